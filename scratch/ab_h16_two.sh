#!/bin/bash
# QRGPU_H16_TWO off / on over batch sizes and the two Hessian forms (mixed h = 16 shard)
for A in ${SIZES:-"1024" "256" "512" "2048" "128"}; do
  echo "== --robots $A"
  bash scratch/ab_env.sh "--steps 64 --warmup 8 --mixed --horizon 16 --robots $A" QRGPU_H16_TWO=0 QRGPU_H16_TWO=1 | grep "rep 2"
done
