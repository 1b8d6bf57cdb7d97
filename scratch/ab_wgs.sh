#!/bin/bash
# A/B of the h <= 11 main pass: threads per workgroup x workgroups per CU, at 1024 and 8192 robots (bench lines to gpurun_out/ab_wgs/)
mkdir -p gpurun_out/ab_wgs
for cfg in "512 2" "256 2" "256 3" "512 3"; do
  set -- $cfg
  for nr in 1024 8192; do
    QRGPU_MAIN_THREADS=$1 QRGPU_MAIN_WGS=$2 timeout -k 10 300 python bench.py --no-side --robots $nr --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('threads $1 wgs $2 robots $nr: %.3f M ticks/s, mpc %.4f ms' % (d['value'] / 1e6, d['roofline']['kernel_ms']))" | tee -a gpurun_out/ab_wgs/out.txt || exit 1
  done
done
