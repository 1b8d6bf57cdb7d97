#!/bin/bash
# planned list: first-call syncs x big-class margin, default (200-step) and driver-length (20-step) runs
mkdir -p gpurun_out/ab_plan
for cfg in "0 6" "2 6" "2 3" "2 1" "0 1"; do
  set -- $cfg
  for steps in 200 200 20; do
    w=20; [ $steps = 20 ] && w=5
    QRGPU_PLAN_SYNC=$1 QRGPU_BIG_MARGIN=$2 timeout -k 10 300 python bench.py --steps $steps --warmup $w --no-side --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('sync $1 margin $2 steps $steps: %.3f M' % (d['value'] / 1e6), [round(x / 1e6, 2) for x in d['config']['ticks_per_s_per_draw']])" | tee -a gpurun_out/ab_plan/out.txt || exit 1
  done
done
