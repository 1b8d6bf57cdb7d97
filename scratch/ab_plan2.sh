#!/bin/bash
# planned list launch: workgroups beyond the host's (stale) count of the list
mkdir -p gpurun_out/ab_plan
for x in 0 1 2; do
  for steps in 200 200 20 20; do
    w=20; [ $steps = 20 ] && w=5
    QRGPU_PLANNED_EXTRA=$x timeout -k 10 300 python bench.py --steps $steps --warmup $w --no-side --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('extra $x steps $steps: %.3f M' % (d['value'] / 1e6), [round(x / 1e6, 2) for x in d['config']['ticks_per_s_per_draw']])" | tee -a gpurun_out/ab_plan/out2.txt || exit 1
  done
done
