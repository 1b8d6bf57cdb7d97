#!/bin/bash
# (record of a measurement: the QRGPU_BIG_COST switch it drives was removed again with the experiment -- DESIGN.md 8)
# planned list extended to last tick's longest solves: QRGPU_BIG_COST in 4096-cycle ticks (0 = off)
mkdir -p gpurun_out/ab_bigcost
for bc in 0 70 62 56 50; do
  QRGPU_BIG_COST=$bc timeout -k 10 300 python bench.py --no-side --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('big_cost $bc: %.3f M ticks/s (min %.3f max %.3f), mpc %.4f ms' % (d['value'] / 1e6, d['config']['ticks_per_s_min'] / 1e6, d['config']['ticks_per_s_max'] / 1e6, d['roofline']['kernel_ms']))" | tee -a gpurun_out/ab_bigcost/out.txt || exit 1
done
