#!/bin/bash
# (record of a measurement: needs scratch/patches/r02_block_downdate.patch applied -- DESIGN.md 8)
# block downdate (a block drop followed by a sweep of S^-1 over the leaving rows) against the from-scratch solve of what stays
mkdir -p gpurun_out/ab_downdate
for nd in 1 0 1 0; do
  QRGPU_NO_BLOCK_DOWNDATE=$nd timeout -k 10 300 python bench.py --no-side --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('h10 no_downdate $nd: %.3f M, mpc %.4f ms' % (d['value'] / 1e6, d['roofline']['kernel_ms']), [round(x / 1e6, 2) for x in d['config']['ticks_per_s_per_draw']])" | tee -a gpurun_out/ab_downdate/out.txt || exit 1
done
for nd in 1 0 1 0; do
  QRGPU_NO_BLOCK_DOWNDATE=$nd timeout -k 10 300 python bench.py --no-side --no-cpu-baseline --mixed --horizon 16 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('h16 no_downdate $nd: %.3f M, mpc %.4f ms' % (d['value'] / 1e6, d['roofline']['kernel_ms']), [round(x / 1e6, 2) for x in d['config']['ticks_per_s_per_draw']], d['config']['max_active_set_changes_per_draw'])" | tee -a gpurun_out/ab_downdate/out.txt || exit 1
done
