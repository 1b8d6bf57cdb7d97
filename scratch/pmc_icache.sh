#!/bin/bash
# Instruction cache of the serial tick's kernels: requests, hits, misses, fetches in flight (1024 robots per launch)
set -o pipefail
OUT=$PWD/gpurun_out/icache; ROOT=$PWD; mkdir -p $OUT; export TMPDIR=/tmp
export QRGPU_TICK_PIPELINE=0 QRGPU_LAB=1 QRGPU_PLANNED_FORK=1 QRGPU_BENCH_OVERLAP=0
cd /tmp
i=0
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQC_TC_INST_REQ SQC_TC_STALL" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 500 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -- python3 $ROOT/bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-side > /dev/null 2> $OUT/p$i.err || echo "pass $i failed"
done
cd $ROOT
python3 - <<'PY'
import glob, pandas as pd
for f in sorted(glob.glob('gpurun_out/icache/p*/*/*_counter_collection.csv')):
    d = pd.read_csv(f)
    def key(k):
        if 'qr_mpc_kernel<2, false, false, 512, 0' in k: return 'mpc_main'
        if 'qr_wbc_kernel' in k: return 'wbc'
    d['k'] = d['Kernel_Name'].map(key); d = d[d['k'].notna()]
    g = d.groupby(['k', 'Counter_Name', 'Dispatch_Id'])['Counter_Value'].sum().groupby(level=[0, 1]).mean()
    for (k, c), v in g.items(): print('%-9s %-28s %14.0f' % (k, c, v))
PY
