#!/bin/bash
# Does what a launch fetches beyond its algorithmic bytes scale with the batch?  FETCH_SIZE of the serial tick's kernels at 1024 and 8192 robots per launch.
set -o pipefail
OUT=$PWD/gpurun_out/fetchscale; ROOT=$PWD; mkdir -p $OUT; export TMPDIR=/tmp
export QRGPU_TICK_PIPELINE=0 QRGPU_LAB=1 QRGPU_PLANNED_FORK=1 QRGPU_BENCH_OVERLAP=0
cd /tmp
for N in 1024 8192; do
  timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/n$N -- python3 $ROOT/bench.py --robots $N --steps 16 --warmup 2 --no-cpu-baseline --no-side > /dev/null 2> $OUT/n$N.err || echo "pass $N failed"
done
cd $ROOT
python3 - <<'PY'
import glob, pandas as pd
for N in (1024, 8192):
    f = glob.glob('gpurun_out/fetchscale/n%d/*/*_counter_collection.csv' % N)[0]
    d = pd.read_csv(f)
    def key(k):
        if 'qr_mpc_kernel<2, false, false, 512, 0' in k: return 'mpc_main'
        if 'qr_wbc_kernel' in k: return 'wbc'
    d['k'] = d['Kernel_Name'].map(key); d = d[d['k'].notna()]
    g = d.groupby(['k', 'Dispatch_Id'])['Counter_Value'].sum().groupby(level=0).mean()
    for k, v in g.items(): print('%5d robots  %-9s FETCH_SIZE %9.1f KiB -> read %.3f MB per launch (2 x 64 B per request)' % (N, k, v, 2 * v * 1024 / 1e6))
PY
