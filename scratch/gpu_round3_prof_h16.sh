#!/bin/bash
# rocprofv3 kernel stats of the configs[4] bench command (h = 16, mixed, two workgroups per CU); program directly after `--`
set -o pipefail
OUT=$PWD/gpurun_out/${1:-r03ph16}
mkdir -p $OUT
export TMPDIR=/tmp
B="python3 $PWD/bench.py --mixed --horizon 16 --steps 96 --warmup 4 --no-cpu-baseline --no-side"
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- $B > $OUT/prof_bench.json 2> $OUT/prof.err
echo stats $?
find $OUT -name "*kernel_stats.csv" | head -2
