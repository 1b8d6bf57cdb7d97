#!/bin/bash
# round 3: the torch-free bench, the single-robot latency line (zero copy and with copies), the N = 2 rehearsal of the self-launcher
set -o pipefail
OUT=gpurun_out/${1:-r03c}
mkdir -p $OUT
timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --mode single > $OUT/bench_single.json 2> $OUT/bench_single.err; echo "single rc=$?"
QRGPU_SINGLE_COPIES=1 timeout -k 10 300 python bench.py --mode single > $OUT/bench_single_copies.json 2> $OUT/bench_single_copies.err; echo "single copies rc=$?"
QRGPU_BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 2 --steps 16 --warmup 2 --no-side --no-cpu-baseline > $OUT/bench_rehearsal2.json 2> $OUT/bench_rehearsal2.err; echo "rehearsal rc=$?"
tail -c 300 $OUT/bench_rehearsal2.json; tail -3 $OUT/bench_rehearsal2.err
