#!/bin/bash
# round 3: every bench line quoted in DESIGN.md 5, one visit
set -o pipefail
OUT=gpurun_out/${1:-r03b}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit=$?" >> $OUT/pytest_gpu.log; tail -2 $OUT/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit=$?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo bench $?
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_style.json 2> $OUT/bench_driver_style.err; echo driver-style $?
QRGPU_TICK_PIPELINE=0 timeout -k 10 600 python bench.py --no-side > $OUT/bench_serial_tick.json 2> $OUT/bench_serial_tick.err; echo serial $?
timeout -k 10 600 python bench.py --mode single > $OUT/bench_single.json 2> $OUT/bench_single.err; echo single $?
timeout -k 10 600 python bench.py --mixed --horizon 16 > $OUT/bench_config4_f32.json 2> $OUT/bench_config4_f32.err; echo cfg4 $?
timeout -k 10 600 python bench.py --mixed --horizon 16 --hessian bf16x3 --no-side > $OUT/bench_config4_bf16x3.json 2> $OUT/bench_config4_bf16x3.err; echo cfg4bf16 $?
timeout -k 10 600 python bench.py --robots 8192 --no-side > $OUT/bench_8192.json 2> $OUT/bench_8192.err; echo 8192 $?
timeout -k 10 600 python bench.py --robots 256 --mode mpc --no-side > $OUT/bench_config1.json 2> $OUT/bench_config1.err; echo cfg1 $?
QRGPU_BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 2 --steps 16 --warmup 2 --no-side --no-cpu-baseline > $OUT/bench_rehearsal2.json 2> $OUT/bench_rehearsal2.err; echo rehearsal $?
