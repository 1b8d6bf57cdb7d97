#!/bin/bash
# One GPU-box visit (round 4): the bench lines, the overlap A/B, rocprofv3 kernel stats of the bench command and the PMC passes (each in its own run;
# program directly after `--`), the tick timeline, the microbenchmarks of LAB_NOTES A.2, the GPU tests.  Outputs -> gpurun_out/$1 ;
# scratch/summarize_profiles4.py turns them into profiles/r04_*.
set -o pipefail
TAG=${1:-r04p}
OUT=$PWD/gpurun_out/$TAG
ROOT=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
PART=${2:-12}
run() { name=$1; shift; timeout -k 10 500 "$@" > $OUT/$name.json 2> $OUT/$name.err || echo "$name failed"; echo "$name done"; }
if [[ $PART == *1* ]]; then
python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -1 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
run bench python bench.py
run bench_driver_style_steps20_warmup5 python bench.py --steps 20 --warmup 5
QRGPU_BENCH_OVERLAP=0 run bench_no_tick_overlap python bench.py --steps 20 --warmup 5 --no-cpu-baseline
scratch/ov_sweep.sh > $OUT/overlap_ab.txt 2>&1; echo "overlap A/B done"
run bench_8192_robots python bench.py --robots 8192 --no-cpu-baseline --no-side
run bench_config1_256_mpc_only python bench.py --robots 256 --mode mpc
run bench_config4_f32 python bench.py --mixed --horizon 16 --no-side
run bench_config4_bf16x3 python bench.py --mixed --horizon 16 --hessian bf16x3 --no-side
run bench_single_latency python bench.py --mode single
QRGPU_BENCH_FORCE_COMM=1 run bench_forced_comm_one_rank python bench.py --no-cpu-baseline --no-side
QRGPU_BENCH_FORCE_COMM=1 QRGPU_COMM_EVENTS=1 run bench_forced_comm_one_rank_events python bench.py --no-cpu-baseline --no-side
QRGPU_BENCH_REHEARSAL=1 run bench_rehearsal_2ranks_one_gpu python bench.py --gpus 2 --steps 40 --warmup 5 --no-cpu-baseline --no-side
if [ -f scratch/ab/tl.so ]; then
  QRGPU_LIB=$ROOT/scratch/ab/tl.so K=14 timeout -k 5 100 python scratch/diag_overlap.py > $OUT/overlap_timeline.txt 2>&1
  QRGPU_LIB=$ROOT/scratch/ab/tl.so K=14 OV=0 timeout -k 5 100 python scratch/diag_overlap.py >> $OUT/overlap_timeline.txt 2>&1
  echo "timeline done"
fi
for u in cumask cumask2 pipes; do [ -x scratch/ubench/$u ] && timeout -k 5 60 scratch/ubench/$u > $OUT/ubench_$u.txt 2>&1; done; echo "ubench done"
fi
[[ $PART == *2* ]] || exit 0
B="python3 $ROOT/bench.py --steps 96 --warmup 4 --no-cpu-baseline --no-side"
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- $B > $OUT/prof_bench.json 2> $OUT/prof.err
echo stats done
# the counter passes time one kernel at a time: the serial tick (QRGPU_TICK_PIPELINE=0), so that no launch shares the machine with another, and
# the planned launch forked with an event (a laboratory switch: a polled go would wait out its 50 ms bound under a profiler that runs one kernel at a time)
export QRGPU_TICK_PIPELINE=0 QRGPU_LAB=1 QRGPU_PLANNED_FORK=1 QRGPU_BENCH_OVERLAP=0
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$i -- python3 $ROOT/bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-side > /dev/null 2> $OUT/pmc_$i.err || echo "pmc pass $i failed"
  echo "pmc $i done"
done
find $OUT -name "*counter_collection.csv" | wc -l
