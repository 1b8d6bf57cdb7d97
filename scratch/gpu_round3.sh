#!/bin/bash
# One GPU-box visit (round 3): rocprofv3 kernel stats of the bench command and the PMC passes (each in its own run; program directly after `--`).
# Outputs -> gpurun_out/$1 ; scratch/summarize_profiles3.py turns them into profiles/r03_*.
set -o pipefail
TAG=${1:-r03p}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="python3 $PWD/bench.py --steps 96 --warmup 4 --no-cpu-baseline --no-side"
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- $B > $OUT/prof_bench.json 2> $OUT/prof.err
echo stats done
# the counter passes time one kernel at a time: the serial tick (QRGPU_TICK_PIPELINE=0), so that no launch shares the machine with another
export QRGPU_TICK_PIPELINE=0
# (counter collection runs one kernel at a time: a gate that polls for another launch's progress would wait out its bound -- 50 ms, then the plan is
#  called off and the planned launch's workgroups leave at once -- so the planned launch is forked with an event here, as in rounds 1-2)
export QRGPU_PLANNED_FORK=1
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$i -- python3 $OLDPWD/bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-side > /dev/null 2> $OUT/pmc_$i.err || echo "pmc pass $i failed"
  echo "pmc $i done"
done
find $OUT -name "*counter_collection.csv" | wc -l
