#!/bin/bash
# One GPU-box visit (round 2): parity tests, smoke, bench lines, rocprofv3 kernel stats and PMC passes (each in its own run).  Outputs -> gpurun_out/$1
set -o pipefail
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit=$?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit=$?" >> $OUT/smoke.log
tail -2 $OUT/smoke.log
timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo bench done
timeout -k 10 600 python bench.py --mixed --horizon 16 > $OUT/bench_config4_f32.json 2> $OUT/bench_config4_f32.err
timeout -k 10 600 python bench.py --mixed --horizon 16 --hessian bf16x3 > $OUT/bench_config4_bf16x3.json 2> $OUT/bench_config4_bf16x3.err
timeout -k 10 600 python bench.py --robots 8192 --no-side > $OUT/bench_8192.json 2> $OUT/bench_8192.err
timeout -k 10 600 python bench.py --robots 256 --mode mpc --no-side > $OUT/bench_config1.json 2> $OUT/bench_config1.err
echo side benches done
B="python3 bench.py --steps 96 --warmup 4 --no-cpu-baseline --no-side"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- $B > $OUT/prof_bench.json 2> $OUT/prof.err
echo stats done
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" "SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$i -- python3 bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-side > /dev/null 2> $OUT/pmc_$i.err || echo "pmc pass $i failed"
  echo "pmc $i done"
done
find $OUT -name "*counter_collection.csv" | wc -l
