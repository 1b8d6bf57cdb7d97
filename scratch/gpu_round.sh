#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprofv3 kernel stats and PMC passes.  Outputs -> gpurun_out/$1
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit=$?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit=$?" >> $OUT/smoke.log
tail -2 $OUT/smoke.log
timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.err
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.err
timeout -k 10 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/pmc_sq.err
find $OUT -name "*.csv" | head -30
