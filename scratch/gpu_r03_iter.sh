#!/bin/bash
# one kernel iteration on the GPU box: parity tests, phase stamps of the main pass, the headline bench
set -o pipefail
OUT=gpurun_out/${1:-r03i}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
timeout -k 10 200 python scratch/diag_mpc_phases.py 1024 > $OUT/phases.txt 2>&1; echo "phases rc=$?"; head -12 $OUT/phases.txt
timeout -k 10 400 python bench.py --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$OUT/bench.json")); r=d["roofline"]; c=d["config"]
print("value %.3f M  ms/step %.4f  mpc %.4f wbc %.4f outside %.4f  draws %s" % (d["value"]/1e6, d["ms_per_step"], r["kernel_ms"], r["other_kernel_ms"], r["outside_kernels_ms"], [round(x/1e6,2) for x in c["ticks_per_s_per_draw"]]))
print("replayed %.3f M  noK12 %.3f M  pred8 %s" % (c.get("ticks_per_s_same_batch_replayed",0)/1e6, c.get("ticks_per_s_without_k12",0)/1e6, c.get("predicted_weak_scaling_8",{}).get("predicted_speedup_at_8")))
PY
