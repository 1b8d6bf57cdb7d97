#!/bin/bash
# A/B of library builds on ONE box (boxes differ by several per cent): scratch/ab/<name>.so, alternated.  usage: ab_bench.sh "<bench args>" name1 name2 ...
ARGS=$1; shift
mkdir -p gpurun_out/ab
for rep in 1 2 3; do
  for L in "$@"; do
    QRGPU_LIB=$PWD/scratch/ab/$L.so timeout -k 10 300 python bench.py --no-cpu-baseline --no-side $ARGS > gpurun_out/ab/$L.$rep.json 2> gpurun_out/ab/$L.$rep.err || echo "$L failed"
    python - <<PY
import json
d=json.load(open("gpurun_out/ab/$L.$rep.json")); r=d["roofline"]
print("%-14s rep $rep  value %.3f M  mpc %.4f ms  wbc %.4f ms  outside %.4f  min %.2f max %.2f" % ("$L", d["value"]/1e6, r["kernel_ms"], r["other_kernel_ms"], r.get("outside_kernels_ms") or 0, d["config"]["ticks_per_s_min"]/1e6, d["config"]["ticks_per_s_max"]/1e6))
PY
  done
done
