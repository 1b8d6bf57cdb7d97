#!/bin/bash
# A/B of environment settings on one box: ab_env.sh "<bench args>" "VAR=a" "VAR=b" ...
ARGS=$1; shift
mkdir -p gpurun_out/abenv
for rep in 1 2; do
  i=0
  for E in "$@"; do
    i=$((i+1))
    env $E timeout -k 10 300 python bench.py --no-cpu-baseline --no-side $ARGS > gpurun_out/abenv/$i.$rep.json 2> gpurun_out/abenv/$i.$rep.err || echo "$E failed"
    python - <<PY
import json
d=json.load(open("gpurun_out/abenv/$i.$rep.json")); r=d["roofline"]
print("%-28s rep $rep  value %.3f M  ms/step %.4f  mpc %.4f  wbc %.4f  min %.2f max %.2f flags %d" % ("$E", d["value"]/1e6, d["ms_per_step"], r["kernel_ms"], r["other_kernel_ms"], d["config"]["ticks_per_s_min"]/1e6, d["config"]["ticks_per_s_max"]/1e6, sum(d["config"]["status_flags_nonzero_per_draw"])))
PY
  done
done
