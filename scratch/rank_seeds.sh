#!/bin/bash
# each rank's synthetic batch of the 8-GPU run, timed one after the other on this one GPU
for r in 0 1 2 3 4 5 6 7; do
  QRGPU_BENCH_SEED_RANK=$r timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/rk.json
  python - <<PY
import json
d = json.load(open("/tmp/rk.json"))
print("rank-seed $r", round(d["value"]), round(d["ms_per_step"], 4), round(d["roofline"]["kernel_ms"], 4), round(d["roofline"]["other_kernel_ms"], 4), d["config"]["mean_active_set_iterations"], d["config"]["status_flags_nonzero"])
PY
done
