#!/bin/bash
set -e
mkdir -p gpurun_out/chunks
for N in 4096 8192 5000; do
  QRGPU_WBC_CHUNKS=1 N=$N OUT=gpurun_out/chunks/on_$N.npz timeout -k 10 300 python scratch/check_chunks.py
  QRGPU_WBC_CHUNKS=0 N=$N OUT=gpurun_out/chunks/off_$N.npz timeout -k 10 300 python scratch/check_chunks.py
  python - <<PY
import numpy as np
a = np.load("gpurun_out/chunks/on_$N.npz"); b = np.load("gpurun_out/chunks/off_$N.npz")
same = all(np.array_equal(a[k], b[k], equal_nan=True) for k in a.files)
print("N = $N: chunked against one launch, every output of every tick bit for bit:", same)
assert same
PY
done
for rep in 1 2; do
  for C in 1 0; do
    QRGPU_WBC_CHUNKS=$C QRGPU_BENCH_OVERLAP=0 timeout -k 10 300 python bench.py --robots 8192 --no-cpu-baseline --no-side > gpurun_out/chunks/bench_$C.$rep.json 2>/dev/null
    python - <<PY
import json
d = json.load(open("gpurun_out/chunks/bench_$C.$rep.json")); print("8192 robots, QRGPU_WBC_CHUNKS=$C rep $rep: %.3f M ticks/s, %.4f ms per step" % (d["value"] / 1e6, d["ms_per_step"]), [round(x / 1e6, 2) for x in d["config"]["ticks_per_s_per_draw"]])
PY
  done
done
