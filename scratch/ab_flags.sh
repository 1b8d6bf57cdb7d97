#!/bin/bash
# compiler scheduling options against the default build (each: rebuild + one bench line)
mkdir -p gpurun_out/ab_flags
while read -r f; do
  export QRGPU_EXTRA_FLAGS="$f"
  if ! python -c "import __graft_entry__ as g; g.build()" > gpurun_out/ab_flags/build.log 2>&1; then echo "flags [$f]: build failed: $(grep -m1 -E 'error|unknown' gpurun_out/ab_flags/build.log)" | tee -a gpurun_out/ab_flags/out.txt; continue; fi
  for i in 1 2; do
  timeout -k 10 300 python bench.py --no-side --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('flags [$f]: %.3f M, mpc %.4f ms, wbc %.4f ms' % (d['value'] / 1e6, d['roofline']['kernel_ms'], d['roofline']['other_kernel_ms']))" | tee -a gpurun_out/ab_flags/out.txt || exit 1
  done
done <<'F'

-mllvm -amdgpu-sched-strategy=max-ilp
-mllvm -amdgpu-sched-strategy=iterative-ilp
-mllvm -amdgpu-sched-strategy=max-memory-clause
F
