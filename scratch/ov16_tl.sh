#!/bin/bash
# configs[4] overlapped ticks: the timeline of several populations, to catch one where the planned class runs late
mkdir -p gpurun_out/ov16tl
for d in ${DRAWS:-0 1 5 7}; do
  QRGPU_LIB=$PWD/scratch/ab/tl.so H=16 MIXED=1 K=${K:-16} DRAW=$d timeout -k 5 100 python scratch/diag_overlap.py > gpurun_out/ov16tl/draw$d.log 2>&1 || echo "draw $d failed"
done
