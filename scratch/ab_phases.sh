#!/bin/bash
# per-phase cycle stamps (instrumented kernels) of several library builds on one box: ab_phases.sh name1 name2 ...
for rep in 1 2; do
for L in "$@"; do
  echo "== $L (rep $rep)"
  QRGPU_LIB=$PWD/scratch/ab/$L.so timeout -k 10 200 python scratch/diag_mpc_phases.py 1024 2>&1 | sed -n 2,9p
done
done
