#!/bin/bash
# rocprofv3 kernel stats of bench.py for several library builds on one box: ab_prof.sh name1 name2 ...
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  rm -rf /tmp/prof_$L
  QRGPU_LIB=$GRAFT_REPO_ROOT/scratch/ab/$L.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$L -- python3 $GRAFT_REPO_ROOT/bench.py --steps 96 --warmup 4 --no-cpu-baseline --no-side > /tmp/prof_$L.json 2> /tmp/prof_$L.err
  echo "== $L"
  f=$(find /tmp/prof_$L -name "*kernel_stats.csv" | head -1)
  python3 - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:8]:
    print("%-70s calls %5s avg %9.1f us  min %8.1f max %9.1f  total %6.1f %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3, float(r["Percentage"])))
PY
done
