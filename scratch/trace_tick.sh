#!/bin/bash
# kernel timeline of a few pipelined ticks (rocprofv3 --kernel-trace): start / end of every launch relative to the tick's main pass
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace1
rocprofv3 --kernel-trace --output-format csv -d /tmp/trace1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-side > /tmp/trace1.json 2> /tmp/trace1.err
f=$(find /tmp/trace1 -name "*kernel_trace.csv" | head -1)
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
ev=[(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:48]) for r in rows]
ev.sort()
mains=[i for i,e in enumerate(ev) if "qr_mpc_kernel<2, false, false, 512>" in e[2]]
for m in mains[40:44]:
    t0=ev[m][0]
    print("tick: main pass %.1f us" % ((ev[m][1]-t0)/1e3))
    for e in ev[max(0,m-6):m+8]:
        if e[0] >= t0-60000 and e[0] <= ev[m][1]+80000:
            print("   %-50s start %8.1f end %8.1f (%.1f us)" % (e[2], (e[0]-t0)/1e3, (e[1]-t0)/1e3, (e[1]-e[0])/1e3))
PY
