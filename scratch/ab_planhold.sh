#!/bin/bash
# default bench (1024 A1, h = 10, 200 steps = 25 per population): overlapped ticks with the plan hold (31 calls on the plain tick after a lane found a plan) against none
mkdir -p gpurun_out/planhold; rm -f gpurun_out/planhold/*
for rep in 1 2; do
  for H in ${HOLDS:-31 0}; do
    QRGPU_OV_PLAN_HOLD=$H timeout -k 10 300 python bench.py --no-cpu-baseline --no-side > gpurun_out/planhold/hold${H}_$rep.json 2> gpurun_out/planhold/hold${H}_$rep.err || echo "hold $H failed"
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/planhold/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); c = d['config']
        print(f, '%.3f M' % (d['value'] / 1e6), [round(r / 1e6, 2) for r in c['ticks_per_s_per_draw']], c['status_flags_nonzero_per_draw'])
    except Exception as e:
        print(f, 'ERR', e)
PY
