#!/bin/bash
# overlapped tick against the pipelined tick without overlap; one box, alternating
mkdir -p gpurun_out/ov; rm -f gpurun_out/ov/*
for rep in 1 2; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side > gpurun_out/ov/on$rep.json 2> gpurun_out/ov/on$rep.err || echo "on failed"
  QRGPU_BENCH_OVERLAP=0 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side > gpurun_out/ov/off$rep.json 2> gpurun_out/ov/off$rep.err || echo "off failed"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/ov/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); c = d['config']
        print(f, '%.3f M' % (d['value'] / 1e6), [round(r / 1e6, 2) for r in c['ticks_per_s_per_draw']], c.get('tick_overlap'), c['status_flags_nonzero_per_draw'], 'main %.3f ms' % d['roofline']['kernel_ms'])
    except Exception as e:
        print(f, 'ERR', e)
PY
