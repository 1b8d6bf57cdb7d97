#!/bin/bash
# default bench (1024 A1, h = 10): overlapped against plain ticks, alternating on one box, the driver's step counts
mkdir -p gpurun_out/absmall; rm -f gpurun_out/absmall/*
for rep in 1 2 3; do
  timeout -k 10 300 python bench.py --steps ${STEPS:-200} --warmup 10 --no-cpu-baseline --no-side > gpurun_out/absmall/on_$rep.json 2> gpurun_out/absmall/on_$rep.err || echo "on failed"
  QRGPU_BENCH_OVERLAP=0 timeout -k 10 300 python bench.py --steps ${STEPS:-200} --warmup 10 --no-cpu-baseline --no-side > gpurun_out/absmall/off_$rep.json 2> gpurun_out/absmall/off_$rep.err || echo "off failed"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/absmall/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); c = d['config']
        print(f, '%.3f M' % (d['value'] / 1e6), [round(r / 1e6, 2) for r in c['ticks_per_s_per_draw']], c.get('tick_overlap'), c['status_flags_nonzero_per_draw'], 'main %.3f ms' % d['roofline']['kernel_ms'])
    except Exception as e:
        print(f, 'ERR', e); print(open(f.replace('.json', '.err')).read()[-600:])
PY
