#!/bin/bash
# configs[4] per GPU, the default 200 steps (25 per population): overlapped ticks of the laboratory against the plain tick, alternating on one box
mkdir -p gpurun_out/ov16b; rm -f gpurun_out/ov16b/*
for rep in 1 2; do
  QRGPU_LAB=1 QRGPU_OV16_COST=${COST:-1} timeout -k 10 400 python bench.py --mixed --horizon 16 --no-cpu-baseline --no-side > gpurun_out/ov16b/on_$rep.json 2> gpurun_out/ov16b/on_$rep.err || echo "on failed"
  QRGPU_BENCH_OVERLAP=0 timeout -k 10 400 python bench.py --mixed --horizon 16 --no-cpu-baseline --no-side > gpurun_out/ov16b/off_$rep.json 2> gpurun_out/ov16b/off_$rep.err || echo "off failed"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/ov16b/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); c = d['config']
        print(f, '%.3f M' % (d['value'] / 1e6), [round(r / 1e6, 2) for r in c['ticks_per_s_per_draw']], c.get('tick_overlap'), c['status_flags_nonzero_per_draw'], 'pooled %.3f' % (c.get('ticks_per_s_all_steps', 0) / 1e6))
    except Exception as e:
        print(f, 'ERR', e); print(open(f.replace('.json', '.err')).read()[-600:])
PY
