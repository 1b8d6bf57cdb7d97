#!/bin/bash
# configs[4] per GPU, default run, overlapped ticks: lingering workgroups of the planned launch (laboratory switch), alternating on one box
mkdir -p gpurun_out/ov16s; rm -f gpurun_out/ov16s/*
for rep in 1 2; do
  for L in ${LINGERS:-4 8 16}; do
    QRGPU_LAB=1 QRGPU_OV16_LINGER=$L timeout -k 10 400 python bench.py --mixed --horizon 16 --no-cpu-baseline --no-side > gpurun_out/ov16s/linger${L}_$rep.json 2> gpurun_out/ov16s/linger${L}_$rep.err || echo "linger $L failed"
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/ov16s/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); c = d['config']
        print(f, '%.3f M' % (d['value'] / 1e6), [round(r / 1e6, 2) for r in c['ticks_per_s_per_draw']], c['status_flags_nonzero_per_draw'])
    except Exception as e:
        print(f, 'ERR', e)
PY
