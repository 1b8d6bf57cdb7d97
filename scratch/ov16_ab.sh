#!/bin/bash
# configs[4] per GPU (512 A1 + 512 Lite3, h = 16): overlapped ticks on the machine split by CU masks against the plain tick; one box, alternating
mkdir -p gpurun_out/ov16; rm -f gpurun_out/ov16/*
for rep in 1 2; do
  for K in ${SIDES:-64}; do
    QRGPU_LAB=1 QRGPU_OV16_SIDE_CUS=$K timeout -k 10 300 python bench.py --mixed --horizon 16 --steps ${STEPS:-40} --warmup 8 --no-cpu-baseline --no-side $EXTRA > gpurun_out/ov16/on${K}_$rep.json 2> gpurun_out/ov16/on${K}_$rep.err || echo "on $K failed"
  done
  QRGPU_BENCH_OVERLAP=0 timeout -k 10 300 python bench.py --mixed --horizon 16 --steps ${STEPS:-40} --warmup 8 --no-cpu-baseline --no-side $EXTRA > gpurun_out/ov16/off_$rep.json 2> gpurun_out/ov16/off_$rep.err || echo "off failed"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/ov16/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); c = d['config']
        print(f, '%.3f M' % (d['value'] / 1e6), [round(r / 1e6, 2) for r in c['ticks_per_s_per_draw']], c.get('tick_overlap'), c['status_flags_nonzero_per_draw'], 'main %.3f ms' % d['roofline']['kernel_ms'], c.get('max_active_set_changes_per_draw'))
    except Exception as e:
        print(f, 'ERR', e); print(open(f.replace('.json', '.err')).read()[-600:])
PY
