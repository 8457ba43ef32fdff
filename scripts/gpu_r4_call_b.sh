#!/bin/bash
# round 4, call B: whole GPU suite + the bench line (after the ray-kernel pipeline and the MX up-conv)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r4b
timeout 1500 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -8 > gpurun_out/r4b/tests.log; cat gpurun_out/r4b/tests.log
timeout 600 python bench.py > gpurun_out/r4b/bench.json 2> gpurun_out/r4b/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4b/bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print('value', d['value'], d['repeats']['fps'], 'single', d.get('value_single_stream'), 'api', d.get('value_synthesis_api',{}).get('value'))
print('conv', r['avg_launch_ms'], r['frac'], 'traffic', r['traffic'], (r['traffic_source'] or '')[:60]); print('upconv', r.get('upconv_fir_f16x3_kernel'))
print('breakdown', d.get('breakdown_ms_per_frame')); print('torso', {k: d['torso_frame'][k] for k in ('fps','fps_3_streams','breakdown_ms_per_frame')})
print('cfg5', d['cfg5_stress']['fps'], d['cfg5_stress']['breakdown_ms_per_batch']); print('alt', d.get('alt_f16x3',{}).get('value')); print('cpu', d.get('cpu_baseline',{}).get('value'))
PY
tail -3 gpurun_out/r4b/bench.err
