#!/bin/bash
# round 4, call A: the new parity pins (hash noise vs oracle, heavy tails, config 5 at full size, precision policy) + this round's baseline bench line
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r4a
timeout 900 python -m pytest tests/test_gpu_pinned_config.py tests/test_gpu_f16x3.py -m gpu -q -s -x --no-header -p no:cacheprovider 2>&1 | grep -E "heavy tail|benchmarked frame|cfg5 full|decoder, planes|render:|passed|failed|FAILED|Error|error|assert" | tail -120 > gpurun_out/r4a/tests.log
tail -60 gpurun_out/r4a/tests.log
timeout 600 python bench.py > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4a/bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print('value', d['value'], d['repeats']['fps'], 'single', d.get('value_single_stream'), 'api', d.get('value_synthesis_api',{}).get('value'))
print('conv', r['avg_launch_ms'], r['frac'], 'traffic', r['traffic'], r['traffic_source'][:60]); print('upconv', r.get('upconv_fir_f16x3_kernel'))
print('breakdown', d.get('breakdown_ms_per_frame')); print('torso', {k: d['torso_frame'][k] for k in ('fps','fps_3_streams','breakdown_ms_per_frame')})
print('cfg5', d['cfg5_stress']['fps'], d['cfg5_stress']['breakdown_ms_per_batch']); print('alt', d.get('alt_f16x3',{}).get('value')); print('cpu', d.get('cpu_baseline'))
PY
tail -5 gpurun_out/r4a/bench.err
