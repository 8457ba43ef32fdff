#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > gpurun_out/bench_d.json
python - <<'PY'
import json; d=json.load(open('gpurun_out/bench_d.json')); r=d['roofline']
print('value', d['value'], d['repeats']['fps'], 'single', d.get('value_single_stream')); print('conv', r['avg_launch_ms'], r['frac'], 'upconv', r.get('upconv_fir_f16x3_kernel'))
PY
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_range_and_sizes.py -m gpu -q -k "sr or SR or block or conv or synthesis" 2>&1 | tail -3
echo "== whole suite with R3D_SR_PRECISION=f16mx"
R3D_SR_PRECISION=f16mx timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED" | tail -30
