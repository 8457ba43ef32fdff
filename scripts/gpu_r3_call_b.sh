#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
timeout 2400 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -E "passed|failed|rror|k=|\* 2\^|r512|sr_resize|raw stub|pattern|SR of|NOTE|assert" | tail -70
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_b.json; python - <<'PY'
import json; d=json.load(open('gpurun_out/bench_b.json')); print({k:d.get(k) for k in ('value','value_single_stream','value_cold_start','breakdown_ms_per_frame')}); print(d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline'].get('upconv_fir_f16x3_kernel')); print(d.get('torso_frame',{}).get('fps'), d.get('alt_neural_render_512'))
PY
