#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
timeout 1500 python -m pytest tests/test_gpu_warp_sr.py tests/test_gpu_bench_contract.py -m gpu -x -q -s 2>&1 | grep -E "passed|failed|rror|sample|assert" | tail -20
timeout 900 python bench.py 2>gpurun_out/bench_c.err | tail -1 > gpurun_out/bench_c.json; tail -3 gpurun_out/bench_c.err
python - <<'PY'
import json; d=json.load(open('gpurun_out/bench_c.json'))
for k in ('value','repeats','value_single_stream','value_cold_start','breakdown_ms_per_frame','value_synthesis_api','clip125_1gpu','cfg5_stress','cpu_baseline','cpu_baseline_reference','opt_in_f16mx'):
    print(k, json.dumps(d.get(k))[:700])
print('torso', d.get('torso_frame',{}).get('fps'), d.get('torso_frame',{}).get('fps_3_streams'), d.get('torso_frame',{}).get('breakdown_ms_per_frame'))
PY
