# round 5, call 12: big ray shapes with ALL colours parked in the workspace (no scratch in <4,4> / <6,6>): parity + config 5 / tri-grid times
timeout 900 python -m pytest tests -m gpu -q -k "trigrid or cfg5 or 96 or render or fuzz or heavy_tailed_planes or run_model" 2>&1 | tail -2
timeout 300 python scripts/stress_cfg5.py 2>/dev/null | cut -c90-330
timeout 300 python scripts/prof_trigrid.py 2>&1 | tail -2
R3D_SR_PRECISION=f16mx timeout 600 python scripts/fuzz_parity.py 61 16 2>&1 | tail -1
