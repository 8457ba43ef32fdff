# round 5, call 4: the product built WITHOUT packed-f32 instructions (+ split8_product, depths via LDS): full -m gpu suite, bench, cfg 5, tri-grid
mkdir -p gpurun_out/r5c4; O=gpurun_out/r5c4
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_full.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_full.log
timeout 600 python bench.py --no-cpu-baseline --no-traffic > $O/bench.log 2> $O/bench.err; echo "bench rc $?"; tail -2 $O/bench.err; cut -c1-300 $O/bench.log
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5c4/bench.log") if l.startswith("{")][0])
for k in ("value", "value_single_stream", "breakdown_ms_per_frame", "alt_f16x3", "value_synthesis_api", "clip125_1gpu"):
    print(k, json.dumps(d.get(k))[:400])
print("cfg5", json.dumps(d.get("cfg5_stress"))[:500])
print("torso", json.dumps(d.get("torso_frame"))[:600])
PY
timeout 300 python scripts/prof_trigrid.py 2>&1 | tail -2
R3D_LIB=$PWD/tests/_build/libr3d_hip_pk.so timeout 300 python scripts/prof_trigrid.py 2>&1 | tail -2
R3D_LIB=$PWD/tests/_build/libr3d_hip_pk.so timeout 300 python bench.py --no-extras --no-cpu-baseline --no-traffic 2>/dev/null | cut -c1-200
timeout 300 python bench.py --no-extras --no-cpu-baseline --no-traffic 2>/dev/null | cut -c1-200
