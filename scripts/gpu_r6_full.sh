# round 6: the full -m gpu suite (measurements printed), then the default bench line
mkdir -p gpurun_out/r6full; O=gpurun_out/r6full
timeout 1500 python -m pytest tests -m gpu -q -rP > $O/pytest_full.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest_full.log
grep -E "heavy tail|dense heavy|f16mx|synthesis golden|FAILED|Error" $O/pytest_full.log | head -150 > $O/pytest_lines.txt
timeout 900 python bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc $?"; tail -2 $O/bench.err; cut -c1-600 $O/bench.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6full/bench.log"))
r = d["roofline"]
print({k: d.get(k) for k in ("value", "ms_per_step", "value_single_stream")})
print("alt_f16x3", d.get("alt_f16x3", {}).get("value"), "api", r.get("value_synthesis_api"), "median5", d["config"].get("value_median_of_5"))
print("breakdown", r.get("breakdown_ms_per_frame"))
print("roofline", {k: r[k] for k in ("kernel", "achieved", "frac", "traffic", "avg_launch_ms") if k in r})
print("torso", d.get("torso_frame", {}).get("fps"), d.get("torso_frame", {}).get("fps_3_streams"), "cfg5", d.get("cfg5_stress", {}).get("ms_per_batch"), d.get("cfg5_stress", {}).get("fps"))
PY
