# round 5, call 10: head frame without the per-frame tail fold (R3D_MX_TAIL_FOLD A/B): parity of the f16mx tests + bench
mkdir -p gpurun_out/r5c10; O=gpurun_out/r5c10
timeout 900 python -m pytest tests/test_gpu_mx.py tests/test_gpu_pinned_config.py tests/test_gpu_parity.py tests/test_gpu_coresidency.py -m gpu -q -rP > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed" $O/pytest.log | tail -1
grep -E "benchmarked frame|sr_full f16mx|synthesis golden" $O/pytest.log | cut -c1-200
for tf in 0 1 0 1; do
R3D_MX_TAIL_FOLD=$tf timeout 300 python bench.py --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tail_fold=$tf', d['value'], d['value_single_stream'], d['value_synthesis_api']['value'], d['value_synthesis_api']['other_precisions'], d['clip125_1gpu']['fps'], d['cfg5_stress']['ms_per_batch'])"
done
