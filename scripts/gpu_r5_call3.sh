# round 5, call 3: the fixed tests; big ray shapes at one wave per SIMD (AGPR-held values, no scratch) vs two (scratch): cfg 5 and tri-grids
mkdir -p gpurun_out/r5c3; O=gpurun_out/r5c3
timeout 900 python -m pytest tests/test_gpu_f16x3.py tests/test_gpu_bench_contract.py "tests/test_gpu_range_and_sizes.py::test_conv_stack_range_sweep" -m gpu -q -rP > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
for occ in 2 1; do
  echo "== BIG_OCC $occ"
  R3D_RENDER_BIG_OCC=$occ timeout 300 python scripts/stress_cfg5.py > $O/cfg5_occ$occ.json 2> $O/cfg5_occ$occ.err; cut -c1-700 $O/cfg5_occ$occ.json
  R3D_RENDER_BIG_OCC=$occ timeout 300 python scripts/prof_trigrid.py 2>&1 | tail -2
  R3D_RENDER_BIG_OCC=$occ timeout 600 python -m pytest tests -m gpu -q -k "trigrid or cfg5 or 96" 2>&1 | tail -2
done
