# round 6: the Winograd kernel behind the generic 3x3 conv layer: conv / stack / warp tests under both precisions, torso frame timing
mkdir -p gpurun_out/r6c5; O=gpurun_out/r6c5
timeout 1500 python -m pytest tests/test_gpu_f16x3.py tests/test_gpu_range_and_sizes.py tests/test_gpu_warp_sr.py tests/test_gpu_mx.py tests/test_gpu_pinned_config.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
for p in f16mx f16x3; do for w in 3 0; do echo "== torso $p wino=$w"; R3D_CONV_WINO=$w R3D_SR_PRECISION=$p python scripts/prof_torso.py 20 2>&1 | tail -1; done; done
