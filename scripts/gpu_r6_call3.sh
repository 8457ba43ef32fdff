# round 6: Winograd kernel iteration: SR parity (quick subset) + SR-only timing, wino vs direct, both precisions
mkdir -p gpurun_out/r6c3; O=gpurun_out/r6c3
R3D_CONV_WINO=1 timeout 900 python -m pytest tests/test_gpu_mx.py tests/test_gpu_parity.py tests/test_gpu_f16x3.py -m gpu -q -x > $O/pytest_sr.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_sr.log
for p in f16mx f16x3; do for w in 1 0; do echo "== wino=$w $p"; R3D_CONV_WINO=$w R3D_SR_PRECISION=$p python scripts/prof_sr.py 20 2>&1 | grep "SR 128"; done; done
