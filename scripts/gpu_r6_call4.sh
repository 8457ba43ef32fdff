# round 6: 16-byte epilogue stores: parity of the SR / conv-stack / warp tests, A/B timing against the 8-byte build, bit-identity of the two builds
mkdir -p gpurun_out/r6c4; O=gpurun_out/r6c4
timeout 1200 python -m pytest tests/test_gpu_mx.py tests/test_gpu_parity.py tests/test_gpu_f16x3.py tests/test_gpu_warp_sr.py tests/test_gpu_range_and_sizes.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
for l in hip x_EPI8; do for p in f16mx f16x3; do echo "== $l $p"; R3D_LIB=$PWD/real3dportrait_amd/lib/libr3d_$l.so R3D_SR_PRECISION=$p python scripts/prof_sr.py 20 2>&1 | grep -E "SR 128|digest"; done; done
for l in hip x_EPI8 hip x_EPI8; do echo "== torso $l"; R3D_LIB=$PWD/real3dportrait_amd/lib/libr3d_$l.so python scripts/prof_torso.py 20 2>&1 | tail -1; done
