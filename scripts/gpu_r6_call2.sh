# round 6: Winograd kernel iteration: SR parity (quick subset), phase stamps, per-kernel stats
mkdir -p gpurun_out/r6c2; O=gpurun_out/r6c2
R3D_CONV_WINO=1 timeout 900 python -m pytest tests/test_gpu_mx.py tests/test_gpu_parity.py tests/test_gpu_f16x3.py -m gpu -q -x > $O/pytest_sr.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_sr.log
for p in f16mx f16x3; do R3D_LIB=$PWD/real3dportrait_amd/lib/libr3d_hip_stamps.so R3D_CONV_WINO=1 python scripts/gpu_wino_stamps.py $p; done 2>&1 | tee $O/stamps.txt
bash scripts/gpu_r6_kstats.sh "1:f16mx 1:f16x3" 2>&1 | grep -E "==|conv_|upconv" | tee $O/kstats.txt
