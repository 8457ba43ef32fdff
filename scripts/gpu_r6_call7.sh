mkdir -p gpurun_out/r6c6
( timeout 900 python -m pytest tests/test_gpu_blend_conv.py tests/test_gpu_warp_sr.py -x -q -m gpu 2>&1 | tail -3 )
for f in 1 0; do for p in f16mx f16x3; do
    echo "R3D_FUSE_BLEND=$f $p: $(R3D_FUSE_BLEND=$f R3D_SR_PRECISION=$p timeout 300 python scripts/prof_torso.py 200 2>&1 | tail -1)"
done; done
bash scripts/gpu_r6_torso_kstats.sh > /dev/null 2>&1; grep -E "blend|conv1x1|sum of" gpurun_out/r6c6/torso_kernels_1.txt
