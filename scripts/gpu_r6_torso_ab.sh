#!/bin/bash
# round 6: torso frame A/B of the two fusions on one box + per-kernel list
for f in "1 1" "0 0" "1 0" "0 1"; do set -- $f; for p in f16mx f16x3; do
    echo "R3D_FUSE_TORSO_CAT=$1 R3D_FUSE_BLEND=$2 $p: $(R3D_FUSE_TORSO_CAT=$1 R3D_FUSE_BLEND=$2 R3D_SR_PRECISION=$p timeout 300 python scripts/prof_torso.py 200 2>&1 | tail -1)"
done; done
bash scripts/gpu_r6_torso_kstats.sh > /dev/null 2>&1; cat gpurun_out/r6c6/torso_kernels_1.txt
