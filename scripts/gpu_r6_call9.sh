#!/bin/bash
# round 6, call 9: torso_encoder writes its half of the head / torso concatenation (r3d_conv_forward_cat): tests, warp goldens, torso frame A/B
( timeout 1200 python -m pytest tests/test_gpu_blend_conv.py tests/test_gpu_warp_sr.py tests/test_gpu_parity.py tests/test_gpu_range_and_sizes.py -x -q -m gpu 2>&1 | grep -E "conv into|passed|failed|Error|assert" | tail -12 )
for f in 1 0; do for p in f16mx f16x3; do
    echo "R3D_FUSE_TORSO_CAT=$f $p: $(R3D_FUSE_TORSO_CAT=$f R3D_SR_PRECISION=$p timeout 300 python scripts/prof_torso.py 200 2>&1 | tail -1)"
done; done
