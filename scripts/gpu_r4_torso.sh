#!/bin/bash
# f16mx on the torso path: parity (the default-precision test bodies re-run under R3D_SR_PRECISION=f16mx) + torso frame time per precision
cd "$(dirname "$0")/.." || exit 1
timeout 900 python -m pytest tests/test_gpu_f16x3.py tests/test_gpu_warp_sr.py tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -k "warp or fusion or f16mx or torso or conv2d or policy" 2>&1 | tail -15
for p in f16x3 f16mx; do R3D_SR_PRECISION=$p python scripts/torso_frames.py 40 2>&1 | grep "torso frame" | sed "s/^/$p /"; done
