#!/bin/bash
# round 6: the ray kernel with layer 2 (and layer 1) of the decoder on the f32 MFMA (experiment builds real3dportrait_amd/lib/libr3d_x_*.so): render parity tests on the
# experiment library, then kernel time next to the product's on this box (three interleaved repeats)
for lib in "$@"; do
  echo "== parity on $lib"; R3D_LIB=$PWD/$lib timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pinned_config.py -x -q -m gpu -k "render or run_model or trigrid or raygen or synthesis or oracle or frame" 2>&1 | tail -3
done
for rep in 1 2 3; do python scripts/gpu_ray_ab.py default "$@"; done
