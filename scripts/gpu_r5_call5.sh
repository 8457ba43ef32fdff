mkdir -p gpurun_out/r5c5; O=gpurun_out/r5c5
for lib in real3dportrait_amd/lib/libr3d_hip.so tests/_build/libr3d_exp.so tests/_build/libr3d_hip_pk.so; do
  echo "== $lib"
  R3D_LIB=$PWD/$lib timeout 600 python -m pytest tests/test_gpu_range_and_sizes.py tests/test_gpu_pinned_config.py -m gpu -q -rP -k "decoder" 2>&1 | grep -E "\* 2\^|passed|failed|planes" | cut -c1-150
done
