# round 5, call 14: the REF ray shape with ALL colours in the workspace -- at 2 waves/SIMD (230 VGPRs, 45 KB of LDS) and at 3 waves/SIMD (168 VGPRs + 60 spilled, 768 blocks)
for lib in real3dportrait_amd/lib/libr3d_hip.so tests/_build/libr3d_gpall.so tests/_build/libr3d_occ3.so real3dportrait_amd/lib/libr3d_hip.so tests/_build/libr3d_occ3.so; do
  echo "== $lib"
  R3D_LIB=$PWD/$lib timeout 300 python scripts/prof_trigrid.py 2>&1 | tail -2 | head -1
  R3D_LIB=$PWD/$lib timeout 300 python bench.py --no-extras --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   value', d['value'], 'median', d['repeats']['median'], 'one stream', d['value_single_stream'])"
done
R3D_LIB=$PWD/tests/_build/libr3d_occ3.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pinned_config.py -m gpu -q -k "render or synthesis or benchmarked or clip" 2>&1 | tail -2
