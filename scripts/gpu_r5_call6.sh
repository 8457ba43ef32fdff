# round 5, call 6: NOPK product (as_rounded splits): full -m gpu suite, bench line, PK partner bench for the price
mkdir -p gpurun_out/r5c6; O=gpurun_out/r5c6
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_full.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_full.log
for i in 1 2; do
timeout 300 python bench.py --no-extras --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('nopk', d['value'], d['value_single_stream'], d['roofline']['avg_launch_ms'])"
R3D_LIB=$PWD/tests/_build/libr3d_hip_pk.so timeout 300 python bench.py --no-extras --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pk  ', d['value'], d['value_single_stream'], d['roofline']['avg_launch_ms'])"
done
for lib in real3dportrait_amd/lib/libr3d_hip.so tests/_build/libr3d_hip_pk.so; do R3D_LIB=$PWD/$lib timeout 300 python scripts/prof_trigrid.py 2>&1 | tail -2; R3D_LIB=$PWD/$lib timeout 300 python scripts/stress_cfg5.py 2>/dev/null | cut -c90-330; done
