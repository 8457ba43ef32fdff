# round 5, call 1: the e5m2 activation records on the GPU (full -m gpu suite with measurements printed), the bench line, NOPK price, depth-3 experiment
mkdir -p gpurun_out/r5c1; O=gpurun_out/r5c1
timeout 1200 python -m pytest tests -m gpu -q -rP > $O/pytest_full.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest_full.log
grep -E "heavy tail|f16mx|synthesis golden|FAILED|Error" $O/pytest_full.log | head -150 > $O/pytest_lines.txt
timeout 600 python bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc $?"; tail -2 $O/bench.err; cat $O/bench.log | cut -c1-1500
R3D_LIB=$PWD/tests/_build/libr3d_hip_nopk.so timeout 300 python bench.py --no-extras --no-cpu-baseline --no-traffic > $O/bench_nopk.log 2> $O/bench_nopk.err; echo "nopk rc $?"; cut -c1-400 $O/bench_nopk.log
timeout 300 python bench.py --no-extras --no-cpu-baseline --no-traffic > $O/bench_pk.log 2> $O/bench_pk.err; cut -c1-400 $O/bench_pk.log
R3D_LIB=$PWD/tests/_build/libr3d_hip_nopk.so timeout 300 python bench.py --no-extras --no-cpu-baseline --no-traffic --sr-precision f16x3 > $O/bench_nopk_x3.log 2> $O/bench_nopk_x3.err; cut -c1-400 $O/bench_nopk_x3.log
timeout 300 python bench.py --no-extras --no-cpu-baseline --no-traffic --sr-precision f16x3 > $O/bench_pk_x3.log 2> $O/bench_pk_x3.err; cut -c1-400 $O/bench_pk_x3.log
R3D_MX_MAX_DEPTH=3 timeout 600 python -m pytest tests/test_gpu_mx.py tests/test_gpu_pinned_config.py tests/test_gpu_f16x3.py tests/test_gpu_coresidency.py -m gpu -q -rP > $O/pytest_depth3.log 2>&1; echo "depth3 rc $?"; tail -3 $O/pytest_depth3.log
grep -E "heavy tail|f16mx|FAILED|Error" $O/pytest_depth3.log | head -150 > $O/pytest_depth3_lines.txt
