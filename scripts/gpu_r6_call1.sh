# round 6, call 1: the Winograd F(2,3) conv on the GPU: SR goldens / sweeps under both precisions, then the A/B bench (R3D_CONV_WINO=0|1)
mkdir -p gpurun_out/r6c1; O=gpurun_out/r6c1
timeout 900 python -m pytest tests/test_gpu_mx.py tests/test_gpu_parity.py tests/test_gpu_f16x3.py tests/test_gpu_pinned_config.py -m gpu -q -x -rP > $O/pytest_sr.log 2>&1; echo "pytest rc $?"; tail -15 $O/pytest_sr.log
for w in 0 1; do
R3D_CONV_WINO=$w timeout 300 python bench.py --no-extras --no-cpu-baseline --no-traffic > $O/bench_w$w.log 2> $O/bench_w$w.err; echo "wino=$w rc $?"; cut -c1-700 $O/bench_w$w.log
R3D_CONV_WINO=$w timeout 300 python bench.py --no-extras --no-cpu-baseline --no-traffic --sr-precision f16x3 > $O/bench_x3_w$w.log 2> $O/bench_x3_w$w.err; echo "x3 wino=$w rc $?"; cut -c1-400 $O/bench_x3_w$w.log
done
