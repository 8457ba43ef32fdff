mkdir -p gpurun_out; export R=$PWD
timeout 600 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -E "passed|failed|error|sr_full|Error" | tail -8
python bench.py --steps 30 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err; tail -2 gpurun_out/bench.err; cat gpurun_out/bench.log
R3D_SR_PRECISION=f32 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/bench_f32.log 2>&1; tail -1 gpurun_out/bench_f32.log | cut -c1-400
