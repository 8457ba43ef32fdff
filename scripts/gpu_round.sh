mkdir -p gpurun_out; export R=$PWD
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 40 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err; tail -2 gpurun_out/bench.err; cat gpurun_out/bench.log
