mkdir -p gpurun_out; export R=$PWD
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 40 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err; tail -2 gpurun_out/bench.err; cat gpurun_out/bench.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r01 -- python $R/bench.py --steps 10 --warmup 2 --streams 1 --no-cpu-baseline --no-extras > $R/gpurun_out/prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o r01 -- python $R/bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-extras > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o r01 -- python $R/bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-extras > $R/gpurun_out/pmc_write.log 2>&1
