#!/bin/bash
# GPU call 1: validate the range-management plumbing + new tests, then bench + profile
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" 
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 40 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/bench.log
