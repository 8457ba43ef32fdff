#!/bin/bash
# DVFS evidence: sample the shader clock (rocm-smi) while bench.py renders a long clip; prints min / median / max of the samples taken while busy
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( for i in $(seq 1 60); do /opt/rocm/bin/rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|Average Graphics Package Power|Current Socket Graphics Package Power|GPU use" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/clock_samples.txt &
python bench.py --steps 6000 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('long run value', d['value'], 'ms/step', d['ms_per_step'])"
wait
head -40 gpurun_out/clock_samples.txt | cut -c1-220
