#!/bin/bash
# round 6, call 8: chain_fold with one load phase: range tests, warp network, torso frame time + per-kernel stats
( timeout 1200 python -m pytest tests/test_gpu_range_and_sizes.py tests/test_gpu_warp_sr.py tests/test_gpu_parity.py tests/test_gpu_mx.py -x -q -m gpu 2>&1 | tail -3 )
for p in f16mx f16x3; do echo "$p: $(R3D_SR_PRECISION=$p timeout 300 python scripts/prof_torso.py 200 2>&1 | tail -1)"; done
bash scripts/gpu_r6_torso_kstats.sh > /dev/null 2>&1; grep -E "chain_fold|absmax|sum of" gpurun_out/r6c6/torso_kernels_1.txt
