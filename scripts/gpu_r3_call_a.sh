#!/bin/bash
# round 3: the quad-shared gather variant built through the packed-f32 rewriter must be independent of co-resident kernels; full GPU suite on the new build pipeline
cd "$(dirname "$0")/.." || exit 1
L=$PWD/real3dportrait_amd/lib
for load in 1 agg:29 agg:31; do
  R3D_LIB=$L/libr3d_hip_ray4096fix.so timeout 300 python scripts/gpu_debug_determinism.py 100 $load 2>&1 | grep -E "^lib|rror" | sed "s/^/[4096 + pk fix | $load] /"
done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -8
