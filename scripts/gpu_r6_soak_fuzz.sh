#!/bin/bash
# round 6 end state: co-residency soak (pipelined vs sequential frames, byte for byte) and the randomised parity sweep, both SR precisions -> gpurun_out/r6soak/soak_and_fuzz.txt
mkdir -p gpurun_out/r6soak; O=gpurun_out/r6soak/soak_and_fuzz.txt; : > $O
for p in f16mx f16x3; do
  echo "== R3D_SR_PRECISION=$p soak 96 x 30" >> $O; R3D_SR_PRECISION=$p timeout 900 python scripts/gpu_soak_pipeline.py 96 30 2>&1 | grep -v amdgpu.ids | tail -2 >> $O
done
for p in f16mx f16x3; do for seed in 61 62 63; do
  echo "== R3D_SR_PRECISION=$p seed $seed, 16 cases" >> $O; R3D_SR_PRECISION=$p timeout 1500 python scripts/fuzz_parity.py $seed 16 2>&1 | grep -v amdgpu.ids >> $O
done; done
grep -E "soak:|FUZZ:|FAIL" $O
