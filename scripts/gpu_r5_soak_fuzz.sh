# round 5: co-residency soak + randomised parity sweep of the end state, both SR precisions -> gpurun_out/r5_soak_fuzz.txt
O=gpurun_out/r5_soak_fuzz.txt; : > $O
for p in f16mx f16x3; do
  echo "== R3D_SR_PRECISION=$p soak 96 x 30" >> $O
  R3D_SR_PRECISION=$p timeout 600 python scripts/gpu_soak_pipeline.py 96 30 2>&1 | tail -2 >> $O
done
for p in f16mx f16x3; do
  for seed in 51 52 53; do
    echo "== R3D_SR_PRECISION=$p seed $seed, 16 cases" >> $O
    R3D_SR_PRECISION=$p timeout 900 python scripts/fuzz_parity.py $seed 16 2>&1 | grep -v "amdgpu.ids" >> $O
  done
done
grep -E "soak:|summary|failures|BAD|bad" $O | head -20
