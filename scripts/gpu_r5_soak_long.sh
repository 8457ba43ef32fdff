# round 5: long co-residency soak (96 frames x 60 passes) + more seeds of the randomised sweep on the FINAL binary (after the workspace parking of the big ray shapes)
O=gpurun_out/r5_soak_long.txt; : > $O
for p in f16mx f16x3; do
  echo "== R3D_SR_PRECISION=$p soak 96 x 60" >> $O
  R3D_SR_PRECISION=$p timeout 900 python scripts/gpu_soak_pipeline.py 96 60 2>&1 | tail -1 >> $O
done
for seed in 71 72 73 74 75 76; do
  p=f16mx; [ $((seed % 2)) -eq 0 ] && p=f16x3
  echo "== R3D_SR_PRECISION=$p seed $seed, 16 cases" >> $O
  R3D_SR_PRECISION=$p timeout 900 python scripts/fuzz_parity.py $seed 16 2>&1 | grep -E "FUZZ|BAD|render R=.* 9[0-9]\+|render R=.* [6-9][0-9]\+[6-9][0-9]" | cut -c1-160 >> $O
done
grep -E "soak:|FUZZ" $O
