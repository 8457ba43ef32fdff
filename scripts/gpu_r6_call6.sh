#!/bin/bash
# round 6, call 6: the fused blend + 1x1 conv (r3d_conv_forward_blend): its tests, the warp-network tests, torso frame A/B
mkdir -p gpurun_out/r6c6
( timeout 900 python -m pytest tests/test_gpu_blend_conv.py tests/test_gpu_warp_sr.py tests/test_gpu_parity.py -x -q -m gpu -s 2>&1 | grep -E "blend conv|stack with|passed|failed|Error|error|assert" | tail -40 ) > gpurun_out/r6c6/tests.txt 2>&1
for f in 1 0; do
  for p in f16mx f16x3; do
    echo "R3D_FUSE_BLEND=$f $p: $(R3D_FUSE_BLEND=$f R3D_SR_PRECISION=$p timeout 300 python scripts/prof_torso.py 200 2>&1 | tail -1)" >> gpurun_out/r6c6/torso_ab.txt
  done
done
export R=$PWD; cd /tmp; export TMPDIR=/tmp
for f in 1 0; do
  rm -rf /tmp/kt; R3D_FUSE_BLEND=$f timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python $R/scripts/prof_torso.py 40 > /dev/null 2>&1
  python - <<PY > $R/gpurun_out/r6c6/torso_kernels_$f.txt
import csv, glob
f = glob.glob("/tmp/kt/**/t_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 42 / 1e3
print("sum of kernel time per frame: %.1f us" % tot)
for r in rows[:30]:
    print("%-100s calls %6s avg %9.1f us  %5.1f %%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
done
cd $R
cat gpurun_out/r6c6/tests.txt gpurun_out/r6c6/torso_ab.txt
