#!/bin/bash
# A/B of the SR kernels between environment settings ("" = product defaults): per-kernel durations under rocprofv3, and a digest of the SR output
# usage: gpu_r4_ab_env.sh "R3D_CONV_W4=1" "R3D_CONV_W4=3 R3D_SR_PRECISION=f16x3" ...
cd "$(dirname "$0")/.." || exit 1
R=$PWD; cd /tmp; export TMPDIR=/tmp
for E in "" "$@"; do
  O=$R/gpurun_out/r4_ab/$(echo "${E:-product}" | tr ' =/' '___'); rm -rf $O; mkdir -p $O
  echo "== ${E:-product}"
  env R3D_SR_PRECISION=f16mx $E rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python $R/scripts/prof_sr.py 30 > $O/log.txt 2>&1
  grep -i "digest" $O/log.txt
  python - <<PY
import csv, glob
f = glob.glob("$O/**/p_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:4]:
    print("%-74s calls %5s avg %9.1f ns" % (r["Name"][:74], r["Calls"], float(r["AverageNs"])))
PY
done
