#!/bin/bash
# A/B of the SR kernels between the product library and other builds (R3D_LIB): per-kernel durations under rocprofv3 + a digest of the SR output
cd "$(dirname "$0")/.." || exit 1
R=$PWD; cd /tmp; export TMPDIR=/tmp
for L in "" "$@"; do
  O=$R/gpurun_out/r4_ab/$(basename "${L:-product}"); rm -rf $O; mkdir -p $O
  echo "== ${L:-product}"
  R3D_LIB=${L:+$R/$L} R3D_SR_PRECISION=${PREC:-f16mx} rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python $R/scripts/prof_sr.py 30 > $O/log.txt 2>&1
  grep digest $O/log.txt
  python - <<PY
import csv, glob
f = glob.glob("$O/**/p_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:4]:
    print("%-74s calls %5s avg %9.1f ns" % (r["Name"][:74], r["Calls"], float(r["AverageNs"])))
PY
done
