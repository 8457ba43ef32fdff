#!/bin/bash
# per-kernel durations of the SR alone (rocprofv3 --kernel-trace --stats), f16mx
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out/r4_sr_stats; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
R3D_SR_PRECISION=${1:-f16mx} rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python $R/scripts/prof_sr.py 30 > $O/log.txt 2>&1
python - <<PY
import csv, glob
f = glob.glob("$O/**/p_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print("%-70s calls %5s avg %9.1f ns min %9s max %9s  %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"], r["Percentage"]))
PY
