#!/bin/bash
# Kernel trace of the torso frame (BASELINE config 4 surrogate, bench.py build_torso_frame): per-kernel time and launch count per frame.
cd $GRAFT_REPO_ROOT; export R=$PWD; O=$R/gpurun_out/torso_trace; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python $R/scripts/torso_frames.py 20 > $O/run.log 2>&1
python - <<'PY'
import csv, glob, os
O = os.environ["R"] + "/gpurun_out/torso_trace"
st = glob.glob(O + "/stats/**/p_kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(st[0])))
out = ["%-90s %6s %10s %10s" % ("kernel", "calls", "total_us", "avg_us")]
for r in rows:
    out.append("%-90s %6s %10.1f %10.2f" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3))
open(O + "/torso_kernel_stats.txt", "w").write("\n".join(out) + "\n")
PY
tail -2 $O/run.log; head -45 $O/torso_kernel_stats.txt
