# kernel-trace --stats of the bench (single stream): per-kernel average durations -> gpurun_out/kstats/
export R=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kstats -o k -- python $R/bench.py --steps 10 --warmup 2 --streams 1 --no-cpu-baseline --no-extras > $R/gpurun_out/kstats.log 2>&1
f=$(find $R/gpurun_out/kstats -name "k_kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print("%-90s calls=%5s avg=%9.1f us  pct=%s" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
