# per-kernel average durations of the bench frame on one stream, for R3D_CONV_WINO / precision combinations given as "w:prec ..." (default all four)
export R=$PWD; mkdir -p gpurun_out/r6k; cd /tmp; export TMPDIR=/tmp
for cfg in ${1:-0:f16mx 1:f16mx 0:f16x3 1:f16x3}; do
w=${cfg%%:*}; p=${cfg##*:}
rm -rf $R/gpurun_out/r6k/k_$w$p
R3D_CONV_WINO=$w rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6k/k_$w$p -o k -- python $R/bench.py --steps 20 --warmup 3 --streams 1 --no-cpu-baseline --no-extras --no-traffic --sr-precision $p > $R/gpurun_out/r6k/log_$w$p.txt 2>&1
f=$(find $R/gpurun_out/r6k/k_$w$p -name "k_kernel_stats.csv" | head -1); echo "== wino=$w $p"; python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print("%-80s calls=%5s avg=%9.1f us  pct=%s" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
cp $f $R/gpurun_out/r6k/kernel_stats_w${w}_$p.csv
rm -rf $R/gpurun_out/r6k/k_$w$p
done
