mkdir -p gpurun_out/r6c6
export R=$PWD; cd /tmp; export TMPDIR=/tmp
for f in 1 0; do
  rm -rf /tmp/kt; R3D_FUSE_BLEND=$f timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python $R/scripts/prof_torso.py 40 > /tmp/log_$f.txt 2>&1
  python - <<PY > $R/gpurun_out/r6c6/torso_kernels_$f.txt
import csv, glob
f = glob.glob("/tmp/kt/**/t_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 42 / 1e3
print("sum of kernel time per frame (incl. set-up kernels / 42): %.1f us" % tot)
for r in rows[:32]:
    print("%-100s calls %6s avg %9.1f us  %5.1f %%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
  tail -1 /tmp/log_$f.txt
done
