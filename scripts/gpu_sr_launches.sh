# per-launch durations of the SR kernels (grouped by kernel name and grid size): rocprofv3 --kernel-trace of scripts/prof_sr.py
export R=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/sr_launches -o t -- python $R/scripts/prof_sr.py 20 > $R/gpurun_out/sr_launches.log 2>&1
python - <<'PY'
import csv, glob, os, collections
R = os.environ["R"]
f = glob.glob(R + "/gpurun_out/sr_launches/**/t_kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0][-60:]
    key = (name, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
    acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)[: max(1, len(v) * 3 // 4)] if len(v) > 8 else v
    print("%-62s grid=%sx%sx%s wg=%s  n=%d  avg=%.1f us" % (k[0], k[1], k[2], k[3], k[4], len(v), sum(v) / len(v)))
PY
