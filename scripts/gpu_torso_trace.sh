# per-launch durations of the torso frame's kernels (rocprofv3 --kernel-trace of scripts/prof_torso.py), in launch order of one frame
export R=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/torso_trace -o t -- python $R/scripts/prof_torso.py 6 > $R/gpurun_out/torso_trace.log 2>&1
tail -1 $R/gpurun_out/torso_trace.log
python - <<'PY'
import csv, glob, os
R = os.environ["R"]
rows = list(csv.DictReader(open(glob.glob(R + "/gpurun_out/torso_trace/**/t_kernel_trace.csv", recursive=True)[0])))
idx = [i for i, r in enumerate(rows) if "render_kernel" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
tot = 0.0
for r in rows[a:b]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; tot += d
    print("%8.1f us  %-64s grid=%s wg=%s" % (d, r["Kernel_Name"].replace("void r3d::", "").split("(")[0][:64], r["Grid_Size_X"] + "x" + r["Grid_Size_Y"], r["Workgroup_Size_X"]))
print("sum %.1f us over %d launches" % (tot, b - a))
PY
