export R=$PWD; mkdir -p gpurun_out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/srq -o p -- python $R/scripts/prof_sr.py 8 > $R/gpurun_out/srq.log 2>&1
tail -1 $R/gpurun_out/srq.log
python - <<'PY'
import csv, collections, os
R=os.environ["R"]
rows=list(csv.DictReader(open(R+"/gpurun_out/srq/p_kernel_trace.csv")))
d=collections.defaultdict(list)
for r in rows: d[(r["Kernel_Name"].split("(")[0][:40], r["Grid_Size_X"])].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
tot=0
for k,v in d.items():
    if len(v)>=5 and "r3d" in k[0]:
        m=sum(v[-5:])/5/1e3; tot+=m; print("%-44s %8s %8.1f us" % (k[0],k[1],m))
print("sum", round(tot,1))
PY
