export R=$PWD; mkdir -p gpurun_out
cd /tmp; export TMPDIR=/tmp
run() {
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/srq_$1 -o p -- python $R/scripts/prof_sr.py 5 > $R/gpurun_out/srq.log 2>&1
python - $1 <<'PY'
import csv, collections, os, sys
R=os.environ["R"]
rows=list(csv.DictReader(open(R+"/gpurun_out/srq_%s/p_kernel_trace.csv" % sys.argv[1])))
d=collections.defaultdict(list)
for r in rows: d[(r["Kernel_Name"].split("(")[0][:40], r["Grid_Size_X"])].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in d.items():
    if len(v)>=5 and "upconv" in k[0]:
        m=sum(v[-5:])/5/1e3; print(sys.argv[1], "%-44s %8s %8.1f us" % (k[0],k[1],m))
PY
}
timeout 300 python -m pytest $R/tests/test_gpu_parity.py -m gpu -x -q -k "sr_ or synthesis" 2>&1 | tail -2
run full
cp $R/real3dportrait_amd/lib/libr3d_hip.so /tmp/keep.so
for v in 32 64; do cp $R/scripts/probes/bin/libr3d_abl$v.so $R/real3dportrait_amd/lib/libr3d_hip.so; run abl$v; done
cp /tmp/keep.so $R/real3dportrait_amd/lib/libr3d_hip.so
