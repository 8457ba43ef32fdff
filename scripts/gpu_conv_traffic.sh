# HBM bytes per launch of the SR kernels for the conv block orders (R3D_CONV_ORDER tuning switch): FETCH_SIZE / WRITE_SIZE in separate --pmc passes
export R=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
for o in ${ORDERS:-0 1 2}; do for c in FETCH_SIZE WRITE_SIZE; do
  R3D_CONV_ORDER=$o rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/ctraf_${o}_$c -o p -- python $R/scripts/prof_sr.py 3 > $R/gpurun_out/ctraf_${o}_$c.log 2>&1
done; done
python - <<'PY'
import csv, glob, os, collections
R = os.environ["R"]
for o in os.environ.get("ORDERS", "0 1 2").split():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(R + "/gpurun_out/ctraf_%s_%s/**/p_counter_collection.csv" % (o, c), recursive=True):
            for r in csv.DictReader(open(f)):
                if "f16x3_kernel" in r["Kernel_Name"]:
                    acc[(r["Kernel_Name"].split("(")[0].replace("void r3d::", "")[:40], r["Grid_Size"], "")][c].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        f = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]); w = sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
        print("order %s  %-40s grid %7s x %s   FETCH %9.0f KiB (x2)  WRITE %9.0f KiB  -> %6.1f MB" % (o, k[0], k[1], k[2], f, w, (2 * f + w) * 1024 / 1e6))
PY
