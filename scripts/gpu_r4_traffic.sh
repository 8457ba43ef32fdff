#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out/r4_traffic; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do R3D_SR_PRECISION=f16mx rocprofv3 --pmc $c --output-format csv -d $O/$c -o p -- python $R/scripts/prof_sr.py 6 > $O/$c.log 2>&1; done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = collections.defaultdict(list)
    for f in glob.glob("$O/%s/**/p_counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c: v[r["Kernel_Name"].split("(")[0].replace("void r3d::", "")].append(float(r["Counter_Value"]))
    for k, x in v.items(): acc[k][c] = sum(x) / len(x)
for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].values()))[:5]:
    print("%-56s FETCH_KiB %10.1f WRITE_KiB %10.1f bytes %12.0f" % (k[:56], d.get("FETCH_SIZE", 0), d.get("WRITE_SIZE", 0), (2 * d.get("FETCH_SIZE", 0) + d.get("WRITE_SIZE", 0)) * 1024))
PY
