# round 5, call 13: end state after the workspace parking of the big ray shapes: full -m gpu suite; config 5 HBM-side traffic (FETCH / WRITE passes)
mkdir -p gpurun_out/r5c13; O=$PWD/gpurun_out/r5c13
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_full.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed" $O/pytest_full.log | tail -1
R=$PWD; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --output-format csv -d $O/cfg5_$c -o p -- python $R/scripts/stress_cfg5.py > $O/cfg5_$c.log 2>&1; done
python - <<'PY'
import csv, glob, collections, os
O = os.environ.get("O", "/tmp")
PY
cd $R
python - <<'PY'
import csv, glob, collections
O = "gpurun_out/r5c13"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(O + "/cfg5_" + c + "/**/p_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c: acc[r["Kernel_Name"].split("(")[0].replace("void r3d::", "")][c].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items(), key=lambda kv: -sum(sum(x) / len(x) for x in kv[1].values()))[:5]:
    f = sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"])); w = sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"]))
    print("%-60s FETCH_KiB %12.0f WRITE_KiB %12.0f bytes %.3e" % (k[:60], f, w, (2 * f + w) * 1024))
PY
