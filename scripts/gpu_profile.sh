# Round profile: rocprofv3 kernel stats + HBM traffic counters (separate --pmc passes) of bench.py on ONE stream.
export R=$PWD; mkdir -p gpurun_out/profile; cd /tmp; export TMPDIR=/tmp
python $R/bench.py --steps 40 --warmup 5 > $R/gpurun_out/profile/bench_n1.json 2> $R/gpurun_out/profile/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/profile/stats -o p -- python $R/bench.py --steps 10 --warmup 2 --streams 1 --no-cpu-baseline --no-extras > $R/gpurun_out/profile/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/profile/pmc_fetch -o p -- python $R/bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-extras > $R/gpurun_out/profile/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/profile/pmc_write -o p -- python $R/bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-extras > $R/gpurun_out/profile/pmc_write.log 2>&1
python - <<'PY'
import csv, glob, os, collections, json, shutil
R = os.environ["R"]; out = R + "/gpurun_out/profile"
st = glob.glob(out + "/stats/**/p_kernel_stats.csv", recursive=True)
if st: shutil.copy(st[0], out + "/kernel_stats_streams1.csv")
lines = []; per = {}
for tag, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = glob.glob(out + "/pmc_%s/**/p_counter_collection.csv" % tag, recursive=True)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == name: acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    lines.append("== %s (KiB per dispatch, rocprofv3 --pmc %s; own pass; bench.py --streams 1) ==" % (name, name))
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]) / len(kv[1])):
        lines.append("%-70s calls %4d  avg_KiB %14.2f" % (k[:70], len(v), sum(v) / len(v)))
        per.setdefault(k, {})[name] = sum(v) / len(v) * 1024
open(out + "/pmc_summary.txt", "w").write("\n".join(lines) + "\n")
conv = [v for k, v in per.items() if "conv_mfma_f16x3_kernel" in k]
if conv:
    c = conv[0]
    json.dump({"conv_bytes_per_launch_f16x3": int(2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)),
               "f16x3_note": "conv_mfma_f16x3_kernel, avg over its 2 launches per frame; 2 x FETCH_SIZE (gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md) + WRITE_SIZE; separate --pmc passes",
               "fetch_size_bytes_raw": int(c.get("FETCH_SIZE", 0)), "write_size_bytes": int(c.get("WRITE_SIZE", 0))},
              open(out + "/traffic_f16x3.json", "w"), indent=1)
print(open(out + "/pmc_summary.txt").read()[:1800])
PY
head -c 600 $R/gpurun_out/profile/bench_n1.json; echo; head -12 $R/gpurun_out/profile/kernel_stats_streams1.csv | cut -c1-150
