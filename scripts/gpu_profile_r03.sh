#!/bin/bash
# Round-3 profile: (1) rocprofv3 kernel stats of bench.py on ONE stream, (2) HBM traffic of the conv kernels (FETCH_SIZE / WRITE_SIZE, own passes),
# (3) the SQ / TCC / GRBM counter table of the ray kernel and of the SR kernels (separate --pmc passes, no trace domains mixed in).
# Writes gpurun_out/profile_r03/; the summaries are copied to profiles/r03/ by hand (tracked).
cd $GRAFT_REPO_ROOT; export R=$PWD; O=$R/gpurun_out/profile_r03; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python $R/bench.py --steps 40 --warmup 5 > $O/bench_n1.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python $R/bench.py --steps 10 --warmup 2 --streams 1 --no-cpu-baseline --no-extras > $O/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o p -- python $R/bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-extras > $O/pmc_$c.log 2>&1
  # the fp32-class precision (alt_f16x3 of the bench line): same passes, its own directory (pmcx_*: not part of the table below)
  R3D_SR_PRECISION=f16x3 rocprofv3 --pmc $c --output-format csv -d $O/pmcx_$c -o p -- python $R/bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-extras > $O/pmcx_$c.log 2>&1
done
R3D_SR_PRECISION=f16x3 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_f16x3 -o p -- python $R/bench.py --steps 10 --warmup 2 --streams 1 --no-cpu-baseline --no-extras > $O/stats_f16x3.log 2>&1
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/pmc_set$i -o p -- python $R/bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-extras > $O/pmc_set$i.log 2>&1
done
python - <<'PY'
import csv, glob, os, collections, json, shutil
O = os.environ["R"] + "/gpurun_out/profile_r03"
st = glob.glob(O + "/stats/**/p_kernel_stats.csv", recursive=True)
if st: shutil.copy(st[0], O + "/kernel_stats_streams1.csv")
st = glob.glob(O + "/stats_f16x3/**/p_kernel_stats.csv", recursive=True)
if st: shutil.copy(st[0], O + "/kernel_stats_streams1_f16x3.csv")
def collect(pattern):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(O + "/" + pattern + "/**/p_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0].replace("void r3d::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc
lines = []
tr = collect("pmc_[FW]*")
per = {}
lines.append("== HBM traffic per dispatch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per pass; bench.py --streams 1) ==")
lines.append("   bytes = 2 x FETCH_SIZE KiB (gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md) + WRITE_SIZE KiB")
for k, v in sorted(tr.items(), key=lambda kv: -sum(sum(x) / len(x) for x in kv[1].values())):
    f = sum(v.get("FETCH_SIZE", [0])) / max(1, len(v.get("FETCH_SIZE", [0]))); w = sum(v.get("WRITE_SIZE", [0])) / max(1, len(v.get("WRITE_SIZE", [0])))
    lines.append("%-64s calls %4d  FETCH_KiB %12.1f  WRITE_KiB %12.1f  bytes %14.0f" % (k[:64], len(v.get("FETCH_SIZE", [])), f, w, (2 * f + w) * 1024))
    per[k] = (2 * f + w) * 1024
lines.append("")
lines.append("== SQ / TCP / TCC / GRBM counters per dispatch (avg over dispatches; 4 separate --pmc passes) ==")
cn = collect("pmc_set*")
for k in sorted(cn):
    if not any(t in k for t in ("render_kernel", "conv_mfma_f16x3", "upconv_fir")): continue
    v = {c: sum(x) / len(x) for c, x in cn[k].items()}
    lines.append(k[:100])
    for c in sorted(v): lines.append("    %-34s %18.0f" % (c, v[c]))
    wc = v.get("SQ_WAVE_CYCLES", 0)
    if wc:
        lines.append("    -> wave-parked (SQ_WAIT_ANY / SQ_WAVE_CYCLES)        %.3f" % (v.get("SQ_WAIT_ANY", 0) / wc))
        lines.append("    -> issue-stalled (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES) %.3f" % (v.get("SQ_WAIT_INST_ANY", 0) / wc))
        lines.append("    -> VALU active (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES) %.3f" % (v.get("SQ_ACTIVE_INST_VALU", 0) / wc))
    if v.get("GRBM_GUI_ACTIVE") and v.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        # SQ_VALU_MFMA_BUSY_CYCLES sums cycles over the 1024 SIMDs (= 32 x the 32x32x16 MFMA count); GRBM_GUI_ACTIVE sums over the 8 XCDs
        lines.append("    -> MFMA busy per SIMD / kernel cycles = (MFMA_BUSY / 1024) / (GUI_ACTIVE / 8)   %.3f" % ((v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (v["GRBM_GUI_ACTIVE"] / 8.0)))
    if v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0):
        lines.append("    -> L2 hit rate TCC_HIT / (HIT + MISS)                 %.4f" % (v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])))
open(O + "/pmc_summary.txt", "w").write("\n".join(lines) + "\n")
tj = {"source": "profiles/r03/pmc_summary.txt (default precision) and the pmcx_* passes of scripts/gpu_profile_r03.sh (R3D_SR_PRECISION=f16x3)"}
for k, v in per.items():
    if "conv_mfma_f16x3_kernel" in k: tj["conv_bytes_per_launch_" + ("f16mx" if "true>" in k else "f16x3")] = int(v)
trx = collect("pmcx_[FW]*")
for k, v in trx.items():
    if "conv_mfma_f16x3_kernel" in k and "true>" not in k:
        f = sum(v.get("FETCH_SIZE", [0])) / max(1, len(v.get("FETCH_SIZE", [0]))); w = sum(v.get("WRITE_SIZE", [0])) / max(1, len(v.get("WRITE_SIZE", [0])))
        tj["conv_bytes_per_launch_f16x3"] = int((2 * f + w) * 1024)
json.dump(tj, open(O + "/traffic.json", "w"), indent=1)
print(tj)
print(open(O + "/pmc_summary.txt").read()[:6000])
PY
tail -c 1500 $O/bench_n1.json; echo; head -14 $O/kernel_stats_streams1.csv | cut -c1-160
bash $R/scripts/gpu_torso_trace.sh > $O/torso_trace.log 2>&1; cp $R/gpurun_out/torso_trace/torso_kernel_stats.txt $O/ 2>/dev/null; tail -3 $O/torso_trace.log
