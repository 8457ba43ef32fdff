#!/bin/bash
# Round-6 profile.  (1) bench line, (2) rocprofv3 kernel stats of bench.py on ONE stream (f16mx = what `value` is measured with, and f16x3),
# (3) HBM traffic (FETCH_SIZE / WRITE_SIZE, own passes) and the SQ / TCC / GRBM counter table of the head frame's kernels, (4) the same two
# kinds of passes for BASELINE config 5 (scripts/stress_cfg5.py) and for the torso frame (scripts/torso_frames.py): VERDICT r3 item 8,
# (5) rocm-smi power while the frame loop runs, (6) per-kernel time of the torso frame.  (The phase stamps of r04 were not re-taken: the kernels' structure is unchanged.)
# Round 6 adds: the same one-stream kernel stats with the Winograd F(2,3) conv forced on for f16mx (R3D_CONV_WINO=1; the product default is f16x3 only),
# its SQ counters next to the direct kernel's (scripts/gpu_r6_pmc_conv.sh) and its phase stamps (instrumented build libr3d_hip_stamps.so, if present).
# Writes gpurun_out/profile_r06/; the summaries are copied to profiles/r06/ by hand (tracked).
cd $GRAFT_REPO_ROOT; export R=$PWD; O=$R/gpurun_out/profile_r06; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python $R/bench.py --steps 40 --warmup 5 > $O/bench_n1.json 2> $O/bench.err
B="python $R/bench.py --steps 10 --warmup 2 --streams 1 --no-cpu-baseline --no-extras --no-traffic"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- $B > $O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_f16x3 -o p -- $B --sr-precision f16x3 > $O/stats_f16x3.log 2>&1
R3D_CONV_WINO=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_f16x3_direct -o p -- $B --sr-precision f16x3 > $O/stats_f16x3_direct.log 2>&1
R3D_CONV_WINO=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_wino_mx -o p -- $B > $O/stats_wino_mx.log 2>&1
(cd $R && bash scripts/gpu_r6_pmc_conv.sh "hip:1 hip:0" > $O/winograd_pmc.txt 2>&1)
if [ -f $R/real3dportrait_amd/lib/libr3d_hip_stamps.so ]; then for p in f16mx f16x3; do R3D_LIB=$R/real3dportrait_amd/lib/libr3d_hip_stamps.so R3D_CONV_WINO=1 python $R/scripts/gpu_wino_stamps.py $p; done > $O/winograd_phase_stamps.txt 2>&1; fi
SETS=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr")
run_passes() {   # $1 = tag, rest = command
  tag=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --output-format csv -d $O/${tag}_pmc_$c -o p -- "$@" > $O/${tag}_pmc_$c.log 2>&1; done
  i=0
  for set in "${SETS[@]}"; do i=$((i+1)); rocprofv3 --pmc $set --output-format csv -d $O/${tag}_pmc_set$i -o p -- "$@" > $O/${tag}_pmc_set$i.log 2>&1; done
}
run_passes head python $R/bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-extras --no-traffic
run_passes cfg5 env R3D_SR_PRECISION=f16mx python $R/scripts/stress_cfg5.py
run_passes torso env R3D_SR_PRECISION=f16mx python $R/scripts/torso_frames.py 4
R3D_SR_PRECISION=f16mx rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg5 -o p -- python $R/scripts/stress_cfg5.py > $O/stats_cfg5.log 2>&1
python - <<'PY'
import csv, glob, os, collections, json, shutil
O = os.environ["R"] + "/gpurun_out/profile_r06"
for src, dst in (("stats", "kernel_stats_streams1.csv"), ("stats_f16x3", "kernel_stats_streams1_f16x3.csv"), ("stats_f16x3_direct", "kernel_stats_streams1_f16x3_direct_conv.csv"),
                 ("stats_wino_mx", "kernel_stats_streams1_f16mx_winograd_forced.csv"), ("stats_cfg5", "kernel_stats_cfg5.csv")):
    st = glob.glob(O + "/" + src + "/**/p_kernel_stats.csv", recursive=True)
    if st: shutil.copy(st[0], O + "/" + dst)
def collect(pattern):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(O + "/" + pattern + "/**/p_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0].replace("void r3d::", "").replace("r3d::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc
lines, tj = [], {"source": "profiles/r06/pmc_summary.txt (scripts/gpu_profile_r06.sh; bench.py re-measures the head frame's figure itself when rocprofv3 is on PATH)"}
BIG = ("render_kernel", "conv_mfma_f16x3", "upconv_fir", "conv1x1", "planes_to_nhwc", "blend_cat", "rgb_finalize")
for tag, title in (("head", "head frame (bench.py --streams 1, f16mx)"), ("cfg5", "BASELINE config 5: N = 8, R = 256, 96 + 96, SR -> 1024^2 (scripts/stress_cfg5.py, f16mx)"),
                   ("torso", "torso frame (scripts/torso_frames.py, f16mx)")):
    tr = collect(tag + "_pmc_[FW]*")
    lines.append("== %s: HBM traffic per dispatch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per pass) ==" % title)
    lines.append("   bytes = 2 x FETCH_SIZE KiB (gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md) + WRITE_SIZE KiB")
    for k, v in sorted(tr.items(), key=lambda kv: -sum(sum(x) / len(x) for x in kv[1].values()))[:12]:
        f = sum(v.get("FETCH_SIZE", [0])) / max(1, len(v.get("FETCH_SIZE", [0]))); w = sum(v.get("WRITE_SIZE", [0])) / max(1, len(v.get("WRITE_SIZE", [0])))
        lines.append("%-66s calls %4d  FETCH_KiB %12.1f  WRITE_KiB %12.1f  bytes %14.0f" % (k[:66], len(v.get("FETCH_SIZE", [])), f, w, (2 * f + w) * 1024))
        if tag == "head" and "conv_mfma_f16x3_kernel" in k: tj["conv_bytes_per_launch_" + ("f16mx" if "true>" in k else "f16x3")] = int((2 * f + w) * 1024)
        if any(t in k for t in BIG): tj.setdefault(tag, {})[k[:60]] = int((2 * f + w) * 1024)
    lines.append("")
    lines.append("== %s: SQ / TCP / TCC / GRBM counters per dispatch (avg over dispatches; 4 separate --pmc passes) ==" % title)
    cn = collect(tag + "_pmc_set*")
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
            lines.append("    -> MFMA busy per SIMD / kernel cycles = (MFMA_BUSY / 1024) / (GUI_ACTIVE / 8)   %.3f" % ((v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (v["GRBM_GUI_ACTIVE"] / 8.0)))
        if v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0):
            lines.append("    -> L2 hit rate TCC_HIT / (HIT + MISS)                 %.4f" % (v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])))
        if v.get("SQ_LDS_BANK_CONFLICT") is not None and v.get("SQ_ACTIVE_INST_LDS"):
            lines.append("    -> LDS conflict cycles / LDS active                   %.3f" % (v["SQ_LDS_BANK_CONFLICT"] / v["SQ_ACTIVE_INST_LDS"]))
    lines.append("")
open(O + "/pmc_summary.txt", "w").write("\n".join(lines) + "\n")
json.dump(tj, open(O + "/traffic.json", "w"), indent=1)
print(open(O + "/pmc_summary.txt").read()[:9000])
PY
tail -c 600 $O/bench_n1.json; echo; head -8 $O/kernel_stats_streams1.csv | cut -c1-150; head -8 $O/kernel_stats_cfg5.csv | cut -c1-150
bash $R/scripts/gpu_power_probe.sh > $O/power_probe.txt 2>&1; cat $O/power_probe.txt
R3D_SR_PRECISION=f16mx bash $R/scripts/gpu_torso_trace.sh > $O/torso_trace.log 2>&1; cp $R/gpurun_out/torso_trace/torso_kernel_stats.txt $O/ 2>/dev/null; head -12 $O/torso_kernel_stats.txt
