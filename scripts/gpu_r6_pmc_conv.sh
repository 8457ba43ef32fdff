# SQ counters of the plain-conv kernels of an SR frame (prof_sr.py), per (library, R3D_CONV_WINO): gpu_r6_pmc_conv.sh "<lib>:<wino> ..."
export R=$PWD; mkdir -p gpurun_out/r6pmc; cd /tmp; export TMPDIR=/tmp
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM")
for cfg in ${1:-hip:1 hip:0}; do
  l=${cfg%%:*}; w=${cfg##*:}; tag=${l}_w$w
  i=0
  for set in "${SETS[@]}"; do i=$((i+1)); rm -rf $R/gpurun_out/r6pmc/${tag}_$i
    R3D_LIB=$R/real3dportrait_amd/lib/libr3d_$l.so R3D_CONV_WINO=$w rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/r6pmc/${tag}_$i -o p -- python $R/scripts/prof_sr.py 3 > $R/gpurun_out/r6pmc/${tag}_$i.log 2>&1
  done
  python - $R/gpurun_out/r6pmc $tag <<'PY'
import csv, glob, sys, collections
d, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("%s/%s_*/**/*counter_collection.csv" % (d, tag), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv_wino" not in k and "conv_mfma" not in k: continue
        key = (k.split("(")[0][-40:], r["Grid_Size"])
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("==", tag)
for key in sorted(acc):
    c = {n: sum(v) / len(v) for n, v in acc[key].items()}
    print("  kernel %s grid %s" % key)
    print("    " + "  ".join("%s=%.4g" % (n.replace("SQ_", ""), v) for n, v in sorted(c.items())))
    wc = c.get("SQ_WAVE_CYCLES", 0)
    if wc:
        print("    per wave-cycle: valu_active %.3f  any_active %.3f  wait_any %.3f  mfma_busy/busy_cycles %.3f" % (c.get("SQ_ACTIVE_INST_VALU", 0) / wc, c.get("SQ_ACTIVE_INST_ANY", 0) / wc, c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1.0, c.get("SQ_BUSY_CYCLES", 1))))
PY
  rm -rf $R/gpurun_out/r6pmc/${tag}_[0-9]
done
