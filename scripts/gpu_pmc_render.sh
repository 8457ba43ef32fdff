export R=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/counters.txt 2>&1
python $R/scripts/prof_render.py 128 48 48 20
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmcr_$tag -o p -- python $R/scripts/prof_render.py 128 48 48 3 > $R/gpurun_out/pmcr_$tag.log 2>&1
done
ls $R/gpurun_out
python - <<'PY'
import csv, glob, os, collections
R=os.environ["R"]
for f in sorted(glob.glob(R+"/gpurun_out/pmcr_*/p_counter_collection.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): print("%-34s %16.0f  (n=%d)" % (k, sum(v)/len(v), len(v)))
PY
