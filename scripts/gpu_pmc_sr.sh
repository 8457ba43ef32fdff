export R=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
python $R/scripts/prof_sr.py 20
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmcs_$i -o p -- python $R/scripts/prof_sr.py 3 > $R/gpurun_out/pmcs_$i.log 2>&1
done
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pmcs_trace -o p -- python $R/scripts/prof_sr.py 3 > $R/gpurun_out/pmcs_trace.log 2>&1
