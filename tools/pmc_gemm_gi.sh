# PMC passes over the GRU input-projection GEMM alone (tools/gemm_gi_only.py); summary to gpurun_out/pmc_gemm_gi.txt
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_gemm_gi.txt; : > $out
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum" \
           "SQ_WAIT_ANY SQ_WAVES GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM"; do
  i=$((i+1)); rm -rf /tmp/pmcg$i
  rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmcg$i -o p -- python3 tools/gemm_gi_only.py > /tmp/pmcg$i.log 2>&1
  db=$(find /tmp/pmcg$i -name "*.db" | head -1)
  python tools/rocpd_pmc.py $db gemm_kernel >> $out 2>&1
done
cat $out | cut -c1-170
