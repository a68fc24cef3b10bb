# SQ counter passes over the two cluster GRU kernels alone (tools/gru_fwd_only.py); summary -> gpurun_out/pmc_gru_kernel.txt
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_gru_kernel.txt; : > $out
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pmcg$i
  rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmcg$i -o p -- python3 tools/gru_fwd_only.py > /tmp/pmcg$i.log 2>&1
  db=$(find /tmp/pmcg$i -name "*.db" | head -1)
  python tools/rocpd_pmc.py $db gru_ >> $out 2>&1
done
cut -c1-150 $out
