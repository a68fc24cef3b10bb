# PMC passes (one counter group per run, kernel-trace only) over tools/conv_vs_gemm.py; summaries to gpurun_out/pmc_conv.txt
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_conv.txt; : > $out
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1)); rm -rf /tmp/pmc$i
  rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc$i -o p -- python3 tools/conv_vs_gemm.py > /tmp/pmc$i.log 2>&1
  db=$(find /tmp/pmc$i -name "*.db" | head -1)
  python tools/rocpd_pmc.py $db ${PMC_FILTER:-gemm_kernel} >> $out 2>&1
done
cat $out | cut -c1-150
