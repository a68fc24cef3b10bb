# PMC passes (one counter group per run, kernel-trace only) over tools/bwd_matrix_bench.py: instruction mix, MFMA-busy, waits, LDS conflicts
# of the backward matrix kernels (conv dgrad / wgrad, dense dW / dX) -> gpurun_out/${TAG}_pmc_bwd_gemm.txt
export TMPDIR=/tmp
TAG=${1:-r03}
out=$PWD/gpurun_out/${TAG}_pmc_bwd_gemm.txt; : > $out
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pmcb$i
  rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmcb$i -o p -- python3 tools/bwd_matrix_bench.py 3 > /tmp/pmcb$i.log 2>&1
  db=$(find /tmp/pmcb$i -name "*.db" | head -1)
  echo "== $grp" >> $out
  python tools/rocpd_pmc.py $db gemm >> $out 2>&1
  python tools/rocpd_pmc.py $db conv3x3 >> $out 2>&1
done
cut -c1-160 $out | head -150
