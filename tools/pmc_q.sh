# PMC passes over the plane kernel alone (tools/q_kernel_run.py) -> gpurun_out/${TAG}_pmc_q.txt
export TMPDIR=/tmp
TAG=${1:-r04}
out=$PWD/gpurun_out/${TAG}_pmc_q.txt; : > $out
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pmcq$i
  rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmcq$i -o p -- python3 tools/q_kernel_run.py > /tmp/pmcq$i.log 2>&1
  db=$(find /tmp/pmcq$i -name "*.db" | head -1)
  echo "== $grp" >> $out
  python tools/rocpd_pmc.py $db ${KERNEL:-pconv_q} >> $out 2>&1
done
cut -c1-150 $out
