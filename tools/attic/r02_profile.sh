# Round-2 measurement bundle, run on the GPU box from the repo root (writes gpurun_out/, copy the summaries into profiles/):
#   1. the bench line (full: roofline + cpu_baseline)          -> gpurun_out/r02_bench_b128_line.json
#   2. rocprofv3 --kernel-trace of the same command            -> gpurun_out/r02_bench_b128_kernel_stats.txt / _families.txt
#   3. three PMC passes (one counter each, kernel-trace only)  -> gpurun_out/r02_pmc_gru_fwd.json / r02_pmc_gru_bwd.json
export TMPDIR=/tmp
TAG=${1:-r02}
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_b128_line.json
rm -rf /tmp/prof; rocprofv3 --kernel-trace -d /tmp/prof -o kt -- python3 bench.py --steps 9 --warmup 3 --no-cpu-baseline --primary-only > /tmp/b.log 2>&1
db=$(find /tmp/prof -name "*.db" | head -1)
python tools/rocpd_stats.py $db 70 > gpurun_out/${TAG}_bench_b128_kernel_stats.txt
python tools/profile_categories.py gpurun_out/${TAG}_bench_b128_kernel_stats.txt 12 > gpurun_out/${TAG}_bench_b128_kernel_families.txt
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --primary-only > /tmp/pmc_$c.log 2>&1
done
python tools/pmc_gru_json.py $(find /tmp/pmc_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/pmc_WRITE_SIZE -name "*.db" | head -1) \
       $(find /tmp/pmc_SQ_VALU_MFMA_BUSY_CYCLES -name "*.db" | head -1) gpurun_out > gpurun_out/${TAG}_pmc_gru.log 2>&1
cat gpurun_out/${TAG}_bench_b128_kernel_families.txt
cut -c1-1200 gpurun_out/${TAG}_bench_b128_line.json
