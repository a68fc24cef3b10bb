# Round-6 measurement bundle, run on the GPU box from the repo root (writes gpurun_out/; the summaries to be judged are copied into profiles/):
#   1. three PMC passes (one counter each, kernel-trace only) -> gpurun_out/r06_pmc_gru_{fwd,bwd}.json  (bench.py's roofline.traffic source)
#   2. the default bench line (roofline + cpu_baseline)        -> gpurun_out/r06_bench_b128_line.json
#   3. rocprofv3 --kernel-trace of eager steps                 -> gpurun_out/r06_bench_b128_kernel_{stats,families}.txt, _queues.txt
#   4. secondary configurations (config 3 / B=256 / config 5 at T = 34 and at the T = 62 performance-only window) -> gpurun_out/r06_bench_*_line.json
#   5. isolated kernel tables (backward matrix family, plane kernels q vs r + ablations, 32-channel kernel, GRU) and PMC passes of the r / c32pp kernels
#   6. whole-step HBM bytes (PMC FETCH_SIZE / WRITE_SIZE)
#   7. round 6: split-K forms alone (tools/splitk_bench.py), un-profiled phase spans of the step (tools/phase_spans.py), plane wgrad DMA ablation
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --primary-only --launch eager > /tmp/pmc_$c.log 2>&1
done
python tools/pmc_gru_json.py $(find /tmp/pmc_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/pmc_WRITE_SIZE -name "*.db" | head -1) \
       $(find /tmp/pmc_SQ_VALU_MFMA_BUSY_CYCLES -name "*.db" | head -1) gpurun_out r06 > gpurun_out/r06_pmc_gru.log 2>&1
cp gpurun_out/r06_pmc_gru_fwd.json gpurun_out/r06_pmc_gru_bwd.json profiles/     # so that the bench line below reads this run's counters
python bench.py --steps 20 --warmup 5 2> gpurun_out/r06_bench_b128.err | tail -1 > gpurun_out/r06_bench_b128_line.json
bash tools/r06_trace.sh r06_bench_b128 > /dev/null 2>&1     # kernel stats / families / queues / main-queue gaps / step sequence / queue overlap from ONE trace
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --batch 256 2>/dev/null | tail -1 > gpurun_out/r06_bench_b256_line.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --expressive 2>/dev/null | tail -1 > gpurun_out/r06_bench_expressive_line.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --expressive --batch 256 2>/dev/null | tail -1 > gpurun_out/r06_bench_expressive_b256_line.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --expressive --batch 256 --bf16 2>/dev/null | tail -1 > gpurun_out/r06_bench_expressive_b256_bf16_line.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --expressive --batch 256 --bf16 --n-poses 62 2>/dev/null | tail -1 > gpurun_out/r06_bench_expressive_b256_bf16_t62_line.json
for f in b128 b256 expressive expressive_b256 expressive_b256_bf16 expressive_b256_bf16_t62; do
  python - <<PY
import json
d = json.loads(open('gpurun_out/r06_bench_${f}_line.json').read())
g = d.get('graph_replay') or {}
print('${f}', d['ms_per_step'], d['value'], d['launch'], 'graph', g.get('ms_per_step'), 'exact', d['exact_fp32_matrix_core']['ms_per_step'], 'warm', d['warmup_phase']['ms_per_step'],
      'roof', d['roofline']['frac'] if d['roofline'] else None, d['roofline'].get('traffic_stale') if d['roofline'] else None)
PY
done
python tools/bwd_matrix_bench.py 2>/dev/null | grep -E "^(mode|70|6 |0 )" > gpurun_out/r06_bwd_matrix_bench.txt
python tools/r_kernel_bench.py 2>/dev/null | grep -vE "Warn|amdgpu" > gpurun_out/r06_r_kernel_bench.txt
python tools/r_kernel_ablate.py 2>/dev/null | grep -vE "Warn|amdgpu" > gpurun_out/r06_r_kernel_ablate.txt
python tools/c32_ablate.py 2>/dev/null | grep -vE "Warn|amdgpu" > gpurun_out/r06_c32_ablate.txt
python tools/gru_fwd3_bench.py 2>/dev/null | grep -vE "Warn|amdgpu" > gpurun_out/r06_gru_fwd3_bench.txt
python tools/gru_fastpath_ab.py 2>/dev/null | tail -4 > gpurun_out/r06_gru_fastpath_ab.txt
python tools/gemm_census.py > gpurun_out/r06_gemm_census.txt 2>/dev/null
python tools/splitk_bench.py 2>/dev/null | grep -vE "Warn|amdgpu" > gpurun_out/r06_splitk_bench.txt
python tools/phase_spans.py 2>/dev/null | grep -vE "Warn|amdgpu" > gpurun_out/r06_phase_spans.txt
python tools/planes_ablate_wgrad.py 2>/dev/null | grep -vE "Warn|amdgpu" > gpurun_out/r06_planes_ablate_wgrad.txt
KERNEL=pconv_r bash tools/pmc_q.sh r06r > /dev/null 2>&1
bash tools/pmc_c32.sh r06 > /dev/null 2>&1
bash tools/step_bytes.sh r06 > /dev/null 2>&1
cat gpurun_out/r06_bench_b128_kernel_families.txt
head -40 gpurun_out/r06_bench_b128_queues.txt
