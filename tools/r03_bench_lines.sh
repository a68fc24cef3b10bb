python bench.py --steps 20 --warmup 5 2> gpurun_out/r03_bench_b128.err | tail -1 > gpurun_out/r03_bench_b128_line.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --batch 256 2>/dev/null | tail -1 > gpurun_out/r03_bench_b256_line.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --expressive 2>/dev/null | tail -1 > gpurun_out/r03_bench_expressive_line.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --expressive --batch 256 2>/dev/null | tail -1 > gpurun_out/r03_bench_expressive_b256_line.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --expressive --batch 256 --bf16 2>/dev/null | tail -1 > gpurun_out/r03_bench_expressive_b256_bf16_line.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --expressive --batch 256 --bf16 --fp32-storage 2>/dev/null | tail -1 > gpurun_out/r03_bench_expressive_b256_bf16_fp32storage_line.json
for f in b128 b256 expressive expressive_b256 expressive_b256_bf16 expressive_b256_bf16_fp32storage; do
  python - <<PY
import json
d = json.loads(open('gpurun_out/r03_bench_${f}_line.json').read())
print('${f}', d['ms_per_step'], d['value'], d['launch'], 'eager', d['eager']['ms_per_step'], 'exact', d['exact_fp32_matrix_core']['ms_per_step'], 'warm', d['warmup_phase']['ms_per_step'],
      'roof', d['roofline']['frac'] if d['roofline'] else None, d['roofline'].get('traffic_stale') if d['roofline'] else None, 'cpu', d.get('cpu_baseline', {}).get('value'))
PY
done
