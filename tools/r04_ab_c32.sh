# A/B of the prefetching 32-channel direct convolution: tests, isolated kernel, step
timeout 900 python -m pytest tests/test_gpu_np3.py tests/test_gpu_kernels.py tests/test_gpu_blocks.py -x -q -k "conv or c32 or block or encoder or taps or trunk" 2>&1 | tail -3
for b in 33 1; do echo C32_PREFETCH=$b; HA2G_C32_PREFETCH=$b timeout 300 python tools/bwd_matrix_bench.py 2>/dev/null | grep -E "^70 +conv3x3 (dgrad|wgrad) C=32"; done
for b in 33 1 33 1; do echo C32_PREFETCH=$b; HA2G_C32_PREFETCH=$b python bench.py --steps 12 --warmup 4 --no-cpu-baseline --primary-only --no-roofline --launch eager 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; done
