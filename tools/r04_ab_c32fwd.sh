# A/B: layer-1 forward on the three-piece anti-phase kernel (default) vs the fp32 direct kernel (HA2G_DIRECT_C32 with bit 6)
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_blocks.py tests/test_gpu_tail.py tests/test_gpu_np3.py -x -q 2>&1 | tail -3
for b in 65 1 65 1; do echo DIRECT_C32=$b; HA2G_DIRECT_C32=$b python bench.py --steps 12 --warmup 4 --no-cpu-baseline --primary-only --no-roofline --launch eager 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; done
