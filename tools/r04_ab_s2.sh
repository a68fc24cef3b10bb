timeout 900 python -m pytest tests/test_gpu_np3.py tests/test_gpu_blocks.py tests/test_gpu_planes.py -x -q 2>&1 | tail -3
for t in 7 0 7 0; do echo TILE3=$t; HA2G_TILE3=$t python bench.py --steps 12 --warmup 4 --no-cpu-baseline --primary-only --launch eager 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernel_times_us'].get('conv_dgrad_planes_s2'))"; done
