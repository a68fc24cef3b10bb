# A/B of the plane kernel's switches on one box: tests, isolated kernel (ablation columns), step time
timeout 900 python -m pytest tests/test_gpu_np3.py tests/test_gpu_blocks.py tests/test_gpu_planes.py -x -q 2>&1 | tail -3
timeout 300 python tools/planes_ablate.py 2>&1 | grep -E "B=128.*np 3 tile 5"
for b in 16 0 16 0; do echo PLANES_DEBUG=$b; HA2G_PLANES_DEBUG=$b python bench.py --steps 12 --warmup 4 --no-cpu-baseline --primary-only --no-roofline --launch eager 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; done
