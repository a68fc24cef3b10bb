export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_blocks.py tests/test_gpu_pointwise.py tests/test_gpu_b16.py -x -q 2>&1 | tail -2
rm -rf /tmp/prof_bn; rocprofv3 --kernel-trace -d /tmp/prof_bn -o kt -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --primary-only --no-roofline --launch eager > /tmp/b_bn.log 2>&1
db=$(find /tmp/prof_bn -name "*.db" | head -1)
python tools/rocpd_stats.py $db 200 | grep -E "pair_final|bn_stats_final|col_partial|pool_final"
python bench.py --steps 12 --warmup 4 --no-cpu-baseline --primary-only --no-roofline --launch eager 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
