for cfg in "" "HA2G_PLANE_GEMM=0" "HA2G_GRU_MERGE_DIRS=0" "HA2G_PLANE_GEMM_MIN=4000000000"; do
  echo "== $cfg"
  env $cfg python bench.py --steps 10 --warmup 3 --no-cpu-baseline --primary-only --no-roofline --launch eager 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['eager']['host_ms_per_step'])"
done
