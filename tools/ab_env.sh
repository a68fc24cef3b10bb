# alternating A/B of one environment switch on the eager step: bash tools/ab_env.sh NAME [bench flags]
NAME=${1:-HA2G_CONV_BN_STATS}; shift
for i in 1 2 3; do for v in 0 1; do echo -n "$NAME=$v: "; env $NAME=$v python bench.py --steps 12 --warmup 4 --no-cpu-baseline --primary-only --no-roofline --launch eager "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; done; done
