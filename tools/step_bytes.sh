# Whole-step HBM bytes of the headline configuration (B = 128 TED-Gesture, GAN phase): PMC FETCH_SIZE / WRITE_SIZE, one counter per run,
# kernel-trace only -> gpurun_out/${TAG}_pmc_step_bytes_b128.txt      usage: bash tools/step_bytes.sh [TAG] [extra bench.py flags]
export TMPDIR=/tmp
TAG=${1:-r05}; shift
W=2; K=4
out=$PWD/gpurun_out/${TAG}_pmc_step_bytes_b128.txt; : > $out
for ctr in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/pmcstep_${ctr}; rm -rf $d
  rocprofv3 --kernel-trace --pmc $ctr -d $d -o p -- python3 bench.py --steps $K --warmup $W --no-cpu-baseline --primary-only --no-roofline --launch eager "$@" > $d.log 2>&1
  db=$(find $d -name "*.db" | head -1)
  echo "== bench.py (B=128, TED-Gesture, GAN phase, eager) $* : $ctr" >> $out
  python tools/pmc_step_bytes.py $db $ctr $((W+K)) 16 >> $out 2>&1
done
cat $out
