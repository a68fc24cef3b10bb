# HBM bytes per train step (PMC FETCH_SIZE / WRITE_SIZE, one counter per run, kernel-trace only) of the bf16 step at B = 256 expressive with the
# audio trunk's activations stored as fp32 ("before") and as bf16 ("after") -> gpurun_out/${TAG}_pmc_b16_bytes.txt
export TMPDIR=/tmp
TAG=${1:-r03}
W=2; K=4
out=$PWD/gpurun_out/${TAG}_pmc_b16_bytes.txt; : > $out
for mode in "--fp32-storage" ""; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    d=/tmp/pmc16_${ctr}_${mode#--}; rm -rf $d
    rocprofv3 --kernel-trace --pmc $ctr -d $d -o p -- python3 bench.py --bf16 $mode --expressive --batch 256 --steps $K --warmup $W --no-cpu-baseline --primary-only --no-roofline --launch eager > $d.log 2>&1
    db=$(find $d -name "*.db" | head -1)
    echo "== bench.py --bf16 $mode --expressive --batch 256 : $ctr" >> $out
    python tools/pmc_step_bytes.py $db $ctr $((W+K)) 12 >> $out 2>&1
  done
done
cat $out
