# One rocprofv3 kernel trace of 12 eager steps (9 analysed) -> every per-trace table of the bundle, from the SAME database:
#   gpurun_out/${TAG}_kernel_stats.txt / _kernel_families.txt / _queues.txt / _main_queue_gaps.txt / _step_seq.txt / _queue_overlap.txt
# usage: bash tools/r06_trace.sh TAG [extra bench args]
export TMPDIR=/tmp
TAG=${1:-r06}; shift
T=$(basename $TAG)
rm -rf /tmp/prof_$T; rocprofv3 --kernel-trace -d /tmp/prof_$T -o kt -- python3 bench.py --steps 9 --warmup 3 --no-cpu-baseline --primary-only --no-roofline --launch eager "$@" > /tmp/b_$T.log 2>&1
tail -1 /tmp/b_$T.log | cut -c1-300
db=$(find /tmp/prof_$T -name "*.db" | head -1)
python tools/rocpd_stats.py $db 90 > gpurun_out/${TAG}_kernel_stats.txt
python tools/profile_categories.py gpurun_out/${TAG}_kernel_stats.txt 12 > gpurun_out/${TAG}_kernel_families.txt
python tools/rocpd_queues.py $db 12 > gpurun_out/${TAG}_queues.txt
python tools/rocpd_gaps.py $db 12 30 > gpurun_out/${TAG}_main_queue_gaps.txt
python tools/rocpd_step_seq.py $db 12 > gpurun_out/${TAG}_step_seq.txt
: > gpurun_out/${TAG}_queue_overlap.txt
for k in conv3x3_c32pp pconv_r_kernel bn_bwd_apply col_partial pconv_wgrad gru_bwd_cluster gru_fwd_cluster3 pconv_q_kernel; do
  python tools/rocpd_overlap.py $db 12 $k 2>/dev/null | tail -14 >> gpurun_out/${TAG}_queue_overlap.txt
done
cat gpurun_out/${TAG}_kernel_families.txt; cat gpurun_out/${TAG}_queues.txt | head -8; head -12 gpurun_out/${TAG}_main_queue_gaps.txt
