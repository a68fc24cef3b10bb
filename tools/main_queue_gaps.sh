export TMPDIR=/tmp
rm -rf /tmp/prof_g; rocprofv3 --kernel-trace -d /tmp/prof_g -o kt -- python3 bench.py --steps 9 --warmup 3 --no-cpu-baseline --primary-only --no-roofline --launch eager > /tmp/b_g.log 2>&1
db=$(find /tmp/prof_g -name "*.db" | head -1)
python tools/rocpd_gaps.py $db 12 30 > gpurun_out/r05_main_queue_gaps.txt
cat gpurun_out/r05_main_queue_gaps.txt
