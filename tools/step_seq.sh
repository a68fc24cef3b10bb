export TMPDIR=/tmp
rm -rf /tmp/prof_s; rocprofv3 --kernel-trace -d /tmp/prof_s -o kt -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --primary-only --no-roofline --launch eager "$@" > /tmp/b_s.log 2>&1
db=$(find /tmp/prof_s -name "*.db" | head -1)
python tools/rocpd_step_seq.py $db 9 > gpurun_out/r05_step_seq.txt
wc -l gpurun_out/r05_step_seq.txt
