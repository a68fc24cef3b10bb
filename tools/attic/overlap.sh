export TMPDIR=/tmp
rm -rf /tmp/prof; rocprofv3 --kernel-trace -d /tmp/prof -o v -- python3 bench.py --steps 9 --warmup 3 --no-cpu-baseline --primary-only > /tmp/b.log 2>&1
db=$(find /tmp/prof -name "*.db" | head -1)
python tools/rocpd_overlap.py $db 0.3
