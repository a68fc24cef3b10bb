export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_v5.json
cut -c1-700 gpurun_out/bench_v5.json
rm -rf /tmp/prof; rocprofv3 --kernel-trace -d /tmp/prof -o v5 -- python3 bench.py --steps 9 --warmup 3 --no-cpu-baseline --primary-only > /tmp/b.log 2>&1
tail -1 /tmp/b.log | cut -c1-200
db=$(find /tmp/prof -name "*.db" | head -1); echo $db
python tools/rocpd_stats.py $db 60 > gpurun_out/stats_v5.txt
python tools/profile_categories.py gpurun_out/stats_v5.txt 12 > gpurun_out/families_v5.txt
cat gpurun_out/families_v5.txt; head -32 gpurun_out/stats_v5.txt | cut -c1-150
