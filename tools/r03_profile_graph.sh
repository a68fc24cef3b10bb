# Kernel trace of hipGraph REPLAYS of the train step (the headline's launch mode) -> per-queue busy / gap view beside the eager one of
# tools/r03_profile.sh.   usage: bash tools/r03_profile_graph.sh TAG [extra bench args]
export TMPDIR=/tmp
TAG=${1:-r03g}; shift
rm -rf /tmp/prof_$TAG; rocprofv3 --kernel-trace -d /tmp/prof_$TAG -o kt -- python3 bench.py --steps 9 --warmup 3 --no-cpu-baseline --primary-only --no-roofline --launch graph "$@" > /tmp/b_$TAG.log 2>&1
tail -2 /tmp/b_$TAG.log | cut -c1-300
db=$(find /tmp/prof_$TAG -name "*.db" | head -1)
python tools/rocpd_queues.py $db 12 > gpurun_out/${TAG}_queues.txt
cat gpurun_out/${TAG}_queues.txt
