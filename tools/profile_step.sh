# Kernel-trace bundle, run on the GPU box from the repo root (writes gpurun_out/, copy the summaries into profiles/):
#   rocprofv3 --kernel-trace of 9 eager primary steps (+ the untimed roofline pass) -> gpurun_out/${TAG}_kernel_stats.txt / _families.txt
# usage: bash tools/profile_step.sh TAG [extra bench args]      (env such as HA2G_PLANES=0 is inherited)
export TMPDIR=/tmp
TAG=${1:-r03}; shift
rm -rf /tmp/prof_$TAG; rocprofv3 --kernel-trace -d /tmp/prof_$TAG -o kt -- python3 bench.py --steps 9 --warmup 3 --no-cpu-baseline --primary-only --no-roofline --launch eager "$@" > /tmp/b_$TAG.log 2>&1
db=$(find /tmp/prof_$TAG -name "*.db" | head -1)
python tools/rocpd_stats.py $db 90 > gpurun_out/${TAG}_kernel_stats.txt
python tools/profile_categories.py gpurun_out/${TAG}_kernel_stats.txt 12 > gpurun_out/${TAG}_kernel_families.txt
python tools/rocpd_queues.py $db 12 > gpurun_out/${TAG}_queues.txt
cat gpurun_out/${TAG}_kernel_families.txt
