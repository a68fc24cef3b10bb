# hipGraph replay of the step under the HIP runtime's graph / launch environment switches (ROCm 7.x: strings of libamdhip64.so):
#   bash tools/graph_env_ab.sh  -> ms per step of `bench.py --launch graph` per setting (eager beside it for reference)
run() { env "$@" python bench.py --steps 12 --warmup 4 --no-cpu-baseline --primary-only --no-roofline --launch graph 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.3f ms/step (graph)' % d['ms_per_step'])"; }
echo -n "eager                                   "; python bench.py --steps 12 --warmup 4 --no-cpu-baseline --primary-only --no-roofline --launch eager 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.3f ms/step' % d['ms_per_step'])"
echo -n "graph, defaults                         "; run X=0
for kv in DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_HIP_FORCE_GRAPH_QUEUES=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=2 DEBUG_HIP_FORCE_GRAPH_QUEUES=4 DEBUG_HIP_GRAPH_BATCH_SIZE=1 DEBUG_HIP_GRAPH_BATCH_SIZE=1000 HIP_FORCE_DEV_KERNARG=1 HIP_FORCE_DEV_KERNARG=0; do
  printf "graph, %-33s" "$kv"; run $kv
done
