cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_b16.py tests/test_gpu_engine.py -x -q -m gpu -k "stem or tower or engine or encoder" 2>&1 | tail -4
bash tools/r06_trace.sh r06e_trace > gpurun_out/r06e_trace.log 2>&1
grep -n "stem_wgrad\|colsum\|se_bn" gpurun_out/r06e_trace_kernel_stats.txt | head; head -30 gpurun_out/r06e_trace_kernel_families.txt
