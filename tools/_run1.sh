cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_blocks.py -x -q -m gpu -k "relu_decisions" 2>&1 | tail -12
