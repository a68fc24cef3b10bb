cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu -k "gru or GRU or bigru" 2>&1 | tail -4 > gpurun_out/r06c_tests3.txt
python tools/phase_spans.py 2>&1 | grep -v amdgpu > gpurun_out/r06c_phase_spans2.txt
cat gpurun_out/r06c_tests3.txt gpurun_out/r06c_phase_spans2.txt
