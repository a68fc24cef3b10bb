cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_step.py tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r06f_tests.txt
rm -f gpurun_out/r06f_ab.txt
for i in 1 2 3; do
python tools/ab_call.py "pass" >> gpurun_out/r06f_ab.txt 2>&1
python tools/ab_call.py "we.SE_WGRAD_BATCH=False" >> gpurun_out/r06f_ab.txt 2>&1
done
cat gpurun_out/r06f_tests.txt; grep -v amdgpu.ids gpurun_out/r06f_ab.txt
