cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_blocks.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r06b_tests2.txt
rm -f gpurun_out/r06b_ab2.txt
for i in 1 2 3; do
python tools/ab_call.py "pass" >> gpurun_out/r06b_ab2.txt 2>&1
python tools/ab_call.py "we.IMAGE_STATS=False" >> gpurun_out/r06b_ab2.txt 2>&1
done
cat gpurun_out/r06b_tests2.txt; grep -v amdgpu.ids gpurun_out/r06b_ab2.txt
