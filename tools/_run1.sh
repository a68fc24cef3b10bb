cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_pointwise.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r06b_tests1.txt
for i in 1 2 3; do
python tools/ab_call.py "pass" >> gpurun_out/r06b_ab1.txt 2>&1
python tools/ab_call.py "we.SE_BN2_FUSED=False" >> gpurun_out/r06b_ab1.txt 2>&1
python tools/ab_call.py "ops.FUSE_IM2COL_DROPOUT=False" >> gpurun_out/r06b_ab1.txt 2>&1
done
cat gpurun_out/r06b_tests1.txt; cat gpurun_out/r06b_ab1.txt
