cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/r06b_ab4.txt
for i in 1 2 3; do
python tools/ab_call.py "pass" >> gpurun_out/r06b_ab4.txt 2>&1
python tools/ab_call.py "ops.DX_FIRST=False" >> gpurun_out/r06b_ab4.txt 2>&1
done
grep -v amdgpu.ids gpurun_out/r06b_ab4.txt
