cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out/r6_call4.txt; : > $o
timeout 600 python -m pytest tests/test_btse_gpu.py -x -q 2>&1 | grep -v amdgpu.ids | tail -5 >> $o
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -s -k "trajectory" > gpurun_out/r6_traj_test.log 2>&1
grep -v amdgpu.ids gpurun_out/r6_traj_test.log | grep -A30 "loss rel err" | head -60 >> $o
tail -3 gpurun_out/r6_traj_test.log >> $o
timeout 1500 python tools/data_path_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_pack_builder.txt
cat gpurun_out/r6_pack_builder.txt >> $o
cat $o
