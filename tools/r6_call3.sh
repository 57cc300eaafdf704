cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out/r6_call3.txt; : > $o
timeout 600 python -m pytest tests/test_btse_gpu.py -x -q 2>&1 | grep -v amdgpu.ids | tail -30 >> $o
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -s -k "trajectory" 2>&1 | grep -v amdgpu.ids | tail -45 >> $o
timeout 900 python -m pytest tests/test_pack_gpu.py tests/test_augment_gpu.py -x -q 2>&1 | tail -4 >> $o
timeout 1200 python tools/data_path_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_pack_builder.txt
cat gpurun_out/r6_pack_builder.txt >> $o
cat $o
