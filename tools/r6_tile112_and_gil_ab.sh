cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out/r6_call5.txt; : > $o
echo "== gemm tests (112-row tile)" >> $o
timeout 1500 python -m pytest tests/test_gemm_gpu.py -x -q 2>&1 | grep -v amdgpu.ids | tail -8 >> $o
timeout 900 python -m pytest tests/test_model_gpu.py -x -q 2>&1 | grep -v amdgpu.ids | tail -4 >> $o
echo "== batch 32 (configs[1]) and pack 11, 112-row tiles off / on, interleaved" >> $o
bash tools/ab_env.sh "SCL_W8_TILE112=0" "SCL_W8_TILE112=1" 3 --batch 32 --rawboost 0 >> $o 2>&1
bash tools/ab_env.sh "SCL_W8_TILE112=0" "SCL_W8_TILE112=1" 2 --batch 11 --rawboost 0 --steps 10 >> $o 2>&1
bash tools/ab_env.sh "SCL_W8_TILE112=0" "SCL_W8_TILE112=1" 2 >> $o 2>&1
echo "== tools/gemm_bench 32 20 (automatic choice vs 128 x 128)" >> $o
for c in "out fwd" "fc2 fwd" "out dgrad" "fc1 dgrad" "qkv dgrad"; do tools/gemm_bench 32 20 "$c" 2>&1 | grep -v "^case" >> $o; done
echo "== data path, interpreter lock kept by launch calls (default) / released (round 5)" >> $o
for g in hold release; do echo "SCL_CTYPES_GIL=$g" >> $o; SCL_CTYPES_GIL=$g PROBE_PARTS=23 timeout 900 python tools/data_path_probe.py 2>&1 | grep -v "amdgpu.ids\|Scores saved\|vocoders" >> $o; done
cat $o
