cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out/r6_call9.txt; : > $o
timeout 600 python -m pytest tests/test_gemm_gpu.py -x -q -k "triple" 2>&1 | grep -v amdgpu.ids | tail -3 >> $o
echo "== vendor library (hipBLASLt through torch.matmul) vs scl_gemm_bf16, plain bf16 outputs, M = 12736 (calibration only)" >> $o
python tools/vendor_vs_ours.py 2>&1 | grep -v amdgpu > gpurun_out/r6_vendor_vs_ours.txt
cat gpurun_out/r6_vendor_vs_ours.txt >> $o
echo "== whole step on the experiment build, groups of 8 rows (shipped) vs 4, interleaved" >> $o
E=$GRAFT_REPO_ROOT/scl-deepfake-audio-detection_amd/build_exp/libscl_hip.so
bash tools/ab_env.sh "SCL_LIB_PATH=$E SCL_GEMM_GROUP_M=8" "SCL_LIB_PATH=$E SCL_GEMM_GROUP_M=4" 3 2>&1 | sed "s#SCL_LIB_PATH=$E ##" >> $o
cat $o
