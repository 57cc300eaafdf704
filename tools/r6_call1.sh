cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
E=$GRAFT_REPO_ROOT/scl-deepfake-audio-detection_amd/build_exp/libscl_hip.so
# the region-grid tile orders must cover every tile exactly once: the bit-identity tests of the wide kernels under two of them
SCL_LIB_PATH=$E SCL_GEMM_XCD_ROWS=4 SCL_GEMM_GROUP_M=4 timeout 600 python -m pytest tests/test_gemm_gpu.py -x -q -k "wide_tile_kernel_equals or grouped or column_sums" 2>&1 | tail -3 > gpurun_out/r6_xcd_order_tests.txt
SCL_LIB_PATH=$E SCL_GEMM_XCD_ROWS=2 SCL_GEMM_GROUP_M=3 timeout 600 python -m pytest tests/test_gemm_gpu.py -x -q -k "wide_tile_kernel_equals or grouped or column_sums" 2>&1 | tail -3 >> gpurun_out/r6_xcd_order_tests.txt
cat gpurun_out/r6_xcd_order_tests.txt
bash tools/r6_gemm_classes.sh
bash tools/xcd_order_probe.sh
