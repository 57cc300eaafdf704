mkdir -p gpurun_out; : > gpurun_out/abl.log
for mode in 1 0; do echo "=== SCL_W8_MODE=$mode" >> gpurun_out/abl.log; SCL_W8_MODE=$mode STAMPS=1 timeout 200 tools/gemm_bench 64 20 2>&1 | sed 's/| t128.*| w8 /| w8 /; s/stamps([0-9]* blocks): //' >> gpurun_out/abl.log; done
