mkdir -p gpurun_out; : > gpurun_out/abl3.log
for dbg in 0 16; do echo "=== SCL_W8_DEBUG=$dbg (16 = no stagger)" >> gpurun_out/abl3.log; SCL_W8_MODE=1 SCL_W8_DEBUG=$dbg STAMPS=1 timeout 200 tools/gemm_bench 64 20 2>&1 | sed 's/| t128.*| w8 /| w8 /; s/stamps([0-9]* blocks): //' >> gpurun_out/abl3.log; done
timeout 250 python -m pytest tests/test_gemm_gpu.py -x -q -k "wide" 2>&1 | tail -2 >> gpurun_out/abl3.log
