mkdir -p gpurun_out
for w in 0 1; do SCL_GEMM_W8=$w timeout 300 python bench.py --batch 64 --rawboost 5 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/e2e_w8_$w.json 2> gpurun_out/e2e_w8_$w.err; done
SCL_GEMM_W8=1 timeout 300 python bench.py --batch 32 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/e2e_b32_w8_1.json 2>> gpurun_out/e2e_w8_1.err
SCL_GEMM_W8=0 timeout 300 python bench.py --batch 32 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/e2e_b32_w8_0.json 2>> gpurun_out/e2e_w8_0.err
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/pt_all.log 2>&1; echo pytest rc=$?; tail -5 gpurun_out/pt_all.log
