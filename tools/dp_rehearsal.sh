# N = 2 code path of bench.py on a one-GPU box: two processes on cuda:0, gloo transport (RCCL refuses two ranks on one device)
export SCL_BENCH_BACKEND=gloo SCL_BENCH_ONE_DEVICE=1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --batch 16 2>gpurun_out/dp_reh.err | tail -1 | cut -c1-2500
grep -iE "error|Traceback" gpurun_out/dp_reh.err | head -5
