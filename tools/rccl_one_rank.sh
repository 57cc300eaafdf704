# default bench with the gradient exchange forced through RCCL on one rank (identity all-reduce), fp32 and bf16 wire, beside the plain run
for mode in plain fp32 bf16; do
  if [ $mode = plain ]; then E=""; else E="SCL_BENCH_FORCE_DP=1 SCL_DP_WIRE=$mode"; fi
  env $E timeout 600 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>gpurun_out/rccl1_$mode.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$mode', 'ms/step %.2f utt/s %.1f'%(d['ms_per_step'], d['value']), json.dumps(d.get('rccl'))[:700])"
done
