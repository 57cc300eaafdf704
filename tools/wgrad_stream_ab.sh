# A/B of the second-stream weight gradients: default bench and the batch-32 line, SCL_WGRAD_STREAM=0/1
for w in 0 1 0 1; do
  SCL_WGRAD_STREAM=$w timeout 600 python bench.py --no-cpu-baseline --steps 8 --warmup 3 2>gpurun_out/ws_$w.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('wgrad_stream=$w b64', 'ms/step %.2f utt/s %.1f frac %.4f loss %.5f'%(d['ms_per_step'], d['value'], d['roofline']['frac'], d['final_loss']))"
done
for w in 0 1; do
  SCL_WGRAD_STREAM=$w timeout 600 python bench.py --no-cpu-baseline --batch 32 --rawboost 0 --steps 8 --warmup 3 2>>gpurun_out/ws_$w.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('wgrad_stream=$w b32', 'ms/step %.2f utt/s %.1f frac %.4f loss %.5f'%(d['ms_per_step'], d['value'], d['roofline']['frac'], d['final_loss']))"
done
