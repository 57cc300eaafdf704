python -m pytest tests/test_resnet_gpu.py tests/test_aasist_gpu.py -q 2>&1 | tail -2
for m in wav2vec2_aasist wav2vec2_resnet_nll; do python bench.py --model $m --batch 32 --rawboost 0 --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', 'ms/step %.2f utt/s %.1f'%(d['ms_per_step'], d['value']))"; done
