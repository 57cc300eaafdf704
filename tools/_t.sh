for i in 1 2; do python bench.py --no-cpu-baseline --model wav2vec2_btse --batch 128 --rawboost 0 --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('btse b128', d['ms_per_step'], d['value'])"; done > gpurun_out/_t.log
python bench.py --no-cpu-baseline --model wav2vec2_btse --batch 64 --rawboost 0 --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('btse b64', d['ms_per_step'], d['value'])" >> gpurun_out/_t.log
python tools/btse_bio_probe.py 2>&1 | grep -v amdgpu > gpurun_out/r5_btse_bio_probe_final.txt
timeout 900 python -m pytest tests/test_btse_gpu.py tests/test_pack_gpu.py -q -m gpu 2>&1 | tail -2 >> gpurun_out/_t.log
