timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q 2>&1 | tail -2
python tools/epilogue_probe.py 2>&1 | grep "dgrad shape" > /tmp/new.txt
b() { python bench.py --no-cpu-baseline --steps 8 --warmup 3 $2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2: ms/step %.2f utt/s %.1f frac %.4f loss %.5f'%(d['ms_per_step'], d['value'], d['roofline']['frac'], d['final_loss']))"; }
b new ""; b new ""
cp tools/_w8_prev.hip.txt scl-deepfake-audio-detection_amd/csrc/gemm_w8.hip
(cd scl-deepfake-audio-detection_amd && python build.py 2>&1 | tail -1)
python tools/epilogue_probe.py 2>&1 | grep "dgrad shape" > /tmp/prev.txt
paste -d'|' /tmp/prev.txt /tmp/new.txt | awk -F'|' '{ split($1,a,":"); n=split($2,b," "); print a[1] ":" a[2] ":" a[3] "   || pk gelu: " b[n-3] " us" }'
b prev ""; b prev ""
