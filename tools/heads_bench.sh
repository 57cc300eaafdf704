mkdir -p gpurun_out
python -m pytest tests/test_resnet_gpu.py tests/test_aasist_gpu.py tests/test_gemm_gpu.py -q -k "not wide and not pingpong and not big_tile" 2>&1 | tail -3
for m in wav2vec2_resnet_nll wav2vec2_aasist; do python bench.py --model $m --batch 32 --rawboost 0 --steps 5 --warmup 2 --no-cpu-baseline 2>gpurun_out/hb_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', 'ms/step %.2f utt/s %.1f'%(d['ms_per_step'], d['value']))"; tail -2 gpurun_out/hb_err.log | cut -c1-300; done
SCL_RESNET_CONV=bf16 python bench.py --model wav2vec2_resnet_nll --batch 32 --rawboost 0 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('resnet bf16 convs', 'ms/step %.2f utt/s %.1f'%(d['ms_per_step'], d['value']))"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for m in wav2vec2_aasist wav2vec2_resnet_nll; do
rm -rf gpurun_out/prof_$m
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$m -o b -- python3 bench.py --model $m --batch 32 --rawboost 0 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 - $m <<'PY'
import csv,glob,sys
f=glob.glob('gpurun_out/prof_%s/**/*kernel_stats.csv'%sys.argv[1],recursive=True)
rows=list(csv.DictReader(open(f[0])))
tot=sum(float(r['TotalDurationNs']) for r in rows); n=4
print(sys.argv[1], "total kernel ms/step %.2f"%(tot/1e6/n), "launches/step", sum(int(r['Calls']) for r in rows)/n)
for r in rows[:26]:
    if 'gemm_w8' in r['Name'] or 'attn' in r['Name'] or 'ln_' in r['Name'] or 'gemm_dma' in r['Name']: continue
    print("  %-90s %6s %8.1f us %7.3f ms/step"%(r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6/n))
PY
cp gpurun_out/prof_$m/b_kernel_stats.csv gpurun_out/r2_bench_${m}_kernel_stats.csv
done
