cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out/r6_call8.txt; : > $o
timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q -k "triple or layernorm_writes or epilogue_kinds" 2>&1 | grep -v amdgpu.ids | tail -5 >> $o
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_aasist_gpu.py -x -q 2>&1 | grep -v amdgpu.ids | tail -5 >> $o
echo "== bench.py --eval, f32-pair kernel (0) vs triple-plane bf16 GEMMs (1), interleaved" >> $o
for i in 1 2; do for v in 0 1; do
  SCL_SCORE_X3PLANES=$v python bench.py --eval --steps 5 --warmup 2 2>/dev/null | grep '^{"metric"' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('SCL_SCORE_X3PLANES=$v  fp32 path %.2f ms = %.0f utt/s | bf16 kernels %.2f ms = %.0f utt/s | max |logprob diff| %.2e' % (d['fp32']['ms_per_batch'], d['fp32']['utterances_per_s'], d['bf16']['ms_per_batch'], d['bf16']['utterances_per_s'], d['bf16_vs_fp32']['max_abs_logprob_diff']))" >> $o
done; done
python bench.py --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1 | cut -c1-200 >> $o
cat $o
