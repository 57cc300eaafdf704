cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out/r6_call7.txt; : > $o
timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q -k "triple or layernorm_writes" 2>&1 | grep -v amdgpu.ids | tail -5 >> $o
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "scoring or eval or fp32" 2>&1 | grep -v amdgpu.ids | tail -5 >> $o
timeout 900 python -m pytest tests/test_augment_gpu.py tests/test_pack_gpu.py -x -q -s -k "reverb or pack or conf5" 2>&1 | grep -v amdgpu.ids | grep "reverb\|passed\|failed\|Error" | tail -8 >> $o
python - >> $o 2>&1 <<'PY'
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from scl_amd import augment
dev = torch.device("cuda:0")
rs = np.random.RandomState(0)
x = torch.from_numpy((0.1 * rs.randn(64000)).astype(np.float32)).to(dev)
rir = torch.from_numpy((np.exp(-np.arange(8000) / 1200.0) * rs.randn(8000) * 0.3).astype(np.float32)).to(dev)
for flag in (False, True, False, True):
    augment.RIR_GEMM = flag
    for _ in range(3): augment.reverb(x, rir)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(50): y = augment.reverb(x, rir)
    torch.cuda.synchronize()
    print("augment.reverb, one 64000-sample clip x 8000-tap RIR, %s: %.1f us per call (host + device, 50 calls back to back)" % ("f32-MFMA GEMM form" if flag else "fir_kernel", (time.time() - t0) / 50 * 1e6))
PY
echo "== bench.py --eval, f32-pair kernel (0) vs triple-plane bf16 GEMMs (1), interleaved" >> $o
for i in 1 2; do for v in 0 1; do
  SCL_SCORE_X3PLANES=$v python bench.py --eval --steps 5 --warmup 2 2>/dev/null | grep '^{"metric"' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('SCL_SCORE_X3PLANES=$v  fp32 path %.2f ms = %.0f utt/s | bf16 kernels %.2f ms = %.0f utt/s | max |logprob diff| %.2e' % (d['fp32']['ms_per_batch'], d['fp32']['utterances_per_s'], d['bf16']['ms_per_batch'], d['bf16']['utterances_per_s'], d['bf16_vs_fp32']['max_abs_logprob_diff']))" >> $o
done; done
cat $o
