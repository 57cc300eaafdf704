python tools/epi_variants_probe.py 2>&1 | grep -v amdgpu
python tools/attn_probe.py 64 2>&1 | grep -v amdgpu
python -m pytest tests/test_gemm_gpu.py tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -1
for i in 1 2; do python bench.py --no-cpu-baseline --steps 12 2>/dev/null | grep "^{\"metric" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d[\"ms_per_step\"],2), round(d[\"value\"],1), round(d[\"roofline\"][\"frac\"],4), d.get(\"final_loss\"))"; done
