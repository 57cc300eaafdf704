timeout 900 python -m pytest tests/test_hipnn_gpu.py -q -m gpu -k batch_norm 2>&1 | tail -25 > gpurun_out/_t.log
