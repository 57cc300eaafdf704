python tools/conv0_probe.py 2>&1 | grep conv0 > gpurun_out/_t.log
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "conv0" 2>&1 | tail -2 >> gpurun_out/_t.log
