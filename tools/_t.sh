python tools/btse_bio_probe.py 2>&1 | grep -v amdgpu > gpurun_out/_t.log
timeout 900 python -m pytest tests/test_btse_gpu.py -q -m gpu 2>&1 | tail -3 >> gpurun_out/_t.log
