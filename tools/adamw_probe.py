"""AdamW over the flat 315 M-parameter buffer: ms per launch and effective HBM rate (30 B per parameter)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scl_amd import ops
dev = torch.device("cuda:0")
n = 315_438_720
p = torch.randn(n, device=dev) * 0.02; g = torch.randn(n, device=dev) * 1e-3
m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev); pb = torch.empty(n, dtype=torch.bfloat16, device=dev)
for s in range(1, 4): ops.adamw_flat(p, g, m, v, pb, n, 1e-5, 0.9, 0.999, 1e-8, 1e-4, s)
ts = []
for r in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for s in range(4, 9): ops.adamw_flat(p, g, m, v, pb, n, 1e-5, 0.9, 0.999, 1e-8, 1e-4, s)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 5)
t = sorted(ts)[2]
print("adamw %d params: %.3f ms  %.2f TB/s  checksum %.9g" % (n, t, n * 30 / t / 1e9, float(p[:1000].double().sum())))
