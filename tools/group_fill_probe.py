"""Grouped weight-gradient launch: 192 tiles (a layer's four gradients) against 256 tiles (a full round of the CUs) at K = 12736."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scl_amd import ops
from scl_amd.ops import Op
dev = torch.device("cuda:0")
K, E, Fd = 12736, 1024, 4096
def prob(Mo, No, seed):
    g = torch.Generator().manual_seed(seed)
    A = (torch.randn(K, Mo, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    B = (torch.randn(K, No, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    return (Op(A, Mo), Op(B, No), torch.empty(Mo, No, device=dev), Mo, No, K)
sets = {"192 tiles (fc2, fc1, out, qkv)": [prob(E, Fd, 1), prob(Fd, E, 2), prob(E, E, 3), prob(3 * E, E, 4)],
        "256 tiles (fc2, fc1, fc2, fc1)": [prob(E, Fd, 1), prob(Fd, E, 2), prob(E, Fd, 5), prob(Fd, E, 6)],
        "128 tiles (fc2, fc1)": [prob(E, Fd, 1), prob(Fd, E, 2)]}
for name, ps in sets.items():
    for _ in range(3): assert ops.gemm_group(ps)
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): ops.gemm_group(ps)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    t = sorted(ts)[2]
    tiles = sum(((p[3] + 255) // 256) * ((p[4] + 255) // 256) for p in ps)
    fl = sum(2.0 * p[3] * p[4] * K for p in ps)
    print("%-34s %7.1f us  %6.0f TFLOP/s  %.2f us per tile" % (name, t, fl / t / 1e6, t / tiles))
