"""conv0 (Conv1d(1, 512, 10, 5) + LayerNorm + GELU fused) forward / backward alone at batch 64 x 64000 samples: us per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scl_amd import ops
dev = torch.device("cuda:0")
B, L, C, k, s = 64, 64000, 512, 10, 5
T0 = (L - k) // s + 1
x = 0.1 * torch.randn(B, L, device=dev); w = 0.3 * torch.randn(C, k, device=dev); b = torch.randn(C, device=dev)
g = torch.randn(C, device=dev); be = torch.randn(C, device=dev)
z = torch.empty(B * T0 * C, dtype=torch.bfloat16, device=dev); stats = torch.empty(2 * B * T0, device=dev)
dz = (0.1 * torch.randn(B * T0 * C, device=dev)).bfloat16()
ws = torch.empty(ops.conv0_bwd_nparts(B, L, k, s) * C * (k + 3), device=dev)
dW = torch.empty(C, k, device=dev); db = torch.empty(C, device=dev); dg = torch.empty(C, device=dev); dbe = torch.empty(C, device=dev)
def t(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
for r in range(3):
    print("conv0 fwd %.1f us | bwd %.1f us" % (t(lambda: ops.conv0_fwd(x, w, b, g, be, z, B, L, C, k, s, stats=stats)),
                                             t(lambda: ops.conv0_bwd(x, w, b, g, be, dz, ws, dW, db, dg, dbe, B, L, C, k, s, stats=stats))), flush=True)
