"""Fused attention kernels alone at the encoder's shape (T = 199, 16 heads x 64): microseconds per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scl_amd import ops
dev = torch.device("cuda:0")
T, H, D = 199, 16, 64
E = H * D
for B in [int(a) for a in sys.argv[1:]] or [64, 32, 11]:
    sets = []
    for i in range(3):      # rotate buffers: 78 MB of qkv per set at B = 64, never L2-resident
        qkv = (0.5 * torch.randn(B * T, 3 * E, device=dev)).bfloat16()
        ctx = torch.empty(B * T, E, device=dev, dtype=torch.bfloat16)
        lse = torch.empty(B * H * T, device=dev)
        dctx = (0.1 * torch.randn(B * T, E, device=dev)).bfloat16()
        dqkv = torch.empty(B * T, 3 * E, device=dev, dtype=torch.bfloat16)
        sets.append((qkv, ctx, lse, dctx, dqkv))
    def fwd(i):
        qkv, ctx, lse, dctx, dqkv = sets[i % 3]
        ops.attn_fwd(qkv, ctx, lse, B, T, H, D, D ** -0.5)
    def bwd(i):
        qkv, ctx, lse, dctx, dqkv = sets[i % 3]
        ops.attn_bwd(qkv, ctx, dctx, lse, dqkv, B, T, H, D, D ** -0.5)
    for name, f in (("fwd", fwd), ("bwd", bwd)):
        for i in range(6):
            fwd(i) if name == "bwd" else None
            f(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for i in range(60):
            f(i)
        e1.record(); torch.cuda.synchronize()
        print("B=%d attn %s: %.1f us" % (B, name, e0.elapsed_time(e1) * 1000 / 60))
