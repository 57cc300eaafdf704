"""fp8 vs bf16 fused attention forward at the encoder's shape (B x 16 heads x T = 199, head dim 64): microseconds per launch and the
error of each against fp32 soft-max attention (profiles/r4_attn_fp8_probe.txt)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scl_amd import ops

dev = torch.device("cuda:0")
for B, T, H in ((64, 199, 16), (32, 199, 16), (64, 49, 16)):
    D, E = 64, H * 64
    qkv = (0.7 * torch.randn(B, T, 3, H, D, generator=torch.Generator().manual_seed(1))).to(torch.bfloat16).to(dev)
    ctx = torch.empty(B, T, E, dtype=torch.bfloat16, device=dev); lse = torch.empty(B, H, T, device=dev)
    q, k, v = (qkv[:, :, i].float().permute(0, 2, 1, 3) for i in range(3))
    ref = (torch.softmax((q @ k.transpose(-1, -2)) * D ** -0.5, -1) @ v).permute(0, 2, 1, 3).reshape(B, T, E)
    for name, fn in (("bf16", lambda: ops.attn_fwd(qkv, ctx, lse, B, T, H, D, D ** -0.5)), ("fp8 ", lambda: ops.attn_fwd_fp8(qkv, ctx, lse, B, T, H, D, D ** -0.5))):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record(); torch.cuda.synchronize()
        err = float((ctx.float() - ref).norm() / ref.norm())
        flops = 4.0 * B * H * T * T * D
        us = e0.elapsed_time(e1) * 1e3 / 50
        print("B=%d T=%d H=%d  %s  %7.1f us  %6.1f TFLOP/s  rel-L2 vs fp32 %.2e" % (B, T, H, name, us, flops / us / 1e6, err))
