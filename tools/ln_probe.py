"""LayerNorm forward / backward alone at the encoder's shapes: us per launch and effective HBM rate (algorithmic bytes / time).
The rows-per-wave choice is fixed (2) since round 4."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scl_amd import ops
dev = torch.device("cuda:0")
for M, C in ((12736, 1024), (64 * 3199, 512), (6368, 1024)):
    sets = []
    for i in range(3):
        x = torch.randn(M, C, device=dev)
        sets.append((x, torch.empty(M, C, dtype=torch.bfloat16, device=dev), torch.empty(M, device=dev), torch.empty(M, device=dev)))
    g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    def fwd(i):
        x, y, mu, rs = sets[i % 3]
        ops.layernorm_fwd(x, g, b, y, None, mu, rs, M, C)
    for i in range(6): fwd(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(60): fwd(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 60
    print("SCL_LN_ROWS=%s ln_fwd f32->bf16 M=%d C=%d: %.1f us  %.2f TB/s" % ("2", M, C, us, M * C * 6 / us / 1e6))
    # backward at the encoder's configuration: x f32, dy bf16, residual gradient f32 in, dx f32 + bf16 out, dgamma / dbeta / colsum(dres) partials
    nparts = ops.layernorm_bwd_nparts(M)
    bsets = []
    for i in range(3):
        bsets.append((torch.randn(M, C, device=dev).bfloat16(), torch.randn(M, C, device=dev), torch.randn(M, C, device=dev), torch.empty(M, C, device=dev),
                      torch.empty(M, C, dtype=torch.bfloat16, device=dev), torch.empty(nparts, 3 * C, device=dev)))
    mu, rs = sets[0][2], sets[0][3]
    def bwd(i):
        dy, x, dres, dxf, dxb, part = bsets[i % 3]
        ops.layernorm_bwd(dy, x, mu, rs, g, b, dres, dxf, dxb, part, M, C, sum_dres=True)
    for i in range(6): bwd(i)
    torch.cuda.synchronize(); e0.record()
    for i in range(60): bwd(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 60
    print("ln_bwd (x f32, dy bf16, dres f32 -> dx f32 + bf16) M=%d C=%d: %.1f us  %.2f TB/s" % (M, C, us, M * C * 16 / us / 1e6))
