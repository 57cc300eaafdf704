"""Positional conv (16 groups x 64 channels, 128 taps, T = 199) alone: the LDS-resident-slab kernel (csrc/posconv.hip) against the
grouped GEMM, forward and data-gradient forms, us per launch at batch 64 / 32."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scl_amd import ops
from scl_amd.ops import Op
dev = torch.device("cuda:0")
T, K, G, Cg = 199, 128, 16, 64
E = G * Cg
for B in [int(a) for a in sys.argv[1:]] or [64, 32]:
    M = B * T
    sets = []
    for i in range(3):
        xpad = torch.zeros(B, T + K, E, device=dev); xpad[:, K // 2: K // 2 + T] = 0.5 * torch.randn(B, T, E, device=dev)
        sets.append((torch.cat([xpad.bfloat16().reshape(-1), torch.zeros(65536, dtype=torch.bfloat16, device=dev)]), (0.02 * torch.randn(G, Cg, K * Cg, device=dev)).bfloat16(),
                     torch.randn(M, E, device=dev), torch.empty(M, E, device=dev), torch.empty(M, E, dtype=torch.bfloat16, device=dev)))
    bias = torch.randn(E, device=dev)
    def run(i, kernel, fwd):
        xpad, w, R, C, c2 = sets[i % 3]
        if kernel == "gemm":
            kw = dict(bias=bias, bias_bs2=Cg, act=1, c2=c2) if fwd else {}
            ops.gemm(Op(xpad, E, rpb=T, rbstride=(T + K) * E, cin=Cg, cout=E, bs2=Cg), Op(w, K * Cg, bs2=Cg * K * Cg), C, M, Cg, K * Cg, nb2=G, ldc=E, c_bs2=Cg,
                     R=R, rmode=1, **kw)
        else:
            ops.posconv_mfma(xpad, w, C, R, B, T, K, G, Cg, bias=bias if fwd else None, c2=c2 if fwd else None)
    dy = torch.zeros(B, T + K, E, device=dev); dy[:, K // 2 - 1: K // 2 - 1 + T] = 0.1 * torch.randn(B, T, E, device=dev)
    dyp = torch.cat([dy.bfloat16().reshape(-1), torch.zeros(65536, dtype=torch.bfloat16, device=dev)])
    dw = torch.empty(G, Cg, K * Cg, device=dev)
    slab = torch.empty(4, G * Cg * K * Cg, device=dev)
    def wg(kernel, i):
        xpad = sets[i % 3][0]
        if kernel == "gemm":
            ops.gemm(Op(dyp, E, rpb=T, rbstride=(T + K) * E, bs2=Cg, offset=(K // 2 - 1) * E), Op(xpad, E, rpb=T, rbstride=(T + K) * E, cin=Cg, cout=E, bs2=Cg),
                     dw, Cg, K * Cg, M, a_t=True, b_t=True, nb2=G, c_bs2=Cg * K * Cg, ldc=K * Cg)
        else:
            ops.posconv_wgrad(dyp, K // 2 - 1, xpad, dw, B, T, K, G, Cg)
    res = {}
    for rnd in range(3):
        for kernel in ("gemm", "mfma"):
            for i in range(3): wg(kernel, i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for i in range(20): wg(kernel, i)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(kernel, []).append(e0.elapsed_time(e1) * 1000 / 20)
    fl = 2.0 * M * Cg * K * Cg * G
    print("B=%d weight gradient: " % B + " | ".join("%s %.1f us %.0f TF" % (k, sorted(v)[1], fl / sorted(v)[1] / 1e6) for k, v in res.items()), flush=True)
    for fwd in (True, False):
        res = {}
        for rnd in range(3):
            for kernel in ("gemm", "mfma"):
                for i in range(3): run(i, kernel, fwd)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); e0.record()
                for i in range(20): run(i, kernel, fwd)
                e1.record(); torch.cuda.synchronize()
                res.setdefault(kernel, []).append(e0.elapsed_time(e1) * 1000 / 20)
        fl = 2.0 * M * Cg * K * Cg * G
        print("B=%d %s: " % (B, "forward (bias, gelu, c2, + R)" if fwd else "data gradient (+ R)") +
              " | ".join("%s %.1f us %.0f TF" % (k, sorted(v)[1], fl / sorted(v)[1] / 1e6) for k, v in res.items()), flush=True)
