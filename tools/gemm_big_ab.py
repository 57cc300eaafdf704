"""A/B of the 256x128 3-stage ring kernel (default for >= 1000 tiles) against the 128x128 LDS-DMA kernel on the conv-stack shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scl_amd import ops
from scl_amd.ops import Op

dev = torch.device("cuda:0")
bf = lambda *s: (torch.randn(*s, device=dev) * 0.1).to(torch.bfloat16)


def run(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


B, C = 32, 512
Ts = [12799, 6399, 3199, 1599, 799]
for i in range(1, 5):
    Tin, Tout, k = Ts[i - 1], Ts[i], 3
    Mi = B * Tout
    z = bf(B * Tin * C + 65536); wk = bf(C, k * C); y = torch.empty(Mi, C, dtype=torch.float32, device=dev)
    dy = bf(Mi, C); dcol = torch.empty(Mi, k * C, dtype=torch.bfloat16, device=dev)
    cases = {"fwd": lambda kw: ops.gemm(Op(z, 2 * C, rpb=Tout, rbstride=Tin * C), Op(wk, k * C), y, Mi, C, k * C, **kw),
             "dgrad": lambda kw: ops.gemm(Op(dy, C), Op(wk, k * C), dcol, Mi, k * C, C, b_t=True, **kw)}
    for name, fn in cases.items():
        fl = 2.0 * Mi * C * k * C
        res = {}
        for var, kw in (("t128", dict(no_big=True)), ("big", dict(force_big=True))):
            fn(kw); torch.cuda.synchronize()
        for _ in range(5):
            for var, kw in (("t128", dict(no_big=True)), ("big", dict(force_big=True))):
                res.setdefault(var, []).append(run(lambda: fn(kw)))
        med = {v: sorted(x)[len(x) // 2] for v, x in res.items()}
        print("conv%d %-5s M=%6d | t128 %7.1f us %5.0f TF | big %7.1f us %5.0f TF | big/t128 speed x%.2f" % (
            i, name, Mi, med["t128"], fl / med["t128"] / 1e6, med["big"], fl / med["big"] / 1e6, med["t128"] / med["big"]))
