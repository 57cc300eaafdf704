"""Square-problem check of the GEMM variants (the guide's 256^2 8-phase template is quoted at 4096^3 / 8192^3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scl_amd import ops
from scl_amd.ops import Op

dev = torch.device("cuda:0")
bf = lambda *s: (torch.rand(*s, device=dev) * 2 - 1).to(torch.bfloat16)


def run(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for S in (4096, 8192):
    for at, bt in ((False, False), (False, True), (True, True)):
        A, Bm = bf(S, S), bf(S, S)
        C = torch.zeros(S, S, dtype=torch.bfloat16, device=dev)
        res = []
        for var, kw in (("t128", dict(no_p8=True, no_big=True)), ("big", dict(no_p8=True, force_big=True)), ("p8", dict(force_p8=True))):
            fn = lambda kw=kw: ops.gemm(Op(A, S), Op(Bm, S), C, S, S, S, a_t=at, b_t=bt, **kw)
            fn(); torch.cuda.synchronize()
            t = sorted(run(fn) for _ in range(3))[1]
            res.append("%s %7.1f us %6.0f TF" % (var, t, 2.0 * S ** 3 / t / 1e6))
        tt = sorted(run(lambda: torch.matmul(A, Bm.t() if not bt else Bm)) for _ in range(3))[1]
        res.append("torch %6.0f TF" % (2.0 * S ** 3 / tt / 1e6))
        print("S=%d at=%d bt=%d | %s" % (S, at, bt, " | ".join(res)))
