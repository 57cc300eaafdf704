import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scl_amd import ops
from scl_amd.ops import Op
dev = torch.device("cuda:0")
M = 12736
bf = lambda *s: torch.empty(*s, dtype=torch.bfloat16, device=dev)
rb = lambda *s: torch.randn(*s, device=dev).bfloat16()
def bench(name, N, K, b_t, variants):
    sets = []
    for i in range(3):
        A = (0.1 * torch.randn(M, K, device=dev)).bfloat16()
        B = (0.1 * torch.randn(K, N, device=dev) if b_t else 0.1 * torch.randn(N, K, device=dev)).bfloat16()
        sets.append((A, B, bf(M, N), bf(M, N), rb(M, N), torch.randn(N, device=dev), torch.empty(62 * 4, N, device=dev)))
    times = {v: [] for v, _ in variants}
    def run(i, mk):
        A, B, C, C2, R, bias, part = sets[i % 3]
        ops.gemm(Op(A, K), Op(B, N if b_t else K), C, M, N, K, b_t=b_t, **mk(C2, R, bias, part))
    for v, mk in variants:
        for i in range(3): run(i, mk)
    for r in range(5):
        for v, mk in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for i in range(12): run(i, mk)
            e1.record(); torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) * 1000 / 12)
    print("%-10s" % name + " | ".join("%s %.1f us" % (v, sorted(times[v])[2]) for v, _ in variants), flush=True)
bench("fc1 fwd", 4096, 1024, False, [("plain", lambda C2, R, b, p: {}), ("bias", lambda C2, R, b, p: dict(bias=b)), ("bias+gelu", lambda C2, R, b, p: dict(bias=b, act=1)),
                                     ("bias+gelu+c2", lambda C2, R, b, p: dict(bias=b, act=1, c2=C2)), ("bias+gelu+dc2", lambda C2, R, b, p: dict(bias=b, act=5, c2=C2)),
                                     ("bias+c2 only", lambda C2, R, b, p: dict(bias=b, c2=C2))])
bench("fc2 dgrad", 4096, 1024, True, [("plain", lambda C2, R, b, p: {}), ("x gelu'(R)", lambda C2, R, b, p: dict(R=R, rmode=2, ract=1)), ("x R", lambda C2, R, b, p: dict(R=R, rmode=2, ract=4)),
                                      ("x gelu'(R) + colsum", lambda C2, R, b, p: dict(R=R, rmode=2, ract=1, colsum_part=p)), ("x R + colsum", lambda C2, R, b, p: dict(R=R, rmode=2, ract=4, colsum_part=p)),
                                      ("+ R (bf16)", lambda C2, R, b, p: dict(R=R, rmode=1))])
