import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scl_amd import ops
from scl_amd.ops import Op
dev = torch.device("cuda:0")
bf = lambda *s: (torch.randn(*s, device=dev) * 0.1).to(torch.bfloat16)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 6368
for N in (1024, 4096):
    for big in (False, True):
        row = []
        for K in (64, 256, 512, 1024, 2048, 4096):
            A, B, C = bf(M, K), bf(N, K), torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            t = timeit(lambda: ops.gemm(Op(A, K), Op(B, K), C, M, N, K, no_big=not big))
            row.append("K=%d: %.1fus (%.0f TF)" % (K, t, 2.0 * M * N * K / t / 1e6))
        print("N=%d %s | " % (N, "big256x128" if big else "dma128x128") + " | ".join(row))
A, B, C = bf(128, 64), bf(128, 64), torch.empty(128, 128, dtype=torch.bfloat16, device=dev)
print("single tile K=64: %.1f us" % timeit(lambda: ops.gemm(Op(A, 64), Op(B, 64), C, 128, 128, 64)))
