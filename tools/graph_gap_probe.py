"""Kernel-boundary cost: N dependent GEMM launches back to back on a stream vs the same launches replayed as one hipGraph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scl_amd import ops
from scl_amd.ops import Op
dev = torch.device("cuda:0")
M = 12736
def mk(N, K):
    A = (0.1 * torch.randn(M, K, device=dev)).bfloat16(); B = (0.1 * torch.randn(N, K, device=dev)).bfloat16()
    return A, B, torch.empty(M, N, dtype=torch.bfloat16, device=dev)
for (N, K, name) in ((1024, 1024, "out-proj shaped (31 us)"), (1024, 4096, "fc2 fwd shaped (89 us)"), (4096, 1024, "fc1 fwd shaped (112 us)")):
    sets = [mk(N, K) for _ in range(3)]
    NL = 24
    def body():
        for i in range(NL):
            A, B, C = sets[i % 3]
            ops.gemm(Op(A, K), Op(B, K), C, M, N, K)
    for _ in range(3): body()
    torch.cuda.synchronize()
    def timeit(fn):
        ts = []
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1000 / NL)
        return sorted(ts)[2]
    t_plain = timeit(body)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            body()
    torch.cuda.synchronize()
    t_graph = timeit(g.replay)
    print("%-28s plain %.2f us per launch | hipGraph replay %.2f us per launch" % (name, t_plain, t_graph), flush=True)
