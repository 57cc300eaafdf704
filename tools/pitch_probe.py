"""Does the row pitch of the GEMM operands matter (L2 channel interleave)?  fc1-forward / fc2-data-gradient shaped launches with the
operands' rows padded by a few hundred bytes, against the dense power-of-two pitch.  us per launch, median of 5 rounds, 3 buffer sets."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scl_amd import ops
from scl_amd.ops import Op
dev = torch.device("cuda:0")
M = 12736
def bench(name, N, K, b_t, pads):
    times = {p: [] for p in pads}
    sets = {}
    for pa, pb, pc in pads:
        s = []
        for i in range(3):
            A = torch.zeros(M, K + pa, device=dev, dtype=torch.bfloat16); A[:, :K] = (0.1 * torch.randn(M, K, device=dev)).bfloat16()
            if b_t:
                B = torch.zeros(K, N + pb, device=dev, dtype=torch.bfloat16); B[:, :N] = (0.1 * torch.randn(K, N, device=dev)).bfloat16()
            else:
                B = torch.zeros(N, K + pb, device=dev, dtype=torch.bfloat16); B[:, :K] = (0.1 * torch.randn(N, K, device=dev)).bfloat16()
            C = torch.empty(M, N + pc, dtype=torch.bfloat16, device=dev)
            s.append((A, B, C))
        sets[(pa, pb, pc)] = s
    def run(p, i):
        A, B, C = sets[p][i % 3]
        ops.gemm(Op(A, K + p[0]), Op(B, (N if b_t else K) + p[1]), C, M, N, K, b_t=b_t, ldc=N + p[2])
    for p in pads:
        for i in range(3): run(p, i)
    for r in range(5):
        for p in pads:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for i in range(12): run(p, i)
            e1.record(); torch.cuda.synchronize()
            times[p].append(e0.elapsed_time(e1) * 1000 / 12)
    print("%-26s" % name + " | ".join("pad A%d B%d C%d: %.1f us" % (p[0], p[1], p[2], sorted(times[p])[2]) for p in pads), flush=True)
pads = [(0, 0, 0), (64, 64, 0), (64, 64, 64), (128, 128, 0), (32, 32, 0), (64, 0, 0), (0, 64, 0), (0, 0, 0)]
bench("warm-up", 4096, 1024, False, pads[:2])
bench("fc1 fwd  N4096 K1024 NN", 4096, 1024, False, pads)
bench("fc2 dgrad N4096 K1024 NT", 4096, 1024, True, pads)
bench("fc2 fwd  N1024 K4096 NN", 1024, 4096, False, pads)
bench("out fwd  N1024 K1024 NN", 1024, 1024, False, pads)
