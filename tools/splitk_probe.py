"""Pack-sized GEMMs (M = 11 x 199 rows): the 128x128 kernel alone vs split-K + slab reduction, per shape.  Decides the split rule
of ops.gemm's small-M path."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scl_amd import ops
from scl_amd.ops import Op
dev = torch.device("cuda:0")
Ms = [int(a) for a in sys.argv[1:]] or [2189]
for M in Ms:
    for (N, K, b_t, name) in ((1024, 1024, False, "out fwd"), (1024, 4096, False, "fc2 fwd"), (1024, 3072, True, "qkv dgrad"), (1024, 4096, True, "fc1 dgrad"),
                              (4096, 1024, False, "fc1 fwd"), (3072, 1024, False, "qkv fwd")):
        sets = []
        for i in range(3):
            A = (0.1 * torch.randn(M, K, device=dev)).bfloat16()
            B = (0.1 * torch.randn(K, N, device=dev) if b_t else 0.1 * torch.randn(N, K, device=dev)).bfloat16()
            sets.append((A, B))
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        slabs = torch.empty(8, M, N, device=dev)
        outf = torch.empty(M, N, device=dev)
        res = []
        for sk, no_w8 in ((1, False), (1, True), (2, True), (3, True), (4, True), (2, False), (3, False), (4, False), (6, True), (8, True)):
            if K // sk < 512:
                continue
            def run(i):
                A, B = sets[i % 3]
                if sk == 1:
                    ops.gemm(Op(A, K), Op(B, N if b_t else K), C, M, N, K, b_t=b_t, no_w8=no_w8)
                else:
                    ops.gemm(Op(A, K), Op(B, N if b_t else K), slabs, M, N, K, b_t=b_t, splitk=sk, c_split_stride=M * N, no_w8=no_w8)
                    ops.reduce_slabs(slabs, outf, M * N, sk, M * N)
            for i in range(5):
                run(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for i in range(40):
                run(i)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1000 / 40
            res.append("sk%d%s %.1f" % (sk, "" if no_w8 else "w", us))
        print("M=%d %-10s N=%d K=%d: %s  (us per GEMM incl. reduce)" % (M, name, N, K, " | ".join(res)))
