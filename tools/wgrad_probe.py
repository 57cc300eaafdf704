"""Weight-gradient GEMMs (both operands transposed) of a pack-sized step (K = 11 x 199 rows): split-K and tile choice, us per wgrad
including the slab reduction."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scl_amd import ops
from scl_amd.ops import Op
dev = torch.device("cuda:0")
for K in [int(a) for a in sys.argv[1:]] or [2189]:
    for (Mo, No, name) in ((1024, 1024, "out"), (3072, 1024, "qkv"), (4096, 1024, "fc1"), (1024, 4096, "fc2")):
        sets = [((0.1 * torch.randn(K, Mo, device=dev)).bfloat16(), (0.1 * torch.randn(K, No, device=dev)).bfloat16()) for _ in range(3)]
        out = torch.empty(Mo, No, device=dev)
        slabs = torch.empty(16, Mo, No, device=dev)
        res = []
        for sk, kw, tag in ((1, dict(no_w8=True), "t1"), (2, dict(no_w8=True), "t2"), (4, dict(no_w8=True), "t4"), (8, dict(no_w8=True), "t8"),
                            (1, dict(force_w8=True), "w1"), (2, dict(force_w8=True), "w2"), (4, dict(force_w8=True), "w4"), (8, dict(force_w8=True), "w8")):
            def run(i):
                A, B = sets[i % 3]
                if sk == 1:
                    ops.gemm(Op(A, Mo), Op(B, No), out, Mo, No, K, a_t=True, b_t=True, **kw)
                else:
                    ops.gemm(Op(A, Mo), Op(B, No), slabs, Mo, No, K, a_t=True, b_t=True, splitk=sk, c_split_stride=Mo * No, **kw)
                    ops.reduce_slabs(slabs, out, Mo * No, sk, Mo * No)
            try:
                for i in range(5):
                    run(i)
            except Exception as e:
                res.append("%s n/a" % tag); continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for i in range(40):
                run(i)
            e1.record(); torch.cuda.synchronize()
            res.append("%s %.1f" % (tag, e0.elapsed_time(e1) * 1000 / 40))
        print("K=%d %s [%d x %d]: %s" % (K, name, Mo, No, " | ".join(res)))
