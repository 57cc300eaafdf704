"""Pack-sized linears (M = 11 x 199 rows and its multiples): the two-stage 128 x 128 kernel vs its deep-ring variant (csrc/gemm.hip
scl_gemm_deep_kernel — an -DSCL_EXPERIMENTS build; SCL_GEMM_DEEP is read per call, ring depth = -DSCL_DEEP_STAGES) with and without split-K,
and the wide kernel where it qualifies ("auto").  us per GEMM launch from the launch's own dispatch stamps (split-K: the partial GEMM only; its slab reduction is a separate ~8-us launch).  With the shipped library the
"deep" columns repeat the two-stage kernel.  Result (profiles/r6_small_m_probe.txt): the ring depth buys nothing at any of the shapes."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scl_amd import ops
from scl_amd.ops import Op
dev = torch.device("cuda:0")
KID_GEMM = 0
Ms = [int(a) for a in sys.argv[1:]] or [2189]
ops._AUTO_SPLITK = False
for M in Ms:
    for (N, K, b_t, name) in ((1024, 1024, False, "out fwd"), (1024, 1024, True, "out dgrad"), (1024, 4096, False, "fc2 fwd"), (1024, 3072, True, "qkv dgrad"),
                              (1024, 4096, True, "fc1 dgrad"), (4096, 1024, False, "fc1 fwd"), (4096, 1024, True, "fc2 dgrad"), (3072, 1024, False, "qkv fwd")):
        sets = []
        for i in range(3):
            A = (0.1 * torch.randn(M, K, device=dev)).bfloat16()
            B = (0.1 * torch.randn(K, N, device=dev) if b_t else 0.1 * torch.randn(N, K, device=dev)).bfloat16()
            sets.append((A, B))
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        slabs = torch.empty(8, M, N, device=dev)
        outf = torch.empty(M, N, device=dev)
        res = []
        ref = None
        for deep, sk, no_w8 in ((0, 1, False), (0, 1, True), (1, 1, True), (0, 2, True), (1, 2, True), (0, 3, True), (1, 3, True), (0, 4, True), (1, 4, True)):
            if sk > 1 and K // sk < 512:
                continue
            os.environ["SCL_GEMM_DEEP"] = str(deep)
            def run(i):
                A, B = sets[i % 3]
                if sk == 1:
                    ops.gemm(Op(A, K), Op(B, N if b_t else K), C, M, N, K, b_t=b_t, no_w8=no_w8)
                else:
                    ops.gemm(Op(A, K), Op(B, N if b_t else K), slabs, M, N, K, b_t=b_t, splitk=sk, c_split_stride=M * N, no_w8=no_w8)
                    ops.reduce_slabs(slabs, outf, M * N, sk, M * N)
            for i in range(6):
                run(i)
            if sk == 1:
                got = C.float().clone()
                if ref is None:
                    ref = got
                elif not torch.equal(ref, got):
                    res.append("MISMATCH")
            # kernel time from the launches' own dispatch stamps (the library's profiling scope, as bench.py's roofline uses it): a Python
            # loop of launches this short is HOST-bound (~18 us per ops.gemm call), wall time per iteration says nothing about the kernel
            torch.cuda.synchronize()
            ops.prof_reserve(KID_GEMM, 64)
            ops.prof_read(KID_GEMM)
            ops.prof_enable(KID_GEMM, True)
            for i in range(30):
                run(i)
            torch.cuda.synchronize()
            ops.prof_enable(KID_GEMM, False)
            ms = [m for m, _ in ops.prof_read_launches(KID_GEMM)]
            ops.prof_read(KID_GEMM)
            us = 1e3 * sum(ms) / max(1, len(ms))
            res.append("%ssk%d%s %.1f" % ("deep " if deep else "", sk, "" if no_w8 else " auto", us))
        print("M=%d %-10s N=%d K=%d: %s" % (M, name, N, K, " | ".join(res)))
