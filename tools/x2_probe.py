"""A/B of the two-blocks-per-CU GEMM (gemm_x2.hip, 208 x 128 tiles) against the automatic choice (wide 208 x 256 tiles / 128 x 128) on the
encoder's shapes and epilogues at batch 64 (M = 12736) or the M given: us per launch, interleaved rounds in one process, operands and
outputs rotated over three buffer sets."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scl_amd import ops
from scl_amd.ops import Op
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 12736
ROUNDS, PER = 5, 12

bf = lambda *s: torch.empty(*s, dtype=torch.bfloat16, device=dev)
f32 = lambda *s: torch.empty(*s, device=dev)
rb = lambda *s: torch.randn(*s, device=dev).bfloat16()


def bench(name, N, K, b_t, make_kw, variants):
    sets = []
    for i in range(3):
        A = (0.1 * torch.randn(M, K, device=dev)).bfloat16()
        B = (0.1 * torch.randn(K, N, device=dev) if b_t else 0.1 * torch.randn(N, K, device=dev)).bfloat16()
        sets.append((A, B, make_kw()))
    def run(i, sel):
        A, B, (C, kw) = sets[i % 3]
        ops.gemm(Op(A, K), Op(B, N if b_t else K), C, M, N, K, b_t=b_t, **kw, **sel)
    times = {v: [] for v, _ in variants}
    for v, sel in variants:
        for i in range(3):
            run(i, sel)
    for r in range(ROUNDS):
        for v, sel in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for i in range(PER):
                run(i, sel)
            e1.record(); torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) * 1000 / PER)
    out = "%-44s N=%-4d K=%-4d" % (name, N, K)
    for v, _ in variants:
        t = sorted(times[v])[len(times[v]) // 2]
        out += " | %s %7.1f us %5.0f TF" % (v, t, 2.0 * M * N * K / t / 1e6)
    print(out, flush=True)


V = [("auto", dict(no_x2=True)), ("x2", dict(force_x2=True))]
for N, K, b_t, tag in ((4096, 1024, False, "fc1 fwd"), (4096, 1024, True, "fc2 dgrad"), (1024, 4096, False, "fc2 fwd"), (1024, 4096, True, "fc1 dgrad"),
                       (1024, 1024, False, "out fwd"), (1024, 1024, True, "out dgrad"), (3072, 1024, False, "qkv fwd"), (1024, 3072, True, "qkv dgrad")):
    bench(tag + ": plain bf16 store", N, K, b_t, lambda: (bf(M, N), {}), V)
    if tag == "fc1 fwd":
        bench(tag + ": + bias, gelu, c2", N, K, b_t, lambda: (bf(M, N), dict(bias=torch.randn(N, device=dev), act=1, c2=bf(M, N))), V)
    if tag == "fc2 dgrad":
        bench(tag + ": * gelu'(R bf16)", N, K, b_t, lambda: (bf(M, N), dict(R=rb(M, N), rmode=2, ract=1)), V)
    if tag in ("fc2 fwd", "out fwd"):
        bench(tag + ": f32 out + bias + f32 residual", N, K, b_t, lambda: (f32(M, N), dict(bias=torch.randn(N, device=dev), R=torch.randn(M, N, device=dev), rmode=1)), V)
    if tag == "qkv fwd":
        bench(tag + ": + bias", N, K, b_t, lambda: (bf(M, N), dict(bias=torch.randn(N, device=dev))), V)
