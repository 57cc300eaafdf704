"""What the fused epilogues of the wide GEMM cost at batch 64 (M = 12736): the same [M, N, K] contraction with a plain bf16 store and
with each of the encoder's epilogues.  us per launch, operands rotated over three buffer sets."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scl_amd import ops
from scl_amd.ops import Op
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 12736

def bench(name, N, K, b_t, make_kw):
    sets = []
    for i in range(3):
        A = (0.1 * torch.randn(M, K, device=dev)).bfloat16()
        B = (0.1 * torch.randn(K, N, device=dev) if b_t else 0.1 * torch.randn(N, K, device=dev)).bfloat16()
        sets.append((A, B, make_kw()))
    def run(i):
        A, B, (C, kw) = sets[i % 3]
        ops.gemm(Op(A, K), Op(B, N if b_t else K), C, M, N, K, b_t=b_t, **kw)
    for i in range(6):
        run(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(30):
        run(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 30
    print("%-46s N=%d K=%d: %7.1f us  %6.0f TFLOP/s" % (name, N, K, us, 2.0 * M * N * K / us / 1e6))

bf = lambda *s: torch.empty(*s, dtype=torch.bfloat16, device=dev)
f32 = lambda *s: torch.empty(*s, device=dev)
rb = lambda *s: torch.randn(*s, device=dev).bfloat16()
for N, K, b_t, tag in ((4096, 1024, False, "fc1 fwd shape"), (4096, 1024, True, "fc2 dgrad shape"), (1024, 4096, False, "fc2 fwd shape"), (1024, 1024, False, "out fwd shape"),
                       (3072, 1024, False, "qkv fwd shape")):
    bench(tag + ": plain bf16 store", N, K, b_t, lambda: (bf(M, N), {}))
    bench(tag + ": + bias", N, K, b_t, lambda: (bf(M, N), dict(bias=torch.randn(N, device=dev))))
    bench(tag + ": + bias, gelu", N, K, b_t, lambda: (bf(M, N), dict(bias=torch.randn(N, device=dev), act=1)))
    bench(tag + ": + bias, gelu, c2 (fc1 fwd)", N, K, b_t, lambda: (bf(M, N), dict(bias=torch.randn(N, device=dev), act=1, c2=bf(M, N))))
    bench(tag + ": * gelu'(R bf16) (fc2 dgrad)", N, K, b_t, lambda: (bf(M, N), dict(R=rb(M, N), rmode=2, ract=1)))
    bench(tag + ": + R bf16 (residual add)", N, K, b_t, lambda: (bf(M, N), dict(R=rb(M, N), rmode=1)))
    bench(tag + ": f32 out + bias + f32 residual", N, K, b_t, lambda: (f32(M, N), dict(bias=torch.randn(N, device=dev), R=torch.randn(M, N, device=dev), rmode=1)))
    bench(tag + ": f32 out", N, K, b_t, lambda: (f32(M, N), {}))
