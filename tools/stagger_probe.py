"""A/B of a start stagger of every other XCD (SCL_W8_STAGGER, units of ~3.9 us) -- derived from: A/B of the persistent wide-tile blocks (gemm_w8.hip, w8p) against one-tile blocks (SCL_GEMM_PERSIST=0) on the encoder's multi-round
shapes and epilogues at batch 64 (M = 12736) or the M given: us per launch, interleaved rounds in one process, operands and outputs
rotated over three buffer sets.  Environment is read per launch, so both variants run in the same process."""
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


def bench(name, N, K, b_t, make_kw, variants, Mrows=None, a_op=None):
    Mr = Mrows or M
    sets = []
    for i in range(3):
        A = (0.1 * torch.randn(Mr, K, device=dev)).bfloat16()
        B = (0.1 * torch.randn(K, N, device=dev) if b_t else 0.1 * torch.randn(N, K, device=dev)).bfloat16()
        sets.append((A, B, make_kw()))
    def run(i, env):
        A, B, (C, kw) = sets[i % 3]
        os.environ["SCL_W8_STAGGER"] = env
        ops.gemm(Op(A, K), Op(B, N if b_t else K), C, Mr, N, K, b_t=b_t, **kw)
    times = {v: [] for v, _ in variants}
    for v, env in variants:
        for i in range(3):
            run(i, env)
    for r in range(ROUNDS):
        for v, env in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for i in range(PER):
                run(i, env)
            e1.record(); torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) * 1000 / PER)
    out = "%-44s M=%-6d N=%-4d K=%-4d" % (name, Mr, N, K)
    for v, _ in variants:
        t = sorted(times[v])[len(times[v]) // 2]
        out += " | %s %7.1f us %5.0f TF" % (v, t, 2.0 * Mr * N * K / t / 1e6)
    print(out, flush=True)


V = [("stagger 0", "0"), ("2", "2"), ("3", "3"), ("4", "4"), ("6", "6")]
for N, K, b_t, tag in ((4096, 1024, False, "fc1 fwd"), (4096, 1024, True, "fc2 dgrad"), (3072, 1024, False, "qkv fwd"), (1024, 4096, False, "fc2 fwd")):
    bench(tag + ": plain bf16 store", N, K, b_t, lambda: (bf(M, N), {}), V)
    if tag == "fc1 fwd":
        bench(tag + ": + bias, gelu, c2", N, K, b_t, lambda: (bf(M, N), dict(bias=torch.randn(N, device=dev), act=1, c2=bf(M, N))), V)
    if tag == "fc2 dgrad":
        bench(tag + ": * gelu'(R bf16)", N, K, b_t, lambda: (bf(M, N), dict(R=rb(M, N), rmode=2, ract=1)), V)
    if tag == "qkv fwd":
        bench(tag + ": + bias", N, K, b_t, lambda: (bf(M, N), dict(bias=torch.randn(N, device=dev))), V)

