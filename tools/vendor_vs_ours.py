"""Same-box comparison (calibration only): vendor GEMM (torch.matmul -> hipBLASLt, plain bf16 output) against scl_gemm_bf16 with a plain bf16
store on the encoder's shapes at M = 12736, interleaved rounds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scl_amd import ops
from scl_amd.ops import Op
dev = torch.device("cuda:0")
M = 12736
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
for name, N, K, b_t in (("fc1 fwd", 4096, 1024, False), ("qkv fwd", 3072, 1024, False), ("out fwd", 1024, 1024, False), ("fc2 fwd", 1024, 4096, False),
                        ("fc2 dgrad", 4096, 1024, True), ("fc1 dgrad", 1024, 4096, True), ("out dgrad", 1024, 1024, True)):
    sets = []
    for i in range(3):
        A = (0.1 * torch.randn(M, K, device=dev)).bfloat16()
        B = (0.1 * torch.randn(K, N, device=dev) if b_t else 0.1 * torch.randn(N, K, device=dev)).bfloat16()
        sets.append((A, B, torch.empty(M, N, dtype=torch.bfloat16, device=dev)))
    cnt = [0]
    def ours():
        A, B, C = sets[cnt[0] % 3]; cnt[0] += 1
        ops.gemm(Op(A, K), Op(B, N if b_t else K), C, M, N, K, b_t=b_t)
    def vendor():
        A, B, C = sets[cnt[0] % 3]; cnt[0] += 1
        torch.matmul(A, B if b_t else B.t(), out=C)
    res = {"ours": [], "vendor": []}
    if os.environ.get("SEPARATE"):
        for r in range(3): res["ours"].append(t(ours))
        for r in range(3): res["vendor"].append(t(vendor))
    else:
        for r in range(3):
            res["ours"].append(t(ours)); res["vendor"].append(t(vendor))
    o, v = sorted(res["ours"])[1], sorted(res["vendor"])[1]
    print("%-10s N=%4d K=%4d %s | ours %6.1f us %5.0f TF | vendor %6.1f us %5.0f TF | ours/vendor %.2f" % (name, N, K, "NN" if b_t else "NT", o, 2.0 * M * N * K / o / 1e6, v, 2.0 * M * N * K / v / 1e6, o / v), flush=True)
