"""Run one GEMM shape repeatedly (for rocprofv3 --pmc passes).  usage: gemm_one.py M N K [nt|nn|tt] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scl_amd import ops
from scl_amd.ops import Op
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mode = sys.argv[4] if len(sys.argv) > 4 else "nt"
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dev = torch.device("cuda:0")
bf = lambda *s: (torch.randn(*s, device=dev) * 0.1).to(torch.bfloat16)
if mode == "nt":
    A, B, C = bf(M, K), bf(N, K), torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    f = lambda: ops.gemm(Op(A, K), Op(B, K), C, M, N, K)
elif mode == "nn":
    A, B, C = bf(M, K), bf(K, N), torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    f = lambda: ops.gemm(Op(A, K), Op(B, N), C, M, N, K, b_t=True)
else:
    A, B, C = bf(K, M), bf(K, N), torch.empty(M, N, dtype=torch.float32, device=dev)
    f = lambda: ops.gemm(Op(A, M), Op(B, N), C, M, N, K, a_t=True, b_t=True)
for _ in range(reps):
    f()
torch.cuda.synchronize()
