"""Exact-f32 MFMA GEMM vs its bf16-pair form (SCL_GEMM_F32X3): time and error against float64 on the shapes the f32 kernel serves."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scl_amd import ops
from scl_amd.ops import Op
dev = torch.device("cuda:0")
def run(M, N, K, a_t=False, b_t=False, x3=False, iters=20):
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn((K, M) if a_t else (M, K), generator=g).to(dev)
    B = torch.randn((K, N) if b_t else (N, K), generator=g).to(dev)
    C = torch.empty(M, N, device=dev)
    call = lambda: ops.gemm(Op(A, M if a_t else K), Op(B, N if b_t else K), C, M, N, K, a_t=a_t, b_t=b_t, x3=x3)
    for _ in range(3): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    ref = (A.double().t() if a_t else A.double()) @ (B.double() if b_t else B.double().t())
    err = float((C.double() - ref).abs().max() / ref.abs().max())
    return us, err
shapes = [("scoring qkv 64x201", 12864, 3072, 1024, False, False), ("scoring fc1", 12864, 4096, 1024, False, False), ("scoring fc2", 12864, 1024, 4096, False, False),
          ("resnet conv 64->64 3x3 (32 utt)", 32 * 65 * 128, 64, 576, False, False), ("resnet conv 128->128", 32 * 33 * 64, 128, 1152, False, False),
          ("resnet conv 256->256", 32 * 17 * 32, 256, 2304, False, False), ("resnet conv 512->512", 32 * 9 * 16, 512, 4608, False, False),
          ("resnet wgrad 128->128 (A^T B)", 128, 1152, 33 * 64, True, True), ("btse mlp 128x128", 12736, 128, 128, False, False)]
for name, M, N, K, at, bt in shapes:
    t0, e0 = run(M, N, K, at, bt, False)
    t1, e1 = run(M, N, K, at, bt, True)
    fl = 2.0 * M * N * K
    print("%-36s exact %8.1f us %6.1f TF err %.1e | x3 %8.1f us %6.1f TF err %.1e | %.2fx" % (name, t0, fl / t0 / 1e6, e0, t1, fl / t1 / 1e6, e1, t0 / t1))
