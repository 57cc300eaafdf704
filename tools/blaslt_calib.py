"""Calibration only (not part of the product path): what does the vendor GEMM (torch.matmul -> hipBLASLt/rocBLAS) reach on
the encoder's shapes?  Gives the distance of scl_gemm_* from a tuned library kernel on the same silicon."""
import torch

dev = torch.device("cuda:0")
import sys
MB = int(sys.argv[1]) if len(sys.argv) > 1 else 12736          # rows of the encoder linears: 12736 = batch 64, 6368 = batch 32
shapes = [("fc1 fwd", MB, 4096, 1024, "nt"), ("fc2 fwd", MB, 1024, 4096, "nt"), ("qkv fwd", MB, 3072, 1024, "nt"),
          ("out fwd", MB, 1024, 1024, "nt"), ("fc1 dgrad", MB, 1024, 4096, "nn"), ("fc2 dgrad", MB, 4096, 1024, "nn"), ("fc1 wgrad", 4096, 1024, MB, "tn"),
          ("fc2 wgrad", 1024, 4096, MB, "tn"), ("out wgrad", 1024, 1024, MB, "tn"), ("conv1", 6399 * (MB // 199), 512, 1536, "nt"),
          ("big square", 8192, 8192, 8192, "nt")]
for name, M, N, K, lay in shapes:
    a = torch.randn((M, K) if lay[0] == "n" else (K, M), device=dev, dtype=torch.bfloat16)
    b = torch.randn((N, K) if lay[1] == "t" else (K, N), device=dev, dtype=torch.bfloat16)
    A = a if lay[0] == "n" else a.t()
    Bm = b.t() if lay[1] == "t" else b
    for _ in range(5):
        c = A @ Bm
    torch.cuda.synchronize()
    n = 30
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        c = A @ Bm
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print("%-10s M=%6d N=%5d K=%5d %s  %8.1f us  %7.1f TFLOP/s" % (name, M, N, K, lay, us, 2.0 * M * N * K / us / 1e6))
