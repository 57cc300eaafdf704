"""Calibration only (not part of the product path): what does the vendor GEMM (torch.matmul -> hipBLASLt/rocBLAS) reach on
the encoder's shapes?  Gives the distance of scl_gemm_* from a tuned library kernel on the same silicon."""
import torch

dev = torch.device("cuda:0")
shapes = [("fc1 fwd", 6368, 4096, 1024, "nt"), ("fc2 fwd", 6368, 1024, 4096, "nt"), ("qkv fwd", 6368, 3072, 1024, "nt"),
          ("out fwd", 6368, 1024, 1024, "nt"), ("fc1 dgrad", 6368, 1024, 4096, "nn"), ("fc1 wgrad", 4096, 1024, 6368, "tn"),
          ("fc2 wgrad", 1024, 4096, 6368, "tn"), ("conv1", 204768, 512, 1536, "nt"), ("conv2", 102368, 512, 1536, "nt"),
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
