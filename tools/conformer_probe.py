"""Times one ConformerBlock (scl_amd/conformer.py) forward + backward at a BTSE-sized shape (configs/conf-5-btse-trans64.yaml: hidden 256,
4 heads; batch 128 x 199 frames = one 64000-sample clip per row) with HIP events, and the same block of oracle/conformer.py on the host
cores (fp32, a bounded sample of the batch) beside it.  Usage: python tools/conformer_probe.py [B] [n] [dim] [heads] [dim_head]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scl_amd import conformer as C  # noqa: E402


def main():
    a = [int(v) for v in sys.argv[1:]]
    B, n, dim, heads, dh = (a + [128, 199, 256, 4, 64][len(a):])[:5]
    dev = "cuda:0"
    torch.manual_seed(0)
    blk = C.ConformerBlock(dim=dim, dim_head=dh, heads=heads).to(dev).train()
    x = torch.randn(B, n, dim, device=dev, requires_grad=True)
    w = torch.randn(B, n, dim, device=dev)

    def step():
        y = blk(x)
        y.backward(w)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 20
    e0.record()
    for _ in range(K):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / K
    fl = 3 * C.flops(B, n, dim, dh, heads)
    print("ConformerBlock B=%d n=%d dim=%d heads=%d dim_head=%d: fwd+bwd %.3f ms, %.1f TFLOP/s (fp32 MFMA roof 157), %.0f sequences/s"
          % (B, n, dim, heads, dh, ms, fl / ms / 1e9, B / ms * 1e3))
    with torch.no_grad():
        blk.eval()
        for _ in range(3):
            blk(x)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(K):
            blk(x)
        e1.record()
        torch.cuda.synchronize()
        print("  eval forward %.3f ms" % (e0.elapsed_time(e1) / K))
    if os.environ.get("SCL_PROBE_CPU", "1") == "1":
        from oracle import conformer as OC
        Bc = max(1, min(B, 8))
        t = {k: v.detach().cpu().clone() for k, v in blk.state_dict().items()}
        for k, v in t.items():
            if v.dtype.is_floating_point and "running" not in k:
                v.requires_grad_(True)
        xc = x.detach()[:Bc].cpu().requires_grad_(True)
        wc = w[:Bc].cpu()
        OC.forward(t, xc, heads, True).backward(wc)
        t0 = time.time()
        reps = 3
        for _ in range(reps):
            OC.forward(t, xc, heads, True).backward(wc)
        dt = (time.time() - t0) / reps
        print("  oracle on %d host threads, %d sequences: %.1f ms = %.0f sequences/s" % (torch.get_num_threads(), Bc, dt * 1e3, Bc / dt))


if __name__ == "__main__":
    main()
