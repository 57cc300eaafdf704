"""Is the bf16-operand implicit-GEMM convolution numerically sound layer by layer?  One 3x3 convolution, forward / dgrad / wgrad
against an fp64 reference, bf16 and fp32 operands, on N(0,1) data and on data with a large mean (post-ReLU-like)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scl_amd import hipnn
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for name, shift in (("zero-mean", 0.0), ("mean 3 (post-ReLU-like)", 3.0)):
    for (B, H, W, Ci, Co, s) in ((4, 22, 128, 64, 64, 1), (4, 22, 128, 64, 128, 2)):
        x = (torch.randn(B, H, W, Ci, generator=g) + shift)
        w = torch.randn(Co, Ci, 3, 3, generator=g) * 0.05
        dy_shape = None
        xr = x.double().permute(0, 3, 1, 2).requires_grad_(True); wr = w.double().requires_grad_(True)
        yr = torch.nn.functional.conv2d(xr, wr, None, s, 1)
        dy = torch.randn(yr.shape, generator=g, dtype=torch.float64)
        dy = dy - dy.mean(dim=(0, 2, 3), keepdim=True)      # as after a train-mode BatchNorm: per-channel sums vanish
        yr.backward(dy)
        for dt in (torch.float32, torch.bfloat16):
            xd = x.to(dev).requires_grad_(True); wd = w.to(dev).requires_grad_(True)
            y = hipnn.conv2d(xd, wd, None, (s, s), (1, 1), dt)
            y.backward(dy.permute(0, 2, 3, 1).float().to(dev).contiguous())
            e = lambda a, b: ((a.double().cpu() - b).abs().max() / b.abs().max()).item()
            l2 = lambda a, b: ((a.double().cpu() - b).norm() / b.norm()).item()
            print("%-26s Ci %3d Co %3d s %d %-8s fwd max %.1e | dx max %.1e l2 %.1e | dw max %.1e l2 %.1e" % (
                name, Ci, Co, s, str(dt).split(".")[-1], e(y.permute(0, 3, 1, 2), yr.detach()), e(xd.grad.permute(0, 3, 1, 2), xr.grad), l2(xd.grad.permute(0, 3, 1, 2), xr.grad),
                e(wd.grad, wr.grad), l2(wd.grad, wr.grad)))
