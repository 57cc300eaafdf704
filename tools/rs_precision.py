"""Is the 1e-3 grad_x deviation of the fused stack against float64 conditioning or a bug?  Same case through torch fp32 (CPU)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_resstack_gpu as T
from scl_amd import resstack
from scl_amd.aasist_head import Residual_block

def case(B, H, W, filts, training, dtype_ref):
    g = torch.Generator().manual_seed(B * 1000 + W)
    blocks = [Residual_block(f, first=(i == 0)) for i, f in enumerate(filts)]
    refs = {torch.float64: [], dtype_ref: []}
    for blk in blocks:
        with torch.no_grad():
            for n, p in blk.named_parameters():
                std = (1.0 / np.sqrt(p[0].numel())) if p.dim() > 1 else 0.2
                p.copy_(torch.randn(p.shape, generator=g) * std + (1.0 if n.endswith("bn2.weight") or n.endswith("bn1.weight") else 0.0))
            for n, b in blk.named_buffers():
                if n.endswith("running_mean"): b.copy_(0.1 * torch.randn(b.shape, generator=g))
                elif n.endswith("running_var"): b.copy_(0.5 + torch.rand(b.shape, generator=g))
        for dt in refs:
            d = {n: p.detach().to(dt).clone().requires_grad_(True) for n, p in blk.named_parameters()}
            d.update({n: b.detach().to(dt).clone() for n, b in blk.named_buffers() if b.dtype.is_floating_point})
            refs[dt].append(d)
        blk.to("cuda:0"); blk.train(training)
    x = torch.randn(B, H, W, 1, generator=g); wout = torch.randn(B, H, W, filts[-1][1], generator=g)
    xg = x.cuda().requires_grad_(True)
    out = resstack.res_stack(xg, blocks); (out * wout.cuda()).sum().backward(); torch.cuda.synchronize()
    res = {}
    for dt, rp in refs.items():
        xr = x.to(dt).permute(0, 3, 1, 2).clone().requires_grad_(True)
        ro = T.ref_stack(xr, rp, training); (ro * wout.to(dt).permute(0, 3, 1, 2)).sum().backward()
        res[dt] = (ro.detach().double(), xr.grad.double(), rp)
    r64, r32 = res[torch.float64], res[dtype_ref]
    e = lambda a, b: float((a - b).abs().max() / b.abs().max())
    print("training", training, "out: hip-vs-f64 %.2e  torchf32-vs-f64 %.2e | grad_x: hip %.2e torchf32 %.2e" % (
        e(out.permute(0, 3, 1, 2).double().cpu(), r64[0]), e(r32[0], r64[0]), e(xg.grad.permute(0, 3, 1, 2).double().cpu(), r64[1]), e(r32[1], r64[1])))
    for i, blk in enumerate(blocks):
        for n, p in blk.named_parameters():
            if p.grad is None: continue
            a, b, c = p.grad.double().cpu(), r64[2][i][n].grad, r32[2][i][n].grad.double()
            print("   block %d %-24s hip %.2e  torchf32 %.2e   (max %.2e)" % (i, n, e(a, b), e(c, b), float(b.abs().max())))

case(3, 42, 66, [[1, 32], [32, 32], [32, 64], [64, 64]], True, torch.float32)
