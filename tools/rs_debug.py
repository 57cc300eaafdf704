"""Compare the fused stack's intermediate gradient maps (plan buffers) with float64 autograd, block by block."""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scl_amd import resstack
from scl_amd.aasist_head import Residual_block

B, H, W, filts, training = 3, 42, 66, [[1, 32], [32, 32], [32, 64], [64, 64]], True
g = torch.Generator().manual_seed(B * 1000 + W)
blocks = [Residual_block(f, first=(i == 0)) for i, f in enumerate(filts)]
refp = []
for blk in blocks:
    with torch.no_grad():
        for n, p in blk.named_parameters():
            std = (1.0 / np.sqrt(p[0].numel())) if p.dim() > 1 else 0.2
            p.copy_(torch.randn(p.shape, generator=g) * std + (1.0 if n.endswith("bn2.weight") or n.endswith("bn1.weight") else 0.0))
        for n, b in blk.named_buffers():
            if n.endswith("running_mean"): b.copy_(0.1 * torch.randn(b.shape, generator=g))
            elif n.endswith("running_var"): b.copy_(0.5 + torch.rand(b.shape, generator=g))
    d = {n: p.detach().double().clone().requires_grad_(True) for n, p in blk.named_parameters()}
    d.update({n: b.detach().double().clone() for n, b in blk.named_buffers() if b.dtype.is_floating_point})
    refp.append(d)
    blk.to("cuda:0"); blk.train(training)
x = torch.randn(B, H, W, 1, generator=g); wout = torch.randn(B, H, W, filts[-1][1], generator=g)
xg = x.cuda().requires_grad_(True)
out = resstack.res_stack(xg, blocks); (out * wout.cuda()).sum().backward(); torch.cuda.synchronize()
pl = resstack._PLANS[-1]
# reference with retained intermediate grads
xs, y1s, zs = [], [], []
xr = x.double().permute(0, 3, 1, 2).clone().requires_grad_(True)
cur = xr
for i, p in enumerate(refp):
    xs.append(cur); cur.retain_grad()
    y1 = F.conv2d(cur, p["conv1.weight"], p["conv1.bias"], padding=(1, 1)); y1.retain_grad(); y1s.append(y1)
    a = F.selu(F.batch_norm(y1, p["bn2.running_mean"], p["bn2.running_var"], p["bn2.weight"], p["bn2.bias"], training, 0.1, 1e-5)); a.retain_grad(); zs.append(a)
    o = F.conv2d(a, p["conv2.weight"], p["conv2.bias"], padding=(0, 1))
    idn = F.conv2d(cur, p["conv_downsample.weight"], p["conv_downsample.bias"], padding=(0, 1)) if "conv_downsample.weight" in p else cur
    cur = o + idn
(cur * wout.double().permute(0, 3, 1, 2)).sum().backward()
def unb(t, c, r_lo, rows):
    v = t[pl.slack * c: (pl.slack + pl.G) * c].view(B, H + 2, W + 2, c)[:, r_lo:r_lo + rows, 1:W + 1].permute(0, 3, 1, 2).double().cpu()
    return v
e = lambda a, b: float((a - b).abs().max() / b.abs().max())
for i in range(len(blocks)):
    c = pl.cps[i]
    gx = unb(pl.dx[i], c, 1, H)[:, :xs[i].shape[1]]
    print("block %d  d(input) err %.2e   forward a err %.2e  y1 err %.2e" % (i, e(gx, xs[i].grad), e(unb(pl.a[i], pl.cps[i + 1], 0, H + 1), zs[i].detach()), e(unb(pl.y1[i], pl.cps[i + 1], 0, H + 1), y1s[i].detach())))
    if i == len(blocks) - 1 or True:
        pass
# the last block's dz buffer is overwritten by earlier blocks; block 0's d_y1 is what pl.dz holds at the end
for i in range(len(blocks)):
    st = pl.stats[i].double().cpu().view(4, -1)
    y = y1s[i].detach()
    m = y.mean(dim=(0, 2, 3)); v = y.var(dim=(0, 2, 3), unbiased=False)
    print("block %d stats: mean err %.2e rstd err %.2e  min var %.3e" % (i, float((st[0] - m).abs().max() / m.abs().max()), float((st[1] - 1 / torch.sqrt(v + 1e-5)).abs().max() / (1 / torch.sqrt(v + 1e-5)).abs().max()), float(v.min())))
print("block 0 d(y1) err %.2e" % e(unb(pl.dz, pl.cps[1], 0, H + 1), y1s[0].grad))
for i in range(len(blocks)):
    d = (unb(pl.dx[i], pl.cps[i], 1, H)[:, :xs[i].shape[1]] - xs[i].grad)
    am = d.abs().amax(dim=(0, 1))
    r, cc = np.unravel_index(int(am.argmax()), am.shape)
    print("block %d worst position row %d col %d; per-row max err" % (i, r, cc), [float("%.1e" % v) for v in d.abs().amax(dim=(0, 1, 3))[:4]], "...", [float("%.1e" % v) for v in d.abs().amax(dim=(0, 1, 3))[-3:]],
          " mean err %.2e (const shift?) %.2e" % (float(d.abs().mean()), float(d.mean())))
# is the localized error a SELU-derivative flip (pre-activation within rounding of zero)?
i = 1
p = refp[i]
y = y1s[i].detach()
m = y.mean(dim=(0, 2, 3), keepdim=True); v = y.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
z = (y - m) / torch.sqrt(v + 1e-5) * p["bn2.weight"].detach().view(1, -1, 1, 1) + p["bn2.bias"].detach().view(1, -1, 1, 1)
k = int(z.abs().argmin()); idx = np.unravel_index(k, z.shape)
print("block 1: smallest |z| = %.3e at (b, c, row, col) = %s; fused a there = %.3e" % (float(z.flatten()[k]), idx, float(unb(pl.a[i], pl.cps[i + 1], 0, H + 1)[idx])))
