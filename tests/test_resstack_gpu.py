"""csrc/resstack.hip — the fused Residual_block stack of the AASIST encoder (model/wav2vec2_aasist.py:377-433) — against a float64
torch restatement of the reference block built from F.conv2d / F.batch_norm / F.selu on the CPU: outputs, input gradient, every
parameter gradient and the BatchNorm buffers, in training and in eval mode, on the real map size (42 x 66) and on an odd small one.
Tolerance 1e-4 of each tensor's largest magnitude (fp32 matrix cores vs float64)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scl_amd import resstack  # noqa: E402
from scl_amd.aasist_head import Residual_block  # noqa: E402

pytestmark = pytest.mark.gpu


def ref_stack(x, blocks, training):
    """x [B, 1, H, W] float64; blocks: list of dicts of float64 tensors (+ running buffers, updated in place)."""
    for i, p in enumerate(blocks):
        if i > 0:       # bn1: result discarded, running statistics updated (training)
            F.batch_norm(x, p["bn1.running_mean"], p["bn1.running_var"], p["bn1.weight"], p["bn1.bias"], training, 0.1, 1e-5)
        out = F.conv2d(x, p["conv1.weight"], p["conv1.bias"], padding=(1, 1))
        out = F.selu(F.batch_norm(out, p["bn2.running_mean"], p["bn2.running_var"], p["bn2.weight"], p["bn2.bias"], training, 0.1, 1e-5))
        out = F.conv2d(out, p["conv2.weight"], p["conv2.bias"], padding=(0, 1))
        idn = F.conv2d(x, p["conv_downsample.weight"], p["conv_downsample.bias"], padding=(0, 1)) if "conv_downsample.weight" in p else x
        x = out + idn
    return x


def close(got, want, name, tol=1e-4):
    got = np.asarray(torch.as_tensor(got).detach().double().cpu())
    want = np.asarray(torch.as_tensor(want).detach().double().cpu())
    assert got.shape == want.shape, (name, got.shape, want.shape)
    err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-9)
    assert err < tol, "%s: rel err %.3e (max |want| %.3e)" % (name, err, np.abs(want).max())


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("B,H,W,filts", [(3, 42, 66, [[1, 32], [32, 32], [32, 64], [64, 64]]), (2, 5, 9, [[1, 32], [32, 64], [64, 64], [64, 64]]),
                                        (5, 42, 67, [[1, 32], [32, 32]])])
def test_fused_stack_matches_float64_reference(B, H, W, filts, training):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B * 1000 + W)
    blocks = [Residual_block(f, first=(i == 0)) for i, f in enumerate(filts)]
    refp = []
    for blk in blocks:
        with torch.no_grad():
            for n, p in blk.named_parameters():
                # fan-in scaled filters: activations stay O(1) through the stack in eval mode too (running statistics do not re-normalise)
                std = (1.0 / np.sqrt(p[0].numel())) if p.dim() > 1 else 0.2
                p.copy_(torch.randn(p.shape, generator=g) * std + (1.0 if n.endswith("bn2.weight") or n.endswith("bn1.weight") else 0.0))
            for n, b in blk.named_buffers():
                if n.endswith("running_mean"):
                    b.copy_(0.1 * torch.randn(b.shape, generator=g))
                elif n.endswith("running_var"):
                    b.copy_(0.5 + torch.rand(b.shape, generator=g))
        d = {n: p.detach().double().clone().requires_grad_(True) for n, p in blk.named_parameters()}
        d.update({n: b.detach().double().clone() for n, b in blk.named_buffers() if b.dtype.is_floating_point})
        refp.append(d)
        blk.to(dev)
        blk.train(training)
    assert resstack.supported(blocks)
    x = torch.randn(B, H, W, 1, generator=g)
    wout = torch.randn(B, H, W, filts[-1][1], generator=g)
    xg = x.to(dev).requires_grad_(True)
    out = resstack.res_stack(xg, blocks)
    (out * wout.to(dev)).sum().backward()
    torch.cuda.synchronize()
    xr = x.double().permute(0, 3, 1, 2).clone().requires_grad_(True)
    ro = ref_stack(xr, refp, training)
    (ro * wout.double().permute(0, 3, 1, 2)).sum().backward()
    close(out.permute(0, 3, 1, 2), ro, "output")
    close(xg.grad.permute(0, 3, 1, 2), xr.grad, "grad_x")
    for i, (blk, d) in enumerate(zip(blocks, refp)):
        for n, p in blk.named_parameters():
            if n.startswith("bn1."):
                assert p.grad is None or float(p.grad.abs().max()) == 0.0      # discarded branch: no gradient
                continue
            # conv1's bias sits in front of a BatchNorm: its true gradient is 0 in training and both sides compute round-off
            tol = 1e-4
            if n == "conv1.bias" and training:
                assert float(p.grad.abs().max()) < 1e-3 * float(blk.conv2.bias.grad.abs().max())
                continue
            close(p.grad, d[n].grad, "block %d %s" % (i, n), tol)
        for n, b in blk.named_buffers():
            if b.dtype.is_floating_point:
                close(b, d[n], "block %d %s" % (i, n), 1e-5)
            elif training:
                assert int(b) == 1, (i, n, int(b))


def test_two_forwards_then_backwards_keep_their_own_activations():
    """A plan (the stack's saved activations) is busy until its backward: a second forward in between must not overwrite them."""
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    blocks = [Residual_block(f, first=(i == 0)).to(dev) for i, f in enumerate([[1, 32], [32, 32]])]
    for b in blocks:
        b.train()
    xa = torch.randn(2, 6, 7, 1, device=dev, requires_grad=True)
    xb = torch.randn(2, 6, 7, 1, device=dev, requires_grad=True)
    oa = resstack.res_stack(xa, blocks)
    ob = resstack.res_stack(xb, blocks)
    oa.sum().backward()
    ga = xa.grad.clone()
    xa.grad = None
    for b in blocks:
        b.zero_grad()
    oa2 = resstack.res_stack(xa, blocks)
    oa2.sum().backward()
    ob.sum().backward()
    torch.cuda.synchronize()
    assert torch.allclose(oa, oa2, rtol=1e-5, atol=1e-6) and torch.allclose(ga, xa.grad, rtol=1e-4, atol=1e-6)
    assert torch.isfinite(xb.grad).all() and float(xb.grad.abs().max()) > 0
