"""scl_amd/conformer.py (csrc/conformer.hip + the fp32 GEMM / LayerNorm / BatchNorm kernels) against
  * the reference's own ConformerBlock through tests/golden/conformer.npz: train, eval + mask, causal, clamped relative positions —
    output, input gradient, every parameter gradient, BatchNorm buffers;
  * oracle/conformer.py in float64 at other sizes (n not a multiple of 4, mask in train mode, a 2-block Conformer);
  * the reference's construction: same seed -> same state dict (keys, order, values).
Everything is fp32 with exact-fp32 products on both sides: bounds are 1e-4 of each tensor's largest magnitude (fp32 round-off of sums in
a different order), 5e-4 for parameter gradients (sums over B n rows)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import conformer as OC  # noqa: E402
from scl_amd import conformer as C  # noqa: E402
from tests import conformer_cases as CC  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(got, want):
    got, want = got.detach().double().cpu(), torch.as_tensor(want).double()
    return float((got.reshape(want.shape) - want).abs().max() / want.abs().max().clamp_min(1e-12))


@pytest.mark.parametrize("case", CC.CASES)
def test_block_matches_the_reference_goldens(case):
    g = CC.load()
    cfg = CC.cfg_of(g, case)
    blk = C.ConformerBlock(**cfg)
    blk.load_state_dict(CC.state_of(g, case))
    blk.to(DEV).train(CC.TRAIN[case])
    x = torch.from_numpy(g[case + ":x"]).to(DEV).requires_grad_(True)
    mask = torch.from_numpy(g[case + ":mask"]).to(DEV) if case + ":mask" in g.files else None
    y = blk(x, mask=mask)
    (y * torch.from_numpy(g[case + ":wout"]).to(DEV)).sum().backward()
    torch.cuda.synchronize()
    assert rel(y, g[case + ":y"]) < 1e-4
    assert rel(x.grad, g[case + ":grad_x"]) < 2e-4
    n = CC.check_grads(g, case, {k: p.grad for k, p in blk.named_parameters()}, 5e-4, "hip")
    assert n == len(list(blk.parameters()))
    for key in g.files:
        if key.startswith(case + ":buf:"):
            np.testing.assert_allclose(blk.state_dict()[key[len(case) + 5:]].cpu().numpy(), g[key], rtol=2e-5, atol=1e-6)


def _against_oracle(blk, x, mask, heads, training, tol=2e-4):
    t = {k: (v.detach().double() if v.dtype.is_floating_point else v.detach()).cpu().clone() for k, v in blk.state_dict().items()}
    for k, v in t.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    xo = x.detach().double().cpu().requires_grad_(True)
    w = torch.randn(x.shape, generator=torch.Generator().manual_seed(3)).double()
    yo = OC.forward(t, xo, heads, training, mask=None if mask is None else mask.cpu())
    (yo * w).sum().backward()
    xg = x.detach().clone().requires_grad_(True)
    y = blk(xg, mask=mask)
    (y * w.float().to(x.device)).sum().backward()
    torch.cuda.synchronize()
    assert rel(y, yo.detach()) < tol
    assert rel(xg.grad, xo.grad) < 2 * tol
    for k, p in blk.named_parameters():
        if training and k == "conv.net.4.conv.bias" and "conv.net.5.weight" in t:
            continue
        assert rel(p.grad, t[k].grad) < 5 * tol, k
    return t


@pytest.mark.parametrize("B,n,training,masked", [(2, 199, True, False), (5, 30, True, True), (1, 3, False, False), (4, 130, False, True)])
def test_block_against_the_float64_oracle(B, n, training, masked):
    torch.manual_seed(B * 1000 + n)
    blk = C.ConformerBlock(dim=128, dim_head=32, heads=4, conv_kernel_size=15).to(DEV).train(training)
    with torch.no_grad():
        blk.conv.net._modules["5"].running_mean.normal_(0, 0.1)
        blk.conv.net._modules["5"].running_var.uniform_(0.5, 1.5)
    x = torch.randn(B, n, 128, device=DEV)
    mask = None
    if masked:
        mask = torch.rand(B, n, device=DEV) > 0.25
        mask[0, 0] = True
    t = _against_oracle(blk, x, mask, 4, training)
    if training:      # running statistics moved like the oracle's
        assert rel(blk.state_dict()["conv.net.5.running_var"], t["conv.net.5.running_var"].detach()) < 1e-5
        assert int(blk.state_dict()["conv.net.5.num_batches_tracked"]) == int(t["conv.net.5.num_batches_tracked"])


def test_same_seed_gives_the_reference_state_dict():
    g = CC.load()
    torch.manual_seed(1234)
    blk = C.ConformerBlock(dim=64, dim_head=16, heads=4)
    sd = blk.state_dict()
    assert list(sd.keys()) == list(g["init:keys"])
    np.testing.assert_allclose([float(v.double().sum()) for v in sd.values()], g["init:fp"], rtol=0, atol=0)


def test_conformer_stack_and_no_grad_forward():
    torch.manual_seed(5)
    net = C.Conformer(64, depth=2, dim_head=16, heads=4, conv_kernel_size=7).to(DEV).eval()
    x = torch.randn(3, 41, 64, device=DEV)
    with torch.no_grad():
        y = net(x)
    xo = x.double().cpu()
    for blk in net.layers:
        t = {k: (v.detach().double() if v.dtype.is_floating_point else v.detach()).cpu() for k, v in blk.state_dict().items()}
        xo = OC.forward(t, xo, 4, False)
    assert rel(y, xo) < 2e-4
    with torch.no_grad():
        assert torch.equal(net(x), y)


def test_dropout_sites():
    """p > 0 in train mode: a different mask every call, finite gradients everywhere, the expected scale; eval mode ignores p."""
    torch.manual_seed(9)
    blk = C.ConformerBlock(dim=64, dim_head=16, heads=4, attn_dropout=0.1, ff_dropout=0.1, conv_dropout=0.1).to(DEV).train()
    x = torch.randn(4, 50, 64, device=DEV, requires_grad=True)
    y1 = blk(x)
    y1.sum().backward()
    y2 = blk(x.detach())
    torch.cuda.synchronize()
    assert not torch.equal(y1, y2)
    assert torch.isfinite(x.grad).all() and all(torch.isfinite(p.grad).all() for p in blk.parameters())
    blk.eval()
    ref = C.ConformerBlock(dim=64, dim_head=16, heads=4).to(DEV).eval()
    ref.load_state_dict(blk.state_dict())
    with torch.no_grad():
        assert torch.equal(blk(x), ref(x))


def test_rejects_what_the_kernels_cannot_address():
    with pytest.raises(ValueError):
        C.ConformerBlock(dim=30)
    with pytest.raises(ValueError):
        C.ConformerBlock(dim=64, conv_kernel_size=33)
    blk = C.ConformerBlock(dim=64, dim_head=16, heads=4)
    with pytest.raises(RuntimeError):
        blk(torch.zeros(1, 4, 64))
