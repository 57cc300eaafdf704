"""csrc/graph.hip — the fused graph module of the AASIST back-end (GAT layers, graph pooling, the two heterogeneous branches, read-out:
model/wav2vec2_aasist.py:62-374, 545-604) — against the per-operation composition (`AasistHead.graph_unfused`), which
tests/test_aasist_gpu.py pins to the reference's own goldens: logits, last_hidden, the gradients of both node sets, every parameter
gradient and the BatchNorm buffers, in training (dropout p = 0) and in eval mode.  Both arms are fp32 on the same device; the graph
pooling's top-k is discontinuous in its scores, so the bound is 2e-4 of each tensor's largest magnitude, and the dropout test
checks the properties that do not depend on the mask values."""
import copy
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scl_amd import graph  # noqa: E402
from scl_amd.aasist_head import UPSTREAM_AASIST, AasistHead  # noqa: E402

pytestmark = pytest.mark.gpu


def _head(seed, dev, training, drop0=True):
    torch.manual_seed(seed)
    h = AasistHead(UPSTREAM_AASIST)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in h.named_parameters():
            if n.endswith("bn.weight"):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            elif p.dim() == 1 or n.startswith("master") or n == "pos_S":
                p.copy_(0.3 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g) / np.sqrt(p.shape[-1] if p.dim() > 1 and p.shape[-1] > 1 else p.shape[0]))
        for n, b in h.named_buffers():
            if n.endswith("running_mean"):
                b.copy_(0.1 * torch.randn(b.shape, generator=g))
            elif n.endswith("running_var"):
                b.copy_(0.5 + torch.rand(b.shape, generator=g))
    h.to(dev)
    h.train(training)
    if drop0:
        for m in h.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
    return h


def close(got, want, name, tol=2e-4):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    assert got.shape == want.shape, (name, got.shape, want.shape)
    err = float((got - want).abs().max() / want.abs().max().clamp_min(1e-9))
    assert err < tol, "%s: rel err %.3e (max |want| %.3e)" % (name, err, float(want.abs().max()))


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("B,nT", [(5, 66), (3, 67), (2, 33)])
def test_fused_graph_module_equals_the_per_operation_composition(training, B, nT):
    dev = torch.device("cuda:0")
    ha = _head(11, dev, training)
    hb = copy.deepcopy(ha)
    g = torch.Generator().manual_seed(B * 100 + nT)
    eS, eT = torch.randn(B, 42, 64, generator=g), torch.randn(B, nT, 64, generator=g)
    wl, wh = torch.randn(B, 2, generator=g).to(dev), torch.randn(B, 160, generator=g).to(dev)
    assert graph.supported(ha, 42, nT)
    outs = []
    for h, fused in ((ha, True), (hb, False)):
        xs, xt = eS.to(dev).requires_grad_(True), eT.to(dev).requires_grad_(True)
        logits, hidden = graph.graph_module(xs, xt, h) if fused else AasistHead.graph_unfused(h, xs, xt)
        ((logits * wl).sum() + (hidden * wh).sum()).backward()
        torch.cuda.synchronize()
        outs.append((logits, hidden, xs.grad, xt.grad))
    for name, a, b in zip(("logits", "hidden", "grad e_S", "grad e_T"), outs[0], outs[1]):
        close(a, b, name)
    pa, pb = dict(ha.named_parameters()), dict(hb.named_parameters())
    checked = 0
    for n, p in pb.items():
        if p.grad is None:
            continue
        assert pa[n].grad is not None, n
        if training and n.endswith(("proj_with_att.bias", "proj_without_att.bias")):
            # a bias in front of a BatchNorm on batch statistics: the true gradient is 0 and both arms compute round-off
            wmax = float(pb[n.replace(".bias", ".weight")].grad.abs().max())
            assert float(pa[n].grad.abs().max()) < 1e-4 * wmax and float(p.grad.abs().max()) < 1e-4 * wmax, n
            continue
        close(pa[n].grad, p.grad, n, tol=5e-4)
        checked += 1
    assert checked >= (90 if not training else 78), checked
    ba, bb = dict(ha.named_buffers()), dict(hb.named_buffers())
    for n, b in bb.items():
        if n.startswith(("GAT_", "HtrgGAT_")):
            if b.dtype.is_floating_point:
                close(ba[n], b, n, tol=1e-5)
            else:
                assert int(ba[n]) == int(b), n


def test_dropout_in_the_fused_graph_module():
    """Train mode with the reference's probabilities: different masks every call, gradients finite, the hidden vector carries the
    p = 0.5 zeros of self.drop; eval mode is deterministic."""
    dev = torch.device("cuda:0")
    h = _head(21, dev, True, drop0=False)
    g = torch.Generator().manual_seed(5)
    eS, eT = torch.randn(8, 42, 64, generator=g).to(dev), torch.randn(8, 66, 64, generator=g).to(dev)
    xs, xt = eS.clone().requires_grad_(True), eT.clone().requires_grad_(True)
    l1, h1 = graph.graph_module(xs, xt, h)
    (l1.sum() + h1.sum()).backward()
    l2, h2 = graph.graph_module(eS, eT, h)
    torch.cuda.synchronize()
    assert not torch.equal(h1, h2)
    z = float((h1 == 0).float().mean())
    assert 0.35 < z < 0.65, z
    assert torch.isfinite(xs.grad).all() and torch.isfinite(xt.grad).all() and float(xs.grad.abs().max()) > 0
    assert all(torch.isfinite(p.grad).all() for p in h.parameters() if p.grad is not None)
    h.eval()
    with torch.no_grad():
        a = graph.graph_module(eS, eT, h)[0].clone()
        b = graph.graph_module(eS, eT, h)[0].clone()
    assert torch.equal(a, b)
