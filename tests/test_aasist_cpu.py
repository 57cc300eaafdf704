"""Pins the ORACLE restatement of the AASIST back-end (oracle/aasist_head.py, plain torch on the CPU) to the reference's own
model/wav2vec2_aasist.py::Model via tests/golden/aasist.npz (oracle/gen_golden.py::gen_aasist).  fp32 on both sides: tolerance 2e-4
relative to the tensor's max magnitude (different op grouping only).  The product back-end (scl_amd/aasist_head.py, HIP kernels) is
checked against the same vectors on the GPU in tests/test_aasist_gpu.py."""
import os

import numpy as np
import pytest
import torch

from oracle.aasist import fill_state
from oracle.aasist_head import UPSTREAM_AASIST, AasistHead

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "aasist.npz"))
TOL = 2e-4


def _close(a, b, name, floor=1e-6):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, name
    err = np.abs(a - b).max() / max(np.abs(b).max(), floor)
    assert err < TOL, "%s: rel err %.3e" % (name, err)


class _WithLL(torch.nn.Module):
    """LL + head under the reference's state-dict names (the encoder is injected as identity in the golden)."""

    def __init__(self):
        super().__init__()
        self.LL = torch.nn.Linear(16, 128)
        head = AasistHead(UPSTREAM_AASIST)
        for n, c in head.named_children():
            self.add_module(n, c)
        for n in ("pos_S", "master1", "master2"):
            self.register_parameter(n, getattr(head, n))

    def forward(self, x):
        return AasistHead.forward(self, self.LL(x))


def analytically_zero(name, training):
    """Parameters whose gradient is zero in exact arithmetic: a per-channel constant in front of a soft-max (the attention block's
    BatchNorm shift and last conv bias, wav2vec2_aasist.py:470-480,566-577) and, with batch statistics, a bias in front of a BatchNorm
    (conv1 of every Residual_block, the stack's last conv2 ahead of first_bn1, the GAT projections ahead of their bn)."""
    if name in ("attention.2.bias", "attention.3.bias"):
        return True
    if not training:
        return False
    return (name.startswith("encoder.") and name.endswith(".conv1.bias")) or name == "encoder.5.0.conv2.bias" or \
        name.endswith(".proj_with_att.bias") or name.endswith(".proj_without_att.bias")


def check_grads(params, pre, close, tol=TOL):
    """Whole tensors (`:grad:`) or (norm, sum, first 16 values) fingerprints (`:gradfp:`) as gen_aasist stored them; returns the count."""
    n = 0
    casemax = max(np.abs(G[k]).max() for k in G.files if k.startswith(pre + ":grad:"))
    for k in G.files:
        if k.startswith(pre + ":grad:"):
            name = k.split(":", 2 + pre.count(":"))[-1]
            if analytically_zero(name, pre.endswith("train")):
                # round-off around zero on both sides (1e-6 .. 4e-5 here against gradients of 1 .. 35): bounded, not compared
                assert np.abs(G[k]).max() < 1e-5 * casemax and float(params[name].grad.abs().max()) < 1e-5 * casemax, k
            else:
                close(params[name].grad, G[k], k)
            n += 1
        elif k.startswith(pre + ":gradfp:"):
            name = k.split(":", 2 + pre.count(":"))[-1]
            g = np.asarray(params[name].grad.detach().cpu(), dtype=np.float64)
            ref = G[k]
            assert abs(np.sqrt((g ** 2).sum()) - ref[0]) < tol * ref[0], k
            assert abs(g.sum() - ref[1]) < tol * max(ref[0], 1e-6) * 10, k
            scale = max(np.abs(g).max(), 1e-12)
            assert np.abs(g.flatten()[:16] - ref[2:]).max() < tol * scale, k
            n += 1
    return n


@pytest.mark.parametrize("tag", ["", "199:", "202:"])
@pytest.mark.parametrize("case", ["eval", "train"])
def test_aasist_head_matches_reference(case, tag):
    m = _WithLL()
    sd = m.state_dict()
    filled = fill_state({k: tuple(v.shape) for k, v in sd.items()}, seed=5)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})     # same key set as the reference or this raises
    if case == "eval":
        m.eval()
    else:
        m.train()
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
    x = torch.from_numpy(G[tag + "x"]).clone().requires_grad_(True)
    logits, hidden = m(x)
    (logits * torch.from_numpy(G[tag + "w_logits"])).sum().add((hidden * torch.from_numpy(G[tag + "w_hidden"])).sum()).backward()
    pre = tag + case
    _close(logits.detach(), G[pre + ":logits"], "logits")
    _close(hidden.detach(), G[pre + ":hidden"], "hidden")
    _close(x.grad, G[pre + ":grad_x"], "grad_x")
    n = check_grads(dict(m.named_parameters()), pre, _close)
    assert n == (12 if not tag else sum(1 for p in m.parameters() if p.grad is not None))
    for k in G.files:
        if k.startswith(pre + ":buf:"):
            _close(m.state_dict()[k.split(":")[-1]], G[k], k)


def test_fill_state_is_order_independent():
    a = fill_state({"b.weight": (3, 2), "a.bias": (4,)}, seed=1)
    b = fill_state({"a.bias": (4,), "b.weight": (3, 2)}, seed=1)
    assert all(np.array_equal(a[k], b[k]) for k in a)
