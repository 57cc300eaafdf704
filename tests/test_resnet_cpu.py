"""ResNet back-end (scl_amd/resnet_head.py, torch-composed) against the reference's own model/wav2vec2_resnet_nll.py::Model via
tests/golden/resnet.npz (oracle/gen_golden.py::gen_resnet): outputs, the reference's loss terms (no 1/bz on this plugin),
gradients and BatchNorm buffers.  fp32 on both sides: 2e-4 relative to the tensor's max magnitude."""
import os

import numpy as np
import pytest
import torch

from oracle import head as OH
from oracle.aasist import fill_state
from scl_amd.resnet_head import DEFAULT_RESNET, ResNetHead

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "resnet.npz"))
TOL = 2e-4


def _close(a, b, name, tol=TOL):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, name
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-6)
    assert err < tol, "%s: rel err %.3e" % (name, err)


class _WithLL(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.LL = torch.nn.Linear(16, 128)
        head = ResNetHead(DEFAULT_RESNET)
        for n, c in head.named_children():
            self.add_module(n, c)

    def forward(self, x):
        feats = self.LL(x)
        out, emb = ResNetHead.forward(self, feats)
        return out, feats, emb


@pytest.mark.parametrize("case", ["eval", "train"])
def test_resnet_head_matches_reference(case):
    m = _WithLL()
    sd = m.state_dict()
    filled = fill_state({k: tuple(v.shape) for k, v in sd.items()}, seed=7)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})     # same key set as the reference or this raises
    m.eval() if case == "eval" else m.train()
    x = torch.from_numpy(G["x"]).clone().requires_grad_(True)
    y = torch.from_numpy(G["y"])
    out, feats, emb = m(x)
    bz = out.shape[0]
    losses = {k: v * bz for k, v in OH.model_loss(out, feats, emb, y, 1).items()}     # this plugin's Model.loss has no 1/bz
    sum(losses.values()).backward()
    _close(out.detach(), G[case + ":logits"], "logits")
    _close(emb.detach(), G[case + ":emb"], "emb")
    _close(feats.detach(), G[case + ":feats"], "feats")
    for k, v in losses.items():
        _close(v.detach(), G[case + ":loss:" + k], "loss " + k)
    _close(x.grad, G[case + ":grad_x"], "grad_x", 1e-3)
    params = dict(m.named_parameters())
    for k in G.files:
        if k.startswith(case + ":grad:"):
            _close(params[k.split(":", 2)[2]].grad, G[k], k, 1e-3)
        if k.startswith(case + ":gradfp:"):
            g = params[k.split(":", 2)[2]].grad
            fp = np.concatenate([[g.norm().item(), g.sum().item()], g.flatten()[:16].numpy()])
            _close(fp, G[k], k, 1e-3)
        if k.startswith(case + ":buf:"):
            _close(m.state_dict()[k.split(":", 2)[2]], G[k], k)
