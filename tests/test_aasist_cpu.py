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


def _close(a, b, name):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, name
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-6)
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


@pytest.mark.parametrize("case", ["eval", "train"])
def test_aasist_head_matches_reference(case):
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
    x = torch.from_numpy(G["x"]).clone().requires_grad_(True)
    logits, hidden = m(x)
    (logits * torch.from_numpy(G["w_logits"])).sum().add((hidden * torch.from_numpy(G["w_hidden"])).sum()).backward()
    _close(logits.detach(), G[case + ":logits"], "logits")
    _close(hidden.detach(), G[case + ":hidden"], "hidden")
    _close(x.grad, G[case + ":grad_x"], "grad_x")
    params = dict(m.named_parameters())
    for k in G.files:
        if k.startswith(case + ":grad:"):
            _close(params[k.split(":", 2)[2]].grad, G[k], k)
        if k.startswith(case + ":buf:"):
            _close(m.state_dict()[k.split(":", 2)[2]], G[k], k)


def test_fill_state_is_order_independent():
    a = fill_state({"b.weight": (3, 2), "a.bias": (4,)}, seed=1)
    b = fill_state({"a.bias": (4,), "b.weight": (3, 2)}, seed=1)
    assert all(np.array_equal(a[k], b[k]) for k in a)
