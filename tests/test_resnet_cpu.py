"""Pins the ORACLE restatement of the ResNet back-end (oracle/resnet_head.py: one functional forward over the reference's state-dict
names) to the reference's own model/wav2vec2_resnet_nll.py::Model via tests/golden/resnet.npz (oracle/gen_golden.py::gen_resnet):
outputs, the reference's loss terms (no 1/bz on this plugin), gradients and BatchNorm buffers.  fp32 on both sides: 2e-4 relative to
the tensor's max magnitude.  The product back-end (scl_amd/resnet_head.py, HIP kernels) is checked against the same vectors on the
GPU in tests/test_resnet_gpu.py."""
import os

import numpy as np
import pytest
import torch

from oracle import head as OH
from oracle.aasist import fill_state
from oracle import resnet_head as ORH

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
        self.tree = ORH.ParamTree(resnet_shapes())
        for n, c in self.tree.named_children():
            self.add_module(n, c)
        del self._modules["tree"]

    def forward(self, x):
        feats = self.LL(x)
        t = dict(self.named_parameters()); t.update(dict(self.named_buffers()))
        out, emb = ORH.forward(t, feats, self.training)
        return out, feats, emb


def resnet_shapes():
    """The reference Model's state-dict key set for resnet_type 18 (written out once; load_state_dict below raises on any mismatch
    with oracle/aasist.py::fill_state's keys, which gen_golden.py took from the reference)."""
    sh = {"first_bn": 1, "first_bn1": 64}
    shapes = {}
    def bn(name, c):
        shapes.update({name + ".weight": (c,), name + ".bias": (c,), name + ".running_mean": (c,), name + ".running_var": (c,),
                       name + ".num_batches_tracked": ()})
    for n, c in sh.items():
        bn(n, c)
    shapes["resnet.conv1.weight"] = (16, 1, 9, 3); bn("resnet.bn1", 16)
    cin = 16
    for s, planes in zip((1, 2, 3, 4), (64, 128, 256, 512)):
        for j in range(2):
            p = "resnet.layer%d.%d." % (s, j)
            bn(p + "bn1", cin); shapes[p + "conv1.weight"] = (planes, cin, 3, 3)
            bn(p + "bn2", planes); shapes[p + "conv2.weight"] = (planes, planes, 3, 3)
            if (s > 1 and j == 0) or cin != planes:
                shapes[p + "shortcut.0.weight"] = (planes, cin, 1, 1)
            cin = planes
    shapes["resnet.conv5.weight"] = (256, 512, 3, 3); bn("resnet.bn5", 256)
    shapes["resnet.fc.weight"] = (2, 256); shapes["resnet.fc.bias"] = (2,)
    return shapes


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
