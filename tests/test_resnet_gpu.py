"""`wav2vec2_resnet_nll` plugin on the GPU: HIP encoder + LL + losses + HIP ResNet back-end (scl_amd/resnet_head.py) on flat-buffer
views.  (1) The back-end alone against the reference's own Model through tests/golden/resnet.npz — exact-fp32 convolutions at
2e-4, the default bf16-operand convolutions at the bf16 bar.  (2) The whole plugin against the CPU chain oracle wav2vec2 (fp32) ->
LL -> oracle/resnet_head.py (pinned to the same golden by tests/test_resnet_cpu.py)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import head as OH  # noqa: E402
from oracle import wav2vec2 as W  # noqa: E402
from oracle.aasist import fill_state  # noqa: E402
from scl_amd.encoder import W2VConfig  # noqa: E402
from scl_amd.model_resnet import Model  # noqa: E402
from scl_amd.optim import FusedAdamW  # noqa: E402
from oracle import resnet_head as ORH  # noqa: E402
from scl_amd.resnet_head import DEFAULT_RESNET, ResNetHead  # noqa: E402

pytestmark = pytest.mark.gpu
ARGS = {"flag_fix_ssl": False, "contra_mode": "all", "loss_type": 1, "resnet": DEFAULT_RESNET}
CONF = {"model": {"contra_mode": "all", "loss_type": 1}}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def rl2(got, ref):
    got, ref = torch.as_tensor(got).float().cpu(), torch.as_tensor(ref).float().cpu()
    return ((got - ref).norm() / ref.norm().clamp_min(1e-12)).item()


class CpuRef(torch.nn.Module):
    """oracle head (functional forward over the reference's state-dict names) + LL on the CPU."""

    def __init__(self, ssl_sd, cfg, head_sd):
        super().__init__()
        self.cfg = cfg
        self.ssl = {k: v.clone() for k, v in ssl_sd.items()}
        self.LL = torch.nn.Linear(cfg.embed, 128)
        tree = ORH.ParamTree({k: tuple(v.shape) for k, v in head_sd.items() if not k.startswith("LL.")})
        for n, c in tree.named_children():
            self.add_module(n, c)
        self.load_state_dict(head_sd)

    def head(self, feats):
        t = dict(self.named_parameters()); t.update(dict(self.named_buffers()))
        return ORH.forward(t, feats, self.training)


def make(dev, seed, args=ARGS):
    cfg = W.W2VConfig.tiny()
    ssl = W.init_state(cfg, seed=seed)
    m = Model(args, dev, w2v_cfg=W2VConfig.tiny())
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items() if not k.startswith("ssl_model.")}
    head_sd = {k: torch.from_numpy(v) for k, v in fill_state(shapes, seed=seed + 1).items()}
    sd = {"ssl_model.model." + k: v for k, v in ssl.items()}
    sd.update(head_sd)
    m.load_state_dict(sd)
    return m, CpuRef(ssl, cfg, head_sd), cfg


def test_state_dict_names(dev):
    m = Model(ARGS, dev, w2v_cfg=W2VConfig.tiny())
    keys = set(m.state_dict().keys())
    for k in ("ssl_model.model.post_extract_proj.weight", "LL.bias", "first_bn.running_var", "first_bn1.weight", "resnet.conv1.weight",
              "resnet.bn1.num_batches_tracked", "resnet.layer1.0.bn1.weight", "resnet.layer2.0.shortcut.0.weight", "resnet.layer4.1.conv2.weight",
              "resnet.conv5.weight", "resnet.bn5.bias", "resnet.fc.weight"):
        assert k in keys, k
    assert not any("downsample" in k for k in keys)          # never registered by the reference either (resnet.py:150-157)
    lo, hi = m.P.flat.data_ptr(), m.P.flat.data_ptr() + 4 * m.P.n_total
    assert all(lo <= p.data_ptr() < hi for p in m.parameters())


def test_eval_forward_and_train_step(dev):
    m, ref, cfg = make(dev, 91)
    x = 0.1 * torch.randn(4, 24000, generator=torch.Generator().manual_seed(2))
    y = torch.tensor([1, 1, 0, 0])
    # --- eval forward against the CPU chain
    m.eval(); ref.eval()
    with torch.no_grad():
        rf = ref.LL(W.forward(ref.ssl, cfg, x))
        ro, re = ref.head(rf)
        out, feats, emb = m(x.to(dev))
    assert out.shape == (4, 2) and emb.shape == (4, 256)
    assert rl2(feats, rf) < 1e-2 and rl2(emb, re) < 3e-2 and rl2(out, ro) < 3e-2, (rl2(feats, rf), rl2(emb, re), rl2(out, ro))
    # --- train step against the CPU head on the same features
    m.train(); ref.train()
    opt = FusedAdamW(m, lr=1e-4, weight_decay=1e-4)
    for use_graphs in (False,):
        out, feats, emb = m(x.to(dev))
        losses = m.loss(out, feats, emb, y.to(dev), CONF)
        opt.zero_grad()
        sum(losses.values()).backward()
        torch.cuda.synchronize()
        head2 = CpuRef(ref.ssl, cfg, {k: v.cpu() for k, v in m.state_dict().items() if not k.startswith("ssl_model.")})
        head2.train()
        f_leaf = feats.detach().cpu().requires_grad_(True)
        o2, e2 = head2.head(f_leaf)
        rl = {k: v * 4 for k, v in OH.model_loss(o2, f_leaf, e2, y, 1).items()}
        sum(rl.values()).backward()
        for k in rl:
            assert abs(float(losses[k].detach()) - float(rl[k].detach())) < 2e-3 * abs(float(rl[k].detach())) + 1e-5, (use_graphs, k)
        refp = dict(head2.named_parameters())
        for k in ("resnet.fc.weight", "resnet.conv5.weight", "resnet.layer2.0.shortcut.0.weight", "resnet.conv1.weight", "first_bn.weight"):
            assert rl2(m.P.g(k), refp[k].grad) < 2e-2, (use_graphs, k, rl2(m.P.g(k), refp[k].grad))   # exact-fp32 HIP convolutions vs the fp32 CPU oracle (summation order), through train-mode BatchNorm at batch 4
        for n in [n for n, _, tr in W.param_shapes(cfg) if tr]:
            ref.ssl[n].requires_grad_(True)
            ref.ssl[n].grad = None
        head2.LL(W.forward(ref.ssl, cfg, x)).backward(f_leaf.grad)
        for n in ("post_extract_proj.weight", "encoder.layers.1.fc1.weight", "feature_extractor.conv_layers.0.0.weight"):
            assert rl2(m.P.g("ssl_model.model." + n), ref.ssl[n].grad) < 8e-2, (use_graphs, n)
        assert rl2(m.P.g("LL.weight"), refp["LL.weight"].grad) < 3e-2
    before = m.P.flat[: m.P.n_train].clone()
    opt.step()
    torch.cuda.synchronize()
    delta = (m.P.flat[: m.P.n_train] - before).abs()
    assert delta[m._head_lo:].max() > 0 and delta[: m._head_lo].max() > 0 and delta.max() <= 2.2e-4


def test_frozen_encoder_gets_no_gradient(dev):
    m, _, _ = make(dev, 95, dict(ARGS, flag_fix_ssl=True))
    m.train()
    x = (0.1 * torch.randn(4, 24000, generator=torch.Generator().manual_seed(3))).to(dev)
    y = torch.tensor([1, 1, 0, 0], device=dev)
    m.P.grad.zero_()
    out, feats, emb = m(x)
    sum(m.loss(out, feats, emb, y, CONF).values()).backward()
    torch.cuda.synchronize()
    lo = m.P.off("LL.weight")
    assert m.P.grad[:lo].abs().max().item() == 0.0 and m.P.grad[lo:].abs().max().item() > 0.0


G = np.load(os.path.join(os.path.dirname(__file__), "golden", "resnet.npz"))


class _HeadWithLL(torch.nn.Module):
    """LL (test harness, torch) + the product back-end under the reference's state-dict names, on the GPU."""

    def __init__(self):
        super().__init__()
        self.LL = torch.nn.Linear(16, 128)
        for n, c in ResNetHead(DEFAULT_RESNET).named_children():
            self.add_module(n, c)

    def forward(self, x):
        feats = self.LL(x)
        out, emb = ResNetHead.forward(self, feats)
        return out, feats, emb


@pytest.mark.parametrize("conv,tol,gtol", [("f32", 2e-4, 2e-3), ("bf16", 2e-2, None)])
@pytest.mark.parametrize("case", ["eval", "train"])
def test_hip_backend_matches_the_reference_golden(dev, monkeypatch, case, conv, tol, gtol):
    """tests/golden/resnet.npz holds inputs, filled weights and the outputs / loss terms / gradients / BatchNorm buffers of the
    REFERENCE's wav2vec2_resnet_nll Model (generated by importing it, oracle/gen_golden.py::gen_resnet).  The HIP back-end — implicit
    GEMM convolutions, fused BatchNorm kernels, HIP pooling and Linear — must reproduce them: 2e-4 with the exact-fp32 convolution
    kernel (the default; gradients 2e-3: fp32 summation order through train-mode BatchNorm at batch 4 — the CPU oracle needs 1e-3),
    forward at the bf16 bar with the opt-in bf16-operand convolutions (their gradients are not asserted: 10-28 % off on this
    golden's early layers, which is why they are opt-in)."""
    monkeypatch.setenv("SCL_RESNET_CONV", conv)
    m = _HeadWithLL().to(dev)
    sd = m.state_dict()
    filled = fill_state({k: tuple(v.shape) for k, v in sd.items()}, seed=7)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
    m.eval() if case == "eval" else m.train()
    x = torch.from_numpy(G["x"]).to(dev).requires_grad_(True)
    y = torch.from_numpy(G["y"])
    out, feats, emb = m(x)
    bz = out.shape[0]
    losses = {k: v * bz for k, v in OH.model_loss(out.cpu(), feats.cpu(), emb.cpu(), y, 1).items()}     # this plugin's Model.loss has no 1/bz
    sum(losses.values()).backward()
    torch.cuda.synchronize()

    def close(a, b, name, t):
        a, b = np.asarray(torch.as_tensor(a).detach().cpu(), dtype=np.float64), np.asarray(b, dtype=np.float64)
        assert a.shape == b.shape, name
        err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-6)
        assert err < t, "%s: rel err %.3e" % (name, err)
    close(out, G[case + ":logits"], "logits", tol); close(emb, G[case + ":emb"], "emb", tol); close(feats, G[case + ":feats"], "feats", 2e-4)
    for k, v in losses.items():
        close(v, G[case + ":loss:" + k], "loss " + k, tol)
    if gtol is None:
        return
    close(x.grad, G[case + ":grad_x"], "grad_x", gtol)
    params = dict(m.named_parameters())
    for k in G.files:
        if k.startswith(case + ":grad:"):
            close(params[k.split(":", 2)[2]].grad, G[k], k, gtol)
        if k.startswith(case + ":gradfp:"):
            g = params[k.split(":", 2)[2]].grad.cpu()
            close(np.concatenate([[g.norm().item(), g.sum().item()], g.flatten()[:16].numpy()]), G[k], k, gtol)
        if k.startswith(case + ":buf:"):
            close(m.state_dict()[k.split(":", 2)[2]], G[k], k, tol)
