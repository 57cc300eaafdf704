"""`wav2vec2_aasist` plugin on the GPU: HIP encoder + LL behind one autograd boundary, the hand-written HIP AASIST back-end
(scl_amd/resstack.py + graph.py) on flat-buffer parameter views.  Reference = oracle wav2vec2 restatement (fp32, CPU) -> LL -> the same AasistHead class that
tests/test_aasist_cpu.py pins to the reference's own Model.  Tolerance: the bf16 bar of BASELINE.json (rel-L2 < 1e-2 on the
encoder-side tensors); the back-end's top-k graph pooling is discontinuous, so tensors after it get 5e-2."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import head as OH  # noqa: E402
from oracle import wav2vec2 as W  # noqa: E402
from oracle.aasist import fill_state  # noqa: E402
from oracle.aasist_head import AasistHead as OracleHead
from scl_amd.aasist_head import UPSTREAM_AASIST, AasistHead  # noqa: E402
from scl_amd.encoder import W2VConfig  # noqa: E402
from scl_amd.model_aasist import Model  # noqa: E402
from scl_amd.optim import FusedAdamW  # noqa: E402

pytestmark = pytest.mark.gpu
ARGS = {"contra_mode": "all", "loss_type": 1, "aasist": UPSTREAM_AASIST}
CONF = {"model": {"contra_mode": "all", "loss_type": 1}}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def rl2(got, ref):
    got, ref = torch.as_tensor(got).float().cpu(), torch.as_tensor(ref).float().cpu()
    return ((got - ref).norm() / ref.norm().clamp_min(1e-12)).item()


class CpuRef(torch.nn.Module):
    def __init__(self, ssl_sd, cfg, head_sd):
        super().__init__()
        self.cfg = cfg
        self.ssl = {k: v.clone() for k, v in ssl_sd.items()}
        self.LL = torch.nn.Linear(cfg.embed, 128)
        head = OracleHead(UPSTREAM_AASIST)          # plain-torch CPU restatement, pinned to the reference by tests/test_aasist_cpu.py
        for n, c in head.named_children():
            self.add_module(n, c)
        for n in ("pos_S", "master1", "master2"):
            self.register_parameter(n, getattr(head, n))
        self.load_state_dict(head_sd)

    def forward(self, x):
        feats = self.LL(W.forward(self.ssl, self.cfg, x))
        out, hid = OracleHead.forward(self, feats)
        return out, feats, hid


def make(dev, seed, drop0=True):
    cfg = W.W2VConfig.tiny()
    ssl = W.init_state(cfg, seed=seed)
    m = Model(ARGS, dev, w2v_cfg=W2VConfig.tiny())
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items() if not k.startswith("ssl_model.")}
    head_sd = {k: torch.from_numpy(v) for k, v in fill_state(shapes, seed=seed + 1).items()}
    sd = {"ssl_model.model." + k: v for k, v in ssl.items()}
    sd.update(head_sd)
    m.load_state_dict(sd)
    ref = CpuRef(ssl, cfg, head_sd)
    if drop0:
        for mod in list(m.modules()) + list(ref.modules()):
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
    return m, ref, cfg


def test_state_dict_names_and_flat_views(dev):
    m = Model(ARGS, dev, w2v_cfg=W2VConfig.tiny())
    keys = set(m.state_dict().keys())
    for k in ("ssl_model.model.post_extract_proj.weight", "LL.weight", "first_bn.running_mean", "first_bn1.weight",
              "encoder.0.0.conv1.weight", "encoder.1.0.bn1.running_var", "encoder.2.0.conv_downsample.bias", "attention.2.num_batches_tracked",
              "pos_S", "master2", "GAT_layer_S.att_weight", "HtrgGAT_layer_ST12.att_weightM", "HtrgGAT_layer_ST21.proj_type2.bias",
              "pool_hT2.proj.weight", "out_layer.bias"):
        assert k in keys, k
    lo, hi = m.P.flat.data_ptr(), m.P.flat.data_ptr() + 4 * m.P.n_total
    for n, p in m.named_parameters():
        assert lo <= p.data_ptr() < hi, n                      # every parameter (torch head included) is a flat-buffer view
        if p.requires_grad:
            assert p.grad is not None and m.P.grad.data_ptr() <= p.grad.data_ptr() < m.P.grad.data_ptr() + 4 * m.P.n_train, n


def test_eval_forward_matches_cpu_reference(dev):
    m, ref, cfg = make(dev, 41)
    m.eval(); ref.eval()
    x = 0.1 * torch.randn(3, 20000, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        ro, rf, rh = ref(x)
        out, feats, hid = m(x.to(dev))
    assert out.shape == (3, 2) and hid.shape == (3, 160) and feats.shape == rf.shape
    assert rl2(feats, rf) < 1e-2, rl2(feats, rf)
    assert rl2(hid, rh) < 5e-2 and rl2(out, ro) < 5e-2, (rl2(hid, rh), rl2(out, ro))
    m.is_train = False
    with torch.no_grad():
        lone = m(x.to(dev))
    assert torch.equal(lone, out)
    m.is_train = True


def test_train_step_gradients_and_update(dev):
    m, ref, cfg = make(dev, 51)
    m.train(); ref.train()
    x = 0.1 * torch.randn(6, 20000, generator=torch.Generator().manual_seed(3))
    y = torch.tensor([1, 1, 1, 0, 0, 0])
    train_names = [n for n, _, tr in W.param_shapes(cfg) if tr]
    for n in train_names:
        ref.ssl[n].requires_grad_(True)
    opt = FusedAdamW(m, lr=1e-4, weight_decay=1e-4, overlap=False)   # two backward passes are inspected before the step
    out, feats, hid = m(x.to(dev))
    losses = m.loss(out, feats, hid, y.to(dev), CONF)
    opt.zero_grad()
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    # (1) back-end + losses: the CPU copy of the head on the SAME features (the graph pooling's top-k is discontinuous, so the
    #     bf16 encoder error must not enter this comparison): fp32 torch on both sides, 2e-3
    f_leaf = feats.detach().cpu().requires_grad_(True)
    ro, rh = OracleHead.forward(ref, f_leaf)
    rl = OH.model_loss(ro, f_leaf, rh, y, 1)
    sum(rl.values()).backward()
    for k in rl:
        assert abs(float(losses[k].detach()) - float(rl[k].detach())) < 2e-3 * abs(float(rl[k].detach())) + 1e-5, (k, float(losses[k].detach()), float(rl[k].detach()))
    refp = dict(ref.named_parameters())
    for k in ("out_layer.weight", "HtrgGAT_layer_ST11.att_weight12", "GAT_layer_T.proj_with_att.weight", "encoder.0.0.conv1.weight",
              "attention.0.weight", "pos_S", "master1", "pool_hT2.proj.weight"):
        assert rl2(m.P.g(k), refp[k].grad) < 2e-3, (k, rl2(m.P.g(k), refp[k].grad))
    # (2) encoder + LL: push the reference d(feats) through the fp32 oracle encoder; bf16 bar on the HIP side
    rf = ref.LL(W.forward(ref.ssl, cfg, x))
    assert rl2(feats, rf) < 1e-2
    rf.backward(f_leaf.grad)
    for k in ("LL.weight", "LL.bias"):
        assert rl2(m.P.g(k), refp[k].grad) < 3e-2, (k, rl2(m.P.g(k), refp[k].grad))
    for n in ("post_extract_proj.weight", "encoder.layers.1.fc1.weight", "encoder.layers.0.self_attn.q_proj.weight", "feature_extractor.conv_layers.0.0.weight"):
        got, want = m.P.g("ssl_model.model." + n), ref.ssl[n].grad
        assert rl2(got, want) < 8e-2, (n, rl2(got, want))
    # a second backward after zero_grad gives the same head gradients (autograd accumulates into the flat views; zero_grad clears them)
    g1 = m.P.grad[m._head_lo:].clone()
    out, feats, hid = m(x.to(dev))
    losses = m.loss(out, feats, hid, y.to(dev), CONF)
    opt.zero_grad()
    sum(losses.values()).backward()
    assert torch.allclose(m.P.grad[m._head_lo:], g1, rtol=1e-3, atol=1e-6 + 1e-3 * g1.abs().max().item())
    before = m.P.flat[: m.P.n_train].clone()
    opt.step()
    torch.cuda.synchronize()
    delta = (m.P.flat[: m.P.n_train] - before).abs()
    assert delta[m._head_lo:].max() > 0 and delta[: m._head_lo].max() > 0 and delta.max() <= 2.2e-4
    # and the next forward sees the updated weights
    out2, _, _ = m(x.to(dev))
    assert not torch.equal(out2, out)


def test_dropout_active_in_train_mode(dev):
    m, _, _ = make(dev, 61, drop0=False)
    m.train()
    x = (0.1 * torch.randn(4, 20000, generator=torch.Generator().manual_seed(1))).to(dev)
    a = m(x)[2]
    b = m(x)[2]
    assert not torch.equal(a, b)


def test_backend_as_hip_graphs_equals_eager(dev):
    """In training the back-end is replayed as captured forward / backward hipGraphs: same outputs, same gradients, same
    BatchNorm running statistics as the eager launch sequence (dropout p = 0 on both sides; capture warm-up must leave no trace)."""
    res = []
    for use_graphs in (False, True):
        m, _, _ = make(dev, 81)
        m.use_graphs = use_graphs
        m.train()
        opt = FusedAdamW(m, lr=1e-4, weight_decay=1e-4, overlap=False)
        gen = torch.Generator().manual_seed(4)
        y = torch.tensor([1, 1, 1, 0, 0, 0], device=dev)
        outs = []
        for step in range(3):
            x = (0.1 * torch.randn(6, 20000, generator=gen)).to(dev)
            out, feats, hid = m(x)
            losses = m.loss(out, feats, hid, y, CONF)
            opt.zero_grad()
            sum(losses.values()).backward()
            outs.append((out.detach().clone(), hid.detach().clone(), m.P.grad.clone()))
            opt.step()
        torch.cuda.synchronize()
        assert bool(m._graphed) == use_graphs and all(v is not False for v in m._graphed.values())
        res.append((outs, m.state_dict()["first_bn1.running_mean"].clone(), m.state_dict()["encoder.1.0.bn1.num_batches_tracked"].item(),
                    m.P.flat[: m.P.n_train].clone()))
    (eo, ebn, ecnt, ep), (go, gbn, gcnt, gp) = res
    assert ecnt == gcnt == 3
    errs = [(rl2(a2, a), rl2(b2, b), rl2(c2, c)) for (a, b, c), (a2, b2, c2) in zip(eo, go)]
    print("graph-vs-eager rel-L2 per step (logits, hidden, grads):", errs)
    # step 0 runs the same function on the same state: round-off only.  Later steps see weights that differ by Adam's
    # +-lr sign noise on near-zero gradients, and the graph pooling's top-k selection is discontinuous in them.
    assert errs[0][0] < 2e-3 and errs[0][1] < 2e-3 and errs[0][2] < 5e-3, errs
    assert all(np.isfinite(e).all() for e in errs)
    assert rl2(gbn, ebn) < 5e-2
    assert (gp - ep).abs().max().item() <= 2.2e-4 * 3


def test_graph_replay_reads_the_live_weights(dev):
    """The captured back-end must not freeze anything derived from the weights (re-laid-out convolution filters are cached between
    optimizer steps in eager mode): after the weights move, replay == eager on the new weights."""
    m, _, _ = make(dev, 83)
    m.use_graphs = True
    m.train()
    x = (0.1 * torch.randn(6, 20000, generator=torch.Generator().manual_seed(5))).to(dev)
    out0 = m(x)[0].detach().clone()                      # captures
    assert m._graphed and all(v is not False for v in m._graphed.values())
    with torch.no_grad():
        m.P.flat[m._head_lo: m.P.n_train].mul_(1.25)     # every back-end weight, in place, as the fused optimizer does
    m.optimizer_stepped(False)
    out_g = m(x)[0].detach().clone()                     # replay
    m.use_graphs = False
    out_e = m(x)[0].detach().clone()
    torch.cuda.synchronize()
    assert rl2(out_g, out_e) < 2e-3, (rl2(out_g, out_e), rl2(out_g, out0))
    assert rl2(out_e, out0) > 1e-2                       # the weights did move the output


G = np.load(os.path.join(os.path.dirname(__file__), "golden", "aasist.npz"))


class _HeadWithLL(torch.nn.Module):
    """LL (test harness, torch) + the product back-end under the reference's state-dict names, on the GPU."""

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


@pytest.mark.parametrize("tag", ["", "199:", "202:"])
@pytest.mark.parametrize("case", ["eval", "train"])
def test_hip_backend_matches_the_reference_golden(dev, case, tag):
    """tests/golden/aasist.npz: inputs, filled weights and the outputs / gradients / BatchNorm buffers of the REFERENCE's own
    wav2vec2_aasist Model (oracle/gen_golden.py::gen_aasist imports it).  The HIP back-end (implicit-GEMM convolutions and
    projections on the exact-fp32 matrix-core kernel, fused BatchNorm+SELU, fused pairwise attention scores, HIP max pool) must
    reproduce them at fp32 round-off: 2e-4 of each tensor's max magnitude.  Sizes: the small map ("" : T = 61 -> 20 temporal nodes), the
    BASELINE map ("199:": 42 x 66, 66 nodes — 33 kept by the first pools, the fused kernels' 256-position tiles) and an odd width ("202:":
    67 nodes); at the two big sizes EVERY parameter gradient is checked and the fused kernels (resstack.hip, graph.hip) must be the path
    that ran.  Top-k ties: the fused pool orders equal scores by index, torch.topk by its own rule; the random inputs hold none.
    Bounds: outputs 2e-5 at every size (measured 2e-6).  Gradients 2e-4 on the small map; on the big maps 2e-3 (round 6; was 5e-3) of the tensor's largest
    magnitude: SELU's derivative jumps 1.758 -> 1.051 at zero, and among the 5 M pre-activations of a 4 x 42 x 66 stack one can land within
    fp32 rounding of zero on the other side of where the reference's own rounding put it — a sparse O(1) difference in ONE element's
    derivative that spreads into the tensors upstream of it (measured with tools/aasist_golden_probe.py: every tensor of `199:train` and
    `202:eval` within 2e-6 .. 6e-6; `199:eval` up to 2.6e-4 in the chain LL <- block 0; `202:train` 1.4e-3 in encoder.5.0.bn2.bias, 1e-4 at
    grad_x).  DESIGN.md section 3 "Round 4" has the float64 demonstration of the same effect."""
    from scl_amd import graph, resstack
    from test_aasist_cpu import check_grads
    m = _HeadWithLL().to(dev)
    sd = m.state_dict()
    filled = fill_state({k: tuple(v.shape) for k, v in sd.items()}, seed=5)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
    if case == "eval":
        m.eval()
    else:
        m.train()
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
    x = torch.from_numpy(G[tag + "x"]).to(dev).requires_grad_(True)
    seen = lambda mod: {(id(pl), getattr(pl, "gen", 0)) for pl in mod._PLANS}
    rs_before, gr_before = seen(resstack), seen(graph)
    logits, hidden = m(x)
    # the fused path leaves its trace: a plan of this map size was acquired (new, or an idle one with a new generation) by the stack node
    # and by the graph node
    T = x.shape[1]
    assert any(pl.key[:3] == (x.shape[0], 42, T // 3) for pl in resstack._PLANS if (id(pl), pl.gen) not in rs_before), "resstack.hip did not run"
    assert any(pl.key == (x.shape[0], 42, T // 3) for pl in graph._PLANS if (id(pl), pl.gen) not in gr_before), "graph.hip did not run"
    (logits * torch.from_numpy(G[tag + "w_logits"]).to(dev)).sum().add((hidden * torch.from_numpy(G[tag + "w_hidden"]).to(dev)).sum()).backward()
    torch.cuda.synchronize()

    def close(a, b, name, tol=2e-4):
        a, b = np.asarray(torch.as_tensor(a).detach().cpu(), dtype=np.float64), np.asarray(b, dtype=np.float64)
        assert a.shape == b.shape, name
        err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-6)
        assert err < tol, "%s: rel err %.3e" % (name, err)
    pre = tag + case
    gtol = 2e-4 if not tag else 2e-3      # round 6: 5e-3 -> 2e-3 (measured <= 1.4e-3: a regression of 40 % is now caught)
    close(logits, G[pre + ":logits"], "logits", 2e-5); close(hidden, G[pre + ":hidden"], "hidden", 2e-5); close(x.grad, G[pre + ":grad_x"], "grad_x", gtol)
    n = check_grads(dict(m.named_parameters()), pre, lambda a, b, name: close(a, b, name, gtol), gtol)
    assert n == (12 if not tag else sum(1 for p in m.parameters() if p.grad is not None)), n
    for k in G.files:
        if k.startswith(pre + ":buf:"):
            close(m.state_dict()[k.split(":")[-1]], G[k], k)


def test_full_size_aasist_step_at_batch_64(dev):
    """BASELINE configs[3] as one rank sees it: XLS-R-300M encoder + AASIST back-end, 64 x 64000-sample clips.  The CPU oracle cannot
    run this size in test time, so the checks are the size-independent ones: (1) in eval mode (BatchNorm on running statistics) every
    utterance is independent of its batch, so rows 0-3 of the batch-64 forward reproduce the batch-4 forward, whose pieces the tests
    above pin to the reference; (2) a train step at this size gives finite losses in the band of a seeded-random-init model and
    finite gradients on every trainable element; (3) every BatchNorm of the back-end saw exactly one batch (per-rank statistics:
    num_batches_tracked == 1, running statistics moved off their initial values) and the optimizer step changes encoder and head."""
    torch.manual_seed(7)
    m = Model(ARGS, dev, w2v_cfg=W2VConfig(), seed=3)
    g = torch.Generator().manual_seed(1234)
    x4 = 0.1 * torch.randn(4, 64000, generator=g)
    x64 = torch.cat([x4, 0.1 * torch.randn(60, 64000, generator=g)]).to(dev)
    m.eval()
    with torch.no_grad():
        o4, f4, h4 = [t.clone() for t in m(x4.to(dev))]
        o64, f64, h64 = m(x64)
    assert o64.shape == (64, 2) and h64.shape == (64, 160) and f64.shape == (64, 199, 128)
    assert rl2(f64[:4], f4.cpu()) < 2e-3, rl2(f64[:4], f4.cpu())
    # the graph pooling's top-k is discontinuous in its scores: compare the back-end outputs where the features agree to bf16 noise
    assert rl2(h64[:4], h4.cpu()) < 5e-2 and rl2(o64[:4], o4.cpu()) < 5e-2, (rl2(h64[:4], h4.cpu()), rl2(o64[:4], o4.cpu()))
    m.train()
    used = [mod for mod in m.modules() if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm)]      # every one of them is on the AASIST path
    opt = FusedAdamW(m, lr=1e-5, weight_decay=1e-4)
    y = torch.tensor(([1] * 29 + [0] * 64)[:64], device=dev)
    before = m.P.flat[: m.P.n_train].clone()
    out, feats, hid = m(x64)
    losses = m.loss(out, feats, hid, y, CONF)
    total = sum(losses.values())
    opt.zero_grad()
    total.backward()
    torch.cuda.synchronize()
    assert 0.0 < total.item() < 2.0 and all(torch.isfinite(v) for v in losses.values()), {k: v.item() for k, v in losses.items()}
    assert torch.isfinite(m.P.grad[: m.P.n_train]).all()
    assert m.P.grad[m._head_lo: m.P.n_train].abs().max() > 0 and m.P.grad[: m._head_lo].abs().max() > 0
    tracked = [int(b.num_batches_tracked) for b in used]
    assert tracked and all(t == 1 for t in tracked), tracked
    assert any((b.running_mean != 0).any() for b in used)
    opt.step()
    torch.cuda.synchronize()
    delta = (m.P.flat[: m.P.n_train] - before).abs()
    assert delta[m._head_lo:].max() > 0 and delta[: m._head_lo].max() > 0 and torch.isfinite(m.P.flat[: m.P.n_train]).all()
