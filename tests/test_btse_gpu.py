"""`wav2vec2_btse` (BASELINE.json configs[4]) on the GPU, through the C ABI.

 * the HIP back-end (scl_amd/btse_head.py: exact-fp32 MLP GEMMs + the fused bio transformer of csrc/btse.hip + the join / fc2 tail) against the
   vectors the REFERENCE's own model/wav2vec2_btse/model.py::Model produced (tests/golden/btse.npz, oracle/gen_golden.py::gen_btse): log-probs, b,
   every parameter gradient, the gradient at the encoder output — fp32 on both sides, 2e-4 of each tensor's largest magnitude;
 * the masking inside the bio transformer (padded tails never reach a score, model.py:236) against the oracle's encoder states;
 * train-mode dropout with host-rebuilt masks; longer token sequences (up to the kernel's 512) against the float64 oracle;
 * the whole plugin (HIP XLS-R encoder + LL + back-end + losses + fused AdamW) against the CPU chain at the bf16 bar, its state-dict keys
   against the reference's key list, and a batch-128 x 64000 step at XLS-R-300M shape (configs[4] as one rank sees it).
"""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import btse as OB  # noqa: E402
from oracle import head as OH  # noqa: E402
from oracle import wav2vec2 as W  # noqa: E402
from oracle.aasist import fill_state  # noqa: E402
from scl_amd import lib, ops  # noqa: E402
from scl_amd.btse_head import BtseHead  # noqa: E402
from scl_amd.encoder import W2VConfig  # noqa: E402
from scl_amd.model_btse import Model  # noqa: E402
from scl_amd.optim import FusedAdamW  # noqa: E402
from test_btse_cpu import CASES, G, case_args, check_grads  # noqa: E402
from test_dropout_gpu import keep_scale  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 2e-4
CONF = {"model": {"contra_mode": "all", "loss_type": 1}}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def rl2(got, ref):
    got, ref = torch.as_tensor(got).double().cpu(), torch.as_tensor(ref).double().cpu()
    return ((got - ref).norm() / ref.norm().clamp_min(1e-30)).item()


def close(a, b, name, tol=TOL):
    a, b = np.asarray(torch.as_tensor(a).detach().cpu(), dtype=np.float64), np.asarray(torch.as_tensor(b).detach().cpu(), dtype=np.float64)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-6)
    assert err < tol, "%s: rel err %.3e" % (name, err)


def golden_head(case, dev):
    """BtseHead on the GPU holding the golden's parameters (+ the golden's LL as plain tensors)."""
    args = case_args(case)
    shapes = OB.state_shapes(args, 16)
    filled = {k: torch.from_numpy(v) for k, v in fill_state(shapes, seed=int(G[case + ":seed"])).items()}
    head = BtseHead(args).to(dev)
    missing, unexpected = head.load_state_dict({k: v for k, v in filled.items() if not k.startswith("backend.LL.")}, strict=True)
    assert not missing and not unexpected
    return args, head, filled


@pytest.mark.parametrize("case", CASES)
def test_hip_backend_matches_the_reference_golden(case, dev):
    args, head, filled = golden_head(case, dev)
    head.eval()      # the golden ran model.train() with every Dropout p = 0
    feats = torch.from_numpy(G[case + ":feats"]).to(dev).requires_grad_(True)          # = LL(x): the plugin's `ssl_feat` output
    bio, lens = torch.from_numpy(G[case + ":bio"]), torch.from_numpy(G[case + ":lens"])
    logp, b = BtseHead.forward(head, feats, bio, lens)
    close(logp, G[case + ":logp"], "logp")
    close(b, G[case + ":b"], "b")
    w = lambda n: torch.from_numpy(G[case + ":" + n]).to(dev)
    ((logp * w("w_logp")).sum() + (b * w("w_b")).sum()).backward()
    torch.cuda.synchronize()
    grads = {k: v.grad.cpu().numpy() for k, v in head.named_parameters() if v.grad is not None}
    assert not any("m_utt_level" in k for k in grads)
    # LL sits in front of the back-end: its gradients and the gradient at the encoder output follow from d(feats) = head part + w_feats
    dfe = (feats.grad + w("w_feats")).double().cpu().reshape(-1, 128)
    x = torch.from_numpy(G[case + ":x"]).double().reshape(-1, 16)
    grads["backend.LL.weight"] = (dfe.t() @ x).numpy()
    grads["backend.LL.bias"] = dfe.sum(0).numpy()
    close((dfe @ filled["backend.LL.weight"].double()).reshape(G[case + ":grad_x"].shape), G[case + ":grad_x"], "grad_x")
    assert check_grads(case, grads, TOL) == len(filled) - 2


def test_padding_is_masked_inside_the_bio_transformer(dev):
    """Utterances shorter than the batch's longest: the encoder output (transformer.py:51) of their valid rows equals the oracle's, their
    padded rows are zero, their score is exactly zero — read from the kernel's scratch row through the C-ABI descriptor."""
    args, head, filled = golden_head("cat", dev)
    head.eval()
    feats = torch.from_numpy(G["cat:feats"]).to(dev)
    bio, lens = torch.from_numpy(G["cat:bio"]), torch.from_numpy(G["cat:lens"])
    with torch.no_grad():
        logp, b = BtseHead.forward(head, feats, bio, lens)
    torch.cuda.synchronize()
    sd = {k: v.double() for k, v in filled.items()}
    s_ref, x_ref = OB.bio_encoder(sd, args, bio, lens, return_x=True)
    pl = next(iter(head.__dict__["_btse_plans"].values()))
    B, Lt, NL = bio.shape[0], bio.shape[1], args["n_layers"]
    per = ops.btse_bio_ws_floats(NL, Lt)
    ws = pl["ws"].view(B, per).cpu()
    x_got = ws[:, NL * 392 * Lt:NL * 392 * Lt + 32 * Lt].view(B, Lt, 32)
    close(x_got, x_ref, "encoder output", 1e-5)
    for i, n in enumerate(lens.tolist()):
        assert (x_got[i, n:] == 0).all()
        if n < Lt:
            assert (b[i, 128:] == 0).all()
    close(b[:, 128:], s_ref, "bio score", 1e-5)


def test_train_mode_dropout_with_host_rebuilt_masks(dev):
    """MLP dropout (linear.py:30,36: p = 0.5 after every frame-level layer) = counter-hash masks drawn in the GEMM epilogue; the same
    masks rebuilt on the host (numpy port of csrc/common.h::hash_u32) and handed to the oracle give the same outputs and gradients."""
    args, head, filled = golden_head("cat", dev)
    head.train()
    BtseHead.reseed(head, 12345)
    feats = torch.from_numpy(G["cat:feats"]).to(dev).requires_grad_(True)
    bio, lens = torch.from_numpy(G["cat:bio"]), torch.from_numpy(G["cat:lens"])
    logp, b = BtseHead.forward(head, feats, bio, lens)
    w = lambda n: torch.from_numpy(G["cat:" + n])
    ((logp * w("w_logp").to(dev)).sum() + (b * w("w_b").to(dev)).sum()).backward()
    s0 = (12345 * 1664525 + 1013904223) & 0x7FFFFFFF
    B, T = feats.shape[:2]
    masks = [keep_scale((s0 + 7919 * j) & 0x7FFFFFFF, B * T * 128, 0.5).view(B, T, 128) for j in range(3)]
    assert all(abs((m == 0).float().mean().item() - 0.5) < 0.05 for m in masks)
    sd = {k: v.clone().requires_grad_(True) for k, v in filled.items()}
    fr = feats.detach().cpu().requires_grad_(True)
    h = fr
    for i in range(3):
        h = torch.nn.functional.leaky_relu(torch.nn.functional.linear(h, sd["backend.mlp.m_frame_level.linear_%d.weight" % i],
                                                                       sd["backend.mlp.m_frame_level.linear_%d.bias" % i]), 0.01) * masks[i]
    emb = h.mean(1)
    s = OB.bio_encoder(sd, args, bio, lens)
    rb = torch.cat((emb, s), 1)
    rlogp = torch.log_softmax(torch.nn.functional.linear(rb, sd["fc2.weight"], sd["fc2.bias"]), 1)
    ((rlogp * w("w_logp")).sum() + (rb * w("w_b")).sum()).backward()
    close(logp, rlogp, "logp")
    close(b, rb, "b")
    close(feats.grad, fr.grad, "d feats")
    hp = dict(head.named_parameters())
    for k in ("backend.mlp.m_frame_level.linear_0.weight", "backend.mlp.m_frame_level.linear_2.bias", "bioScoring.encoder.attn_layers.1.emb_rel_v",
              "bioScoring.bio_embedding.weight", "fc2.weight"):
        close(hp[k].grad, sd[k].grad, k)
    # a second call draws fresh masks; eval mode draws none and is repeatable
    with torch.no_grad():
        l2, _ = BtseHead.forward(head, feats.detach(), bio, lens)
        head.eval()
        e1, _ = BtseHead.forward(head, feats.detach(), bio, lens)
        e2, _ = BtseHead.forward(head, feats.detach(), bio, lens)
    assert not torch.equal(l2, logp.detach()) and torch.equal(e1, e2)


@pytest.mark.parametrize("Lt,B,is_add", [(64, 3, False), (65, 2, False), (199, 4, False), (200, 2, True), (512, 2, False)])
def test_long_token_sequences_against_the_float64_oracle(Lt, B, is_add, dev):
    """Token counts across the kernel's key-tile boundaries (64 per lane slot, 8 slots = 512), all utterances full length so that every one
    contributes: outputs 1e-4 / gradients 5e-4 of the float64 oracle (fp32 accumulation over up to 512 keys)."""
    args = OB.default_args(is_add=is_add, bio_out=128 if is_add else 64)
    shapes = OB.state_shapes(args, 16)
    filled = {k: torch.from_numpy(v) for k, v in fill_state(shapes, seed=900 + Lt).items()}
    head = BtseHead(args).to(dev)
    head.load_state_dict({k: v for k, v in filled.items() if not k.startswith("backend.LL.")})
    head.eval()
    rs = np.random.RandomState(Lt)
    T = 9
    f0 = torch.from_numpy(rs.standard_normal((B, T, 128)).astype(np.float32))
    bio = torch.from_numpy(rs.randint(0, 3, size=(B, Lt)).astype(np.int32))
    lens = torch.full((B,), Lt, dtype=torch.int32)
    wl, wb = torch.from_numpy(rs.standard_normal((B, 2)).astype(np.float32)), None
    feats = f0.to(dev).requires_grad_(True)
    logp, b = BtseHead.forward(head, feats, bio, lens)
    wb = torch.from_numpy(rs.standard_normal(tuple(b.shape)).astype(np.float32))
    ((logp * wl.to(dev)).sum() + (b * wb.to(dev)).sum()).backward()
    sd = {k: v.double().requires_grad_(True) for k, v in filled.items()}
    # the oracle's forward applies LL first: feed it through an identity LL
    sd["backend.LL.weight"] = torch.eye(128, dtype=torch.float64)
    sd["backend.LL.bias"] = torch.zeros(128, dtype=torch.float64)
    fr = f0.double().requires_grad_(True)
    rlogp, _, rb = OB.forward(sd, args, fr, bio, lens)
    ((rlogp * wl.double()).sum() + (rb * wb.double()).sum()).backward()
    close(logp, rlogp, "logp", 1e-4)
    close(b, rb, "b", 1e-4)
    close(feats.grad, fr.grad, "d feats", 5e-4)
    scale = max(float(v.grad.abs().max()) for k, v in sd.items() if v.grad is not None and k.startswith("bioScoring"))
    for k, p in head.named_parameters():
        if p.grad is None:
            continue
        ref = sd[k].grad
        err = float((p.grad.double().cpu() - ref).abs().max()) / max(float(ref.abs().max()), 1e-3 * scale)
        assert err < 5e-4, (k, err)


def test_c_abi_refuses_what_the_kernel_does_not_serve(dev):
    Lb = lib.load()
    assert Lb.scl_btse_bio_supported(32, 4, 128, 3, 4, 64, 17) == 1
    for bad in ((64, 4, 128, 3, 4, 64, 17), (32, 8, 128, 3, 4, 64, 17), (32, 4, 256, 3, 4, 64, 17), (32, 4, 128, 9, 4, 64, 17),
                (32, 4, 128, 3, 10, 64, 17), (32, 4, 128, 3, 4, 300, 17), (32, 4, 128, 3, 4, 64, 513), (32, 4, 128, 3, 4, 64, 0)):
        assert Lb.scl_btse_bio_supported(*bad) == 0, bad
    d = lib.SclBtseBio()
    d.bio_dim, d.n_heads, d.pf_dim, d.n_layers, d.window, d.bio_out, d.L, d.B = 32, 4, 128, 3, 4, 64, 600, 2
    assert Lb.scl_btse_bio_fwd(ctypes.byref(d), None) == -3 and b"unsupported shape" in Lb.scl_last_error()
    d.L = 17
    assert Lb.scl_btse_bio_fwd(ctypes.byref(d), None) == -1          # null pointers
    assert Lb.scl_btse_bio_bwd(ctypes.byref(d), None) == -1
    assert Lb.scl_btse_join_fwd(None, None, None, None, None, 2, 128, 64, 0, None) == -1
    head = BtseHead(OB.default_args()).to(dev)
    with pytest.raises(lib.SclError, match="no fallback"):
        BtseHead.forward(head, torch.zeros(1, 4, 128, device=dev), torch.zeros(1, 513, dtype=torch.int32), torch.tensor([513]))
    with pytest.raises(IndexError):
        BtseHead.forward(head, torch.zeros(1, 4, 128, device=dev), torch.full((1, 5), 3, dtype=torch.int32), torch.tensor([5]))
    with pytest.raises(ValueError, match="is_add needs bio_out"):
        BtseHead(OB.default_args(is_add=True, bio_out=64))
    # device-resident tokens (bench.py, training): no host round trip in the forward — the check is queued on the device and surfaces at
    # the next forward (or check_tokens()); in-range tokens pass silently
    feats = torch.zeros(1, 4, 128, device=dev)
    BtseHead.forward(head, feats, torch.full((1, 5), 2, dtype=torch.int32, device=dev), torch.tensor([5], device=dev))
    BtseHead.check_tokens(head)
    BtseHead.forward(head, feats, torch.full((1, 5), 7, dtype=torch.int32, device=dev), torch.tensor([5], device=dev))      # clamped by the kernel
    torch.cuda.synchronize()
    with pytest.raises(IndexError, match="earlier forward"):
        BtseHead.forward(head, feats, torch.full((1, 5), 1, dtype=torch.int32, device=dev), torch.tensor([5], device=dev))
    BtseHead.forward(head, feats, torch.tensor([[0, 1, -1, 2, 0]], dtype=torch.int32, device=dev), torch.tensor([5], device=dev))
    with pytest.raises(IndexError):
        BtseHead.check_tokens(head)
    BtseHead.check_tokens(head)          # raised once, then clear


# ---- the whole plugin ------------------------------------------------------------------------------------------------------------------
ARGS = dict(OB.default_args(), name="wav2vec2_btse")


def make_plugin(dev, seed):
    cfg = W.W2VConfig.tiny()
    ssl = W.init_state(cfg, seed=seed)
    m = Model(ARGS, dev, w2v_cfg=W2VConfig.tiny())
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items() if not k.startswith("backend.ssl_model.")}
    head_sd = {k: torch.from_numpy(v) for k, v in fill_state(shapes, seed=seed + 1).items()}
    sd = {"backend.ssl_model.model." + k: v for k, v in ssl.items()}
    sd.update(head_sd)
    m.load_state_dict(sd)
    return m, ssl, head_sd, cfg


def test_state_dict_keys_are_the_references(dev):
    m = Model(ARGS, dev, w2v_cfg=W2VConfig.tiny())
    keys = list(m.state_dict().keys())
    own = [k for k in keys if not k.startswith("backend.ssl_model.")]
    assert sorted(own) == sorted(G["cat:keys"].tolist())          # the reference Model's own key list (SSL encoder injected there)
    sd = m.state_dict()
    for k, shp in zip(G["cat:keys"].tolist(), G["cat:shapes"].tolist()):
        if k != "backend.LL.weight":                               # 16-wide stand-in encoder in the golden
            assert repr(tuple(sd[k].shape)) == shp, k
    assert "backend.ssl_model.model.encoder.layers.0.self_attn.q_proj.weight" in keys and "backend.ssl_model.model.mask_emb" in keys
    lo, hi = m.P.flat.data_ptr(), m.P.flat.data_ptr() + 4 * m.P.n_total
    for n, p in m.named_parameters():
        assert lo <= p.data_ptr() < hi, n
        frozen = "m_utt_level" in n
        if p.requires_grad:
            assert not frozen and m.P.grad.data_ptr() <= p.grad.data_ptr() < m.P.grad.data_ptr() + 4 * m.P.n_train, n
    assert not m.get_parameter("backend.mlp.m_utt_level.weight").requires_grad
    with pytest.raises(RuntimeError, match="needs bio tokens"):
        m(torch.zeros(2, 4000, device=dev))


def test_plugin_forward_and_train_step_against_the_cpu_chain(dev):
    m, ssl, head_sd, cfg = make_plugin(dev, 71)
    m.eval()
    B, Lt = 6, 23
    x = 0.1 * torch.randn(B, 8000, generator=torch.Generator().manual_seed(5))
    y = torch.tensor([1, 1, 1, 0, 0, 0])
    rs = np.random.RandomState(3)
    bio = torch.from_numpy(rs.randint(0, 3, size=(B, Lt)).astype(np.int32))
    lens = torch.tensor([Lt, Lt, 7, Lt, 1, Lt], dtype=torch.int32)
    train_names = [n for n, _, tr in W.param_shapes(cfg) if tr]
    for n in train_names:
        ssl[n].requires_grad_(True)
    hsd = {k: v.clone().requires_grad_(True) for k, v in head_sd.items()}
    enc = W.forward(ssl, cfg, x)
    rlogp, rfeats, rb = OB.forward(hsd, OB.default_args(), enc, bio, lens)
    rloss = OH.model_loss(rlogp, rfeats, rb, y, 1)
    sum(rloss.values()).backward()
    opt = FusedAdamW(m, lr=1e-3, weight_decay=1e-2, overlap=False)
    before = {k: v.clone() for k, v in m.state_dict().items() if "m_utt_level" in k or k == "fc2.weight"}
    out, feats, b = m(x.to(dev), bio, lens)
    assert out.shape == (B, 2) and feats.shape == rfeats.shape and b.shape == (B, 192)
    assert rl2(feats, rfeats) < 1e-2 and rl2(b, rb) < 1e-2 and rl2(out, rlogp) < 1e-2, (rl2(feats, rfeats), rl2(b, rb), rl2(out, rlogp))
    losses = m.loss(out, feats, b, y.to(dev), CONF)
    opt.zero_grad()
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    for k in rloss:
        got, ref = losses[k].item(), rloss[k].item()
        assert abs(got - ref) <= 1e-2 * max(abs(ref), 1e-3), (k, got, ref)
    for k in ("fc2.weight", "bioScoring.bio_scoring.weight", "bioScoring.encoder.attn_layers.0.conv_v.weight", "bioScoring.encoder.ffn_layers.2.conv_1.weight",
              "backend.mlp.m_frame_level.linear_1.weight", "backend.LL.weight"):
        name = k[len("backend."):] if k.startswith("backend.LL.") else k
        assert rl2(m.P.g(name), hsd[k].grad) < 5e-2, (k, rl2(m.P.g(name), hsd[k].grad))
    for n in ("post_extract_proj.weight", "encoder.layers.1.fc1.weight", "feature_extractor.conv_layers.0.0.weight"):
        assert rl2(m.P.g("ssl_model.model." + n), ssl[n].grad) < 8e-2, n
    opt.step()
    torch.cuda.synchronize()
    after = m.state_dict()
    assert torch.equal(after["backend.mlp.m_utt_level.weight"], before["backend.mlp.m_utt_level.weight"])      # no gradient: AdamW skips it
    assert torch.equal(after["backend.mlp.m_utt_level.bias"], before["backend.mlp.m_utt_level.bias"])
    assert not torch.equal(after["fc2.weight"], before["fc2.weight"])
    m.is_train = False
    with torch.no_grad():
        lone = m(x.to(dev), bio, lens)
    assert lone.shape == (B, 2)


def test_tokenizer_hook(dev, tmp_path, monkeypatch):
    (tmp_path / "my_bio.py").write_text("def tok(x, fs):\n    assert fs == 16000\n    return [[1, 0, 2, 1, 1] for _ in range(x.shape[0])]\n")
    monkeypatch.syspath_prepend(str(tmp_path))
    m = Model(dict(ARGS, bio_tokenizer="my_bio:tok"), dev, w2v_cfg=W2VConfig.tiny())
    m.eval()
    x = 0.1 * torch.randn(2, 4000, generator=torch.Generator().manual_seed(1)).to(dev)
    with torch.no_grad():
        a, _, _ = m(x)
        b_, _, _ = m(x, torch.tensor([[1, 0, 2, 1, 1]] * 2, dtype=torch.int32), torch.tensor([5, 5], dtype=torch.int32))
    assert torch.equal(a, b_)


def test_batch_128_step_at_xlsr_shape(dev):
    """configs[4] as one rank sees it: XLS-R-300M encoder, batch 128 x 64000 samples, 199-token bio sequences.  The CPU oracle cannot run
    this size in test time; every utterance is independent in eval mode (no BatchNorm anywhere in this plugin), so rows 0-3 of the batch
    must reproduce the batch-4 forward; then one full train step: finite losses and gradients, every bio-transformer tensor non-zero."""
    m = Model(ARGS, dev, w2v_cfg=W2VConfig())
    m.eval()
    g = torch.Generator().manual_seed(1234)
    B, Lt = 128, 199
    x = (0.1 * torch.randn(B, 64000, generator=g)).to(dev)
    rs = np.random.RandomState(9)
    bio = torch.from_numpy(rs.randint(0, 3, size=(B, Lt)).astype(np.int32))
    lens = torch.full((B,), Lt, dtype=torch.int32)
    with torch.no_grad():
        o4, f4, b4 = [t.clone() for t in m(x[:4], bio[:4], lens[:4])]
        o, f, b = m(x, bio, lens)
    assert rl2(o[:4], o4) < 2e-3 and rl2(f[:4], f4) < 2e-3 and rl2(b[:4], b4) < 2e-3, (rl2(o[:4], o4), rl2(f[:4], f4), rl2(b[:4], b4))
    m.train()
    y = torch.tensor([1] * 58 + [0] * 70).to(dev)
    opt = FusedAdamW(m, lr=1e-5, weight_decay=1e-4)
    out, feats, bvec = m(x, bio, lens)
    losses = m.loss(out, feats, bvec, y, CONF)
    opt.zero_grad()
    sum(losses.values()).backward()
    opt.step()
    torch.cuda.synchronize()
    assert all(torch.isfinite(v).item() for v in losses.values()), losses
    assert 0.3 < losses["L_CE"].item() * B < 3.0
    assert torch.isfinite(m.P.grad).all()
    for n, p in m.named_parameters():
        if n.startswith("bioScoring.") and "conv_k.bias" not in n:
            assert float(p.grad.abs().max()) > 0, n
