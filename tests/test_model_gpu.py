"""End-to-end parity of the HIP model path (tiny wav2vec2 config) against the reference-generated
golden of one train_epoch iteration (tests/golden/train_step.npz, produced by the reference's own
Model / loss / AdamW with the oracle encoder injected) and against the oracle on fresh inputs.

The HIP path computes GEMMs with bf16 operands and fp32 accumulation, so the tolerance is
BASELINE.json's bf16 bar: relative L2 error < 1e-2 (and no single element off by more than 3e-2 of
the tensor's max) for outputs, 2e-2 for the loss terms; gradients (which chain ~40 bf16 GEMMs) are
checked at 4e-2 of each tensor's max and by cosine similarity > 0.995."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from scl_amd.encoder import W2VConfig  # noqa: E402
from scl_amd.model_linear import Model, loss_custom  # noqa: E402
from scl_amd.optim import FusedAdamW  # noqa: E402
from oracle import head as OH  # noqa: E402
from oracle import wav2vec2 as W  # noqa: E402

G = os.path.join(os.path.dirname(__file__), "golden")
ARGS = {"flag_fix_ssl": False, "contra_mode": "all", "loss_type": 1}
CONF = {"model": {"contra_mode": "all", "loss_type": 1}}


def build(dev, ssl_sd, head_sd):
    m = Model(ARGS, dev, w2v_cfg=W2VConfig.tiny())
    sd = {"ssl_model.model." + k: v for k, v in ssl_sd.items()}
    sd.update(head_sd)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("first_bn" in k for k in missing), (missing, unexpected)
    return m


def relerr(got, ref):
    got = torch.as_tensor(got).float().cpu()
    ref = torch.as_tensor(ref).float().cpu()
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-12)).item()


def rl2(got, ref):
    """relative L2 error — the '1e-2 bf16' bar of BASELINE.json is applied to this."""
    got = torch.as_tensor(got).float().cpu()
    ref = torch.as_tensor(ref).float().cpu()
    return ((got - ref).norm() / ref.norm().clamp_min(1e-12)).item()


def close_bf16(got, ref):
    return rl2(got, ref) < 1e-2 and relerr(got, ref) < 3e-2


def cosine(a, b):
    a = torch.as_tensor(a).float().cpu().flatten(); b = torch.as_tensor(b).float().cpu().flatten()
    return (a @ b / (a.norm() * b.norm()).clamp_min(1e-30)).item()


def test_train_step_matches_reference_golden(dev):
    g = np.load(os.path.join(G, "train_step.npz"))
    ssl = W.init_state(W.W2VConfig.tiny(), seed=11)
    head = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd:") and "first_bn" not in k}
    m = build(dev, ssl, head)
    m.eval()   # the golden was generated with dropout off (its RNG cannot be shared)
    x = torch.from_numpy(g["x"]).to(dev); y = torch.from_numpy(g["y"]).to(dev)
    opt = FusedAdamW(m, lr=1e-3, weight_decay=1e-4)
    out, feats, emb = m(x)
    assert close_bf16(out, g["out"]) and close_bf16(emb, g["emb"]) and close_bf16(feats, g["feats"]), \
        (rl2(out, g["out"]), rl2(emb, g["emb"]), rl2(feats, g["feats"]))
    losses = m.loss(out, feats, emb, y, CONF)
    for k in ("L_CE", "L_CF1", "L_CF2"):
        ref = float(g["loss:" + k])
        assert abs(losses[k].item() - ref) <= 1e-2 * max(abs(ref), 1e-3), (k, losses[k].item(), ref)   # north_star: 1e-2 rel at bf16
    train_loss = 0.0
    for v in losses.values():
        train_loss = train_loss + v
    opt.zero_grad()
    train_loss.backward()
    torch.cuda.synchronize()
    report, bad = [], []
    for k in g.files:
        if k.startswith("grad:"):
            name = k[5:]
            got, ref = m.P.g(name), g[k]
            if np.abs(ref).max() < 1e-6:
                # mathematically zero gradient (k_proj.bias: softmax is invariant to a per-query shift);
                # the reference holds fp32 round-off there, we must hold bf16 round-off of the same sum
                ok = got.abs().max().item() < 1e-3
                report.append((name, "zero-grad", got.abs().max().item()))
            else:
                e, c = relerr(got, ref), cosine(got, ref)
                ok = c > 0.995 and e < 4e-2
                report.append((name, e, c))
            if not ok:
                bad.append(report[-1])
    print("\n".join(str(r) for r in report))
    assert not bad, bad
    opt.step()
    torch.cuda.synchronize()
    for k in g.files:
        if k.startswith("post:"):
            name = k[5:]
            ref = torch.from_numpy(g[k])
            got = m.P.f32(name).cpu()
            # AdamW's first step moves every weight by ~lr*sign(g): compare the UPDATE direction and size
            before = (ssl[name[len("ssl_model.model."):]] if name.startswith("ssl_model.model.") else head[name])
            du_ref, du_got = ref - before, got - before
            if ("grad:" + name) in g.files and np.abs(g["grad:" + name]).max() < 1e-6:
                assert (got - ref).abs().max().item() <= 2.2e-3, name   # Adam turns pure round-off into +-lr: direction is noise
                continue
            # Adam's first update is lr*sign(g): a bf16-noise sign flip on a near-zero gradient element is legitimate
            assert (torch.sign(du_got) == torch.sign(du_ref)).float().mean().item() >= 0.93, name
            # |first Adam update| <= lr per element; a sign flip on a near-zero gradient costs at most 2*lr
            assert (got - ref).abs().max().item() <= 2.2e-3, name


def run_trajectory(dev, g, m, opt, sched, report=None):
    """Six steps of main.run_epoch's body (= the reference's train_epoch, main.py:47-84) on the golden's packs; the scheduler steps after
    every third pack (main.py:416).  Returns per-step (losses, lr) and the watched weights after each step."""
    import main as product_main
    names = [k[2:] for k in g.files if k.startswith("w:")]
    losses, lrs, ws = [], [], {n: [] for n in names}
    step = 0
    for ep in range(2):
        for i in range(3):
            x, n, n_packs = product_main._as_model_input(torch.from_numpy(g["x"][step]), dev)          # [1, L, V] pack -> [V, L]
            y = torch.from_numpy(g["y"][step]).view(-1).type(torch.int64).to(dev)
            lrs.append(opt.param_groups[0]["lr"])
            out, feats, emb = m(x)
            l = m.loss(out, feats, emb, y, CONF, "pack%d" % step)
            total = None
            for v in l.values():
                total = v if total is None else total + v
            opt.zero_grad()
            if getattr(opt, "grad_sync", None) is not None:
                opt.grad_sync.begin()
            total.backward()
            opt.step()
            losses.append([l[k].item() for k in ("L_CE", "L_CF1", "L_CF2")])
            for nme in names:
                ws[nme].append(m.P.f32(nme).detach().cpu().clone())
            step += 1
        sched.step()
    return np.array(losses), np.array(lrs), ws


def test_six_step_trajectory_matches_the_reference_train_epoch_and_scheduler(dev, monkeypatch):
    """tests/golden/trajectory.npz: the reference's own train_epoch x AdamW x per-epoch CyclicLR over six packs (oracle/gen_golden.py::
    gen_trajectory; the CPU suite pins the oracle to it at fp32 round-off).  Here the SAME six steps run through the HIP model, FusedAdamW
    and torch's scheduler writing `lr` into it: steps 2..6 exercise what one step cannot — Adam's bias correction at t > 1, the bf16 working
    copy refreshed by every update and read by the next forward, the scheduler's new rate reaching the fused kernel.
    Bars: learning rate exact; L_CE / L_CF2 at north_star's bf16 bar, L_CF1 at 6e-2 (see below) — and the trained trajectory is told apart
    from an untrained one (golden `losses_frozen`: L_CF1 of the untrained weights is 5 % / 14 % away at steps 3 / 6, the GPU must be
    three times closer to the trained value);
    cumulative weight updates w_k - w_0 by direction (cosine) and size.  Adam divides by sqrt(v): where a gradient element is bf16 noise
    around zero its update is +-lr at random, so element-wise equality of updates is not a property the bf16 path can have (the one-step
    test documents the same); the cosine bound is what the GPU measured with margin."""
    from scl_amd import model_linear
    monkeypatch.setattr(model_linear, "DROP_P", 0.0)          # the golden ran train mode with every Dropout p = 0
    g = np.load(os.path.join(G, "trajectory.npz"))
    ssl = W.init_state(W.W2VConfig.tiny(), seed=11)
    head = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd:") and "first_bn" not in k}
    m = build(dev, ssl, head)
    m.train()
    max_lr, min_lr, wd = (float(v) for v in g["hyper"])
    opt = FusedAdamW(m, lr=max_lr, weight_decay=wd)                                                            # main.py:339
    sched = torch.optim.lr_scheduler.CyclicLR(opt, base_lr=min_lr, max_lr=max_lr, step_size_up=3, mode="exp_range", gamma=0.85,
                                              cycle_momentum=False)                                            # main.py:341
    losses, lrs, ws = run_trajectory(dev, g, m, opt, sched)
    assert np.array_equal(lrs, g["lr"]), (lrs, g["lr"])
    ref, frozen = g["losses"], g["losses_frozen"]
    rel = np.abs(losses - ref) / np.maximum(np.abs(ref), 1e-3)
    print("loss rel err per step\n", rel.round(5))
    print("L_CF1 got / ref / frozen\n", np.stack([losses[:, 1], ref[:, 1], frozen[:, 1]], 1).round(4))
    # L_CE / L_CF2 at the bf16 bar; L_CF1 (SupCon over the frame sequence at temperature 0.07: logits of +-14 on a 4-view pack) amplifies
    # the encoder's bf16 error to 3.6e-2 already at step 1, BEFORE any update — its bar is 6e-2, and what pins the trajectory is the
    # comparison below, not this bound
    assert rel[:, [0, 2]].max() < 1e-2 and rel[:, 1].max() < 6e-2, rel
    # the trajectory is distinguishable from not training: wherever the untrained weights' L_CF1 is more than 4 % away from the
    # reference's trained value, the GPU's value is closer to the trained one, and at the most separated step (the last: 14 %) at least
    # twice as close
    far = [k for k in range(1, 6) if abs(frozen[k, 1] - ref[k, 1]) > 4e-2 * abs(ref[k, 1])]
    table = np.stack([losses[:, 1], ref[:, 1], frozen[:, 1]], 1).round(4).tolist()
    assert len(far) >= 2, far
    for k in far:
        assert abs(losses[k, 1] - ref[k, 1]) < abs(losses[k, 1] - frozen[k, 1]), (k, table)
    kmax = max(far, key=lambda k: abs(frozen[k, 1] - ref[k, 1]) / abs(ref[k, 1]))
    assert 2 * abs(losses[kmax, 1] - ref[kmax, 1]) < abs(losses[kmax, 1] - frozen[kmax, 1]), (kmax, table)
    rows = []
    for name in ws:
        w0 = (ssl[name[len("ssl_model.model."):]] if name.startswith("ssl_model.model.") else head[name]).float()
        for k in (0, 2, 5):
            du_ref = torch.from_numpy(g["w:" + name][k]) - w0
            du_got = ws[name][k] - w0
            rows.append((name, k + 1, cosine(du_got, du_ref), (du_got.norm() / du_ref.norm()).item(), (du_got - du_ref).abs().max().item()))
    print("\n".join("%-60s step %d  cos %.4f  |du| ratio %.4f  max abs diff %.2e" % r for r in rows))
    lr_sum = np.cumsum(g["lr"])
    for name, k, c, ratio, mx in rows:
        assert c > TRAJ_COS[k] and 0.9 < ratio < 1.1 and mx <= 2.05 * lr_sum[k - 1], (name, k, c, ratio, mx)


TRAJ_COS = {1: 0.95, 3: 0.98, 6: 0.99}      # measured 0.972 - 1.000 / 0.990 - 1.000 / 0.997 - 1.000 (step 1 = lr * sign(g): every noise-sign element counts fully; later steps average the noise)


@pytest.mark.parametrize("model_kind", ["linear", "aasist"])
def test_adamw_under_the_backward_equals_adamw_after_it(dev, model_kind):
    """FusedAdamW's default on one GPU applies the update slice by slice on a side stream while the backward is still
    running; the numbers must be exactly those of the plain step (same kernel, same slices of the same buffers)."""
    cfg = W.W2VConfig.tiny()
    finals = []
    for overlap in (False, True):
        if model_kind == "linear":
            m = build(dev, W.init_state(cfg, seed=71), OH.init_head(cfg.embed, seed=72))
        else:
            from scl_amd.model_aasist import Model as AModel
            m = AModel({"contra_mode": "all", "loss_type": 1}, dev, w2v_cfg=W2VConfig.tiny(), seed=5)
        m.eval()   # dropout off: both runs see the same function
        opt = FusedAdamW(m, lr=1e-3, weight_decay=1e-4, overlap=overlap)
        assert (opt.overlap is not None) == overlap
        gen = torch.Generator().manual_seed(9)
        for _ in range(3):
            x = (0.1 * torch.randn(4, 20000, generator=gen)).to(dev)
            y = torch.tensor([1, 1, 0, 0], device=dev)
            out, feats, emb = m(x)
            opt.zero_grad()
            sum(m.loss(out, feats, emb, y, CONF).values()).backward()
            opt.step()
        torch.cuda.synchronize()
        finals.append((m.P.flat[: m.P.n_train].clone(), m.P.bf16.clone(), opt.exp_avg.clone(), opt.step_count))
    assert finals[0][3] == finals[1][3] == 3
    if model_kind == "linear":
        assert all(torch.equal(a, b) for a, b in zip(finals[0][:3], finals[1][:3]))
    else:   # the back-ends' pooled / scattered gradients use float atomics (order not fixed between runs): compare to round-off
        assert (finals[0][0] - finals[1][0]).abs().max().item() < 5e-3 * 1e-3 + 2.1e-3


def test_every_parameter_gradient_matches_oracle_autograd(dev):
    """All 300+ trainable tensors, not just the golden's 13: the hand-scheduled backward (bias gradients riding on LayerNorm
    backwards, phase-split conv backward-data, split-K weight gradients, weight-norm backward ...) against torch autograd through
    the fp32 oracle on the same tiny model and batch.  bf16 bar on a tiny noisy model: cosine > 0.99 and max error < 20 % of the tensor max (a wiring mistake gives cosine ~ 0)."""
    import copy
    cfg = W.W2VConfig.tiny()
    ssl, head = W.init_state(cfg, seed=91), OH.init_head(cfg.embed, seed=92)
    m = build(dev, ssl, head)
    m.eval()   # dropout off on both sides
    x = 0.1 * torch.randn(6, 12000, generator=torch.Generator().manual_seed(8))
    y = torch.tensor([1, 1, 1, 0, 0, 0])
    out, feats, emb = m(x.to(dev))
    sum(m.loss(out, feats, emb, y.to(dev), CONF).values()).backward()
    torch.cuda.synchronize()
    _, grads, _, _ = OH.train_step(copy.deepcopy(ssl), copy.deepcopy(head), cfg, x, y, lr=0.0, wd=0.0)
    bad, n = [], 0
    for name, ref in grads.items():
        got = m.P.g(name).cpu()
        n += 1
        if ref.abs().max().item() < 1e-6:
            ok = got.abs().max().item() < 2e-3          # mathematically zero (k_proj.bias): bf16 round-off only
        else:
            # head tensors: a LeakyReLU slope that flips under the bf16 encoder's perturbation is an O(1) error in one element of
            # d(pre); on this 222-frame batch a handful of flips is the whole budget (the head itself is exact:
            # test_head_backward_is_exact_given_the_encoder_output)
            lim = 0.25 if name.startswith("backend.m_frame_level") else 0.2
            ok = cosine(got, ref) > 0.99 and relerr(got, ref) < lim
        if not ok:
            bad.append((name, cosine(got, ref), relerr(got, ref)))
    assert n > 60 and not bad, bad


def test_backward_is_bitwise_reproducible(dev):
    """Every reduction on the path has a fixed order (split-K slabs, ticket-finished column sums, 4-wave combines, SupCon Gram
    chunks): the same step from the same state gives the same bits, outputs and all 300+ gradient tensors."""
    cfg = W.W2VConfig.tiny()
    m = build(dev, W.init_state(cfg, seed=81), OH.init_head(cfg.embed, seed=82))
    m.eval()
    x = (0.1 * torch.randn(5, 9000, generator=torch.Generator().manual_seed(6))).to(dev)
    y = torch.tensor([1, 1, 0, 0, 0], device=dev)
    runs = []
    for _ in range(3):
        out, feats, emb = m(x)
        sum(m.loss(out, feats, emb, y, CONF).values()).backward()
        torch.cuda.synchronize()
        runs.append((out.clone(), feats.clone(), m.P.grad.clone()))
    for r in runs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(runs[0], r))


def test_second_stream_weight_gradients_equal_the_single_stream_backward(dev, monkeypatch):
    """SCL_WGRAD_STREAM=1 moves the weight / bias gradients of the 24-layer loop to a second stream (event-ordered, three rotating
    residual-gradient buffers): same bits as the one-stream backward, on the recorded step and on its replays."""
    cfg = W.W2VConfig.tiny()
    x = (0.1 * torch.randn(5, 9000, generator=torch.Generator().manual_seed(7))).to(dev)
    y = torch.tensor([1, 1, 0, 0, 0], device=dev)
    grads = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("SCL_WGRAD_STREAM", flag)
        m = build(dev, W.init_state(cfg, seed=83), OH.init_head(cfg.embed, seed=84))
        m.eval()
        assert (m.encoder.wstream is not None) == (flag == "1")
        for _ in range(3):
            out, feats, emb = m(x)
            sum(m.loss(out, feats, emb, y, CONF).values()).backward()
        torch.cuda.synchronize()
        grads[flag] = m.P.grad.clone()
    assert torch.equal(grads["0"], grads["1"])


def test_forward_matches_oracle_on_fresh_input_and_eval_scores(dev):
    cfg = W.W2VConfig.tiny()
    ssl = W.init_state(cfg, seed=21)
    head = OH.init_head(cfg.embed, seed=22)
    m = build(dev, ssl, head)
    m.eval()
    gen = torch.Generator().manual_seed(5)
    for B, L in ((3, 4000), (5, 1700), (2, 16000)):
        x = 0.1 * torch.randn(B, L, generator=gen)
        with torch.no_grad():
            ro, rf, re = OH.full_forward(ssl, head, cfg, x)
            out, feats, emb = m(x.to(dev))
        assert out.shape == ro.shape and feats.shape == rf.shape
        assert close_bf16(out, ro) and close_bf16(feats, rf) and close_bf16(emb, re), (B, L, rl2(feats, rf), relerr(feats, rf))
    m.is_train = False   # --predict path (main.py:188): forward returns log-probs only
    with torch.no_grad():
        lp = m(x.to(dev))
    assert lp.shape == (2, 2) and close_bf16(lp, ro)
    m.is_train = True


def test_frozen_ssl_trains_the_head_only(dev):
    """flag_fix_ssl: true (xlsr.py:31-38 runs the encoder under no_grad in eval mode; wav2vec2_linear_nll.py:120-137): same outputs as the
    trainable model in eval mode, head gradients identical, the encoder receives no gradient and AdamW leaves every encoder weight as it was."""
    cfg = W.W2VConfig.tiny()
    ssl, head = W.init_state(cfg, seed=61), OH.init_head(cfg.embed, seed=62)
    sd = {"ssl_model.model." + k: v for k, v in ssl.items()}
    sd.update(head)
    x = (0.1 * torch.randn(4, 4000, generator=torch.Generator().manual_seed(2))).to(dev)
    y = torch.tensor([1, 1, 0, 0], device=dev)
    res = {}
    for frozen in (False, True):
        m = Model(dict(ARGS, flag_fix_ssl=frozen), dev, w2v_cfg=W2VConfig.tiny())
        m.load_state_dict(sd, strict=False)
        m.eval()                                    # dropout off everywhere: the two models differ only in where the backward stops
        opt = FusedAdamW(m, lr=1e-3, weight_decay=1e-2)
        before = m.P.flat[: m.P.n_train].clone()
        out, feats, emb = m(x)
        opt.zero_grad()
        sum(m.loss(out, feats, emb, y, CONF).values()).backward()
        torch.cuda.synchronize()
        grad = m.P.grad[: m.P.n_train].clone()
        opt.step()
        torch.cuda.synchronize()
        res[frozen] = (out, feats, emb, grad, m.P.flat[: m.P.n_train] - before, m.P.off("LL.weight"))
    lo = res[True][5]
    for a, b in zip(res[False][:3], res[True][:3]):
        assert torch.equal(a, b)
    assert torch.equal(res[True][3][lo:], res[False][3][lo:]) and res[False][3][:lo].abs().max().item() > 0
    assert res[True][3][:lo].abs().max().item() == 0.0
    assert res[True][4][:lo].abs().max().item() == 0.0 and res[True][4][lo:].abs().max().item() > 0      # no weight decay on frozen weights either
    assert torch.equal(res[True][4][lo:], res[False][4][lo:])


def test_train_mode_dropout_is_applied_and_backward_consistent(dev):
    cfg = W.W2VConfig.tiny()
    m = build(dev, W.init_state(cfg, seed=31), OH.init_head(cfg.embed, seed=32))
    m.train()
    x = (0.1 * torch.randn(4, 4000, generator=torch.Generator().manual_seed(1))).to(dev)
    y = torch.tensor([1, 1, 0, 0], device=dev)
    out1, _, emb1 = m(x)
    out2, _, emb2 = m(x)
    assert not torch.allclose(emb1, emb2)   # fresh dropout mask per forward
    losses = m.loss(*m(x), y, CONF)
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    gnorm = m.P.grad.norm().item()
    assert np.isfinite(gnorm) and gnorm > 0


def test_state_dict_names_follow_the_reference(dev):
    m = Model(ARGS, dev, w2v_cfg=W2VConfig.tiny())
    keys = set(m.state_dict().keys())
    for k in ("ssl_model.model.feature_extractor.conv_layers.0.0.weight", "ssl_model.model.feature_extractor.conv_layers.6.2.1.bias",
              "ssl_model.model.post_extract_proj.weight", "ssl_model.model.encoder.pos_conv.0.weight_g",
              "ssl_model.model.encoder.layers.1.self_attn.k_proj.bias", "ssl_model.model.encoder.layer_norm.weight",
              "ssl_model.model.mask_emb", "ssl_model.model.quantizer.vars", "ssl_model.model.final_proj.weight", "LL.weight",
              "first_bn.running_mean", "first_bn1.num_batches_tracked", "backend.m_frame_level.3.weight", "backend.m_utt_level.bias"):
        assert k in keys, k
    ref_names = {"ssl_model.model." + n for n, _, _ in W.param_shapes(W.W2VConfig.tiny())}
    assert ref_names <= keys


@pytest.mark.parametrize("nclip,L", [(5, 6000), (4, 80000)])
def test_head_dim_64_config_uses_fused_attention_and_matches_oracle(dev, nclip, L):
    """A small encoder with 64-wide heads takes the fused attention kernels (scores stay on chip): forward, losses and
    gradients must still match the oracle's autograd.  80000-sample clips (T = 249 frames: longer than the fused backward's 224 and than
    any clip the reference's loaders produce, but a legal trim_length) must take the materialised-score path and the pos-conv fall-backs."""
    from scl_amd.encoder import W2VConfig
    kw = dict(conv_dim=32, embed=128, layers=2, heads=2, ffn=256, pos_k=16, pos_groups=4, final_dim=16, latent_vars=8, latent_groups=2)
    ocfg, cfg = W.W2VConfig(**kw), W2VConfig(**kw)
    ssl, head = W.init_state(ocfg, seed=41), OH.init_head(ocfg.embed, seed=42)
    m = Model(ARGS, dev, w2v_cfg=cfg)
    sd = {"ssl_model.model." + k: v for k, v in ssl.items()}
    sd.update(head)
    m.load_state_dict(sd, strict=False)
    m.eval()
    x = 0.1 * torch.randn(nclip, L, generator=torch.Generator().manual_seed(3))
    y = torch.tensor([1, 1, 1, 0, 0] if nclip == 5 else [1, 1, 0, 0])
    out, feats, emb = m(x.to(dev))
    assert m.encoder.bufs(nclip, L)["fused_attn"] == (L == 6000)
    losses = m.loss(out, feats, emb, y.to(dev), CONF)
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    ref_losses, ref_grads, (ro, rf, re), _ = OH.train_step(ssl, head, ocfg, x, y)
    assert close_bf16(out, ro) and close_bf16(feats, rf) and close_bf16(emb, re), (rl2(feats, rf), relerr(feats, rf))
    for k, v in ref_losses.items():
        assert abs(losses[k].item() - v) <= 2e-2 * max(abs(v), 1e-3), k
    for name in ("ssl_model.model.encoder.layers.0.self_attn.q_proj.weight", "ssl_model.model.encoder.layers.0.self_attn.v_proj.weight",
                 "ssl_model.model.encoder.layers.1.self_attn.k_proj.weight", "ssl_model.model.encoder.layers.1.self_attn.out_proj.weight",
                 "ssl_model.model.feature_extractor.conv_layers.2.0.weight", "LL.weight"):
        c = cosine(m.P.g(name), ref_grads[name])
        assert c > (0.99 if name == "LL.weight" else 0.995), (name, c)      # LL sits under the head's LeakyReLU slope flips (90 frames here)


def test_full_size_xlsr_forward_matches_oracle(dev):
    """XLS-R-300M shape (24 layers, 1024 wide, 16 heads of 64), seeded random weights, 2 x 16000-sample clips (BASELINE
    config-1 length): log-probs / feats / emb of the HIP path against the fp32 CPU oracle at the bf16 bar (1e-2 rel-L2)."""
    from scl_amd.encoder import W2VConfig
    ocfg = W.W2VConfig()
    ssl, head = W.init_state(ocfg, seed=51), OH.init_head(ocfg.embed, seed=52)
    m = Model(ARGS, dev, w2v_cfg=W2VConfig())
    sd = {"ssl_model.model." + k: v for k, v in ssl.items()}
    sd.update(head)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected
    m.eval()
    x = 0.1 * torch.randn(2, 16000, generator=torch.Generator().manual_seed(1234))
    with torch.no_grad():
        ro, rf, re = OH.full_forward(ssl, head, ocfg, x)
        out, feats, emb = m(x.to(dev))
    print("full-size rel-L2: out %.2e feats %.2e emb %.2e ; max-rel feats %.2e" % (rl2(out, ro), rl2(feats, rf), rl2(emb, re), relerr(feats, rf)))
    assert rl2(out, ro) < 1e-2 and rl2(feats, rf) < 1e-2 and rl2(emb, re) < 1e-2
    assert (out.argmax(1).cpu() == ro.argmax(1)).all()


def _full_size_model(dev, seed_ssl=61, seed_head=62):
    from scl_amd.encoder import W2VConfig
    ocfg = W.W2VConfig()
    ssl, head = W.init_state(ocfg, seed=seed_ssl), OH.init_head(ocfg.embed, seed=seed_head)
    m = Model(ARGS, dev, w2v_cfg=W2VConfig())
    sd = {"ssl_model.model." + k: v for k, v in ssl.items()}
    sd.update(head)
    _, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected
    m.eval()          # dropout off: the oracle's masks cannot be shared
    return m, ssl, head, ocfg


# one tensor per GEMM layout class of the backward (conv wgrad with overlapping rows, grouped pos-conv weight-norm pair, q/k/v/out, fc1/fc2,
# LayerNorm parameters, the head's LL) in a shallow, a middle and the deepest layer
FULL_SIZE_GRADS = [
    "ssl_model.model.feature_extractor.conv_layers.0.0.weight", "ssl_model.model.feature_extractor.conv_layers.1.0.weight",
    "ssl_model.model.feature_extractor.conv_layers.6.0.weight", "ssl_model.model.feature_extractor.conv_layers.3.2.1.weight",
    "ssl_model.model.post_extract_proj.weight", "ssl_model.model.encoder.pos_conv.0.weight_v", "ssl_model.model.encoder.pos_conv.0.weight_g",
    "ssl_model.model.encoder.layers.0.self_attn.q_proj.weight", "ssl_model.model.encoder.layers.0.self_attn.k_proj.weight",
    "ssl_model.model.encoder.layers.0.self_attn.v_proj.weight", "ssl_model.model.encoder.layers.0.self_attn.out_proj.weight",
    "ssl_model.model.encoder.layers.0.fc1.weight", "ssl_model.model.encoder.layers.0.fc2.weight", "ssl_model.model.encoder.layers.0.fc1.bias",
    "ssl_model.model.encoder.layers.11.fc1.weight", "ssl_model.model.encoder.layers.11.self_attn.q_proj.bias",
    "ssl_model.model.encoder.layers.23.self_attn.out_proj.weight", "ssl_model.model.encoder.layers.23.fc2.weight",
    "ssl_model.model.encoder.layers.23.final_layer_norm.weight", "ssl_model.model.encoder.layer_norm.bias", "LL.weight", "LL.bias",
    "backend.m_frame_level.0.weight", "backend.m_utt_level.weight",
]


def head_only_reference(m, B, L, y, head_sd):
    """The oracle's head + losses (torch fp32 autograd, CPU) run on the encoder output the GPU step itself produced, with LL.weight at
    the bf16 rounding the GPU's LL GEMM multiplies by: what the hand-written head backward must reproduce when nothing upstream differs.
    Returns (gradients of the head tensors, gradient w.r.t. the encoder output [B*T, E])."""
    d = m.encoder.bufs(B, L)
    T, E = d["T"], m.cfg.embed
    enc = d["out"][: B * T * E].float().cpu().view(B, T, E).clone().requires_grad_(True)
    sd = {k: v.clone().float() for k, v in head_sd.items()}
    sd["LL.weight"] = sd["LL.weight"].to(torch.bfloat16).float()
    for v in sd.values():
        v.requires_grad_(True)
    out, feats, emb = OH.head_forward(sd, enc)
    sum(OH.model_loss(out, feats, emb, y).values()).backward()
    return {k: v.grad for k, v in sd.items()}, enc.grad.reshape(B * T, E)


HEAD_TENSORS = ["backend.m_frame_level.0.weight", "backend.m_frame_level.0.bias", "backend.m_frame_level.3.weight", "backend.m_frame_level.3.bias",
                "backend.m_frame_level.6.weight", "backend.m_frame_level.6.bias", "backend.m_utt_level.weight", "backend.m_utt_level.bias"]


def test_head_backward_is_exact_given_the_encoder_output(dev):
    """The frame-level head (three 128 x 128 linears + LeakyReLU + mean pool + utterance linear + losses) runs in f32 on the exact-fp32
    GEMM: fed the SAME encoder output, its parameter gradients agree with torch's fp32 autograd to 1e-4 rel-L2 (LL, whose GEMMs take
    bf16 operands, and the gradient handed to the encoder: 1e-2).  What is left between the whole step and the fp32 oracle in these
    tensors (tests below) therefore comes from upstream: a bf16-perturbed pre-activation that changes sign flips its LeakyReLU slope."""
    cfg = W.W2VConfig.tiny()
    ssl, head = W.init_state(cfg, seed=91), OH.init_head(cfg.embed, seed=92)
    m = build(dev, ssl, head)
    m.eval()
    B, L = 6, 12000
    x = 0.1 * torch.randn(B, L, generator=torch.Generator().manual_seed(8))
    y = torch.tensor([1, 1, 1, 0, 0, 0])
    out, feats, emb = m(x.to(dev))
    sum(m.loss(out, feats, emb, y.to(dev), CONF).values()).backward()
    torch.cuda.synchronize()
    ref, denc = head_only_reference(m, B, L, y, head)
    for name in HEAD_TENSORS:
        assert rl2(m.P.g(name), ref[name]) < 1e-4, (name, rl2(m.P.g(name), ref[name]))
    assert rl2(m.P.g("LL.weight"), ref["LL.weight"]) < 1e-2 and rl2(m.P.g("LL.bias"), ref["LL.bias"]) < 1e-4
    T = m.encoder.bufs(B, L)["T"]
    got = m._head_bufs(B, T)["denc"][: B * T * cfg.embed].float().cpu().view(B * T, cfg.embed)
    assert rl2(got, denc) < 1e-2


def test_full_size_train_step_matches_oracle_at_baseline_shape(dev):
    """BASELINE shape: XLS-R-300M encoder (24 x 1024 x 16 heads x 4096), 4 x 64000-sample clips (T = 199), dropout off — forward
    outputs, the three loss terms (north_star bar: 1e-2 relative at bf16) and 24 gradient tensors covering every GEMM layout
    class of the backward, against oracle.head.train_step (torch fp32 autograd on the CPU, pinned to the reference's Model /
    loss by tests/golden/train_step.npz).  Model.loss here is the reference's (wav2vec2_linear_nll.py:158-192)."""
    m, ssl, head, ocfg = _full_size_model(dev)
    x = 0.1 * torch.randn(4, 64000, generator=torch.Generator().manual_seed(1234))
    y = torch.tensor([1, 1, 0, 0])
    out, feats, emb = m(x.to(dev))
    losses = m.loss(out, feats, emb, y.to(dev), CONF)
    total = sum(losses.values())
    for p in m.parameters():
        p.grad = None
    total.backward()
    torch.cuda.synchronize()
    href, denc = head_only_reference(m, 4, 64000, y, head)      # before train_step: its AdamW update moves `head` in place
    head0 = {k: v.clone() for k, v in head.items()}
    ref_losses, ref_grads, (ro, rf, re), _ = OH.train_step(ssl, head, ocfg, x, y)
    print("4 x 64000 rel-L2: out %.2e feats %.2e emb %.2e" % (rl2(out, ro), rl2(feats, rf), rl2(emb, re)))
    assert rl2(out, ro) < 1e-2 and rl2(feats, rf) < 1e-2 and rl2(emb, re) < 1e-2
    for k, v in ref_losses.items():
        print("loss %s: %.6f vs oracle %.6f (rel %.2e)" % (k, losses[k].item(), v, abs(losses[k].item() - v) / max(abs(v), 1e-6)))
        assert abs(losses[k].item() - v) <= 1e-2 * max(abs(v), 1e-3), (k, losses[k].item(), v)
    bad = []
    for name in FULL_SIZE_GRADS:
        got, ref = m.P.g(name).float().cpu().flatten(), ref_grads[name].flatten()
        e, c = rl2(got, ref), cosine(got, ref)
        print("grad %-70s rel-L2 %.2e cos %.6f" % (name, e, c))
        # the frame-level head tensors sit behind three LeakyReLUs: where the bf16 encoder moves a pre-activation across zero the slope
        # jumps 1 <-> 0.01, a sparse O(1) error in d(pre) that no head precision removes (f32 head 7.1e-2, bf16 head 8.1e-2 here);
        # the head itself is exact — checked right below on the GPU's own encoder output
        lim_e, lim_c = (1.0e-1, 0.995) if name.startswith("backend.m_frame_level") else (6e-2, 0.998)
        if not (e < lim_e and c > lim_c):
            bad.append((name, e, c))
    assert not bad, bad
    for name in HEAD_TENSORS:
        e = rl2(m.P.g(name), href[name])
        print("head-only %-40s rel-L2 %.2e" % (name, e))
        assert e < 1e-4, (name, e)
    assert rl2(m.P.g("LL.weight"), href["LL.weight"]) < 1e-2
    flips = 0.0   # fraction of frame-level pre-activations whose sign differs between the step and the fp32 oracle's forward
    with torch.no_grad():
        h_g = torch.relu(feats.float().cpu()); h_o = torch.relu(rf)
        for idx in (0, 3, 6):
            w, b_ = head0["backend.m_frame_level.%d.weight" % idx], head0["backend.m_frame_level.%d.bias" % idx]
            pg, po = torch.nn.functional.linear(h_g, w, b_), torch.nn.functional.linear(h_o, w, b_)
            flips = max(flips, ((pg > 0) != (po > 0)).float().mean().item())
            h_g, h_o = torch.nn.functional.leaky_relu(pg, 0.01), torch.nn.functional.leaky_relu(po, 0.01)
    print("LeakyReLU slope flips vs the fp32 oracle: %.3f %% of the pre-activations of the worst layer" % (100 * flips))
    assert 0.0 < flips < 0.02


@pytest.mark.parametrize("B", [64, 32])
def test_batch_64_auto_selected_tiles_agree_with_the_validated_small_batch(dev, B):
    """bench.py's shapes — BASELINE configs[2] (batch 64 x 64000: M = 12736 rows, where the wide-tile GEMM kernels and their split-K plans
    are picked automatically) and configs[1] (batch 32: M = 6368 rows, which leaves the N = 1024 linears on the 128 x 128 kernel and picks
    other split-K plans) — cannot be run through the CPU oracle in test time.  Every utterance's forward is independent of the others, and
    the row tiles only re-partition M, so the first 4 rows of a batch-64 step must reproduce the batch-4 step that the test above
    checks against the oracle: outputs to bf16 round-off, and the per-utterance losses entering a batch of the same labels."""
    m, _, _, _ = _full_size_model(dev)
    g = torch.Generator().manual_seed(1234)
    x4 = 0.1 * torch.randn(4, 64000, generator=g)
    x64 = torch.cat([x4, 0.1 * torch.randn(B - 4, 64000, generator=g)]).to(dev)
    with torch.no_grad():
        o4, f4, e4 = [t.clone() for t in m(x4.to(dev))]
        o64, f64, e64 = m(x64)
    print("batch %d" % B, "vs batch 4 rel-L2: out %.2e feats %.2e emb %.2e" % (rl2(o64[:4], o4.cpu()), rl2(f64[:4], f4.cpu()), rl2(e64[:4], e4.cpu())))
    assert rl2(o64[:4], o4.cpu()) < 2e-3 and rl2(f64[:4], f4.cpu()) < 2e-3 and rl2(e64[:4], e4.cpu()) < 2e-3
    # one full train step at this size: finite losses in the band of a random-init model, finite gradients everywhere
    m.train()
    y = torch.tensor(([1] * (29 * B // 64) + [0] * B)[:B], device=dev)
    out, feats, emb = m(x64)
    losses = m.loss(out, feats, emb, y, CONF)
    total = sum(losses.values())
    total.backward()
    torch.cuda.synchronize()
    assert 0.0 < total.item() < 64.0 / B and torch.isfinite(m.P.grad[: m.P.n_train]).all()


def test_fp32_scoring_path_matches_oracle_to_1e3_at_xlsr_shape(dev):
    """main.py --eval / --predict / --emb score with fp32 activations, the fp32 master weights and the exact-fp32 matrix-core GEMM
    (Model._score_fp32 / Encoder.forward_f32): XLS-R-300M shape, 2 x 64600-sample clips (the eval pad length, T = 201), against the
    fp32 CPU oracle — per-utterance log-probs, embeddings and frame features within north_star's fp32 bar, 1e-3 of the tensor's
    scale (the bf16 training kernels give ~1e-2 on the same input, printed for comparison)."""
    import scl_amd.model_linear as ML
    m, ssl, head, ocfg = _full_size_model(dev, 71, 72)
    x = 0.1 * torch.randn(2, 64600, generator=torch.Generator().manual_seed(99))
    with torch.no_grad():
        ro, rf, re = OH.full_forward(ssl, head, ocfg, x)
        m.is_train = True
        out, feats, emb = m(x.to(dev))
        old, ML.SCORE_FP32 = ML.SCORE_FP32, False
        try:
            ob, fb, eb = m(x.to(dev))
        finally:
            ML.SCORE_FP32 = old
    mx = lambda a, b: ((a.float().cpu() - b).abs().max() / b.abs().max()).item()
    print("fp32 scoring: max-rel logp %.2e emb %.2e feats %.2e | bf16 kernels: logp %.2e emb %.2e feats %.2e" %
          (mx(out, ro), mx(emb, re), mx(feats, rf), mx(ob, ro), mx(eb, re), mx(fb, rf)))
    assert mx(out, ro) < 1e-3 and mx(emb, re) < 1e-3 and mx(feats, rf) < 1e-3
    assert (out.argmax(1).cpu() == ro.argmax(1)).all()


def test_fp32_scoring_path_on_confident_scores(dev):
    """The same bar on scores that look like a trained model's: the utterance-level layer amplified until the widest margin is ~8 nats, so
    that the log-probs spread over 0 ... -8 instead of sitting at log 0.5 (random-init weights say little about the tails, where EER is decided).  Six clips of different
    loudness; the error of every log-prob against the fp32 CPU oracle within 1e-3 of the largest |log-prob|, decisions identical."""
    m, ssl, head, ocfg = _full_size_model(dev, 73, 74)
    g = torch.Generator().manual_seed(100)
    x = torch.randn(6, 64600, generator=g) * torch.tensor([0.02, 0.05, 0.1, 0.2, 0.4, 0.8])[:, None]
    with torch.no_grad():
        _, _, re = OH.full_forward(ssl, head, ocfg, x)
        w, b = head["backend.m_utt_level.weight"], head["backend.m_utt_level.bias"]
        margin = (re @ (w[0] - w[1])) + (b[0] - b[1])
        amp = 8.0 / (margin - margin.mean()).abs().max().item()          # the widest margin becomes ~8 nats around the mean
        head = dict(head)
        head["backend.m_utt_level.weight"] = w * amp
        head["backend.m_utt_level.bias"] = b * amp - torch.stack([margin.mean() * amp / 2, -margin.mean() * amp / 2])
        m.load_state_dict({k: head[k] for k in ("backend.m_utt_level.weight", "backend.m_utt_level.bias")}, strict=False)
        ro = torch.log_softmax(torch.nn.functional.linear(re, head["backend.m_utt_level.weight"], head["backend.m_utt_level.bias"]), 1)
        out, feats, emb = m(x.to(dev))
    spread = ro.abs().max().item()
    err = (out.float().cpu() - ro).abs().max().item()
    print("confident scores: log-probs\n", ro.numpy().round(4), "\nmax abs err %.2e (spread %.2f)" % (err, spread))
    assert spread > 2.0, spread                              # the amplification did produce tails
    assert err < 1e-3 * spread
    assert (out.argmax(1).cpu() == ro.argmax(1)).all()


@pytest.mark.parametrize("layerdrop", [0.0, 0.35])
def test_carried_over_weight_gradient_launches_equal_the_split_k_path(dev, layerdrop):
    """The grouped weight-gradient launches with carry-over (scl_amd/encoder.py::_flush_slabs: tile ranges of several layers' problems
    in one launch, operands alive one layer longer) against one split-K launch per gradient, on the SAME forward: XLS-R-300M shape,
    6 x 64000 samples (M = 1194 rows: zero rows up to 1216 behind the reduction operands), with and without LayerDrop (skipped layers
    contribute no tiles; the flush rule counts processed layers).  Every encoder weight gradient must agree to fp32 summation round-off
    (the two paths add the same products in different orders) — a clobbered operand would show as O(1)."""
    from scl_amd import encoder as E
    from scl_amd.encoder import W2VConfig
    torch.manual_seed(11)
    m = Model(ARGS, dev, w2v_cfg=W2VConfig(encoder_layerdrop=layerdrop))
    m.train()
    x = (0.1 * torch.randn(6, 64000, generator=torch.Generator().manual_seed(5))).to(dev)
    y = torch.tensor([1, 1, 1, 0, 0, 0], device=dev)
    grads = {}
    saved = (E.WGRAD_GROUP, E.WGRAD_CARRY)
    try:
        for mode, (grp, carry) in (("split-k", (False, False)), ("group", (True, False)), ("carry", (True, True))):
            E.WGRAD_GROUP, E.WGRAD_CARRY = grp, carry
            for st in m._states.values():
                st["plans"].clear()                   # a recorded launch plan would replay the first mode's launches
            torch.manual_seed(123)                    # the same layers are dropped in every mode
            m._step_seed = 777                        # ... and the same head-dropout masks drawn
            out, feats, emb = m(x)
            losses = m.loss(out, feats, emb, y, CONF)
            m.P.grad.zero_()
            sum(losses.values()).backward()
            torch.cuda.synchronize()
            grads[mode] = m.P.grad.clone()
    finally:
        E.WGRAD_GROUP, E.WGRAD_CARRY = saved
    ref = grads["split-k"]
    assert torch.isfinite(ref).all() and float(ref.abs().max()) > 0
    for mode in ("group", "carry"):
        for name in ("encoder.layers.23.fc2.weight", "encoder.layers.23.self_attn.q_proj.weight", "encoder.layers.12.fc1.weight",
                     "encoder.layers.11.self_attn.out_proj.weight", "encoder.layers.0.fc2.weight", "encoder.layers.0.self_attn.v_proj.weight"):
            o, n_, shape, _ = m.P.index["ssl_model.model." + name]
            a, b = grads[mode][o:o + n_], ref[o:o + n_]
            scale = float(b.abs().max())
            if scale == 0:                            # a dropped layer: zero on both sides
                assert float(a.abs().max()) == 0, (mode, name)
                continue
            assert float((a - b).abs().max()) < 2e-5 * scale, (mode, name, float((a - b).abs().max()) / scale)
        whole = float((grads[mode] - ref).norm() / ref.norm())
        assert whole < 1e-5, (mode, whole)
    if layerdrop > 0:
        zero_layers = sum(1 for n in range(24) if float(ref[m.P.index["ssl_model.model.encoder.layers.%d.fc1.weight" % n][0]:][:1024].abs().max()) == 0)
        assert 1 <= zero_layers <= 20, zero_layers
