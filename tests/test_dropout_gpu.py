"""Element dropout inside the encoder (fairseq Wav2Vec2Config.dropout / attention_dropout / activation_dropout / dropout_input; the reference
runs XLS-R in train mode with whatever its checkpoint's cfg carries, model/xlsr.py:33-41).  The kernels draw their keep-masks from a
counter hash of (site seed, element index); the test re-implements that hash in numpy, hands the SAME masks to the fp32 oracle
(oracle.wav2vec2.forward(masks=...)) and compares outputs, losses and gradients — on the recorded step and on replayed launch plans."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from scl_amd import ops  # noqa: E402
from scl_amd.encoder import Encoder, W2VConfig  # noqa: E402
from scl_amd.model_linear import DROP_P, Model  # noqa: E402
from oracle import head as OH  # noqa: E402
from oracle import wav2vec2 as W  # noqa: E402

ARGS = {"flag_fix_ssl": False, "contra_mode": "all", "loss_type": 1}
CONF = {"model": {"contra_mode": "all", "loss_type": 1}}
M32 = np.uint64(0xFFFFFFFF)


def hash_u32(seed, idx):
    """csrc/common.h::hash_u32, bit for bit (idx: uint64 array)."""
    idx = idx.astype(np.uint64)
    seed = np.uint64(seed)
    x = ((idx & M32) * np.uint64(0x9E3779B1) & M32) ^ (((idx >> np.uint64(32)) * np.uint64(0x85EBCA77)) & M32) ^ seed
    x ^= x >> np.uint64(16); x = x * np.uint64(0x7feb352d) & M32
    x ^= x >> np.uint64(15); x = x * np.uint64(0x846ca68b) & M32
    x ^= x >> np.uint64(16)
    x = (x + (seed * np.uint64(0xC2B2AE3D) & M32)) & M32
    x ^= x >> np.uint64(15); x = x * np.uint64(0x2c1b3c6d) & M32
    x ^= x >> np.uint64(12); x = x * np.uint64(0x297a2d39) & M32
    x ^= x >> np.uint64(15)
    return x


def keep_scale(seed, n, p):
    """csrc/common.h::dropout_scale over element indices 0 .. n-1 -> float32 factors (0 or 1 / (1 - p))."""
    u = (hash_u32(seed, np.arange(n, dtype=np.uint64)) >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return torch.from_numpy(np.where(u >= np.float32(p), np.float32(1.0) / (np.float32(1.0) - np.float32(p)), np.float32(0.0)).astype(np.float32))


def rl2(got, ref):
    got, ref = torch.as_tensor(got).float().cpu(), torch.as_tensor(ref).float().cpu()
    return ((got - ref).norm() / ref.norm().clamp_min(1e-12)).item()


def cosine(a, b):
    a = torch.as_tensor(a).float().cpu().flatten(); b = torch.as_tensor(b).float().cpu().flatten()
    return (a @ b / (a.norm() * b.norm()).clamp_min(1e-30)).item()


def test_dropout_kernel_draws_the_mask_the_numpy_port_predicts(dev):
    n, seed, p = 100_003, 0x1234567, 0.3
    x = torch.randn(n, generator=torch.Generator().manual_seed(0))
    y = torch.empty(n, device=dev); yb = torch.empty(n, dtype=torch.bfloat16, device=dev)
    ops.dropout(x.to(dev), y, yb, n, seed, p)
    k = keep_scale(seed, n, p)
    assert torch.equal(y.cpu(), x * k) and torch.equal(yb.cpu(), (x * k).to(torch.bfloat16))
    assert abs((k == 0).float().mean().item() - p) < 5e-3


def enc_masks_for(step_seed, cfg, B, T, probs):
    p_res, p_attn, p_act, p_in = probs
    E, H, Fd = cfg.embed, cfg.heads, cfg.ffn
    ss = lambda layer, site: Encoder.site_seed(step_seed, layer, site)
    masks = {}
    if p_in > 0:
        masks["in"] = keep_scale(ss(-1, Encoder.SITE_IN), B * T * E, p_in).view(B, T, E)
    if p_res > 0:
        masks["enc"] = keep_scale(ss(-1, Encoder.SITE_ENC), B * T * E, p_res).view(B, T, E)
    for n in range(cfg.layers):
        m = {}
        if p_attn > 0:
            m["attn"] = keep_scale(ss(n, Encoder.SITE_ATTN), B * H * T * T, p_attn).view(B, H, T, T)
        if p_res > 0:
            m["d1"] = keep_scale(ss(n, Encoder.SITE_1), B * T * E, p_res).view(B, T, E)
            m["d3"] = keep_scale(ss(n, Encoder.SITE_3), B * T * E, p_res).view(B, T, E)
        if p_act > 0:
            m["d2"] = keep_scale(ss(n, Encoder.SITE_2), B * T * Fd, p_act).view(B, T, Fd)
        masks[n] = m
    return masks


@pytest.mark.parametrize("probs,heads", [((0.1, 0.1, 0.1, 0.1), 2), ((0.2, 0.0, 0.0, 0.0), 2), ((0.0, 0.15, 0.05, 0.0), 2),
                                         ((0.0, 0.15, 0.05, 0.0), 4), ((0.1, 0.1, 0.1, 0.1), 4)])
def test_encoder_dropout_matches_the_oracle_given_the_same_masks(dev, probs, heads):
    """(dropout, attention_dropout, activation_dropout, dropout_input) on a 2-layer encoder with 64-wide heads (the fused attention
    kernels) or 32-wide heads (the un-fused path: materialised probabilities, scl_dropout_rows with the fused kernels' mask index),
    head dropout on as well (train mode, p = 0.5): three steps, the last two replayed from launch plans; after each the
    step's masks are rebuilt on the host and the oracle's autograd through them must agree at the bf16 bar."""
    kw = dict(conv_dim=32, embed=128, layers=2, heads=heads, ffn=256, pos_k=16, pos_groups=4, final_dim=16, latent_vars=8, latent_groups=2)
    ocfg = W.W2VConfig(**kw)
    cfg = W2VConfig(dropout=probs[0], attention_dropout=probs[1], activation_dropout=probs[2], dropout_input=probs[3], **kw)
    ssl, head = W.init_state(ocfg, seed=41), OH.init_head(ocfg.embed, seed=42)
    m = Model(ARGS, dev, w2v_cfg=cfg)
    sd = {"ssl_model.model." + k: v for k, v in ssl.items()}
    sd.update(head)
    m.load_state_dict(sd, strict=False)
    m.train()
    B, L = 5, 6000
    x = 0.1 * torch.randn(B, L, generator=torch.Generator().manual_seed(3))
    y = torch.tensor([1, 1, 1, 0, 0])
    T = cfg.conv_lens(L)[-1]
    for step in range(3):
        out, feats, emb = m(x.to(dev))
        losses = m.loss(out, feats, emb, y.to(dev), CONF)
        for p_ in m.parameters():
            p_.grad = None
        sum(losses.values()).backward()
        torch.cuda.synchronize()
        if step == 1:
            continue
        step_seed = m._step_seed
        enc_masks = enc_masks_for(step_seed, cfg, B, T, probs)
        head_masks = [keep_scale((step_seed + 7919 * j) & 0x7FFFFFFF, B * T * 128, DROP_P).view(B, T, 128) for j in range(3)]
        import copy
        ref_losses, ref_grads, (ro, rf, re), _ = OH.train_step(copy.deepcopy(ssl), copy.deepcopy(head), ocfg, x, y, lr=0.0, wd=0.0,
                                                               dropout_masks=head_masks, enc_masks=enc_masks)
        assert rl2(feats, rf) < 1.5e-2 and rl2(emb, re) < 2e-2 and rl2(out, ro) < 2e-2, (step, rl2(feats, rf), rl2(emb, re), rl2(out, ro))
        for k, v in ref_losses.items():
            assert abs(losses[k].item() - v) <= 3e-2 * max(abs(v), 1e-3), (step, k, losses[k].item(), v)
        for name in ("ssl_model.model.encoder.layers.0.self_attn.q_proj.weight", "ssl_model.model.encoder.layers.0.self_attn.v_proj.weight",
                     "ssl_model.model.encoder.layers.0.self_attn.q_proj.bias", "ssl_model.model.encoder.layers.0.self_attn.out_proj.bias",
                     "ssl_model.model.encoder.layers.1.self_attn.k_proj.weight", "ssl_model.model.encoder.layers.1.self_attn.out_proj.weight",
                     "ssl_model.model.encoder.layers.0.fc1.weight", "ssl_model.model.encoder.layers.0.fc1.bias", "ssl_model.model.encoder.layers.1.fc2.weight",
                     "ssl_model.model.encoder.layers.1.fc2.bias", "ssl_model.model.encoder.layers.1.final_layer_norm.weight",
                     "ssl_model.model.post_extract_proj.weight", "ssl_model.model.post_extract_proj.bias", "ssl_model.model.encoder.pos_conv.0.weight_v",
                     "ssl_model.model.feature_extractor.conv_layers.2.0.weight"):
            c = cosine(m.P.g(name), ref_grads[name])
            assert c > 0.99, (step, name, c)
    # a different mask every step, and none in eval mode
    o1 = m(x.to(dev))[1].clone(); o2 = m(x.to(dev))[1].clone()
    assert not torch.equal(o1, o2)
    m.eval()
    with torch.no_grad():
        e1 = m(x.to(dev))[1].clone(); e2 = m(x.to(dev))[1].clone()
    assert torch.equal(e1, e2)


def test_checkpoint_probabilities_reach_the_encoder(dev, tmp_path):
    """A fairseq checkpoint whose cfg asks for element dropout is no longer refused: its probabilities become the encoder's."""
    from scl_amd import checkpoint
    cfg = W2VConfig.tiny()
    m = Model(ARGS, dev, w2v_cfg=cfg)
    sd = {k[len("ssl_model.model."):]: v.detach().cpu().clone() for k, v in m.state_dict().items() if k.startswith("ssl_model.model.")}
    import argparse
    path = str(tmp_path / "w2v.pt")
    torch.save({"model": sd, "args": argparse.Namespace(dropout=0.1, attention_dropout=0.05, activation_dropout=0.0, dropout_input=0.2,
                                                        dropout_features=0.3, encoder_layerdrop=0.0)}, path)
    probs = checkpoint.load_pretrained_into(m, path)
    assert (m.cfg.dropout, m.cfg.attention_dropout, m.cfg.activation_dropout, m.cfg.dropout_input) == (0.1, 0.05, 0.0, 0.2)
    assert probs["dropout_features"] == 0.3
