"""ORACLE (test infrastructure).  CPU fp32 restatement, in plain torch ops, of the forward the
reference obtains from fairseq (third-party, pinned at a54021305d6b3c4c5959ac9395135f63202db8f1 by
00_envsetup.sh:29, NOT under /root/reference):

    model/xlsr.py:41   self.model(input_tmp, mask=False, features_only=True)['x']

i.e. fairseq `Wav2Vec2Model.forward` with layer_norm_first=True, extractor_mode="layer_norm",
conv_bias=True (the XLS-R-300M recipe).  PARITY UNPINNED w.r.t. fairseq
itself (cannot be installed here); pinned instead against `transformers.Wav2Vec2Model` with copied
weights (oracle/gen_golden.py, tests/test_oracle_golden.py).

State-dict keys use fairseq's names so that the reference's checkpoints load unchanged
(SURVEY.md §8b "Checkpoint compatibility").
"""
import math

import torch
import torch.nn.functional as F


class W2VConfig:
    def __init__(self, conv_dim=512, conv_kernels=(10, 3, 3, 3, 3, 2, 2), conv_strides=(5, 2, 2, 2, 2, 2, 2),
                 embed=1024, layers=24, heads=16, ffn=4096, pos_k=128, pos_groups=16, final_dim=768,
                 latent_vars=320, latent_groups=2):
        self.conv_dim, self.conv_kernels, self.conv_strides = conv_dim, tuple(conv_kernels), tuple(conv_strides)
        self.embed, self.layers, self.heads, self.ffn = embed, layers, heads, ffn
        self.pos_k, self.pos_groups = pos_k, pos_groups
        self.final_dim, self.latent_vars, self.latent_groups = final_dim, latent_vars, latent_groups

    @staticmethod
    def tiny():
        return W2VConfig(conv_dim=32, embed=64, layers=2, heads=4, ffn=128, pos_k=16, pos_groups=4, final_dim=16,
                         latent_vars=8, latent_groups=2)

    def out_len(self, L):
        for k, s in zip(self.conv_kernels, self.conv_strides):
            L = (L - k) // s + 1
        return L


def param_shapes(cfg):
    """Ordered (name, shape, trainable_on_this_path) in fairseq naming."""
    C, E = cfg.conv_dim, cfg.embed
    out = [("mask_emb", (E,), False)]
    cin = 1
    for i, k in enumerate(cfg.conv_kernels):
        out += [("feature_extractor.conv_layers.%d.0.weight" % i, (C, cin, k), True),
                ("feature_extractor.conv_layers.%d.0.bias" % i, (C,), True),
                ("feature_extractor.conv_layers.%d.2.1.weight" % i, (C,), True),
                ("feature_extractor.conv_layers.%d.2.1.bias" % i, (C,), True)]
        cin = C
    out += [("post_extract_proj.weight", (E, C), True), ("post_extract_proj.bias", (E,), True)]
    vd = cfg.final_dim // cfg.latent_groups
    out += [("quantizer.vars", (1, cfg.latent_vars * cfg.latent_groups, vd), False),
            ("quantizer.weight_proj.weight", (cfg.latent_vars * cfg.latent_groups, C), False),
            ("quantizer.weight_proj.bias", (cfg.latent_vars * cfg.latent_groups,), False),
            ("project_q.weight", (cfg.final_dim, cfg.final_dim), False), ("project_q.bias", (cfg.final_dim,), False)]
    out += [("encoder.pos_conv.0.bias", (E,), True),
            ("encoder.pos_conv.0.weight_g", (1, 1, cfg.pos_k), True),
            ("encoder.pos_conv.0.weight_v", (E, E // cfg.pos_groups, cfg.pos_k), True)]
    for n in range(cfg.layers):
        p = "encoder.layers.%d." % n
        for proj in ("k_proj", "v_proj", "q_proj", "out_proj"):
            out += [(p + "self_attn.%s.weight" % proj, (E, E), True), (p + "self_attn.%s.bias" % proj, (E,), True)]
        out += [(p + "self_attn_layer_norm.weight", (E,), True), (p + "self_attn_layer_norm.bias", (E,), True),
                (p + "fc1.weight", (cfg.ffn, E), True), (p + "fc1.bias", (cfg.ffn,), True),
                (p + "fc2.weight", (E, cfg.ffn), True), (p + "fc2.bias", (E,), True),
                (p + "final_layer_norm.weight", (E,), True), (p + "final_layer_norm.bias", (E,), True)]
    out += [("encoder.layer_norm.weight", (E,), True), ("encoder.layer_norm.bias", (E,), True),
            ("layer_norm.weight", (C,), True), ("layer_norm.bias", (C,), True),
            ("final_proj.weight", (cfg.final_dim, E), False), ("final_proj.bias", (cfg.final_dim,), False)]
    return out


def init_state(cfg, seed=0, dtype=torch.float32):
    """Seeded random weights of the right shapes (no checkpoint is available).  Scales are chosen
    so activations stay O(1) through the stack: conv weights kaiming-normal (as fairseq), linears
    N(0, 0.02) like BERT init, LayerNorm affine near identity with small noise so that parity
    tests see non-trivial gamma/beta."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape, _ in param_shapes(cfg):
        if name.endswith("2.1.weight") or "layer_norm.weight" in name:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("2.1.bias") or "layer_norm.bias" in name:
            t = 0.05 * torch.randn(shape, generator=g)
        elif "conv_layers" in name and name.endswith("0.weight"):
            fan_in = shape[1] * shape[2]
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
        elif name == "encoder.pos_conv.0.weight_v":
            std = math.sqrt(4.0 / (cfg.pos_k * cfg.embed))
            t = torch.randn(shape, generator=g) * std
        elif name == "encoder.pos_conv.0.weight_g":
            t = None  # set below to ||v|| as torch weight_norm does at wrap time
        elif name.endswith(".bias"):
            t = 0.02 * torch.randn(shape, generator=g)
        else:
            t = 0.02 * torch.randn(shape, generator=g)
        sd[name] = t
    v = sd["encoder.pos_conv.0.weight_v"]
    sd["encoder.pos_conv.0.weight_g"] = v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt() * (
        1.0 + 0.1 * torch.randn((1, 1, cfg.pos_k), generator=g))
    return {k: t.to(dtype) for k, t in sd.items()}


def pos_conv_weight(sd):
    """torch.nn.utils.weight_norm(conv, name="weight", dim=2): w = g * v / ||v||, the norm taken
    over dims (0, 1) for every kernel tap."""
    v, g = sd["encoder.pos_conv.0.weight_v"], sd["encoder.pos_conv.0.weight_g"]
    return g * v / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()


def conv_stack(sd, cfg, x):
    """Appendix A step 1: 7 x (Conv1d(bias) -> LayerNorm over channels (fp32, eps 1e-5) -> GELU(erf)).
    x: [B, L] -> list of per-layer outputs in channels-last layout [B, T_i, C]."""
    h = x[:, None, :]
    outs = []
    for i, s in enumerate(cfg.conv_strides):
        p = "feature_extractor.conv_layers.%d." % i
        h = F.conv1d(h, sd[p + "0.weight"], sd[p + "0.bias"], stride=s)
        h = F.layer_norm(h.transpose(1, 2), (cfg.conv_dim,), sd[p + "2.1.weight"], sd[p + "2.1.bias"], 1e-5)
        h = F.gelu(h)
        outs.append(h)
        h = h.transpose(1, 2)
    return outs


def encoder_layer(sd, cfg, n, x, return_attn=False, masks=None):
    """fairseq TransformerSentenceEncoderLayer.forward, layer_norm_first=True.  masks (optional, train mode): dict of explicit
    keep / (1 - p) factors for this layer — "attn" [B, H, T, T] on the attention probabilities (MultiheadAttention.dropout_module),
    "d1" [B, T, E] (dropout1, on out_proj's output), "d2" [B, T, ffn] (dropout2, after the activation), "d3" [B, T, E] (dropout3, on
    fc2's output); a missing key = no dropout at that site."""
    masks = masks or {}
    p = "encoder.layers.%d." % n
    B, T, E = x.shape
    H = cfg.heads
    D = E // H
    res = x
    h = F.layer_norm(x, (E,), sd[p + "self_attn_layer_norm.weight"], sd[p + "self_attn_layer_norm.bias"], 1e-5)
    q = F.linear(h, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"]) * (D ** -0.5)
    k = F.linear(h, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"])
    v = F.linear(h, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"])
    q = q.view(B, T, H, D).transpose(1, 2)
    k = k.view(B, T, H, D).transpose(1, 2)
    v = v.view(B, T, H, D).transpose(1, 2)
    attn = torch.softmax(q @ k.transpose(-1, -2), dim=-1)
    pa = attn * masks["attn"] if "attn" in masks else attn
    ctx = (pa @ v).transpose(1, 2).reshape(B, T, E)
    o = F.linear(ctx, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
    x = res + (o * masks["d1"] if "d1" in masks else o)
    res = x
    h = F.layer_norm(x, (E,), sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"], 1e-5)
    h = F.gelu(F.linear(h, sd[p + "fc1.weight"], sd[p + "fc1.bias"]))
    if "d2" in masks:
        h = h * masks["d2"]
    o = F.linear(h, sd[p + "fc2.weight"], sd[p + "fc2.bias"])
    x = res + (o * masks["d3"] if "d3" in masks else o)
    return (x, attn) if return_attn else x


def forward(sd, cfg, x, return_all=False, masks=None):
    """x: [B, L] fp32 raw waveform (NOT normalised — the reference feeds it as is, Appendix A).
    Returns [B, T, embed]; with return_all also a dict of intermediates.  masks (optional): explicit element-dropout factors as
    fairseq's train mode would draw them — "in" [B, T, E] (Wav2Vec2Model.dropout_input on the projected features), "enc" [B, T, E]
    (TransformerEncoder: F.dropout after the positional-conv residual add) and per layer n a dict under key n (see encoder_layer)."""
    masks = masks or {}
    inter = {}
    feats = conv_stack(sd, cfg, x)
    inter["conv"] = feats
    h = F.layer_norm(feats[-1], (cfg.conv_dim,), sd["layer_norm.weight"], sd["layer_norm.bias"], 1e-5)
    h = F.linear(h, sd["post_extract_proj.weight"], sd["post_extract_proj.bias"])
    if "in" in masks:
        h = h * masks["in"]
    inter["proj"] = h
    # positional conv: Conv1d(E, E, k, padding=k//2, groups) then drop the last frame (even k), GELU
    w = pos_conv_weight(sd)
    pc = F.conv1d(h.transpose(1, 2), w, sd["encoder.pos_conv.0.bias"], padding=cfg.pos_k // 2, groups=cfg.pos_groups)
    if cfg.pos_k % 2 == 0:
        pc = pc[:, :, :-1]
    h = h + F.gelu(pc).transpose(1, 2)
    if "enc" in masks:
        h = h * masks["enc"]
    inter["pos"] = h
    # (fairseq pads T to a multiple of 2 with a key-padding mask here: a mathematical no-op)
    layers = []
    for n in range(cfg.layers):
        h = encoder_layer(sd, cfg, n, h, masks=masks.get(n))
        layers.append(h)
    inter["layers"] = layers
    h = F.layer_norm(h, (cfg.embed,), sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"], 1e-5)
    return (h, inter) if return_all else h


# ---- key maps to transformers.Wav2Vec2Model (used only by gen_golden.py / tests) ----------------
def to_hf_state(sd, cfg):
    m = {}
    for i in range(len(cfg.conv_kernels)):
        m["feature_extractor.conv_layers.%d.conv.weight" % i] = sd["feature_extractor.conv_layers.%d.0.weight" % i]
        m["feature_extractor.conv_layers.%d.conv.bias" % i] = sd["feature_extractor.conv_layers.%d.0.bias" % i]
        m["feature_extractor.conv_layers.%d.layer_norm.weight" % i] = sd["feature_extractor.conv_layers.%d.2.1.weight" % i]
        m["feature_extractor.conv_layers.%d.layer_norm.bias" % i] = sd["feature_extractor.conv_layers.%d.2.1.bias" % i]
    m["feature_projection.layer_norm.weight"] = sd["layer_norm.weight"]
    m["feature_projection.layer_norm.bias"] = sd["layer_norm.bias"]
    m["feature_projection.projection.weight"] = sd["post_extract_proj.weight"]
    m["feature_projection.projection.bias"] = sd["post_extract_proj.bias"]
    m["encoder.pos_conv_embed.conv.bias"] = sd["encoder.pos_conv.0.bias"]
    m["encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = sd["encoder.pos_conv.0.weight_g"]
    m["encoder.pos_conv_embed.conv.parametrizations.weight.original1"] = sd["encoder.pos_conv.0.weight_v"]
    for n in range(cfg.layers):
        a, b = "encoder.layers.%d." % n, "encoder.layers.%d." % n
        for proj in ("q_proj", "k_proj", "v_proj", "out_proj"):
            for wb in ("weight", "bias"):
                m[b + "attention.%s.%s" % (proj, wb)] = sd[a + "self_attn.%s.%s" % (proj, wb)]
        for wb in ("weight", "bias"):
            m[b + "layer_norm." + wb] = sd[a + "self_attn_layer_norm." + wb]
            m[b + "feed_forward.intermediate_dense." + wb] = sd[a + "fc1." + wb]
            m[b + "feed_forward.output_dense." + wb] = sd[a + "fc2." + wb]
            m[b + "final_layer_norm." + wb] = sd[a + "final_layer_norm." + wb]
    m["encoder.layer_norm.weight"] = sd["encoder.layer_norm.weight"]
    m["encoder.layer_norm.bias"] = sd["encoder.layer_norm.bias"]
    m["masked_spec_embed"] = sd["mask_emb"]
    return m
