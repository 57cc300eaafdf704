"""ORACLE (test infrastructure).  CPU restatement (plain torch over a state dict, autograd-capable, any float dtype) of the
reference's `wav2vec2_btse` plugin behind the SSL encoder — BASELINE.json configs[4]:

  model/wav2vec2_btse/backend.py:17-47      Model: LL 1024 -> 128, MLP back-end
  model/wav2vec2_btse/linear.py:5-67        MLP: 3 x (Linear 128 -> 128, LeakyReLU(0.01), Dropout 0.5), mean over T, m_utt_level (its logits
                                            are computed and DISCARDED by model.py:324)
  model/wav2vec2_btse/model.py:210-238      bioEncoderTransformersmall: Embedding * sqrt(bio_dim), transformer.Encoder, 1x1 conv, LAST position
  model/wav2vec2_btse/transformer.py:17-52  Encoder (post-LN: x = LN(x + attn), x = LN(x + ffn))
  model/wav2vec2_btse/transformer.py:105-260 MultiHeadAttention with window_size = 4 relative keys AND values, heads_share = True
  model/wav2vec2_btse/transformer.py:261-306 FFN (kernel 1, ReLU)
  model/wav2vec2_btse/modules.py:27-39      LayerNorm over channels (gamma / beta, eps 1e-5)
  model/wav2vec2_btse/model.py:321-343      Model.forward(x, bio, bio_lengths): concat (or fc1 + add) -> fc2 -> log_softmax

The reference's pad / reshape skewing (`_relative_position_to_absolute_position`, `_absolute_position_to_relative_position`,
`_get_relative_embeddings`, transformer.py:189-243) is restated as what it computes:
    scores[i, j] += (q_i / sqrt(dk)) . emb_rel_k[j - i + w]          for |j - i| <= w
    out[i]       += sum_{|j - i| <= w} p[i, j] * emb_rel_v[j - i + w]
Pinned to the reference's own classes (imported with a stand-in for the absent `biosegment` module, the SSL encoder injected) by
tests/golden/btse.npz (oracle/gen_golden.py::gen_btse), tests/test_btse_cpu.py.

What has NO oracle: the tokeniser `Wav2bioCNN.wav2bio` (model.py:306-319; the `biosegment` package is a dangling symlink in the
reference) — bio tokens are an INPUT here — and `Model.loss` (model.py:345-375 reads attributes that are never set).
"""
import math

import torch
import torch.nn.functional as F

WINDOW = 4      # transformer.py:18 (Encoder's default window_size, never overridden by model.py:220-225)
LN_EPS = 1e-5   # modules.py:28
MASK_FILL = -1e4  # transformer.py:168


def default_args(**over):
    """configs/conf-5-btse-trans64.yaml `model:` block."""
    a = dict(flag_fix_ssl=False, contra_mode="all", loss_type=1, bio_out=64, nb_classes=2, is_add=False, bio_hid=256, n_heads=4,
             pf_dim=128, n_layers=3, n_bios=3, bio_dim=32)
    a.update(over)
    return a


def state_shapes(args, embed):
    """{state-dict key: shape} of reference Model(args) without `backend.ssl_model.*` (registration order of the reference)."""
    D, Fd, NL, dk = args["bio_dim"], args["pf_dim"], args["n_layers"], args["bio_dim"] // args["n_heads"]
    s = {"backend.LL.weight": (128, embed), "backend.LL.bias": (128,)}
    for i in range(3):
        s["backend.mlp.m_frame_level.linear_%d.weight" % i] = (128, 128)
        s["backend.mlp.m_frame_level.linear_%d.bias" % i] = (128,)
    s["backend.mlp.m_utt_level.weight"] = (2, 128)
    s["backend.mlp.m_utt_level.bias"] = (2,)
    s["bioScoring.bio_embedding.weight"] = (args["n_bios"], D)
    p = "bioScoring.encoder."
    for i in range(NL):
        s[p + "attn_layers.%d.emb_rel_k" % i] = (1, 2 * WINDOW + 1, dk)
        s[p + "attn_layers.%d.emb_rel_v" % i] = (1, 2 * WINDOW + 1, dk)
        for c in "qkvo":
            s[p + "attn_layers.%d.conv_%s.weight" % (i, c)] = (D, D, 1)
            s[p + "attn_layers.%d.conv_%s.bias" % (i, c)] = (D,)
    for i in range(NL):
        s[p + "norm_layers_1.%d.gamma" % i] = (D,)
        s[p + "norm_layers_1.%d.beta" % i] = (D,)
    for i in range(NL):
        s[p + "ffn_layers.%d.conv_1.weight" % i] = (Fd, D, 1)
        s[p + "ffn_layers.%d.conv_1.bias" % i] = (Fd,)
        s[p + "ffn_layers.%d.conv_2.weight" % i] = (D, Fd, 1)
        s[p + "ffn_layers.%d.conv_2.bias" % i] = (D,)
    for i in range(NL):
        s[p + "norm_layers_2.%d.gamma" % i] = (D,)
        s[p + "norm_layers_2.%d.beta" % i] = (D,)
    s["bioScoring.bio_scoring.weight"] = (args["bio_out"], D, 1)
    s["bioScoring.bio_scoring.bias"] = (args["bio_out"],)
    if args["is_add"]:
        s["fc1.weight"] = (args["bio_out"], 128)
        s["fc1.bias"] = (args["bio_out"],)
        s["fc2.weight"] = (args["nb_classes"], 128)
    else:
        s["fc2.weight"] = (args["nb_classes"], 128 + args["bio_out"])
    s["fc2.bias"] = (args["nb_classes"],)
    return s


def _rel_band(L, dtype):
    """[L, L] index j - i + w and validity |j - i| <= w."""
    i = torch.arange(L)
    d = i[None, :] - i[:, None]
    return (d + WINDOW).clamp(0, 2 * WINDOW), (d.abs() <= WINDOW).to(dtype)


def attention(sd, pre, x, attn_mask, n_heads):
    """transformer.py:138-186.  x [B, L, D] (channels last here; the reference keeps [B, D, L]); attn_mask [B, L, L] of 0 / 1."""
    B, L, D = x.shape
    dk = D // n_heads
    lin = lambda c: F.linear(x, sd[pre + "conv_%s.weight" % c][:, :, 0], sd[pre + "conv_%s.bias" % c])
    q, k, v = (lin(c).view(B, L, n_heads, dk).transpose(1, 2) for c in "qkv")          # [B, h, L, dk]  (:150-153)
    qs = q / math.sqrt(dk)
    scores = qs @ k.transpose(-2, -1)                                                   # :155
    idx, band = _rel_band(L, x.dtype)
    Ek = sd[pre + "emb_rel_k"][0]                                                       # [2w+1, dk], shared by the heads
    Ev = sd[pre + "emb_rel_v"][0]
    rel = qs @ Ek.t()                                                                   # [B, h, L, 2w+1]   (:158-160)
    scores = scores + torch.gather(rel, 3, idx.expand(B, n_heads, L, L)) * band        # :161
    scores = scores.masked_fill(attn_mask[:, None] == 0, MASK_FILL)                     # :168
    p = torch.softmax(scores, dim=-1)                                                   # :173  (dropout p = 0: transformer.py:17)
    out = p @ v                                                                         # :175
    pb = p * band                                                                       # :177-179: only |j - i| <= w meets a non-zero row
    relw = torch.zeros(B, n_heads, L, 2 * WINDOW + 1, dtype=x.dtype).scatter_add(3, idx.expand(B, n_heads, L, L), pb)
    out = out + relw @ Ev
    out = out.transpose(1, 2).reshape(B, L, D)                                          # :180
    return F.linear(out, sd[pre + "conv_o.weight"][:, :, 0], sd[pre + "conv_o.bias"])   # :146


def bio_encoder(sd, args, bio, bio_lengths, return_x=False):
    """model.py:227-238 -> [B, bio_out]: the scoring conv's output at the LAST (padded) position, times its mask
    (return_x: also the encoder output x * mask [B, L, D], transformer.py:51)."""
    D, H, NL = args["bio_dim"], args["n_heads"], args["n_layers"]
    B, L = bio.shape
    e = sd["bioScoring.bio_embedding.weight"][bio.long()] * math.sqrt(D)                # :228
    mask = (torch.arange(L)[None, :] < bio_lengths[:, None].long()).to(e.dtype)         # :230 commons.sequence_mask
    m3 = mask[:, :, None]
    x = e * m3                                                                          # :232 + transformer.py:42
    attn_mask = mask[:, :, None] * mask[:, None, :]                                     # transformer.py:41
    p = "bioScoring.encoder."
    for i in range(NL):
        y = attention(sd, p + "attn_layers.%d." % i, x, attn_mask, H)
        x = F.layer_norm(x + y, (D,), sd[p + "norm_layers_1.%d.gamma" % i], sd[p + "norm_layers_1.%d.beta" % i], LN_EPS)
        f = p + "ffn_layers.%d." % i
        h = torch.relu(F.linear(x * m3, sd[f + "conv_1.weight"][:, :, 0], sd[f + "conv_1.bias"]))       # transformer.py:283-288
        y = F.linear(h * m3, sd[f + "conv_2.weight"][:, :, 0], sd[f + "conv_2.bias"]) * m3              # :290-291
        x = F.layer_norm(x + y, (D,), sd[p + "norm_layers_2.%d.gamma" % i], sd[p + "norm_layers_2.%d.beta" % i], LN_EPS)
    x = x * m3                                                                          # transformer.py:51
    s = F.linear(x, sd["bioScoring.bio_scoring.weight"][:, :, 0], sd["bioScoring.bio_scoring.bias"]) * m3   # model.py:234
    return (s[:, -1, :], x) if return_x else s[:, -1, :]                                # :236


def forward(sd, args, x_ssl, bio, bio_lengths, dropout_masks=None):
    """x_ssl [B, T, E] (the SSL encoder's output) -> (log_probs [B, nb_classes], feats [B, T, 128], b [B, 128 + bio_out] or [B, bio_out]).
    dropout_masks: None (eval) or three [B, T, 128] tensors of keep / (1 - p) factors (train)."""
    feats = F.linear(x_ssl, sd["backend.LL.weight"], sd["backend.LL.bias"])             # backend.py:41
    h = feats
    for i in range(3):                                                                  # linear.py:27-36,58
        h = F.leaky_relu(F.linear(h, sd["backend.mlp.m_frame_level.linear_%d.weight" % i], sd["backend.mlp.m_frame_level.linear_%d.bias" % i]), 0.01)
        if dropout_masks is not None:
            h = h * dropout_masks[i]
    emb = h.mean(1)                                                                     # linear.py:62
    s = bio_encoder(sd, args, bio, bio_lengths)                                         # model.py:328
    if args["is_add"]:
        b = F.linear(emb, sd["fc1.weight"], sd["fc1.bias"]) + s                         # :330-331
    else:
        b = torch.cat((emb, s), 1)                                                      # :333
    logp = torch.log_softmax(F.linear(b, sd["fc2.weight"], sd["fc2.bias"]), dim=1)      # :336-338
    return logp, feats, b
