"""ORACLE (test infrastructure; never imported by the product path).  CPU restatement of the reference's Conformer block as ONE functional
forward over a state dict — model/conformer.py:180-216 (ConformerBlock.forward: x = 0.5 ff1(x) + x; x = attn(x) + x; x = conv(x) + x;
x = 0.5 ff2(x) + x; post_norm), with :128-145 (FeedForward: Linear, Swish, Linear under a pre-LayerNorm), :68-126 (Attention: bias-free
q / kv projections, Shaw relative positions `dist = clamp(i - j, +-max_pos) + max_pos`, scores (q.k + q.rel_pos_emb[dist]) * scale, the
mask as masked_fill(-finfo.max), soft-max, to_out), :147-178 (ConformerConvModule: LayerNorm, 1x1 conv to 2 x inner, GLU over channels,
depthwise conv with the "same" padding of calc_same_padding :17-19 or the causal (k - 1, 0), BatchNorm1d unless causal, Swish, 1x1 conv).
The shape of the block is read off the tensors (heads and max_pos are the two numbers a state dict does not hold).  Runs in the dtype of
the tensors it is given: float32 to be compared with the golden vectors of the reference itself (tests/golden/conformer.npz, written
by oracle/gen_golden.py::gen_conformer, which imports /root/reference/model/conformer.py), float64 as the yardstick of the GPU tests.
Autograd supplies the gradients.  Dropout is not restated (the reference's block defaults are all 0).
"""
import torch
import torch.nn.functional as F


def _ln(x, t, name):
    return F.layer_norm(x, x.shape[-1:], t[name + ".weight"], t[name + ".bias"], 1e-5)


def _swish(x):
    return x * torch.sigmoid(x)


def _ff(x, t, pre):
    h = _ln(x, t, pre + ".fn.norm")
    h = F.linear(h, t[pre + ".fn.fn.net.0.weight"], t[pre + ".fn.fn.net.0.bias"])
    return F.linear(_swish(h), t[pre + ".fn.fn.net.3.weight"], t[pre + ".fn.fn.net.3.bias"])


def _attention(x, t, heads, mask):
    B, n, _ = x.shape
    h = _ln(x, t, "attn.norm")
    q = F.linear(h, t["attn.fn.to_q.weight"])
    kv = F.linear(h, t["attn.fn.to_kv.weight"])
    hd = q.shape[-1]
    dh = hd // heads
    split = lambda z: z.reshape(B, n, heads, dh).permute(0, 2, 1, 3)         # noqa: E731   [B, H, n, dh]
    q, k, v = split(q), split(kv[..., :hd]), split(kv[..., hd:])
    scale = dh ** -0.5
    E = t["attn.fn.rel_pos_emb.weight"]
    max_pos = (E.shape[0] - 1) // 2
    pos = torch.arange(n)
    dist = (pos[:, None] - pos[None, :]).clamp(-max_pos, max_pos) + max_pos  # [i][j]
    rel = E[dist]                                                            # [n, n, dh]
    dots = torch.matmul(q, k.transpose(-1, -2)) * scale
    dots = dots + torch.matmul(q.unsqueeze(-2), rel.transpose(-1, -2)).squeeze(-2) * scale      # sum_d q[b,h,i,d] rel[i,j,d]
    if mask is not None:
        pair = mask[:, None, :, None] & mask[:, None, None, :]
        dots = dots.masked_fill(~pair, -torch.finfo(dots.dtype).max)
    p = torch.softmax(dots, dim=-1)
    o = torch.matmul(p, v).permute(0, 2, 1, 3).reshape(B, n, hd)
    return F.linear(o, t["attn.fn.to_out.weight"], t["attn.fn.to_out.bias"])


def _conv_module(x, t, training, buffers):
    h = _ln(x, t, "conv.net.0").transpose(1, 2)                               # [B, dim, n]
    h = F.conv1d(h, t["conv.net.2.weight"], t["conv.net.2.bias"])
    inner = h.shape[1] // 2
    h = h[:, :inner] * torch.sigmoid(h[:, inner:])
    w = t["conv.net.4.conv.weight"]
    k = w.shape[-1]
    causal = "conv.net.5.weight" not in t
    pad = (k - 1, 0) if causal else (k // 2, k // 2 - (k + 1) % 2)
    h = F.conv1d(F.pad(h, pad), w, t["conv.net.4.conv.bias"], groups=inner)
    if not causal:
        rm, rv = buffers["conv.net.5.running_mean"], buffers["conv.net.5.running_var"]
        h = F.batch_norm(h, rm, rv, t["conv.net.5.weight"], t["conv.net.5.bias"], training, 0.1, 1e-5)
        if training:
            buffers["conv.net.5.num_batches_tracked"] += 1
    h = F.conv1d(_swish(h), t["conv.net.7.weight"], t["conv.net.7.bias"])
    return h.transpose(1, 2)


def forward(t, x, heads, training, mask=None, buffers=None):
    """t: {reference state-dict name: tensor} (parameters; the BatchNorm buffers may live in `buffers` instead — they are updated in place
    when training).  x [B, n, dim]; mask: optional bool [B, n]."""
    buffers = t if buffers is None else buffers
    x = 0.5 * _ff(x, t, "ff1") + x
    x = _attention(x, t, heads, mask) + x
    x = _conv_module(x, t, training, buffers) + x
    x = 0.5 * _ff(x, t, "ff2") + x
    return _ln(x, t, "post_norm")
