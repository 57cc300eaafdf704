"""ORACLE (test infrastructure).  CPU fp32 restatement (plain torch, autograd-capable) of the
reference's linear head, losses and train step:

  model/wav2vec2_linear_nll.py:30-93   BackEnd
  model/wav2vec2_linear_nll.py:120-137 Model._forward
  model/wav2vec2_linear_nll.py:158-192 Model.loss   (== model/loss_metrics.py:498-532 loss_custom)
  model/loss_metrics.py:85-209         sim_metric_seq, supcon_loss
  main.py:47-84                        train_epoch (one iteration), AdamW of main.py:339

Pinned against the reference's own classes (imported with a stub `fairseq`) by
tests/golden/head_loss.npz.
"""
import torch
import torch.nn.functional as F

from . import wav2vec2 as W

HEAD_SHAPES = [
    ("LL.weight", (128, None)), ("LL.bias", (128,)),
    ("backend.m_frame_level.0.weight", (128, 128)), ("backend.m_frame_level.0.bias", (128,)),
    ("backend.m_frame_level.3.weight", (128, 128)), ("backend.m_frame_level.3.bias", (128,)),
    ("backend.m_frame_level.6.weight", (128, 128)), ("backend.m_frame_level.6.bias", (128,)),
    ("backend.m_utt_level.weight", (2, 128)), ("backend.m_utt_level.bias", (2,)),
]


def init_head(embed, seed=1):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in HEAD_SHAPES:
        shape = tuple(embed if s is None else s for s in shape)
        fan_in = shape[-1] if len(shape) > 1 else 128
        sd[name] = (torch.rand(shape, generator=g) * 2 - 1) / (fan_in ** 0.5)
    return sd


def head_forward(sd, x_ssl, dropout_masks=None):
    """x_ssl [B,T,E] -> (log_probs [B,2], feats [B,T,128] (pre-ReLU LL output), emb [B,128]).
    dropout_masks: None (eval) or 3 tensors [B,T,128] of keep/(1-p) factors (train)."""
    feats = F.linear(x_ssl, sd["LL.weight"], sd["LL.bias"])                       # linear_nll:127-128
    h = torch.relu(feats)                                                         # :129
    for j, idx in enumerate((0, 3, 6)):                                           # BackEnd :49-63
        h = F.leaky_relu(F.linear(h, sd["backend.m_frame_level.%d.weight" % idx],
                                  sd["backend.m_frame_level.%d.bias" % idx]), 0.01)
        if dropout_masks is not None:
            h = h * dropout_masks[j]
    emb = h.mean(1)                                                               # :88
    logits = F.linear(emb, sd["backend.m_utt_level.weight"], sd["backend.m_utt_level.bias"])
    return F.log_softmax(logits, dim=1), feats, emb                               # :134


def sim_metric_seq(mat1, mat2):
    """loss_metrics.py:85-86"""
    return torch.bmm(mat1.permute(1, 0, 2), mat2.permute(1, 2, 0)).mean(0)


def supcon_loss(input_feat, labels, t=0.07):
    """loss_metrics.py:87-209 specialised to the call made by Model.loss: n_views == 1,
    contra_mode 'all', sim_metric = sim_metric_seq, no length norm.  input_feat [bs, 1, T', d]."""
    feat = input_feat
    bs = feat.shape[0]
    labels = labels.view(-1, 1)
    mask = torch.eq(labels, labels.T).type(feat.dtype)
    contrast = torch.cat(torch.unbind(feat, dim=1), dim=0)
    logits_mat = torch.div(sim_metric_seq(contrast, contrast), t)
    self_mask = torch.scatter(torch.ones_like(mask), 1, torch.arange(bs).view(-1, 1), 0)
    mask_ = mask * self_mask
    logits_max, _ = torch.max(logits_mat * self_mask, dim=1, keepdim=True)
    logits_mat_ = logits_mat - logits_max.detach()
    exp_logits = torch.exp(logits_mat_ * self_mask) * self_mask
    log_prob = logits_mat_ - torch.log(exp_logits.sum(1, keepdim=True))
    mean_log_prob_pos = (mask_ * log_prob).sum(1) / mask_.sum(1)
    return (-mean_log_prob_pos).view(1, bs).mean()


def model_loss(output, feats, emb, labels, loss_type=1):
    """Model.loss, linear_nll:158-192."""
    bz = output.shape[0]
    L_CE = 1 / bz * F.cross_entropy(output, labels)   # CE applied to log-probs, then /bz again (sic)
    L_CF1 = 1 / bz * supcon_loss(feats.unsqueeze(1), labels)
    L_CF2 = 1 / bz * supcon_loss(emb.unsqueeze(1).unsqueeze(-1), labels)
    return {1: {"L_CE": L_CE, "L_CF1": L_CF1, "L_CF2": L_CF2}, 2: {"L_CE": L_CE, "L_CF1": L_CF1},
            3: {"L_CE": L_CE, "L_CF2": L_CF2}, 4: {"L_CE": L_CE}, 5: {"L_CF1": L_CF1, "L_CF2": L_CF2}}[loss_type]


def full_forward(ssl_sd, head_sd, cfg, x, dropout_masks=None, enc_masks=None):
    return head_forward(head_sd, W.forward(ssl_sd, cfg, x, masks=enc_masks), dropout_masks)


def train_step(ssl_sd, head_sd, cfg, x, labels, loss_type=1, lr=1e-5, wd=1e-4, dropout_masks=None, opt_state=None, enc_masks=None):
    """One iteration of main.py:53-80 on CPU with torch autograd + torch.optim.AdamW (main.py:339).
    Mutates the parameter tensors in place; returns (losses dict of floats, grads dict, outputs)."""
    params = {}
    for k, v in ssl_sd.items():
        params["ssl_model.model." + k] = v
    params.update(head_sd)
    names = [n for n, _, tr in W.param_shapes(cfg) if tr]
    train = {("ssl_model.model." + n): params["ssl_model.model." + n] for n in names}
    train.update(head_sd)
    for p in train.values():
        p.requires_grad_(True)
        p.grad = None
    out, feats, emb = full_forward(ssl_sd, head_sd, cfg, x, dropout_masks, enc_masks)
    losses = model_loss(out, feats, emb, labels, loss_type)
    total = sum(losses.values())
    total.backward()
    grads = {k: p.grad.detach().clone() for k, p in train.items()}
    opt = opt_state if opt_state is not None else torch.optim.AdamW(list(train.values()), lr=lr, weight_decay=wd)
    opt.step()
    for p in train.values():
        p.requires_grad_(False)
    return ({k: float(v.detach()) for k, v in losses.items()}, grads,
            (out.detach(), feats.detach(), emb.detach()), opt)
