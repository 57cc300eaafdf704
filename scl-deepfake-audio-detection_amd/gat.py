"""Autograd wrapper of the fused AASIST pairwise attention score (csrc/gat.hip).

    gat_score(x [B,N,D] f32, att_proj.weight [Do,D], att_proj.bias [Do], a [3,Do] or [1,Do], n1) -> s [B,N,N] f32

replaces `tanh(att_proj(x_i * x_j)) @ att_weight` of GraphAttentionLayer / HtrgGraphAttentionLayer
(model/wav2vec2_aasist.py:107-135, 259-291); the temperature and the softmax stay torch ops on the [B,N,N] result.
"""
import torch

from . import ops


class _GatScoreFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, bias, a, n1):
        B, N, D = x.shape
        Do = W.shape[0]
        xc, Wc, bc = x.contiguous().float(), W.contiguous().float(), bias.contiguous().float()
        a3 = torch.zeros(3, Do, device=x.device, dtype=torch.float32)
        a3[: a.shape[0]] = a
        s = torch.empty(B, N, N, device=x.device, dtype=torch.float32)
        ops.gat_score_fwd(xc, Wc, bc, a3, s, B, N, D, Do, n1)
        ctx.save_for_backward(xc, Wc, bc, a3)
        ctx.n1, ctx.na = n1, a.shape[0]
        return s

    @staticmethod
    def backward(ctx, ds):
        x, W, bias, a3 = ctx.saved_tensors
        B, N, D = x.shape
        Do = W.shape[0]
        nb = ops.gat_score_nblocks(N) * B
        ncol = Do * D + 4 * Do
        dP = torch.empty(B * N * N * D, device=x.device, dtype=torch.float32)
        part = torch.empty(nb, ncol, device=x.device, dtype=torch.float32)
        dx = torch.empty_like(x)
        ops.gat_score_bwd(x, W, bias, a3, ds.contiguous().float(), dP, part, dx, B, N, D, Do, ctx.n1)
        red = torch.empty(ncol, device=x.device, dtype=torch.float32)
        ops.colreduce(part, red, nb, ncol)
        dW = red[: Do * D].view(Do, D)
        db = red[Do * D: Do * D + Do]
        da = red[Do * D + Do:].view(3, Do)[: ctx.na]
        return dx, dW, db, da, None


def gat_score(x, W, bias, a, n1=None):
    """a: [1, Do] (homogeneous layer) or [3, Do] = (a11, a22, a12) with n1 = number of first-type nodes."""
    return _GatScoreFn.apply(x, W, bias, a, x.shape[1] if n1 is None else n1)
