"""Conformer block on the HIP kernels — the configs[4] slice of the hot path (BASELINE.json: "conformer head").

Surface = /root/reference/model/conformer.py: `ConformerBlock(dim=, dim_head=, heads=, ff_mult=, conv_expansion_factor=,
conv_kernel_size=, attn_dropout=, ff_dropout=, conv_dropout=, conv_causal=)` (:180-216) and `Conformer(dim, depth=, ...)` (:219-253),
same constructor arguments, same state-dict keys (`ff1.fn.fn.net.0.weight`, `attn.fn.rel_pos_emb.weight`, `conv.net.4.conv.weight`
...), parameters created by the same torch modules in the same order (one seed gives both the same initial weights), `forward(x,
mask=None)` on [B, n, dim] fp32.

What runs: one autograd node per block.  Its forward and its hand-written backward are sequences of C-ABI calls — LayerNorm
(scl_layernorm_*), every Linear / 1x1 Conv1d / attention contraction on the exact-fp32 MFMA GEMM (scl_gemm_bf16 with f32 operands; Scale(0.5)
and the residual adds are its epilogue), BatchNorm1d (scl_bn_*), and csrc/conformer.hip for Swish, GLU, the depthwise convolution and Shaw's
relative positions.  The reference materialises rel_pos_emb(dist) as an [n, n, d] tensor per call (:98-105); here the 2n - 1 distances that
occur are gathered once, `q Eu^T` is a GEMM and the soft-max kernel reads it skewed.  Dropout (all three default to 0, as in the
reference's constructor) uses the counter hash of the other kernels: masks are a function of (seed, element), recomputed in the backward,
statistically equivalent to nn.Dropout, not bit-equal to torch's generator.  There is no CPU fallback: without libscl_hip.so the first
call raises."""
import torch
from torch import nn

from . import ops
from .hipnn import ACT_NONE, _colsum
from .ops import Op

MAX_N = 1024           # scl_relpos_softmax_*: one wave per probability row
MAX_TAPS = 32          # scl_dwconv1d_*


class _Node(nn.Module):
    """Holds sub-modules under the attribute names the reference's wrappers (Scale / PreNorm / FeedForward / nn.Sequential) give them, so
    that state-dict keys match; never called."""

    def __init__(self, **kids):
        super().__init__()
        for k, v in kids.items():
            self.add_module(k.lstrip("_"), v)


def _same_padding(k):
    pad = k // 2
    return pad, pad - (k + 1) % 2


_CALLS = [0]


def _next_seed(base, site):
    return (int(base) * 2654435761 + site * 40503 + 977) & 0x7FFFFFFF


def _wgrad(dy, x, dw, M, N, K, alpha=1.0, a_ld=None, a_off=0):
    """dw [N, K] = alpha * dy[:, a_off : a_off + N]^T x over M rows (split over the rows when the output is a few tiles, slabs summed in order)."""
    A = Op(dy, a_ld or N, offset=a_off)
    tiles = ((N + 63) // 64) * ((K + 63) // 64)
    sk = max(1, min(32, 256 // tiles, M // 128))
    if sk > 1:
        slabs = torch.empty(sk, N, K, device=dy.device)
        ops.gemm(A, Op(x, K), slabs, N, K, M, a_t=True, b_t=True, splitk=sk, c_split_stride=N * K, alpha=alpha)
        ops.reduce_slabs(slabs, dw, N * K, sk, N * K)
    else:
        ops.gemm(A, Op(x, K), dw, N, K, M, a_t=True, b_t=True, alpha=alpha)


class _BlockFn(torch.autograd.Function):
    """x [B, n, dim] -> ConformerBlock.forward (conformer.py:209-215).  Inputs after `blk`: the block's parameters in `blk._plist` order."""

    @staticmethod
    def forward(ctx, x, mask, blk, *params):
        P = dict(zip(blk._pnames, [p.detach() for p in params]))
        training = blk.training
        B, n, D = x.shape
        M = B * n
        dev = x.device
        H, dh, HD, Fd, Ci, k = blk.heads, blk.dim_head, blk.heads * blk.dim_head, blk.dim * blk.ff_mult, blk.inner, blk.kernel_size
        x0 = x.detach().contiguous().float().view(M, D)
        f = lambda *s: torch.empty(*s, device=dev)          # noqa: E731
        sv = {}
        _CALLS[0] += 1
        base = _next_seed(blk.seed, _CALLS[0])
        sv["seeds"] = [_next_seed(base, s) for s in range(8)]

        def ln(xin, name, key):
            h, mean, rstd = f(M, D), f(M), f(M)
            ops.layernorm_fwd(xin, P[name + ".weight"], P[name + ".bias"], None, h, mean, rstd, M, D, eps=1e-5)
            sv[key] = (xin, h, mean, rstd)
            return h

        def out_proj(a, K, wname, bname, xin, scale, p, site):
            """xin + scale * dropout(a W^T + b)."""
            w, b = P[wname].view(D, K), P[bname]
            xo = f(M, D)
            if p > 0 and training:
                y = f(M, D)
                ops.gemm(Op(a, K), Op(w, K), y, M, D, K, bias=b)
                ops.dropout(y, y, None, M * D, sv["seeds"][site], p)
                ops.axpby(xin, y, 1.0, scale, xo, M * D)
            elif scale == 1.0:
                ops.gemm(Op(a, K), Op(w, K), xo, M, D, K, bias=b, R=xin, rmode=1)
            else:
                hb = f(D)
                ops.axpby(b, None, scale, 0.0, hb, D)
                ops.gemm(Op(a, K), Op(w, K), xo, M, D, K, bias=hb, alpha=scale, R=xin, rmode=1)
            return xo

        def ff(xin, pre, key, site):
            h = ln(xin, pre + ".fn.norm", key + "ln")
            u, a = f(M, Fd), f(M, Fd)
            ops.gemm(Op(h, D), Op(P[pre + ".fn.fn.net.0.weight"], D), u, M, Fd, D, bias=P[pre + ".fn.fn.net.0.bias"])
            ops.swish_fwd(u, a, M * Fd)
            if blk.ff_dropout > 0 and training:
                ops.dropout(a, a, None, M * Fd, sv["seeds"][site], blk.ff_dropout)
            sv[key] = (u, a)
            return out_proj(a, Fd, pre + ".fn.fn.net.3.weight", pre + ".fn.fn.net.3.bias", xin, 0.5, blk.ff_dropout, site + 1)

        x1 = ff(x0, "ff1", "ff1", 0)

        # ---- attention (conformer.py:88-120) ----
        h = ln(x1, "attn.norm", "aln")
        np_, Nr = (n + 3) // 4 * 4, (2 * n - 1 + 3) // 4 * 4
        qkv = f(M, 3 * HD)
        ops.gemm(Op(h, D), Op(P["attn.fn.to_q.weight"], D), qkv, M, HD, D, ldc=3 * HD)
        ops.gemm(Op(h, D), Op(P["attn.fn.to_kv.weight"], D), qkv, M, 2 * HD, D, ldc=3 * HD, c_offset=HD)
        qop = Op(qkv, 3 * HD, bs1=n * 3 * HD, bs2=dh)
        kop = Op(qkv, 3 * HD, bs1=n * 3 * HD, bs2=dh, offset=HD)
        vop = Op(qkv, 3 * HD, bs1=n * 3 * HD, bs2=dh, offset=2 * HD)
        S, Rl, Pm, Eu = f(B * H * n, np_), f(B * H * n, Nr), f(B * H * n, np_), f(Nr, dh)
        bq = dict(nb1=B, nb2=H)
        ops.gemm(qop, kop, S, n, n, dh, ldc=np_, c_bs1=H * n * np_, c_bs2=n * np_, **bq)
        ops.relpos_gather(P["attn.fn.rel_pos_emb.weight"], Eu, n, Nr, dh, blk.max_pos_emb)
        ops.gemm(qop, Op(Eu, dh), Rl, n, Nr, dh, ldc=Nr, c_bs1=H * n * Nr, c_bs2=n * Nr, **bq)
        mk = None if mask is None else mask.to(device=dev, dtype=torch.bool).contiguous().view(B, n).to(torch.uint8)
        ops.relpos_softmax_fwd(S, Rl, mk, Pm, B, H, n, np_, Nr, np_, blk.scale)
        cx = f(M, HD)
        ops.gemm(Op(Pm, np_, bs1=H * n * np_, bs2=n * np_), vop, cx, n, dh, n, b_t=True, ldc=HD, c_bs1=n * HD, c_bs2=dh, **bq)
        x2 = out_proj(cx, HD, "attn.fn.to_out.weight", "attn.fn.to_out.bias", x1, 1.0, blk.attn_dropout, 4)
        sv["attn"] = (qkv, Pm, Eu, cx, mk, S, Rl)      # S / Rl are reused as dP / dR in the backward

        # ---- convolution module (conformer.py:147-174) ----
        h = ln(x2, "conv.net.0", "cln")
        u, g, c = f(M, 2 * Ci), f(M, Ci), f(M, Ci)
        ops.gemm(Op(h, D), Op(P["conv.net.2.weight"].view(2 * Ci, D), D), u, M, 2 * Ci, D, bias=P["conv.net.2.bias"])
        ops.glu_fwd(u, g, M, Ci)
        ops.dwconv1d_fwd(g, P["conv.net.4.conv.weight"].view(Ci, k), P["conv.net.4.conv.bias"], c, B, n, Ci, k, blk.pad[0])
        if blk.causal:
            y, bnst = c, None
        else:
            bn = blk.conv.net._modules["5"]
            part = torch.empty(ops.bn_nslabs(M) * 2 * Ci, dtype=torch.float64, device=dev)
            mean, rstd, y = f(Ci), f(Ci), f(M, Ci)
            use_batch = training or bn.running_mean is None
            ops.bn_fwd(c, M, Ci, P["conv.net.5.weight"], P["conv.net.5.bias"], bn.running_mean, bn.running_var,
                       bn.num_batches_tracked if use_batch else None, use_batch, 0.1 if bn.momentum is None else bn.momentum, bn.eps, ACT_NONE,
                       part, mean, rstd, y)
            bnst = (mean, rstd, use_batch)
        s = f(M, Ci)
        ops.swish_fwd(y, s, M * Ci)
        x3 = out_proj(s, Ci, "conv.net.7.weight", "conv.net.7.bias", x2, 1.0, blk.conv_dropout, 5)
        sv["conv"] = (u, g, c, y, s, bnst)

        x4 = ff(x3, "ff2", "ff2", 2)
        out = ln(x4, "post_norm", "pln")
        ctx.sv, ctx.blk, ctx.P, ctx.dims, ctx.training = sv, blk, P, (B, n), training
        return out.view(B, n, D)

    @staticmethod
    def backward(ctx, dout):
        sv, blk, P, (B, n), training = ctx.sv, ctx.blk, ctx.P, ctx.dims, ctx.training
        D = blk.dim
        M = B * n
        dev = dout.device
        H, dh, HD, Fd, Ci, k = blk.heads, blk.dim_head, blk.heads * blk.dim_head, blk.dim * blk.ff_mult, blk.inner, blk.kernel_size
        f = lambda *s: torch.empty(*s, device=dev)          # noqa: E731
        G = {}
        nparts = ops.layernorm_bwd_nparts(M)

        def ln_bwd(dh_, key, name, dres):
            xin, _h, mean, rstd = sv[key]
            dx, part, gb = f(M, D), f(nparts, 2 * D), f(2 * D)
            ops.layernorm_bwd(dh_, xin, mean, rstd, P[name + ".weight"], P[name + ".bias"], dres, dx, None, part, M, D)
            ops.colreduce(part, gb, nparts, 2 * D)
            G[name + ".weight"], G[name + ".bias"] = gb[:D], gb[D:]
            return dx

        def out_proj_bwd(dxo, a, K, wname, bname, scale, p, site):
            """gradient wrt `a` of xin + scale * dropout(a W^T + b); parameter gradients into G."""
            w = P[wname].view(D, K)
            dy = dxo
            if p > 0 and training:
                dy = f(M, D)
                ops.dropout(dxo, dy, None, M * D, sv["seeds"][site], p)
            da, dw = f(M, K), f(D, K)
            ops.gemm(Op(dy, D), Op(w, K), da, M, K, D, b_t=True, alpha=scale)
            _wgrad(dy, a, dw, M, D, K, alpha=scale)
            db = _colsum(dy, M, D)
            if scale != 1.0:
                ops.axpby(db, None, scale, 0.0, db, D)
            G[wname], G[bname] = dw.view(P[wname].shape), db
            return da

        def ff_bwd(dxo, pre, key, site):
            u, a = sv[key]
            h = sv[key + "ln"][1]
            da = out_proj_bwd(dxo, a, Fd, pre + ".fn.fn.net.3.weight", pre + ".fn.fn.net.3.bias", 0.5, blk.ff_dropout, site + 1)
            if blk.ff_dropout > 0 and training:
                ops.dropout(da, da, None, M * Fd, sv["seeds"][site], blk.ff_dropout)
            du = f(M, Fd)
            ops.swish_bwd(da, u, du, M * Fd)
            w1 = P[pre + ".fn.fn.net.0.weight"]
            dh_, dw1 = f(M, D), f(Fd, D)
            ops.gemm(Op(du, Fd), Op(w1, D), dh_, M, D, Fd, b_t=True)
            _wgrad(du, h, dw1, M, Fd, D)
            G[pre + ".fn.fn.net.0.weight"], G[pre + ".fn.fn.net.0.bias"] = dw1, _colsum(du, M, Fd)
            return ln_bwd(dh_, key + "ln", pre + ".fn.norm", dxo)

        dx4 = ln_bwd(dout.contiguous().float().view(M, D), "pln", "post_norm", None)
        dx3 = ff_bwd(dx4, "ff2", "ff2", 2)

        # ---- convolution module ----
        u, g, c, y, s, bnst = sv["conv"]
        h = sv["cln"][1]
        ds = out_proj_bwd(dx3, s, Ci, "conv.net.7.weight", "conv.net.7.bias", 1.0, blk.conv_dropout, 5)
        dy = f(M, Ci)
        ops.swish_bwd(ds, y, dy, M * Ci)
        if bnst is None:
            dc = dy
        else:
            mean, rstd, use_batch = bnst
            part = torch.empty(ops.bn_nslabs(M) * 2 * Ci, dtype=torch.float64, device=dev)
            sums, dgam, dbet, dc = f(2 * Ci), f(Ci), f(Ci), f(M, Ci)
            ops.bn_bwd(dy, y, c, mean, rstd, P["conv.net.5.weight"], M, Ci, ACT_NONE, use_batch, part, sums, dgam, dbet, dc)
            G["conv.net.5.weight"], G["conv.net.5.bias"] = dgam, dbet
        wd = P["conv.net.4.conv.weight"].view(Ci, k)
        dg, dwd, dbd = f(M, Ci), f(Ci, k), f(Ci)
        ops.dwconv1d_fwd(dc, wd, None, dg, B, n, Ci, k, k - 1 - blk.pad[0], flip=True)
        ops.dwconv1d_wgrad(g, dc, f(ops.dwconv1d_wgrad_nslabs(B, n) * (k + 1) * Ci), dwd, dbd, B, n, Ci, k, blk.pad[0])
        G["conv.net.4.conv.weight"], G["conv.net.4.conv.bias"] = dwd.view(Ci, 1, k), dbd
        du = f(M, 2 * Ci)
        ops.glu_bwd(dg, u, du, M, Ci)
        wp1 = P["conv.net.2.weight"].view(2 * Ci, D)
        dh_, dwp1 = f(M, D), f(2 * Ci, D)
        ops.gemm(Op(du, 2 * Ci), Op(wp1, D), dh_, M, D, 2 * Ci, b_t=True)
        _wgrad(du, h, dwp1, M, 2 * Ci, D)
        G["conv.net.2.weight"], G["conv.net.2.bias"] = dwp1.view(2 * Ci, D, 1), _colsum(du, M, 2 * Ci)
        dx2 = ln_bwd(dh_, "cln", "conv.net.0", dx3)

        # ---- attention ----
        qkv, Pm, Eu, cx, mk, dP, dR = sv["attn"]
        h = sv["aln"][1]
        np_, Nr = Pm.shape[1], dR.shape[1]
        dcx = out_proj_bwd(dx2, cx, HD, "attn.fn.to_out.weight", "attn.fn.to_out.bias", 1.0, blk.attn_dropout, 4)
        qop = Op(qkv, 3 * HD, bs1=n * 3 * HD, bs2=dh)
        kop = Op(qkv, 3 * HD, bs1=n * 3 * HD, bs2=dh, offset=HD)
        vop = Op(qkv, 3 * HD, bs1=n * 3 * HD, bs2=dh, offset=2 * HD)
        dcop = Op(dcx, HD, bs1=n * HD, bs2=dh)
        pop = Op(Pm, np_, bs1=H * n * np_, bs2=n * np_)
        bq = dict(nb1=B, nb2=H)
        dqkv = f(M, 3 * HD)
        cq = dict(ldc=3 * HD, c_bs1=n * 3 * HD, c_bs2=dh)
        ops.gemm(pop, dcop, dqkv, n, dh, n, a_t=True, b_t=True, c_offset=2 * HD, **cq, **bq)                     # dV = P^T dctx
        ops.gemm(dcop, vop, dP, n, n, dh, ldc=np_, c_bs1=H * n * np_, c_bs2=n * np_, **bq)                       # dP = dctx V^T
        dS = f(B * H * n, np_)
        ops.relpos_softmax_bwd(Pm, dP, mk, dS, dR, B, H, n, np_, Nr, blk.scale)
        dsop = Op(dS, np_, bs1=H * n * np_, bs2=n * np_)
        drop = Op(dR, Nr, bs1=H * n * Nr, bs2=n * Nr)
        ops.gemm(dsop, kop, dqkv, n, dh, n, b_t=True, **cq, **bq)                                               # dQ = dS K
        ops.gemm(drop, Op(Eu, dh), dqkv, n, dh, Nr, b_t=True, R=dqkv, rmode=1, **cq, **bq)                      #    + dR Eu
        ops.gemm(dsop, qop, dqkv, n, dh, n, a_t=True, b_t=True, c_offset=HD, **cq, **bq)                        # dK = dS^T Q
        slabs, dEu = f(B * H, Nr * dh), f(Nr, dh)
        ops.gemm(drop, qop, slabs, Nr, dh, n, a_t=True, b_t=True, ldc=dh, c_bs1=H * Nr * dh, c_bs2=Nr * dh, **bq)  # dEu per (b, h) = dR^T Q
        ops.reduce_slabs(slabs, dEu, Nr * dh, B * H, Nr * dh)
        dE = f(2 * blk.max_pos_emb + 1, dh)
        ops.relpos_scatter_grad(dEu, dE, n, dh, blk.max_pos_emb)
        G["attn.fn.rel_pos_emb.weight"] = dE
        wq, wkv = P["attn.fn.to_q.weight"], P["attn.fn.to_kv.weight"]
        dh_, dwq, dwkv = f(M, D), f(HD, D), f(2 * HD, D)
        ops.gemm(Op(dqkv, 3 * HD), Op(wq, D), dh_, M, D, HD, b_t=True)
        ops.gemm(Op(dqkv, 3 * HD, offset=HD), Op(wkv, D), dh_, M, D, 2 * HD, b_t=True, R=dh_, rmode=1)
        _wgrad(dqkv, h, dwq, M, HD, D, a_ld=3 * HD)
        _wgrad(dqkv, h, dwkv, M, 2 * HD, D, a_ld=3 * HD, a_off=HD)
        G["attn.fn.to_q.weight"], G["attn.fn.to_kv.weight"] = dwq, dwkv
        dx1 = ln_bwd(dh_, "aln", "attn.norm", dx2)

        dx0 = ff_bwd(dx1, "ff1", "ff1", 0)
        ctx.sv = None
        return (dx0.view(B, n, D), None, None) + tuple(G[nm] for nm in blk._pnames)


class ConformerBlock(nn.Module):
    """model/conformer.py:180-216."""

    def __init__(self, *, dim, dim_head=64, heads=8, ff_mult=4, conv_expansion_factor=2, conv_kernel_size=31, attn_dropout=0.,
                 ff_dropout=0., conv_dropout=0., conv_causal=False, max_pos_emb=512, seed=0):
        super().__init__()
        inner, hd = dim * conv_expansion_factor, dim_head * heads
        if dim % 8 or dim_head % 4 or inner % 8 or (dim * ff_mult) % 8 or hd % 8:
            raise ValueError("ConformerBlock on HIP: dim, dim * ff_mult, dim * conv_expansion_factor and heads * dim_head must be multiples of 8, "
                             "dim_head of 4 (16-byte operand vectors)")
        if not 1 <= conv_kernel_size <= MAX_TAPS:
            raise ValueError("ConformerBlock on HIP: 1 <= conv_kernel_size <= %d" % MAX_TAPS)
        self.dim, self.dim_head, self.heads, self.ff_mult, self.inner, self.kernel_size = dim, dim_head, heads, ff_mult, inner, conv_kernel_size
        self.scale, self.max_pos_emb, self.causal = dim_head ** -0.5, max_pos_emb, conv_causal
        self.attn_dropout, self.ff_dropout, self.conv_dropout, self.seed = attn_dropout, ff_dropout, conv_dropout, seed
        self.pad = _same_padding(conv_kernel_size) if not conv_causal else (conv_kernel_size - 1, 0)

        def feed_forward():
            return _Node(_0=nn.Linear(dim, dim * ff_mult), _3=nn.Linear(dim * ff_mult, dim))

        # creation order = the reference's (ff1, attention, convolution module, ff2, then the norms): one seed, the same initial weights
        ff1 = feed_forward()
        att = _Node(to_q=nn.Linear(dim, hd, bias=False), to_kv=nn.Linear(dim, hd * 2, bias=False), to_out=nn.Linear(hd, dim),
                    rel_pos_emb=nn.Embedding(2 * max_pos_emb + 1, dim_head))
        cn0 = nn.LayerNorm(dim)
        cn2 = nn.Conv1d(dim, inner * 2, 1)
        cn4 = _Node(conv=nn.Conv1d(inner, inner, conv_kernel_size, groups=inner))
        kids = dict(_0=cn0, _2=cn2, _4=cn4)
        if not conv_causal:
            kids["_5"] = nn.BatchNorm1d(inner)
        kids["_7"] = nn.Conv1d(inner, dim, 1)
        ff2 = feed_forward()
        self.ff1 = _Node(fn=_Node(fn=_Node(net=ff1), norm=nn.LayerNorm(dim)))
        self.attn = _Node(fn=att, norm=nn.LayerNorm(dim))
        self.conv = _Node(net=_Node(**kids))
        self.ff2 = _Node(fn=_Node(fn=_Node(net=ff2), norm=nn.LayerNorm(dim)))
        self.post_norm = nn.LayerNorm(dim)
        self._pnames = [nm for nm, _ in self.named_parameters()]

    def forward(self, x, mask=None):
        if x.dim() != 3 or x.shape[-1] != self.dim:
            raise ValueError("ConformerBlock: x must be [B, n, %d]" % self.dim)
        if not x.is_cuda:
            raise RuntimeError("ConformerBlock runs on the HIP kernels only: x must live on the GPU (no CPU fallback)")
        if x.shape[1] > MAX_N:
            raise ValueError("ConformerBlock on HIP: n = %d > %d positions" % (x.shape[1], MAX_N))
        return _BlockFn.apply(x, mask, self, *[p for _, p in self.named_parameters()])


class Conformer(nn.Module):
    """model/conformer.py:219-253 (its blocks take the default dropouts, as there)."""

    def __init__(self, dim, *, depth, dim_head=64, heads=8, ff_mult=4, conv_expansion_factor=2, conv_kernel_size=31, attn_dropout=0.,
                 ff_dropout=0., conv_dropout=0., conv_causal=False):
        super().__init__()
        self.dim = dim
        self.layers = nn.ModuleList([ConformerBlock(dim=dim, dim_head=dim_head, heads=heads, ff_mult=ff_mult,
                                                    conv_expansion_factor=conv_expansion_factor, conv_kernel_size=conv_kernel_size,
                                                    conv_causal=conv_causal) for _ in range(depth)])

    def forward(self, x):
        for block in self.layers:
            x = block(x)
        return x


def flops(B, n, dim, dim_head=64, heads=8, ff_mult=4, conv_expansion_factor=2, conv_kernel_size=31):
    """Multiply-add FLOPs of one block's forward (x 3 for forward + backward)."""
    M, hd, inner = B * n, dim_head * heads, dim * conv_expansion_factor
    lin = 2 * (2 * dim * dim * ff_mult) + 3 * dim * hd + hd * dim + dim * 2 * inner + inner * dim
    att = B * heads * (2 * n * n * dim_head + n * (2 * n - 1) * dim_head)
    return 2 * (M * lin + att + M * inner * conv_kernel_size)


__all__ = ["ConformerBlock", "Conformer", "flops"]
