"""Autograd-visible building blocks of the AASIST / ResNet back-ends, every one of them a HIP kernel behind the C ABI.

The two back-ends (model/wav2vec2_aasist.py:62-604, model/resnet.py:47-191) are a few hundred small layers; what they need from a
device library is: 2-D convolution, BatchNorm (+ReLU / SELU), Linear, a batched matrix product, a 3x3 max pool and global average
pooling — forward and backward.  Here those are

  * convolutions as IMPLICIT GEMMs over channels-last maps: the input is copied once into a zero-padded [B, Hp, Wp, C] map and the
    GEMM kernel addresses the sliding windows in place (rows = output positions with overlapping windows, K = (kh; kw*C + c)
    through the operand's 2-level contiguous index, one batch entry per utterance) — forward, backward-data (a stride-1
    correlation of the zero-dilated, padded output gradient with the flipped weights) and backward-weights (both operands
    transposed, one slab per utterance + a deterministic reduce).  fp32 operands run on the exact-fp32 matrix-core kernel
    (csrc/gemm_f32.hip), bf16 operands on the bf16 kernels (csrc/gemm.hip);
  * BatchNorm + activation, pooling: csrc/nn.hip;  Linear / bmm: the same GEMM entry point.

torch is the tensor container and the autograd tape; no torch.nn.functional compute op of those kinds is called.
Maps are channels-last throughout ([B, H, W, C]); the callers (aasist_head.py, resnet_head.py) are written for that layout.
"""
import weakref

import torch

from . import ops
from .lib import FLAT
from .ops import Op

ACT_NONE, ACT_RELU, ACT_SELU = 0, 1, 2

# Everything here runs on the EXACT f32 matrix-core kernel unless a caller asks otherwise.  With the bf16-pair form (ops.F32X3,
# csrc/gemm_f32.hip, 1.3 - 1.8 x faster per launch) in the FORWARD the reference goldens' outputs still hold 2e-4, but a 5e-6 perturbation of
# the pre-activations flips ReLU / SELU masks and the ResNet's earliest gradients drift to 4e-3 (eval) / 1.5e-2 (train) of their scale
# through 18 layers (bound 2e-3).  The BACKWARD products of a convolution (data and weight gradient: two thirds of the work) are linear maps
# GIVEN the forward's masks: conv2d(..., x3_bwd=True) runs them in the pair form — the masks stay the reference's, the gradients move by
# ~1e-5 (the ResNet back-end asks for it; the per-layer AASIST composition, whose small bias gradients sit at 8e-4 with it, does not).
def _gemm(*a, **kw):
    return ops.gemm(*a, x3=False, **kw)


# Re-laid-out copies of the convolution weights ([Co][kh][kw][Cp] for the forward / wgrad, flipped [Ci][kh][kw][Cop] for the dgrad) are
# pure functions of the weights: they are rebuilt once per optimizer step (the model bumps the epoch), not once per call.
_WEIGHT_EPOCH = 0


def weights_changed():
    """Called by the model after every optimizer step / load_state_dict (the fused AdamW kernel writes through raw pointers, which
    torch's per-tensor version counter does not see)."""
    global _WEIGHT_EPOCH
    _WEIGHT_EPOCH += 1


def _packed(weight, kind, dtype, build):
    if torch.cuda.is_current_stream_capturing():
        return build()          # inside a hipGraph capture the re-layout must be part of the graph: every replay reads the live weights
    cache = weight.__dict__.setdefault("_scl_packed", {})        # lives and dies with the parameter object
    hit = cache.get((kind, dtype))
    if hit is None or hit[0] != (_WEIGHT_EPOCH, weight._version):
        hit = ((_WEIGHT_EPOCH, weight._version), build())
        cache[(kind, dtype)] = hit
    return hit[1]


def _packed_conv(weight, dtype, Co, Ci, kh, kw, Cp, Cop):
    """(forward / weight-gradient operand [Co, kh*kw*Cp], data-gradient operand [Ci, kh*kw*Cop]) of a Conv2d weight: one HIP launch per
    convolution and optimizer step (scl_conv_pack_weights) instead of two fills, a flip and two strided copies."""
    def build():
        w = weight.detach()
        if w.dtype != torch.float32 or not w.is_contiguous():
            w = w.float().contiguous()
        fwd = torch.empty(Co, kh * kw * Cp, dtype=torch.float32, device=w.device)
        bwd = torch.empty(Ci, kh * kw * Cop, dtype=torch.float32, device=w.device)
        ops.conv_pack_weights(w, fwd, bwd, Co, Ci, kh, kw, Cp, Cop)
        return fwd.to(dtype), bwd.to(dtype)
    return _packed(weight, "conv", dtype, build)


# Zero-bordered / zero-dilated staging maps are written in their interior only (pad_nhwc), so a buffer that was zero-filled ONCE can serve
# every later call with the same geometry: 93 torch fills per ResNet step (0.6 ms at batch 32) disappear.  A forward's padded input lives
# until its backward (busy flag, released there or when the node dies); the backward's own maps are free again as soon as their GEMMs
# are queued (stream order).  Inside a hipGraph capture the fill stays part of the graph.
_ZERO_POOL = {}


def _zeros_acquire(key, numel, dtype, dev):
    """-> entry [tensor, busy, generation].  The generation is bumped on every hand-out: whoever releases (or reads, in a backward) passes
    the generation it was given, so a stale owner — the finaliser of a graph whose backward already released the entry, after a later
    forward took it — can neither free nor silently read a buffer that now belongs to someone else (round-5 advisory)."""
    if torch.cuda.is_current_stream_capturing():
        return [torch.zeros(numel, dtype=dtype, device=dev), True, 1]
    lst = _ZERO_POOL.setdefault((key, numel, dtype, dev.index, torch.cuda.current_stream(dev).cuda_stream), [])
    for ent in lst:
        if not ent[1]:
            ent[1] = True
            ent[2] += 1
            return ent
    ent = [torch.zeros(numel, dtype=dtype, device=dev), True, 1]
    if len(lst) < 8:
        lst.append(ent)
    return ent


def _zeros_release(ent, gen=None):
    if gen is None or ent[2] == gen:
        ent[1] = False


def _ceil(v, m):
    return (v + m - 1) // m * m


def _colsum(x2d, M, N):
    """Column sums of a contiguous f32 [M, N] (bias gradients): the self-finishing HIP reduction when its 8-column vectors fit."""
    if N % 8:
        return x2d.view(M, N).sum(0)          # the 2-class / 1-score layers (computed 4 wide): a handful of columns
    out = torch.empty(N, device=x2d.device)
    part = torch.empty(ops.colsum_reduce_nparts(M, N) * N, device=x2d.device)
    ops.colsum_reduce(x2d, part, out, M, N)
    return out


# ---- convolution ---------------------------------------------------------------------------------------------------------------------
class _Conv2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, dtype, x3_bwd=False, grad_in_place=False, residual=None):
        """x [B, H, W, Ci] f32; weight [Co, Ci, kh, kw] (torch layout); bias [Co] or None -> y [B, OH, OW, Co] f32 (+ residual, same shape)."""
        B, H, W, Ci = x.shape
        Co, _, kh, kw = weight.shape
        sh, sw = stride
        ph, pw = padding
        vec = 8 if dtype == torch.bfloat16 else 4
        Cp = _ceil(Ci, vec)
        Hp, Wp = H + 2 * ph, W + 2 * pw
        OH, OW = (Hp - kh) // sh + 1, (Wp - kw) // sw + 1
        dev = x.device
        rowmap = (W, H * W, Hp * Wp * Cp, Wp * Cp, Cp, (ph * Wp + pw) * Cp)
        ent = _zeros_acquire(("xp", id(weight), Ci) + rowmap, B * Hp * Wp * Cp + 4096, dtype, dev)        # tail slack: tile rows past the map are masked, not skipped
        xp = ent[0]
        xc = x.contiguous()
        ops.pad_nhwc(xc, B * H * W, Ci, xp, rowmap)
        ctx.xp_ent, ctx.xp_gen = ent, ent[2]
        weakref.finalize(ctx, _zeros_release, ent, ent[2])
        wk = _packed_conv(weight, dtype, Co, Ci, kh, kw, Cp, _ceil(Co, vec))[0]
        K = kh * kw * Cp
        y = torch.empty(B, OH, OW, Co, dtype=torch.float32, device=dev)
        _gemm(Op(xp, sw * Cp, rpb=OW, rbstride=sh * Wp * Cp, cin=kw * Cp, cout=Wp * Cp, bs1=Hp * Wp * Cp), Op(wk, K), y, OH * OW, Co, K,
                 nb1=B, c_bs1=OH * OW * Co, bias=None if bias is None else bias.detach().float().contiguous(),
                 **({} if residual is None else dict(R=residual.detach().contiguous().float(), rmode=1)))       # the block's skip connection in the epilogue
        ctx.save_for_backward(xp, weight)
        ctx.geom = (B, H, W, Ci, Co, kh, kw, sh, sw, ph, pw, Cp, Hp, Wp, OH, OW, dtype, bias is not None)
        ctx.x3_bwd = bool(x3_bwd)
        ctx.grad_in_place = weight if grad_in_place else None      # the Parameter itself: saved_tensors hands back a fresh tensor object without .grad
        return y

    @staticmethod
    def backward(ctx, dy):
        xp, weight = ctx.saved_tensors
        if ctx.xp_ent[2] != ctx.xp_gen:
            raise RuntimeError("hipnn.conv2d: this node's padded input was handed to a later forward (generation %d, node holds %d) — "
                               "a second backward through a graph whose first backward already released it" % (ctx.xp_ent[2], ctx.xp_gen))
        B, H, W, Ci, Co, kh, kw, sh, sw, ph, pw, Cp, Hp, Wp, OH, OW, dtype, has_bias = ctx.geom
        x3_bwd = ctx.x3_bwd
        dev = dy.device
        dy = dy.contiguous().float()
        vec = 8 if dtype == torch.bfloat16 else 4
        Cop = _ceil(Co, vec)
        K = kh * kw * Cp
        db = None
        if has_bias and ctx.needs_input_grad[2]:
            db = _colsum(dy, B * OH * OW, Co)
        dx = dw = None
        need_dw, need_dx = ctx.needs_input_grad[1], ctx.needs_input_grad[0]
        # A stride-1 "same" convolution (k - 1 = 2 p: every 3x3 / 1x1 of the ResNet body) has a padded output-gradient map dyp of exactly
        # the padded input's geometry, and the weight gradient is then ONE long reduction over the flat padded grid of the whole batch:
        #   dW[co][(r,s,c)] = sum_q dyp[q + off][co] * xp[q + r*Wp + s][c],   off = (k-1-p) * (Wp + 1)
        # (border positions contribute dyp = 0).  Split-K slabs replace the per-utterance slabs: 2-8 partial [Co, K] images instead of
        # 32-256, and hundreds of K steps per block instead of 5 on the deep layers' 9 x 16 maps.  Measured at batch 32 (same box, kernel trace):
        # weight-gradient GEMMs 2.47 -> 2.46 ms, their finishing kernel 0.85 -> 0.68 ms per step.
        flat = need_dw and sh == 1 and sw == 1 and kh - 1 == 2 * ph and kw - 1 == 2 * pw and dtype == torch.float32
        ent_p = dyp = None
        Hd, Wd = H + kh - 1, W + kw - 1
        if need_dx or flat:
            rm = (OW, OH * OW, Hd * Wd * Cop, sh * Wd * Cop, sw * Cop, ((kh - 1 - ph) * Wd + (kw - 1 - pw)) * Cop)
            ent_p = _zeros_acquire(("dyp", Co) + rm, B * Hd * Wd * Cop + 4096, dtype, dev)
            dyp = ent_p[0]
            ops.pad_nhwc(dy, B * OH * OW, Co, dyp, rm)
        if need_dw:
            tile = 64 if dtype == torch.float32 else 128
            ent_c = None
            if flat:
                off = (kh - 1 - ph) * Wp + (kw - 1 - pw)
                R = B * Hp * Wp - (kh - 1) * Wp - (kw - 1)          # the last window that still holds an output position
                # the f32 kernel takes 128 x 128 tiles (2 blocks per CU) once 512 of them exist, 64 x 64 (4 per CU) otherwise: aim at
                # ~1024 blocks, two rounds of the former or one of the latter
                t = 128 if (Co >= 128 and K >= 128) else 64
                tiles = ((Co + t - 1) // t) * ((K + t - 1) // t)
                nslab = max(1, min(128, (1024 + tiles // 2) // tiles, (R // 32) // 8))
                slabs = torch.empty(nslab, Co, K, dtype=torch.float32, device=dev)
                ops.gemm(Op(dyp[off * Cop:], Cop), Op(xp, Cp, cin=kw * Cp, cout=Wp * Cp), slabs, Co, K, R, a_t=True, b_t=True, splitk=nslab,
                         c_split_stride=Co * K if nslab > 1 else 0, x3=x3_bwd)
            else:
                # dW[co][(kh,kw,c)] = sum_b sum_(oh,ow) dy[b,oh,ow,co] * xp[b, oh*sh+kh, ow*sw+kw, c]: both operands transposed (rows = the
                # reduction index), one [Co, K] slab per utterance, summed in a fixed order
                dyc = dy if (dtype == torch.float32 and Cop == Co) else None
                if dyc is None:
                    rmc = (OW, OH * OW, OH * OW * Cop, OW * Cop, Cop, 0)
                    ent_c = _zeros_acquire(("dyc", Co) + rmc, B * OH * OW * Cop + 4096, dtype, dev)
                    dyc = ent_c[0]
                    ops.pad_nhwc(dy, B * OH * OW, Co, dyc, rmc)
                tiles = ((Co + tile - 1) // tile) * ((K + tile - 1) // tile) * B
                sk = max(1, min(8, 1024 // max(tiles, 1), (OH * OW) // 256))          # enough blocks to hide the single-stage prefetch
                nslab = B * sk
                slabs = torch.empty(nslab, Co, K, dtype=torch.float32, device=dev)
                ops.gemm(Op(dyc, Cop, bs1=OH * OW * Cop), Op(xp, sw * Cp, rpb=OW, rbstride=sh * Wp * Cp, cin=kw * Cp, cout=Wp * Cp, bs1=Hp * Wp * Cp),
                         slabs, Co, K, OH * OW, a_t=True, b_t=True, nb1=B, c_bs1=sk * Co * K, splitk=sk, c_split_stride=Co * K if sk > 1 else 0, x3=x3_bwd)
            # slab sum (fixed order) + [Co][kh][kw][Cp] -> torch layout in one kernel; with grad_in_place (parameters whose .grad is an
            # attached flat-buffer view: the model plugins) also autograd's accumulation, and no gradient is handed back to the engine
            g = ctx.grad_in_place.grad if ctx.grad_in_place is not None else None
            if g is not None and g.dtype == torch.float32 and g.is_contiguous() and not torch.cuda.is_current_stream_capturing():
                ops.conv_wgrad_finish(slabs, g, nslab, Co, Ci, kh, kw, Cp, accumulate=True)
            else:
                dw = torch.empty(Co, Ci, kh, kw, dtype=torch.float32, device=dev)
                ops.conv_wgrad_finish(slabs, dw, nslab, Co, Ci, kh, kw, Cp, accumulate=False)
            if ent_c is not None:
                _zeros_release(ent_c)
        if need_dx:
            # dx = stride-1 correlation of the zero-dilated output gradient (padded by k-1-p) with the flipped, transposed weights
            Kd = kh * kw * Cop
            wd = _packed_conv(weight, dtype, Co, Ci, kh, kw, Cp, Cop)[1]
            dx = torch.empty(B, H, W, Ci, dtype=torch.float32, device=dev)
            ops.gemm(Op(dyp, Cop, rpb=W, rbstride=Wd * Cop, cin=kw * Cop, cout=Wd * Cop, bs1=Hd * Wd * Cop), Op(wd, Kd), dx, H * W, Ci, Kd,
                     nb1=B, c_bs1=H * W * Ci, x3=x3_bwd)
        if ent_p is not None:
            _zeros_release(ent_p)
        _zeros_release(ctx.xp_ent, ctx.xp_gen)
        return dx, dw, db, None, None, None, None, None, (dy if ctx.needs_input_grad[8] else None)


def conv2d(x, weight, bias=None, stride=(1, 1), padding=(0, 0), dtype=torch.float32, x3_bwd=False, grad_in_place=False, residual=None):
    """Channels-last 2-D convolution (cross-correlation, as nn.Conv2d): x [B, H, W, Ci] -> [B, OH, OW, Co] (+ residual [B, OH, OW, Co], added in
    the GEMM epilogue; its gradient is the output gradient).  grad_in_place: the weight
    gradient is ADDED into `weight.grad` by the finishing kernel (when that tensor exists) instead of being returned to autograd — for
    `.backward()` callers whose parameters carry attached .grad views (the plugins' flat gradient buffer); torch.autograd.grad callers
    leave it off."""
    return _Conv2dFn.apply(x, weight, bias, tuple(stride), tuple(padding), dtype, x3_bwd, grad_in_place, residual)


# ---- BatchNorm (+ activation) --------------------------------------------------------------------------------------------------------
class _BatchNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, nbt, training, momentum, eps, act, grad_in_place=False):
        C = x.shape[-1]
        xc = x.contiguous().float()
        N = xc.numel() // C
        dev = x.device
        part = torch.empty(ops.bn_nslabs(N) * 2 * C, dtype=torch.float64, device=dev)
        mean, rstd = torch.empty(C, device=dev), torch.empty(C, device=dev)
        y = torch.empty_like(xc)
        ops.bn_fwd(xc, N, C, weight, bias, running_mean, running_var, nbt, training, momentum, eps, act, part, mean, rstd, y)
        ctx.save_for_backward(xc, y, mean, rstd, weight)
        ctx.cfg = (N, C, act, training, weight is not None, bias is not None)
        ctx.in_place = (weight, bias) if grad_in_place else None
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, y, mean, rstd, weight = ctx.saved_tensors
        N, C, act, training, has_w, has_b = ctx.cfg
        dev = dy.device
        dyc = dy.contiguous().float()
        part = torch.empty(ops.bn_nslabs(N) * 2 * C, dtype=torch.float64, device=dev)
        sums = torch.empty(2 * C, device=dev)
        dx = torch.empty_like(xc)
        if ctx.in_place is not None and has_w and has_b and not torch.cuda.is_current_stream_capturing():
            # both parameters carry an attached, contiguous f32 .grad (the plugins' flat gradient buffer): the finishing kernel adds into
            # them (what AccumulateGrad would do with two more launches) and autograd gets no parameter gradient back
            gw, gb = ctx.in_place[0].grad, ctx.in_place[1].grad
            if all(g is not None and g.dtype == torch.float32 and g.is_contiguous() for g in (gw, gb)):
                ops.bn_bwd(dyc, y, xc, mean, rstd, weight, N, C, act, training, part, sums, gw, gb, dx, accumulate=True)
                return dx, None, None, None, None, None, None, None, None, None, None
        dg = torch.empty(C, device=dev) if has_w else None
        db = torch.empty(C, device=dev) if has_b else None
        ops.bn_bwd(dyc, y, xc, mean, rstd, weight, N, C, act, training, part, sums, dg, db, dx)
        return dx, dg, db, None, None, None, None, None, None, None, None


def batch_norm(x, bn, act=ACT_NONE, grad_in_place=False):
    """nn.BatchNorm1d / 2d semantics over the LAST dimension of x (channels-last), fused with `act`; `bn` supplies weight, bias, the
    running statistics and the training flag (its forward is never called).  grad_in_place: as conv2d's."""
    training = bn.training or bn.running_mean is None
    momentum = 0.1 if bn.momentum is None else bn.momentum
    return _BatchNormFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked if training else None, training,
                              momentum, bn.eps, act, grad_in_place)


# ---- Linear / bmm ----------------------------------------------------------------------------------------------------------------------
class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        N, K = weight.shape
        xc = x.contiguous().float().view(-1, K)
        M = xc.shape[0]
        wc = weight.detach().contiguous().float()
        y = torch.empty(M, N, device=x.device)
        _gemm(Op(xc, K), Op(wc, K), y, M, N, K, bias=None if bias is None else bias.detach().contiguous().float())
        ctx.save_for_backward(xc, wc)
        ctx.meta = (x.shape, bias is not None)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        xc, wc = ctx.saved_tensors
        xshape, has_b = ctx.meta
        N, K = wc.shape
        M = xc.shape[0]
        dev = dy.device
        dyc = dy.contiguous().float().view(M, N)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, device=dev)
            _gemm(Op(dyc, N), Op(wc, K), dx, M, K, N, b_t=True)
            dx = dx.view(xshape)
        if ctx.needs_input_grad[1]:
            dw = torch.empty(N, K, device=dev)
            # a [<= 128, <= 128] output is one or two tiles: split the reduction over the rows so that the launch is not a single
            # block walking M / 32 steps at HBM latency, then sum the slabs in a fixed order
            tiles = ((N + 63) // 64) * ((K + 63) // 64)
            sk = max(1, min(32, 256 // tiles, M // 128))
            if sk > 1:
                slabs = torch.empty(sk, N, K, device=dev)
                _gemm(Op(dyc, N), Op(xc, K), slabs, N, K, M, a_t=True, b_t=True, splitk=sk, c_split_stride=N * K)
                ops.reduce_slabs(slabs, dw, N * K, sk, N * K)
            else:
                _gemm(Op(dyc, N), Op(xc, K), dw, N, K, M, a_t=True, b_t=True)
        if has_b and ctx.needs_input_grad[2]:
            db = _colsum(dyc, M, N)
        return dx, dw, db


def linear(x, weight, bias=None):
    """nn.Linear on the exact-fp32 GEMM.  16-byte operand vectors need in_features % 4 == 0 and out_features % 4 == 0: narrower outputs
    (the 2-class and 1-score layers) are computed 4 wide with zero rows and sliced."""
    N, K = weight.shape
    if K % 4:
        pad = 4 - K % 4
        x = torch.nn.functional.pad(x, (0, pad))
        weight = torch.nn.functional.pad(weight, (0, pad))
    if N % 4:
        pad = 4 - N % 4
        weight = torch.nn.functional.pad(weight, (0, 0, 0, pad))
        bias = None if bias is None else torch.nn.functional.pad(bias, (0, pad))
        return _LinearFn.apply(x, weight, bias)[..., :N]
    return _LinearFn.apply(x, weight, bias)


class _BmmFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        """a [B, M, K] @ b [B, K, N] -> [B, M, N]   (K % 4 == 0 and N % 4 == 0)."""
        B, M, K = a.shape
        N = b.shape[2]
        ac, bc = a.contiguous().float(), b.contiguous().float()
        c = torch.empty(B, M, N, device=a.device)
        _gemm(Op(ac, K, bs1=M * K), Op(bc, N, bs1=K * N), c, M, N, K, b_t=True, nb1=B, c_bs1=M * N)
        ctx.save_for_backward(ac, bc)
        return c

    @staticmethod
    def backward(ctx, dc):
        ac, bc = ctx.saved_tensors
        B, M, K = ac.shape
        N = bc.shape[2]
        dcc = dc.contiguous().float()
        da = db = None
        if ctx.needs_input_grad[0]:          # dA[m][k] = sum_n dC[m][n] B[k][n]
            da = torch.empty(B, M, K, device=dc.device)
            _gemm(Op(dcc, N, bs1=M * N), Op(bc, N, bs1=K * N), da, M, K, N, nb1=B, c_bs1=M * K)
        if ctx.needs_input_grad[1]:          # dB[k][n] = sum_m A[m][k] dC[m][n]
            db = torch.empty(B, K, N, device=dc.device)
            _gemm(Op(ac, K, bs1=M * K), Op(dcc, N, bs1=M * N), db, K, N, M, a_t=True, b_t=True, nb1=B, c_bs1=K * N)
        return da, db


def bmm(a, b):
    """torch.bmm on the exact-fp32 GEMM; the contraction and the output width are zero-padded to multiples of 4 (graph sizes such as 42
    or 66 nodes are not)."""
    K, N = a.shape[2], b.shape[2]
    pk, pn = (-K) % 4, (-N) % 4
    if pk:
        a = torch.nn.functional.pad(a, (0, pk))
        b = torch.nn.functional.pad(b, (0, 0, 0, pk))
    if pn:
        b = torch.nn.functional.pad(b, (0, pn))
    c = _BmmFn.apply(a, b)
    return c[..., :N] if pn else c


# ---- pooling ---------------------------------------------------------------------------------------------------------------------------
class _MaxPool3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        """x [B, H, W] f32 (any strides) -> [B, H // 3, W // 3]  (F.max_pool2d(x, (3, 3)) of a single-channel map)."""
        B, H, W = x.shape
        y = torch.empty(B, H // 3, W // 3, device=x.device)
        idx = torch.empty(B, H // 3, W // 3, dtype=torch.int32, device=x.device)
        xf = x.float()
        ops.maxpool3_fwd(xf, xf.stride(1), xf.stride(2), xf.stride(0), H, W, B, y, idx)
        ctx.save_for_backward(idx)
        ctx.shape = (B, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        B, H, W = ctx.shape
        dx = torch.zeros(B, H, W, device=dy.device)
        ops.maxpool3_bwd(dy.contiguous().float(), idx, H, W, B, dx, W, 1, H * W)
        return dx


def max_pool3(x):
    return _MaxPool3Fn.apply(x)


class _AvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        """x [B, R, C] -> [B, C] mean over R."""
        B, R, C = x.shape
        y = torch.empty(B, C, device=x.device)
        ops.avgpool_fwd(x.contiguous().float(), B, R, C, y)
        ctx.shape = (B, R, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, R, C = ctx.shape
        dx = torch.empty(B, R, C, device=dy.device)
        ops.avgpool_bwd(dy.contiguous().float(), B, R, C, dx)
        return dx


def avg_pool_rows(x):
    return _AvgPoolFn.apply(x)
