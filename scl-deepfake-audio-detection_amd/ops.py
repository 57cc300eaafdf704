"""Tensor-level wrappers over the C ABI (torch is used for device memory and the stream only)."""
import ctypes
import os

import torch

from . import lib as L

FLAT = L.FLAT


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---- launch plans ---------------------------------------------------------------------------------
# Every buffer of a train step is static per (batch, length), so the ~1100 C-ABI calls of a step are
# identical from step to step.  The first step RECORDS them as (function, argument list) pairs; later
# steps REPLAY the list (a ctypes call with pre-built arguments costs ~2 us instead of ~40 us of Python
# descriptor building), which keeps the host ahead of the GPU.
# The recording state is PER THREAD: the prefetch thread (scl_amd/prefetch.py) launches augmentation kernels while the main thread may be
# recording a train step — its calls must not end up in that plan (they would be replayed later on freed buffers).
import threading

_TLS = threading.local()


def _rec():
    return getattr(_TLS, "rec", None)


def start_recording():
    _TLS.rec = []
    return _TLS.rec


def stop_recording():
    plan, _TLS.rec = _rec(), None
    return plan


def replay(plan):
    for fn, args, name, _keep in plan:
        rc = fn(*args)
        if rc != 0:
            L.check(rc, name)


def host_callback(fn, *args):
    """Run a Python callback now and, when recording, at the same position of every replay (DP bucket launches)."""
    rec, _TLS.rec = _rec(), None    # C-ABI calls the callback makes itself (optimizer slices) belong to the callback, not to the plan
    try:
        fn(*args)
    finally:
        _TLS.rec = rec
    if rec is not None:
        def _cb(*a, _fn=fn):
            _fn(*a)
            return 0
        rec.append((_cb, list(args), getattr(fn, "__name__", "callback"), None))


def stream_wait(waiter, signaler):
    """`waiter` (torch stream) does not start later work before `signaler`'s work so far is done; recorded into launch plans."""
    _call("scl_stream_wait_stream", ctypes.c_void_p(waiter.cuda_stream), ctypes.c_void_p(signaler.cuda_stream))


_BLOCKING = ("scl_prof_read", "scl_flac", "scl_debug")      # entry points that may wait or run long on the host: they release the interpreter lock


def _call(name, *args, keep=None):
    fn = getattr(L.load() if name.startswith(_BLOCKING) else L.load_launch(), name)      # launches keep the lock (lib.load_launch)
    rc = fn(*args)
    if rc != 0:
        L.check(rc, name)
    rec = _rec()
    if rec is not None:
        rec.append((fn, list(args), name, keep))   # `keep` holds objects the arguments point into (GEMM descriptors)
        return rec[-1]
    return None


def _ptr(t, offset=0):
    return t.data_ptr() + offset * t.element_size()


class Op:
    """One bf16 GEMM operand: element offset(row r, contiguous c) =
    (r // rpb) * rbstride + (r % rpb) * ld + (c // cin) * cout + c % cin (+ batch strides)."""

    __slots__ = ("t", "ld", "rpb", "rbstride", "cin", "cout", "bs1", "bs2", "offset")

    def __init__(self, t, ld, rpb=FLAT, rbstride=0, cin=FLAT, cout=0, bs1=0, bs2=0, offset=0):
        assert t.dtype in (torch.bfloat16, torch.float32), "GEMM operands are bf16, or f32 for the exact-fp32 kernel"
        self.t, self.ld, self.rpb, self.rbstride = t, ld, rpb, rbstride
        self.cin, self.cout, self.bs1, self.bs2, self.offset = cin, cout, bs1, bs2, offset

    def c(self):
        return L.SclOperand(_ptr(self.t, self.offset), self.bs1, self.bs2, self.rbstride, self.cout,
                            self.rpb, self.ld, self.cin, 0)


# f32-operand GEMMs of the back-ends (AASIST / ResNet / BTSE / Conformer convolutions and linears) and of the scoring path as bf16 pairs
# on the bf16 matrix cores (csrc/gemm_f32.hip "f32 x3": ~2e-5 relative per product at ~5x the matrix-core rate); SCL_F32X3=0: the exact
# v_mfma_f32_16x16x4_f32 kernel everywhere.  The losses' Gram matrices and the linear plugin's frame-level head stay on the exact kernel.
F32X3 = os.environ.get("SCL_F32X3", "1") != "0"

_SPLITK_WS = {}      # per stream: f32 slabs of the automatic split-K path (old, smaller ones stay alive: recorded plans point into them)
_AUTO_SPLITK = True


def _auto_splitk(M, N, K, a_t, nb1, nb2, splitk, f32):
    """Few output tiles and a long reduction (N = 1024 linears of a pack-sized step: 144 tiles of 128 x 128, K = 3072 / 4096): three
    partial GEMMs + scl_gemm_splitk_finish beat the single launch 57 -> 37 us (tools/splitk_probe.py); shorter reductions and
    problems that fill the CUs do not."""
    if not _AUTO_SPLITK or splitk != 1 or nb1 != 1 or nb2 != 1 or a_t or f32 or K < 3072 or K % 64:
        return 1
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    if tiles > 170:
        return 1
    sk = min(K // 1024, 512 // tiles, 4)
    return sk if sk >= 2 else 1


def gemm(A, B, C, M, N, K, *, a_t=False, b_t=False, bias=None, act=0, c2=None, R=None, rmode=0, ract=0,
         alpha=1.0, nb1=1, nb2=1, splitk=1, ldc=None, c_rpb=FLAT, c_rbstride=0, c_bs1=0, c_bs2=0,
         c_offset=0, bias_bs2=0, bias_offset=0, drop_p=0.0, drop_seed=0, c_split_stride=0, no_dma=False, no_big=False, no_p8=False, force_p8=False, force_big=False, no_w8=False, force_w8=False, force_x2=False, no_x2=False, colsum_part=None, x3=None, split3=False):
    """C[z][m][n] = epilogue(alpha * sum_k A[z][m][k] B[z][n][k]) — see include/scl_hip.h.
    x3 (f32 operands only): the bf16-pair form of the f32 kernel (SCL_GEMM_F32X3); None = the module default F32X3, False = the exact kernel.
    split3: C (bf16, ldc = 3 N) receives the triple-plane row image of the result (see scl_split3_f32_bf16)."""
    d = _gemm_desc(A, B, C, M, N, K, a_t=a_t, b_t=b_t, bias=bias, act=act, c2=c2, R=R, rmode=rmode, ract=ract, alpha=alpha, nb1=nb1,
                   nb2=nb2, splitk=splitk, ldc=ldc, c_rpb=c_rpb, c_rbstride=c_rbstride, c_bs1=c_bs1, c_bs2=c_bs2, c_offset=c_offset,
                   bias_bs2=bias_bs2, bias_offset=bias_offset, drop_p=drop_p, drop_seed=drop_seed, c_split_stride=c_split_stride,
                   no_dma=no_dma, no_big=no_big, no_p8=no_p8, force_p8=force_p8, force_big=force_big, no_w8=no_w8, force_w8=force_w8,
                   force_x2=force_x2, no_x2=no_x2, colsum_part=colsum_part, x3=x3)
    if split3:      # bf16 C as [hi | hi | lo] planes of N columns (ldc = 3 N): SCL_GEMM_C_SPLIT3, wide tiles only (SclError otherwise)
        d.flags |= L.GEMM_C_SPLIT3
    sk = 1 if split3 else _auto_splitk(M, N, K, a_t, nb1, nb2, splitk, A.t.dtype == torch.float32)
    if sk > 1 and not (force_w8 or force_p8 or force_big or force_x2):
        key = (C.device.index, torch.cuda.current_stream(C.device).cuda_stream)
        ws = _SPLITK_WS.get(key)
        if ws is None or ws[-1].numel() < sk * M * N:
            ws = (ws or []) + [torch.empty(sk * M * N, dtype=torch.float32, device=C.device)]
            _SPLITK_WS[key] = ws
        slabs = ws[-1]
        part = _gemm_desc(A, B, slabs, M, N, K, b_t=b_t, alpha=alpha, splitk=sk, c_split_stride=M * N, no_dma=no_dma, no_w8=True)
        _call("scl_gemm_bf16", ctypes.byref(part), _stream(), keep=part)
        _call("scl_gemm_splitk_finish", ctypes.byref(d), _ptr(slabs), sk, M * N, _stream(), keep=d)
        return d
    _call("scl_gemm_bf16", ctypes.byref(d), _stream(), keep=d)
    return d


def gemm_group(problems):
    """problems: list of (A, B, C, M, N, K) weight-gradient contractions C[M, N] (f32) = A^T B over K rows — one launch, whole reduction per
    block, no split-K slabs (include/scl_hip.h: scl_gemm_bf16_group).  Returns False (nothing launched) when the list does not qualify."""
    arr = (L.SclGemmDesc * len(problems))()
    for i, (A, B, C, M, N, K) in enumerate(problems):
        d = _gemm_desc(A, B, C, M, N, K, a_t=True, b_t=True)
        ctypes.memmove(ctypes.addressof(arr[i]), ctypes.addressof(d), ctypes.sizeof(L.SclGemmDesc))
    if len(problems) > 8 or not L.load().scl_gemm_bf16_group_ok(arr, len(problems)):
        return False
    _call("scl_gemm_bf16_group", arr, len(problems), _stream(), keep=arr)
    return True


def gemm_group_tiles(A, B, C, M, N, K):
    """256 x 256 output tiles of one weight-gradient problem as a group member; 0 if it does not qualify."""
    d = _gemm_desc(A, B, C, M, N, K, a_t=True, b_t=True)
    return L.load().scl_gemm_bf16_group_tiles(ctypes.byref(d))


def gemm_group_part(parts):
    """parts: list of ((A, B, C, M, N, K), first_tile, n_tiles) — one launch over tile RANGES of up to 8 weight-gradient problems."""
    n = len(parts)
    arr = (L.SclGemmDesc * n)()
    t0, nt = (ctypes.c_int32 * n)(), (ctypes.c_int32 * n)()
    for i, ((A, B, C, M, N, K), first, count) in enumerate(parts):
        d = _gemm_desc(A, B, C, M, N, K, a_t=True, b_t=True)
        ctypes.memmove(ctypes.addressof(arr[i]), ctypes.addressof(d), ctypes.sizeof(L.SclGemmDesc))
        t0[i], nt[i] = first, count
    _call("scl_gemm_bf16_group_part", arr, t0, nt, n, _stream(), keep=(arr, t0, nt))


def gemm_colsum_rows(A, B, C, M, N, K, **kw):
    """Partial rows [rows, N] f32 a gemm(..., colsum_part=...) with the same arguments writes (sum them with colreduce); 0 when the
    call would not run on the wide tiles, which alone produce the fused column sums."""
    d = _gemm_desc(A, B, C, M, N, K, **kw)
    return L.load().scl_gemm_colsum_rows(ctypes.byref(d))


def gemm_wide_kind(A, B, C, M, N, K, **kw):
    """0 if scl_gemm_bf16 would use the 128x128 tiles for this call, 1 / 2 for the wide tiles (gemm_w8.hip); same arguments as gemm()."""
    d = _gemm_desc(A, B, C, M, N, K, **kw)
    return L.load().scl_gemm_uses_wide_tiles(ctypes.byref(d))


def _gemm_desc(A, B, C, M, N, K, *, a_t=False, b_t=False, bias=None, act=0, c2=None, R=None, rmode=0, ract=0,
               alpha=1.0, nb1=1, nb2=1, splitk=1, ldc=None, c_rpb=FLAT, c_rbstride=0, c_bs1=0, c_bs2=0,
               c_offset=0, bias_bs2=0, bias_offset=0, drop_p=0.0, drop_seed=0, c_split_stride=0, no_dma=False, no_big=False, no_p8=False,
               force_p8=False, force_big=False, no_w8=False, force_w8=False, force_x2=False, no_x2=False, colsum_part=None, x3=None):
    d = L.SclGemmDesc()
    d.A, d.B = A.c(), B.c()
    flags = 0
    assert A.t.dtype == B.t.dtype, "A and B must have the same element type"
    if A.t.dtype == torch.float32:
        flags |= L.GEMM_AB_F32
        if F32X3 if x3 is None else x3:
            flags |= L.GEMM_F32X3
    if a_t:
        flags |= L.GEMM_A_T
    if b_t:
        flags |= L.GEMM_B_T
    if C.dtype == torch.float32:
        flags |= L.GEMM_C_F32
    else:
        assert C.dtype == torch.bfloat16
    d.C = _ptr(C, c_offset)
    if c2 is not None:
        flags |= L.GEMM_HAS_C2
        if c2.dtype == torch.float32:
            flags |= L.GEMM_C2_F32
        d.C2 = _ptr(c2, c_offset)
    if R is not None:
        assert rmode in (1, 2)
        if R.dtype == torch.float32:
            flags |= L.GEMM_R_F32
        d.R = _ptr(R, c_offset)
    if bias is not None:
        assert bias.dtype == torch.float32
        flags |= L.GEMM_HAS_BIAS
        d.bias = _ptr(bias, bias_offset)
    if drop_p > 0.0:
        flags |= L.GEMM_DROPOUT
    if colsum_part is not None:
        assert colsum_part.dtype == torch.float32
        d.colsum_part = _ptr(colsum_part)
    if no_dma:
        flags |= L.GEMM_NO_DMA
    if no_big:
        flags |= L.GEMM_NO_BIG
    if force_big:
        flags |= L.GEMM_FORCE_BIG
    if no_p8:
        flags |= L.GEMM_NO_P8
    if force_p8:
        flags |= L.GEMM_FORCE_P8
    if no_w8:
        flags |= L.GEMM_NO_W8
    if force_w8:
        flags |= L.GEMM_FORCE_W8
    if force_x2:
        flags |= L.GEMM_FORCE_X2
    if no_x2:
        flags |= L.GEMM_NO_X2
    flags |= (act << L.ACT_SHIFT) | (rmode << L.RMODE_SHIFT) | (ract << L.RACT_SHIFT)
    d.c_bs1, d.c_bs2, d.c_rbstride, d.c_split_stride, d.bias_bs2 = c_bs1, c_bs2, c_rbstride, c_split_stride, bias_bs2
    d.c_rpb, d.ldc = c_rpb, (N if ldc is None else ldc)
    d.M, d.N, d.K, d.nb1, d.nb2, d.splitk = M, N, K, nb1, nb2, splitk
    d.flags, d.alpha, d.drop_p, d.drop_seed = flags, alpha, drop_p, drop_seed
    return d


def reduce_slabs(slabs, out, n, nslabs, stride):
    _call("scl_reduce_slabs_f32", _ptr(slabs), _ptr(out), n, nslabs, stride, _stream())


def prof_enable(kid, on):
    _call("scl_prof_enable", kid, 1 if on else 0)


def prof_reserve(kid, n_pairs):
    _call("scl_prof_reserve", kid, n_pairs)


def prof_read_launches(kid, cap=65536):
    """[(ms, (M, N, K, flags, z, variant, 0, 0))] of the launches scl_prof_read would sum, in issue order; call before prof_read."""
    n = ctypes.c_int64(0)
    _call("scl_prof_read_launches", kid, 0, None, None, ctypes.byref(n))
    k = min(int(n.value), cap)
    if k == 0:
        return []
    ms = (ctypes.c_float * k)()
    meta = (ctypes.c_int32 * (8 * k))()
    _call("scl_prof_read_launches", kid, k, ms, meta, ctypes.byref(n))
    return [(float(ms[i]), tuple(int(meta[8 * i + j]) for j in range(8))) for i in range(k)]


def prof_read(kid):
    n, ms, fl = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
    _call("scl_prof_read", kid, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl))
    return n.value, ms.value, fl.value


# ------------------------------------------------------------------------------------------------
# thin wrappers (one per C entry point); `None` tensors become NULL
# ------------------------------------------------------------------------------------------------
def _p(t, offset=0):
    return None if t is None else _ptr(t, offset)


def _isf32(t):
    return 1 if t.dtype == torch.float32 else 0


def layernorm_fwd(x, gamma, beta, y_bf16, y_f32, mean, rstd, M, C, act=0, ldx=None, ldy=None, eps=1e-5):
    _call("scl_layernorm_fwd", _p(x), _isf32(x), _p(gamma), _p(beta), _p(y_bf16), _p(y_f32), _p(mean), _p(rstd),
                                       M, C, ldx or C, ldy or C, eps, act, _stream())


def layernorm_bwd_nparts(M):
    return L.load().scl_layernorm_bwd_nparts(M)


LN_BWD_DIN_SEED, LN_BWD_DOUT_SEED = 22, 24      # positions of the two mask seeds in a recorded scl_layernorm_bwd call
ATTN_FWD_SEED, ATTN_BWD_SEED, DROPOUT_SEED = 9, 12, 4


def layernorm_bwd(dy, x, mean, rstd, gamma, beta, dres, dx_f32, dx_bf16, part, M, C, act=0, sum_dres=False, out_rpb=0,
                  out_rbstride=0, out_off=0, din=(0, 0.0), dout=(0, 0.0)):
    """part: f32 [nparts, 2*C] — per-slab (dgamma | dbeta) partial sums; [nparts, 3*C] with sum_dres (| colsum(dres)).
    out_rpb > 0: the bf16 output is written with per-utterance row padding (see include/scl_hip.h)."""
    return _call("scl_layernorm_bwd", _p(dy), _isf32(dy), _p(x), _isf32(x), _p(mean), _p(rstd), _p(gamma), _p(beta),
                 _p(dres), _p(dx_f32), _p(dx_bf16), _p(part), M, C, C, C, C, act, int(sum_dres), out_rpb, out_rbstride, out_off,
                 int(din[0]), float(din[1]), int(dout[0]), float(dout[1]), _stream())


def colreduce(part, out, nparts, C, pstride=None, accumulate=False):
    _call("scl_colreduce_f32", _p(part), _p(out), nparts, C, pstride or C, 1 if accumulate else 0, _stream())


def colreduce_multi(jobs):
    """jobs: list of (part, out, nparts, C[, out2, split]) — up to 8 column reductions in one launch."""
    arr = (L.SclReduceJob * len(jobs))()
    for i, jb in enumerate(jobs):
        part, out, nparts, C = jb[:4]
        out2, split = (jb[4], jb[5]) if len(jb) > 4 and jb[4] is not None else (None, 0)
        arr[i].part, arr[i].out, arr[i].out2 = _p(part), _p(out), _p(out2)
        arr[i].pstride, arr[i].nparts, arr[i].C, arr[i].split = C, nparts, C, split
    _call("scl_colreduce_multi", arr, len(jobs), _stream(), keep=arr)


def colsum_nparts(M):
    return L.load().scl_colsum_nparts(M)


def colsum_reduce_nparts(M, N):
    """Partial rows colsum_reduce writes for an [M, N] input (more than colsum_nparts(M) when N <= 128)."""
    return L.load().scl_colsum_reduce_nparts(M, N)


def colsum(x, part, M, N, ld=None):
    _call("scl_colsum", _p(x), _isf32(x), _p(part), M, N, ld or N, _stream())


_COUNTERS = {}


def _counters(device):
    """Ticket counters of the self-finishing reductions (zero between launches; one array per device, launches are stream-ordered)."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)   # one set per stream
    if key not in _COUNTERS:
        _COUNTERS[key] = torch.zeros(128, dtype=torch.int32, device=device)
    return _COUNTERS[key]


_SCRATCH = {}


def colreduce_seg(part, out, nparts, C, pstride=None, accumulate=False, out2=None, split=0):
    """colreduce over 8x more blocks, finished in-launch (LayerNorm parameter gradients: 768 partial rows x 2C or 3C);
    columns >= split go to out2 when given."""
    key = (part.device.type, part.device.index, torch.cuda.current_stream(part.device).cuda_stream if part.is_cuda else 0)
    if key not in _SCRATCH:
        _SCRATCH[key] = torch.empty(8 * 4096, dtype=torch.float32, device=part.device)
    assert C <= 4096
    _call("scl_colreduce_seg_f32", _p(part), _p(out), nparts, C, pstride or C, 1 if accumulate else 0, _p(_SCRATCH[key]),
          _p(_counters(part.device)), _p(out2), split, _stream())


def colsum_reduce(x, part, out, M, N, ld=None):
    """out[n] = sum_m x[m][n] in one launch (part: f32 [colsum_nparts(M), N] scratch)."""
    _call("scl_colsum_reduce", _p(x), _isf32(x), _p(part), _p(_counters(x.device)), _p(out), M, N, ld or N, _stream())


def cast_bf16(src, dst, n=None):
    _call("scl_cast_f32_bf16", _p(src), _p(dst), n or src.numel(), _stream())


def split3(x, rows, K, out, order, ldx=None, x_offset=0):
    """f32 [rows][K] -> bf16 [rows][3 K] as [hi | hi | lo] (order 0) / [hi | lo | hi] (order 1): csrc/elementwise.hip split3_kernel."""
    _call("scl_split3_f32_bf16", _ptr(x, x_offset), rows, K, ldx or K, _p(out), order, _stream())


def add_f32(a, b, out, out_bf16, n):
    _call("scl_add_f32", _p(a), _p(b), _p(out), _p(out_bf16), n, _stream())


def pad_rows(src, dst, B, T, C, rows_out, pad_before, pre=None, ract=0):
    _call("scl_pad_rows_bf16", _p(src), _isf32(src), _p(dst), _p(pre), ract, B, T, C, rows_out, pad_before, _stream())


def posconv_supported(T, K, G, Cg):
    return bool(L.load().scl_posconv_supported(T, K, G, Cg))


def posconv_mfma(xpad, w, C, R, B, T, K, G, Cg, bias=None, c2=None):
    """Grouped positional conv with the utterance slab resident in LDS (csrc/posconv.hip): forward form (bias, GELU, pre-activation copy
    c2, + R) when bias is given, data-gradient form (+ R) otherwise."""
    return _call("scl_posconv_mfma", _p(xpad), _p(w), _p(C), _p(bias), _p(c2), _p(R), B, T, K, G, Cg, 1 if bias is not None else 0, _stream())


def posconv_wgrad_supported(T, K, G, Cg):
    return bool(L.load().scl_posconv_wgrad_supported(T, K, G, Cg))


def posconv_wgrad(dypad, dy_row0, xpad, dw, B, T, K, G, Cg):
    """dw[g][o][tap * Cg + c] (f32, overwritten) = sum over utterances and frames of dY x shifted X (csrc/posconv.hip)."""
    return _call("scl_posconv_wgrad", _p(dypad), dy_row0, _p(xpad), _p(dw), B, T, K, G, Cg, _stream())


def col2im(dcol, dz, B, Tin, Tout, C, k, s):
    _call("scl_col2im_bf16", _p(dcol), _p(dz), B, Tin, Tout, C, k, s, _stream())


def conv_weight_pack(w, wk, Co, Ci, k, wd=None, stride=1):
    _call("scl_conv_weight_pack", _p(w), _p(wk), _p(wd), Co, Ci, k, stride, _stream())


def conv_weight_unpack_grad(dwk, dw, Co, Ci, k):
    _call("scl_conv_weight_unpack_grad", _p(dwk), _p(dw), Co, Ci, k, _stream())


def posconv_weight_pack(v, g, norm, wf, wd, E, Cg, K):
    _call("scl_posconv_weight_pack", _p(v), _p(g), _p(norm), _p(wf), _p(wd), E, Cg, K, _stream())


def posconv_weight_bwd(dwf, v, g, norm, sdot_ws, dv, dg, E, Cg, K):
    _call("scl_posconv_weight_bwd", _p(dwf), _p(v), _p(g), _p(norm), _p(sdot_ws), _p(dv), _p(dg), E, Cg, K, _stream())


def meanpool_fwd(h, emb, B, T, C):
    _call("scl_meanpool_fwd_f32" if _isf32(h) else "scl_meanpool_fwd", _p(h), _p(emb), B, T, C, _stream())


def meanpool_bwd(demb, pre, dpre, B, T, C, ract, drop_p=0.0, seed=0):
    assert pre.dtype == dpre.dtype
    return _call("scl_meanpool_bwd_f32" if _isf32(pre) else "scl_meanpool_bwd", _p(demb), _p(pre), _p(dpre), B, T, C, ract, drop_p, seed, _stream())


def utt_head_fwd(emb, W, bias, logp, B, C, NC):
    _call("scl_utt_head_fwd", _p(emb), _p(W), _p(bias), _p(logp), B, C, NC, _stream())


def utt_head_bwd(dlogp, logp, emb, W, demb_in, demb, dW, db, ws, B, C, NC):
    _call("scl_utt_head_bwd", _p(dlogp), _p(logp), _p(emb), _p(W), _p(demb_in), _p(demb), _p(dW), _p(db), _p(ws),
                                      B, C, NC, _stream())


def softmax_fwd(S, P, R, T, ldS, Tp):
    _call("scl_softmax_fwd", _p(S), _p(P), R, T, ldS, Tp, _stream())


def softmax_fwd_f32(S, P, R, T, ldS, Tp):
    _call("scl_softmax_fwd_f32", _p(S), _p(P), R, T, ldS, Tp, _stream())


def softmax_bwd(P, dP, dS, R, T, lddP, Tp):
    _call("scl_softmax_bwd", _p(P), _p(dP), _p(dS), R, T, lddP, Tp, _stream())


def attn_fwd(qkv, ctx, lse, B, T, H, D, scale, drop_p=0.0, drop_seed=0):
    return _call("scl_attn_fwd", _p(qkv), _p(ctx), _p(lse), B, T, H, D, scale, float(drop_p), int(drop_seed), _stream())


def attn_fwd_fp8(qkv, ctx, lse, B, T, H, D, scale):
    """fp8 (e4m3) operands, fp32 accumulation: the configs[4] variant of the fused attention forward (no dropout)."""
    return _call("scl_attn_fwd_fp8", _p(qkv), _p(ctx), _p(lse), B, T, H, D, float(scale), _stream())


def attn_bwd(qkv, ctx, dctx, lse, dqkv, B, T, H, D, scale, bias_part=None, drop_p=0.0, drop_seed=0):
    """bias_part: optional f32 [B, 3*H*D] — per-utterance column sums of dqkv (colreduce over B gives the q/k/v bias gradients)."""
    return _call("scl_attn_bwd", _p(qkv), _p(ctx), _p(dctx), _p(lse), _p(dqkv), _p(bias_part), B, T, H, D, scale, float(drop_p), int(drop_seed), _stream())


def dropout(x, y_f32, y_bf16, n, seed, p):
    """y = x * keep-mask(seed, i) / (1 - p) to f32 and / or bf16 (in place allowed)."""
    return _call("scl_dropout_f32", _p(x), _p(y_f32), _p(y_bf16), n, int(seed), float(p), _stream())


DROPOUT_ROWS_SEED = 6


def dropout_rows(x, y, R, T, ld, seed, p):
    """Attention dropout on the un-fused path: y[r][j] = x[r][j] * keep-mask(seed, r * T + j) / (1 - p), zero padding columns (bf16 or f32)."""
    return _call("scl_dropout_rows", _p(x), _p(y), R, T, ld, 1 if x.dtype == torch.float32 else 0, int(seed), float(p), _stream())


def conv0_fwd(x, w, b, gamma, beta, z, B, Lx, C, k, stride, eps=1e-5, stats=None):
    _call("scl_conv0_fwd", _p(x), _p(w), _p(b), _p(gamma), _p(beta), _p(z), _p(stats), B, Lx, C, k, stride, eps, _stream())


def conv0_fwd_f32(x, w, b, gamma, beta, z, B, Lx, C, k, stride, eps=1e-5):
    _call("scl_conv0_fwd_f32", _p(x), _p(w), _p(b), _p(gamma), _p(beta), _p(z), B, Lx, C, k, stride, eps, _stream())


def conv0_bwd_nparts(B, Lx, k, stride):
    return L.load().scl_conv0_bwd_nparts(B, Lx, k, stride)


def conv0_bwd(x, w, b, gamma, beta, dz, part_ws, dW, db, dgamma, dbeta, B, Lx, C, k, stride, eps=1e-5, stats=None):
    _call("scl_conv0_bwd", _p(x), _p(w), _p(b), _p(gamma), _p(beta), _p(dz), _p(stats), _p(part_ws), _p(dW), _p(db), _p(dgamma),
                                   _p(dbeta), B, Lx, C, k, stride, eps, _stream())


def supcon_nchunks(K):
    return L.load().scl_supcon_nchunks(K)


def supcon_ws_floats(bz, K):
    """floats of scl_supcon_fwd's workspace: partial Gram matrices + one row of per-utterance losses (used by batches of more than 128)"""
    return int(L.load().scl_supcon_ws_floats(bz, K))


def supcon_fwd(F, labels, bz, K, ldF, Tprime, temperature, ws, G, loss_out, S_out=None):
    assert G.numel() >= 2 * bz * bz and ws.numel() >= supcon_nchunks(K) * bz * bz + (bz if bz > 128 else 0), \
        "supcon: G holds 2 bz^2 floats (dL/dS + backward scratch), ws supcon_ws_floats(bz, K)"
    _call("scl_supcon_fwd", _p(F), _p(labels), bz, K, ldF, Tprime, temperature, _p(ws), _p(G), _p(loss_out), _p(S_out),
                                    _stream())


def supcon_bwd(F, G, upstream, coef, bz, K, ldF, Tprime, temperature, dF, dF_bf16=None, accumulate=False):
    assert G.numel() >= 2 * bz * bz, "supcon: G holds 2 bz^2 floats (dL/dS + backward scratch)"
    _call("scl_supcon_bwd", _p(F), _p(G), _p(upstream), coef, bz, K, ldF, Tprime, temperature, _p(dF), _p(dF_bf16),
                                    1 if accumulate else 0, _stream())


def nll_fwd(logp, labels, bz, NC, loss_out, dlogp_coef):
    _call("scl_nll_fwd", _p(logp), _p(labels), bz, NC, _p(loss_out), _p(dlogp_coef), _stream())


def adamw_flat(p, g, m, v, p_bf16, n, lr, beta1, beta2, eps, wd, step, grad_scale=1.0):
    _call("scl_adamw_flat", _p(p), _p(g), _p(m), _p(v), _p(p_bf16), n, lr, beta1, beta2, eps, wd, step, grad_scale,
                                    _stream())


def reduce_slabs_multi(jobs):
    """jobs: list of (slabs, out, n, nslabs, stride) — up to 8 split-K combines in one launch, each bit-identical to reduce_slabs."""
    arr = (L.SclSlabJob * len(jobs))()
    for i, (slabs, out, n, nslabs, stride) in enumerate(jobs):
        arr[i].slabs, arr[i].out, arr[i].n, arr[i].stride, arr[i].nslabs = _p(slabs), _p(out), n, stride, nslabs
    _call("scl_reduce_slabs_multi", arr, len(jobs), _stream(), keep=arr)


def fir_nblocks(Lout):
    return L.load().scl_fir_nblocks(Lout)


def fir_multi(x, ldx, Lin, taps, tap_off, tap_len, tap_h, nclip, nf, use_pow, y, ldy, Lout, part=None):
    _call("scl_fir_multi_f32", _p(x), ldx, Lin, _p(taps), _p(tap_off), _p(tap_len), _p(tap_h), nclip, nf,
                                       1 if use_pow else 0, _p(y), ldy, Lout, _p(part), _stream())


def clip_stats(x, ldx, Lx, nclip, part):
    _call("scl_clip_stats_f32", _p(x), ldx, Lx, nclip, _p(part), _stream())


def isd_scatter(y, ldy, pos, fr, clip_off, nclip, max_per_clip, g_sd):
    _call("scl_isd_scatter_f32", _p(y), ldy, _p(pos), _p(fr), _p(clip_off), nclip, max_per_clip, g_sd, _stream())


AFF_CENTER_PEAK_COND, AFF_PEAK_COND, AFF_PEAK_ALWAYS, AFF_SSI_MIX, AFF_PEAK_QUANT_I16 = 0, 1, 2, 3, 4


def clip_affine(mode, x, ldx, partx, out, ldo, Lx, nclip, z=None, ldz=0, partz=None, snr_db=None):
    _call("scl_clip_affine_f32", mode, _p(x), ldx, _p(z), ldz, _p(partx), _p(partz), _p(snr_db), _p(out), ldo, Lx, nclip,
                                         _stream())


def f32_to_i16_wrap(x, out_i16, n):
    _call("scl_f32_to_i16_wrap", _p(x), _p(out_i16), n, _stream())


def i16_sumsq(x_i16, n, part64, nparts):
    _call("scl_i16_sumsq", _p(x_i16), n, _p(part64), nparts, _stream())


def i16_gain_overlay(speech_i16, n, noise_i16, nn, factor, out_f32=None, out_i16=None):
    _call("scl_i16_gain_overlay", _p(speech_i16), n, _p(noise_i16), nn, factor, _p(out_f32), _p(out_i16), _stream())


def multiview_crop(src, off, lens, V, firstlen, start, out_len, repeat_pad, out, ldo):
    _call("scl_multiview_crop_f32", _p(src), _p(off), _p(lens), V, firstlen, start, out_len, 1 if repeat_pad else 0,
                                            _p(out), ldo, _stream())


def i16_append_xfade(out_i16, n1, chunk_i16, n2, a0, R, fade1, m2, fade2, tail_off, tail_n, frames_per_ms):
    """fade = (per_ms, from_power, scale_step).  `chunk_i16` may be a view into the source clip."""
    _call("scl_i16_append_xfade", _p(out_i16), n1, _p(chunk_i16), n2, a0, R, int(fade1[0]), float(fade1[1]), float(fade1[2]), m2,
          int(fade2[0]), float(fade2[1]), float(fade2[2]), tail_off, tail_n, frames_per_ms, _stream())


def stft_nframes(n):
    return L.load().scl_stft_nframes(n)


def stft(y, n, D, nframes):
    _call("scl_stft_f32", _p(y), n, _p(D), nframes, _stream())


def phase_vocoder(D, nframes, rate, out, nsteps):
    _call("scl_phase_vocoder_c64", _p(D), nframes, float(rate), _p(out), nsteps, _stream())


def istft(D, nframes, frames_ws, y, length):
    _call("scl_istft_f32", _p(D), nframes, _p(frames_ws), _p(y), length, _stream())


def resample_sinc(x, n_in, ratio, out, n_out):
    _call("scl_resample_sinc_f32", _p(x), n_in, float(ratio), _p(out), n_out, _stream())


def gat_score_nblocks(N):
    return L.load().scl_gat_score_nblocks(N)


def gat_score_fwd(x, W, bias, a, s, B, N, D, Do, n1):
    _call("scl_gat_score_fwd", _p(x), _p(W), _p(bias), _p(a), _p(s), B, N, D, Do, n1, _stream())


def gat_score_bwd(x, W, bias, a, ds, dP, part, dx, B, N, D, Do, n1):
    _call("scl_gat_score_bwd", _p(x), _p(W), _p(bias), _p(a), _p(ds), _p(dP), _p(part), _p(dx), B, N, D, Do, n1, _stream())


# ---- back-end pieces over channels-last fp32 maps (csrc/nn.hip) ----------------------------------------------------------------------
def bn_nslabs(N):
    return L.load().scl_bn_nslabs(N)


def bn_fwd(x, N, C, gamma, beta, running_mean, running_var, nbt, training, momentum, eps, act, part, mean, rstd, y, y2=None, rowmap=None):
    """rowmap = (W, HW, bs, rs, cs, base) of the mapped second output y2 (f32 or bf16)."""
    W_, HW, bs, rs, cs, base = rowmap if rowmap is not None else (1, 1, 0, 0, 0, 0)
    _call("scl_bn_fwd", _p(x), N, C, _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(nbt), 1 if training else 0, momentum, eps, act,
          _p(part), _p(mean), _p(rstd), _p(y), _p(y2), 1 if (y2 is not None and y2.dtype == torch.bfloat16) else 0, W_, HW, bs, rs, cs, base, _stream())


def bn_bwd(dy, y, x, mean, rstd, gamma, N, C, act, training, part, sums, dgamma, dbeta, dx, accumulate=False):
    _call("scl_bn_bwd", _p(dy), _p(y), _p(x), _p(mean), _p(rstd), _p(gamma), N, C, act, 1 if training else 0, _p(part), _p(sums), _p(dgamma),
          _p(dbeta), _p(dx), 1 if accumulate else 0, _stream())


def pad_nhwc(src, rows, C, dst, rowmap):
    W_, HW, bs, rs, cs, base = rowmap
    _call("scl_pad_nhwc_f32", _p(src), rows, C, _p(dst), 1 if dst.dtype == torch.bfloat16 else 0, W_, HW, bs, rs, cs, base, _stream())


def conv_pack_weights(w, fwd, bwd, Co, Ci, kh, kw, Cp, Cop):
    _call("scl_conv_pack_weights", _p(w), _p(fwd), _p(bwd), Co, Ci, kh, kw, Cp, Cop, _stream())


def conv_wgrad_finish(slabs, grad, nslab, Co, Ci, kh, kw, Cp, accumulate=True):
    _call("scl_conv_wgrad_finish", _p(slabs), _p(grad), nslab, Co, Ci, kh, kw, Cp, 1 if accumulate else 0, _stream())


def maxpool3_fwd(x, xs_h, xs_w, xs_b, H, W_, B, y, idx):
    _call("scl_maxpool3_fwd", _p(x), xs_h, xs_w, xs_b, H, W_, B, _p(y), _p(idx), _stream())


def maxpool3_bwd(dy, idx, H, W_, B, dx, xs_h, xs_w, xs_b):
    _call("scl_maxpool3_bwd", _p(dy), _p(idx), H, W_, B, _p(dx), xs_h, xs_w, xs_b, _stream())


def avgpool_fwd(x, B, R, C, y):
    _call("scl_avgpool_fwd", _p(x), B, R, C, _p(y), _stream())


def avgpool_bwd(dy, B, R, C, dx):
    _call("scl_avgpool_bwd", _p(dy), B, R, C, _p(dx), _stream())


# ---- Conformer block operators (csrc/conformer.hip) ---------------------------------------------------------------------------------
def swish_fwd(x, y, n):
    _call("scl_swish_fwd", _p(x), _p(y), n, _stream())


def swish_bwd(dy, x, dx, n):
    _call("scl_swish_bwd", _p(dy), _p(x), _p(dx), n, _stream())


def glu_fwd(x, y, M, C):
    _call("scl_glu_fwd", _p(x), _p(y), M, C, _stream())


def glu_bwd(dy, x, dx, M, C):
    _call("scl_glu_bwd", _p(dy), _p(x), _p(dx), M, C, _stream())


def axpby(a, b, sa, sb, out, n):
    _call("scl_axpby_f32", _p(a), _p(b), float(sa), float(sb), _p(out), n, _stream())


def dwconv1d_fwd(x, w, bias, y, B, n, C, k, pad_l, flip=False):
    _call("scl_dwconv1d_fwd", _p(x), _p(w), _p(bias), _p(y), B, n, C, k, pad_l, int(flip), _stream())


def dwconv1d_wgrad_nslabs(B, n):
    return L.load().scl_dwconv1d_wgrad_nslabs(B, n)


def dwconv1d_wgrad(x, dy, part, dw, db, B, n, C, k, pad_l):
    _call("scl_dwconv1d_wgrad", _p(x), _p(dy), _p(part), _p(dw), _p(db), B, n, C, k, pad_l, _stream())


def relpos_gather(E, Eu, n, Nr, D, max_pos):
    _call("scl_relpos_gather", _p(E), _p(Eu), n, Nr, D, max_pos, _stream())


def relpos_scatter_grad(dEu, dE, n, D, max_pos):
    _call("scl_relpos_scatter_grad", _p(dEu), _p(dE), n, D, max_pos, _stream())


def relpos_softmax_fwd(S, R, mask, P, B, H, n, ldS, ldR, ldP, scale):
    _call("scl_relpos_softmax_fwd", _p(S), _p(R), _p(mask), _p(P), B, H, n, ldS, ldR, ldP, float(scale), _stream())


def relpos_softmax_bwd(P, dP, mask, dS, dR, B, H, n, ldP, ldR, scale):
    _call("scl_relpos_softmax_bwd", _p(P), _p(dP), _p(mask), _p(dS), _p(dR), B, H, n, ldP, ldR, float(scale), _stream())


# ---- csrc/btse.hip ------------------------------------------------------------------------------------------------------------------------
def btse_bio_supported(bio_dim, n_heads, pf_dim, n_layers, window, bio_out, n_tokens):
    return bool(L.load().scl_btse_bio_supported(bio_dim, n_heads, pf_dim, n_layers, window, bio_out, n_tokens))


def btse_bio_ws_floats(n_layers, L_tokens):
    return L.load().scl_btse_bio_ws_floats(n_layers, L_tokens)


def btse_bio_fwd(desc):
    return _call("scl_btse_bio_fwd", ctypes.byref(desc), _stream(), keep=desc)


def btse_bio_bwd(desc):
    return _call("scl_btse_bio_bwd", ctypes.byref(desc), _stream(), keep=desc)


def btse_join_fwd(emb, s, W1, b1, b, B, C, bio_out, is_add):
    _call("scl_btse_join_fwd", _p(emb), _p(s), _p(W1), _p(b1), _p(b), B, C, bio_out, int(is_add), _stream())


def btse_join_bwd(db, emb, W1, demb, ds, dW1, db1, B, C, bio_out, is_add):
    _call("scl_btse_join_bwd", _p(db), _p(emb), _p(W1), _p(demb), _p(ds), _p(dW1), _p(db1), B, C, bio_out, int(is_add), _stream())
