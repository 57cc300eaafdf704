"""Tensor-level wrappers over the C ABI (torch is used for device memory and the stream only)."""
import ctypes

import torch

from . import lib as L

FLAT = L.FLAT


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t, offset=0):
    return t.data_ptr() + offset * t.element_size()


class Op:
    """One bf16 GEMM operand: element offset(row r, contiguous c) =
    (r // rpb) * rbstride + (r % rpb) * ld + (c // cin) * cout + c % cin (+ batch strides)."""

    __slots__ = ("t", "ld", "rpb", "rbstride", "cin", "cout", "bs1", "bs2", "offset")

    def __init__(self, t, ld, rpb=FLAT, rbstride=0, cin=FLAT, cout=0, bs1=0, bs2=0, offset=0):
        assert t.dtype == torch.bfloat16, "GEMM operands are bf16"
        self.t, self.ld, self.rpb, self.rbstride = t, ld, rpb, rbstride
        self.cin, self.cout, self.bs1, self.bs2, self.offset = cin, cout, bs1, bs2, offset

    def c(self):
        return L.SclOperand(_ptr(self.t, self.offset), self.bs1, self.bs2, self.rbstride, self.cout,
                            self.rpb, self.ld, self.cin, 0)


def gemm(A, B, C, M, N, K, *, a_t=False, b_t=False, bias=None, act=0, c2=None, R=None, rmode=0, ract=0,
         alpha=1.0, nb1=1, nb2=1, splitk=1, ldc=None, c_rpb=FLAT, c_rbstride=0, c_bs1=0, c_bs2=0,
         c_offset=0, bias_bs2=0, bias_offset=0, drop_p=0.0, drop_seed=0, c_split_stride=0):
    """C[z][m][n] = epilogue(alpha * sum_k A[z][m][k] B[z][n][k]) — see include/scl_hip.h."""
    lib = L.load()
    d = L.SclGemmDesc()
    d.A, d.B = A.c(), B.c()
    flags = 0
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
    flags |= (act << L.ACT_SHIFT) | (rmode << L.RMODE_SHIFT) | (ract << L.RACT_SHIFT)
    d.c_bs1, d.c_bs2, d.c_rbstride, d.c_split_stride, d.bias_bs2 = c_bs1, c_bs2, c_rbstride, c_split_stride, bias_bs2
    d.c_rpb, d.ldc = c_rpb, (N if ldc is None else ldc)
    d.M, d.N, d.K, d.nb1, d.nb2, d.splitk = M, N, K, nb1, nb2, splitk
    d.flags, d.alpha, d.drop_p, d.drop_seed = flags, alpha, drop_p, drop_seed
    L.check(lib.scl_gemm_bf16(ctypes.byref(d), _stream()), "scl_gemm_bf16")


def reduce_slabs(slabs, out, n, nslabs, stride):
    L.check(L.load().scl_reduce_slabs_f32(_ptr(slabs), _ptr(out), n, nslabs, stride, _stream()), "scl_reduce_slabs_f32")


def prof_enable(kid, on):
    L.check(L.load().scl_prof_enable(kid, 1 if on else 0), "scl_prof_enable")


def prof_read(kid):
    n, ms, fl = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
    L.check(L.load().scl_prof_read(kid, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl)), "scl_prof_read")
    return n.value, ms.value, fl.value
