"""Parity of the generic bf16 MFMA contraction (scl_gemm_bf16) against torch fp32 matmul on the
same bf16-rounded operands.  Tolerance: fp32 accumulation of bf16 products -> 2e-3 relative to the
row scale for fp32 outputs, one bf16 ulp (8e-3 rel) for bf16 outputs."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from scl_amd import ops  # noqa: E402
from scl_amd.lib import FLAT  # noqa: E402



def _has_experiments():
    return bool(ops.L.load().scl_build_flags() & 1)


# The opt-in GEMM experiments (256x128 ring, 256x256 ping-pong, two workgroups per CU, persistent blocks) are not part of the shipped
# library; their bit-identity tests run against a build with SCL_BUILD_DEFINES=-DSCL_EXPERIMENTS (see csrc/gemm.hip).
needs_experiments = pytest.mark.skipif("not _has_experiments()", reason="library built without -DSCL_EXPERIMENTS")

def _rand(shape, dev, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(torch.bfloat16).to(dev)


def _close(got, ref, tol, what=""):
    got, ref = got.float().cpu(), ref.float().cpu()
    denom = ref.abs().max().clamp_min(1e-6)
    err = ((got - ref).abs().max() / denom).item()
    assert err < tol, "%s: rel err %.3e >= %.1e" % (what, err, tol)


@pytest.mark.parametrize("a_t,b_t", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(300, 200, 136), (128, 128, 64), (257, 129, 199 + 1), (64, 72, 1000)])
def test_layouts(dev, a_t, b_t, M, N, K):
    # storage is padded to multiples of 8 along the contiguous dim
    pad8 = lambda v: (v + 7) // 8 * 8
    A = _rand((M, K), dev, 1)
    B = _rand((N, K), dev, 2)
    ref = A.float() @ B.float().t()
    if a_t:
        ldA = pad8(M)
        At = torch.zeros(K, ldA, dtype=torch.bfloat16, device=dev)
        At[:, :M] = A.t()
        opA = ops.Op(At, ldA)
    else:
        ldA = pad8(K)
        Ap = torch.zeros(M, ldA, dtype=torch.bfloat16, device=dev)
        Ap[:, :K] = A
        opA = ops.Op(Ap, ldA)
    if b_t:
        ldB = pad8(N)
        Bt = torch.zeros(K, ldB, dtype=torch.bfloat16, device=dev)
        Bt[:, :N] = B.t()
        opB = ops.Op(Bt, ldB)
    else:
        ldB = pad8(K)
        Bp = torch.zeros(N, ldB, dtype=torch.bfloat16, device=dev)
        Bp[:, :K] = B
        opB = ops.Op(Bp, ldB)
    C = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev)
    ops.gemm(opA, opB, C, M, N, K, a_t=a_t, b_t=b_t)
    torch.cuda.synchronize()
    _close(C, ref, 2e-3, "layout a_t=%s b_t=%s" % (a_t, b_t))


@pytest.mark.parametrize("a_t,b_t", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(304, 200, 136), (128, 128, 64), (264, 136, 200), (72, 64, 1000), (1000, 520, 264)])
def test_lds_dma_staging_equals_register_staging(dev, a_t, b_t, M, N, K):
    """All dims % 8 == 0 -> the LDS-DMA kernel is selected; it must agree BIT FOR BIT with the register-staged
    kernel (same LDS image, same MFMA order) and with the fp32 reference; memory around the operands is NaN."""
    A = _rand((M, K), dev, 21); B = _rand((N, K), dev, 22)
    ref = A.float() @ B.float().t()
    def padded(mat, rows, cols):
        buf = torch.full((rows + 3, cols + 8), float("nan"), dtype=torch.bfloat16, device=dev)
        buf[:rows, :cols] = mat
        return buf, cols + 8
    bufA, ldA = padded(A.t().contiguous(), K, M) if a_t else padded(A, M, K)
    bufB, ldB = padded(B.t().contiguous(), K, N) if b_t else padded(B, N, K)
    outs = []
    for no_dma in (False, True):
        C = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev)
        ops.gemm(ops.Op(bufA, ldA), ops.Op(bufB, ldB), C, M, N, K, a_t=a_t, b_t=b_t, no_dma=no_dma)
        outs.append(C)
    assert torch.equal(outs[0], outs[1])
    _close(outs[0], ref, 2e-3, "dma a_t=%s b_t=%s" % (a_t, b_t))


@needs_experiments
@pytest.mark.parametrize("a_t,b_t", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(5128, 1280, 264), (6368, 1024, 1024), (2600, 2560, 192)])
def test_big_tile_ring_kernel_equals_128_tile_kernel(dev, a_t, b_t, M, N, K):
    """The opt-in 256x128 / 3-stage LDS-DMA ring kernel: every output element sums K in the same order as in the 128x128
    kernels, so the results must be bit-identical."""
    A = _rand((M, K), dev, 31, 0.3); B = _rand((N, K), dev, 32, 0.3)
    opA = ops.Op(A.t().contiguous(), M) if a_t else ops.Op(A, K)
    opB = ops.Op(B.t().contiguous(), N) if b_t else ops.Op(B, K)
    bias = torch.randn(N, device=dev)
    outs = []
    for force_big in (True, False):
        C = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
        ops.gemm(opA, opB, C, M, N, K, a_t=a_t, b_t=b_t, bias=bias, act=1, force_big=force_big, no_w8=True)
        outs.append(C)
    assert torch.equal(outs[0], outs[1])
    _close(outs[0], torch.nn.functional.gelu(A.float() @ B.float().t() + bias), 8e-3, "big a_t=%s b_t=%s" % (a_t, b_t))


@needs_experiments
@pytest.mark.parametrize("a_t,b_t", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K,splitk", [(5128, 1288, 264, 1), (6368, 1024, 1024, 1), (520, 776, 4096, 3), (256, 256, 64, 1), (1000, 3072, 512, 1)])
def test_pingpong_256_tile_kernel_equals_128_tile_kernel(dev, a_t, b_t, M, N, K, splitk):
    """The opt-in 256x256 two-wave-group kernel (raw barriers, counted vmcnt, LDS-DMA in flight across ds_reads): same K order
    per output element as the 128x128 kernels => bit-identical, on ragged M / N edges, with split-K slabs, repeated to give a
    mis-ordered LDS read a chance to show."""
    A = _rand((M, K), dev, 41, 0.3); B = _rand((N, K), dev, 42, 0.3)
    opA = ops.Op(A.t().contiguous(), M) if a_t else ops.Op(A, K)
    opB = ops.Op(B.t().contiguous(), N) if b_t else ops.Op(B, K)
    outs = []
    for kw in (dict(no_p8=True, no_big=True, no_w8=True), dict(force_p8=True), dict(force_p8=True), dict(force_p8=True)):
        if splitk > 1:
            C = torch.full((splitk, M, N), float("nan"), dtype=torch.float32, device=dev)
            ops.gemm(opA, opB, C, M, N, K, a_t=a_t, b_t=b_t, splitk=splitk, c_split_stride=M * N, **kw)
        else:
            C = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
            ops.gemm(opA, opB, C, M, N, K, a_t=a_t, b_t=b_t, act=2, **kw)
        outs.append(C)
    for o in outs[1:]:
        assert torch.equal(outs[0], o)
    ref = A.float() @ B.float().t()
    got = outs[1].sum(0) if splitk > 1 else outs[1]
    _close(got, ref if splitk > 1 else torch.relu(ref), 8e-3, "p8 a_t=%s b_t=%s" % (a_t, b_t))


def test_kcontig_tail_needs_no_zero_padding(dev):
    # K = 199: the last 16-byte vector is partially valid; garbage (NaN) beyond K must be masked
    M, N, K = 70, 90, 199
    A = _rand((M, K), dev, 3); B = _rand((N, K), dev, 4)
    Ap = torch.full((M, 208), float("nan"), dtype=torch.bfloat16, device=dev); Ap[:, :K] = A
    Bp = torch.full((N, 208), float("nan"), dtype=torch.bfloat16, device=dev); Bp[:, :K] = B
    C = torch.empty(M, N, dtype=torch.float32, device=dev)
    ops.gemm(ops.Op(Ap, 208), ops.Op(Bp, 208), C, M, N, K)
    _close(C, A.float() @ B.float().t(), 2e-3, "k tail")
    # transposed operands: rows >= K and cols >= M/N hold NaN and must not leak
    At = torch.full((K + 9, 72), float("nan"), dtype=torch.bfloat16, device=dev); At[:K, :M] = A.t()
    Bt = torch.full((K + 9, 96), float("nan"), dtype=torch.bfloat16, device=dev); Bt[:K, :N] = B.t()
    C2 = torch.empty(M, N, dtype=torch.float32, device=dev)
    ops.gemm(ops.Op(At, 72), ops.Op(Bt, 96), C2, M, N, K, a_t=True, b_t=True)
    _close(C2, A.float() @ B.float().t(), 2e-3, "t tail")


def test_epilogues(dev):
    M, N, K = 200, 264, 320
    A = _rand((M, K), dev, 5, 0.2); B = _rand((N, K), dev, 6, 0.2)
    bias = torch.randn(N, device=dev)
    lin = A.float() @ B.float().t() + bias
    # bias + gelu with pre-activation second output (bf16)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    C2 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ops.gemm(ops.Op(A, K), ops.Op(B, K), C, M, N, K, bias=bias, act=1, c2=C2)
    _close(C2, lin, 8e-3, "pre-act")
    _close(C, torch.nn.functional.gelu(lin), 8e-3, "gelu")
    # bias + f32 residual -> f32
    R = torch.randn(M, N, device=dev)
    C = torch.empty(M, N, dtype=torch.float32, device=dev)
    ops.gemm(ops.Op(A, K), ops.Op(B, K), C, M, N, K, bias=bias, R=R, rmode=1)
    _close(C, lin + R, 2e-3, "residual")
    # alpha and grad-multiply by gelu'(H) with H bf16
    H = _rand((M, N), dev, 7)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ops.gemm(ops.Op(A, K), ops.Op(B, K), C, M, N, K, alpha=0.5, R=H, rmode=2, ract=1)
    h = H.float().requires_grad_(True)
    torch.nn.functional.gelu(h).sum().backward()
    _close(C, 0.5 * (A.float() @ B.float().t()) * h.grad, 8e-3, "gelu-grad-mul")
    # relu / leaky
    for act, fn in ((2, torch.relu), (3, lambda t: torch.nn.functional.leaky_relu(t, 0.01))):
        C = torch.empty(M, N, dtype=torch.float32, device=dev)
        ops.gemm(ops.Op(A, K), ops.Op(B, K), C, M, N, K, bias=bias, act=act)
        _close(C, fn(lin), 2e-3, "act %d" % act)


def test_dropout_mask_is_reproducible_and_unbiased(dev):
    M, N, K = 256, 256, 64
    A = torch.ones(M, K, dtype=torch.bfloat16, device=dev); B = torch.ones(N, K, dtype=torch.bfloat16, device=dev)
    C1 = torch.empty(M, N, dtype=torch.float32, device=dev); C2 = torch.empty_like(C1)
    ops.gemm(ops.Op(A, K), ops.Op(B, K), C1, M, N, K, drop_p=0.5, drop_seed=123)
    ops.gemm(ops.Op(A, K), ops.Op(B, K), C2, M, N, K, drop_p=0.5, drop_seed=123)
    assert torch.equal(C1, C2)
    vals = set(C1.unique().tolist())
    assert vals == {0.0, 128.0}
    keep = (C1 > 0).float().mean().item()
    assert abs(keep - 0.5) < 0.02


def test_conv_rows_and_split_k_index(dev):
    # conv1d k=3, stride 2, channels-last == GEMM with overlapping rows (ld = stride*C < K)
    Bsz, Tin, Cc, Co, k, s = 3, 41, 32, 48, 3, 2
    Tout = (Tin - k) // s + 1
    x = _rand((Bsz, Tin, Cc), dev, 8)
    w = _rand((Co, Cc, k), dev, 9, 0.2)   # torch layout [co, ci, j]
    ref = torch.nn.functional.conv1d(x.float().transpose(1, 2), w.float(), stride=s).transpose(1, 2)  # [B,Tout,Co]
    wk = w.permute(0, 2, 1).contiguous().view(Co, k * Cc)  # [co, j*C + ci]
    xpad = torch.zeros(Bsz * Tin * Cc + 64, dtype=torch.bfloat16, device=dev); xpad[:x.numel()] = x.flatten()
    C = torch.empty(Bsz * Tout, Co, dtype=torch.float32, device=dev)
    ops.gemm(ops.Op(xpad, s * Cc, rpb=Tout, rbstride=Tin * Cc), ops.Op(wk, k * Cc), C, Bsz * Tout, Co, k * Cc)
    _close(C.view(Bsz, Tout, Co), ref, 2e-3, "conv rows")
    # the same contraction as a wgrad: dW[co][j*C+ci] = sum_{b,t} dy[b,t,co] * x[b, s*t+j, ci]
    dy = _rand((Bsz * Tout, Co), dev, 10)
    xf = x.float().transpose(1, 2).requires_grad_(False)
    wv = w.float().clone().requires_grad_(True)
    y = torch.nn.functional.conv1d(xf, wv, stride=s).transpose(1, 2).reshape(Bsz * Tout, Co)
    (y * dy.float()).sum().backward()
    dw_ref = wv.grad.permute(0, 2, 1).reshape(Co, k * Cc)
    dW = torch.empty(Co, k * Cc, dtype=torch.float32, device=dev)
    ops.gemm(ops.Op(dy, Co), ops.Op(xpad, s * Cc, rpb=Tout, rbstride=Tin * Cc), dW, Co, k * Cc, Bsz * Tout,
             a_t=True, b_t=True)
    _close(dW, dw_ref, 2e-3, "conv wgrad")
    # 2-level contiguous index (grouped positional conv): k = (j, ci) -> j*ldx + ci, groups via batch
    G, Cg, kk, T = 2, 16, 8, 37
    Ctot = G * Cg
    xp = _rand((Bsz, T + kk, Ctot), dev, 11)
    wg = _rand((Ctot, Cg, kk), dev, 12, 0.3)   # [co_total, ci, j]
    ref = torch.nn.functional.conv1d(xp.float().transpose(1, 2), wg.float(), groups=G).transpose(1, 2)[:, :T]
    wgk = wg.view(G, Cg, Cg, kk).permute(0, 1, 3, 2).contiguous()  # [g][co][j][ci]
    out = torch.zeros(Bsz * T, Ctot, dtype=torch.float32, device=dev)
    ops.gemm(ops.Op(xp, Ctot, rpb=T, rbstride=(T + kk) * Ctot, cin=Cg, cout=Ctot, bs2=Cg),
             ops.Op(wgk, kk * Cg, bs2=Cg * kk * Cg), out, Bsz * T, Cg, kk * Cg,
             nb2=G, ldc=Ctot, c_bs2=Cg)
    _close(out.view(Bsz, T, Ctot), ref, 2e-3, "grouped conv")


def test_batched_attention_shapes_and_splitk(dev):
    Bsz, H, T, D = 2, 3, 199, 64
    qkv = _rand((Bsz, T, 3, H, D), dev, 13, 0.5)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]   # [B,T,H,D]
    ld = 3 * H * D
    S = torch.empty(Bsz, H, T, T, dtype=torch.float32, device=dev)
    ops.gemm(ops.Op(qkv, ld, bs1=T * ld, bs2=D), ops.Op(qkv, ld, bs1=T * ld, bs2=D, offset=H * D), S, T, T, D,
             nb1=Bsz, nb2=H, alpha=0.125, c_bs1=H * T * T, c_bs2=T * T)
    ref = torch.einsum("bihd,bjhd->bhij", q.float(), k.float()) * 0.125
    _close(S, ref, 2e-3, "QK^T")
    Tp = 208
    P = torch.zeros(Bsz, H, T, Tp, dtype=torch.bfloat16, device=dev)
    P[..., :T] = torch.softmax(ref, -1).to(torch.bfloat16)
    O = torch.zeros(Bsz, T, H * D, dtype=torch.bfloat16, device=dev)
    ops.gemm(ops.Op(P, Tp, bs1=H * T * Tp, bs2=T * Tp), ops.Op(qkv, ld, bs1=T * ld, bs2=D, offset=2 * H * D),
             O, T, D, T, b_t=True, nb1=Bsz, nb2=H, ldc=H * D, c_bs1=T * H * D, c_bs2=D)
    refO = torch.einsum("bhij,bjhd->bihd", P[..., :T].float(), v.float()).reshape(Bsz, T, H * D)
    _close(O, refO, 8e-3, "PV")
    # split-K slabs + deterministic reduce (wgrad shape: long reduction, small output)
    M, N, K = 256, 128, 6368
    A = _rand((K, M), dev, 14, 0.1); Bm = _rand((K, N), dev, 15, 0.1)
    slabs = torch.empty(4, M, N, dtype=torch.float32, device=dev)
    ops.gemm(ops.Op(A, M), ops.Op(Bm, N), slabs, M, N, K, a_t=True, b_t=True, splitk=4, c_split_stride=M * N)
    out = torch.empty(M, N, dtype=torch.float32, device=dev)
    ops.reduce_slabs(slabs, out, M * N, 4, M * N)
    _close(out, A.float().t() @ Bm.float(), 2e-3, "split-K")


def test_gemm_speed_report(dev, capsys):
    M, N, K = 6368, 4096, 1024
    A = _rand((M, K), dev, 16, 0.1); B = _rand((N, K), dev, 17, 0.1)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for name, kw, a, b in (("NT", {}, ops.Op(A, K), ops.Op(B, K)),):
        for _ in range(3):
            ops.gemm(a, b, C, M, N, K, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm(a, b, C, M, N, K, **kw)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        with capsys.disabled():
            print("\n[gemm %s %dx%dx%d] %.3f ms  %.1f TFLOP/s" % (name, M, N, K, ms, 2 * M * N * K / ms / 1e9))


@pytest.mark.parametrize("a_t,b_t", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K,splitk", [(12736, 1024, 1024, 1), (5128, 1288, 256, 1), (2184, 3072, 1024, 1), (1024, 4096, 12736, 4),
                                          (520, 776, 4096, 3), (200, 256, 64, 1), (1000, 200, 512, 1)])
@pytest.mark.parametrize("loop", ["1", "0", "1-no112"])
def test_wide_tile_kernel_equals_128_tile_kernel(dev, monkeypatch, a_t, b_t, M, N, K, splitk, loop):
    """The wide-tile kernels (gemm_w8.hip: runtime row pitch, 7+6 / 8+8 row blocks per wave row, scalar-offset K advance), both K
    loops (SCL_W8_MODE 1 = single barrier with counted LDS waits, the default for every layout; 0 = two-barrier ping-pong): same K
    order per output element as the 128x128 kernels => bit-identical, on ragged M / N edges, row pitches that are not multiples of
    16, split-K slabs with an uneven last slab, repeated to give a mis-ordered LDS read a chance to show.
    Memory behind the operands is NaN: a fetch past a row limit that is not range-checked to zero would poison the output."""
    # round 6: with K-contiguous A the plan may take 112-row tiles (4 + 3 row blocks) for shapes that fill the CUs better that way
    # ((2184, 3072), (200, 256), (1000, 200) here); "1-no112" keeps those shapes on the 208- / 256-row tiles they were written for
    monkeypatch.setenv("SCL_W8_MODE", loop[0])
    monkeypatch.setenv("SCL_W8_TILE112", "0" if loop.endswith("no112") else "1")
    A = _rand((M, K), dev, 51, 0.3); B = _rand((N, K), dev, 52, 0.3)
    def nanpad(mat):
        buf = torch.full((mat.numel() + 4096,), float("nan"), dtype=torch.bfloat16, device=dev)
        buf[:mat.numel()] = mat.reshape(-1)
        return buf
    opA = ops.Op(nanpad(A.t().contiguous()), M) if a_t else ops.Op(nanpad(A), K)
    opB = ops.Op(nanpad(B.t().contiguous()), N) if b_t else ops.Op(nanpad(B), K)
    outs = []
    for kw in (dict(no_p8=True, no_big=True, no_w8=True), dict(force_w8=True), dict(force_w8=True), dict(force_w8=True)):
        if splitk > 1:
            C = torch.full((splitk, M, N), float("nan"), dtype=torch.float32, device=dev)
            ops.gemm(opA, opB, C, M, N, K, a_t=a_t, b_t=b_t, splitk=splitk, c_split_stride=M * N, **kw)
        else:
            C = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
            ops.gemm(opA, opB, C, M, N, K, a_t=a_t, b_t=b_t, act=2, **kw)
        outs.append(C)
    for o in outs[1:]:
        assert torch.equal(outs[0], o)
    ref = A.float() @ B.float().t()
    got = outs[1].sum(0) if splitk > 1 else outs[1]
    _close(got, ref if splitk > 1 else torch.relu(ref), 8e-3, "w8 a_t=%s b_t=%s" % (a_t, b_t))


@pytest.mark.parametrize("a_t,b_t", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K,splitk", [(12736, 1024, 1024, 1), (5128, 1288, 96, 1), (2184, 3072, 1024, 1), (1024, 4096, 12736, 4),
                                          (520, 776, 4128, 3), (200, 256, 32, 1), (1000, 200, 544, 1), (208, 128, 64, 1)])
@needs_experiments
def test_two_blocks_per_cu_kernel_equals_128_tile_kernel(dev, a_t, b_t, M, N, K, splitk):
    """gemm_x2.hip (208 x 128 tiles, 4-wave workgroups, two per CU, K step 32, rings of three stages): same K order per output
    element as the 128 x 128 kernels => bit-identical, on ragged M / N edges, K that is a multiple of 32 but not of 64, fewer K steps
    than ring stages, uneven split-K slabs; operands are followed by NaN, so a fetch that is not range-checked shows."""
    A = _rand((M, K), dev, 71, 0.3); B = _rand((N, K), dev, 72, 0.3)
    def nanpad(mat):
        buf = torch.full((mat.numel() + 4096,), float("nan"), dtype=torch.bfloat16, device=dev)
        buf[:mat.numel()] = mat.reshape(-1)
        return buf
    opA = ops.Op(nanpad(A.t().contiguous()), M) if a_t else ops.Op(nanpad(A), K)
    opB = ops.Op(nanpad(B.t().contiguous()), N) if b_t else ops.Op(nanpad(B), K)
    outs = []
    for kw in (dict(no_p8=True, no_big=True, no_w8=True, no_x2=True), dict(force_x2=True), dict(force_x2=True), dict(force_x2=True)):
        if splitk > 1:
            C = torch.full((splitk, M, N), float("nan"), dtype=torch.float32, device=dev)
            ops.gemm(opA, opB, C, M, N, K, a_t=a_t, b_t=b_t, splitk=splitk, c_split_stride=M * N, **kw)
        else:
            C = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
            ops.gemm(opA, opB, C, M, N, K, a_t=a_t, b_t=b_t, act=2, **kw)
        outs.append(C)
    d = ops._gemm_desc(opA, opB, outs[-1], M, N, K, a_t=a_t, b_t=b_t, splitk=splitk, c_split_stride=M * N if splitk > 1 else 0, force_x2=True)
    assert ops.L.load().scl_gemm_uses_wide_tiles(ops.ctypes.byref(d)) == 3
    for o in outs[1:]:
        assert torch.equal(outs[0], o)
    ref = A.float() @ B.float().t()
    got = outs[1].sum(0) if splitk > 1 else outs[1]
    _close(got, ref if splitk > 1 else torch.relu(ref), 8e-3, "x2 a_t=%s b_t=%s" % (a_t, b_t))


@needs_experiments
def test_two_blocks_per_cu_kernel_epilogues_and_conv_rows(dev):
    """Every fused epilogue of the encoder (bias + GELU + second output; x gelu'(R); f32 residual; dropout) and the utterance-batched
    overlapping rows of a conv layer through gemm_x2.hip against the 128 x 128 kernel, bit for bit."""
    M, N, K = 3 * 199 + 57, 1024, 512
    A = _rand((M, K), dev, 81, 0.3); W = _rand((N, K), dev, 82, 0.05); Wt = W.t().contiguous()
    bias = torch.randn(N, device=dev); Rb = _rand((M, N), dev, 83, 1.0); Rf = torch.randn(M, N, device=dev)
    cases = [
        (dict(bias=bias, act=1), torch.bfloat16, True),                      # fc1 forward: GELU + pre-activation copy
        (dict(R=Rb, rmode=2, ract=1), torch.bfloat16, False),                # fc2 data gradient
        (dict(bias=bias, R=Rf, rmode=1), torch.float32, False),              # out-proj / fc2 forward: f32 residual stream
        (dict(bias=bias, act=3, drop_p=0.5, drop_seed=1234), torch.bfloat16, True),
        (dict(bias=bias, act=5, drop_p=0.25, drop_seed=77), torch.bfloat16, True),   # fc1 forward, round 3: second output = gelu'(pre-activation)
        (dict(R=Rb, rmode=2, ract=4), torch.bfloat16, False),                # fc2 data gradient, round 3: x stored derivative
    ]
    for kw, cdt, c2 in cases:
        for b_t, opB in ((False, ops.Op(W, K)), (True, ops.Op(Wt, N))):
            outs = []
            for sel in (dict(no_w8=True, no_p8=True, no_big=True, no_x2=True), dict(force_x2=True)):
                C = torch.full((M, N), float("nan"), dtype=cdt, device=dev)
                C2 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev) if c2 else None
                ops.gemm(ops.Op(A, K), opB, C, M, N, K, b_t=b_t, c2=C2, **kw, **sel)
                outs.append((C, C2))
            assert torch.equal(outs[0][0], outs[1][0]), kw.keys()
            if c2:
                assert torch.equal(outs[0][1], outs[1][1]), kw.keys()
    Bz, Tin, C_, k, s = 6, 1601, 512, 3, 2
    Tout = (Tin - k) // s + 1
    z = _rand((Bz * Tin * C_ + 65536,), dev, 61, 0.3); wk = _rand((C_, k * C_), dev, 62, 0.05)
    cb = torch.randn(C_, device=dev)
    outs = []
    for sel in (dict(no_w8=True, no_p8=True, no_big=True, no_x2=True), dict(force_x2=True)):
        y = torch.full((Bz * Tout, C_), float("nan"), dtype=torch.float32, device=dev)
        ops.gemm(ops.Op(z, s * C_, rpb=Tout, rbstride=Tin * C_), ops.Op(wk, k * C_), y, Bz * Tout, C_, k * C_, bias=cb, **sel)
        outs.append(y)
    assert torch.equal(outs[0], outs[1])


def _persistent_launches():
    return ops.L.load().scl_debug_gemm_persistent_launches()


@pytest.mark.parametrize("a_t,b_t", [(False, False), (False, True), (True, False)])
@pytest.mark.parametrize("M,N,K,blocks", [(12736, 4096, 1024, "1"), (12736, 3072, 1024, "1"), (2184, 3072, 1024, "8"), (5128, 1288, 192, "16"),
                                          (3000, 1000, 320, "8"), (5 * 208 + 8, 2 * 256 + 40, 448, "8"), (6400 * 2, 512, 1536, "24")])
@needs_experiments
def test_persistent_wide_kernel_equals_128_tile_kernel(dev, monkeypatch, a_t, b_t, M, N, K, blocks):
    """gemm_w8.hip's persistent blocks (w8p): every resident block walks several tiles, requests the next tile's first K stage two K
    steps before its epilogue and the second one from inside its own epilogue regions.  Same tiles and K order as the one-tile
    blocks => bit-identical to the 128 x 128 kernel: the encoder's multi-round shapes with the automatic grid, and small problems
    with 8-24 resident blocks (many tiles per block, a ragged last round, 3 K steps = the minimum, an odd number of K steps = the
    B images swap roles from tile to tile, ragged M / N edges); operands are followed by NaN so an unchecked fetch shows."""
    A = _rand((M, K), dev, 171, 0.3); B = _rand((N, K), dev, 172, 0.3)
    def nanpad(mat):
        buf = torch.full((mat.numel() + 4096,), float("nan"), dtype=torch.bfloat16, device=dev)
        buf[:mat.numel()] = mat.reshape(-1)
        return buf
    opA = ops.Op(nanpad(A.t().contiguous()), M) if a_t else ops.Op(nanpad(A), K)
    opB = ops.Op(nanpad(B.t().contiguous()), N) if b_t else ops.Op(nanpad(B), K)
    outs = []
    for persist, kw in (("0", dict(no_p8=True, no_big=True, no_w8=True, no_x2=True)), ("0", dict(force_w8=True)), (blocks, dict(force_w8=True)),
                        (blocks, dict(force_w8=True))):
        monkeypatch.setenv("SCL_GEMM_PERSIST", persist)
        n0 = _persistent_launches()
        C = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
        ops.gemm(opA, opB, C, M, N, K, a_t=a_t, b_t=b_t, act=2, **kw)
        assert _persistent_launches() - n0 == (0 if persist == "0" else 1), (persist, kw)
        outs.append(C)
    for o in outs[1:]:
        assert torch.equal(outs[0], o)
    _close(outs[2], torch.relu(A.float() @ B.float().t()), 8e-3, "w8p a_t=%s b_t=%s" % (a_t, b_t))


@needs_experiments
def test_persistent_wide_kernel_epilogues_column_sums_and_conv_rows(dev, monkeypatch):
    """Every fused epilogue of the encoder through the persistent blocks (two 16-row blocks per pass, R staged in the free B image)
    against the 128 x 128 kernel, bit for bit; the per-tile column sums must equal the one-tile kernel's partial rows bit for bit
    (same rows in the same order per lane); utterance-batched overlapping conv rows."""
    M, N, K = 9 * 199 + 57, 1024, 512
    A = _rand((M, K), dev, 181, 0.3); W = _rand((N, K), dev, 182, 0.05); Wt = W.t().contiguous()
    bias = torch.randn(N, device=dev); Rb = _rand((M, N), dev, 183, 1.0); Rf = torch.randn(M, N, device=dev)
    cases = [
        (dict(bias=bias, act=1), torch.bfloat16, True),                      # fc1 forward: GELU + pre-activation copy
        (dict(R=Rb, rmode=2, ract=1), torch.bfloat16, False),                # fc2 data gradient
        (dict(bias=bias, R=Rf, rmode=1), torch.float32, False),              # out-proj / fc2 forward: f32 residual stream
        (dict(bias=bias, R=Rb, rmode=1), torch.float32, False),
        (dict(R=Rf, rmode=2, ract=1), torch.bfloat16, False),
        (dict(bias=bias, act=3, drop_p=0.5, drop_seed=1234), torch.bfloat16, True),
        (dict(bias=bias, act=5), torch.bfloat16, True),
        (dict(R=Rb, rmode=2, ract=4), torch.bfloat16, False),
    ]
    for kw, cdt, c2 in cases:
        for b_t, opB in ((False, ops.Op(W, K)), (True, ops.Op(Wt, N))):
            outs = []
            for persist, sel in (("0", dict(no_w8=True, no_p8=True, no_big=True, no_x2=True)), ("8", dict(force_w8=True))):
                monkeypatch.setenv("SCL_GEMM_PERSIST", persist)
                n0 = _persistent_launches()
                C = torch.full((M, N), float("nan"), dtype=cdt, device=dev)
                C2 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev) if c2 else None
                ops.gemm(ops.Op(A, K), opB, C, M, N, K, b_t=b_t, c2=C2, **kw, **sel)
                assert _persistent_launches() - n0 == int(persist != "0")
                outs.append((C, C2))
            assert torch.equal(outs[0][0], outs[1][0]), kw.keys()
            if c2:
                assert torch.equal(outs[0][1], outs[1][1]), kw.keys()
    # column sums: partial rows of the persistent blocks == partial rows of the one-tile blocks
    kw = dict(b_t=True, R=Rb, rmode=2, ract=1, force_w8=True)
    parts = []
    for persist in ("0", "8"):
        monkeypatch.setenv("SCL_GEMM_PERSIST", persist)
        C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        rows = ops.gemm_colsum_rows(ops.Op(A, K), ops.Op(Wt, N), C, M, N, K, **kw)
        part = torch.full((rows, N), float("nan"), device=dev)
        ops.gemm(ops.Op(A, K), ops.Op(Wt, N), C, M, N, K, colsum_part=part, **kw)
        parts.append((C, part))
    assert torch.equal(parts[0][0], parts[1][0]) and torch.equal(parts[0][1], parts[1][1]) and torch.isfinite(parts[1][1]).all()
    Bz, Tin, C_, k, s = 6, 1601, 512, 3, 2
    Tout = (Tin - k) // s + 1
    z = _rand((Bz * Tin * C_ + 65536,), dev, 61, 0.3); wk = _rand((C_, k * C_), dev, 62, 0.05)
    cb = torch.randn(C_, device=dev)
    outs = []
    for persist, sel in (("0", dict(no_w8=True, no_p8=True, no_big=True, no_x2=True)), ("8", dict(force_w8=True))):
        monkeypatch.setenv("SCL_GEMM_PERSIST", persist)
        y = torch.full((Bz * Tout, C_), float("nan"), dtype=torch.float32, device=dev)
        ops.gemm(ops.Op(z, s * C_, rpb=Tout, rbstride=Tin * C_), ops.Op(wk, k * C_), y, Bz * Tout, C_, k * C_, bias=cb, **sel)
        outs.append(y)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("M,N,K,b_t", [(12736, 4096, 1024, True), (1000, 520, 256, False), (3 * 206 + 5, 1024, 512, True)])
def test_wide_tile_epilogue_column_sums(dev, M, N, K, b_t):
    """SclGemmDesc.colsum_part: the wide tiles also write per-tile column sums of the f32 values they store (the bias gradient of the
    Linear whose output gradient this GEMM produces).  Against the column sum of an f32-output run of the same GEMM (1e-5 of the
    column's absolute sum: only the summation order differs), with the x gelu'(R) epilogue of the fc2 data gradient and a row count
    that leaves the last tile ragged; the bf16 output itself must not change, and a launch that cannot honour the request refuses."""
    A = _rand((M, K), dev, 91, 0.3); W = _rand((N, K), dev, 92, 0.05)
    opB = ops.Op(W.t().contiguous(), N) if b_t else ops.Op(W, K)
    R = _rand((M, N), dev, 93, 1.0)
    kw = dict(b_t=b_t, R=R, rmode=2, ract=1, force_w8=M < 12000)      # the small cases would otherwise go to the 128 x 128 tiles
    C0 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ops.gemm(ops.Op(A, K), opB, C0, M, N, K, **kw)
    rows = ops.gemm_colsum_rows(ops.Op(A, K), opB, C0, M, N, K, **kw)
    assert rows > 0 and rows % 4 == 0
    part = torch.full((rows, N), float("nan"), device=dev)
    C1 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ops.gemm(ops.Op(A, K), opB, C1, M, N, K, colsum_part=part, **kw)
    assert torch.equal(C0, C1)
    Cf = torch.empty(M, N, dtype=torch.float32, device=dev)
    ops.gemm(ops.Op(A, K), opB, Cf, M, N, K, **kw)
    out = torch.empty(N, device=dev)
    ops.colreduce(part, out, rows, N)
    assert torch.isfinite(part).all()
    ref = Cf.double().sum(0)
    scale = Cf.double().abs().sum(0)
    assert ((out.double() - ref).abs() / scale).max().item() < 1e-5
    assert (out - part.sum(0)).abs().max().item() <= 1e-4 * scale.max().item()
    from scl_amd.lib import SclError
    with pytest.raises(SclError):          # 128 x 128 tiles cannot: refused, not silently ignored
        ops.gemm(ops.Op(A, K), opB, C1, M, N, K, colsum_part=part, no_w8=True, **kw)


def test_wide_tile_kernel_conv_and_grouped_addressing(dev):
    """Operands the wide kernel must address exactly as the 128x128 kernels do: utterance-batched overlapping rows (conv layer as a
    GEMM: rpb / rbstride / ld < K), a strided utterance-batched output (c_rpb / c_rbstride / ldc, the phase-split transposed
    convolution), and the 2-level contiguous index of the grouped positional conv (cin = 64) with batch strides, bias, GELU, second
    output and residual."""
    Bz, Tin, C, k, s = 6, 1601, 512, 3, 2
    Tout = (Tin - k) // s + 1
    z = _rand((Bz * Tin * C + 65536,), dev, 61, 0.3); wk = _rand((C, k * C), dev, 62, 0.05)
    bias = torch.randn(C, device=dev)
    outs = []
    for kw in (dict(no_w8=True, no_p8=True, no_big=True), dict(force_w8=True)):
        y = torch.full((Bz * Tout, C), float("nan"), dtype=torch.float32, device=dev)
        ops.gemm(ops.Op(z, s * C, rpb=Tout, rbstride=Tin * C), ops.Op(wk, k * C), y, Bz * Tout, C, k * C, bias=bias, **kw)
        outs.append(y)
    assert torch.equal(outs[0], outs[1])
    # strided, utterance-batched output with a transposed weight operand
    dy = _rand((Bz * (Tout + 2) * C + 65536,), dev, 63, 0.3); wd = _rand((2 * C, C), dev, 64, 0.05)
    outs = []
    for kw in (dict(no_w8=True, no_p8=True, no_big=True), dict(force_w8=True)):
        dz = torch.zeros(Bz * Tin * C + 65536, dtype=torch.bfloat16, device=dev)
        ops.gemm(ops.Op(dy, C, rpb=Tout, rbstride=(Tout + 2) * C), ops.Op(wd, C), dz, Bz * Tout, C, 2 * C, b_t=True, ldc=s * C,
                 c_rpb=Tout, c_rbstride=Tin * C, c_offset=C, **kw)
        outs.append(dz)
    assert torch.equal(outs[0], outs[1])
    # grouped positional conv: 16 groups, N = 64 per group, k index = (tap, channel-in-group)
    Bz, T, E, K, G = 8, 199, 1024, 128, 16
    Cg = E // G
    xpad = _rand((Bz * (T + K) * E + 65536,), dev, 65, 0.3); wf = _rand((G, Cg, K * Cg), dev, 66, 0.02)
    x0 = torch.randn(Bz * T, E, device=dev); pbias = torch.randn(E, device=dev)
    outs = []
    for kw in (dict(no_w8=True, no_p8=True, no_big=True), dict(force_w8=True)):
        xo = torch.full((Bz * T, E), float("nan"), dtype=torch.float32, device=dev)
        pre = torch.full((Bz * T, E), float("nan"), dtype=torch.bfloat16, device=dev)
        ops.gemm(ops.Op(xpad, E, rpb=T, rbstride=(T + K) * E, cin=Cg, cout=E, bs2=Cg), ops.Op(wf, K * Cg, bs2=Cg * K * Cg), xo, Bz * T, Cg,
                 K * Cg, nb2=G, ldc=E, c_bs2=Cg, bias=pbias, bias_bs2=Cg, act=1, c2=pre, R=x0, rmode=1, **kw)
        outs.append((xo, pre))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("a_t,b_t", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K,splitk", [(300, 200, 136, 1), (128, 128, 64, 1), (2112, 64, 64, 1), (1000, 520, 260, 1), (64, 72, 1000, 3), (6432, 1024, 512, 1)])
@pytest.mark.parametrize("x3", [False, True])
def test_f32_operand_kernel_matches_fp64(dev, a_t, b_t, M, N, K, splitk, x3):
    """scl_gemm_bf16 with f32 operands (gemm_f32.hip) against an fp64 reference.  Exact form (v_mfma_f32_16x16x4_f32, exact fp32 products,
    fp32 accumulation): 2e-6 of the row scale — the fp32 bar of north_star (1e-3) with three orders of margin.  Bf16-pair form
    (SCL_GEMM_F32X3: a = hi + lo, three bf16 MFMAs per term, fp32 accumulation): 2e-5 (measured 4 - 6e-6).  Ragged M / N / K (K % 4 != 0 rows
    are zero-padded to the 16-byte vector), split-K slabs, bias + GELU epilogue."""
    pad4 = lambda v: (v + 3) // 4 * 4
    g = torch.Generator().manual_seed(7)
    A = torch.randn(M, K, generator=g); B = torch.randn(N, K, generator=g)
    ref = (A.double() @ B.double().t())
    def operand(mat, t):
        m = mat.t().contiguous() if t else mat
        buf = torch.full((m.shape[0] + 2, pad4(m.shape[1]) + 4), float("nan"))
        buf[: m.shape[0], : m.shape[1]] = m
        if not t:
            buf[: m.shape[0], m.shape[1]: pad4(m.shape[1])] = 0.0   # K tail inside the last 16-byte vector must be finite zeros
        return buf.to(dev), buf.shape[1]
    bufA, ldA = operand(A, a_t); bufB, ldB = operand(B, b_t)
    if splitk > 1:
        C = torch.full((splitk, M, N), float("nan"), device=dev)
        ops.gemm(ops.Op(bufA, ldA), ops.Op(bufB, ldB), C, M, N, K, a_t=a_t, b_t=b_t, splitk=splitk, c_split_stride=M * N, x3=x3)
        got = C.sum(0)
    else:
        bias = torch.randn(N, generator=g)
        got = torch.full((M, N), float("nan"), device=dev)
        ops.gemm(ops.Op(bufA, ldA), ops.Op(bufB, ldB), got, M, N, K, a_t=a_t, b_t=b_t, bias=bias.to(dev), act=1, x3=x3)
        ref = torch.nn.functional.gelu(ref + bias.double())
    torch.cuda.synchronize()
    err = ((got.double().cpu() - ref).abs().max() / ref.abs().max()).item()
    assert err < (2e-5 if x3 else 2e-6), err
    if x3:
        assert err > 1e-7      # it really is the other kernel


def test_f32_kernel_addresses_a_2d_convolution_in_place(dev):
    """A 3x3, stride-(2,1) convolution over a channels-last, zero-padded map as ONE batched GEMM without an im2col buffer: rows =
    output positions (rpb = OW, rbstride = sh * Wp * C, ld = sw * C, one batch entry per utterance), K = (kh; kw*C + c) through the
    2-level contiguous index with cin = 3*C = 48 — not a power of two (magic division in col_off)."""
    Bz, H, W, C, Co, kh, kw, sh, sw = 3, 21, 18, 16, 24, 3, 3, 2, 1
    g = torch.Generator().manual_seed(3)
    x = torch.randn(Bz, C, H, W, generator=g); w = torch.randn(Co, C, kh, kw, generator=g); b = torch.randn(Co, generator=g)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), stride=(sh, sw), padding=1)          # [B, Co, OH, OW]
    OH, OW = ref.shape[2], ref.shape[3]
    Hp, Wp = H + 2, W + 2
    xp = torch.zeros(Bz, Hp, Wp, C); xp[:, 1:-1, 1:-1] = x.permute(0, 2, 3, 1)
    wk = w.permute(0, 2, 3, 1).reshape(Co, kh * kw * C).contiguous()                                       # [Co][kh][kw][C]
    xp_d = torch.cat([xp.reshape(-1), torch.zeros(4096)]).to(dev)
    y = torch.full((Bz, OH * OW, Co), float("nan"), device=dev)
    ops.gemm(ops.Op(xp_d, sw * C, rpb=OW, rbstride=sh * Wp * C, cin=kw * C, cout=Wp * C, bs1=Hp * Wp * C), ops.Op(wk.to(dev), kh * kw * C), y,
             OH * OW, Co, kh * kw * C, nb1=Bz, c_bs1=OH * OW * Co, bias=b.to(dev))
    torch.cuda.synchronize()
    got = y.view(Bz, OH, OW, Co).permute(0, 3, 1, 2).double().cpu()
    assert ((got - ref).abs().max() / ref.abs().max()).item() < 2e-5      # the module default (bf16-pair form); the exact form is pinned above


@pytest.mark.parametrize("case", ["plain_bf16", "bias_resid_f32", "gelu_c2", "gelu_grad_mul", "ragged"])
def test_auto_split_k_with_fused_finish_matches_the_single_launch(dev, case, monkeypatch):
    """Pack-sized linears (144 output tiles, K = 3072 / 4096) go through ops.gemm's automatic split-K + scl_gemm_splitk_finish: the
    epilogue of the original descriptor applied to the summed slabs.  Same values as the one-launch kernel up to the f32 summation
    order (three partial sums instead of one chain): bf16 outputs within one bf16 step, f32 outputs within 2e-5 of the scale."""
    from scl_amd import ops
    from scl_amd.ops import Op
    g = torch.Generator().manual_seed(11)
    M, N, K = (2189, 1024, 4096) if case != "ragged" else (2189, 1000, 3072)
    A = (0.05 * torch.randn(M, K, generator=g)).bfloat16().to(dev)
    b_t = case in ("gelu_grad_mul", "ragged")
    B = (0.05 * torch.randn(K, N, generator=g) if b_t else 0.05 * torch.randn(N, K, generator=g)).bfloat16().to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    R32 = torch.randn(M, N, generator=g).to(dev)
    Rbf = torch.randn(M, N, generator=g).bfloat16().to(dev)

    def run(auto):
        monkeypatch.setattr(ops, "_AUTO_SPLITK", auto)
        kw, outs = dict(b_t=b_t), []
        if case == "plain_bf16":
            C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        elif case == "bias_resid_f32":
            C = torch.empty(M, N, device=dev); kw.update(bias=bias, R=R32, rmode=1)
        elif case == "gelu_c2":
            C = torch.empty(M, N, dtype=torch.bfloat16, device=dev); c2 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            kw.update(bias=bias, act=1, c2=c2); outs.append(c2)
        elif case == "gelu_grad_mul":
            C = torch.empty(M, N, dtype=torch.bfloat16, device=dev); kw.update(R=Rbf, rmode=2, ract=1)
        else:
            C = torch.empty(M, N, device=dev); kw.update(bias=bias)
        assert ops._auto_splitk(M, N, K, False, 1, 1, 1, False) == (3 if auto else 1)
        ops.gemm(Op(A, K), Op(B, N if b_t else K), C, M, N, K, **kw)
        torch.cuda.synchronize()
        return [C.float().cpu()] + [o.float().cpu() for o in outs]

    one, split = run(False), run(True)
    for a, b in zip(one, split):
        scale = a.abs().max().item()
        tol = 2e-5 * scale if case in ("bias_resid_f32", "ragged") else scale * 2 ** -7
        assert (a - b).abs().max().item() <= tol, (case, (a - b).abs().max().item(), scale)
        assert ((a - b).abs() > 1e-6 * scale).float().mean().item() < 0.5


def test_gelu_with_stored_derivative_epilogues(dev):
    """ACT 5 (forward: C = gelu(x + b), C2 = gelu'(x + b)) and RMODE 2 / RACT 4 (backward: C = acc x R, R = the stored derivative) against
    torch: the GELU output must equal the ACT 1 epilogue's bit for bit, the stored derivative must be torch's erf-GELU derivative of the
    f32 pre-activation within bf16 rounding, and the backward product must equal the plain GEMM times R rounded once; 128 x 128 tiles,
    wide tiles and the split-K finishing pass all agree bit for bit."""
    M, N, K = 2 * 208 + 33, 512, 256
    A = _rand((M, K), dev, 201, 0.5); W = _rand((N, K), dev, 202, 0.1)
    bias = torch.randn(N, device=dev)
    pre = A.float() @ W.float().t() + bias
    outs = {}
    for name, sel in (("t128", dict(no_w8=True, no_p8=True, no_big=True, no_x2=True)), ("w8", dict(force_w8=True))):
        C1 = torch.empty(M, N, dtype=torch.bfloat16, device=dev); P1 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        ops.gemm(ops.Op(A, K), ops.Op(W, K), C1, M, N, K, bias=bias, act=1, c2=P1, **sel)
        C5 = torch.empty(M, N, dtype=torch.bfloat16, device=dev); D5 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        ops.gemm(ops.Op(A, K), ops.Op(W, K), C5, M, N, K, bias=bias, act=5, c2=D5, **sel)
        assert torch.equal(C1, C5)
        outs[name] = (C5, D5)
    assert torch.equal(outs["t128"][0], outs["w8"][0]) and torch.equal(outs["t128"][1], outs["w8"][1])
    x = pre.double()
    ref_d = 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)
    _close(outs["w8"][1], ref_d, 6e-3, "stored gelu'")
    # backward: dY @ W2 (x) stored derivative
    dY = _rand((M, K), dev, 203, 0.5)
    D = outs["w8"][1]
    plain = torch.empty(M, N, dtype=torch.float32, device=dev)
    ops.gemm(ops.Op(dY, K), ops.Op(W, K), plain, M, N, K, no_w8=True, no_p8=True, no_big=True, no_x2=True)
    want = (plain * D.float()).to(torch.bfloat16)
    for sel in (dict(no_w8=True, no_p8=True, no_big=True, no_x2=True), dict(force_w8=True)):
        G = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        ops.gemm(ops.Op(dY, K), ops.Op(W, K), G, M, N, K, R=D, rmode=2, ract=4, **sel)
        assert torch.equal(G, want)
    slabs = torch.empty(2, M, N, device=dev)
    ops.gemm(ops.Op(dY, K), ops.Op(W, K), slabs, M, N, K, splitk=2, c_split_stride=M * N, no_w8=True, no_p8=True, no_big=True, no_x2=True)
    G = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    dsc = ops._gemm_desc(ops.Op(dY, K), ops.Op(W, K), G, M, N, K, R=D, rmode=2, ract=4)
    ops._call("scl_gemm_splitk_finish", ops.ctypes.byref(dsc), ops._ptr(slabs), 2, M * N, ops._stream(), keep=dsc)
    assert torch.equal(G, ((slabs[0] + slabs[1]) * D.float()).to(torch.bfloat16))


@pytest.mark.parametrize("splitk", [1, 3, 16])
def test_wide_tile_kernel_utterance_batched_reduction_rows(dev, splitk):
    """Conv-stack weight gradient as a GEMM: both operands transposed with the K rows batched per utterance (A = the padded output-gradient
    layout, rows at an offset; B = overlapping im2col rows with stride 2): the wide ping-pong kernel computes its LDS-DMA offsets per
    K step (row_off divides by the frames per utterance) where flat operands advance a scalar offset.  Bit-identical to the 128 x 128
    kernel, slab by slab; NaN behind both buffers."""
    Bz, Tin, C, k, s = 8, 321, 512, 3, 2
    Tout = (Tin - k) // s + 1            # 160 frames -> K = 1280 rows = 20 K steps, utterance edges inside K steps
    Q, Rp = 2, Tout + 5
    z = _rand((Bz * Tin * C,), dev, 301, 0.3)
    dyp = torch.zeros(Bz, Rp, C)
    dyp[:, Q: Q + Tout] = torch.randn(Bz, Tout, C, generator=torch.Generator().manual_seed(302)) * 0.1
    dyp = dyp.to(torch.bfloat16).to(dev)
    def nanpad(t):
        buf = torch.full((t.numel() + 65536,), float("nan"), dtype=torch.bfloat16, device=dev)
        buf[:t.numel()] = t.reshape(-1)
        return buf
    zb, dyb = nanpad(z), nanpad(dyp)
    M, N, K = C, k * C, Bz * Tout
    outs = []
    for sel in (dict(no_w8=True, no_p8=True, no_big=True, no_x2=True), dict(force_w8=True), dict(force_w8=True)):
        out = torch.full((splitk, M, N), float("nan"), device=dev)
        ops.gemm(ops.Op(dyb, C, rpb=Tout, rbstride=Rp * C, offset=Q * C), ops.Op(zb, s * C, rpb=Tout, rbstride=Tin * C), out, M, N, K, a_t=True, b_t=True,
                 splitk=splitk, c_split_stride=M * N if splitk > 1 else 0, **sel)
        outs.append(out)
    d = ops._gemm_desc(ops.Op(dyb, C, rpb=Tout, rbstride=Rp * C, offset=Q * C), ops.Op(zb, s * C, rpb=Tout, rbstride=Tin * C), outs[-1], M, N, K, a_t=True,
                       b_t=True, splitk=splitk, c_split_stride=M * N if splitk > 1 else 0, force_w8=True)
    assert ops.L.load().scl_gemm_uses_wide_tiles(ops.ctypes.byref(d)) in (1, 2)
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    # against torch: dW[co][tap*C + ci] = sum_{b,t} dy[b][t][co] x[b][t*s + tap][ci]
    x = z.float().view(Bz, Tin, C).cpu(); dy = dyp.float()[:, Q: Q + Tout].cpu()
    want = torch.cat([torch.einsum("bto,btc->oc", dy, x[:, tap: tap + s * Tout: s][:, :Tout]) for tap in range(k)], dim=1)
    _close(outs[1].sum(0), want, 2e-3, "conv wgrad")


@pytest.mark.parametrize("t112", ["1", "0"])
@pytest.mark.parametrize("M,N", [(3 * 208 + 57, 1024), (2 * 208 + 1, 2 * 256 + 72), (12736, 1024), (6368, 1024)])
def test_wide_tile_epilogue_kinds_equal_the_128_tile_kernel(dev, monkeypatch, M, N, t112):
    """The compile-time epilogue kinds of the 208-row single-barrier kernels (gemm_w8_epi.h: 1 bf16 [+ bias], 2 fc1 forward with the
    stored gelu', 3 fc2 data gradient x stored derivative [+ column sums], 4 f32 [+ bias] + f32 residual) against the 128 x 128
    kernel and against the generic run-time-flag loop of the same kernel (SCL_W8_EPI_GENERIC is read once per process, so the
    generic arm is reached through a flag combination no kind matches: dropout with p = 0), bit for bit — ragged last row tile,
    a column edge inside a tile (the element-wise path inside a kind), both weight layouts."""
    # t112 (round 6): the two ragged shapes and the batch-32 shape (6368 rows: 57 x 4 tiles) plan 112-row tiles (4 + 3 row blocks, the second
    # wave row's single 3-block epilogue pass) when allowed; "0" keeps them on the 208-row tiles
    monkeypatch.setenv("SCL_W8_TILE112", t112)
    K = 512
    A = _rand((M, K), dev, 401, 0.3); W = _rand((N, K), dev, 402, 0.05); Wt = W.t().contiguous()
    if M == 6368:
        d0 = ops._gemm_desc(ops.Op(A, K), ops.Op(W, K), torch.empty(M, N, dtype=torch.bfloat16, device=dev), M, N, K)
        assert ops.L.load().scl_gemm_uses_wide_tiles(ops.ctypes.byref(d0)) == (4 if t112 == "1" else 0)      # the AUTOMATIC choice at batch 32
    bias = torch.randn(N, device=dev); Rb = _rand((M, N), dev, 403, 1.0); Rf = torch.randn(M, N, device=dev)
    cases = [
        (dict(), torch.bfloat16, False), (dict(bias=bias), torch.bfloat16, False),                                   # kind 1
        (dict(bias=bias, act=5), torch.bfloat16, True),                                                             # kind 2
        (dict(R=Rb, rmode=2, ract=4), torch.bfloat16, False),                                                       # kind 3
        (dict(R=Rf, rmode=1), torch.float32, False), (dict(bias=bias, R=Rf, rmode=1), torch.float32, False),        # kind 4
    ]
    for kw, cdt, c2 in cases:
        for b_t, opB in ((False, ops.Op(W, K)), (True, ops.Op(Wt, N))):
            outs = []
            for sel in (dict(no_w8=True, no_p8=True, no_big=True, no_x2=True), dict(force_w8=True), dict(force_w8=True, drop_p=0.0, drop_seed=0, _drop_flag=True)):
                sel = dict(sel)
                generic = sel.pop("_drop_flag", False)
                C = torch.full((M, N), float("nan"), dtype=cdt, device=dev)
                C2 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev) if c2 else None
                d = ops._gemm_desc(ops.Op(A, K), opB, C, M, N, K, b_t=b_t, c2=C2, **kw, **sel)
                if generic:
                    d.flags |= 0x80          # SCL_GEMM_DROPOUT with p = 0: every keep factor is 1, and no kind matches the flag word
                ops._call("scl_gemm_bf16", ops.ctypes.byref(d), ops._stream(), keep=d)
                outs.append((C, C2))
            for o in outs[1:]:
                assert torch.equal(outs[0][0], o[0]), (kw.keys(), b_t)
                if c2:
                    assert torch.equal(outs[0][1], o[1]), (kw.keys(), b_t)
    if N % 256 == 0:      # kind 3 with the per-tile column sums: partial rows equal the generic loop's
        parts = []
        for generic in (False, True):
            C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            kw = dict(b_t=True, R=Rb, rmode=2, ract=4, force_w8=True)
            rows = ops.gemm_colsum_rows(ops.Op(A, K), ops.Op(Wt, N), C, M, N, K, **kw)
            part = torch.full((rows, N), float("nan"), device=dev)
            d = ops._gemm_desc(ops.Op(A, K), ops.Op(Wt, N), C, M, N, K, colsum_part=part, **kw)
            if generic:
                d.flags |= 0x80
            ops._call("scl_gemm_bf16", ops.ctypes.byref(d), ops._stream(), keep=d)
            parts.append((C, part))
        assert torch.equal(parts[0][0], parts[1][0]) and torch.equal(parts[0][1], parts[1][1])


@pytest.mark.parametrize("Kr", [192, 448, 12736])
def test_grouped_weight_gradient_launch_equals_the_single_launches(dev, Kr):
    """scl_gemm_bf16_group: four A^T B problems of different output shapes in ONE launch (one block per 256 x 256 tile, whole reduction per
    block).  Every member must equal, BIT FOR BIT, the single wide-tile launch of the same problem without split-K (same tile, same K order,
    same epilogue) and torch's fp32 product of the bf16 operands to accumulation round-off; lists that do not qualify are refused."""
    E, Fd = (1024, 4096) if Kr > 1000 else (256, 512)
    shapes = [(E, Fd), (Fd, E), (E, E), (3 * E, E)]          # fc2, fc1, out-proj, qkv weight gradients: [rows of dY, rows of X]
    probs, refs, singles = [], [], []
    for i, (Mo, No) in enumerate(shapes):
        A = _rand((Kr, Mo), dev, 10 + i, 0.5)                  # dY [Kr, Mo]
        B = _rand((Kr, No), dev, 20 + i, 0.5)                  # X  [Kr, No]
        C = torch.full((Mo, No), float("nan"), device=dev)
        probs.append((ops.Op(A, Mo), ops.Op(B, No), C, Mo, No, Kr))
        refs.append(A.float().t() @ B.float())
        S = torch.empty(Mo, No, device=dev)
        ops.gemm(ops.Op(A, Mo), ops.Op(B, No), S, Mo, No, Kr, a_t=True, b_t=True, force_w8=True)
        singles.append(S)
    assert ops.gemm_group(probs) is True
    torch.cuda.synchronize()
    for (A, B, C, Mo, No, _), ref, S in zip(probs, refs, singles):
        _close(C, ref, 2e-5 * math.sqrt(Kr), "group vs fp32 (%d x %d)" % (Mo, No))
        assert torch.equal(C, S), "group member [%d, %d] differs from its single launch" % (Mo, No)
    # one member alone, and the refusals: K not a multiple of 64, fewer than 3 K steps, nine members, a bias
    C1 = torch.empty_like(probs[2][2])
    assert ops.gemm_group([probs[2][:2] + (C1,) + probs[2][3:]]) and torch.equal(C1, probs[2][2])
    A, B, C, Mo, No, _ = probs[0]
    assert ops.gemm_group([(A, B, C, Mo, No, Kr - 32)]) is False
    assert ops.gemm_group([(A, B, C, Mo, No, 128)]) is False
    assert ops.gemm_group(probs + probs + [probs[0]]) is False
    import ctypes
    d = ops._gemm_desc(A, B, C, Mo, No, Kr, a_t=True, b_t=True, bias=torch.zeros(No, device=dev))
    assert ops.L.load().scl_gemm_bf16_group_ok(ctypes.byref(d), 1) == 0
    assert ops.L.load().scl_gemm_bf16_group(ctypes.byref(d), 1, None) == -3


def test_grouped_launch_over_tile_ranges_covers_every_tile_once(dev):
    """scl_gemm_bf16_group_part: members are RANGES of their problems' tiles (the encoder carries a layer's remainder into the next
    layer's launch so that every launch is a whole round of 256 tiles).  Any partition of a problem's tiles over several launches must
    reproduce the single launch bit for bit; ranges outside a problem are refused."""
    import ctypes
    Kr, E, Fd = 448, 256, 512
    shapes = [(E, Fd), (Fd, E), (E, E), (3 * E, E), (Fd, Fd)]
    probs, singles = [], []
    for i, (Mo, No) in enumerate(shapes):
        A = _rand((Kr, Mo), dev, 30 + i, 0.5)
        B = _rand((Kr, No), dev, 40 + i, 0.5)
        probs.append((ops.Op(A, Mo), ops.Op(B, No), torch.full((Mo, No), float("nan"), device=dev), Mo, No, Kr))
        S = torch.empty(Mo, No, device=dev)
        ops.gemm(ops.Op(A, Mo), ops.Op(B, No), S, Mo, No, Kr, a_t=True, b_t=True, force_w8=True)
        singles.append(S)
    tiles = [ops.gemm_group_tiles(*p) for p in probs]
    assert tiles == [2, 2, 1, 3, 4]
    # three launches of 4 tiles each, oldest first, a problem split across launches where the cut falls
    queue = [[p, t, 0] for p, t in zip(probs, tiles)]
    launches = 0
    while queue:
        take, parts = 4, []
        while take and queue:
            it = queue[0]
            c = min(take, it[1] - it[2])
            parts.append((it[0], it[2], c)); it[2] += c; take -= c
            if it[2] == it[1]:
                queue.pop(0)
        ops.gemm_group_part(parts)
        launches += 1
    torch.cuda.synchronize()
    assert launches == 3
    for p, S in zip(probs, singles):
        assert torch.equal(p[2], S)
    with pytest.raises(ops.L.SclError, match="covers tiles"):
        ops.gemm_group_part([(probs[0], 1, 2)])


def test_triple_plane_bf16_gemm_reproduces_the_f32_product(dev):
    """Round 6, the scoring path's linears: scl_split3_f32_bf16 writes [hi | hi | lo] / [hi | lo | hi] rows (hi = bf16(x), lo = bf16(x - hi)),
    and ONE bf16 GEMM over 3 K then returns hi.hi + hi.lo + lo.hi.  Checked: the planes themselves (hi + lo reproduces x to 2^-16 relative, hi
    duplicated where the layout says so), the product against float64 (the f32-pair kernel's 4 - 6e-6 class) and against the f32-pair kernel
    on the same operands."""
    M, N, K = 1000, 520, 256
    g = torch.Generator().manual_seed(77)
    A = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    a3 = torch.empty(M * 3 * K, dtype=torch.bfloat16, device=dev)
    w3 = torch.empty(N * 3 * K, dtype=torch.bfloat16, device=dev)
    ops.split3(A, M, K, a3, 0)
    ops.split3(W, N, K, w3, 1)
    av, wv = a3.view(M, 3, K).float(), w3.view(N, 3, K).float()
    assert torch.equal(av[:, 0], av[:, 1]) and torch.equal(wv[:, 0], wv[:, 2])
    assert torch.equal(av[:, 0], A.to(torch.bfloat16).float()) and torch.equal(wv[:, 0], W.to(torch.bfloat16).float())
    for x, hi, lo in ((A, av[:, 0], av[:, 2]), (W, wv[:, 0], wv[:, 1])):
        assert ((hi + lo - x).abs() <= x.abs() * 2.0 ** -16 + 1e-30).all()
    C = torch.full((M, N), float("nan"), device=dev)
    ops.gemm(ops.Op(a3, 3 * K), ops.Op(w3, 3 * K), C, M, N, 3 * K)
    ref = A.double() @ W.double().t()
    err = ((C.double() - ref).abs().max() / ref.abs().max()).item()
    assert err < 2e-5, err
    C2 = torch.empty(M, N, device=dev)
    ops.gemm(ops.Op(A, K), ops.Op(W, K), C2, M, N, K, x3=True)
    assert ((C - C2).abs().max() / ref.abs().max()).item() < 2e-5
    # a row pitch larger than K (a view into a wider buffer) and the refusals
    wide = torch.randn(64, 320, generator=g).to(dev)
    o = torch.empty(64 * 3 * 256, dtype=torch.bfloat16, device=dev)
    ops.split3(wide, 64, 256, o, 0, ldx=320)
    assert torch.equal(o.view(64, 3, 256)[:, 0].float(), wide[:, :256].to(torch.bfloat16).float())
    with pytest.raises(ops.L.SclError):
        ops.split3(wide, 64, 100, o, 0, ldx=320)          # K not a multiple of 8


def test_layernorm_writes_the_triple_plane_operand_itself(dev):
    """scl_layernorm_fwd with act | 0x100: the [hi | hi | lo] row image must equal the split pass applied to the kernel's own f32 output."""
    M, C = 403, 1024
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(M, C, generator=g) * 3 + 0.7).to(dev)
    gam, bet = torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    y = torch.empty(M, C, device=dev)
    ops.layernorm_fwd(x, gam, bet, None, y, mean, rstd, M, C)
    want = torch.empty(M * 3 * C, dtype=torch.bfloat16, device=dev)
    ops.split3(y, M, C, want, 0)
    got = torch.full((M * 3 * C,), float("nan"), dtype=torch.bfloat16, device=dev)
    ops.layernorm_fwd(x, gam, bet, got, None, mean, rstd, M, C, act=0x100)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    ref = torch.nn.functional.layer_norm(x.double(), (C,), gam.double(), bet.double())
    v = got.view(M, 3, C).float()
    assert (((v[:, 0] + v[:, 2]).double() - ref).abs() <= ref.abs() * 2.0 ** -15 + 1e-5).all()      # hi + lo carries 16 mantissa bits


def test_wide_tile_epilogue_writes_the_triple_plane_image(dev):
    """SCL_GEMM_C_SPLIT3 (round 6: the scoring path's fc1 -> fc2 hand-over): the wide-tile epilogue stores bias + GELU of the product as the
    row image [hi | hi | lo] (ldc = 3 N) — bit for bit what scl_split3_f32_bf16 makes of the same launch's f32 output — on 208- and 112-row
    tiles; a launch that would not run on the wide tiles is refused, not served wrongly."""
    for M, N, K in ((12864, 4096, 3072), (6368, 1024, 512)):      # 208-row tiles; 112-row tiles (57 x 4)
        g = torch.Generator().manual_seed(M)
        A = _rand((M, K), dev, 61, 0.3); W = _rand((N, K), dev, 62, 0.05)
        bias = torch.randn(N, generator=g).to(dev)
        Cf = torch.empty(M, N, device=dev)
        ops.gemm(ops.Op(A, K), ops.Op(W, K), Cf, M, N, K, bias=bias, act=1, force_w8=True)
        want = torch.empty(M * 3 * N, dtype=torch.bfloat16, device=dev)
        ops.split3(Cf, M, N, want, 0)
        got = torch.full((M * 3 * N,), float("nan"), dtype=torch.bfloat16, device=dev)
        ops.gemm(ops.Op(A, K), ops.Op(W, K), got, M, N, K, bias=bias, act=1, ldc=3 * N, split3=True)
        assert torch.equal(got.view(torch.int16), want.view(torch.int16)), (M, N, K)
    small = torch.empty(64 * 3 * 64, dtype=torch.bfloat16, device=dev)
    with pytest.raises(ops.L.SclError, match="wide-tile kernel only"):
        ops.gemm(ops.Op(_rand((64, 64), dev, 1, 0.3), 64), ops.Op(_rand((64, 64), dev, 2, 0.3), 64), small, 64, 64, 64, ldc=192, split3=True)
    with pytest.raises(ops.L.SclError, match="ldc = 3 N"):
        ops.gemm(ops.Op(A, K), ops.Op(W, K), got, M, N, K, ldc=N, split3=True)
