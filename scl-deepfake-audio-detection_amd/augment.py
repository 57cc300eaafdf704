"""Host side of the on-GPU waveform augmentation.

Random parameters are sampled on the host with the reference's distributions and — so that a
seeded run is reproducible against the reference — from the global `np.random` stream in exactly
the order the reference consumes it (datautils/RawBoost.py, datautils/asvspoof_2019_augall_3.py:
377-439, core_scripts/data_io/wav_augmentation.py:255-274); the waveforms themselves never leave the
GPU: every stage below is a HIP kernel from csrc/augment.hip reached through the C ABI.
"""
import math

import threading

import numpy as np
import torch
from scipy import signal

from . import ops


# ---- parameter sampling (A1, RawBoost.py:14-48) ------------------------------------------------
def _rand_range(x1, x2, integer):
    y = np.random.uniform(low=x1, high=x2, size=(1,))
    return int(y[0]) if integer else y


def gen_notch_coeffs(nBands, minF, maxF, minBW, maxBW, minCoeff, maxCoeff, minG, maxG, fs):
    b = 1
    for _ in range(nBands):
        fc = _rand_range(minF, maxF, 0)
        bw = _rand_range(minBW, maxBW, 0)
        c = _rand_range(minCoeff, maxCoeff, 1)
        if c / 2 == int(c / 2):
            c = c + 1
        f1 = fc - bw / 2
        f2 = fc + bw / 2
        if f1 <= 0:
            f1 = 1 / 1000
        if f2 >= fs / 2:
            f2 = fs / 2 - 1 / 1000
        b = np.convolve(signal.firwin(c, [float(np.ravel(f1)[0]), float(np.ravel(f2)[0])], window="hamming", fs=fs), b)
    G = _rand_range(minG, maxG, 0)
    _, h = signal.freqz(b, 1, fs=fs)
    return np.asarray(pow(10, G / 20) * b / np.amax(np.abs(h)), dtype=np.float64).ravel()


def _draw_notch_params(nBands, minF, maxF, minBW, maxBW, minCoeff, maxCoeff, minG, maxG):
    """The random draws of genNotchCoeffs (RawBoost.py:25-47) in its order — per band fc, bw, c, then G — without the filter design."""
    fc, bw, c = np.empty(nBands), np.empty(nBands), np.empty(nBands, dtype=np.int64)
    for k in range(nBands):
        fc[k] = _rand_range(minF, maxF, 0)[0]
        bw[k] = _rand_range(minBW, maxBW, 0)[0]
        ck = _rand_range(minCoeff, maxCoeff, 1)
        c[k] = ck + 1 if ck / 2 == int(ck / 2) else ck
    G = _rand_range(minG, maxG, 0)[0]
    return fc, bw, c, G


# SCL_RAWBOOST_SCIPY=1: design every notch filter with scipy.signal.firwin / np.convolve / freqz exactly as the reference does (one call
# chain per band: 100 firwin calls per 11-view pack = 40 % of the pack builder's host time, profiles/r4_pack_builder.txt).  Default: the
# same draws in the same order, the filters of a clip designed at once in closed form (design_notch_filters: tests/test_host_cpu.py pins
# it to the scipy chain at 1e-12; the RawBoost goldens hold at their 3e-5 bar either way).
_SCIPY_DESIGN = __import__("os").environ.get("SCL_RAWBOOST_SCIPY", "0") == "1"
RIR_GEMM = __import__("os").environ.get("SCL_RIR_GEMM", "1") != "0"      # RIR convolution (>= 1024 taps) on the f32 matrix cores (below); 0: fir_kernel


def _draw_lnl(a, fs):
    taps = []
    minG, maxG = a.minG, a.maxG
    draws = []
    for i in range(a.N_f):
        if i == 1:
            minG = minG - a.minBiasLinNonLin
            maxG = maxG - a.maxBiasLinNonLin
        if _SCIPY_DESIGN:
            taps.append(gen_notch_coeffs(a.nBands, a.minF, a.maxF, a.minBW, a.maxBW, a.minCoeff, a.maxCoeff, minG, maxG, fs))
        else:
            draws.append(_draw_notch_params(a.nBands, a.minF, a.maxF, a.minBW, a.maxBW, a.minCoeff, a.maxCoeff, minG, maxG))
    if _SCIPY_DESIGN:
        return taps
    return design_notch_filters(np.stack([d[0] for d in draws]), np.stack([d[1] for d in draws]), np.stack([d[2] for d in draws]),
                                np.array([d[3] for d in draws]), fs)


def _draw_isd(a, L):
    beta = _rand_range(0, a.P, 0)
    n = int(L * (beta[0] / 100))
    p = np.random.permutation(L)[:n]
    f_r = np.multiply((2 * np.random.rand(p.shape[0])) - 1, (2 * np.random.rand(p.shape[0])) - 1)
    return p.astype(np.int32), f_r.astype(np.float32)


def _draw_ssi(a, L, fs):
    noise = np.random.normal(0, 1, L)
    if _SCIPY_DESIGN:
        b = gen_notch_coeffs(a.nBands, a.minF, a.maxF, a.minBW, a.maxBW, a.minCoeff, a.maxCoeff, a.minG, a.maxG, fs)
    else:
        fc, bw, c, G = _draw_notch_params(a.nBands, a.minF, a.maxF, a.minBW, a.maxBW, a.minCoeff, a.maxCoeff, a.minG, a.maxG)
        b = design_notch_filters(fc[None], bw[None], c[None], np.array([G]), fs)[0]
    snr = _rand_range(a.SNRmin, a.SNRmax, 0)
    return noise.astype(np.float32), b, float(snr[0])


# ---- fast parameter sampling (throughput mode) ----------------------------------------------------
# Same distributions as above, but (a) drawn from a private numpy Generator instead of the global legacy stream, so it
# is NOT draw-for-draw reproducible against the reference, and (b) all notch filters of a batch are designed at once:
# firwin in closed form, the five band-stop sections multiplied in the frequency domain (zero-phase: real spectra from one matrix
# product with a cosine table, design_notch_filters below), freqz = the first 512 bins of the 1024-point response.  ~50x less host time
# per clip than the per-filter scipy calls of the reference-compatible sampler.
# A numpy Generator is not safe under concurrent draws, and the pack builder runs on several threads (scl_amd/prefetch.py): every thread
# gets its own generator, spawned from one seed sequence in the order the threads first ask (the launch thread of bench.py is always the
# first: its stream is np.random.default_rng(seed)'s own, as before).
_FAST_ROOT = {"seq": np.random.SeedSequence(1234), "epoch": 0, "first": True}
_FAST_LOCAL = threading.local()
_FAST_LOCK = threading.Lock()


def _fast_rng():
    loc = _FAST_LOCAL
    if getattr(loc, "epoch", -1) != _FAST_ROOT["epoch"]:
        with _FAST_LOCK:
            if _FAST_ROOT["first"]:
                loc.rng = np.random.default_rng(_FAST_ROOT["seq"])          # == default_rng(seed) for the first (main) thread
                _FAST_ROOT["first"] = False
            else:
                loc.rng = np.random.default_rng(_FAST_ROOT["seq"].spawn(1)[0])
            loc.epoch = _FAST_ROOT["epoch"]
    return loc.rng


def seed_fast_sampler(seed):
    with _FAST_LOCK:
        _FAST_ROOT.update(seq=np.random.SeedSequence(seed), epoch=_FAST_ROOT["epoch"] + 1, first=True)


_COS_TABLES = {}      # nfft -> [rows, nfft / 2 + 1] float64: 2 cos(2 pi i k / nfft), row 0 = 1 (see _cos_table)
_HAMMING = {}         # c -> right half (centre first) of the symmetric Hamming window of length c


try:      # numpy's BLAS threads sleep between calls and take tens of ms to wake for a 2-ms product (measured: 0.4 ms warm, 26 - 190 ms after a
    # pause, 2.3 ms on the calling thread alone): the one matrix product of the filter design runs single-threaded
    from threadpoolctl import ThreadpoolController as _TPC
    _BLAS_CTL = _TPC()
except Exception:      # noqa: BLE001 - optional
    _BLAS_CTL = None


_BLAS_LOCK = threading.Lock()      # the limit is process-wide state: two threads entering / leaving it out of order would leave it changed


def _matmul_1thread(a, b):
    if _BLAS_CTL is None:
        return a @ b
    with _BLAS_LOCK, _BLAS_CTL.limit(limits=1, user_api="blas"):
        return a @ b


def _cos_table(kmax, nfft):
    """rows 0 .. kmax of the table for this transform size; ONE table per nfft, grown when a longer section turns up (the largest half
    length differs from call to call — a table per (kmax, nfft) was rebuilt on almost every one-clip call of the pack builder)"""
    t = _COS_TABLES.get(nfft)
    if t is None or t.shape[0] < kmax + 1:
        rows = max(kmax + 1, 64)
        k = np.arange(rows)[:, None]
        i = np.arange(nfft // 2 + 1)[None, :]
        t = 2.0 * np.cos((2.0 * np.pi / nfft) * ((k * i) % nfft))      # A(w_i) = h[0] + 2 sum_{k >= 1} h[k] cos(w_i k)
        t[0] = 1.0
        if len(_COS_TABLES) > 4:
            _COS_TABLES.clear()
        _COS_TABLES[nfft] = t
    return t[: kmax + 1]


def _hamming_half(c, kmax):
    """[..., kmax + 1]: w[k] of the symmetric Hamming window of odd length c at distance k from its centre (0 beyond (c - 1) / 2)."""
    out = np.zeros(c.shape + (kmax + 1,))
    for cv in np.unique(c):
        w = _HAMMING.get(int(cv))
        if w is None:
            a = (int(cv) - 1) // 2
            w = 0.54 - 0.46 * np.cos(2.0 * np.pi * (a + np.arange(a + 1)) / max(int(cv) - 1, 1)) if cv > 1 else np.ones(1)
            _HAMMING[int(cv)] = w
        out[c == cv, :len(w)] = w
    return out


_DESIGN_CHUNK = 64


def design_notch_filters(fc, bw, c, G, fs):
    """Vectorised genNotchCoeffs for n filters: fc, bw [n, nBands] (Hz), c [n, nBands] tap counts, G [n] (dB).
    Returns a list of n float64 tap vectors (length sum(c) - nBands + 1).

    Odd tap counts (what genNotchCoeffs draws, RawBoost.py:33-36) take the zero-phase route: every section is symmetric about its centre
    tap, so h[-k] = h[k] and its spectrum is REAL, A(w) = h[0] + 2 sum_k h[k] cos(w k) — one [sections, half taps] x [half taps, bins] matrix
    product with a cached cosine table instead of 1024-point FFTs of 101-tap sequences, half the sines of the closed-form firwin, the
    window from a table per length, a real product over the sections and one inverse transform per filter: 20 -> 8 ms per 320 filters."""
    n, nb = fc.shape
    c = np.asarray(c)
    if n > _DESIGN_CHUNK:      # [filters, sections, bins] float64 intermediates of one chunk stay cache-resident: 35 vs 56 us per filter
        out = []
        for i in range(0, n, _DESIGN_CHUNK):
            j = i + _DESIGN_CHUNK
            out += design_notch_filters(fc[i:j], bw[i:j], c[i:j], G[i:j], fs)
        return out
    nyq = fs / 2.0
    f1 = fc - bw / 2.0
    f2 = fc + bw / 2.0
    f1 = np.where(f1 <= 0, 1 / 1000, f1) / nyq
    f2 = np.where(f2 >= nyq, nyq - 1 / 1000, f2) / nyq
    if not bool((c % 2 == 1).all()):
        return _design_notch_filters_general(f1, f2, c, G)
    lens = c.sum(axis=1) - nb + 1
    nfft = 1024
    while nfft < int(lens.max()):                               # non-default --nBands / --maxCoeff: the circular product must not alias
        nfft *= 2
    alpha = (c - 1) // 2                                        # centre tap of every section
    kmax = int(alpha.max())
    k = np.arange(1, kmax + 1)[None, None, :]
    pik = np.pi * k
    # scipy.signal.firwin, pass_zero band-stop with cut-offs f1 < f2 (normalised to Nyquist): h[m] = f1 sinc(f1 m) + sinc(m) - f2 sinc(f2 m)
    # at distance m from the centre; for integer m != 0 that is (sin(pi f1 m) - sin(pi f2 m)) / (pi m), and f1 + 1 - f2 at the centre
    half = np.empty((n, nb, kmax + 1))
    half[..., 0] = f1 - f2 + 1.0
    half[..., 1:] = (np.sin(f1[..., None] * pik) - np.sin(f2[..., None] * pik)) / pik
    half *= _hamming_half(c, kmax)                              # zero beyond the section's own half length
    half /= (half[..., :1] + 2.0 * half[..., 1:].sum(axis=-1, keepdims=True))      # unity gain at DC
    A = _matmul_1thread(half.reshape(n * nb, kmax + 1), _cos_table(kmax, nfft)).reshape(n, nb, nfft // 2 + 1)
    spec = A.prod(axis=1)                                       # zero-phase response of the nBands sections in series
    bz = np.fft.irfft(spec, nfft, axis=-1)                      # zero-phase taps: bz[k] = bz[-k], k <= (lens - 1) / 2 < nfft / 2
    # freqz(b, 1, fs): 512 points on [0, fs/2) = every (nfft / 1024)-th bin; |H| = |A| (the linear phase has unit modulus)
    Hmag = np.abs(spec[:, ::nfft // 1024][:, :512]).max(axis=-1)
    bz *= (10.0 ** (G / 20.0) / Hmag)[:, None]
    htot = (lens - 1) // 2
    lmax = int(lens.max())
    taps = np.take_along_axis(bz, np.abs(np.arange(lmax)[None, :] - htot[:, None]) % nfft, axis=1)      # causal: b[t] = bz[|t - htot|]
    return [taps[i, :lens[i]].copy() for i in range(n)]


def _design_notch_filters_general(f1, f2, c, G):
    """any tap counts (even ones have no centre tap): closed-form firwin, the sections multiplied in the frequency domain by FFT"""
    n, nb = f1.shape
    cmax = int(c.max())
    idx = np.arange(cmax)[None, None, :]                        # [1,1,cmax]
    alpha = 0.5 * (c[..., None] - 1)
    m = idx - alpha
    valid = idx < c[..., None]
    h = f1[..., None] * np.sinc(f1[..., None] * m) + (np.sinc(m) - f2[..., None] * np.sinc(f2[..., None] * m))
    win = 0.54 - 0.46 * np.cos(2.0 * np.pi * idx / np.maximum(c[..., None] - 1, 1))   # symmetric Hamming of length c
    h = np.where(valid, h * win, 0.0)
    h = h / h.sum(axis=-1, keepdims=True)                       # unity gain at DC
    lens = c.sum(axis=1) - nb + 1
    nfft = 1024
    while nfft < int(lens.max()):
        nfft *= 2
    spec = np.fft.rfft(h, nfft, axis=-1).prod(axis=1)           # product of the nBands sections
    b = np.fft.irfft(spec, nfft, axis=-1)                       # [n, nfft]; exact linear convolution (total length <= nfft)
    Hmag = np.abs(spec[:, ::nfft // 1024][:, :512]).max(axis=-1)      # freqz(b, 1, fs): the transform of b IS spec
    b = (10.0 ** (G / 20.0) / Hmag)[:, None] * b
    return [b[i, :lens[i]].copy() for i in range(n)]


def _fast_notch_params(a, n, minG, maxG):
    r = _fast_rng()
    fc = r.uniform(a.minF, a.maxF, (n, a.nBands))
    bw = r.uniform(a.minBW, a.maxBW, (n, a.nBands))
    c = r.uniform(a.minCoeff, a.maxCoeff, (n, a.nBands)).astype(np.int64)   # int() truncation, then made odd (RawBoost.py:33-36)
    c = np.where(c % 2 == 0, c + 1, c)
    G = minG + (maxG - minG) * r.random(n)     # the reference draws uniform(-5, -20): low > high is legal in legacy numpy
    return fc, bw, c, G


def _fast_notch(a, n, minG, maxG, fs):
    return design_notch_filters(*_fast_notch_params(a, n, minG, maxG), fs)


def _fast_lnl(a, n, fs):
    """per clip: the linear branch's filter, then the N_f - 1 non-linear ones (lower gains) — drawn in that order, designed in ONE batch
    (a one-clip call of the pack builder pays the design's fixed cost once, not twice)"""
    lin = _fast_notch_params(a, n, a.minG, a.maxG)
    if a.N_f <= 1:
        taps = design_notch_filters(*lin, fs)
        return [[taps[i]] for i in range(n)]
    nl = _fast_notch_params(a, n * (a.N_f - 1), a.minG - a.minBiasLinNonLin, a.maxG - a.maxBiasLinNonLin)
    taps = design_notch_filters(*(np.concatenate([p, q]) for p, q in zip(lin, nl)), fs)
    k = a.N_f - 1
    return [[taps[i]] + taps[n + i * k: n + (i + 1) * k] for i in range(n)]


def _fast_isd(a, n, L):
    r = _fast_rng()
    draws = []
    for beta in r.uniform(0, a.P, n):
        k = int(L * (beta / 100))
        p = r.choice(L, k, replace=False).astype(np.int32)       # uniform k-subset, as permutation(L)[:k]
        draws.append((p, ((2 * r.random(k)) - 1) * ((2 * r.random(k)) - 1)))
    return [(p, f.astype(np.float32)) for p, f in draws]


def _fast_ssi(a, n, L, fs):
    r = _fast_rng()
    taps = _fast_notch(a, n, a.minG, a.maxG, fs)
    snr = r.uniform(a.SNRmin, a.SNRmax, n)
    return [(r.standard_normal(L).astype(np.float32), taps[i], float(snr[i])) for i in range(n)]


# ---- device stages -----------------------------------------------------------------------------
_LAST_TAP_TOTAL = 0


def last_tap_total():
    """Sum of FIR tap counts over all clips and branches of the most recent LnL / SSI stage (bench.py: achieved FLOP/s)."""
    return _LAST_TAP_TOTAL


def _h2d_pack(arrays, dev):
    """Several host arrays -> device tensors through ONE pinned staging buffer and ONE asynchronous copy.
    A `.to(device)` of pageable memory is a blocking hipMemcpy: it returns only after everything queued on the stream before it has run,
    i.e. the host re-joined the GPU at the start of every train step (and once more inside the loss), and the GPU then idled through
    the host's filter design and launch work: 2 - 3 ms of every 45 ms step (profiles/r5_bench_default_gaps.txt).  Pinned memory from
    torch's caching host allocator + non_blocking keeps the copy stream-ordered without stopping the host; the allocator re-uses a
    block only after the copy that reads it has completed."""
    arrays = [np.ascontiguousarray(a) for a in arrays]
    offs, pos = [], 0
    for a in arrays:
        offs.append(pos)
        pos += (a.nbytes + 15) // 16 * 16
    host = torch.empty(max(pos, 16), dtype=torch.uint8, pin_memory=True)
    hv = host.numpy()
    for a, o in zip(arrays, offs):
        hv[o:o + a.nbytes] = a.view(np.uint8).reshape(-1)
    d = upload_async(host, dev)
    return [d[o:o + a.nbytes].view(_TORCH_DT[a.dtype.type]).view(a.shape) for a, o in zip(arrays, offs)]


# Host -> device copies made by a thread that runs AHEAD of the GPU (the trainer's launch thread) go out on an UPLOAD STREAM of their own;
# the stream that will use the data waits for the copy's event.  A copy queued on the compute stream runs only when the GPU gets there
# and the consumer then waits for the host-memory read; on its own stream it runs when it is issued and the consumer's wait is long
# satisfied.  Measured with events inside the un-profiled default step (tools/augment_gaps.py, profiles/r6_augment_gaps.txt): the ISD
# upload + scatter 59 -> 20 us, the RawBoost chain 328 -> 273 us per step (its kernels: ~250); profiles/r6_upload_stream_ab.txt: default
# step -0.3 ms, pack-11 step -0.18 ms.  A thread whose stream is nearly EMPTY gains nothing and pays the extra stream switch + event per
# copy (the pack builder alone: 234 -> 198 packs/s), so the prefetcher's builder threads opt out with `use_upload_stream(False)`.
_UPLOAD = {}
_UPLOAD_LOCAL = threading.local()
UPLOAD_STREAM = __import__("os").environ.get("SCL_UPLOAD_STREAM", "1") != "0"


def use_upload_stream(on):
    """per-thread switch (default: on, unless SCL_UPLOAD_STREAM=0)"""
    _UPLOAD_LOCAL.on = bool(on)


def upload_async(host, dev):
    """pinned host tensor -> device tensor, asynchronously; safe to use on the CURRENT stream of the calling thread."""
    dev = torch.device(dev)
    if dev.type != "cuda" or not UPLOAD_STREAM or not getattr(_UPLOAD_LOCAL, "on", True):
        return host.to(dev, non_blocking=True)
    cur = torch.cuda.current_stream(dev)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), threading.get_ident())
    up = _UPLOAD.get(key)
    if up is None:
        up = _UPLOAD[key] = torch.cuda.Stream(device=dev)      # one per launching thread: the pack builder's copies do not queue behind the trainer's
    with torch.cuda.stream(up):
        d = host.to(dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(up)
    cur.wait_event(ev)
    d.record_stream(cur)      # allocated on the upload stream's pool, consumed (and eventually freed) on `cur`
    return d


_TORCH_DT = {np.float32: torch.float32, np.int32: torch.int32, np.int64: torch.int64, np.float64: torch.float64}


def _taps_to_device(taps_per_clip, dev, centred=True):
    """taps_per_clip: list (clips) of lists (filters) of float64 arrays."""
    global _LAST_TAP_TOTAL
    _LAST_TAP_TOTAL = sum(len(b) for taps in taps_per_clip for b in taps)
    flat, off, ln, hh = [], [], [], []
    pos = 0
    for taps in taps_per_clip:
        for b in taps:
            flat.append(np.asarray(b, dtype=np.float32))
            off.append(pos)
            ln.append(len(b))
            hh.append((len(b) + 1) // 2 if centred else 0)
            pos += len(b)
    return tuple(_h2d_pack([np.concatenate(flat), np.asarray(off, dtype=np.int32), np.asarray(ln, dtype=np.int32), np.asarray(hh, dtype=np.int32)], dev))


def _lnl_stage(x, taps_per_clip):
    n, L = x.shape
    nblk = ops.fir_nblocks(L)
    taps, off, ln, hh = _taps_to_device(taps_per_clip, x.device)
    y = torch.empty_like(x)
    part = torch.empty(n * nblk * 4, device=x.device)
    ops.fir_multi(x, L, L, taps, off, ln, hh, n, len(taps_per_clip[0]), True, y, L, L, part)
    out = torch.empty_like(x)
    ops.clip_affine(ops.AFF_CENTER_PEAK_COND, y, L, part, out, L, L, n)
    return out


def _isd_stage(x, draws, g_sd):
    n, L = x.shape
    y = x.clone()
    offs = np.zeros(n + 1, dtype=np.int32)
    for i, (p, _) in enumerate(draws):
        offs[i + 1] = offs[i] + len(p)
    if offs[-1] > 0:
        pos, fr, offs_d = _h2d_pack([np.concatenate([p for p, _ in draws]).astype(np.int32), np.concatenate([f for _, f in draws]).astype(np.float32),
                                     offs], x.device)
        ops.isd_scatter(y, L, pos, fr, offs_d, n, int(np.diff(offs).max()), float(g_sd))
    part = torch.empty(n * ops.fir_nblocks(L) * 4, device=x.device)
    ops.clip_stats(y, L, L, n, part)
    out = torch.empty_like(x)
    ops.clip_affine(ops.AFF_PEAK_COND, y, L, part, out, L, L, n)
    return out


def _ssi_stage(x, draws):
    n, L = x.shape
    dev = x.device
    nblk = ops.fir_nblocks(L)
    noise, snr = _h2d_pack([np.stack([d[0] for d in draws]).astype(np.float32), np.asarray([d[2] for d in draws], dtype=np.float32)], dev)
    taps, off, ln, hh = _taps_to_device([[d[1]] for d in draws], dev)
    nf = torch.empty_like(x)
    part_n = torch.empty(n * nblk * 4, device=dev)
    ops.fir_multi(noise, L, L, taps, off, ln, hh, n, 1, False, nf, L, L, part_n)
    part_x = torch.empty(n * nblk * 4, device=dev)
    ops.clip_stats(x, L, L, n, part_x)
    out = torch.empty_like(x)
    ops.clip_affine(ops.AFF_SSI_MIX, nf, L, part_n, out, L, L, n, z=x, ldz=L, partz=part_x, snr_db=snr)
    return out


def _peak_cond(x):
    n, L = x.shape
    part = torch.empty(n * ops.fir_nblocks(L) * 4, device=x.device)
    ops.clip_stats(x, L, L, n, part)
    out = torch.empty_like(x)
    ops.clip_affine(ops.AFF_PEAK_COND, x, L, part, out, L, L, n)
    return out


_CHAINS = {1: "L", 2: "I", 3: "S", 4: "LIS", 5: "LI", 6: "LS", 7: "IS"}


def rawboost_batch(x, args, algo, sr=16000, sampler="reference"):
    """process_Rawboost_feature (augall_3:377-439) applied to every row of x [n, L] (fp32, on the
    GPU).  sampler="reference": draws come from the global np.random stream, clip by clip and stage by
    stage, exactly like n successive reference calls; sampler="fast": same distributions from a private
    generator with batched filter design (throughput mode)."""
    assert x.dim() == 2 and x.dtype == torch.float32 and x.is_cuda
    x = x.contiguous()
    n, L = x.shape
    fast = sampler == "fast"
    if algo == 8:
        if fast:
            draws = list(zip(_fast_lnl(args, n, sr), _fast_isd(args, n, L)))
        else:
            draws = [(_draw_lnl(args, sr), _draw_isd(args, L)) for _ in range(n)]
        f1 = _lnl_stage(x, [d[0] for d in draws])
        f2 = _isd_stage(x, [d[1] for d in draws], args.g_sd)
        s = torch.empty_like(x)
        ops.add_f32(f1, f2, s, None, x.numel())
        return _peak_cond(s)
    chain = _CHAINS.get(algo)
    if chain is None:
        return x
    if fast:
        per_stage = {st: (_fast_lnl(args, n, sr) if st == "L" else (_fast_isd(args, n, L) if st == "I" else _fast_ssi(args, n, L, sr)))
                     for st in chain}
        draws = [{st: per_stage[st][i] for st in chain} for i in range(n)]
    else:
        draws = []
        for _ in range(n):
            d = {}
            for st in chain:
                d[st] = _draw_lnl(args, sr) if st == "L" else (_draw_isd(args, L) if st == "I" else _draw_ssi(args, L, sr))
            draws.append(d)
    y = x
    for st in chain:
        if st == "L":
            y = _lnl_stage(y, [d["L"] for d in draws])
        elif st == "I":
            y = _isd_stage(y, [d["I"] for d in draws], args.g_sd)
        else:
            y = _ssi_stage(y, [d["S"] for d in draws])
    return y


# ---- RIR convolution on the f32 matrix cores (round 6) -----------------------------------------------------------------------------------
# SURVEY.md 8(d) asks for a formulation of the 8000-tap RIR convolution in which the matrix cores do the products.  Taking the outputs 64 at a
# time makes it a GEMM without a band of wasted products:
#     y[64 a + b] = sum_k A[a][k] * B[b][k],   A[a][k] = xpad[64 a + k]            (rows of the SIGNAL that overlap: row pitch 64 < K = R + 63)
#                                              B[b][k] = h[b + R - 1 - k] or 0     (the taps as a 64-row Toeplitz image, built once per RIR)
# i.e. M = Lout / 64 rows, N = 64, K = R + 63 (0.8 % extra products at R = 8000) on the library's exact-f32 MFMA kernel (an fmaf chain per
# output: v_mfma_f32_16x16x4_f32; bf16 operands would break the +-1-LSB property of A7), the overlapping-row operand being the same
# descriptor the conv stack uses; split-K (the output is only 18 tiles) with the slabs summed in index order.  Measured on MI355X
# (profiles/r6_fir_toeplitz_probe.txt): ONE clip x 8000 taps — what reverb_wrapper runs — 296 us on fir_kernel (one clip is a handful of
# workgroups: 3.5 TFLOP/s) -> 26 us; 16 clips 339 -> 176 us (93 TFLOP/s of useful work).  RawBoost's 121 - 411-tap filters stay on
# fir_kernel: the GEMM form measured 235 vs 202 us there (24 % band waste, and five powered / shifted signal copies to lay out).
_RIR_GEMM_MIN_TAPS = 1024
_TOEPLITZ = {}      # (data_ptr, R, device) -> (weak reference to the RIR tensor, image [64, Kp], Kp)


def _toeplitz_image(rir):
    import weakref
    R = rir.numel()
    key = (rir.data_ptr(), R, str(rir.device))
    ent = _TOEPLITZ.get(key)
    if ent is not None and ent[0]() is rir:
        return ent[1], ent[2]
    Kp = (R + 63 + 15) // 16 * 16
    b = torch.arange(64, device=rir.device)[:, None]
    k = torch.arange(Kp, device=rir.device)[None, :]
    t = b + (R - 1) - k
    img = torch.where((t >= 0) & (t < R), rir.float()[t.clamp(0, R - 1)], torch.zeros((), device=rir.device)).contiguous()
    if len(_TOEPLITZ) > 256:
        _TOEPLITZ.clear()
    _TOEPLITZ[key] = (weakref.ref(rir), img, Kp)      # RIR tensors live in the audio bank (scl_amd/pack.py): the image is built once per file
    return img, Kp


def _rir_full_conv_gemm(x, rir):
    """y[m] = sum_t rir[t] x[m - t], m in [0, L + R - 1): numpy.convolve(x, rir) (reverb.py:37) as the GEMM above.  Returns y and the
    partial statistics clip_affine needs."""
    L, R = x.numel(), rir.numel()
    Lout = L + R - 1
    dev = x.device
    img, Kp = _toeplitz_image(rir)
    M = (Lout + 63) // 64
    Lp = 64 * (M - 1) + Kp + 64
    xpad = torch.zeros(Lp, device=dev)
    xpad[R - 1: R - 1 + L] = x
    nk = Kp // 16
    splitk = max(1, min(32, nk // 8, (4 * 256) // max(1, ((M + 63) // 64))))
    y = torch.empty(M * 64, device=dev)
    if splitk > 1:
        slabs = torch.empty(splitk * M * 64, device=dev)
        ops.gemm(ops.Op(xpad, 64), ops.Op(img, Kp), slabs, M, 64, Kp, splitk=splitk, c_split_stride=M * 64, x3=False)
        ops.reduce_slabs(slabs, y, M * 64, splitk, M * 64)
    else:
        ops.gemm(ops.Op(xpad, 64), ops.Op(img, Kp), y, M, 64, Kp, x3=False)
    part = torch.empty(ops.fir_nblocks(Lout) * 4, device=dev)
    ops.clip_stats(y, Lout, Lout, 1, part)
    return y, part, Lout


def reverb(x, rir):
    """ReverbAugmentor.transform (reverb.py:33-44): full convolution, peak normalise, int16 C cast;
    returns the int16 VALUES as fp32 (pydub_to_librosa keeps them unscaled, utils.py:20-22)."""
    L, R = x.numel(), rir.numel()
    Lout = L + R - 1
    dev = x.device
    if R >= _RIR_GEMM_MIN_TAPS and RIR_GEMM:
        y, part, _ = _rir_full_conv_gemm(x.contiguous().float(), rir)
    else:
        z0, zr = _h2d_pack([np.zeros(1, dtype=np.int32), np.array([R], dtype=np.int32)], dev)
        y = torch.empty(Lout, device=dev)
        part = torch.empty(ops.fir_nblocks(Lout) * 4, device=dev)
        ops.fir_multi(x.contiguous(), L, L, rir.contiguous().float(), z0, zr, z0, 1, 1, False, y, Lout, Lout, part)
    out = torch.empty(Lout, device=dev)
    ops.clip_affine(ops.AFF_PEAK_QUANT_I16, y, Lout, part, out, Lout, Lout, 1)
    return out


def to_int16(x):
    out = torch.empty(x.numel(), dtype=torch.int16, device=x.device)
    ops.f32_to_i16_wrap(x.contiguous(), out, x.numel())
    return out


def _dbfs(sumsq, n):
    rms = math.isqrt(int(sumsq) // int(n)) if n else 0
    return -float("inf") if rms == 0 else 20 * math.log(rms / 32768.0, 10)      # pydub.utils.ratio_to_db spells log10 this way


def host_i16_sumsq(x):
    """sum of squares of librosa_to_pydub(x) (utils.py:24-30: int16(x * 32768) by C cast — truncation toward zero, +1.0 wraps to -32768),
    evaluated on the host from the float32 samples: the exact integer scl_f32_to_i16_wrap + scl_i16_sumsq produce on the device.  Lets
    background_noise() run without its host <- device round trip (round 6: that `.tolist()` was the pack builder's one synchronisation per
    pack — the builder thread sat out the queueing delay of every kernel it had issued before it, on a GPU saturated by the training step)."""
    v = np.trunc(np.asarray(x, dtype=np.float32) * np.float32(32768.0)).astype(np.int64)
    v = ((v + 32768) % 65536) - 32768
    return int((v * v).sum())


def background_noise(x, noise_i16, snr_db, sumsq=None):
    """BackgroundNoiseAugmentor.transform (background_noise.py:40-56) with the noise file and
    SNR_dB = random.randint(5, 15) given; pydub / audioop integer semantics, bit-exact.
    sumsq = (speech, noise) sums of squares of the two int16 signals when the caller already has them (host_i16_sumsq: no device sync)."""
    dev = x.device
    sp = to_int16(x)
    n, nn = sp.numel(), noise_i16.numel()
    if sumsq is not None:
        sums = [int(sumsq[0]), int(sumsq[1])]
    else:
        parts = torch.zeros(2, 64, dtype=torch.int64, device=dev)
        ops.i16_sumsq(sp, n, parts[0], 64)
        ops.i16_sumsq(noise_i16.contiguous(), nn, parts[1], 64)
        sums = parts.sum(1).tolist()                      # host sync: two integers per clip
    sig_db, noi_db = _dbfs(sums[0], n), _dbfs(sums[1], nn)
    gain = snr_db * noi_db / sig_db                   # (sic) background_noise.py:52
    factor = 10.0 ** (gain / 20.0)
    out = torch.empty(n, device=dev)
    ops.i16_gain_overlay(sp, n, noise_i16, nn, factor, out_f32=out)
    return out


def _fade_params(from_gain_db, to_gain_db, duration_ms, sr):
    """pydub AudioSegment.fade's gain ramp: (one step per millisecond?, from_power, scale_step) — fades longer than 100 ms step once
    per millisecond, shorter ones once per frame.  Python float arithmetic, as pydub evaluates it."""
    from_power = 10 ** (float(from_gain_db) / 20)
    gain_delta = 10 ** (float(to_gain_db) / 20) - from_power
    if duration_ms > 100:
        return 1, from_power, gain_delta / duration_ms
    return 0, from_power, gain_delta / (duration_ms * (sr / 1000.0))          # frame_count(ms=end) - frame_count(ms=0)


def speed(x, speed_factor, sr=16000, chunk_size=150, crossfade=25):
    """SpeedAugmentor.transform (speed.py:29-33) = pydub 0.25.1 AudioSegment.speedup(speed_factor) on the clip's int16 image
    (librosa_to_pydub, utils.py:24-30); returns the int16 VALUES as fp32 (pydub_to_librosa).  The chunk list and pydub's
    millisecond bookkeeping are walked here; every AudioSegment.append is one in-place kernel over the running output
    (scl_i16_append_xfade).  For speed_factor < 1 pydub's crossfade comes out negative and the output is a few hundred
    milliseconds long (see include/scl_hip.h) — the reference's behaviour, kept."""
    if sr % 1000:
        raise ValueError("speed: sample rates that are not whole frames per millisecond are not supported (got %d)" % sr)
    F = sr // 1000
    src = to_int16(x)
    N = int(src.numel())
    len_ms = round(1000 * (float(N) / sr))                                # AudioSegment.__len__
    atk = 1.0 / speed_factor
    if speed_factor < 2.0:
        remove = int(chunk_size * (1 - atk) / atk)
    else:
        remove = int(chunk_size)
        chunk_size = int(atk * chunk_size / (1 - atk))
    crossfade = min(crossfade, remove - 1)
    step = chunk_size + remove
    if step <= 0:
        raise ValueError("speed: chunk length %d ms" % step)
    nchunks = int(math.ceil(len_ms / float(step)))
    if nchunks < 2:
        raise ValueError("Could not speed up AudioSegment, it was too short %.2fs for %d ms chunks at %.2fx" % (N / sr, chunk_size, speed_factor))
    remove -= crossfade
    if not 0 < remove < step:
        raise ValueError("speed: cannot drop %d ms of a %d ms chunk" % (remove, step))
    Lc = F * (step - remove)                                              # frames of every chunk but the last
    s_last = F * (nchunks - 1) * step
    e_last = F * min(nchunks * step, len_ms)
    have_last = max(0, min(e_last, N) - s_last)
    c = abs(crossfade)
    cap = Lc + (nchunks - 2) * (Lc if crossfade >= 0 else F * c) + (e_last - s_last) + 64
    out = torch.zeros(cap, dtype=torch.int16, device=x.device)           # zeros: pydub pads a short last slice with silence
    none = (0, 1.0, 0.0)
    ops.i16_append_xfade(out, 0, src, Lc, 0, 0, none, 0, none, 0, Lc, F)                     # out = chunks[0]
    n = Lc
    for i in range(1, nchunks - 1):
        chunk = src[F * i * step:]
        if crossfade == 0:
            ops.i16_append_xfade(out, n, chunk, Lc, 0, 0, none, 0, none, 0, Lc, F)
            n += Lc
            continue
        if crossfade > n // F or crossfade > Lc // F:
            raise ValueError("speed: crossfade %d ms is longer than the segment" % crossfade)
        if crossfade > 0:
            R = m2 = F * crossfade
            a0, tail_off, tail_n = n - R, R, Lc - R
        else:                                                             # seg1[c:] faded out under looped seg2[:-c]; then seg2[-c:]
            a0, R, m2 = F * c, n - F * c, Lc - F * c
            tail_off, tail_n = Lc - F * c, F * c
            if R <= 0 or m2 <= 0:
                raise ValueError("speed: empty cross-fade (pydub divides by zero / never returns here)")
        ops.i16_append_xfade(out, n, chunk, Lc, a0, R, _fade_params(0, -120, R // F, sr), m2, _fade_params(-120, 0, m2 // F, sr), tail_off, tail_n, F)
        n += tail_n
    if have_last:
        ops.i16_append_xfade(out, n, src[s_last:], have_last, 0, 0, none, 0, none, 0, have_last, F)     # out += last_chunk
    n += e_last - s_last                                                  # frames missing from the last slice stay silent
    return out[:n].to(torch.float32)


def pitch_shift(x, n_steps, sr=16000):
    """PitchAugmentor.transform (pitch.py:31-38) = librosa 0.10.0 effects.pitch_shift(data, sr, n_steps) followed by the int16
    round trip (librosa_to_pydub, pydub_to_librosa): stft -> phase vocoder (rate 2^(-n/12)) -> istft -> resample by `rate` ->
    fix_length; returns the int16 VALUES as fp32.  The resampler is a Kaiser-windowed sinc where librosa calls soxr_hq."""
    dev = x.device
    x = x.contiguous().float()
    L = int(x.numel())
    rate = 2.0 ** (-float(n_steps) / 12)
    nfr = ops.stft_nframes(L)
    D = torch.empty(nfr * 1025 * 2, device=dev)
    ops.stft(x, L, D, nfr)
    nsteps = int(math.ceil(nfr / rate))                                   # len(np.arange(0, nfr, rate))
    Ds = torch.empty(nsteps * 1025 * 2, device=dev)
    ops.phase_vocoder(D, nfr, rate, Ds, nsteps)
    len_stretch = int(round(L / rate))
    nuse = min(nsteps, int(math.ceil((len_stretch + 2048) / 512)))
    ws = torch.empty(nuse * 2048, device=dev)
    ys = torch.empty(len_stretch, device=dev)
    ops.istft(Ds, nuse, ws, ys, len_stretch)
    if rate != 1.0:                                                       # librosa.resample returns its input when the rates agree
        n_out = int(math.ceil(len_stretch * rate))
        yr = torch.empty(n_out, device=dev)
        ops.resample_sinc(ys, len_stretch, rate, yr, n_out)
    else:
        yr = ys
    fixed = torch.zeros(L, device=dev)
    m = min(L, int(yr.numel()))
    fixed[:m] = yr[:m]
    return to_int16(fixed).to(torch.float32)


def multiview_crop(views, length, repeat_pad, random_trim=True):
    """batch_pad_for_multiview (wav_augmentation.py:209-282) for a list of 1-D device tensors;
    returns [V, out_len].  Draws one np.random.rand() under the reference's condition."""
    dev = views[0].device
    lens = [int(v.numel()) for v in views]
    firstlen = lens[0]
    if firstlen < length:
        start, out_len = 0, (length if repeat_pad else firstlen)
    elif random_trim:
        start, out_len = int(np.random.rand() * (firstlen - length)), length
    else:
        start, out_len = 0, length
    src = torch.cat([v.reshape(-1).float() for v in views])
    # offsets and lengths through one pinned, asynchronous copy (a torch.tensor(..., device=) is a blocking copy behind the whole pack's chain)
    off, lens_d = _h2d_pack([np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64), np.asarray(lens, dtype=np.int32)], dev)
    out = torch.empty(len(views), out_len, device=dev)
    ops.multiview_crop(src, off, lens_d, len(views), firstlen, start, out_len, repeat_pad, out, out_len)
    return out
