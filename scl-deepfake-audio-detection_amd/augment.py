"""Host side of the on-GPU waveform augmentation.

Random parameters are sampled on the host with the reference's distributions and — so that a
seeded run is reproducible against the reference — from the global `np.random` stream in exactly
the order the reference consumes it (datautils/RawBoost.py, datautils/asvspoof_2019_augall_3.py:
377-439, core_scripts/data_io/wav_augmentation.py:255-274); the waveforms themselves never leave the
GPU: every stage below is a HIP kernel from csrc/augment.hip reached through the C ABI.
"""
import math

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


def _draw_lnl(a, fs):
    taps = []
    minG, maxG = a.minG, a.maxG
    for i in range(a.N_f):
        if i == 1:
            minG = minG - a.minBiasLinNonLin
            maxG = maxG - a.maxBiasLinNonLin
        taps.append(gen_notch_coeffs(a.nBands, a.minF, a.maxF, a.minBW, a.maxBW, a.minCoeff, a.maxCoeff, minG, maxG, fs))
    return taps


def _draw_isd(a, L):
    beta = _rand_range(0, a.P, 0)
    n = int(L * (beta[0] / 100))
    p = np.random.permutation(L)[:n]
    f_r = np.multiply((2 * np.random.rand(p.shape[0])) - 1, (2 * np.random.rand(p.shape[0])) - 1)
    return p.astype(np.int32), f_r.astype(np.float32)


def _draw_ssi(a, L, fs):
    noise = np.random.normal(0, 1, L)
    b = gen_notch_coeffs(a.nBands, a.minF, a.maxF, a.minBW, a.maxBW, a.minCoeff, a.maxCoeff, a.minG, a.maxG, fs)
    snr = _rand_range(a.SNRmin, a.SNRmax, 0)
    return noise.astype(np.float32), b, float(snr[0])


# ---- device stages -----------------------------------------------------------------------------
def _taps_to_device(taps_per_clip, dev, centred=True):
    """taps_per_clip: list (clips) of lists (filters) of float64 arrays."""
    flat, off, ln, hh = [], [], [], []
    pos = 0
    for taps in taps_per_clip:
        for b in taps:
            flat.append(np.asarray(b, dtype=np.float32))
            off.append(pos)
            ln.append(len(b))
            hh.append((len(b) + 1) // 2 if centred else 0)
            pos += len(b)
    t = lambda arr, dt: torch.from_numpy(np.asarray(arr, dtype=dt)).to(dev)
    return t(np.concatenate(flat), np.float32), t(off, np.int32), t(ln, np.int32), t(hh, np.int32)


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
        pos = torch.from_numpy(np.concatenate([p for p, _ in draws]).astype(np.int32)).to(x.device)
        fr = torch.from_numpy(np.concatenate([f for _, f in draws]).astype(np.float32)).to(x.device)
        ops.isd_scatter(y, L, pos, fr, torch.from_numpy(offs).to(x.device), n, int(np.diff(offs).max()), float(g_sd))
    part = torch.empty(n * ops.fir_nblocks(L) * 4, device=x.device)
    ops.clip_stats(y, L, L, n, part)
    out = torch.empty_like(x)
    ops.clip_affine(ops.AFF_PEAK_COND, y, L, part, out, L, L, n)
    return out


def _ssi_stage(x, draws):
    n, L = x.shape
    dev = x.device
    nblk = ops.fir_nblocks(L)
    noise = torch.from_numpy(np.stack([d[0] for d in draws])).to(dev)
    taps, off, ln, hh = _taps_to_device([[d[1]] for d in draws], dev)
    snr = torch.tensor([d[2] for d in draws], dtype=torch.float32, device=dev)
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


def rawboost_batch(x, args, algo, sr=16000):
    """process_Rawboost_feature (augall_3:377-439) applied to every row of x [n, L] (fp32, on the
    GPU).  Draws for clip i are taken before those of clip i+1, stage by stage, like n successive
    reference calls."""
    assert x.dim() == 2 and x.dtype == torch.float32 and x.is_cuda
    x = x.contiguous()
    n, L = x.shape
    if algo == 8:
        draws = [(_draw_lnl(args, sr), _draw_isd(args, L)) for _ in range(n)]
        f1 = _lnl_stage(x, [d[0] for d in draws])
        f2 = _isd_stage(x, [d[1] for d in draws], args.g_sd)
        s = torch.empty_like(x)
        ops.add_f32(f1, f2, s, None, x.numel())
        return _peak_cond(s)
    chain = _CHAINS.get(algo)
    if chain is None:
        return x
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


def reverb(x, rir):
    """ReverbAugmentor.transform (reverb.py:33-44): full convolution, peak normalise, int16 C cast;
    returns the int16 VALUES as fp32 (pydub_to_librosa keeps them unscaled, utils.py:20-22)."""
    L, R = x.numel(), rir.numel()
    Lout = L + R - 1
    dev = x.device
    i32 = lambda v: torch.tensor([v], dtype=torch.int32, device=dev)
    y = torch.empty(Lout, device=dev)
    part = torch.empty(ops.fir_nblocks(Lout) * 4, device=dev)
    ops.fir_multi(x.contiguous(), L, L, rir.contiguous().float(), i32(0), i32(R), i32(0), 1, 1, False, y, Lout, Lout, part)
    out = torch.empty(Lout, device=dev)
    ops.clip_affine(ops.AFF_PEAK_QUANT_I16, y, Lout, part, out, Lout, Lout, 1)
    return out


def to_int16(x):
    out = torch.empty(x.numel(), dtype=torch.int16, device=x.device)
    ops.f32_to_i16_wrap(x.contiguous(), out, x.numel())
    return out


def _dbfs(sumsq, n):
    rms = math.isqrt(int(sumsq) // int(n)) if n else 0
    return -float("inf") if rms == 0 else 20.0 * math.log10(rms / 32768.0)


def background_noise(x, noise_i16, snr_db):
    """BackgroundNoiseAugmentor.transform (background_noise.py:40-56) with the noise file and
    SNR_dB = random.randint(5, 15) given; pydub / audioop integer semantics, bit-exact."""
    dev = x.device
    sp = to_int16(x)
    n, nn = sp.numel(), noise_i16.numel()
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
    off = torch.tensor(np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64), device=dev)
    out = torch.empty(len(views), out_len, device=dev)
    ops.multiview_crop(src, off, torch.tensor(lens, dtype=torch.int32, device=dev), len(views), firstlen, start, out_len,
                       repeat_pad, out, out_len)
    return out
