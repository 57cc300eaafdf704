"""ORACLE (test infrastructure — never imported by the product path).

CPU restatement, in numpy float64, of the reference's RawBoost augmentation
(datautils/RawBoost.py) and its algo dispatch (datautils/asvspoof_2019_augall_3.py:377-439).

Randomness contract: every function that draws takes `rng` (default: the global `np.random`
module) and consumes it in EXACTLY the order the reference does, so that seeding the global
numpy RNG and calling either implementation gives identical outputs; every function also has a
"given draws" form (`*_apply`) that is what the HIP kernels are compared against.

Pinned by tests/test_oracle_golden.py against vectors produced by the reference itself
(oracle/gen_golden.py imports /root/reference in the build container).
"""
import numpy as np
from scipy import signal


def rand_range(x1, x2, integer, rng=np.random):
    """RawBoost.py:14-18"""
    y = rng.uniform(low=x1, high=x2, size=(1,))
    if integer:
        y = int(y[0])
    return y


def norm_wav(x, always):
    """RawBoost.py:20-25"""
    peak = np.amax(np.abs(x))
    if always:
        x = x / peak
    elif peak > 1:
        x = x / peak
    return x


def gen_notch_coeffs(nBands, minF, maxF, minBW, maxBW, minCoeff, maxCoeff, minG, maxG, fs, rng=np.random):
    """RawBoost.py:28-48 — returns the taps b (float64, length sum(c_i) - nBands + 1 <= 501)."""
    b = 1
    for _ in range(nBands):
        fc = rand_range(minF, maxF, 0, rng)
        bw = rand_range(minBW, maxBW, 0, rng)
        c = rand_range(minCoeff, maxCoeff, 1, rng)
        if c / 2 == int(c / 2):
            c = c + 1
        f1 = fc - bw / 2
        f2 = fc + bw / 2
        if f1 <= 0:
            f1 = 1 / 1000
        if f2 >= fs / 2:
            f2 = fs / 2 - 1 / 1000
        b = np.convolve(signal.firwin(c, [float(np.ravel(f1)[0]), float(np.ravel(f2)[0])], window="hamming", fs=fs), b)
    G = rand_range(minG, maxG, 0, rng)
    _, h = signal.freqz(b, 1, fs=fs)
    b = pow(10, G / 20) * b / np.amax(np.abs(h))
    return np.asarray(b, dtype=np.float64).ravel()


def filter_fir(x, b):
    """RawBoost.py:51-56 in closed form: y[n] = sum_k b[k] * x[n + N//2 - k], N = len(b)+1,
    x zero outside [0, L) (lfilter on the end-padded signal, cropped by N/2 on both sides)."""
    x = np.asarray(x, dtype=np.float64)
    N = b.shape[0] + 1
    full = np.convolve(np.concatenate([x, np.zeros(N)]), b)[: x.shape[0] + N]  # == lfilter(b, 1, xpad)
    return full[int(N / 2): int(full.shape[0] - N / 2)]


def lnl_apply(x, taps):
    """RawBoost.py:59-69 with the N_f tap vectors given."""
    x = np.asarray(x)  # powers are taken in the input dtype (float32 clips stay float32 here), as np.power does
    y = np.zeros(x.shape[0])
    for i, b in enumerate(taps):
        y = y + filter_fir(np.power(x, i + 1), b)
    y = y - np.mean(y)
    return norm_wav(y, 0)


def lnl_draw(N_f, nBands, minF, maxF, minBW, maxBW, minCoeff, maxCoeff, minG, maxG, minBias, maxBias, fs, rng=np.random):
    taps = []
    for i in range(N_f):
        if i == 1:
            minG = minG - minBias
            maxG = maxG - maxBias
        taps.append(gen_notch_coeffs(nBands, minF, maxF, minBW, maxBW, minCoeff, maxCoeff, minG, maxG, fs, rng))
    return taps


def lnl(x, N_f, nBands, minF, maxF, minBW, maxBW, minCoeff, maxCoeff, minG, maxG, minBias, maxBias, fs, rng=np.random):
    return lnl_apply(x, lnl_draw(N_f, nBands, minF, maxF, minBW, maxBW, minCoeff, maxCoeff, minG, maxG, minBias, maxBias, fs, rng))


def isd_draw(L, P, rng=np.random):
    """RawBoost.py:74-80 draw order: beta, permutation, rand, rand."""
    beta = rand_range(0, P, 0, rng)
    n = int(L * (beta[0] / 100))
    p = rng.permutation(L)[:n]
    f_r = np.multiply((2 * rng.rand(p.shape[0])) - 1, (2 * rng.rand(p.shape[0])) - 1)
    return p, f_r


def isd_apply(x, p, f_r, g_sd):
    """RawBoost.py:76-84"""
    x = np.asarray(x)
    y = x.copy()
    r = g_sd * x[p] * f_r
    y[p] = x[p] + r
    return norm_wav(y, 0)


def isd(x, P, g_sd, rng=np.random):
    p, f_r = isd_draw(np.asarray(x).shape[0], P, rng)
    return isd_apply(x, p, f_r, g_sd)


def ssi_draw(L, SNRmin, SNRmax, nBands, minF, maxF, minBW, maxBW, minCoeff, maxCoeff, minG, maxG, fs, rng=np.random):
    """RawBoost.py:90-94 draw order: noise, notch taps, SNR."""
    noise = rng.normal(0, 1, L)
    b = gen_notch_coeffs(nBands, minF, maxF, minBW, maxBW, minCoeff, maxCoeff, minG, maxG, fs, rng)
    # SNR is drawn AFTER filtering in the reference, but filtering draws nothing
    snr = rand_range(SNRmin, SNRmax, 0, rng)
    return noise, b, float(snr[0])


def ssi_apply(x, noise, b, snr):
    """RawBoost.py:91-97"""
    x = np.asarray(x)
    n = filter_fir(noise, b)
    n = norm_wav(n, 1)
    n = n / np.linalg.norm(n, 2) * np.linalg.norm(x, 2) / 10.0 ** (0.05 * snr)
    return x + n


def ssi(x, SNRmin, SNRmax, nBands, minF, maxF, minBW, maxBW, minCoeff, maxCoeff, minG, maxG, fs, rng=np.random):
    noise, b, snr = ssi_draw(np.asarray(x).shape[0], SNRmin, SNRmax, nBands, minF, maxF, minBW, maxBW, minCoeff, maxCoeff, minG, maxG, fs, rng)
    return ssi_apply(x, noise, b, snr)


class RawBoostArgs:
    """Defaults of main.py:258-298."""
    algo = 5
    nBands, minF, maxF, minBW, maxBW = 5, 20, 8000, 100, 1000
    minCoeff, maxCoeff, minG, maxG = 10, 100, 0, 0
    minBiasLinNonLin, maxBiasLinNonLin, N_f = 5, 20, 5
    P, g_sd, SNRmin, SNRmax = 10, 2, 10, 40


def process_rawboost_feature(feature, sr, args, algo, rng=np.random):
    """datautils/asvspoof_2019_augall_3.py:377-439"""
    a = args

    def _lnl(f):
        return lnl(f, a.N_f, a.nBands, a.minF, a.maxF, a.minBW, a.maxBW, a.minCoeff, a.maxCoeff, a.minG, a.maxG,
                   a.minBiasLinNonLin, a.maxBiasLinNonLin, sr, rng)

    def _isd(f):
        return isd(f, a.P, a.g_sd, rng)

    def _ssi(f):
        return ssi(f, a.SNRmin, a.SNRmax, a.nBands, a.minF, a.maxF, a.minBW, a.maxBW, a.minCoeff, a.maxCoeff, a.minG,
                   a.maxG, sr, rng)

    if algo == 1:
        return _lnl(feature)
    if algo == 2:
        return _isd(feature)
    if algo == 3:
        return _ssi(feature)
    if algo == 4:
        return _ssi(_isd(_lnl(feature)))
    if algo == 5:
        return _isd(_lnl(feature))
    if algo == 6:
        return _ssi(_lnl(feature))
    if algo == 7:
        return _ssi(_isd(feature))
    if algo == 8:
        f1 = _lnl(feature)
        f2 = _isd(feature)
        return norm_wav(f1 + f2, 0)
    return feature
