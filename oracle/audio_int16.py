"""ORACLE (test infrastructure).  pydub 0.25.1 / CPython audioop integer semantics and the two
augmenters built on them.  Pinning: `reverb` and `librosa_to_int16` reproduce, bit for bit, vectors produced by the REFERENCE's
own ReverbAugmentor.transform / librosa_to_pydub / pydub_to_librosa (tests/golden/audio_int16.npz, oracle/gen_golden.py::
gen_audio_int16 — pydub.AudioSegment stood in by a bare sample container).  `rms_int` / `apply_gain` / `overlay` are pinned to
CPython 3.10's own `audioop.rms` / `audioop.mul` / `audioop.add` — the C code pydub 0.25.1 calls for them — executed in the build
container on random, saturating, silent and extreme inputs, and `background_noise` to the chain of background_noise.py:40-56
built from those real primitives (tests/golden/audioop.npz, gen_golden.py::gen_audioop; tests/test_oracle_golden.py).  What
stays restated from pydub's published source because pydub itself is absent: that dBFS is 20 * log(rms / 32768, 10), that
apply_gain passes 10 ** (dB / 20) to audioop.mul, and that overlay() adds from position 0 over min(len) samples (SURVEY.md
Appendix B).  Anchored on the reference's call sites:

  datautils/audio_augmentor/utils.py:20-30      librosa_to_pydub / pydub_to_librosa
  datautils/audio_augmentor/reverb.py:33-44     ReverbAugmentor.transform
  datautils/audio_augmentor/background_noise.py:40-56  BackgroundNoiseAugmentor.transform
"""
import math

import numpy as np


def librosa_to_int16(x):
    """utils.py:26  np.array(x * (1<<15), dtype=np.int16): C cast, truncation toward zero, and
    wrap-around modulo 2^16 for out-of-range values (+1.0 -> 32768 -> -32768)."""
    v = np.trunc(np.asarray(x, dtype=np.float64) * 32768.0).astype(np.int64)
    return ((v + 32768) % 65536 - 32768).astype(np.int16)


def rms_int(samples):
    """audioop.rms: floor(sqrt(sum(x^2) / n)) in integer arithmetic."""
    s = np.asarray(samples, dtype=np.int64)
    if s.size == 0:
        return 0
    return int(math.isqrt(int((s * s).sum()) // s.size))


def dbfs(samples):
    """pydub AudioSegment.dBFS = ratio_to_db(rms / max_possible_amplitude) = 20 * log(rms / 32768, 10) (pydub/utils.py spells the
    base-10 logarithm as math.log(x, 10)); -inf when rms == 0."""
    r = rms_int(samples)
    if r == 0:
        return -float("inf")
    return 20 * math.log(r / 32768.0, 10)


def apply_gain(samples, gain_db):
    """pydub apply_gain -> audioop.mul(data, 2, 10**(dB/20)): floor(clip(x*f, -32768, 32767))."""
    f = 10.0 ** (gain_db / 20.0)
    v = np.asarray(samples, dtype=np.float64) * f
    v = np.clip(v, -32768.0, 32767.0)
    return np.floor(v).astype(np.int16)


def overlay(a, b):
    """pydub a.overlay(b) with defaults: position 0, no loop; audioop.add saturates."""
    a = np.asarray(a, dtype=np.int16)
    b = np.asarray(b, dtype=np.int16)
    out = a.copy()
    n = min(a.shape[0], b.shape[0])
    out[:n] = np.clip(a[:n].astype(np.int32) + b[:n].astype(np.int32), -32768, 32767).astype(np.int16)
    return out


def background_noise(speech_f32, noise_i16, snr_db):
    """background_noise.py:40-56 with the noise file and SNR_dB = random.randint(5, 15) given.
    Returns int16-valued samples (pydub_to_librosa keeps raw int16, utils.py:20-22)."""
    sp = librosa_to_int16(speech_f32)
    sig_db, noi_db = dbfs(sp), dbfs(noise_i16)
    gain = snr_db * noi_db / sig_db  # (sic) background_noise.py:52
    return overlay(apply_gain(sp, gain), noise_i16), gain


def reverb(speech_f32, rir_f32):
    """reverb.py:33-44: full convolution in float32 (np.convolve on float32 inputs), peak
    normalise, int16 conversion (wrap at +1.0).  Returns int16-valued samples, length L+R-1."""
    y = np.convolve(np.asarray(speech_f32, dtype=np.float32), np.asarray(rir_f32, dtype=np.float32))
    y = y / np.max(np.abs(y))
    return librosa_to_int16(y)
