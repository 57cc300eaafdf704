"""ORACLE (test infrastructure).  The two conf-5 augmenters whose arithmetic lives in third-party packages that are absent from
this image and from /root/reference:

  speed   datautils/audio_augmentor/speed.py:29-33   -> pydub 0.25.1 (00_envsetup.sh:48) AudioSegment.speedup(speed_factor)
  pitch   datautils/audio_augmentor/pitch.py:31-38   -> librosa 0.10.0 (00_envsetup.sh:45) effects.pitch_shift(data, sr, n_steps)

PARITY PARTLY PINNED: neither package can be imported here.  The speed path's ARITHMETIC is pinned: pydub does it in CPython's
`audioop` (mul / add), which is in this container's standard library — tests/golden/audioop.npz holds the whole speedup()
sequence below executed with the real audioop.mul / audioop.add in place of `_mul` / `_add` (factors below and above 1, a
speech-like and a full-scale clip; oracle/gen_golden.py::gen_audioop, tests/test_oracle_golden.py).  What remains a restatement
of pydub's published source is the sequence itself (slicing, fade steps, looped overlay, append).  Of the pitch path (librosa) the
two transforms at its ends, `stft` and `istft` (and the periodic Hann window), are pinned to scipy.signal's stft / istft with
librosa's framing (tests/test_speed_pitch_cpu.py); the phase vocoder and the resampler stay UNPINNED restatements.  Both are anchored on the reference's call sites above and on its converters (utils.py:20-30, pinned in
oracle/audio_int16.py).

`Seg` restates the part of pydub.AudioSegment that speedup() touches, for mono 16-bit audio, statement by statement: millisecond
slicing (`__getitem__`, `_parse_position`, rounded `len`), `fade` (one gain step per millisecond above 100 ms, per frame below),
`overlay` (audioop.add, saturating, optional looping), `append` (cross-fade), `make_chunks`, `speedup`.  audioop.mul is
floor(clip(x * f)) (CPython audioop.c fbound()).  Note what speedup() does for playback_speed < 1 (half of the reference's draws,
speed factor ~ U(0.9, 1.1)): ms_to_remove_per_chunk and hence `crossfade` come out NEGATIVE, append() then keeps the first |c| ms
of the running output, fades the whole rest out under a looped fade-in of the next chunk and appends that chunk's last |c| ms —
the result grows by |c| ms per chunk (a 4 s clip comes back ~0.6 s long).  That is the library's behaviour for this argument
range and therefore the reference's; it is restated, not repaired.

`pitch_shift` restates librosa 0.10.0: stft (n_fft 2048, hop 512, periodic Hann, centred, zero padding) -> phase_vocoder(rate =
2^(-n_steps/12)) -> istft(length = round(len / rate)) -> resample(orig_sr = sr / rate -> sr) -> fix_length.  The resampler is the
one deliberate deviation: librosa calls soxr ("soxr_hq"), whose filter is not a published algorithm; here it is a Kaiser-windowed
sinc interpolator (32 zero crossings, beta 14.77, cut-off 0.95 x the lower Nyquist).  The pitch path is floating point; tests
compare the HIP path with this restatement within +-1 int16 step.
"""
import math

import numpy as np

from .audio_int16 import librosa_to_int16


# ---- pydub -----------------------------------------------------------------------------------------------------------------------------
def _mul(frames, factor):
    """audioop.mul(fragment, 2, factor): floor(clip(x * factor)) per sample (audioop.c fbound)."""
    v = np.asarray(frames, dtype=np.float64) * float(factor)
    return np.floor(np.clip(v, -32768.0, 32767.0)).astype(np.int16)


def _add(a, b):
    """audioop.add(a, b, 2): saturating; the fragments have equal length."""
    return np.clip(a.astype(np.int32) + b.astype(np.int32), -32768, 32767).astype(np.int16)


def db_to_float(db):
    return 10 ** (float(db) / 20)


class Seg:
    """pydub.AudioSegment, mono, sample_width 2 (frame_width 2)."""

    def __init__(self, frames, rate=16000):
        self.f = np.asarray(frames, dtype=np.int16)
        self.rate = rate

    def _spawn(self, frames):
        return Seg(frames, self.rate)

    def frame_count(self, ms=None):
        if ms is not None:
            return ms * (self.rate / 1000.0)
        return float(len(self.f))

    def __len__(self):
        return round(1000 * (self.frame_count() / self.rate))

    def _parse_position(self, val):
        if val < 0:
            val = len(self) - abs(val)
        val = self.frame_count(ms=len(self)) if val == float("inf") else self.frame_count(ms=val)
        return int(val)

    def __getitem__(self, ms):
        if isinstance(ms, slice):
            start = ms.start if ms.start is not None else 0
            end = ms.stop if ms.stop is not None else len(self)
            start = min(start, len(self))
            end = min(end, len(self))
        else:
            start, end = ms, ms + 1
        start = self._parse_position(start)
        end = self._parse_position(end)
        data = self.f[start:end] if end >= start else self.f[0:0]
        missing = (end - start) - len(data)
        if missing > 0:
            if missing > self.frame_count(ms=2):
                raise ValueError("TooManyMissingFrames")
            silence = np.zeros(1 if len(data) else 0, dtype=np.int16)        # audioop.mul(data[:frame_width], 2, 0)
            data = np.concatenate([data] + [silence] * missing)
        return self._spawn(data)

    def get_frame(self, index):
        return self.f[index:index + 1]

    def fade(self, to_gain=0, from_gain=0, start=None, end=None):
        if to_gain == 0 and from_gain == 0:
            return self
        start = min(len(self), start) if start is not None else None
        end = min(len(self), end) if end is not None else None
        if start is not None and start < 0:
            start += len(self)
        if end is not None and end < 0:
            end += len(self)
        duration = end - start
        from_power = db_to_float(from_gain)
        out = []
        before = self[:start].f
        if from_gain != 0:
            before = _mul(before, from_power)
        out.append(before)
        gain_delta = db_to_float(to_gain) - from_power
        if duration > 100:
            scale_step = gain_delta / duration
            for i in range(duration):
                volume_change = from_power + (scale_step * i)
                out.append(_mul(self[start + i].f, volume_change))
        else:
            start_frame = self.frame_count(ms=start)
            end_frame = self.frame_count(ms=end)
            fade_frames = end_frame - start_frame
            scale_step = gain_delta / fade_frames            # ZeroDivisionError on an empty fade, as pydub
            for i in range(int(fade_frames)):
                volume_change = from_power + (scale_step * i)
                out.append(_mul(self.get_frame(int(start_frame + i)), volume_change))
        after = self[end:].f
        if to_gain != 0:
            after = _mul(after, db_to_float(to_gain))
        out.append(after)
        return self._spawn(np.concatenate(out))

    def overlay(self, seg, position=0, loop=False):
        times = -1 if loop else 1
        out = [self[:position].f]
        seg1 = self[position:].f
        seg2 = seg.f
        pos = 0
        while times:
            remaining = max(0, len(seg1) - pos)
            if len(seg2) >= remaining:
                seg2 = seg2[:remaining]
                times = 1
            elif len(seg2) == 0:
                raise ValueError("overlay(loop=True) of an empty segment never terminates in pydub")
            out.append(_add(seg1[pos:pos + len(seg2)], seg2))
            pos += len(seg2)
            times -= 1
        out.append(seg1[pos:])
        return self._spawn(np.concatenate(out))

    def append(self, seg, crossfade=100):
        if not crossfade:
            return self._spawn(np.concatenate([self.f, seg.f]))
        if crossfade > len(self) or crossfade > len(seg):
            raise ValueError("Crossfade is longer than the segment")
        xf = self[-crossfade:].fade(to_gain=-120, start=0, end=float("inf"))
        xf = xf.overlay(seg[:crossfade].fade(from_gain=-120, start=0, end=float("inf")), position=0, loop=True)       # xf *= ...
        return self._spawn(np.concatenate([self[:-crossfade].f, xf.f, seg[crossfade:].f]))


def make_chunks(seg, chunk_length):
    n = math.ceil(len(seg) / float(chunk_length))
    return [seg[i * chunk_length:(i + 1) * chunk_length] for i in range(int(n))]


def speedup(seg, playback_speed=1.5, chunk_size=150, crossfade=25):
    """pydub/effects.py speedup()."""
    atk = 1.0 / playback_speed
    if playback_speed < 2.0:
        ms_to_remove_per_chunk = int(chunk_size * (1 - atk) / atk)
    else:
        ms_to_remove_per_chunk = int(chunk_size)
        chunk_size = int(atk * chunk_size / (1 - atk))
    crossfade = min(crossfade, ms_to_remove_per_chunk - 1)
    chunks = make_chunks(seg, chunk_size + ms_to_remove_per_chunk)
    if len(chunks) < 2:
        raise ValueError("Could not speed up AudioSegment, it was too short")
    ms_to_remove_per_chunk -= crossfade
    last_chunk = chunks[-1]
    chunks = [chunk[:-ms_to_remove_per_chunk] for chunk in chunks[:-1]]
    out = chunks[0]
    for chunk in chunks[1:]:
        out = out.append(chunk, crossfade=crossfade)
    return out.append(last_chunk, crossfade=0)          # out += last_chunk


def speed(speech_f32, speed_factor, sr=16000):
    """SpeedAugmentor.load + transform (speed.py:20-33) then pydub_to_librosa: int16-VALUED samples."""
    return speedup(Seg(librosa_to_int16(speech_f32), sr), speed_factor).f


# ---- librosa ---------------------------------------------------------------------------------------------------------------------------
N_FFT, HOP = 2048, 512


def hann_periodic(n):
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n))


def stft(y):
    """librosa.stft(y) defaults: centred frames over the zero-padded signal, 1 + len // hop frames; complex64 [1025, frames]."""
    y = np.asarray(y, dtype=np.float32)
    ypad = np.pad(y.astype(np.float64), (N_FFT // 2, N_FFT // 2))
    nfr = 1 + len(y) // HOP
    w = hann_periodic(N_FFT)
    frames = np.stack([ypad[t * HOP: t * HOP + N_FFT] * w for t in range(nfr)], axis=1)
    return np.fft.rfft(frames, axis=0).astype(np.complex64)


def phase_vocoder(D, rate):
    """librosa.phase_vocoder (core/spectrum.py), hop = n_fft / 4."""
    nb, nfr = D.shape
    time_steps = np.arange(0, nfr, rate, dtype=np.float64)
    out = np.zeros((nb, len(time_steps)), dtype=D.dtype)
    phi_advance = np.linspace(0, np.pi * HOP, nb)
    phase_acc = np.angle(D[:, 0])                               # float32, accumulated in float32 as librosa's in-place +=
    Dp = np.pad(D, [(0, 0), (0, 2)], mode="constant")
    for t, step in enumerate(time_steps):
        cols = Dp[:, int(step): int(step + 2)]
        alpha = np.mod(step, 1.0)
        mag = (1.0 - alpha) * np.abs(cols[:, 0]) + alpha * np.abs(cols[:, 1])
        out[:, t] = mag * np.exp(1j * phase_acc)                # util.phasor(phase_acc, mag=mag)
        dphase = np.angle(cols[:, 1]) - np.angle(cols[:, 0]) - phi_advance
        dphase = dphase - 2.0 * np.pi * np.round(dphase / (2.0 * np.pi))
        phase_acc += phi_advance + dphase
    return out


def istft(D, length):
    """librosa.istft(D, length=length): windowed inverse frames overlap-added, divided by the window sum of squares where it is
    not tiny, centre padding removed, fixed to `length`."""
    nb, nfr = D.shape
    w = hann_periodic(N_FFT)
    nfr = min(nfr, int(np.ceil((length + N_FFT) / HOP)))
    y = np.zeros(N_FFT + HOP * (nfr - 1))
    wss = np.zeros_like(y)
    fr = np.fft.irfft(D[:, :nfr].astype(np.complex128), n=N_FFT, axis=0)
    for t in range(nfr):
        y[t * HOP: t * HOP + N_FFT] += w * fr[:, t]
        wss[t * HOP: t * HOP + N_FFT] += w * w
    nz = wss > np.finfo(np.float32).tiny
    y[nz] /= wss[nz]
    y = y[N_FFT // 2:]
    return fix_length(y, length).astype(np.float32)


def fix_length(y, size):
    if len(y) >= size:
        return y[:size]
    return np.concatenate([y, np.zeros(size - len(y), dtype=y.dtype)])


SINC_ZEROS, SINC_BETA, SINC_ROLLOFF = 32, 14.769656459379492, 0.95


def resample_sinc(y, ratio):
    """Band-limited resampling by `ratio` = target_sr / orig_sr (stands where librosa calls soxr_hq, see the module header):
    out[n] = sum_k y[k] h(n / ratio - k), h(u) = 2 fc sinc(2 fc u) kaiser(u / W), fc = 0.5 * rolloff * min(1, ratio),
    W = zeros / (2 fc) the half-width; n_out = ceil(len * ratio) (librosa.resample)."""
    y = np.asarray(y, dtype=np.float64)
    n_out = int(np.ceil(len(y) * ratio))
    fc = 0.5 * SINC_ROLLOFF * min(1.0, ratio)
    W = SINC_ZEROS / (2.0 * fc)
    out = np.zeros(n_out)
    i0b = np.i0(SINC_BETA)
    for n in range(n_out):
        t = n / ratio
        k0, k1 = max(0, int(math.ceil(t - W))), min(len(y) - 1, int(math.floor(t + W)))
        if k1 < k0:
            continue
        u = t - np.arange(k0, k1 + 1)
        win = np.i0(SINC_BETA * np.sqrt(np.maximum(0.0, 1.0 - (u / W) ** 2))) / i0b
        out[n] = np.dot(y[k0:k1 + 1], 2.0 * fc * np.sinc(2.0 * fc * u) * win)
    return out.astype(np.float32)


def time_stretch(y, rate):
    return istft(phase_vocoder(stft(y), rate), int(round(len(y) / rate)))


def pitch_shift(y, sr, n_steps):
    """librosa.effects.pitch_shift(y, sr=sr, n_steps=n_steps) with the defaults the reference uses (pitch.py:36)."""
    rate = 2.0 ** (-float(n_steps) / 12)
    ys = time_stretch(np.asarray(y, dtype=np.float32), rate)
    if rate != 1.0:                                  # librosa.resample returns its input when orig_sr == target_sr
        ys = resample_sinc(ys, rate)
    return fix_length(ys, len(y))


def pitch(speech_f32, n_steps, sr=16000):
    """PitchAugmentor.transform (pitch.py:31-38) + pydub_to_librosa: int16-VALUED samples."""
    return librosa_to_int16(pitch_shift(speech_f32, sr, n_steps))
