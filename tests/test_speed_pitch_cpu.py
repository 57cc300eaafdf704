"""oracle/audio_speed_pitch.py (pydub speedup / librosa pitch_shift restated; parity UNPINNED — the packages are absent) checked
against hand-derived known answers of the published algorithms, so that the GPU parity tests compare against something that is at
least self-consistent with pydub's and librosa's documented behaviour."""
import numpy as np
import pytest

from oracle import audio_speed_pitch as SP
from oracle.audio_int16 import librosa_to_int16


def test_audioop_mul_floors_and_saturates():
    assert SP._mul(np.array([-3, 3, 30000, -30000], np.int16), 0.5).tolist() == [-2, 1, 15000, -15000]
    assert SP._mul(np.array([30000, -30000], np.int16), 2.0).tolist() == [32767, -32768]
    assert SP._add(np.array([30000, -30000], np.int16), np.array([10000, -10000], np.int16)).tolist() == [32767, -32768]


def test_millisecond_slicing_and_rounded_length():
    s = SP.Seg((np.arange(66010) % 1000).astype(np.int16), 16000)          # 4125.6 ms -> len 4126: the last slice is padded with silence
    assert len(s) == 4126
    last = s[4100:4200]
    assert len(last.f) == 26 * 16 and last.f[-6:].tolist() == [0] * 6 and last.f[-7] == (66009 % 1000)
    s = SP.Seg((np.arange(70001) % 1000).astype(np.int16), 16000)          # 4375.06 ms -> len 4375: the trailing frame is never reached
    assert len(s) == 4375 and len(s[4300:].f) == 75 * 16
    assert s[-10:].f[0] == (16 * 4365) % 1000 and len(s[:-1].f) == 16 * 4374


def test_speedup_chunk_arithmetic_for_a_positive_crossfade():
    """factor 1.05: 7 ms to remove per 150 ms, crossfade 6 ms, 157 ms chunks trimmed by 1 ms: a 4 s clip is 26 chunks ->
    25 * 2496 - 24 * 96 + 1200 frames; chunk interiors are copies, the crossfade follows pydub's per-frame ramp."""
    rs = np.random.RandomState(0)
    x = (0.2 * rs.randn(64000)).astype(np.float32)
    src = librosa_to_int16(x).astype(np.int64)
    y = SP.speed(x, 1.05)
    assert len(y) == 25 * 2496 - 24 * 96 + 1200
    assert np.array_equal(y[:2400], src[:2400])                             # chunk 0 up to its cross-fade
    assert np.array_equal(y[2496:2496 + 2304], src[16 * 157 + 96: 16 * 157 + 2400])      # chunk 1 between its two cross-fades
    assert np.array_equal(y[-1200:], src[16 * 157 * 25:])                   # the untrimmed last chunk
    j = 37                                                                  # one cross-faded frame, by hand
    g_out = 1.0 + ((10 ** (-120 / 20) - 1.0) / 96.0) * j
    g_in = 10 ** (-120 / 20) + ((1.0 - 10 ** (-120 / 20)) / 96.0) * j
    want = np.floor(src[2400 + j] * g_out) + np.floor(src[16 * 157 + j] * g_in)
    assert y[2400 + j] == np.clip(want, -32768, 32767)


def test_speedup_below_one_takes_pydubs_negative_crossfade_path():
    """factor 0.95: ms_to_remove = int(-7.5) = -7, crossfade = -8, 143 ms chunks trimmed by 1 ms; every append keeps 8 ms, re-fades
    the rest and adds 8 ms: 28 chunks of a 4 s clip -> 142 + 26 * 8 + 139 ms."""
    rs = np.random.RandomState(1)
    x = (0.2 * rs.randn(64000)).astype(np.float32)
    src = librosa_to_int16(x)
    y = SP.speed(x, 0.95)
    assert len(y) == 16 * (142 + 26 * 8 + (4000 - 27 * 143))
    assert np.array_equal(y[:128], src[:128])                               # the first 8 ms are never touched again
    assert np.array_equal(y[-16 * 139:], src[16 * 27 * 143:])
    assert np.array_equal(y[-16 * 147: -16 * 139], src[16 * (26 * 143 + 134): 16 * (26 * 143 + 142)])    # chunk 26's last 8 ms


def test_speedup_refuses_short_audio():
    with pytest.raises(ValueError):
        SP.speed(np.zeros(2000, np.float32), 1.05)


def test_stft_of_a_bin_centred_cosine_and_inverse():
    n = np.arange(8192)
    y = np.cos(2 * np.pi * 64 * n / 2048).astype(np.float32)
    D = SP.stft(y)
    assert D.shape == (1025, 17)
    mid = np.abs(D[:, 8])
    assert np.argmax(mid) == 64 and abs(mid[64] - 512.0) < 1e-2 and abs(mid[63] - 256.0) < 1e-2 and mid[70] < 1e-2      # Hann main lobe
    back = SP.istft(D, len(y))
    assert np.abs(back - y).max() < 1e-5


def test_phase_vocoder_keeps_magnitudes_and_advances_phase_at_the_bin_frequency():
    n = np.arange(16384)
    y = np.cos(2 * np.pi * 100 * n / 2048).astype(np.float32)
    D = SP.stft(y)
    S = SP.phase_vocoder(D, 0.5)
    assert S.shape[1] == 2 * D.shape[1]
    assert abs(np.abs(S[100, 20]) - np.abs(D[100, 10])) < 1e-2
    # a stationary partial at bin 100 advances by 2 pi * 100 * hop / n_fft = 50 pi per output hop: the same phase every second frame
    ph = np.angle(S[100, 10:20])
    d = np.angle(np.exp(1j * np.diff(ph)))
    assert np.abs(d).max() < 1e-2


def test_pitch_shift_moves_a_sine_by_the_requested_interval():
    t = np.arange(16000) / 16000.0
    x = (0.4 * np.sin(2 * np.pi * 500 * t)).astype(np.float32)
    for n_steps in (12, 1):
        y = SP.pitch_shift(x, 16000, n_steps)
        assert y.shape == x.shape
        seg = y[2048:-2048]
        spec = np.abs(np.fft.rfft(seg * np.hanning(len(seg))))
        peak = np.argmax(spec) * 16000.0 / len(seg)
        assert abs(peak - 500 * 2 ** (n_steps / 12)) < 3.0, (n_steps, peak)
        assert 0.3 < np.abs(seg).max() < 0.5


def test_resampler_passes_the_band_and_rejects_above_the_cutoff():
    t = np.arange(4000)
    lo = np.sin(2 * np.pi * 0.05 * t).astype(np.float32)               # 0.05 cycles / sample: far inside the pass band
    out = SP.resample_sinc(lo, 0.9)
    want = np.sin(2 * np.pi * 0.05 * np.arange(len(out)) / 0.9)
    assert len(out) == int(np.ceil(4000 * 0.9)) and np.abs(out[100:-100] - want[100:-100]).max() < 1e-3
    hi = np.sin(2 * np.pi * 0.49 * t).astype(np.float32)               # above the new Nyquist (0.45): removed
    assert np.abs(SP.resample_sinc(hi, 0.9)[100:-100]).max() < 1e-3


def test_stft_and_istft_restatements_match_scipy_signal():
    """librosa is absent, but the two transforms at either end of pitch_shift have a second public implementation in this image:
    scipy.signal.stft / istft with librosa's framing (periodic Hann, n_fft 2048, hop 512, centred with zero padding) — equal up to
    scipy's 'spectrum' scaling by the window sum.  Pins `stft`, `istft` and the window; the phase vocoder and the resampler stay
    restated (see the oracle's header)."""
    import scipy.signal as ss
    from oracle import audio_speed_pitch as SP
    rs = np.random.RandomState(3)
    y = (0.2 * rs.randn(16000 + 333)).astype(np.float32)
    w = SP.hann_periodic(SP.N_FFT)
    assert np.allclose(w, ss.get_window("hann", SP.N_FFT, fftbins=True), atol=1e-15)
    f, t, Z = ss.stft(y.astype(np.float64), window=w, nperseg=SP.N_FFT, noverlap=SP.N_FFT - SP.HOP, nfft=SP.N_FFT, boundary="zeros", padded=False,
                      return_onesided=True, scaling="spectrum")
    D = SP.stft(y)
    assert D.shape == Z.shape == (1025, 1 + len(y) // SP.HOP)
    scale = np.abs(Z).max() * w.sum()
    assert np.abs(D - Z * w.sum()).max() / scale < 1e-6
    back = SP.istft(D, len(y))
    assert np.abs(back - y).max() < 1e-5                       # the round trip is the identity away from nothing at all (COLA at hop = n_fft / 4)
    _, x2 = ss.istft(Z, window=w, nperseg=SP.N_FFT, noverlap=SP.N_FFT - SP.HOP, nfft=SP.N_FFT, boundary=True, scaling="spectrum")
    n = min(len(x2), len(y))
    assert np.abs(back[:n] - x2[:n]).max() < 1e-5
