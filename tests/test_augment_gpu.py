"""Parity of the waveform-augmentation kernels against the oracle (which is pinned to the
reference) on the golden clips.  fp32 direct-form FIR vs the reference's float64: 2e-5 absolute on
signals of O(0.1..1); int16 paths are bit-exact except where noted."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from scl_amd import augment as AUG  # noqa: E402
from oracle import audio_int16 as AI  # noqa: E402
from oracle import multiview as OM  # noqa: E402
from oracle import rawboost as RB  # noqa: E402

G = os.path.join(os.path.dirname(__file__), "golden")


def test_rawboost_algos_match_reference_goldens(dev):
    g = np.load(os.path.join(G, "rawboost.npz"))
    args = RB.RawBoostArgs()
    for seed in (0, 1):
        x = g["x_s%d" % seed]
        for algo in range(0, 9):
            np.random.seed(1000 * algo + seed)   # the host sampler consumes np.random in the reference's order
            xs = torch.from_numpy(x)[None].to(dev)
            y = AUG.rawboost_batch(xs, args, algo, 16000)
            ref = g["algo%d_s%d" % (algo, seed)]
            err = np.abs(y[0].cpu().numpy().astype(np.float64) - ref).max()
            assert err < 3e-5, "algo %d seed %d: %.3e" % (algo, seed, err)


def test_lnl_long_clip_with_normalisation(dev):
    g = np.load(os.path.join(G, "rawboost.npz"))
    np.random.seed(77)
    y = AUG.rawboost_batch(torch.from_numpy(g["x_long"])[None].to(dev), RB.RawBoostArgs(), 1, 16000)
    assert np.abs(y[0].cpu().numpy() - g["lnl_long"]).max() < 3e-5


def test_batched_clips_get_independent_draws(dev):
    rs = np.random.RandomState(0)
    xs = (0.1 * rs.randn(5, 6000)).astype(np.float32)
    np.random.seed(3)
    y = AUG.rawboost_batch(torch.from_numpy(xs).to(dev), RB.RawBoostArgs(), 5, 16000).cpu().numpy()
    np.random.seed(3)
    for i in range(5):
        ref = RB.process_rawboost_feature(xs[i], 16000, RB.RawBoostArgs(), 5)
        assert np.abs(y[i] - ref).max() < 3e-5


def test_reverb_matches_the_reference_run_up_to_rounding_ties(dev):
    """A7 against vectors produced by the reference's own ReverbAugmentor.transform (tests/golden/audio_int16.npz).  The int16 result
    is trunc(32768 * y / max|y|) of a float32 convolution; the HIP kernel sums its products in a different order than np.convolve,
    so the two float32 values can fall on different sides of an integer only where the exact value is itself within float32
    round-off of that integer.  Asserted: every sample equals the reference except at such ties, where it is off by exactly one
    LSB (or the +-32768 wrap of the peak sample) — ties are identified from the float64 convolution stored with the golden, with the
    a-priori bound 32768 * R * 2^-24 * sum|x||h| / max|y| on the float32 summation error (R taps)."""
    g = np.load(os.path.join(G, "audio_int16.npz"))
    for name in ("short", "clip16000", "long_rir"):
        sp, rir, ref = g[name + ":speech"], g[name + ":rir"], g[name + ":out"].astype(np.int64)
        got = AUG.reverb(torch.from_numpy(sp).to(dev), torch.from_numpy(rir).to(dev)).cpu().numpy().astype(np.int64)
        assert got.shape == ref.shape
        y64 = g[name + ":conv64"]
        v = 32768.0 * y64 / np.abs(y64).max()                                   # exact pre-truncation value in LSB
        bound = 32768.0 * np.convolve(np.abs(sp).astype(np.float64), np.abs(rir).astype(np.float64)) * len(rir) * 2.0 ** -24 / np.abs(y64).max()
        bound = np.minimum(np.maximum(bound, 1e-3), 0.5)
        dist = np.abs(v - np.round(v))                                           # distance to the nearest truncation boundary
        d = np.abs(got - ref)
        wrap = d >= 65535                                                        # +1.0 * 32768 wraps to -32768 (utils.py:26)
        bad = (d != 0) & ~wrap
        assert np.all(d[bad] == 1), (name, np.unique(d[bad]))
        assert np.all(dist[bad] <= 2 * bound[bad] + 1e-6), (name, float((dist[bad] - 2 * bound[bad]).max()))
        assert bad.mean() < 1e-2, (name, bad.mean())     # a fraction ~ the tie band (<= 0.5 LSB wide) of all samples sits on a tie
        print("reverb %s: %d of %d samples differ by one LSB, all at float32 rounding ties" % (name, int(bad.sum()), d.size))


def test_background_noise_and_int16_conversion(dev):
    rs = np.random.RandomState(1)
    sp = (0.1 * rs.randn(5000)).astype(np.float32)
    noise = (500 * rs.randn(4000)).astype(np.int16)
    for snr in (5, 10, 15):
        got = AUG.background_noise(torch.from_numpy(sp).to(dev), torch.from_numpy(noise).to(dev), snr).cpu().numpy()
        ref, _ = AI.background_noise(sp, noise, snr)
        assert np.array_equal(got, ref.astype(np.float32))
        # round 6: the two integer powers handed in from the host (no host <- device round trip in the pack builder): same result
        nsq = int((noise.astype(np.int64) ** 2).sum())
        got2 = AUG.background_noise(torch.from_numpy(sp).to(dev), torch.from_numpy(noise).to(dev), snr, sumsq=(AUG.host_i16_sumsq(sp), nsq)).cpu().numpy()
        assert np.array_equal(got2, got)
    edge = np.array([1.0, -1.0, 0.99999, -0.00002, 0.5, 0.999985, -0.999985], dtype=np.float32)      # the +1.0 wrap and truncation toward zero
    i16 = AUG.to_int16(torch.from_numpy(edge).to(dev)).cpu().numpy().astype(np.int64)
    assert AUG.host_i16_sumsq(edge) == int((i16 * i16).sum())
    g = np.load(os.path.join(G, "audio_int16.npz"))      # librosa_to_pydub -> pydub_to_librosa executed by the reference
    assert AUG.to_int16(torch.from_numpy(g["conv:in"]).to(dev)).cpu().numpy().tolist() == g["conv:out"].tolist()


def test_multiview_crop_matches_reference_goldens(dev):
    g = np.load(os.path.join(G, "multiview.npz"))
    for name, n in (("longer", 4), ("shorter", 4), ("exact", 3)):
        views = [g["%s_in%d" % (name, i)][:, 0].astype(np.float32) for i in range(n)]
        for rp in (False, True):
            np.random.seed(5)
            out = AUG.multiview_crop([torch.from_numpy(v).to(dev) for v in views], 2000, rp).cpu().numpy()
            for i in range(n):
                ref = g["%s_rp%d_out%d" % (name, int(rp), i)][:, 0].astype(np.float32)
                assert out[i].shape == ref.shape and np.array_equal(out[i], ref), (name, rp, i)


# ---- conf-5 augmenters: speed (pydub speedup) and pitch (librosa pitch_shift) ---------------------------------------------------------
from oracle import audio_speed_pitch as SP  # noqa: E402


@pytest.mark.parametrize("n", [64000, 70001, 66010, 9000])
def test_speed_is_bit_exact_against_the_pydub_restatement(dev, n):
    """Every regime of AudioSegment.speedup the reference's U(0.9, 1.1) draw reaches: negative crossfades (factor < 1 and the first
    0.66 % above 1), the plain-concatenation case (ms_to_remove 1), positive crossfades of 1..14 ms; clip lengths that are whole
    milliseconds, that round down (trailing frames dropped) and up (silence-padded last slice).  Integer work: bit-exact."""
    rs = np.random.RandomState(n)
    x = (0.25 * rs.randn(n)).astype(np.float32)
    x[100:140] = 1.5                                    # int16 wrap-around in the conversion, saturation in the overlays
    for f in (0.9, 0.931, 0.97, 0.999, 1.0, 1.004, 1.0067, 1.01, 1.02, 1.05, 1.0999, 1.1):
        ref = SP.speed(x, f)
        got = AUG.speed(torch.from_numpy(x).to(dev), f).cpu().numpy()
        assert got.dtype == np.float32 and got.shape == ref.shape, (f, got.shape, ref.shape)
        assert np.array_equal(got.astype(np.int32), ref.astype(np.int32)), (f, np.abs(got - ref).max())


def test_speed_raises_like_pydub_on_clips_shorter_than_two_chunks(dev):
    x = torch.zeros(2000, device=dev)
    with pytest.raises(ValueError):
        AUG.speed(x, 1.05)
    with pytest.raises(ValueError):
        SP.speed(np.zeros(2000, np.float32), 1.05)


def test_stft_phase_vocoder_istft_match_the_librosa_restatement(dev):
    """The three stages one by one against oracle/audio_speed_pitch.py (fp32 FFT vs numpy's fp64: 2e-6 of the spectrum's scale)."""
    from scl_amd import ops
    rs = np.random.RandomState(3)
    L = 16000
    y = (0.1 * rs.randn(L)).astype(np.float32)
    yd = torch.from_numpy(y).to(dev)
    nfr = ops.stft_nframes(L)
    assert nfr == 1 + L // 512
    D = torch.empty(nfr * 1025 * 2, device=dev)
    ops.stft(yd, L, D, nfr)
    Dg = torch.view_as_complex(D.view(nfr, 1025, 2)).cpu().numpy().T
    Dr = SP.stft(y)
    assert Dg.shape == Dr.shape
    assert np.abs(Dg - Dr).max() < 2e-6 * np.abs(Dr).max() * 10
    for rate in (2.0 ** (1 / 12), 2.0 ** (-1 / 12), 1.0):
        nsteps = len(np.arange(0, nfr, rate))
        Dref = SP.phase_vocoder(Dr, rate)
        Din = torch.view_as_real(torch.from_numpy(np.ascontiguousarray(Dr.T)).to(dev)).contiguous()
        Ds = torch.empty(nsteps * 1025 * 2, device=dev)
        ops.phase_vocoder(Din, nfr, rate, Ds, nsteps)
        Dsg = torch.view_as_complex(Ds.view(nsteps, 1025, 2)).cpu().numpy().T
        assert Dsg.shape == Dref.shape
        # float32 phase accumulators: a last-bit difference in an arctangent can move a phase by one float32 step of a value ~1e5
        rel = np.linalg.norm(Dsg - Dref) / np.linalg.norm(Dref)
        assert rel < 5e-3, (rate, rel)
        assert np.abs(np.abs(Dsg) - np.abs(Dref)).max() < 1e-4 * np.abs(Dref).max()
        length = int(round(L / rate))
        nuse = min(nsteps, int(np.ceil((length + 2048) / 512)))
        ws = torch.empty(nuse * 2048, device=dev)
        out = torch.empty(length, device=dev)
        Dd = torch.view_as_real(torch.from_numpy(np.ascontiguousarray(Dref.T)).to(dev)).contiguous()
        ops.istft(Dd, nuse, ws, out, length)
        ref = SP.istft(Dref, length)
        assert np.abs(out.cpu().numpy() - ref).max() < 2e-6, rate


def test_sinc_resampler_matches_its_restatement(dev):
    from scl_amd import ops
    rs = np.random.RandomState(4)
    x = (0.1 * rs.randn(6000)).astype(np.float32)
    for ratio in (2.0 ** (1 / 12), 2.0 ** (-1 / 12)):
        n_out = int(np.ceil(len(x) * ratio))
        out = torch.empty(n_out, device=dev)
        ops.resample_sinc(torch.from_numpy(x).to(dev), len(x), ratio, out, n_out)
        ref = SP.resample_sinc(x, ratio)
        assert np.abs(out.cpu().numpy() - ref).max() < 1e-6


def test_pitch_matches_the_librosa_restatement(dev):
    """End to end, int16 values out.  n_steps = 0 still goes through the phase vocoder (librosa does) and is NOT the identity."""
    t = np.arange(16000) / 16000.0
    rs = np.random.RandomState(5)
    x = (0.3 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 1230 * t + 1.0) + 0.01 * rs.randn(16000)).astype(np.float32)
    for n in (-1, 0, 1):
        ref = SP.pitch(x, n).astype(np.float64)
        got = AUG.pitch_shift(torch.from_numpy(x).to(dev), n).cpu().numpy().astype(np.float64)
        assert got.shape == ref.shape == (16000,)
        rel = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        print("pitch n_steps %+d: rel-L2 %.2e, max |diff| %.0f LSB, |diff| > 1 LSB on %.3f %%" % (n, rel, np.abs(got - ref).max(), 100 * (np.abs(got - ref) > 1).mean()))
        assert rel < 2e-4 and np.abs(got - ref).max() <= 2
        # and the pitch did move: the 440 Hz partial sits at 440 * 2^(n/12)
        spec = np.abs(np.fft.rfft(got[2048:-2048] * np.hanning(len(got) - 4096)))
        peak = np.argmax(spec[: int(800 * (len(got) - 4096) / 16000)]) * 16000.0 / (len(got) - 4096)
        assert abs(peak - 440 * 2 ** (n / 12)) < 3.0, (n, peak)
