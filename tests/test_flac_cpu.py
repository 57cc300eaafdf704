"""csrc/flac.hip (host code: runs without a GPU) against streams written by tests/flac_writer.py: every subframe type, residual
coding and stereo mode decodes to the encoder's input exactly, CRC / MD5 corruption is detected, and pack.load_audio reads .flac.
No third-party FLAC encoder exists in this image (`import soundfile`, `flac`, `ffmpeg`, `sox`: all absent, no .flac file on the disk), so
the streams here that were NOT written by this repository's own encoder are the three examples of the format specification (RFC 9639,
appendix D.1 - D.3, typed from the published text): their frame CRC-8s, CRC-16s and the MD5s of the audio were produced by the reference
encoder ("reference libFLAC 1.3.3 20190804" in the second one's comment block), and the decoder verifies every one of them."""
import ctypes
import os

import numpy as np
import pytest

from flac_writer import write_flac
from scl_amd import lib as L


def decode(raw, check_md5=1):
    lib = L.load()
    buf = ctypes.create_string_buffer(raw, len(raw))
    fs, ch, bits, total = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int64(0)
    L.check(lib.scl_flac_info(buf, len(raw), ctypes.byref(fs), ctypes.byref(ch), ctypes.byref(bits), ctypes.byref(total)), "info")
    out = np.empty((max(total.value, 1), ch.value), dtype=np.int32)
    got = ctypes.c_int64(0)
    L.check(lib.scl_flac_decode_i32(buf, len(raw), out.ctypes.data_as(ctypes.c_void_p), out.shape[0], ctypes.byref(got), check_md5), "decode")
    return out[: got.value], fs.value, bits.value


def speechlike(n, seed, amp=6000):
    rs = np.random.RandomState(seed)
    t = np.arange(n)
    x = amp * (np.sin(2 * np.pi * t / 57.0) * np.sin(2 * np.pi * t / 1900.0) + 0.3 * np.sin(2 * np.pi * t / 13.3)) + 40 * rs.randn(n)
    return np.round(x).astype(np.int64)


def test_mono_16_bit_like_asvspoof():
    x = speechlike(64000 + 777, 0)
    plans = [("fixed", dict(order=o, porder=p)) for o in range(5) for p in (0, 3)] + [("lpc", dict(coefs=[1876, -930, 101], shift=10, precision=12, porder=4)),
             ("lpc", dict(coefs=[3000, -2500, 900, -300, 80, -10, 3, 1], shift=11, precision=13, porder=2, rice2=True)), ("verbatim", {})]
    raw = write_flac(x, 16000, 16, 4096, plan=lambda f, c: plans[f % len(plans)])
    y, fs, bits = decode(raw)
    assert fs == 16000 and bits == 16 and y.shape == (len(x), 1)
    assert np.array_equal(y[:, 0], x)


def test_constant_wasted_bits_escape_partitions_and_odd_block_sizes():
    x = speechlike(3 * 1000 + 137, 1)
    x[1000:2000] = -1234                                   # a CONSTANT frame
    x[2000:3000] = (x[2000:3000] >> 3) << 3                # three wasted bits
    def plan(f, c):
        if f == 1:
            return "constant", {}
        if f == 2:
            return "fixed", dict(order=1, porder=1, wasted=3)
        return "fixed", dict(order=2, porder=2, escape_parts=(1, 2))
    raw = write_flac(x, 8000, 16, 1000, plan=plan, id3=True)          # 1000-sample blocks: the 16-bit explicit block size; 137: the 8-bit one
    y, fs, _ = decode(raw)
    assert fs == 8000 and np.array_equal(y[:, 0], x)


@pytest.mark.parametrize("mode", ["independent", "left_side", "right_side", "mid_side"])
@pytest.mark.parametrize("bps", [16, 24])
def test_stereo_decorrelation_modes(mode, bps):
    amp = 6000 if bps == 16 else 1_500_000
    l, r = speechlike(9000, 2, amp), speechlike(9000, 3, amp)
    r = (0.7 * l + 0.3 * r).astype(np.int64) + 1           # correlated channels, odd sums for the mid/side parity bit
    x = np.stack([l, r], axis=1)
    raw = write_flac(x, 44100, bps, 2048, plan=lambda f, c: ("fixed", dict(order=2, porder=3, rice2=(bps == 24))), stereo_mode=mode)
    y, fs, bits = decode(raw)
    assert fs == 44100 and bits == bps and np.array_equal(y, x)


def test_corruption_is_detected():
    x = speechlike(20000, 4)
    raw = bytearray(write_flac(x, 16000, 16, 4096))
    ok, _, _ = decode(bytes(raw))
    assert np.array_equal(ok[:, 0], x)
    bad = bytearray(raw); bad[len(bad) // 2] ^= 0x10        # a flipped bit inside a frame: CRC-16 (or an impossible field)
    with pytest.raises(L.SclError):
        decode(bytes(bad))
    bad = bytearray(raw); bad[4 + 4 + 18 + 3] ^= 0xFF       # a flipped byte of the MD5 signature
    with pytest.raises(L.SclError, match="MD5"):
        decode(bytes(bad))
    decode(bytes(bad), check_md5=0)
    with pytest.raises(L.SclError):
        decode(bytes(raw[: len(raw) - 100]))                # truncated stream
    with pytest.raises(L.SclError):
        decode(b"RIFF" + bytes(raw[4:]))


def test_load_audio_reads_flac_like_librosa(tmp_path):
    from scl_amd import pack
    pack.set_audio_loader(None)
    x = speechlike(16000, 5)
    p = str(tmp_path / "LA_T_1000137.flac")
    open(p, "wb").write(write_flac(x, 16000, 16, 4096))
    y = pack.load_audio(p, 16000)
    assert y.dtype == np.float32 and np.array_equal(y, (x / 32768.0).astype(np.float32))
    st = np.stack([x, -x // 2], axis=1)
    p2 = str(tmp_path / "stereo48k.flac")
    open(p2, "wb").write(write_flac(st, 48000, 16, 4096, stereo_mode="mid_side"))
    y2 = pack.load_audio(p2, 16000)                         # mono mix-down + 3:1 resampling
    assert abs(len(y2) - 16000 // 3) <= 1 and np.isfinite(y2).all()


def test_truncation_inside_a_frame_header_is_an_error_not_an_overread():
    x = speechlike(9000, 6)
    raw = write_flac(x, 16000, 16, 4096)
    first = raw.index(b"\xff\xf8", 4 + 4 + 34)                # sync code of frame 0
    for cut in range(first + 2, first + 7):                   # every cut inside the header (before / at its CRC-8 byte)
        tail = raw[:cut]
        lib = L.load()
        buf = ctypes.create_string_buffer(tail, len(tail))      # exact-size buffer: an over-read would leave it
        out = np.empty((9000, 1), dtype=np.int32)
        got = ctypes.c_int64(0)
        assert lib.scl_flac_decode_i32(buf, len(tail), out.ctypes.data_as(ctypes.c_void_p), 9000, ctypes.byref(got), 0) != 0


def test_stream_of_unknown_length_with_constant_frames_is_read_by_growing_the_output(tmp_path):
    from scl_amd import pack
    pack.set_audio_loader(None)
    x = np.zeros(40 * 4096, dtype=np.int64)                    # digital silence: ~11 bytes per 4096-sample CONSTANT frame,
    x[:4096] = speechlike(4096, 7)                            # far beyond any fixed samples-per-byte bound
    raw = write_flac(x, 16000, 16, 4096, plan=lambda f, c: ("fixed", dict(order=2, porder=2)) if f == 0 else ("constant", {}), unknown_total=True)
    assert len(raw) * 16 < len(x)
    p = str(tmp_path / "streamed.flac")
    open(p, "wb").write(raw)
    y = pack.load_audio(p, 16000)
    assert np.array_equal(y, (x / 32768.0).astype(np.float32))
    buf = ctypes.create_string_buffer(raw, len(raw))
    out = np.empty((4096, 1), dtype=np.int32)
    got = ctypes.c_int64(0)
    assert L.load().scl_flac_decode_i32(buf, len(raw), out.ctypes.data_as(ctypes.c_void_p), 4096, ctypes.byref(got), 0) != 0
    assert b"output too small" in L.load().scl_last_error()


def test_offline_cache_keeps_the_source_name_but_holds_wav_bytes(tmp_path):
    """online_aug: false on the .flac corpus (augall_3:285-291 exports out_format='wav' under the utterance's own name)."""
    import types
    import torch
    from scl_amd import pack
    pack.set_audio_loader(None)
    args = types.SimpleNamespace(aug_dir=str(tmp_path), device="cpu")
    y = torch.tensor([0.0, 1000.0, -32768.0, 32767.0, 12.0])
    got = pack._offline_cached("reverb", None, args, 16000, "/corpus/LA_T_1.flac", lambda: y.clone(), int16_values=True)
    cached = os.path.join(str(tmp_path), "reverb", "LA_T_1.flac")
    assert open(cached, "rb").read(4) == b"RIFF"
    assert torch.equal(got, y / 32768.0)
    again = pack._offline_cached("reverb", None, args, 16000, "/corpus/LA_T_1.flac", lambda: 1 / 0, int16_values=True)   # a hit: make() not called
    assert np.array_equal(np.asarray(again.cpu()), (y / 32768.0).numpy())


def test_stream_from_the_format_specification():
    """RFC 9639 appendix D.1: 44.1 kHz, 2 channels, 16 bits, one inter-channel sample, verbatim subframes with wasted bits.  The decoder
    checks the frame header's CRC-8 (0xbf), the frame's CRC-16 (0xaa9a) and STREAMINFO's MD5 of the decoded audio — three numbers this
    repository did not produce."""
    raw = bytes.fromhex("664c614380000022100010000000 0f00000f0ac442f0000000013e84b41807dc690307586a3dad1a2e0f"
                        "fff869180000bf0358fd03128baa9a".replace(" ", ""))
    assert len(raw) == 57
    pcm, fs, bits = decode(raw, check_md5=1)
    assert (fs, bits) == (44100, 16) and pcm.shape == (1, 2)
    assert pcm.tolist() == [[25588, 10416]]
    bad = bytearray(raw)
    bad[-4] ^= 1                      # one bit of the second subframe: the frame CRC-16 must catch it
    with pytest.raises(L.SclError):
        decode(bytes(bad))


def test_second_and_third_streams_from_the_format_specification():
    """RFC 9639 appendix D.2: STREAMINFO + SEEKTABLE + VORBIS_COMMENT + PADDING, two frames (16 and 3 inter-channel samples), side-channel
    stereo, fixed predictors of order 1 / 2 / 3, Rice partitions, a wasted-bits verbatim frame.  D.3: 32 kHz, one channel, 8 bits, one
    frame of 24 samples with a quantised linear predictor of order 3 (precision 4, shift 2) and an escaped-free Rice partition."""
    d2 = bytes.fromhex(
        "66 4c 61 43 00 00 00 22 00 10 00 10 00 00 17 00 00 44 0a c4 42 f0 00 00 00 13 d5 b0 56 49 75 e9 8b 8d 8b 93 04 22 75 7b 81 03"
        "03 00 00 12 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 10"
        "04 00 00 3a 20 00 00 00 72 65 66 65 72 65 6e 63 65 20 6c 69 62 46 4c 41 43 20 31 2e 33 2e 33 20 32 30 31 39 30 38 30 34"
        "01 00 00 00 0e 00 00 00 54 49 54 4c 45 3d d7 a9 d7 9c d7 95 d7 9d"
        "81 00 00 06 00 00 00 00 00 00"
        "ff f8 69 98 00 0f 99 12 08 67 01 62 3d 14 42 99 8f 5d f7 0d 6f e0 0c 17 ca eb 21 00 0e e7 a7 7a 24 a1 59 0c 12 17 b6 03 09 7b 78 4f"
        "aa 9a 33 d2 85 e0 70 ad 5b 1b 48 51 b4 01 0d 99 d2 cd 1a 68 f1 e6 b8 10"
        "ff f8 69 18 01 02 a4 02 c3 82 c4 0b c1 4a 03 ee 48 dd 03 b6 7c 13 30")
    assert len(d2) == 227
    pcm, fs, bits = decode(d2, check_md5=1)
    assert (fs, bits) == (44100, 16) and pcm.shape == (19, 2)
    assert pcm[:, 0].tolist() == [10372, 18041, 14942, 17876, 15627, 17899, 16242, 18077, 16824, 18263, 17295, -14418, -15201, -14508, -15195,
                                  -14818, -15486, -15349, -16054]
    assert pcm[:, 1].tolist() == [6070, 10545, 8743, 10449, 9143, 10463, 9502, 10569, 9840, 10680, 10113, -8428, -8895, -8476, -8896, -8653,
                                  -9072, -8958, -9410]
    d3 = bytes.fromhex("66 4c 61 43 80 00 00 22 10 00 10 00 00 00 1f 00 00 1f 07 d0 00 70 00 00 00 18 f8 f9 e3 96 f5 cb cf c6 dc 80 7f 99 77 90 6b 32"
                       "ff f8 68 02 00 17 e9 44 00 4f 6f 31 3d 10 47 d2 27 cb 6d 09 08 31 45 2b dc 28 22 22 80 57 a3")
    assert len(d3) == 73
    pcm, fs, bits = decode(d3, check_md5=1)
    assert (fs, bits) == (32000, 8) and pcm.shape == (24, 1)
    assert pcm[:, 0].tolist() == [0, 79, 111, 78, 8, -61, -90, -68, -13, 42, 67, 53, 13, -27, -46, -38, -12, 14, 24, 19, 6, -4, -5, 0]


@pytest.mark.parametrize("nch,bps,mode", [(1, 16, "independent"), (2, 16, "mid_side"), (2, 24, "left_side"), (3, 12, "independent"), (1, 20, "independent")])
def test_one_pass_mono_float_decode_equals_the_numpy_formulation(nch, bps, mode):
    """scl_flac_decode_mono_f32 (round 6: the reader of the pack builder and of the scoring loop) against the two numpy passes it replaces —
    int32 image -> float32 / 2^(bits-1) -> mean over the channel axis, i.e. librosa.load(mono=True) on soundfile's floats — bit for bit,
    with the MD5 of STREAMINFO checked on the way (computed over the frames as they are decoded) and a capacity that is too small refused."""
    n = 10000
    amp = (1 << (bps - 1)) // 3
    x = np.stack([speechlike(n, 30 + c, amp=amp) * (1 if c % 2 == 0 else -1) // (c + 1) for c in range(nch)], axis=1)
    raw = write_flac(x if nch > 1 else x[:, 0], 16000, bps, 1152, stereo_mode=mode)
    ints, fs, bits = decode(raw)
    ref = ints.astype(np.float32) / np.float32(1 << (bits - 1))
    ref = ref.mean(axis=1) if nch > 1 else ref[:, 0]
    lib = L.load()
    buf = ctypes.create_string_buffer(raw, len(raw))
    out = np.full(n + 7, np.nan, dtype=np.float32)
    got = ctypes.c_int64(0)
    L.check(lib.scl_flac_decode_mono_f32(buf, len(raw), out.ctypes.data_as(ctypes.c_void_p), n + 7, ctypes.byref(got), 1), "decode f32")
    assert got.value == n and np.array_equal(out[:n], ref) and np.isnan(out[n:]).all()
    assert lib.scl_flac_decode_mono_f32(buf, len(raw), out.ctypes.data_as(ctypes.c_void_p), n - 1, ctypes.byref(got), 1) != 0
    assert b"output too small" in lib.scl_last_error()
    bad = bytearray(raw); bad[len(bad) // 2] ^= 0x10
    cb = ctypes.create_string_buffer(bytes(bad), len(bad))
    assert lib.scl_flac_decode_mono_f32(cb, len(bad), out.ctypes.data_as(ctypes.c_void_p), n + 7, ctypes.byref(got), 1) != 0
