"""csrc/flac.hip (host code: runs without a GPU) against streams written by tests/flac_writer.py: every subframe type, residual
coding and stereo mode decodes to the encoder's input exactly, CRC / MD5 corruption is detected, and pack.load_audio reads .flac."""
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
