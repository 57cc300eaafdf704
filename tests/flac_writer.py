"""A small FLAC ENCODER for the tests (test infrastructure): writes streams that exercise every branch of csrc/flac.hip — STREAMINFO
with MD5, fixed-blocksize frames with 8 / 16-bit explicit block sizes, CONSTANT / VERBATIM / FIXED(0-4) / LPC subframes, wasted
bits, Rice and Rice2 partitions incl. escape partitions, independent / left-side / right-side / mid-side stereo, CRC-8 / CRC-16.
Written from the format specification (RFC 9639); the decoder is a separate implementation in C++."""
import hashlib

import numpy as np


class BitWriter:
    def __init__(self):
        self.bits = []

    def put(self, v, n):
        v &= (1 << n) - 1
        self.bits.extend(((v >> (n - 1 - i)) & 1) for i in range(n))

    def unary(self, q):
        self.bits.extend([0] * q + [1])

    def align(self):
        while len(self.bits) % 8:
            self.bits.append(0)

    def tobytes(self):
        assert len(self.bits) % 8 == 0
        return np.packbits(np.array(self.bits, dtype=np.uint8)).tobytes()


def crc8(b):
    c = 0
    for x in b:
        c ^= x
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
    return c


def crc16(b):
    c = 0
    for x in b:
        c ^= x << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
    return c


def _utf8(n):
    if n < 0x80:
        return [n]
    out, lead, cont = [], 0xC0, 1
    while n >= (1 << (6 * cont + (6 - cont))):
        cont += 1
    lead = (0xFF << (7 - cont)) & 0xFF
    for i in range(cont):
        out.insert(0, 0x80 | (n & 0x3F))
        n >>= 6
    return [lead | n] + out


FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def _residual(w, res, blocksize, order, porder, rice2=False, escape_parts=()):
    w.put(1 if rice2 else 0, 2)
    w.put(porder, 4)
    pbits, esc = (5, 31) if rice2 else (4, 15)
    i = 0
    for part in range(1 << porder):
        count = (blocksize >> porder) - (order if part == 0 else 0)
        seg = res[i:i + count]
        i += count
        if part in escape_parts:
            nb = max(1, int(np.max(np.abs(seg))).bit_length() + 1) if len(seg) else 1
            w.put(esc, pbits)
            w.put(nb, 5)
            for v in seg:
                w.put(int(v), nb)
            continue
        u = [(int(v) << 1) if v >= 0 else ((-int(v) << 1) - 1) for v in seg]
        mean = (sum(u) / len(u)) if u else 0
        k = max(0, min(esc - 1, int(np.log2(mean + 1))))
        w.put(k, pbits)
        for x in u:
            w.unary(x >> k)
            if k:
                w.put(x & ((1 << k) - 1), k)


def _subframe(w, s, bps, kind, **kw):
    s = np.asarray(s, dtype=np.int64)
    n = len(s)
    wasted = kw.get("wasted", 0)
    if wasted:
        assert np.all(s % (1 << wasted) == 0)
        s = s >> wasted
        bps -= wasted
    w.put(0, 1)
    if kind == "constant":
        w.put(0, 6)
    elif kind == "verbatim":
        w.put(1, 6)
    elif kind == "fixed":
        w.put(8 + kw["order"], 6)
    else:
        w.put(32 + len(kw["coefs"]) - 1, 6)
    if wasted:
        w.put(1, 1)
        w.unary(wasted - 1)
    else:
        w.put(0, 1)
    if kind == "constant":
        w.put(int(s[0]), bps)
    elif kind == "verbatim":
        for v in s:
            w.put(int(v), bps)
    else:
        if kind == "fixed":
            order, coefs, shift = kw["order"], FIXED[kw["order"]], 0
        else:
            coefs, shift, order = kw["coefs"], kw["shift"], len(kw["coefs"])
        for v in s[:order]:
            w.put(int(v), bps)
        if kind == "lpc":
            prec = kw.get("precision", 12)
            w.put(prec - 1, 4)
            w.put(shift, 5)
            for c in coefs:
                w.put(int(c), prec)
        res = []
        for i in range(order, n):
            p = sum(int(coefs[j]) * int(s[i - 1 - j]) for j in range(order))
            res.append(int(s[i]) - (p >> shift))
        _residual(w, np.array(res, dtype=np.int64), n, order, kw.get("porder", 0), kw.get("rice2", False), kw.get("escape_parts", ()))


def write_flac(x, sample_rate=16000, bps=16, blocksize=4096, plan=None, stereo_mode="independent", md5=True, id3=False, unknown_total=False):
    """x: int array [n] or [n, channels].  plan(frame_index, channel) -> (kind, kwargs) picks the subframe coding."""
    x = np.asarray(x, dtype=np.int64)
    if x.ndim == 1:
        x = x[:, None]
    n, nch = x.shape
    plan = plan or (lambda f, c: ("fixed", dict(order=2, porder=2)))
    frames = []
    for fi, start in enumerate(range(0, n, blocksize)):
        blk = x[start:start + blocksize]
        bs = len(blk)
        w = BitWriter()
        w.put(0x3FFE, 14); w.put(0, 1); w.put(0, 1)
        codes = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13, 16384: 14, 32768: 15}
        bs_code = codes.get(bs, 6 if bs <= 256 else 7)
        w.put(bs_code, 4)
        w.put({8000: 4, 16000: 5, 22050: 6, 24000: 7, 32000: 8, 44100: 9, 48000: 10}.get(sample_rate, 0), 4)
        chans = [blk[:, c] for c in range(nch)]
        bpss = [bps] * nch
        if nch == 2 and stereo_mode != "independent":
            l, r = chans
            if stereo_mode == "left_side":
                w.put(8, 4); chans = [l, l - r]; bpss = [bps, bps + 1]
            elif stereo_mode == "right_side":
                w.put(9, 4); chans = [l - r, r]; bpss = [bps + 1, bps]
            else:
                w.put(10, 4); chans = [(l + r) >> 1, l - r]; bpss = [bps, bps + 1]
        else:
            w.put(nch - 1, 4)
        w.put({8: 1, 12: 2, 16: 4, 20: 5, 24: 6}.get(bps, 0), 3)
        w.put(0, 1)
        for b in _utf8(fi):
            w.put(b, 8)
        if bs_code == 6:
            w.put(bs - 1, 8)
        elif bs_code == 7:
            w.put(bs - 1, 16)
        hdr = w.tobytes()
        w.put(crc8(hdr), 8)
        for c in range(nch):
            kind, kw = plan(fi, c)
            kw = dict(kw)
            if "porder" in kw:
                while kw["porder"] > 0 and ((bs >> kw["porder"]) << kw["porder"] != bs or (bs >> kw["porder"]) <= kw.get("order", len(kw.get("coefs", [])))):
                    kw["porder"] -= 1
            _subframe(w, chans[c], bpss[c], kind, **kw)
        w.align()
        body = w.tobytes()
        frames.append(body + crc16(body).to_bytes(2, "big"))
    nbytes = (bps + 7) // 8
    raw = b"".join(int(v).to_bytes(nbytes, "little", signed=True) for v in x.reshape(-1))
    digest = hashlib.md5(raw).digest() if md5 else b"\0" * 16
    si = BitWriter()
    si.put(blocksize, 16); si.put(blocksize, 16); si.put(0, 24); si.put(0, 24)
    si.put(sample_rate, 20); si.put(nch - 1, 3); si.put(bps - 1, 5); si.put(0 if unknown_total else n, 36)      # 0: a streamed encoder that never went back to fill it in
    head = b"fLaC" + bytes([0x00]) + (34).to_bytes(3, "big") + si.tobytes() + digest
    pad = bytes([0x81]) + (8).to_bytes(3, "big") + b"\0" * 8         # a PADDING block, flagged last
    tag = (b"ID3\x04\x00\x00" + bytes([0, 0, 0, 10]) + b"\0" * 10) if id3 else b""
    return tag + head + pad + b"".join(frames)
