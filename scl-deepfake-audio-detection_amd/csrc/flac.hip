// flac.hip — host-side FLAC decoder of the pack builder's file reader (no device code in this file).
//
// The reference reads every utterance with librosa.load (datautils/asvspoof_2019_augall_3.py:97-100) and MUSAN / RIR files with
// AudioSegment.from_file / librosa.load (audio_augmentor/background_noise.py:22-28, reverb.py:30-31); ASVspoof ships as 16 kHz mono
// 16-bit FLAC.  libsndfile / ffmpeg are not part of this image, so the decoder lives here: the whole of the FLAC subset format
// (RFC 9639) that encoders produce — STREAMINFO, fixed- and variable-blocksize frames, CONSTANT / VERBATIM / FIXED / LPC subframes,
// wasted bits, Rice and Rice2 partitioned residuals with escape partitions, left-side / right-side / mid-side decorrelation,
// 4..32 bits per sample, up to 8 channels.  Every frame header's CRC-8 and every frame's CRC-16 are checked, and — on request —
// the MD5 of the decoded samples against the one the encoder stored in STREAMINFO, so a decoding error cannot pass silently.
#include <stdint.h>
#include <string.h>
#include <vector>
#include "common.h"

namespace {

struct Bits {
    const uint8_t* p; size_t n, pos = 0; uint64_t acc = 0; int have = 0; bool bad = false;
    Bits(const uint8_t* p_, size_t n_) : p(p_), n(n_) {}
    // after fill(): have >= 57.  Fast path (round 6: the byte loop was a third of the decoder's time): one unaligned big-endian 8-byte
    // load, of which the (64 - have) / 8 whole bytes that fit are taken.
    inline void fill() {
        if (have > 56) return;
        if (pos + 8 <= n) {
            uint64_t w;
            memcpy(&w, p + pos, 8);
            w = __builtin_bswap64(w);
            const int take = (64 - have) >> 3;          // 1 .. 8 bytes
            const int tot = have + 8 * take;            // 57 .. 64 valid bits afterwards; the partial byte behind them stays unread
            acc |= (w >> have) & (tot >= 64 ? ~0ull : ~(~0ull >> tot));
            pos += (size_t)take;
            have += 8 * take;
            return;
        }
        while (have <= 56) {
            uint64_t b = 0;
            if (pos < n) b = p[pos]; else if (pos >= n + 8) { bad = true; }
            ++pos;
            acc |= b << (56 - have);
            have += 8;
        }
    }
    inline uint32_t read(int k) {            // k <= 32
        if (k == 0) return 0;
        fill();
        const uint32_t v = (uint32_t)(acc >> (64 - k));
        acc <<= k; have -= k;
        return v;
    }
    inline int32_t read_signed(int k) {
        if (k == 0) return 0;
        const uint32_t v = read(k);
        return (int32_t)(v << (32 - k)) >> (32 - k);
    }
    inline uint32_t read_unary() {           // number of 0 bits before the next 1 bit
        uint32_t q = 0;
        for (;;) {
            fill();
            if (acc == 0) { q += have; acc = 0; have = 0; if (pos > n + 8) { bad = true; return q; } continue; }
            const int z = __builtin_clzll(acc);
            if (z >= have) { q += have; acc = 0; have = 0; continue; }
            q += z;
            acc <<= (z + 1); have -= (z + 1);
            return q;
        }
    }
    // one Rice-coded residual with parameter k (< 32): unary quotient, stop bit, k remainder bits — from ONE refill when they fit
    // the 57+ bits a fill() guarantees (quotients are short: the encoder picks k so that they average ~1)
    inline int32_t read_rice(int k) {
        fill();
        uint32_t u;
        if (acc != 0) {
            const int z = __builtin_clzll(acc);
            if (z + 1 + k <= have) {
                const uint64_t a = acc << (z + 1);
                const uint32_t rem = k ? (uint32_t)(a >> (64 - k)) : 0u;
                acc = k ? (a << k) : a;
                have -= z + 1 + k;
                u = ((uint32_t)z << k) | rem;
                return (int32_t)(u >> 1) ^ -(int32_t)(u & 1);
            }
        }
        const uint32_t q = read_unary();
        u = (q << k) | read(k);
        return (int32_t)(u >> 1) ^ -(int32_t)(u & 1);
    }
    inline size_t byte_pos() const { return pos - (size_t)(have / 8); }     // valid when aligned
    inline void align() { const int r = have & 7; acc <<= r; have -= r; }
};

uint8_t crc8(const uint8_t* p, size_t n) {
    uint8_t c = 0;
    for (size_t i = 0; i < n; ++i) {
        c ^= p[i];
        for (int b = 0; b < 8; ++b) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : (c << 1));
    }
    return c;
}
uint16_t crc16(const uint8_t* p, size_t n) {
    static uint16_t tab[256]; static bool init = false;
    if (!init) {
        for (int i = 0; i < 256; ++i) {
            uint16_t c = (uint16_t)(i << 8);
            for (int b = 0; b < 8; ++b) c = (uint16_t)((c & 0x8000) ? (c << 1) ^ 0x8005 : (c << 1));
            tab[i] = c;
        }
        init = true;
    }
    uint16_t c = 0;
    for (size_t i = 0; i < n; ++i) c = (uint16_t)((c << 8) ^ tab[(c >> 8) ^ p[i]]);
    return c;
}

// ---- MD5 (RFC 1321) ------------------------------------------------------------------------------------------------------------------
struct Md5 {
    uint32_t h[4] = {0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u};
    uint8_t buf[64]; size_t fill = 0; uint64_t total = 0;
    static inline uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
    void block(const uint8_t* m) {
        uint32_t w[16];
        memcpy(w, m, 64);      // little-endian host (x86-64 / the image's only target)
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3];
#define MD5_F(x, y, z) ((z) ^ ((x) & ((y) ^ (z))))
#define MD5_G(x, y, z) ((y) ^ ((z) & ((x) ^ (y))))
#define MD5_H(x, y, z) ((x) ^ (y) ^ (z))
#define MD5_I(x, y, z) ((y) ^ ((x) | ~(z)))
#define MD5_STEP(f, a, b, c, d, x, t, s) (a) += f((b), (c), (d)) + (x) + (t); (a) = rol((a), (s)); (a) += (b);
        MD5_STEP(MD5_F, a, b, c, d, w[0], 0xd76aa478, 7) MD5_STEP(MD5_F, d, a, b, c, w[1], 0xe8c7b756, 12) MD5_STEP(MD5_F, c, d, a, b, w[2], 0x242070db, 17) MD5_STEP(MD5_F, b, c, d, a, w[3], 0xc1bdceee, 22)
        MD5_STEP(MD5_F, a, b, c, d, w[4], 0xf57c0faf, 7) MD5_STEP(MD5_F, d, a, b, c, w[5], 0x4787c62a, 12) MD5_STEP(MD5_F, c, d, a, b, w[6], 0xa8304613, 17) MD5_STEP(MD5_F, b, c, d, a, w[7], 0xfd469501, 22)
        MD5_STEP(MD5_F, a, b, c, d, w[8], 0x698098d8, 7) MD5_STEP(MD5_F, d, a, b, c, w[9], 0x8b44f7af, 12) MD5_STEP(MD5_F, c, d, a, b, w[10], 0xffff5bb1, 17) MD5_STEP(MD5_F, b, c, d, a, w[11], 0x895cd7be, 22)
        MD5_STEP(MD5_F, a, b, c, d, w[12], 0x6b901122, 7) MD5_STEP(MD5_F, d, a, b, c, w[13], 0xfd987193, 12) MD5_STEP(MD5_F, c, d, a, b, w[14], 0xa679438e, 17) MD5_STEP(MD5_F, b, c, d, a, w[15], 0x49b40821, 22)
        MD5_STEP(MD5_G, a, b, c, d, w[1], 0xf61e2562, 5) MD5_STEP(MD5_G, d, a, b, c, w[6], 0xc040b340, 9) MD5_STEP(MD5_G, c, d, a, b, w[11], 0x265e5a51, 14) MD5_STEP(MD5_G, b, c, d, a, w[0], 0xe9b6c7aa, 20)
        MD5_STEP(MD5_G, a, b, c, d, w[5], 0xd62f105d, 5) MD5_STEP(MD5_G, d, a, b, c, w[10], 0x02441453, 9) MD5_STEP(MD5_G, c, d, a, b, w[15], 0xd8a1e681, 14) MD5_STEP(MD5_G, b, c, d, a, w[4], 0xe7d3fbc8, 20)
        MD5_STEP(MD5_G, a, b, c, d, w[9], 0x21e1cde6, 5) MD5_STEP(MD5_G, d, a, b, c, w[14], 0xc33707d6, 9) MD5_STEP(MD5_G, c, d, a, b, w[3], 0xf4d50d87, 14) MD5_STEP(MD5_G, b, c, d, a, w[8], 0x455a14ed, 20)
        MD5_STEP(MD5_G, a, b, c, d, w[13], 0xa9e3e905, 5) MD5_STEP(MD5_G, d, a, b, c, w[2], 0xfcefa3f8, 9) MD5_STEP(MD5_G, c, d, a, b, w[7], 0x676f02d9, 14) MD5_STEP(MD5_G, b, c, d, a, w[12], 0x8d2a4c8a, 20)
        MD5_STEP(MD5_H, a, b, c, d, w[5], 0xfffa3942, 4) MD5_STEP(MD5_H, d, a, b, c, w[8], 0x8771f681, 11) MD5_STEP(MD5_H, c, d, a, b, w[11], 0x6d9d6122, 16) MD5_STEP(MD5_H, b, c, d, a, w[14], 0xfde5380c, 23)
        MD5_STEP(MD5_H, a, b, c, d, w[1], 0xa4beea44, 4) MD5_STEP(MD5_H, d, a, b, c, w[4], 0x4bdecfa9, 11) MD5_STEP(MD5_H, c, d, a, b, w[7], 0xf6bb4b60, 16) MD5_STEP(MD5_H, b, c, d, a, w[10], 0xbebfbc70, 23)
        MD5_STEP(MD5_H, a, b, c, d, w[13], 0x289b7ec6, 4) MD5_STEP(MD5_H, d, a, b, c, w[0], 0xeaa127fa, 11) MD5_STEP(MD5_H, c, d, a, b, w[3], 0xd4ef3085, 16) MD5_STEP(MD5_H, b, c, d, a, w[6], 0x04881d05, 23)
        MD5_STEP(MD5_H, a, b, c, d, w[9], 0xd9d4d039, 4) MD5_STEP(MD5_H, d, a, b, c, w[12], 0xe6db99e5, 11) MD5_STEP(MD5_H, c, d, a, b, w[15], 0x1fa27cf8, 16) MD5_STEP(MD5_H, b, c, d, a, w[2], 0xc4ac5665, 23)
        MD5_STEP(MD5_I, a, b, c, d, w[0], 0xf4292244, 6) MD5_STEP(MD5_I, d, a, b, c, w[7], 0x432aff97, 10) MD5_STEP(MD5_I, c, d, a, b, w[14], 0xab9423a7, 15) MD5_STEP(MD5_I, b, c, d, a, w[5], 0xfc93a039, 21)
        MD5_STEP(MD5_I, a, b, c, d, w[12], 0x655b59c3, 6) MD5_STEP(MD5_I, d, a, b, c, w[3], 0x8f0ccc92, 10) MD5_STEP(MD5_I, c, d, a, b, w[10], 0xffeff47d, 15) MD5_STEP(MD5_I, b, c, d, a, w[1], 0x85845dd1, 21)
        MD5_STEP(MD5_I, a, b, c, d, w[8], 0x6fa87e4f, 6) MD5_STEP(MD5_I, d, a, b, c, w[15], 0xfe2ce6e0, 10) MD5_STEP(MD5_I, c, d, a, b, w[6], 0xa3014314, 15) MD5_STEP(MD5_I, b, c, d, a, w[13], 0x4e0811a1, 21)
        MD5_STEP(MD5_I, a, b, c, d, w[4], 0xf7537e82, 6) MD5_STEP(MD5_I, d, a, b, c, w[11], 0xbd3af235, 10) MD5_STEP(MD5_I, c, d, a, b, w[2], 0x2ad7d2bb, 15) MD5_STEP(MD5_I, b, c, d, a, w[9], 0xeb86d391, 21)
#undef MD5_STEP
#undef MD5_F
#undef MD5_G
#undef MD5_H
#undef MD5_I
        h[0] += a; h[1] += b; h[2] += c; h[3] += d;
    }
    void update(const uint8_t* p, size_t n) {
        total += n;
        if (fill == 0) while (n >= 64) { block(p); p += 64; n -= 64; }      // whole blocks straight from the source
        while (n) {
            const size_t k = 64 - fill < n ? 64 - fill : n;
            memcpy(buf + fill, p, k); fill += k; p += k; n -= k;
            if (fill == 64) { block(buf); fill = 0; }
        }
    }
    void final(uint8_t out[16]) {
        const uint64_t bits = total * 8;
        const uint8_t one = 0x80, zero = 0;
        update(&one, 1);
        while (fill != 56) update(&zero, 1);
        uint8_t len[8];
        for (int i = 0; i < 8; ++i) len[i] = (uint8_t)(bits >> (8 * i));
        update(len, 8);
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out[4 * i + j] = (uint8_t)(h[i] >> (8 * j));
    }
};

struct StreamInfo { int sample_rate = 0, channels = 0, bps = 0, max_block = 0; int64_t total = 0; uint8_t md5[16]; size_t first_frame = 0; };

bool parse_header(const uint8_t* p, size_t n, StreamInfo* si) {
    size_t pos = 0;
    if (n >= 10 && !memcmp(p, "ID3", 3)) {      // an ID3v2 tag in front of the stream
        const size_t sz = ((size_t)(p[6] & 0x7f) << 21) | ((size_t)(p[7] & 0x7f) << 14) | ((size_t)(p[8] & 0x7f) << 7) | (size_t)(p[9] & 0x7f);
        pos = 10 + sz;
    }
    if (pos + 4 > n || memcmp(p + pos, "fLaC", 4)) return false;
    pos += 4;
    bool got = false;
    for (;;) {
        if (pos + 4 > n) return false;
        const bool last = p[pos] & 0x80;
        const int type = p[pos] & 0x7f;
        const size_t len = ((size_t)p[pos + 1] << 16) | ((size_t)p[pos + 2] << 8) | p[pos + 3];
        pos += 4;
        if (pos + len > n) return false;
        if (type == 0) {
            if (len < 34) return false;
            const uint8_t* s = p + pos;
            si->max_block = (s[2] << 8) | s[3];
            si->sample_rate = (s[10] << 12) | (s[11] << 4) | (s[12] >> 4);
            si->channels = ((s[12] >> 1) & 7) + 1;
            si->bps = (((s[12] & 1) << 4) | (s[13] >> 4)) + 1;
            si->total = ((int64_t)(s[13] & 0xf) << 32) | ((int64_t)s[14] << 24) | ((int64_t)s[15] << 16) | ((int64_t)s[16] << 8) | s[17];
            memcpy(si->md5, s + 18, 16);
            got = true;
        }
        pos += len;
        if (last) break;
    }
    si->first_frame = pos;
    return got;
}

bool decode_residual(Bits& br, int32_t* out, int blocksize, int pred_order) {
    const int method = (int)br.read(2);
    if (method > 1) return false;
    const int pbits = method ? 5 : 4, esc = method ? 31 : 15;
    const int porder = (int)br.read(4);
    const int nparts = 1 << porder;
    if ((blocksize >> porder) << porder != blocksize && porder > 0) return false;
    int i = pred_order;
    for (int part = 0; part < nparts; ++part) {
        int count = (blocksize >> porder) - (part == 0 ? pred_order : 0);
        if (count < 0) return false;
        const int param = (int)br.read(pbits);
        if (param == esc) {
            const int nb = (int)br.read(5);
            for (int k = 0; k < count; ++k) out[i++] = br.read_signed(nb);
        } else {
            for (int k = 0; k < count; ++k) out[i++] = br.read_rice(param);
        }
        if (br.bad) return false;
    }
    return i == blocksize;
}

// s[i] += (sum_j coef[j] * s[i - 1 - j]) >> shift: the order as a compile-time constant for the orders encoders use (the loop is one
// dependent chain per sample; unrolled, the products of the older samples run ahead of it)
template <int ORDER>
void lpc_restore_n(int32_t* s, int blocksize, const int32_t* coef, int shift) {
    int64_t c[ORDER];
    for (int j = 0; j < ORDER; ++j) c[j] = coef[j];
    for (int i = ORDER; i < blocksize; ++i) {
        int64_t p = 0;
#pragma unroll
        for (int j = 0; j < ORDER; ++j) p += c[j] * s[i - 1 - j];
        s[i] = (int32_t)(s[i] + (p >> shift));
    }
}
void lpc_restore(int32_t* s, int blocksize, int order, const int32_t* coef, int shift) {
    switch (order) {
        case 1: return lpc_restore_n<1>(s, blocksize, coef, shift);
        case 2: return lpc_restore_n<2>(s, blocksize, coef, shift);
        case 3: return lpc_restore_n<3>(s, blocksize, coef, shift);
        case 4: return lpc_restore_n<4>(s, blocksize, coef, shift);
        case 5: return lpc_restore_n<5>(s, blocksize, coef, shift);
        case 6: return lpc_restore_n<6>(s, blocksize, coef, shift);
        case 7: return lpc_restore_n<7>(s, blocksize, coef, shift);
        case 8: return lpc_restore_n<8>(s, blocksize, coef, shift);
        case 9: return lpc_restore_n<9>(s, blocksize, coef, shift);
        case 10: return lpc_restore_n<10>(s, blocksize, coef, shift);
        case 11: return lpc_restore_n<11>(s, blocksize, coef, shift);
        case 12: return lpc_restore_n<12>(s, blocksize, coef, shift);
        default: break;
    }
    for (int i = order; i < blocksize; ++i) {
        int64_t p = 0;
        for (int j = 0; j < order; ++j) p += (int64_t)coef[j] * s[i - 1 - j];
        s[i] = (int32_t)(s[i] + (p >> shift));
    }
}

bool decode_subframe(Bits& br, int32_t* s, int blocksize, int bps) {
    if (br.read(1)) return false;
    const int type = (int)br.read(6);
    int wasted = 0;
    if (br.read(1)) { wasted = (int)br.read_unary() + 1; bps -= wasted; if (bps <= 0) return false; }
    if (type == 0) {
        const int32_t v = br.read_signed(bps);
        for (int i = 0; i < blocksize; ++i) s[i] = v;
    } else if (type == 1) {
        for (int i = 0; i < blocksize; ++i) s[i] = br.read_signed(bps);
    } else if (type >= 8 && type <= 12) {
        const int order = type - 8;
        if (order > blocksize) return false;
        for (int i = 0; i < order; ++i) s[i] = br.read_signed(bps);
        if (!decode_residual(br, s, blocksize, order)) return false;
        // the sums can exceed 32 bits for 32-bit streams only in malformed input; int64 keeps it defined
        switch (order) {      // (the switch outside the loops)
            case 1: for (int i = 1; i < blocksize; ++i) s[i] = (int32_t)(s[i] + (int64_t)s[i - 1]); break;
            case 2: for (int i = 2; i < blocksize; ++i) s[i] = (int32_t)(s[i] + 2 * (int64_t)s[i - 1] - s[i - 2]); break;
            case 3: for (int i = 3; i < blocksize; ++i) s[i] = (int32_t)(s[i] + 3 * (int64_t)s[i - 1] - 3 * (int64_t)s[i - 2] + s[i - 3]); break;
            case 4: for (int i = 4; i < blocksize; ++i) s[i] = (int32_t)(s[i] + 4 * (int64_t)s[i - 1] - 6 * (int64_t)s[i - 2] + 4 * (int64_t)s[i - 3] - s[i - 4]); break;
            default: break;
        }
    } else if (type >= 32) {
        const int order = (type & 31) + 1;
        if (order > blocksize) return false;
        for (int i = 0; i < order; ++i) s[i] = br.read_signed(bps);
        const int prec = (int)br.read(4) + 1;
        if (prec == 16) return false;
        const int shift = br.read_signed(5);
        if (shift < 0) return false;
        int32_t coef[32];
        for (int j = 0; j < order; ++j) coef[j] = br.read_signed(prec);
        if (!decode_residual(br, s, blocksize, order)) return false;
        lpc_restore(s, blocksize, order, coef, shift);
    } else {
        return false;      // reserved subframe type
    }
    if (wasted) for (int i = 0; i < blocksize; ++i) s[i] = (int32_t)((uint32_t)s[i] << wasted);
    return !br.bad;
}

// one frame starting at byte `pos`; appends blocksize x channels interleaved samples to out[written...]
constexpr int FLAC_ENOSPC = -100;      // internal: the frame is sound but does not fit the caller's buffer
int decode_frame(const uint8_t* p, size_t n, size_t* pos, const StreamInfo& si, std::vector<int32_t>& ch, int32_t* out, int64_t capacity,
                 int64_t* written, int* frame_bps) {
    Bits br(p + *pos, n - *pos);
    if (br.read(14) != 0x3FFE) return SCL_EINVAL;
    if (br.read(1)) return SCL_EINVAL;
    br.read(1);                                   // blocking strategy: the sample / frame number is not needed to decode
    const int bs_code = (int)br.read(4), sr_code = (int)br.read(4), ch_code = (int)br.read(4), ss_code = (int)br.read(3);
    if (br.read(1)) return SCL_EINVAL;
    {   // UTF-8-like coded number
        const uint32_t b0 = br.read(8);
        int extra = 0;
        if (b0 & 0x80) { uint32_t m = 0x40; while (b0 & m) { ++extra; m >>= 1; } if (extra == 0 || extra > 6) return SCL_EINVAL; }
        for (int i = 0; i < extra; ++i) if ((br.read(8) & 0xC0) != 0x80) return SCL_EINVAL;
    }
    int blocksize;
    if (bs_code == 0) return SCL_EINVAL;
    else if (bs_code == 1) blocksize = 192;
    else if (bs_code <= 5) blocksize = 576 << (bs_code - 2);
    else if (bs_code == 6) blocksize = (int)br.read(8) + 1;
    else if (bs_code == 7) blocksize = (int)br.read(16) + 1;
    else blocksize = 256 << (bs_code - 8);
    if (sr_code == 12) br.read(8); else if (sr_code == 13 || sr_code == 14) br.read(16); else if (sr_code == 15) return SCL_EINVAL;
    static const int ss_tab[8] = {0, 8, 12, -1, 16, 20, 24, 32};
    int bps = ss_tab[ss_code];
    if (bps < 0) return SCL_EINVAL;
    if (bps == 0) bps = si.bps;
    *frame_bps = bps;
    br.align();
    const size_t hdr_len = br.byte_pos();
    const uint32_t c8 = br.read(8);
    if (br.bad || *pos + hdr_len + 1 > n || crc8(p + *pos, hdr_len) != c8) return SCL_EINVAL;      // header (+ its CRC byte) inside the buffer
    int nch, mode = 0;                            // mode 1 left/side, 2 side/right, 3 mid/side
    if (ch_code < 8) nch = ch_code + 1; else if (ch_code <= 10) { nch = 2; mode = ch_code - 7; } else return SCL_EINVAL;
    if (nch != si.channels) return SCL_EINVAL;
    if ((int)ch.size() < nch * blocksize) ch.resize((size_t)nch * blocksize);
    for (int c = 0; c < nch; ++c) {
        const int extra = (mode == 1 && c == 1) || (mode == 2 && c == 0) || (mode == 3 && c == 1);
        if (bps + extra > 32) return SCL_EINVAL;
        if (!decode_subframe(br, ch.data() + (size_t)c * blocksize, blocksize, bps + extra)) return SCL_EINVAL;
    }
    br.align();
    const size_t body_len = br.byte_pos();
    const uint32_t c16 = br.read(16);
    if (br.bad || *pos + body_len + 2 > n || crc16(p + *pos, body_len) != c16) return SCL_EINVAL;
    int32_t* a = ch.data(); int32_t* b = ch.data() + blocksize;
    if (mode == 1) for (int i = 0; i < blocksize; ++i) b[i] = a[i] - b[i];
    else if (mode == 2) for (int i = 0; i < blocksize; ++i) a[i] = a[i] + b[i];
    else if (mode == 3) for (int i = 0; i < blocksize; ++i) {
        const int32_t side = b[i];
        const int32_t mid = (int32_t)(((uint32_t)a[i] << 1) | (uint32_t)(side & 1));
        a[i] = (mid + side) >> 1; b[i] = (mid - side) >> 1;
    }
    if (*written + blocksize > capacity) return FLAC_ENOSPC;
    for (int c = 0; c < nch; ++c) {
        const int32_t* src = ch.data() + (size_t)c * blocksize;
        int32_t* dst = out + *written * nch + c;
        for (int i = 0; i < blocksize; ++i) dst[(size_t)i * nch] = src[i];
    }
    *written += blocksize;
    *pos += body_len + 2;
    return SCL_OK;
}

}  // namespace

extern "C" int scl_flac_info(const void* data, int64_t nbytes, int* sample_rate, int* channels, int* bits_per_sample, int64_t* total_samples) {
    SCL_REQUIRE(data && nbytes > 42 && sample_rate && channels && bits_per_sample && total_samples, "flac_info: bad args");
    StreamInfo si;
    SCL_REQUIRE(parse_header((const uint8_t*)data, (size_t)nbytes, &si), "flac_info: not a FLAC stream (no fLaC marker / STREAMINFO)");
    *sample_rate = si.sample_rate; *channels = si.channels; *bits_per_sample = si.bps; *total_samples = si.total;
    return SCL_OK;
}

// Shared body of the two decode entry points: frames are decoded one at a time; `sink_i` (interleaved int32 [capacity][channels]) or
// `sink_f` (mono float [capacity]: channel mean of sample / 2^(bits-1), the arithmetic of librosa.load(mono=True) on soundfile's float
// samples — what scl_amd/pack.py did with two numpy passes over the int32 image) receives them, and the MD5 runs over the frames as they come.
static int flac_decode_any(const void* data, int64_t nbytes, int32_t* sink_i, float* sink_f, int64_t capacity_samples, int64_t* decoded_samples,
                           int check_md5) {
    const uint8_t* p = (const uint8_t*)data;
    const size_t n = (size_t)nbytes;
    StreamInfo si;
    SCL_REQUIRE(parse_header(p, n, &si), "flac_decode: not a FLAC stream (no fLaC marker / STREAMINFO)");
    SCL_REQUIRE(si.channels >= 1 && si.channels <= 8 && si.bps >= 4 && si.bps <= 32, "flac_decode: %d channels, %d bits", si.channels, si.bps);
    std::vector<int32_t> ch, frame;
    size_t pos = si.first_frame;
    int64_t written = 0;
    int bps = si.bps;
    bool all0 = true;
    for (int i = 0; i < 16; ++i) all0 = all0 && si.md5[i] == 0;
    const bool md5_on = check_md5 && !all0;      // an all-zero signature means the encoder did not compute one
    Md5 md;
    const int bytes = (si.bps + 7) / 8, nch = si.channels;
    std::vector<uint8_t> tmp;
    const float inv = 1.0f / (float)(1ull << (si.bps - 1));
    while (pos + 2 <= n && (si.total == 0 || written < si.total)) {
        if (!(p[pos] == 0xFF && (p[pos + 1] & 0xFE) == 0xF8)) break;      // trailing bytes that are not a frame (e.g. an ID3v1 tag)
        // one frame into `frame` (interleaved), then into the sink
        int64_t fw = 0;
        const size_t frame_cap = 65536;      // the largest block a frame header can declare
        if (frame.size() < frame_cap * (size_t)nch) frame.resize(frame_cap * (size_t)nch);
        const int rc = decode_frame(p, n, &pos, si, ch, frame.data(), (int64_t)frame_cap, &fw, &bps);
        if (rc != SCL_OK) {
            scl_set_error("flac_decode: corrupt frame at byte %zu (sample %lld): sync / CRC / reserved-field check failed", pos, (long long)written);
            return rc;
        }
        if (written + fw > capacity_samples) {      // told apart from a corrupt frame: a caller that could not size the output (STREAMINFO total = 0) grows it and retries
            scl_set_error("flac_decode: output too small (%lld samples per channel hold the stream only up to byte %zu)", (long long)capacity_samples, pos);
            return SCL_EINVAL;
        }
        const int32_t* f = frame.data();
        if (sink_i) memcpy(sink_i + written * nch, f, (size_t)fw * nch * sizeof(int32_t));
        if (sink_f) {
            float* o = sink_f + written;
            if (nch == 1) for (int64_t i = 0; i < fw; ++i) o[i] = (float)f[i] * inv;      // exact: a power-of-two scale
            else for (int64_t i = 0; i < fw; ++i) {
                float a = 0.f;
                for (int c = 0; c < nch; ++c) a += (float)f[i * nch + c] * inv;         // numpy's mean over the channel axis: sequential float32 sum, then / channels
                o[i] = a / (float)nch;
            }
        }
        if (md5_on) {
            const int64_t m = fw * nch;
            if (tmp.size() < (size_t)m * bytes) tmp.resize((size_t)m * bytes);
            if (bytes == 2) {
                int16_t* t16 = reinterpret_cast<int16_t*>(tmp.data());
                for (int64_t i = 0; i < m; ++i) t16[i] = (int16_t)f[i];
            } else {
                for (int64_t i = 0; i < m; ++i) {
                    const uint32_t v = (uint32_t)f[i];
                    for (int b = 0; b < bytes; ++b) tmp[(size_t)i * bytes + b] = (uint8_t)(v >> (8 * b));
                }
            }
            md.update(tmp.data(), (size_t)m * bytes);
        }
        written += fw;
    }
    SCL_REQUIRE(si.total == 0 || written == si.total, "flac_decode: stream ends after %lld of %lld samples", (long long)written, (long long)si.total);
    *decoded_samples = written;
    if (md5_on) {
        uint8_t dig[16];
        md.final(dig);
        SCL_REQUIRE(!memcmp(dig, si.md5, 16), "flac_decode: MD5 of the decoded audio differs from the signature in STREAMINFO");
    }
    return SCL_OK;
}

extern "C" int scl_flac_decode_i32(const void* data, int64_t nbytes, int32_t* out, int64_t capacity_samples, int64_t* decoded_samples,
                                   int check_md5) {
    SCL_REQUIRE(data && nbytes > 42 && out && decoded_samples && capacity_samples > 0, "flac_decode: bad args");
    return flac_decode_any(data, nbytes, out, nullptr, capacity_samples, decoded_samples, check_md5);
}

extern "C" int scl_flac_decode_mono_f32(const void* data, int64_t nbytes, float* out, int64_t capacity_samples, int64_t* decoded_samples,
                                        int check_md5) {
    SCL_REQUIRE(data && nbytes > 42 && out && decoded_samples && capacity_samples > 0, "flac_decode: bad args");
    return flac_decode_any(data, nbytes, nullptr, out, capacity_samples, decoded_samples, check_md5);
}
