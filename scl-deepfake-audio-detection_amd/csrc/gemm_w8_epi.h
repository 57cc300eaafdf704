// gemm_w8_epi.h — the LDS-staged epilogue shared by the wide-tile GEMM kernels (gemm_w8.hip: one 8-wave block per CU; gemm_x2.hip: two
// 4-wave blocks per CU).  A wave hands over up to 64 rows x 64 columns of f32 accumulators per pass.
#pragma once
#include "gemm_common.h"

namespace sclg {

// ---- epilogue through LDS: whole 128-B lines per store ----------------------------------------------------------------
// The MFMA leaves a lane with 4 consecutive columns of 16 different rows, so a direct store touches 16 cache lines with 32
// (bf16: 16) bytes each; with one 8-wave block per CU nothing overlaps the epilogue and those stores (plus the residual loads of
// the same shape) took 16-27 us of a 45-55 us block (stamps, profiles/r2_gemm_stamps.txt).  Here each wave parks up to 64 rows x
// 64 columns of f32 accumulators in a private 16-KiB LDS block ([row][256 B], 16-byte chunks XOR-swizzled with row & 15: the
// ds_write_b128 of a 16-lane group and the two ds_read_b128 per lane are bank-conflict free) and reads them back row-contiguous:
// a lane owns 8 consecutive columns of one row, 8 lanes own a row's 64 columns, so bias / residual loads and the C / C2 stores of
// one wave-instruction cover 8 rows x 128 (bf16) or 256 (f32) contiguous bytes.  Arithmetic per element is the old epilogue's,
// in the same order: results are bit-identical.
// NMT: 16-row blocks the accumulator array holds (4: 16-KiB block per wave; the persistent kernel hands over 2 at a time: 8 KiB).
// cs_carry (optional, 8 floats of the caller): the column sums START from it and are handed back in it instead of being reduced and
// stored per pass (the caller finishes with w8_colsum_store): two 2-block passes then add a lane's rows in the order one 4-block pass does.
// blk_stride: byte distance between the wave's 4-KiB (16-row) transposition blocks (4096: one contiguous block).
// EK: the epilogue KIND.  The flag word is a launch constant, yet the rolled row loop below re-tested ~20 of its bits for every 8 rows
// (the ISA showed the loop body as a chain of ~40 small blocks ending in s_cbranch).  The encoder's five hot epilogues get the flags as
// compile-time constants (their loop is straight-line code); everything else runs the generic form (EK 0).  Same arithmetic either way.
//   1: bf16 C [+ bias]                      — QKV forward, the plain data gradients
//   2: bf16 C = gelu(acc + bias), bf16 C2 = gelu'  — fc1 forward (ACT 5)
//   3: bf16 C = acc x R, R bf16 the stored derivative (RMODE 2 / RACT 4)  — fc2 data gradient
//   4: f32 C = acc [+ bias] + R, R f32      — out-proj / fc2 forward (residual stream)
//   5: f32 C [+ bias]                       — split-K slabs of the weight gradients, conv layers
// Kinds 1-5 also assume 8-element alignment of every row start (ldc, c_rbstride, cbase multiples of 8: checked by the dispatcher).
template <int NMT, int EK>
static __device__ __forceinline__ void w8_epilogue_pass_k(const GemmK& d, f32x4 (&acc)[NMT][4], int nmt, char* wlds, char* wextra, int mbase, int nbase,
                                                   int mlimit, long long cbase, const float* bias, int lane, float* csum_row,
                                                   float* cs_carry, const int blk_stride) {
    // The generic form evaluates every step of the chain (alpha, bias, activation, x R, + R, column sum) in its own basic block, so no
    // multiply-add pair of DIFFERENT steps is ever fused; the straight-line kinds must not fuse them either (seen: the column sum
    // cs += (acc x R) became one fma and the fc1 bias gradient changed in the last bits).
#pragma clang fp contract(off)
    const int flags = d.flags;
    constexpr bool KN = EK != 0;
    const bool c_f32 = KN ? (EK == 4 || EK == 5) : bool(flags & SCL_GEMM_C_F32);
    const bool c2_f32 = KN ? false : bool(flags & SCL_GEMM_C2_F32);
    const bool r_f32 = KN ? (EK == 4) : bool(flags & SCL_GEMM_R_F32);
    const bool has_bias = EK == 2 ? true : (EK == 3 ? false : bool(flags & SCL_GEMM_HAS_BIAS));
    const bool has_c2 = KN ? (EK == 2) : bool(flags & SCL_GEMM_HAS_C2);
    const bool drop = KN ? false : bool(flags & SCL_GEMM_DROPOUT);
    const int act = KN ? (EK == 2 ? 5 : 0) : ((flags >> SCL_GEMM_ACT_SHIFT) & 0xF);
    const int rmode = KN ? (EK == 3 ? 2 : (EK == 4 ? 1 : 0)) : ((flags >> SCL_GEMM_RMODE_SHIFT) & 0xF);
    const int ract = KN ? (EK == 3 ? 4 : 0) : ((flags >> SCL_GEMM_RACT_SHIFT) & 0xF);
    const int g = lane >> 4, lc = lane & 15;
    const int c = lane & 7, rsub = lane >> 3;
    const int col = nbase + 8 * c;
    const bool colv = d.vec_ok && col + 8 <= d.N;
    // The R operand (activation-gradient input / residual) comes in through LDS-DMA, several rows at a time: loads and stores share
    // the in-order vmcnt counter, so a wait for row i's R vector requested after row i-1's stores also waits for those stores'
    // acknowledgement — one memory round trip per row (tools/epilogue_probe.py: reading R cost 52 us per launch where a second
    // store costs 14).  Staged this way the queue is drained once per batch, and the loop stays rolled (the kernels are 44 KB of
    // code: unrolling the epilogue to keep R in registers slowed every variant, R or not, by 20-36 us).  bf16 R: batches of four
    // rows in the wave's 4 KiB above the transposition blocks.  f32 R (twice the bytes): rows 0-1 and 2-3 there, rows 4-7 in the
    // first half of the wave's own transposition block, whose rows have been consumed by then.
    // The FIRST batch is requested before the accumulators are parked in LDS: its round trip runs under those ds_writes.
    const bool r_dma = rmode && d.vec_ok && !(d.debug & 2) && !(d.ldc & 7) && !(d.c_rbstride & 7) && !(cbase & 7) && !((unsigned long long)d.R & 15);
    __amdgpu_buffer_rsrc_t r_rsrc = make_rsrc(reinterpret_cast<const char*>(r_dma ? d.R : d.C));
    auto r_issue = [&](int i) {
        const int nrow = r_f32 ? (i == 4 ? 4 : 2) : 4;
        char* dst = (r_f32 && i == 4) ? wlds : wextra;
        for (int u = 0; u < nrow; ++u) {
            const int row2 = mbase + 8 * (i + u) + rsub;
            unsigned offb = OOB;
            if (i + u < 2 * nmt && row2 < mlimit && colv) {
                const unsigned q2 = udiv_magic((unsigned)row2, d.c_magic, d.c_shift);
                const long long o2 = cbase + (long long)q2 * d.c_rbstride + (long long)((unsigned)row2 - q2 * d.c_rpb) * d.ldc + col;
                offb = (unsigned)(o2 << (r_f32 ? 2 : 1));
            }
            if (r_f32) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r_rsrc, (lds_void*)(dst + u * 2048), 16, offb, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r_rsrc, (lds_void*)(dst + u * 2048 + 1024), 16, offb == OOB ? OOB : offb + 16, 0, 0, 0);
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r_rsrc, (lds_void*)(dst + u * 1024), 16, offb, 0, 0, 0);
            }
        }
        return dst;
    };
    if (r_dma) r_issue(0);
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) {
        if (mt < nmt) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                *reinterpret_cast<f32x4*>(wlds + mt * blk_stride + lc * 256 + (((nt * 4 + g) ^ lc) << 4)) = acc[mt][nt];
        }
    }
    float bb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (has_bias && colv) {
        const float4 b0 = *reinterpret_cast<const float4*>(bias + col), b1 = *reinterpret_cast<const float4*>(bias + col + 4);
        bb[0] = b0.x; bb[1] = b0.y; bb[2] = b0.z; bb[3] = b0.w; bb[4] = b1.x; bb[5] = b1.y; bb[6] = b1.z; bb[7] = b1.w;
    }
    const EpiArgs ea = {d.C, d.C2, d.R, d.N, d.flags, d.drop_seed, d.drop_p};
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // column sums of this lane's 8 columns over the rows it stores (csum_row)
    if (cs_carry) {
#pragma unroll
        for (int j = 0; j < 8; ++j) cs[j] = cs_carry[j];
    }
    const char* rstage = wextra;
    for (int i = 0; i < 2 * nmt; ++i) {
        const bool issue = r_dma && (r_f32 ? (i == 0 || i == 2 || i == 4) : (i & 3) == 0);
        if (issue) {
            if (r_f32 && i == 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's reads of rows 0-31 have returned
            char* dst = i == 0 ? wextra : r_issue(i);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            rstage = dst - (r_f32 ? i * 2048 : i * 1024);      // row-batch i lives at rstage + i * (2048 | 1024)
        }
        const int r = 8 * i + rsub;
        const char* wrow = wlds + (r >> 4) * blk_stride + (r & 15) * 256;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(wrow + (((2 * c) ^ (r & 15)) << 4));
        const f32x4 hi = *reinterpret_cast<const f32x4*>(wrow + (((2 * c + 1) ^ (r & 15)) << 4));
        const int row = mbase + r;
        if (row >= mlimit) continue;
        const unsigned q = udiv_magic((unsigned)row, d.c_magic, d.c_shift);
        const long long off = cbase + (long long)q * d.c_rbstride + (long long)((unsigned)row - q * d.c_rpb) * d.ldc + col;
        float v[8] = {d.alpha * lo[0], d.alpha * lo[1], d.alpha * lo[2], d.alpha * lo[3], d.alpha * hi[0], d.alpha * hi[1], d.alpha * hi[2], d.alpha * hi[3]};
        if (colv) {
            if (has_bias) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += bb[j];
            }
            float w[8] = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};      // second output: the pre-activation, or (ACT 5) gelu' of it
            if (act == 5) {
#pragma unroll
                for (int j = 0; j < 8; j += 2) gelu_both2(v[j], v[j + 1], w[j], w[j + 1]);
            }
            if (has_c2) {
                if (c2_f32) {
                    float* p = reinterpret_cast<float*>(d.C2) + off;
                    *reinterpret_cast<float4*>(p) = make_float4(w[0], w[1], w[2], w[3]);
                    *reinterpret_cast<float4*>(p + 4) = make_float4(w[4], w[5], w[6], w[7]);
                } else {
                    bf16_t* p = reinterpret_cast<bf16_t*>(d.C2) + off;
                    if (KN || (off & 7) == 0) *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf2(w[0], w[1]), pack_bf2(w[2], w[3]), pack_bf2(w[4], w[5]), pack_bf2(w[6], w[7]));
                    else { *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf2(w[0], w[1]), pack_bf2(w[2], w[3])); *reinterpret_cast<uint2*>(p + 4) = make_uint2(pack_bf2(w[4], w[5]), pack_bf2(w[6], w[7])); }
                }
            }
            if (act == 5) {
            } else if (act == 1) {
#pragma unroll
                for (int j = 0; j < 8; j += 2) gelu2(v[j], v[j + 1]);
            } else if (act) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = act_f(act, v[j]);
            }
            float rr[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (rmode) {
                if (r_f32) {
                    const float* p = r_dma ? reinterpret_cast<const float*>(rstage + i * 2048 + lane * 16) : reinterpret_cast<const float*>(d.R) + off;
                    const float4 t0 = *reinterpret_cast<const float4*>(p), t1 = *reinterpret_cast<const float4*>(p + (r_dma ? 256 : 4));
                    rr[0] = t0.x; rr[1] = t0.y; rr[2] = t0.z; rr[3] = t0.w; rr[4] = t1.x; rr[5] = t1.y; rr[6] = t1.z; rr[7] = t1.w;
                } else {
                    const bf16_t* p = reinterpret_cast<const bf16_t*>(d.R) + off;
                    uint2 t0, t1;
                    if (r_dma) { const uint4 t = *reinterpret_cast<const uint4*>(rstage + i * 1024 + lane * 16); t0 = make_uint2(t.x, t.y); t1 = make_uint2(t.z, t.w); }
                    else if (KN || (off & 7) == 0) { const uint4 t = *reinterpret_cast<const uint4*>(p); t0 = make_uint2(t.x, t.y); t1 = make_uint2(t.z, t.w); }
                    else { t0 = *reinterpret_cast<const uint2*>(p); t1 = *reinterpret_cast<const uint2*>(p + 4); }
                    rr[0] = __uint_as_float(t0.x << 16); rr[1] = __uint_as_float(t0.x & 0xFFFF0000u);
                    rr[2] = __uint_as_float(t0.y << 16); rr[3] = __uint_as_float(t0.y & 0xFFFF0000u);
                    rr[4] = __uint_as_float(t1.x << 16); rr[5] = __uint_as_float(t1.x & 0xFFFF0000u);
                    rr[6] = __uint_as_float(t1.y << 16); rr[7] = __uint_as_float(t1.y & 0xFFFF0000u);
                }
            }
            if (rmode == 2 && ract == 1) {
#pragma unroll
                for (int j = 0; j < 8; j += 2) { gelu_grad2(rr[j], rr[j + 1]); v[j] *= rr[j]; v[j + 1] *= rr[j + 1]; }
            } else if (rmode == 2 && ract == 4) {      // R holds the derivative itself (ACT 5 forward): no per-element switch
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] *= rr[j];
            } else if (rmode == 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] *= act_grad_f(ract, rr[j]);
            }
            if (drop) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] *= dropout_scale(d.drop_seed, (uint64_t)(off + j), d.drop_p);
            }
            if (rmode == 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += rr[j];
            }
            if (csum_row || cs_carry) {
#pragma unroll
                for (int j = 0; j < 8; ++j) cs[j] += v[j];
            }
            if (c_f32) {
                float* p = reinterpret_cast<float*>(d.C) + off;
                *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
            } else if (!KN && ((unsigned)flags & SCL_GEMM_C_SPLIT3)) {
                // triple-plane row image [hi | hi | lo] (planes N apart, ldc = 3 N; the host checked alignment and N % 8 == 0): the scoring
                // path's fc1 hands fc2 its left operand without an f32 copy and a split pass (scl_split3_f32_bf16's arithmetic)
                bf16_t* p = reinterpret_cast<bf16_t*>(d.C) + off;
                const uint4 h = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
                const unsigned hw[4] = {h.x, h.y, h.z, h.w};
                float l8[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    l8[2 * j] = v[2 * j] - __uint_as_float(hw[j] << 16);
                    l8[2 * j + 1] = v[2 * j + 1] - __uint_as_float(hw[j] & 0xFFFF0000u);
                }
                *reinterpret_cast<uint4*>(p) = h;
                *reinterpret_cast<uint4*>(p + d.N) = h;
                *reinterpret_cast<uint4*>(p + 2 * (long long)d.N) = make_uint4(pack_bf2(l8[0], l8[1]), pack_bf2(l8[2], l8[3]), pack_bf2(l8[4], l8[5]), pack_bf2(l8[6], l8[7]));
            } else {
                bf16_t* p = reinterpret_cast<bf16_t*>(d.C) + off;
                if (KN || (off & 7) == 0) *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
                else { *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])); *reinterpret_cast<uint2*>(p + 4) = make_uint2(pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])); }
            }
        } else {
            // column-edge tile / unaligned C: element-wise path
#pragma unroll
            for (int j = 0; j < 8; ++j) epi_scalar(ea, v[j], off + j, col + j, bias);
        }
    }
    if (cs_carry) {
#pragma unroll
        for (int j = 0; j < 8; ++j) cs_carry[j] = cs[j];
    } else if (csum_row) {      // lanes that share c = lane & 7 hold the same 8 columns for rows rsub, rsub + 8, ...: fixed-order butterfly over rsub
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            cs[j] += __shfl_xor(cs[j], 8, 64);
            cs[j] += __shfl_xor(cs[j], 16, 64);
            cs[j] += __shfl_xor(cs[j], 32, 64);
        }
        if (lane < 8 && colv) {
            *reinterpret_cast<float4*>(csum_row + col) = make_float4(cs[0], cs[1], cs[2], cs[3]);
            *reinterpret_cast<float4*>(csum_row + col + 4) = make_float4(cs[4], cs[5], cs[6], cs[7]);
        }
    }
}


// SPEC: bit k set = kind k may be taken.  Kind 5 is compiled out by default: measured on one box it is 5-6 % SLOWER than the generic loop
// (f32 slabs of the weight gradients 127 -> 135 us, conv forward 395 -> 416), unlike kinds 1-4 (plain bf16 41 -> 36, f32 residual 47 -> 45,
// fc1 / fc2 above).  The 256-row single-barrier kernels sit at the 256-register limit (any extra copy pushed them into scratch, and a
// kernel with a private segment pays a queue-side set-up per dispatch) and the ping-pong kernels only ever store f32 slabs: both pass 0.
template <int NMT, int SPEC = 0x1E>
static __device__ __forceinline__ void w8_epilogue_pass(const GemmK& d, f32x4 (&acc)[NMT][4], int nmt, char* wlds, char* wextra, int mbase, int nbase,
                                                 int mlimit, long long cbase, const float* bias, int lane, float* csum_row = nullptr,
                                                 float* cs_carry = nullptr, const int blk_stride = 4096) {
    constexpr int KEY = SCL_GEMM_C_F32 | SCL_GEMM_C2_F32 | SCL_GEMM_R_F32 | SCL_GEMM_HAS_BIAS | SCL_GEMM_HAS_C2 | SCL_GEMM_DROPOUT |
                        (0xF << SCL_GEMM_ACT_SHIFT) | (0xF << SCL_GEMM_RMODE_SHIFT) | (0xF << SCL_GEMM_RACT_SHIFT);
    const int f = d.flags & KEY, fb = f & ~SCL_GEMM_HAS_BIAS;
    if (SPEC != 0 && !(d.debug & 16) && !((unsigned)d.flags & SCL_GEMM_C_SPLIT3) && d.vec_ok && !(d.ldc & 7) && !(d.c_rbstride & 7) && !(cbase & 7)) {      // debug bit 4: SCL_W8_EPI_GENERIC=1 (A/B); the triple-plane store lives in the generic loop only
        if ((SPEC & 2) && fb == 0) return w8_epilogue_pass_k<NMT, 1>(d, acc, nmt, wlds, wextra, mbase, nbase, mlimit, cbase, bias, lane, csum_row, cs_carry, blk_stride);
        if ((SPEC & 4) && f == (SCL_GEMM_HAS_BIAS | SCL_GEMM_HAS_C2 | (5 << SCL_GEMM_ACT_SHIFT)))
            return w8_epilogue_pass_k<NMT, 2>(d, acc, nmt, wlds, wextra, mbase, nbase, mlimit, cbase, bias, lane, csum_row, cs_carry, blk_stride);
        if ((SPEC & 8) && f == ((2 << SCL_GEMM_RMODE_SHIFT) | (4 << SCL_GEMM_RACT_SHIFT)))
            return w8_epilogue_pass_k<NMT, 3>(d, acc, nmt, wlds, wextra, mbase, nbase, mlimit, cbase, bias, lane, csum_row, cs_carry, blk_stride);
        if ((SPEC & 16) && fb == (SCL_GEMM_C_F32 | SCL_GEMM_R_F32 | (1 << SCL_GEMM_RMODE_SHIFT)))
            return w8_epilogue_pass_k<NMT, 4>(d, acc, nmt, wlds, wextra, mbase, nbase, mlimit, cbase, bias, lane, csum_row, cs_carry, blk_stride);
        if ((SPEC & 32) && fb == SCL_GEMM_C_F32) return w8_epilogue_pass_k<NMT, 5>(d, acc, nmt, wlds, wextra, mbase, nbase, mlimit, cbase, bias, lane, csum_row, cs_carry, blk_stride);
    }
    w8_epilogue_pass_k<NMT, 0>(d, acc, nmt, wlds, wextra, mbase, nbase, mlimit, cbase, bias, lane, csum_row, cs_carry, blk_stride);
}

// finish a carried column sum: butterfly over the 8 row lanes that share a column group, lanes 0-7 store 8 columns each
static __device__ __forceinline__ void w8_colsum_store(const GemmK& d, float (&cs)[8], float* csum_row, int nbase, int lane) {
    const int col = nbase + 8 * (lane & 7);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        cs[j] += __shfl_xor(cs[j], 8, 64);
        cs[j] += __shfl_xor(cs[j], 16, 64);
        cs[j] += __shfl_xor(cs[j], 32, 64);
    }
    if (lane < 8 && d.vec_ok && col + 8 <= d.N) {
        *reinterpret_cast<float4*>(csum_row + col) = make_float4(cs[0], cs[1], cs[2], cs[3]);
        *reinterpret_cast<float4*>(csum_row + col + 4) = make_float4(cs[4], cs[5], cs[6], cs[7]);
    }
}

}  // namespace sclg
