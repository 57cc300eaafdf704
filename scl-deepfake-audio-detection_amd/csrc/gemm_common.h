// gemm_common.h — device-side pieces shared by the bf16 GEMM kernel families (gemm.hip, gemm_w8.hip):
// operand descriptors, buffer-descriptor construction, the XCD-aware tile order, LDS fragment reads and the fused epilogue.
#pragma once
#include "common.h"

namespace sclg {

constexpr int BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;  // 16 KiB: one 128-row (or 128-column) operand sub-tile of one K step
constexpr unsigned OOB = 0xFFFFFFFFu;

struct OpK {             // device-side operand description (bytes, 32-bit)
    const void* ptr;
    long long bs1, bs2;  // batch strides in BYTES
    unsigned rb_bytes, ld_bytes, cout_bytes;
    unsigned rpb, rpb_magic, rpb_shift;
    unsigned cin, cin_magic, cin_mshift, esz_shift;   // contiguous-index block length (any value, magic division); log2(element bytes)
};
struct GemmK {
    OpK A, B;
    void* C; void* C2; const void* R; const float* bias; float* colsum;
    long long c_bs1, c_bs2, c_rbstride, c_split_stride, bias_bs2;  // elements
    unsigned c_rpb, c_magic, c_shift;
    int ldc, M, N, K, nb2, splitk, flags, vec_ok, group_m, tile_m, debug;
    float alpha, drop_p;
    unsigned drop_seed;
};

__device__ __forceinline__ unsigned udiv_magic(unsigned n, unsigned magic, unsigned shift) {
    return (unsigned)(((unsigned long long)__umulhi(magic, n) + n) >> shift);
}
__device__ __forceinline__ unsigned row_off(const OpK& o, unsigned r) {
    const unsigned q = udiv_magic(r, o.rpb_magic, o.rpb_shift);
    return q * o.rb_bytes + (r - q * o.rpb) * o.ld_bytes;
}
__device__ __forceinline__ unsigned col_off(const OpK& o, unsigned c) {
    const unsigned q = udiv_magic(c, o.cin_magic, o.cin_mshift);
    return q * o.cout_bytes + ((c - q * o.cin) << o.esz_shift);
}

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// buffer descriptor from PROVABLY wave-uniform words, or hipcc wraps every buffer_load in a waterfall loop
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const char* base) {
    const unsigned long long b = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, 0xFFFFFFFF, 0x00020000);
}

__device__ __forceinline__ u32x4 mask_tail(u32x4 v, int nvalid) {  // keep the first nvalid (<= 8) bf16
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const unsigned m = (2 * d + 1 < nvalid) ? 0xFFFFFFFFu : ((2 * d < nvalid) ? 0x0000FFFFu : 0u);
        v[d] &= m;
    }
    return v;
}

// workgroup id -> output tile.  (1) XCD-aware: blocks b and b+8 share an XCD (round-robin dispatch), so every XCD gets a
// contiguous run of tile ids; (2) grouped order inside the run: 8 tile-rows are walked for one tile-column before moving to
// the next column, so the ~64 tiles resident on an XCD at a time touch 8 A row-panels and 8 B column-panels (~4 MiB = its L2).
// SEQ: `bid` already IS the position in the tile sequence (the grouped launch lays ITS blocks out XCD-major over all members).
template <bool SEQ = false>
__device__ __forceinline__ void tile_coords(int bid, int ntiles, int tiles_m, int tiles_n, int& tm, int& tn, int GROUP_M = 8) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = ntiles >> 3, r = ntiles & 7;
    int tile = SEQ ? bid : (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
#ifdef SCL_EXPERIMENTS
    // A/B of the XCD ownership axis (round 6, tools/xcd_order_probe.sh): bits 8..11 of the group word = R, the chip's 8 XCDs as an
    // R x (8 / R) grid of regions over the tile matrix (R row bands x 8 / R column bands).  The tile sequence walks region after region
    // (grouped order inside each), and XCD x still owns the x-th eighth of that sequence, so its run is its region up to a few tiles.
    // R = 0: the shipped order (one grouped walk over the whole matrix).
    const int R = (GROUP_M >> 8) & 15;
    GROUP_M &= 255;
    int row0 = 0, col0 = 0;
    if (R > 0) {
        const int C = 8 / R;
        int left = tile, found = 0;
        for (int reg = 0; reg < 8; ++reg) {
            const int ri = reg / C, ci = reg - ri * C;
            const int r0 = (tiles_m * ri) / R, r1 = (tiles_m * (ri + 1)) / R, c0 = (tiles_n * ci) / C, c1 = (tiles_n * (ci + 1)) / C;
            const int cnt = (r1 - r0) * (c1 - c0);
            if (!found && left < cnt) { found = 1; row0 = r0; col0 = c0; tiles_m = r1 - r0; tiles_n = c1 - c0; tile = left; }
            if (!found) left -= cnt;
        }
    }
#endif
    const int per_group = GROUP_M * tiles_n;
    const int group = tile / per_group;
    const int first_m = group * GROUP_M;
    const int gsize = min(tiles_m - first_m, GROUP_M);
    const int in_group = tile - group * per_group;
    // block-uniform results, but the integer divisions above run on the vector ALU and the compiler keeps everything derived from them
    // (K offsets of the LDS-DMA pieces, output bases) in VGPRs: a buffer load whose scalar offset sits in a VGPR becomes a waterfall loop
    // (readfirstlane + compare + exec mask + branch, per instruction).  Hand the values back as SGPRs.
#ifdef SCL_EXPERIMENTS
    tm = __builtin_amdgcn_readfirstlane(row0 + first_m + in_group % gsize);
    tn = __builtin_amdgcn_readfirstlane(col0 + in_group / gsize);
#else
    tm = __builtin_amdgcn_readfirstlane(first_m + in_group % gsize);
    tn = __builtin_amdgcn_readfirstlane(in_group / gsize);
#endif
}

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) void lds_void;

// fragment of a K-contiguous tile: 16 rows x 32 k; lane l holds row (l&15), k = 8*(l>>4)+0..7
__device__ __forceinline__ bf16x8 frag_k(const char* tile, int rowblk, int ks, int lane) {
    const int row = rowblk * 16 + (lane & 15);
    const int c = 4 * ks + (lane >> 4);
    const char* p = tile + row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
    return *reinterpret_cast<const bf16x8*>(p);
}
// fragment of a transposed tile ([k][col]): the same register image, through ds_read_b64_tr_b16
__device__ __forceinline__ bf16x8 frag_t(const char* tile, int colblk, int ks, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const int krow = 32 * ks + 8 * g + (i >> 2);
    const int sw = (krow & 3) | (((krow >> 3) & 1) << 2);
    const char* p = tile + krow * 256 + ((colblk ^ sw) << 5) + ((i & 3) << 3);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * 256));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// The same fragment for kernels that keep LDS-DMA writes in flight across their ds_reads: hipcc (ROCm 7.2) cannot prove
// that the tr-read builtin does not alias a pending `buffer_load ... lds` and puts `s_waitcnt vmcnt(0)` in front of it, which
// drains the prefetch every K step (seen in the ISA of the dma / big / p8 kernels; plain ds_read_b128 is not affected).
// Issued as inline asm the read is invisible to that pass — and to the compiler's lgkmcnt bookkeeping, so the CALLER must
// execute `s_waitcnt lgkmcnt(0)` between these reads and the first use of their results.
__device__ __forceinline__ bf16x8 frag_t_raw(const char* tile, int colblk, int ks, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const int krow = 32 * ks + 8 * g + (i >> 2);
    const int sw = (krow & 3) | (((krow >> 3) & 1) << 2);
    const char* p = tile + krow * 256 + ((colblk ^ sw) << 5) + ((i & 3) << 3);
    const unsigned a = (unsigned)(uintptr_t)(lds_void*)(p);
    s16x4 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(hi) : "v"(a) : "memory");
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// The wait that belongs to frag_t_raw reads.  A bare `asm volatile("s_waitcnt lgkmcnt(0)")` orders memory operations only: an MFMA has
// none, so the scheduler may lift one ABOVE the wait (found in the ISA of scl_gemm_dma_kernel<*, *>: one MFMA of the second k sub-step
// read a fragment ~25 instructions after its ds_read_b64_tr_b16 and before the wait — right almost always, a stale operand when LDS is
// slow).  Here the fragments are read-write operands of the wait, so every use of them depends on it.
__device__ __forceinline__ void lds_wait_frags(bf16x8 (&a)[4], bf16x8 (&b)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) :: "memory");
}

// Counted LDS wait: at most n LDS operations (ds_read_b128 = 1, a frag_t_raw fragment = 2) still outstanding.  LDS operations retire in
// order, so this releases the OLDEST reads while younger ones (the other sub-step's fragments) stay in flight.  n is a constant after
// unrolling; the counter has 4 bits.
__device__ __forceinline__ void lds_wait_upto(int n) {
    // s_waitcnt simm16 on gfx9: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14; 0xC07F leaves vmcnt and expcnt unconstrained.
    // The builtin (not inline asm) so that the compiler's own wait insertion sees it and adds nothing redundant behind it.
    switch (n < 0 ? 0 : (n > 15 ? 15 : n)) {
        case 0: __builtin_amdgcn_s_waitcnt(0xC07F); break;
        case 1: __builtin_amdgcn_s_waitcnt(0xC17F); break;
        case 2: __builtin_amdgcn_s_waitcnt(0xC27F); break;
        case 3: __builtin_amdgcn_s_waitcnt(0xC37F); break;
        case 4: __builtin_amdgcn_s_waitcnt(0xC47F); break;
        case 5: __builtin_amdgcn_s_waitcnt(0xC57F); break;
        case 6: __builtin_amdgcn_s_waitcnt(0xC67F); break;
        case 7: __builtin_amdgcn_s_waitcnt(0xC77F); break;
        case 8: __builtin_amdgcn_s_waitcnt(0xC87F); break;
        case 9: __builtin_amdgcn_s_waitcnt(0xC97F); break;
        case 10: __builtin_amdgcn_s_waitcnt(0xCA7F); break;
        case 11: __builtin_amdgcn_s_waitcnt(0xCB7F); break;
        case 12: __builtin_amdgcn_s_waitcnt(0xCC7F); break;
        case 13: __builtin_amdgcn_s_waitcnt(0xCD7F); break;
        case 14: __builtin_amdgcn_s_waitcnt(0xCE7F); break;
        default: __builtin_amdgcn_s_waitcnt(0xCF7F); break;
    }
}

// element-wise epilogue for edge tiles / unaligned outputs (rare path, kept out of line)
struct EpiArgs { void* C; void* C2; const void* R; int N, flags; unsigned drop_seed; float drop_p; };
static __device__ __noinline__ void epi_scalar(EpiArgs d, float t, long long o, int col, const float* bias) {
    if (col >= d.N) return;
    const int flags = d.flags;
    const int act = (flags >> SCL_GEMM_ACT_SHIFT) & 0xF, rmode = (flags >> SCL_GEMM_RMODE_SHIFT) & 0xF, ract = (flags >> SCL_GEMM_RACT_SHIFT) & 0xF;
    if (flags & SCL_GEMM_HAS_BIAS) t += bias[col];
    float c2v = t;
    if (act == 5) gelu_both_f(t, t, c2v);
    if (flags & SCL_GEMM_HAS_C2) {
        if (flags & SCL_GEMM_C2_F32) reinterpret_cast<float*>(d.C2)[o] = c2v;
        else reinterpret_cast<bf16_t*>(d.C2)[o] = f2bf(c2v);
    }
    if (act != 5) t = act_f(act, t);
    float r = 0.f;
    if (rmode) r = (flags & SCL_GEMM_R_F32) ? reinterpret_cast<const float*>(d.R)[o] : bf2f(reinterpret_cast<const bf16_t*>(d.R)[o]);
    if (rmode == 2) t *= act_grad_f(ract, r);
    if (flags & SCL_GEMM_DROPOUT) t *= dropout_scale(d.drop_seed, (uint64_t)o, d.drop_p);
    if (rmode == 1) t += r;
    if (flags & SCL_GEMM_C_F32) reinterpret_cast<float*>(d.C)[o] = t;
    else reinterpret_cast<bf16_t*>(d.C)[o] = f2bf(t);
}

// Fused epilogue of one wave's [NMT x 16 rows] x [4 x 16 columns] accumulator block whose first element is C[mbase][nbase]:
// lane holds C[row = mbase + mt*16 + (lane&15)][col = nbase + nt*16 + 4*(lane>>4) + 0..3].  Rows >= mlimit (the end of the
// problem or of this tile's row range) and row blocks >= nmt are not written.
// EK: epilogue kind, as in gemm_w8_epi.h (0 generic: every flag tested per 16 x 16 sub-block, 16 times per wave; 1 bf16 [+ bias],
// 2 fc1 forward with the stored gelu', 3 x stored derivative, 4 f32 [+ bias] + f32 residual, 5 f32 [+ bias]: flags as constants).
template <int NMT, int EK>
__device__ __forceinline__ void gemm_epilogue_blk_k(const GemmK& d, f32x4 (&acc)[NMT][4], int mbase, int nbase, int mlimit, int nmt,
                                                    int z1, int z2, int ksplit, int lane) {
#pragma clang fp contract(off)      // as the generic form's basic blocks imply: no multiply-add of two different epilogue steps is fused
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
    const long long cbase = z1 * d.c_bs1 + z2 * d.c_bs2 + (long long)ksplit * d.c_split_stride;
    const float* bias = has_bias ? d.bias + z2 * d.bias_bs2 : nullptr;
    const int g = lane >> 4, lc = lane & 15;
    const EpiArgs ea = {d.C, d.C2, d.R, d.N, d.flags, d.drop_seed, d.drop_p};

#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) {
        if (mt < nmt) {
        const int row = mbase + mt * 16 + lc;
        const bool rok = row < mlimit;
        const unsigned q = udiv_magic((unsigned)(rok ? row : 0), d.c_magic, d.c_shift);
        const long long roff = cbase + (long long)q * d.c_rbstride + (long long)((unsigned)(rok ? row : 0) - q * d.c_rpb) * d.ldc;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int col = nbase + nt * 16 + 4 * g;
            const long long off = roff + col;
            float v[4] = {d.alpha * acc[mt][nt][0], d.alpha * acc[mt][nt][1], d.alpha * acc[mt][nt][2], d.alpha * acc[mt][nt][3]};
            if (rok && d.vec_ok && col + 4 <= d.N) {
                if (has_bias) {
                    const float4 bb = *reinterpret_cast<const float4*>(bias + col);
                    v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
                }
                float c2v[4] = {v[0], v[1], v[2], v[3]};
                if (act == 5) { gelu_both_f(v[0], v[0], c2v[0]); gelu_both_f(v[1], v[1], c2v[1]); gelu_both_f(v[2], v[2], c2v[2]); gelu_both_f(v[3], v[3], c2v[3]); }
                if (has_c2) {
                    if (c2_f32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(d.C2) + off) = make_float4(c2v[0], c2v[1], c2v[2], c2v[3]);
                    else *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(d.C2) + off) = make_uint2(pack_bf2(c2v[0], c2v[1]), pack_bf2(c2v[2], c2v[3]));
                }
                if (act && act != 5) {
                    v[0] = act_f(act, v[0]); v[1] = act_f(act, v[1]); v[2] = act_f(act, v[2]); v[3] = act_f(act, v[3]);
                }
                float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
                if (rmode) {
                    if (r_f32) {
                        const float4 t = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(d.R) + off);
                        r0 = t.x; r1 = t.y; r2 = t.z; r3 = t.w;
                    } else {
                        const uint2 t = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(d.R) + off);
                        r0 = __uint_as_float(t.x << 16); r1 = __uint_as_float(t.x & 0xFFFF0000u);
                        r2 = __uint_as_float(t.y << 16); r3 = __uint_as_float(t.y & 0xFFFF0000u);
                    }
                }
                if (rmode == 2 && ract == 4) {
                    v[0] *= r0; v[1] *= r1; v[2] *= r2; v[3] *= r3;
                } else if (rmode == 2) {
                    v[0] *= act_grad_f(ract, r0); v[1] *= act_grad_f(ract, r1); v[2] *= act_grad_f(ract, r2); v[3] *= act_grad_f(ract, r3);
                }
                if (drop) {
                    v[0] *= dropout_scale(d.drop_seed, (uint64_t)(off + 0), d.drop_p); v[1] *= dropout_scale(d.drop_seed, (uint64_t)(off + 1), d.drop_p);
                    v[2] *= dropout_scale(d.drop_seed, (uint64_t)(off + 2), d.drop_p); v[3] *= dropout_scale(d.drop_seed, (uint64_t)(off + 3), d.drop_p);
                }
                if (rmode == 1) { v[0] += r0; v[1] += r1; v[2] += r2; v[3] += r3; }
                if (c_f32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(d.C) + off) = make_float4(v[0], v[1], v[2], v[3]);
                else *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(d.C) + off) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
            } else if (rok) {
                // edge tile / unaligned C: element-wise path
                epi_scalar(ea, v[0], off + 0, col + 0, bias);
                epi_scalar(ea, v[1], off + 1, col + 1, bias);
                epi_scalar(ea, v[2], off + 2, col + 2, bias);
                epi_scalar(ea, v[3], off + 3, col + 3, bias);
            }
        }
        }
    }
}

template <int NMT>
__device__ __forceinline__ void gemm_epilogue_blk(const GemmK& d, f32x4 (&acc)[NMT][4], int mbase, int nbase, int mlimit, int nmt,
                                                  int z1, int z2, int ksplit, int lane) {
    constexpr int KEY = SCL_GEMM_C_F32 | SCL_GEMM_C2_F32 | SCL_GEMM_R_F32 | SCL_GEMM_HAS_BIAS | SCL_GEMM_HAS_C2 | SCL_GEMM_DROPOUT |
                        (0xF << SCL_GEMM_ACT_SHIFT) | (0xF << SCL_GEMM_RMODE_SHIFT) | (0xF << SCL_GEMM_RACT_SHIFT);
    const int f = d.flags & KEY, fb = f & ~SCL_GEMM_HAS_BIAS;
    if (!(d.debug & 16)) {      // debug bit 4: SCL_W8_EPI_GENERIC=1 (A/B)
        if (fb == 0) return gemm_epilogue_blk_k<NMT, 1>(d, acc, mbase, nbase, mlimit, nmt, z1, z2, ksplit, lane);
        if (f == (SCL_GEMM_HAS_BIAS | SCL_GEMM_HAS_C2 | (5 << SCL_GEMM_ACT_SHIFT))) return gemm_epilogue_blk_k<NMT, 2>(d, acc, mbase, nbase, mlimit, nmt, z1, z2, ksplit, lane);
        if (f == ((2 << SCL_GEMM_RMODE_SHIFT) | (4 << SCL_GEMM_RACT_SHIFT))) return gemm_epilogue_blk_k<NMT, 3>(d, acc, mbase, nbase, mlimit, nmt, z1, z2, ksplit, lane);
        if (fb == (SCL_GEMM_C_F32 | SCL_GEMM_R_F32 | (1 << SCL_GEMM_RMODE_SHIFT))) return gemm_epilogue_blk_k<NMT, 4>(d, acc, mbase, nbase, mlimit, nmt, z1, z2, ksplit, lane);
        if (fb == SCL_GEMM_C_F32) return gemm_epilogue_blk_k<NMT, 5>(d, acc, mbase, nbase, mlimit, nmt, z1, z2, ksplit, lane);
    }
    gemm_epilogue_blk_k<NMT, 0>(d, acc, mbase, nbase, mlimit, nmt, z1, z2, ksplit, lane);
}

// the 64x64 per-wave block of the 128x128 / 256x128 / 256x256 kernels
__device__ __forceinline__ void gemm_epilogue(const GemmK& d, f32x4 (&acc)[4][4], int m0, int n0, int z1, int z2, int ksplit,
                                              int lane, int wr, int wc) {
    gemm_epilogue_blk<4>(d, acc, m0 + wr * 64, n0 + wc * 64, d.M, 4, z1, z2, ksplit, lane);
}





// gemm.hip (host side)
void make_magic(unsigned dv, unsigned* magic, unsigned* shift);
bool fill_operand(const SclOperand& o, const char* name, long long rows, long long contig, OpK* k, int esz);
int scl_gemm_f32_launch(const SclGemmDesc& d, GemmK& k, hipStream_t s);   // gemm_f32.hip

// gemm_w8.hip (host side)
struct W8Plan { int variant, tiles_m, tile_m; long long tiles, cost; int ncu = 0; };
bool scl_gemm_w8_plan(const GemmK& k, bool at, bool bt, const SclGemmDesc& d, long long zdim, int ncu, W8Plan* plan);
int scl_gemm_read_stamps(unsigned long long* out, int nblocks);
long long scl_gemm_w8p_launches();
int scl_gemm_w8_launch(GemmK& k, bool at, bool bt, const W8Plan& plan, long long zdim, hipStream_t s);
constexpr int W8_GROUP_MAX = 8;      // members of a grouped launch (scl_gemm_bf16_group / _part)
bool scl_gemm_w8_group_member_ok(const GemmK& k, bool at, bool bt, const SclGemmDesc& d);
int scl_gemm_w8_group_tiles(const GemmK& k);
int scl_gemm_w8_group_launch(GemmK* ks, int n, const int* tile0, const int* ntile, hipStream_t s);

// gemm_x2.hip (host side): 208 x 128 tiles, two 4-wave workgroups per CU (plan->variant = 2)
bool scl_gemm_x2_plan(const GemmK& k, bool at, bool bt, const SclGemmDesc& d, long long zdim, int ncu, W8Plan* plan);
int scl_gemm_x2_launch(GemmK& k, bool at, bool bt, const W8Plan& plan, long long zdim, hipStream_t s);

}  // namespace sclg
