// attention.hip — row softmax (fp32 statistics) between the QK^T and PV contractions of
// fairseq MultiheadAttention (reached from model/xlsr.py:41), forward and backward.
//
//   fwd:  P[r][j]  = softmax_j S[r][j]                       S f32 [R, T] (ld = lds), P bf16 [R, Tp]
//   bwd:  dS[r][j] = P[r][j] * (dP[r][j] - sum_j dP[r][j] P[r][j])
//
// One wave per row (T <= 4*64 per pass, looped for longer rows); columns T..Tp-1 of P / dS are
// written as zero so the following transposed-operand GEMMs can read whole 16-byte vectors.
#include "common.h"

namespace {

constexpr int MAXV = 8;  // up to 512 columns kept in registers

template <typename TP>      // TP: bf16_t (training path: P is a GEMM operand) or float (fp32 scoring path)
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ S, TP* __restrict__ P, int64_t R,
                                                          int T, int ldS, int Tp) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const float* s = S + row * ldS;
    float v[MAXV];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 64 + lane;
        v[i] = c < T ? s[c] : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 64 + lane;
        v[i] = c < T ? __expf(v[i] - mx) : 0.f;
        sum += v[i];
    }
    const float inv = 1.0f / wave_sum(sum);
    TP* p = P + row * Tp;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < Tp) {
            if constexpr (sizeof(TP) == 4) p[c] = v[i] * inv;
            else p[c] = f2bf(v[i] * inv);
        }
    }
}

__global__ __launch_bounds__(256) void softmax_bwd_kernel(const bf16_t* __restrict__ P, const float* __restrict__ dP,
                                                          bf16_t* __restrict__ dS, int64_t R, int T, int lddP, int Tp) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const bf16_t* p = P + row * Tp;
    const float* dp = dP + row * lddP;
    float pv[MAXV], dv[MAXV];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 64 + lane;
        pv[i] = c < T ? bf2f(p[c]) : 0.f;
        dv[i] = c < T ? dp[c] : 0.f;
        dot += pv[i] * dv[i];
    }
    dot = wave_sum(dot);
    bf16_t* o = dS + row * Tp;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < Tp) o[c] = f2bf(pv[i] * (dv[i] - dot));
    }
}

}  // namespace

extern "C" int scl_softmax_fwd(const float* S, void* P, int64_t R, int T, int ldS, int Tp, void* stream) {
    SCL_REQUIRE(S && P && R > 0 && T > 0 && T <= 512 && Tp >= T && Tp <= 512 && (Tp & 7) == 0, "softmax_fwd: need T <= Tp <= 512, Tp %% 8 == 0");
    hipLaunchKernelGGL(softmax_fwd_kernel<bf16_t>, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, S, (bf16_t*)P, R, T, ldS, Tp);
    return scl_check_launch("scl_softmax_fwd");
}

extern "C" int scl_softmax_fwd_f32(const float* S, float* P, int64_t R, int T, int ldS, int Tp, void* stream) {
    SCL_REQUIRE(S && P && R > 0 && T > 0 && T <= 512 && Tp >= T && Tp <= 512 && (Tp & 3) == 0, "softmax_fwd_f32: need T <= Tp <= 512, Tp %% 4 == 0");
    hipLaunchKernelGGL(softmax_fwd_kernel<float>, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, S, P, R, T, ldS, Tp);
    return scl_check_launch("scl_softmax_fwd_f32");
}

extern "C" int scl_softmax_bwd(const void* P, const float* dP, void* dS, int64_t R, int T, int lddP, int Tp, void* stream) {
    SCL_REQUIRE(P && dP && dS && R > 0 && T > 0 && T <= 512 && Tp >= T && Tp <= 512 && (Tp & 7) == 0, "softmax_bwd: need T <= Tp <= 512, Tp %% 8 == 0");
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)P, dP, (bf16_t*)dS, R, T, lddP, Tp);
    return scl_check_launch("scl_softmax_bwd");
}

// =====================================================================================================
// Fused attention (head dim 64, T <= 256): scores never leave the chip.
//
// Forward, one workgroup per (utterance, head), 4 waves.  K and V of the head are staged once in LDS
// (K with the GEMM's 16-byte-slot swizzle for ds_read_b128 row reads, V with a 32-byte-chunk swizzle for
// ds_read_b64_tr_b16 transposed reads).  A wave takes 16 queries at a time:
//   S^T[key][q] = K Q^T      MFMA(A = K rows from LDS, B = Q rows straight from global)  -> lane owns ONE query
//                            (column q = lane&15) and keys 16t + 4*(lane>>4) + reg: the row softmax is an in-lane
//                            reduction plus two cross-lane steps (xor 16, xor 32), fp32, exp via the scaled max.
//   O^T[d][q]   = V^T P^T    the normalised probabilities, packed to bf16, ARE the B operand of the next MFMA
//                            (k index permuted; the A operand = V^T is fetched with the same permutation by the
//                            transposed LDS read), so P never goes through LDS.
// Writes ctx (bf16, [B,T,H*64] slice) and the row log-sum-exp (fp32) that the backward recomputes P from.
// Reference arithmetic: fairseq MultiheadAttention inside Wav2Vec2Model.forward (model/xlsr.py:41):
// softmax((q*scale) k^T) v with fp32 softmax.
// =====================================================================================================
namespace {

typedef __attribute__((address_space(3))) s16x4 lds_s16x4_a;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4a;
constexpr int ATT_D = 64;
constexpr int ATT_NTMAX = 16;   // key tiles of 16 -> T <= 256

__device__ __forceinline__ int att_k_off(int key, int c) { return key * 128 + ((c ^ ((key >> 1) & 7)) << 4); }          // row-read image
__device__ __forceinline__ int att_t_off(int key, int d) { return key * 128 + ((((d >> 4) ^ ((key >> 1) & 3))) << 5) + ((d & 15) << 1); }  // tr-read image

__device__ __forceinline__ bf16x8 att_frag_rows(const char* tile, int rowblk, int ks, int lane) {
    const int row = rowblk * 16 + (lane & 15);
    return *reinterpret_cast<const bf16x8*>(tile + att_k_off(row, 4 * ks + (lane >> 4)));
}
// A operand [i = d (16 of them, block dt)][k = 8 keys]: keys rowa + 4g + 0..3 and rowb + 4g + 0..3
__device__ __forceinline__ bf16x8 att_frag_tr(const char* tile, int rowa, int rowb, int dt, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const int ra = rowa + 4 * g + (i >> 2), rb = rowb + 4 * g + (i >> 2);
    const int col = 16 * dt + 4 * (i & 3);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_a*)(tile + att_t_off(ra, col)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_a*)(tile + att_t_off(rb, col)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// One 16-query tile of the forward against the staged K / V images: scores, soft-max, P V, ctx and log-sum-exp stores.
template <bool DROP, int NTC>
__device__ __forceinline__ void att_fwd_tile(const char* Kt, const char* Vt, bf16x8 (&qf)[2], const bf16_t* qnext, int qb, int b, int h, int T, int H,
                                             int NT, float scale, float sl2, float drop_p, uint32_t drop_seed, bf16_t* __restrict__ ctx,
                                             float* __restrict__ lse, int lane) {
    constexpr int NTB = NTC ? NTC : ATT_NTMAX, NT2B = (NTB + 1) / 2;      // unroll bounds
    const int NT2 = (NT + 1) / 2, E = H * ATT_D;
    const int lc = lane & 15, g = lane >> 4;
    const int q = qb * 16 + lc;
    f32x4 s[NTB + 1];
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < NTB; ++t) {
        s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t < NT) {
#ifdef ATT_ABL_NOQK
            s[t] = f32x4{(float)lane, __builtin_bit_cast(float, __builtin_bit_cast(u32x4a, qf[0])[0]), 2.f, (float)t};
#else
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_frag_rows(Kt, t, ks, lane), qf[ks], s[t], 0, 0, 0);
#endif
            if (t == NT - 1) {      // only the last key tile can hold keys >= T (the softmax below is VALU-bound: no per-element test elsewhere)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (16 * t + 4 * g + r >= T) s[t][r] = -INFINITY;
            }
            m = fmaxf(m, fmaxf(fmaxf(s[t][0], s[t][1]), fmaxf(s[t][2], s[t][3])));
            if (NTC && (t & 1)) __builtin_amdgcn_sched_barrier(0);      // keeps the K-fragment reads of later tiles from being hoisted (128-register budget)
        }
    }
#ifndef ATT_NO_QPRE
    // the query fragments are dead from here: the lane's row of the NEXT tile this wave will work on (qnext; null = none) travels
    // under the soft-max and the P V products
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        uint4 u = make_uint4(0, 0, 0, 0);
        if (qnext) u = *reinterpret_cast<const uint4*>(qnext + 32 * ks + 8 * g);
        qf[ks] = __builtin_bit_cast(bf16x8, u);
    }
#endif
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
    const float msl = -m * sl2;
#pragma unroll
    for (int t = 0; t < NTB; ++t) {
        if (t < NT) {
#pragma unroll
#ifdef ATT_ABL_NOEXP
            for (int r = 0; r < 4; ++r) { s[t][r] = __builtin_fmaf(s[t][r], sl2, msl) * 1e-3f; l += s[t][r]; }
#else
            for (int r = 0; r < 4; ++r) { s[t][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][r], sl2, msl)); l += s[t][r]; }      // = exp(scale * (s - m))
#endif
        }
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    if (g == 0 && q < T) lse[((int64_t)b * H + h) * T + q] = scale * m + __logf(l);
    if (DROP) {      // attention dropout (fairseq MultiheadAttention dropout_module on the probabilities): keep-mask by (row, key)
        const uint64_t rowbase = (((uint64_t)b * H + h) * T + (uint64_t)(q < T ? q : 0)) * T;
#pragma unroll
        for (int t = 0; t < NTB; ++t) {
            if (t < NT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) s[t][r] *= dropout_scale(drop_seed, rowbase + (uint64_t)(16 * t + 4 * g + r), drop_p);
            }
        }
    }
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NT2B; ++u) {
        if (u < NT2) {
            const int ta = 2 * u, tb = 2 * u + 1;
            // the un-normalised exponentials (<= 1) are the B operand; 1 / l multiplies the 16 outputs instead of the 4 NT probabilities
            float pb[4] = {0.f, 0.f, 0.f, 0.f};
            if (tb < NT) { pb[0] = s[tb][0]; pb[1] = s[tb][1]; pb[2] = s[tb][2]; pb[3] = s[tb][3]; }
            u32x4a pk;
            pk[0] = pack_bf2(s[ta][0], s[ta][1]); pk[1] = pack_bf2(s[ta][2], s[ta][3]);
            pk[2] = pack_bf2(pb[0], pb[1]); pk[3] = pack_bf2(pb[2], pb[3]);
            const bf16x8 pf = __builtin_bit_cast(bf16x8, pk);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#ifdef ATT_ABL_NOPV
                o[dt][0] += __builtin_bit_cast(float, pk[dt]);
#else
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_frag_tr(Vt, 16 * ta, 16 * tb, dt, lane), pf, o[dt], 0, 0, 0);
#endif
            if (NTC) __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (q < T) {
        bf16_t* dst = ctx + ((int64_t)b * T + q) * E + h * ATT_D + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            *reinterpret_cast<uint2*>(dst + 16 * dt) = make_uint2(pack_bf2(o[dt][0] * inv, o[dt][1] * inv), pack_bf2(o[dt][2] * inv, o[dt][3] * inv));
    }
}

// NTC: number of 16-key tiles as a compile-time constant (0 = run time).  With a run-time tile count every step of the unrolled tile
// loops is predicated (v_cndmask on 4 x 16 score registers, the tile counter and masks in SGPRs that spill to lanes: 150 cndmask +
// 300 readlane / writelane in a 1300-instruction kernel that is VALU-bound); the encoder's lengths give 13 tiles (T = 193 .. 208: 64000- and
// 64600-sample clips) or 4 (16000-sample clips), everything else takes the generic form.
template <bool DROP, int NTC>      // attention dropout compiled in only where asked for: its index arithmetic pushes the plain kernel over its 128 registers
__global__ __launch_bounds__(512, DROP ? 2 : 4) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx, float* __restrict__ lse,
                                                       int T, int H, float scale, float drop_p, uint32_t drop_seed) {
    extern __shared__ __attribute__((aligned(16))) char asmem[];
    const int E = H * ATT_D;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int NT = NTC ? NTC : (T + 15) / 16, NT2 = (NT + 1) / 2, rows = 32 * NT2;
    char* Kt = asmem;                 // [rows][128 B] row-read image
    char* Vt = asmem + rows * 128;    // [rows][128 B] tr-read image
    const bf16_t* base = qkv + (int64_t)b * T * 3 * E + h * ATT_D;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int lc = lane & 15, g = lane >> 4;
    // the wave's first query tile is requested ahead of the K / V staging (one memory round trip instead of two in a row), the next one
    // while the current tile is multiplied
    bf16x8 qf[2];
    auto load_q = [&](int qb, bf16x8 (&dst)[2]) {
        const int q = qb * 16 + lc;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 u = make_uint4(0, 0, 0, 0);
            if (q < T) u = *reinterpret_cast<const uint4*>(base + (int64_t)q * 3 * E + 32 * ks + 8 * g);
            dst[ks] = __builtin_bit_cast(bf16x8, u);
        }
    };
#ifndef ATT_NO_QPRE
    load_q(wave, qf);
#endif
    // K / V staging: all of a thread's loads are requested before the first LDS write (a rolled load -> write loop paid one memory
    // round trip per iteration; rows * 8 <= 2048 vectors = 4 per thread)
    {
        uint4 kv[4], vv[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = threadIdx.x + 512 * it, key = idx >> 3, c = idx & 7;
            kv[it] = make_uint4(0, 0, 0, 0); vv[it] = make_uint4(0, 0, 0, 0);
#ifndef ATT_ABL_NOSTAGE
            if (idx < rows * 8 && key < T) {
#else
            if (idx < rows * 8 && key < T && T < 0) {
#endif
                kv[it] = *reinterpret_cast<const uint4*>(base + (int64_t)key * 3 * E + E + 8 * c);
                vv[it] = *reinterpret_cast<const uint4*>(base + (int64_t)key * 3 * E + 2 * E + 8 * c);
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = threadIdx.x + 512 * it, key = idx >> 3, c = idx & 7;
            if (idx < rows * 8) {
                *reinterpret_cast<uint4*>(Kt + att_k_off(key, c)) = kv[it];
                *reinterpret_cast<uint4*>(Vt + att_t_off(key, 8 * c)) = vv[it];
            }
        }
    }
    __syncthreads();
    const float sl2 = scale * 1.4426950408889634f;
    for (int qb = wave; qb < NT; qb += (int)(blockDim.x >> 6)) {      // 8 waves: two blocks per CU = 4 waves per SIMD
#ifdef ATT_NO_QPRE
        load_q(qb, qf);
#endif
        const int qn = (qb + (int)(blockDim.x >> 6)) * 16 + lc;      // this lane's query of the wave's next tile
        const bf16_t* qnext = (qb + (int)(blockDim.x >> 6) < NT && qn < T) ? base + (int64_t)qn * 3 * E : nullptr;
        att_fwd_tile<DROP, NTC>(Kt, Vt, qf, qnext, qb, b, h, T, H, NT, scale, sl2, drop_p, drop_seed, ctx, lse, lane);
    }
}


// Workgroup barrier for LDS hand-offs only: __syncthreads() also drains vmcnt (its release fence covers global memory), which made
// every step of the backward wait for the acknowledgement of its dQ stores (barrier A), for the query-tile loads it had just
// requested for the NEXT step (barrier B), and the bias-sum tail for the dK / dV stores.  Here: this wave's LDS operations retired,
// then s_barrier; global loads stay in flight (the compiler still waits for them where their registers are first used).
#define ATT_LDS_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
// -----------------------------------------------------------------------------------------------------
// Backward, one workgroup per (utterance, head), T <= 224.  A wave OWNS a block of keys and keeps dK^T and dV^T of those keys
// in accumulator registers while the workgroup sweeps the queries 32 at a time:
//   S[q][key] = Q K^T, dP[q][key] = dO V^T         (lane owns one key column, registers hold 4 queries)
//   P = exp(scale*S - LSE[q]),  dS = P * (dP - delta[q]),  delta[q] = <dO[q], O[q]>   (computed here, in the tile loader)
//   dV^T += dO^T P,  dK^T += Q^T dS                (P / dS accumulators ARE the B operands; dO^T / Q^T by transposed LDS reads)
//   dQ    = dS K                                   (dS crosses LDS once)
// Scores, probabilities and their gradients never touch HBM.  dQ/dK carry the softmax scale.
// -----------------------------------------------------------------------------------------------------
__device__ __forceinline__ bf16x8 att_frag_tr_nat(const char* tile, int base, int dt, int lane) {   // k = base + 8g + 0..7
    const int i = lane & 15, g = lane >> 4;
    const int ra = base + 8 * g + (i >> 2);
    const int col = 16 * dt + 4 * (i & 3);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_a*)(tile + att_t_off(ra, col)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_a*)(tile + att_t_off(ra + 4, col)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 att_pack8(const f32x4& a, const f32x4& b) {
    u32x4a pk;
    pk[0] = pack_bf2(a[0], a[1]); pk[1] = pack_bf2(a[2], a[3]); pk[2] = pack_bf2(b[0], b[1]); pk[3] = pack_bf2(b[2], b[3]);
    return __builtin_bit_cast(bf16x8, pk);
}

// -----------------------------------------------------------------------------------------------------
// The launched form: 8 waves, so that TWO waves share each SIMD (a first 4-wave form — 64 keys per wave, 388 registers, one wave
// per SIMD, per-wave f32 dQ partials summed through LDS — exposed every LDS / global latency: 8 % MFMA busy, 90 us; this: 55 us):
//   * wave w OWNS keys 32w..32w+31: dK^T / dV^T of those keys in 64 accumulators (was 128), P / dS of 32 queries x 32 keys in 32;
//   * dS of all waves goes to ONE LDS image of eight [32 q][32 keys] sub-images (sub-image = owning wave = one 32-deep k step);
//   * dQ is produced without partial sums: wave w computes the [16 q x 16 d] output tile (q-tile w&1, d-tile w>>1) over ALL keys
//     (<= 8 MFMAs) and stores it, so the 32 KiB of per-wave f32 partials and the 4-wave combine pass are gone;
//   * wave 7 owns no keys (T <= 224): it alone stages the query tiles (Q, dO, O for delta, LSE), fetched two steps ahead;
//   * query tiles and the dS image are double-buffered, so a step has ONE barrier:  [S, dP, P, dS, dV, dK of tile u | stage tile u+1]
//     barrier  [dQ of tile u], and a wave runs dQ(u) and the MFMA phase of u+1 back to back;
//   * barriers are LDS-only (s_waitcnt lgkmcnt(0); s_barrier): global loads stay in flight across them;
//   * dK / dV leave through the wave's own K / V LDS rows as whole 128-byte rows; the 8- and 16-lane sums run on the DPP path.
// Stamps of each stage of this restructuring (28.8 -> 22.6 us per block, 111.9 -> 94.7 us per launch): profiles/r3_attn_bwd_stamps.txt.
// LDS: K (row image), K (tr image), V (row image) + 2 x 16 KiB query tiles + 2 x 16 KiB dS = 148.5 KiB for T = 199.
// -----------------------------------------------------------------------------------------------------
__device__ __forceinline__ int att_s_off(int q, int key32) {   // dS sub-image [32 q][32 keys], 64-B rows, 16-B chunk ^= (q>>2)&3
    return q * 64 + ((((key32 >> 3) ^ (q >> 2)) & 3) << 4) + ((key32 & 7) << 1);
}

template <bool DROP, int NTC>      // attention dropout compiled in only where asked for; NTC: compile-time key-tile count (0 = run time), as in attn_fwd_kernel
__global__ __launch_bounds__(512, 2) void attn_bwd8_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ ctx,
                                                           const bf16_t* __restrict__ dctx, const float* __restrict__ lse,
                                                           bf16_t* __restrict__ dqkv, float* __restrict__ bias_part, int T, int H, float scale,
                                                           float drop_p, uint32_t drop_seed) {
    extern __shared__ __attribute__((aligned(16))) char asmem[];
    const int E = H * ATT_D;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int NT = NTC ? NTC : (T + 15) / 16, NT2 = (NT + 1) / 2, rows = 32 * NT2;
    char* Kk = asmem;
    char* Kt = Kk + rows * 128;
    char* Vk = Kt + rows * 128;
    char* QO = Vk + rows * 128;            // 2 buffers x {Q rows, Q tr, dO rows, dO tr} of 4 KiB, then 2 dS images of 16 KiB
    float* lseS = reinterpret_cast<float*>(QO + 65536);   // 2 x {lse[32], delta[32]}
    const bf16_t* base = qkv + (int64_t)b * T * 3 * E + h * ATT_D;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0..7
    // Staging role: wave 7 never owns keys (T <= 224 = 14 key tiles, two per wave), so it alone carries the query tiles — Q, dO, O
    // (for delta) and LSE of 32 queries, four 16-byte pieces of each per lane, fetched two steps ahead of their use and written to the
    // idle tile buffer while waves 0-6 are in their MFMA phase.
    const bool stager = wave == 7;
    const int sc = lane & 7, sr0 = lane >> 3;      // piece i of this lane: query row sr0 + 8 i, 16-byte chunk sc
    uint4 vq[4], vo[4], vc[4];
    float lqn[4];
    auto fetch_tile = [&](int u) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = 32 * u + sr0 + 8 * i;
            vq[i] = make_uint4(0, 0, 0, 0); vo[i] = make_uint4(0, 0, 0, 0); vc[i] = make_uint4(0, 0, 0, 0);
            lqn[i] = 1e30f;      // rows past T: exp(0 - huge) = 0, no per-element row test
            if (q < T) {
                vq[i] = *reinterpret_cast<const uint4*>(base + (int64_t)q * 3 * E + 8 * sc);
                vo[i] = *reinterpret_cast<const uint4*>(dctx + ((int64_t)b * T + q) * E + h * ATT_D + 8 * sc);
                vc[i] = *reinterpret_cast<const uint4*>(ctx + ((int64_t)b * T + q) * E + h * ATT_D + 8 * sc);
                if (sc == 0) lqn[i] = lse[((int64_t)b * H + h) * T + q];
            }
        }
    };
    if (stager) fetch_tile(0);
    {   // all loads first, then the LDS writes (see attn_fwd_kernel); rows * 8 <= 1792 vectors = 4 per thread
        uint4 kv[4], vv[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = threadIdx.x + 512 * it, key = idx >> 3, c = idx & 7;
            kv[it] = make_uint4(0, 0, 0, 0); vv[it] = make_uint4(0, 0, 0, 0);
            if (idx < rows * 8 && key < T) {
                kv[it] = *reinterpret_cast<const uint4*>(base + (int64_t)key * 3 * E + E + 8 * c);
                vv[it] = *reinterpret_cast<const uint4*>(base + (int64_t)key * 3 * E + 2 * E + 8 * c);
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = threadIdx.x + 512 * it, key = idx >> 3, c = idx & 7;
            if (idx < rows * 8) {
                *reinterpret_cast<uint4*>(Kk + att_k_off(key, c)) = kv[it];
                *reinterpret_cast<uint4*>(Kt + att_t_off(key, 8 * c)) = kv[it];
                *reinterpret_cast<uint4*>(Vk + att_k_off(key, c)) = vv[it];
            }
        }
    }
    const int lc = lane & 15, g = lane >> 4;
    int nkt = NT - 2 * wave; nkt = nkt < 0 ? 0 : (nkt > 2 ? 2 : nkt);     // 16-key tiles this wave owns

    f32x4 dVt[4][2], dKt[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { dVt[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; dKt[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    f32x4 dqsum = f32x4{0.f, 0.f, 0.f, 0.f};      // bias_part: column sums of this wave's dQ tiles over the query steps

    // One barrier per step: the query tiles and the dS image are double-buffered (buffer = step parity), so while a step's dS is
    // being consumed by the dQ tiles (after the barrier) nothing a later step writes can touch it, and the NEXT step's query tiles are
    // staged before the same barrier.  A wave therefore runs  dQ(u) | S, dP, P, dS, dV, dK (u+1)  back to back without a barrier in
    // between (the latency-bound dQ chain of one wave overlaps the fragment reads of the next step).
    auto stage_tile = [&](int p) {      // registers (tile fetched earlier) -> tile buffer p
        char* qo = QO + p * 16384;
        float* ls = lseS + p * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int sr = sr0 + 8 * i;
            *reinterpret_cast<uint4*>(qo + att_k_off(sr, sc)) = vq[i];
            *reinterpret_cast<uint4*>(qo + 4096 + att_t_off(sr, 8 * sc)) = vq[i];
            *reinterpret_cast<uint4*>(qo + 8192 + att_k_off(sr, sc)) = vo[i];
            *reinterpret_cast<uint4*>(qo + 12288 + att_t_off(sr, 8 * sc)) = vo[i];
            const unsigned ow[4] = {vo[i].x, vo[i].y, vo[i].z, vo[i].w}, cw[4] = {vc[i].x, vc[i].y, vc[i].z, vc[i].w};
            float dot = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                dot += __uint_as_float(ow[k] << 16) * __uint_as_float(cw[k] << 16);
                dot += __uint_as_float(ow[k] & 0xFFFF0000u) * __uint_as_float(cw[k] & 0xFFFF0000u);
            }
            dot = lanes8_sum(dot);
            if (sc == 0) { ls[32 + sr] = dot; ls[sr] = lqn[i]; }
        }
    };
    auto main_phase = [&](int u) {      // S, dP, P, dS, dV^T, dK^T of query tile u (tile buffer u & 1) and this wave's keys; dS -> image u & 1
        const int p = u & 1;
        const char* Qk = QO + p * 16384;
        const char* Qt = Qk + 4096;
        const char* Ok = Qk + 8192;
        const char* Ot = Qk + 12288;
        char* dSs = QO + 32768 + p * 16384;      // [8 waves][32 q][64 B]
        char* dSw = dSs + wave * 2048;
        const float* lseP = lseS + p * 64;
        const float* delP = lseP + 32;
        {
            f32x4 P[2][2], dS[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int j = 0; j < 2; ++j) { P[a][j] = f32x4{0.f, 0.f, 0.f, 0.f}; dS[a][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            if (nkt > 0) {
                bf16x8 qa[2][2], oa[2][2];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) { qa[a][ks] = att_frag_rows(Qk, a, ks, lane); oa[a][ks] = att_frag_rows(Ok, a, ks, lane); }
                float lq[2][4], dq_[2][4];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const float4 l4 = *reinterpret_cast<const float4*>(lseP + 16 * a + 4 * g), d4 = *reinterpret_cast<const float4*>(delP + 16 * a + 4 * g);
                    lq[a][0] = l4.x * 1.4426950408889634f; lq[a][1] = l4.y * 1.4426950408889634f; lq[a][2] = l4.z * 1.4426950408889634f; lq[a][3] = l4.w * 1.4426950408889634f;
                    dq_[a][0] = d4.x; dq_[a][1] = d4.y; dq_[a][2] = d4.z; dq_[a][3] = d4.w;
                }
                const float sc2 = scale * 1.4426950408889634f;      // P = 2^(sc2 * S - lse * log2 e): one fma + v_exp per element
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (j < nkt) {
                        const int kb = 2 * wave + j;
                        const bf16x8 k0 = att_frag_rows(Kk, kb, 0, lane), k1 = att_frag_rows(Kk, kb, 1, lane);
                        const bf16x8 vv0 = att_frag_rows(Vk, kb, 0, lane), vv1 = att_frag_rows(Vk, kb, 1, lane);
                        const int key = 16 * kb + lc;
                        const bool key_dead = key >= T;      // only the last key tile can hold one (K rows past T are zeros: S = 0, P would be e^-lse)
#pragma unroll
                        for (int a = 0; a < 2; ++a) {
                            f32x4 sv = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
                            sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[a][0], k0, sv, 0, 0, 0);
                            sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[a][1], k1, sv, 0, 0, 0);
                            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(oa[a][0], vv0, dp, 0, 0, 0);
                            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(oa[a][1], vv1, dp, 0, 0, 0);
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int qq = 32 * u + 16 * a + 4 * g + r;
                                const float pv = key_dead ? 0.f : __builtin_amdgcn_exp2f(__builtin_fmaf(sv[r], sc2, -lq[a][r]));
                                // attention dropout: O = (P x mask) V, so dV takes P x mask and dP = (dO V^T) x mask; delta = <dO, O> is unchanged
                                const float mk = DROP ? dropout_scale(drop_seed, (((uint64_t)b * H + h) * T + (uint64_t)(qq < T ? qq : 0)) * T + (uint64_t)key, drop_p) : 1.f;
                                P[a][j][r] = pv * mk;
                                dS[a][j][r] = pv * (dp[r] * mk - dq_[a][r]);
                            }
                        }
                    }
                }
                // ---- dV^T += dO^T P ; dK^T += Q^T dS   (k = the 32 queries of this step)
                bf16x8 pP[2], pS[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) { pP[j] = att_pack8(P[0][j], P[1][j]); pS[j] = att_pack8(dS[0][j], dS[1][j]); }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const bf16x8 ot = att_frag_tr(Ot, 0, 16, dt, lane);
                    const bf16x8 qt = att_frag_tr(Qt, 0, 16, dt, lane);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (j < nkt) {
                            dVt[dt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ot, pP[j], dVt[dt][j], 0, 0, 0);
                            dKt[dt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, pS[j], dKt[dt][j], 0, 0, 0);
                        }
                    }
                }
            }
            // ---- dS -> this wave's sub-image [32 q][32 keys] (zeros for keys it does not own inside the K image)
            if (2 * wave < 2 * NT2) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            *reinterpret_cast<bf16_t*>(dSw + att_s_off(16 * a + 4 * g + r, 16 * j + lc)) = f2bf(dS[a][j][r]);
            }
        }
    };
    auto dq_phase = [&](int u) {
        char* dSs = QO + 32768 + (u & 1) * 16384;
        {   // ---- dQ tile [16 q (tile qa_) x 16 d (tile dt)] over all keys; lane ends up with 4 consecutive d of one query
            const int qa_ = wave & 1, dt = wave >> 1;
            f32x4 dq = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < (NTC ? (NTC + 1) / 2 : ATT_NTMAX / 2); ++ks) {
                if (!NTC && ks >= NT2) break;
                const int row = 16 * qa_ + lc;
                const bf16x8 dsf = *reinterpret_cast<const bf16x8*>(dSs + ks * 2048 + row * 64 + (((g ^ (row >> 2)) & 3) << 4));
                const bf16x8 kf = att_frag_tr_nat(Kt, 32 * ks, dt, lane);
                dq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, dsf, dq, 0, 0, 0);      // D[d = 4g + r][q = lc]
            }
            const int q = 32 * u + 16 * qa_ + lc;
            if (q < T)
                *reinterpret_cast<uint2*>(dqkv + ((int64_t)b * T + q) * 3 * E + h * ATT_D + 16 * dt + 4 * g) =
                    make_uint2(pack_bf2(dq[0] * scale, dq[1] * scale), pack_bf2(dq[2] * scale, dq[3] * scale));
            dqsum += dq;      // rows q >= T hold zeros (their P is masked to 0)
        }
    };
    if (stager) {
        stage_tile(0);      // tile 0 was requested before the K / V loads
        if (1 < NT2) fetch_tile(1);
        ATT_LDS_BARRIER();      // K / V images and query tile 0 are in LDS
        for (int u = 0; u < NT2; ++u) {
            if (u + 1 < NT2) {
                stage_tile((u & 1) ^ 1);      // tile u+1 -> the other buffer: its last readers finished before the previous barrier
                if (u + 2 < NT2) fetch_tile(u + 2);
            }
            ATT_LDS_BARRIER();
            dq_phase(u);
        }
    } else {
        ATT_LDS_BARRIER();
        for (int u = 0; u < NT2; ++u) {
            main_phase(u);
            ATT_LDS_BARRIER();   // every wave's dS of this step and the next step's query tiles are in LDS
            dq_phase(u);
        }
    }
    // ---- dK, dV of this wave's keys: lane owns key 16(2w+j)+lc, d = 16dt + 4g + 0..3 — 8 bytes of a 128-byte row.  Stored directly
    // that is 16 instructions of sixteen 32-byte row pieces each (stamps: 1.8 us to issue them, and the block's tail waited on their
    // drain); instead the wave turns its [32 keys][64 d] blocks around in LDS — the K and V row images of ITS OWN keys, which
    // nobody reads after the last step's barrier C — and stores whole 128-byte rows, 8 rows per instruction.
    if (nkt > 0) {
        char* tk = Kk + wave * 4096;
        char* tv = Vk + wave * 4096;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = 16 * j + lc;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int off = row * 128 + ((((2 * dt + (g >> 1)) ^ (row & 7))) << 4) + ((g & 1) << 3);
                *reinterpret_cast<uint2*>(tk + off) = make_uint2(pack_bf2(dKt[dt][j][0] * scale, dKt[dt][j][1] * scale),
                                                                 pack_bf2(dKt[dt][j][2] * scale, dKt[dt][j][3] * scale));
                *reinterpret_cast<uint2*>(tv + off) = make_uint2(pack_bf2(dVt[dt][j][0], dVt[dt][j][1]), pack_bf2(dVt[dt][j][2], dVt[dt][j][3]));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // wave-local: no barrier
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 8 * i + (lane >> 3), cc = lane & 7;
            const int key = 32 * wave + row;
            const uint4 kq = *reinterpret_cast<const uint4*>(tk + row * 128 + ((cc ^ (row & 7)) << 4));
            const uint4 vq = *reinterpret_cast<const uint4*>(tv + row * 128 + ((cc ^ (row & 7)) << 4));
            if (key < T) {
                bf16_t* dst = dqkv + ((int64_t)b * T + key) * 3 * E + h * ATT_D + 8 * cc;
                *reinterpret_cast<uint4*>(dst + E) = kq;
                *reinterpret_cast<uint4*>(dst + 2 * E) = vq;
            }
        }
    }
    if (bias_part == nullptr) return;
    // ---- column sums of this (utterance, head)'s dQ / dK / dV block -> bias_part[b][3E]: the q/k/v bias gradient is the column sum
    // of dqkv (autograd of F.linear), summed here from the f32 accumulators instead of by a pass over the 78 MB tensor.
    // Fixed order: butterfly over the 16 key / query lanes, then waves in index order => deterministic.
    // [8 waves][16 (dq) + 64 (dk) + 64 (dv)] f32 = 4.5 KiB in the query-tile buffer the LAST step did not use: its readers (main phase of
    // step NT2 - 2) finished before that step's barrier, and the last step staged nothing into it — no barrier needed before the partials
    // are written (round 6; rounds 3 - 5 waited for the dS image here: one block-wide barrier of the tail gone)
    float* red = reinterpret_cast<float*>(QO + (((NT2 - 1) & 1) ^ 1) * 16384);
    {
        float v[4] = {dqsum[0] * scale, dqsum[1] * scale, dqsum[2] * scale, dqsum[3] * scale};
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = lanes16_sum(v[r]);
        if (lc == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave * 144 + 4 * g + r] = v[r];
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            float kk[4], vv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {      // keys >= T and tiles this wave does not own hold zeros
                kk[r] = (dKt[dt][0][r] + dKt[dt][1][r]) * scale;
                vv[r] = dVt[dt][0][r] + dVt[dt][1][r];
                kk[r] = lanes16_sum(kk[r]);
                vv[r] = lanes16_sum(vv[r]);
            }
            if (lc == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { red[wave * 144 + 16 + 16 * dt + 4 * g + r] = kk[r]; red[wave * 144 + 80 + 16 * dt + 4 * g + r] = vv[r]; }
            }
        }
    }
    ATT_LDS_BARRIER();
    if (threadIdx.x < 192) {
        const int which = threadIdx.x >> 6, d = threadIdx.x & 63;      // 0: q, 1: k, 2: v
        float t = 0.f;
        if (which == 0) t = red[(2 * (d >> 4)) * 144 + (d & 15)] + red[(2 * (d >> 4) + 1) * 144 + (d & 15)];      // the two query tiles of d-tile d >> 4
        else {
#pragma unroll
            for (int w = 0; w < 8; ++w) t += red[w * 144 + 16 + 64 * (which - 1) + d];
        }
        bias_part[(int64_t)b * 3 * E + which * E + h * ATT_D + d] = t;
    }
}

}  // namespace

extern "C" int scl_attn_fwd(const void* qkv, void* ctx, float* lse, int B, int T, int H, int D, float scale, float drop_p, uint32_t drop_seed,
                            void* stream) {
    SCL_REQUIRE(qkv && ctx && lse && B > 0 && H > 0 && drop_p >= 0.f && drop_p < 1.f, "attn_fwd: bad args");
    SCL_REQUIRE(D == ATT_D && T >= 1 && T <= 256, "attn_fwd: fused path needs head dim 64 and T <= 256 (got D=%d, T=%d)", D, T);
    const int NT = (T + 15) / 16, rows = 32 * ((NT + 1) / 2);
    const size_t lds = (size_t)2 * rows * 128;
#define ATT_FWD(DR, N) hipLaunchKernelGGL((attn_fwd_kernel<DR, N>), dim3(B * H), dim3(512), lds, (hipStream_t)stream, (const bf16_t*)qkv, (bf16_t*)ctx, lse, T, H, scale, drop_p, drop_seed)
    if (drop_p > 0.f) ATT_FWD(true, 0);
    else if (NT == 13) ATT_FWD(false, 13);
    else if (NT == 4) ATT_FWD(false, 4);
    else ATT_FWD(false, 0);
#undef ATT_FWD
    return scl_check_launch("scl_attn_fwd");
}

extern "C" int scl_attn_bwd(const void* qkv, const void* ctx, const void* dctx, const float* lse, void* dqkv, float* bias_part, int B, int T,
                            int H, int D, float scale, float drop_p, uint32_t drop_seed, void* stream) {
    SCL_REQUIRE(qkv && ctx && dctx && lse && dqkv && B > 0 && H > 0 && drop_p >= 0.f && drop_p < 1.f, "attn_bwd: bad args");
    SCL_REQUIRE(D == ATT_D && T >= 1 && T <= 224, "attn_bwd: fused path needs head dim 64 and T <= 224 (got D=%d, T=%d)", D, T);
    const int NT = (T + 15) / 16, rows = 32 * ((NT + 1) / 2);
    const size_t lds = (size_t)3 * rows * 128 + 2 * 16384 + 2 * 16384 + 128 * 4;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)attn_bwd8_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)attn_bwd8_kernel<false, 13>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)attn_bwd8_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
#define ATT_BWD(DR, N) hipLaunchKernelGGL((attn_bwd8_kernel<DR, N>), dim3(B * H), dim3(512), lds, (hipStream_t)stream, (const bf16_t*)qkv, (const bf16_t*)ctx, \
                                          (const bf16_t*)dctx, lse, (bf16_t*)dqkv, bias_part, T, H, scale, drop_p, drop_seed)
    if (drop_p > 0.f) ATT_BWD(true, 0);
    else if (NT == 13) ATT_BWD(false, 13);
    else ATT_BWD(false, 0);
#undef ATT_BWD
    return scl_check_launch("scl_attn_bwd");
}
