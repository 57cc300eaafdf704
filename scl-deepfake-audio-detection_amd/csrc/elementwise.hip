// elementwise.hip — HBM-bound glue of the hot path: dtype casts, zero-padded row copies for the
// positional conv, col2im of the conv-stack dgrad, weight re-layouts (conv [co,ci,j] <-> GEMM
// [co, j*C+ci]; weight-norm of encoder.pos_conv.0), and the tiny tail of the linear head
// (model/wav2vec2_linear_nll.py:88-93,134: mean over frames, m_utt_level, log_softmax).
#include "common.h"

namespace {

__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n) {
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    const int64_t step = (int64_t)gridDim.x * blockDim.x * 8;
    for (int64_t i = i0; i < n; i += step) {
        if (i + 8 <= n) {
            const float4 a = *reinterpret_cast<const float4*>(src + i);
            const float4 b = *reinterpret_cast<const float4*>(src + i + 4);
            uint4 u;
            u.x = pack_bf2(a.x, a.y); u.y = pack_bf2(a.z, a.w); u.z = pack_bf2(b.x, b.y); u.w = pack_bf2(b.z, b.w);
            *reinterpret_cast<uint4*>(dst + i) = u;
        } else {
            for (int64_t j = i; j < n; ++j) dst[j] = f2bf(src[j]);
        }
    }
}

// f32 [rows][K] (row pitch ldx) -> bf16 [rows][3 K]: every value as hi = bf16(x) and lo = bf16(x - hi), laid out per row as
// [hi | hi | lo] (order 0: the left operand) or [hi | lo | hi] (order 1: the right operand), so that ONE bf16 GEMM over 3 K computes
// hi.hi + hi.lo + lo.hi — the f32 product to ~2^-17 relative (the lo.lo term is dropped), accumulated in f32.  Round 6: the scoring
// path's linears on the wide bf16 kernel (operands by LDS-DMA, no conversion in the K loop) instead of the f32-pair kernel that splits
// inside its loop at 26 % matrix-pipe occupancy.
__global__ void split3_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, int64_t rows, int K, int64_t ldx, int order) {
    const int kv = K >> 3;
    const int64_t total = rows * kv;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
        const int64_t r = i / kv;
        const int c = (int)(i - r * kv) * 8;
        const float4 a = *reinterpret_cast<const float4*>(x + r * ldx + c);
        const float4 b = *reinterpret_cast<const float4*>(x + r * ldx + c + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        float lo[8];
        uint4 h, l;
        unsigned hw[4], lw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            hw[j] = pack_bf2(v[2 * j], v[2 * j + 1]);
            lo[2 * j] = v[2 * j] - __uint_as_float(hw[j] << 16);
            lo[2 * j + 1] = v[2 * j + 1] - __uint_as_float(hw[j] & 0xFFFF0000u);
            lw[j] = pack_bf2(lo[2 * j], lo[2 * j + 1]);
        }
        h = make_uint4(hw[0], hw[1], hw[2], hw[3]); l = make_uint4(lw[0], lw[1], lw[2], lw[3]);
        bf16_t* o = out + r * 3 * (int64_t)K + c;
        *reinterpret_cast<uint4*>(o) = h;
        *reinterpret_cast<uint4*>(o + K) = order ? l : h;
        *reinterpret_cast<uint4*>(o + 2 * (int64_t)K) = order ? h : l;
    }
}

// dst[b][r][c] (bf16, rows_out per item) = src[b][r - pad_before][c] (f32 or bf16), zero outside; optional
// multiply by act'(pre[b][r-pad][c]) (positional-conv backward: dc = d_out * gelu'(pre)).
template <bool SRC_F32>
__global__ void pad_rows_kernel(const void* __restrict__ src, bf16_t* __restrict__ dst, const bf16_t* __restrict__ pre,
                                int B, int T, int C, int rows_out, int pad_before, int ract) {
    const int64_t total = (int64_t)B * rows_out * (C / 8);
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    const int c8n = C / 8;
    for (int64_t i = i0; i < total; i += step) {
        const int c = (int)(i % c8n) * 8;
        const int64_t br = i / c8n;
        const int r = (int)(br % rows_out), b = (int)(br / rows_out);
        const int t = r - pad_before;
        uint4 u = make_uint4(0, 0, 0, 0);
        if (t >= 0 && t < T) {
            const int64_t so = ((int64_t)b * T + t) * C + c;
            float v[8];
            if (SRC_F32) {
                const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(src) + so);
                const float4 bb = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(src) + so + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = bb.x; v[5] = bb.y; v[6] = bb.z; v[7] = bb.w;
            } else {
                const uint4 w = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(src) + so);
                const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[2 * k] = __uint_as_float(ww[k] << 16); v[2 * k + 1] = __uint_as_float(ww[k] & 0xFFFF0000u); }
            }
            if (pre) {
                const uint4 w = *reinterpret_cast<const uint4*>(pre + so);
                const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v[2 * k] *= act_grad_f(ract, __uint_as_float(ww[k] << 16));
                    v[2 * k + 1] *= act_grad_f(ract, __uint_as_float(ww[k] & 0xFFFF0000u));
                }
            }
            u.x = pack_bf2(v[0], v[1]); u.y = pack_bf2(v[2], v[3]); u.z = pack_bf2(v[4], v[5]); u.w = pack_bf2(v[6], v[7]);
        }
        *reinterpret_cast<uint4*>(dst + ((int64_t)b * rows_out + r) * C + c) = u;
    }
}

// conv-stack dgrad, second half: dz[b][r][c] = sum_j dcol[b][(r-j)/s][j*C + c] over taps j with (r-j) % s == 0
__global__ void col2im_kernel(const bf16_t* __restrict__ dcol, bf16_t* __restrict__ dz, int B, int Tin, int Tout, int C,
                              int k, int s) {
    const int c8n = C / 8;
    const int64_t total = (int64_t)B * Tin * c8n;
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = i0; i < total; i += step) {
        const int c = (int)(i % c8n) * 8;
        const int64_t br = i / c8n;
        const int r = (int)(br % Tin), b = (int)(br / Tin);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int j = 0; j < k; ++j) {
            const int rj = r - j;
            if (rj < 0 || (rj % s) != 0) continue;
            const int t = rj / s;
            if (t >= Tout) continue;
            const uint4 w = *reinterpret_cast<const uint4*>(dcol + (((int64_t)b * Tout + t) * k + j) * C + c);
            const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) { acc[2 * q] += __uint_as_float(ww[q] << 16); acc[2 * q + 1] += __uint_as_float(ww[q] & 0xFFFF0000u); }
        }
        uint4 u;
        u.x = pack_bf2(acc[0], acc[1]); u.y = pack_bf2(acc[2], acc[3]); u.z = pack_bf2(acc[4], acc[5]); u.w = pack_bf2(acc[6], acc[7]);
        *reinterpret_cast<uint4*>(dz + ((int64_t)b * Tin + r) * C + c) = u;
    }
}

// conv weight [co][ci][j] f32  ->  GEMM layout [co][j*Ci + ci] bf16, and (optional) the transposed-convolution layout used by the
// phase-split dgrad: wd[blk(j)][co][ci], tap blocks ordered phase by phase (p = j mod stride), inside a phase by DESCENDING tap
// (the A rows of that GEMM are [dy[u-q_max] ... dy[u]], so block q' holds tap p + stride*(nq-1-q')).
__global__ void conv_w_pack_kernel(const float* __restrict__ w, bf16_t* __restrict__ wk, bf16_t* __restrict__ wd, int Co, int Ci, int k,
                                   int stride) {
    const int64_t n = (int64_t)Co * Ci * k;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Ci);
        const int j = (int)((i / Ci) % k);
        const int co = (int)(i / ((int64_t)Ci * k));
        const bf16_t v = f2bf(w[((int64_t)co * Ci + ci) * k + j]);
        wk[i] = v;
        if (wd) {
            const int p = j % stride, q = j / stride;
            int blk = 0;
            for (int pp = 0; pp < p; ++pp) blk += (k - pp + stride - 1) / stride;     // taps in the earlier phases
            const int nq = (k - p + stride - 1) / stride;
            blk += nq - 1 - q;
            wd[((int64_t)blk * Co + co) * Ci + ci] = v;
        }
    }
}
// gradient back: dwk [co][j*Ci+ci] f32 -> dw [co][ci][j] f32
__global__ void conv_w_unpack_grad_kernel(const float* __restrict__ dwk, float* __restrict__ dw, int Co, int Ci, int k) {
    const int64_t n = (int64_t)Co * Ci * k;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % k);
        const int ci = (int)((i / k) % Ci);
        const int co = (int)(i / ((int64_t)Ci * k));
        dw[i] = dwk[((int64_t)co * k + j) * Ci + ci];
    }
}

// ---- positional conv weight: torch weight_norm(dim=2): w[co][ci][j] = g[j] * v[co][ci][j] / ||v[:,:,j]|| ----
// Two coalesced stages (a first version ran one block per tap over a stride-K gather: 16x read amplification, 150 us).
// Stage 1: block b sums v^2 over its slab of (co, ci) rows, thread t owns tap j = t % K: rows are read as they lie (K contiguous
// floats).  part[b][j]; blockDim must be a multiple of K.
__global__ __launch_bounds__(256) void posconv_norm_part_kernel(const float* __restrict__ v, float* __restrict__ part, int nrows, int K,
                                                                int rows_per_block) {
    __shared__ float red[256];
    const int j = threadIdx.x % K, rl = threadIdx.x / K, rstep = blockDim.x / K;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(nrows, r0 + rows_per_block);
    float s = 0.f;
    for (int r = r0 + rl; r < r1; r += rstep) { const float t = v[(int64_t)r * K + j]; s += t * t; }
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < K) {
        float t = 0.f;
        for (int q = 0; q < rstep; ++q) t += red[q * K + threadIdx.x];
        part[(int64_t)blockIdx.x * K + threadIdx.x] = t;
    }
}
// Stage 2 (one block of 1024 threads): out[j] = f(sum_p part[p][j]), f = sqrt or identity; fixed order => deterministic
__global__ __launch_bounds__(1024) void posconv_vec_finish_kernel(const float* __restrict__ part, float* __restrict__ out, int nparts, int K,
                                                                   int do_sqrt) {
    __shared__ float red[1024];
    const int j = threadIdx.x % K, seg = threadIdx.x / K, nseg = blockDim.x / K;
    float s = 0.f;
    for (int p = seg; p < nparts; p += nseg) s += part[(int64_t)p * K + j];
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < K) {
        float t = 0.f;
        for (int q = 0; q < nseg; ++q) t += red[q * K + threadIdx.x];
        out[threadIdx.x] = do_sqrt ? sqrtf(t) : t;
    }
}
// wf[g][co][j*Cg + ci] (forward GEMM B operand) and wd[g][ci][j'*Cg + co] with j' = K-1-j (dgrad operand)
__global__ void posconv_pack_kernel(const float* __restrict__ v, const float* __restrict__ g, const float* __restrict__ norm,
                                    bf16_t* __restrict__ wf, bf16_t* __restrict__ wd, int E, int Cg, int K) {
    const int64_t n = (int64_t)E * Cg * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % K);
        const int ci = (int)((i / K) % Cg);
        const int cot = (int)(i / ((int64_t)Cg * K));  // global output channel
        const int grp = cot / Cg, co = cot % Cg;
        const bf16_t val = f2bf(g[j] * v[i] / norm[j]);
        wf[(((int64_t)grp * Cg + co) * K + j) * Cg + ci] = val;
        wd[(((int64_t)grp * Cg + ci) * K + (K - 1 - j)) * Cg + co] = val;
    }
}
// backward of the weight norm from dwf[g][co][j*Cg+ci] (f32):
//   s_j = sum_{co,ci} dw * v ;  dg[j] = s_j / norm_j ;  dv = g_j/norm_j * dw - g_j * s_j / norm_j^3 * v
// Stage 1 of s_j: one block per output channel `cot`.  Its v block [Cg ci][K j] and its dwf block [K j][Cg ci] are both
// contiguous: staged in LDS as they lie (dwf rows padded by one float), then thread (j, half) walks ci.  part[cot][j].
__global__ __launch_bounds__(256) void posconv_wbwd_dot_part_kernel(const float* __restrict__ dwf, const float* __restrict__ v,
                                                                    float* __restrict__ part, int Cg, int K) {
    extern __shared__ float sm[];
    float* vl = sm;                     // [Cg][K]
    float* dl = sm + Cg * K;            // [K][Cg + 1]
    float* red = dl + K * (Cg + 1);     // [blockDim]
    const int cot = blockIdx.x;
    const float* vb = v + (int64_t)cot * Cg * K;
    const float* db = dwf + (int64_t)cot * K * Cg;
    for (int i = threadIdx.x; i < Cg * K; i += blockDim.x) {
        vl[i] = vb[i];
        dl[(i / Cg) * (Cg + 1) + (i % Cg)] = db[i];
    }
    __syncthreads();
    const int j = threadIdx.x % K, h = threadIdx.x / K, nh = blockDim.x / K;
    float s = 0.f;
    for (int ci = h; ci < Cg; ci += nh) s += dl[j * (Cg + 1) + ci] * vl[ci * K + j];
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < K) {
        float t = 0.f;
        for (int q = 0; q < nh; ++q) t += red[q * K + threadIdx.x];
        part[(int64_t)cot * K + threadIdx.x] = t;
    }
}
__global__ void posconv_wbwd_apply_kernel(const float* __restrict__ dwf, const float* __restrict__ v, const float* __restrict__ g,
                                          const float* __restrict__ norm, const float* __restrict__ sdot,
                                          float* __restrict__ dv, float* __restrict__ dg, int E, int Cg, int K) {
    const int64_t n = (int64_t)E * Cg * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % K);
        const int ci = (int)((i / K) % Cg);
        const int cot = (int)(i / ((int64_t)Cg * K));
        const int grp = cot / Cg, co = cot % Cg;
        const float dw = dwf[(((int64_t)grp * Cg + co) * K + j) * Cg + ci];
        const float nj = norm[j];
        dv[i] = g[j] / nj * dw - g[j] * sdot[j] / (nj * nj * nj) * v[i];
        if (i < K) dg[i] = sdot[i] / norm[i];
    }
}

// ---- head tail --------------------------------------------------------------------------------
// emb[b][c] = mean_t h[b][t][c]   (h bf16 [B,T,C])
// 512 threads = 4 frame groups x 128 channel lanes: group g adds frames g, g + 4, ...; the four partial sums are combined in a
// fixed order through LDS (one thread per channel walking all T frames left the launch at 49 us for 64 utterances)
__device__ __forceinline__ float ld_act(const bf16_t* p, int64_t i) { return bf2f(p[i]); }
__device__ __forceinline__ float ld_act(const float* p, int64_t i) { return p[i]; }
__device__ __forceinline__ void st_act(bf16_t* p, int64_t i, float v) { p[i] = f2bf(v); }
__device__ __forceinline__ void st_act(float* p, int64_t i, float v) { p[i] = v; }
template <typename TA>
__global__ __launch_bounds__(512) void meanpool_fwd_kernel(const TA* __restrict__ h, float* __restrict__ emb, int T, int C) {
    __shared__ float red[4][128];
    const int b = blockIdx.x, g = threadIdx.x >> 7, l = threadIdx.x & 127;
    for (int c0 = 0; c0 < C; c0 += 128) {
        const int c = c0 + l;
        float s = 0.f;
        if (c < C)
            for (int t = g; t < T; t += 4) s += ld_act(h, ((int64_t)b * T + t) * C + c);
        red[g][l] = s;
        __syncthreads();
        if (g == 0 && c < C) emb[(int64_t)b * C + c] = (((red[0][l] + red[1][l]) + red[2][l]) + red[3][l]) / (float)T;
        __syncthreads();
    }
}
// d_pre[b][t][c] = d_emb[b][c] / T * dropmask(seed, idx) * act'(pre[b][t][c])
template <typename TA>
__global__ void meanpool_bwd_kernel(const float* __restrict__ demb, const TA* __restrict__ pre, TA* __restrict__ dpre,
                                    int B, int T, int C, int ract, float drop_p, uint32_t seed) {
    const int64_t n = (int64_t)B * T * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int b = (int)(i / ((int64_t)T * C));
        float v = demb[(int64_t)b * C + c] / (float)T;
        if (drop_p > 0.f) v *= dropout_scale(seed, (uint64_t)i, drop_p);
        v *= act_grad_f(ract, ld_act(pre, i));
        st_act(dpre, i, v);
    }
}
// logits = emb W^T + b ; logp = log_softmax(logits)   (num classes NC <= 8, C <= 1024); one wave per item
__global__ void utt_head_fwd_kernel(const float* __restrict__ emb, const float* __restrict__ W, const float* __restrict__ bias,
                                    float* __restrict__ logp, int B, int C, int NC) {
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= B) return;
    float lg[8];
    float mx = -INFINITY;
    for (int k = 0; k < NC; ++k) {
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += emb[(int64_t)b * C + c] * W[(int64_t)k * C + c];
        s = wave_sum(s) + bias[k];
        lg[k] = s;
        mx = fmaxf(mx, s);
    }
    float se = 0.f;
    for (int k = 0; k < NC; ++k) se += expf(lg[k] - mx);
    const float lse = mx + logf(se);
    if (lane == 0) for (int k = 0; k < NC; ++k) logp[(int64_t)b * NC + k] = lg[k] - lse;
}
// backward of log_softmax + linear: dlogits = dlogp - softmax * sum(dlogp); demb (+)= dlogits W;
// dW[k][c] = sum_b dlogits[b][k] emb[b][c]; db[k] = sum_b dlogits[b][k].   Single block (B, C small).
__global__ void utt_head_bwd_kernel(const float* __restrict__ dlogp, const float* __restrict__ logp, const float* __restrict__ emb,
                                    const float* __restrict__ W, const float* __restrict__ demb_in, float* __restrict__ demb,
                                    float* __restrict__ dW, float* __restrict__ db, float* __restrict__ dlogits_ws, int B,
                                    int C, int NC) {
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < NC; ++k) s += dlogp[i * NC + k];
        for (int k = 0; k < NC; ++k) dlogits_ws[i * NC + k] = dlogp[i * NC + k] - expf(logp[i * NC + k]) * s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < B * C; i += blockDim.x) {
        const int b = i / C, c = i % C;
        float s = demb_in ? demb_in[i] : 0.f;
        for (int k = 0; k < NC; ++k) s += dlogits_ws[b * NC + k] * W[(int64_t)k * C + c];
        demb[i] = s;
    }
    for (int i = threadIdx.x; i < NC * C; i += blockDim.x) {
        const int k = i / C, c = i % C;
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dlogits_ws[b * NC + k] * emb[(int64_t)b * C + c];
        dW[i] = s;
    }
    for (int k = threadIdx.x; k < NC; k += blockDim.x) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dlogits_ws[b * NC + k];
        db[k] = s;
    }
}

// out = a + b (f32), optional bf16 copy
__global__ void add_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                               bf16_t* __restrict__ out_bf, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = a[i] + (b ? b[i] : 0.f);
        if (out) out[i] = v;
        if (out_bf) out_bf[i] = f2bf(v);
    }
}

inline int grid_for(int64_t n, int per_thread = 1) {
    int64_t b = (n / per_thread + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" int scl_cast_f32_bf16(const float* src, void* dst, int64_t n, void* stream) {
    SCL_REQUIRE(src && dst && n > 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "cast: bad args");
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n, 8)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n);
    return scl_check_launch("scl_cast_f32_bf16");
}

extern "C" int scl_split3_f32_bf16(const float* x, int64_t rows, int K, int64_t ldx, void* out, int order, void* stream) {
    SCL_REQUIRE(x && out && rows > 0 && K >= 8 && (K & 7) == 0 && ldx >= K && (ldx & 3) == 0 && (order == 0 || order == 1), "split3: bad args (K a multiple of 8)");
    SCL_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 15) == 0, "split3: operands must be 16-byte aligned");
    hipLaunchKernelGGL(split3_kernel, dim3(grid_for(rows * (K >> 3))), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)out, rows, K, ldx, order);
    return scl_check_launch("scl_split3_f32_bf16");
}

extern "C" int scl_pad_rows_bf16(const void* src, int src_f32, void* dst, const void* pre, int ract, int B, int T, int C,
                                 int rows_out, int pad_before, void* stream) {
    SCL_REQUIRE(src && dst && B > 0 && T > 0 && C > 0 && (C & 7) == 0 && rows_out >= 1, "pad_rows: bad args");
    const int64_t total = (int64_t)B * rows_out * (C / 8);
    hipStream_t s = (hipStream_t)stream;
    if (src_f32) hipLaunchKernelGGL((pad_rows_kernel<true>), dim3(grid_for(total)), dim3(256), 0, s, src, (bf16_t*)dst, (const bf16_t*)pre, B, T, C, rows_out, pad_before, ract);
    else hipLaunchKernelGGL((pad_rows_kernel<false>), dim3(grid_for(total)), dim3(256), 0, s, src, (bf16_t*)dst, (const bf16_t*)pre, B, T, C, rows_out, pad_before, ract);
    return scl_check_launch("scl_pad_rows_bf16");
}

extern "C" int scl_col2im_bf16(const void* dcol, void* dz, int B, int Tin, int Tout, int C, int k, int s, void* stream) {
    SCL_REQUIRE(dcol && dz && B > 0 && Tin > 0 && Tout > 0 && (C & 7) == 0 && k >= 1 && s >= 1, "col2im: bad args");
    const int64_t total = (int64_t)B * Tin * (C / 8);
    hipLaunchKernelGGL(col2im_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dcol, (bf16_t*)dz, B, Tin, Tout, C, k, s);
    return scl_check_launch("scl_col2im_bf16");
}

extern "C" int scl_conv_weight_pack(const float* w, void* wk, void* wd, int Co, int Ci, int k, int stride, void* stream) {
    SCL_REQUIRE(w && wk && Co > 0 && Ci > 0 && k > 0 && stride >= 1, "conv_weight_pack: bad args");
    hipLaunchKernelGGL(conv_w_pack_kernel, dim3(grid_for((int64_t)Co * Ci * k)), dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)wk,
                       (bf16_t*)wd, Co, Ci, k, stride);
    return scl_check_launch("scl_conv_weight_pack");
}
extern "C" int scl_conv_weight_unpack_grad(const float* dwk, float* dw, int Co, int Ci, int k, void* stream) {
    SCL_REQUIRE(dwk && dw && Co > 0 && Ci > 0 && k > 0, "conv_weight_unpack_grad: bad args");
    hipLaunchKernelGGL(conv_w_unpack_grad_kernel, dim3(grid_for((int64_t)Co * Ci * k)), dim3(256), 0, (hipStream_t)stream, dwk, dw, Co, Ci, k);
    return scl_check_launch("scl_conv_weight_unpack_grad");
}

extern "C" int scl_posconv_weight_pack(const float* v, const float* g, float* norm, void* wf, void* wd, int E, int Cg, int K, void* stream) {
    SCL_REQUIRE(v && g && norm && wf && wd && E > 0 && Cg > 0 && K > 0 && E % Cg == 0, "posconv_weight_pack: bad args");
    hipStream_t s = (hipStream_t)stream;
    {   // ||v[:,:,j]|| in two coalesced stages; `wd` doubles as the [<= 256][K] f32 partial buffer before it is written below
        SCL_REQUIRE(K <= 256 && 256 % K == 0 && E * Cg >= 2, "posconv_weight_pack: need K | 256");
        const int nrows = E * Cg;
        int nb = nrows / 2 < 256 ? nrows / 2 : 256;          // nb * K floats must fit into wd's E*Cg*K bf16
        const int rpb = (nrows + nb - 1) / nb;
        float* part = reinterpret_cast<float*>(wd);
        hipLaunchKernelGGL(posconv_norm_part_kernel, dim3((nrows + rpb - 1) / rpb), dim3(256), 0, s, v, part, nrows, K, rpb);
        hipLaunchKernelGGL(posconv_vec_finish_kernel, dim3(1), dim3(1024 / K * K), 0, s, part, norm, (nrows + rpb - 1) / rpb, K, 1);
    }
    hipLaunchKernelGGL(posconv_pack_kernel, dim3(grid_for((int64_t)E * Cg * K)), dim3(256), 0, s, v, g, norm, (bf16_t*)wf, (bf16_t*)wd, E, Cg, K);
    return scl_check_launch("scl_posconv_weight_pack");
}
extern "C" int scl_posconv_weight_bwd(const float* dwf, const float* v, const float* g, const float* norm, float* sdot_ws,
                                      float* dv, float* dg, int E, int Cg, int K, void* stream) {
    SCL_REQUIRE(dwf && v && g && norm && sdot_ws && dv && dg, "posconv_weight_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    {   // s_j = sum dW * v: per-output-channel partials (sdot_ws: K + E*K floats), then one finishing block
        SCL_REQUIRE(K <= 256 && 256 % K == 0, "posconv_weight_bwd: need K | 256");
        float* part = sdot_ws + K;
        const size_t lds = (size_t)(Cg * K + K * (Cg + 1) + 256) * sizeof(float);
        static bool attr_set = false;
        if (!attr_set) {
            hipFuncSetAttribute((const void*)posconv_wbwd_dot_part_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set = true;
        }
        SCL_REQUIRE(lds <= 160 * 1024, "posconv_weight_bwd: Cg*K too large for the LDS tile");
        hipLaunchKernelGGL(posconv_wbwd_dot_part_kernel, dim3(E), dim3(256), lds, s, dwf, v, part, Cg, K);
        hipLaunchKernelGGL(posconv_vec_finish_kernel, dim3(1), dim3(1024 / K * K), 0, s, part, sdot_ws, E, K, 0);
    }
    hipLaunchKernelGGL(posconv_wbwd_apply_kernel, dim3(grid_for((int64_t)E * Cg * K)), dim3(256), 0, s, dwf, v, g, norm, sdot_ws, dv, dg, E, Cg, K);
    return scl_check_launch("scl_posconv_weight_bwd");
}

extern "C" int scl_meanpool_fwd(const void* h, float* emb, int B, int T, int C, void* stream) {
    SCL_REQUIRE(h && emb && B > 0 && T > 0 && C > 0, "meanpool_fwd: bad args");
    hipLaunchKernelGGL(meanpool_fwd_kernel<bf16_t>, dim3(B), dim3(512), 0, (hipStream_t)stream, (const bf16_t*)h, emb, T, C);
    return scl_check_launch("scl_meanpool_fwd");
}
extern "C" int scl_meanpool_fwd_f32(const float* h, float* emb, int B, int T, int C, void* stream) {
    SCL_REQUIRE(h && emb && B > 0 && T > 0 && C > 0, "meanpool_fwd_f32: bad args");
    hipLaunchKernelGGL(meanpool_fwd_kernel<float>, dim3(B), dim3(512), 0, (hipStream_t)stream, h, emb, T, C);
    return scl_check_launch("scl_meanpool_fwd_f32");
}
extern "C" int scl_meanpool_bwd(const float* demb, const void* pre, void* dpre, int B, int T, int C, int ract, float drop_p,
                                uint32_t seed, void* stream) {
    SCL_REQUIRE(demb && pre && dpre && B > 0 && T > 0 && C > 0, "meanpool_bwd: bad args");
    hipLaunchKernelGGL(meanpool_bwd_kernel<bf16_t>, dim3(grid_for((int64_t)B * T * C)), dim3(256), 0, (hipStream_t)stream, demb, (const bf16_t*)pre, (bf16_t*)dpre, B, T, C, ract, drop_p, seed);
    return scl_check_launch("scl_meanpool_bwd");
}
extern "C" int scl_meanpool_bwd_f32(const float* demb, const float* pre, float* dpre, int B, int T, int C, int ract, float drop_p,
                                    uint32_t seed, void* stream) {
    SCL_REQUIRE(demb && pre && dpre && B > 0 && T > 0 && C > 0, "meanpool_bwd_f32: bad args");
    hipLaunchKernelGGL(meanpool_bwd_kernel<float>, dim3(grid_for((int64_t)B * T * C)), dim3(256), 0, (hipStream_t)stream, demb, pre, dpre, B, T, C, ract, drop_p, seed);
    return scl_check_launch("scl_meanpool_bwd_f32");
}
extern "C" int scl_utt_head_fwd(const float* emb, const float* W, const float* bias, float* logp, int B, int C, int NC, void* stream) {
    SCL_REQUIRE(emb && W && bias && logp && B > 0 && C > 0 && NC >= 1 && NC <= 8, "utt_head_fwd: bad args");
    hipLaunchKernelGGL(utt_head_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, emb, W, bias, logp, B, C, NC);
    return scl_check_launch("scl_utt_head_fwd");
}
extern "C" int scl_utt_head_bwd(const float* dlogp, const float* logp, const float* emb, const float* W, const float* demb_in,
                                float* demb, float* dW, float* db, float* ws, int B, int C, int NC, void* stream) {
    SCL_REQUIRE(dlogp && logp && emb && W && demb && dW && db && ws && B > 0 && NC <= 8, "utt_head_bwd: bad args");
    hipLaunchKernelGGL(utt_head_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, dlogp, logp, emb, W, demb_in, demb, dW, db, ws, B, C, NC);
    return scl_check_launch("scl_utt_head_bwd");
}
// y[i] = x[i] * keep(seed, i) / (1 - p): the element-dropout sites of the encoder that no GEMM epilogue covers (fairseq
// TransformerEncoder.extract_features: F.dropout after the positional-conv residual add; its backward; the backward of dropout_input)
__global__ void dropout_f32_kernel(const float* __restrict__ x, float* __restrict__ y, bf16_t* __restrict__ y_bf, int64_t n, uint32_t seed, float p) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[i] * dropout_scale(seed, (uint64_t)i, p);
        if (y) y[i] = v;
        if (y_bf) y_bf[i] = f2bf(v);
    }
}
extern "C" int scl_dropout_f32(const float* x, float* y_f32, void* y_bf16, int64_t n, uint32_t seed, float p, void* stream) {
    SCL_REQUIRE(x && (y_f32 || y_bf16) && n > 0 && p >= 0.f && p < 1.f, "dropout_f32: bad args");
    hipLaunchKernelGGL(dropout_f32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y_f32, (bf16_t*)y_bf16, n, seed, p);
    return scl_check_launch("scl_dropout_f32");
}
// attention dropout on the UN-fused attention path (T > 224 or head dim != 64): the keep-mask of probability row r = (b, h, q), key j is
// hash(seed, r * T + j) — the index the fused kernels of attention.hip use — so both paths drop the same elements for one seed.
template <typename TX>
__global__ void dropout_rows_kernel(const TX* __restrict__ x, TX* __restrict__ y, int64_t R, int T, int ld, uint32_t seed, float p) {
    const int64_t n = R * ld;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / ld;
        const int j = (int)(i - r * ld);
        float v = 0.f;
        if (j < T) {
            if constexpr (sizeof(TX) == 2) v = bf2f(x[i]); else v = x[i];
            v *= dropout_scale(seed, (uint64_t)(r * T + j), p);
        }
        if constexpr (sizeof(TX) == 2) y[i] = f2bf(v); else y[i] = v;
    }
}
extern "C" int scl_dropout_rows(const void* x, void* y, int64_t R, int T, int ld, int is_f32, uint32_t seed, float p, void* stream) {
    SCL_REQUIRE(x && y && R > 0 && T > 0 && ld >= T && p >= 0.f && p < 1.f, "dropout_rows: bad args");
    if (is_f32) hipLaunchKernelGGL(dropout_rows_kernel<float>, dim3(grid_for(R * ld)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, R, T, ld, seed, p);
    else hipLaunchKernelGGL(dropout_rows_kernel<bf16_t>, dim3(grid_for(R * ld)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, R, T, ld, seed, p);
    return scl_check_launch("scl_dropout_rows");
}
extern "C" int scl_add_f32(const float* a, const float* b, float* out, void* out_bf16, int64_t n, void* stream) {
    SCL_REQUIRE(a && (out || out_bf16) && n > 0, "add_f32: bad args");
    hipLaunchKernelGGL(add_f32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, (bf16_t*)out_bf16, n);
    return scl_check_launch("scl_add_f32");
}
