// btse.hip — the "bio" branch of the reference's wav2vec2_btse plugin (BASELINE.json configs[4]) as two kernels, and the join in front
// of fc2.  Reference: model/wav2vec2_btse/model.py:210-238 (bioEncoderTransformersmall), :321-343 (Model.forward);
// model/wav2vec2_btse/transformer.py:17-52 (Encoder), :105-260 (MultiHeadAttention, window_size = 4 relative keys and values shared by
// the heads), :261-306 (FFN); model/wav2vec2_btse/modules.py:27-39 (LayerNorm over channels).
//
// The encoder is 40 752 parameters on a few hundred tokens: one WORKGROUP PER UTTERANCE runs all of it (embedding -> n_layers x
// {QKV, relative-position attention, out-projection + residual + LayerNorm, FFN + residual + LayerNorm} -> scoring conv at the last
// position) in ONE launch forward and ONE launch backward, fp32 throughout.  Per phase a thread owns one output channel and keeps that
// channel's weight row in registers while it walks the rows; K and V of the layer sit in LDS at a 33-float pitch (lane j reads row j:
// conflict-free) and one wave owns one query row, so the L x L scores, the soft-max and the relative-position skew (scores[i][j] +=
// q_i . emb_rel_k[j - i + 4], out[i] += p[i][j] emb_rel_v[j - i + 4] for |j - i| <= 4: what the reference's pad / reshape tricks at
// transformer.py:189-243 compute) live in registers.  The activations the backward needs go to a per-utterance scratch row in HBM
// (L2-resident: 392 L floats per layer); the backward recomputes the probabilities from the saved row log-sum-exp, first per query row
// (dq, delta, the relative-embedding gradients), then per key row (dk, dv) with Q and dA in LDS.  Parameter gradients leave as one slab
// row per utterance; the host sums the rows in index order.
// model.py:236 reads the LAST padded position times its mask: an utterance shorter than L scores exactly zero and contributes no
// gradient — the backward writes a zero row for it and returns.
#include "common.h"

namespace {
constexpr int BD = 32, BH = 4, BK = 8, BF = 128, BW = 4, NR = 2 * BW + 1, KP = 33, MAXJ = 8, NT = 256, MAXL = 64 * MAXJ;
constexpr float QSCALE = 0.35355339059327373f;      // 1 / sqrt(k_channels = 8), transformer.py:155
constexpr float EMB_SCALE = 5.656854249492381f;     // sqrt(bio_dim = 32), model.py:228
constexpr float FILL = -1e4f, EPS = 1e-5f;          // transformer.py:168, modules.py:28
enum { I_WQ, I_BQ, I_WK, I_BK, I_WV, I_BV, I_WO, I_BO, I_EK, I_EV, I_G1, I_B1, I_W1, I_C1, I_W2, I_C2, I_G2, I_B2 };
// per-utterance scratch, in units of L floats: layer l at l * O_LAYER, the encoder output at n_layers * O_LAYER, the backward's own
// buffers behind it
enum : int { O_XIN = 0, O_Q = 32, O_K = 64, O_V = 96, O_A = 128, O_S1 = 160, O_X1 = 192, O_S2 = 224, O_H = 256, O_ST = 384, O_LSE = 388, O_LAYER = 392 };
enum : int { G_DX = 0, G_Q = 32, G_K = 64, G_V = 96, G_DA = 128, G_DS1 = 160, G_DS2 = 192, G_DH = 224, G_DELTA = 352, G_TOTAL = 356 };

__device__ __forceinline__ float half_sum(float v) {      // sum over the 32 lanes that share a row (lanes 0-31 / 32-63 of the wave)
    v = lanes16_sum(v);
    return v + __shfl_xor(v, 16);
}
__device__ __forceinline__ void wave_sync_lds() {      // same-wave LDS hand-over: LDS operations of a wave retire in order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// sum of v over the 8 row groups of the (row group, channel) thread mapping, in index order; every thread gets its channel's total
__device__ __forceinline__ float rg_total(float v, float* red, int rg, int c) {
    red[rg * BD + c] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) s += red[g * BD + c];
    __syncthreads();
    return s;
}

// y[c] = bias + sum_k x[k] w[k] over a 32-float row read as 8 float4 (all 32 lanes of a row group read the same address)
__device__ __forceinline__ float dot32(const float* __restrict__ row, const float (&w)[BD], float acc) {
    const float4* r4 = (const float4*)row;
#pragma unroll
    for (int k4 = 0; k4 < BD / 4; ++k4) {
        const float4 x = r4[k4];
        acc = fmaf(x.x, w[4 * k4], acc); acc = fmaf(x.y, w[4 * k4 + 1], acc); acc = fmaf(x.z, w[4 * k4 + 2], acc); acc = fmaf(x.w, w[4 * k4 + 3], acc);
    }
    return acc;
}

__global__ __launch_bounds__(NT) void btse_bio_fwd_kernel(const SclBtseBio p) {
    extern __shared__ float lds[];
    __shared__ float Eks[NR * BK], Evs[NR * BK];
    const int b = blockIdx.x, t = threadIdx.x, L = p.L;
    const int len = min(max(p.lens[b], 0), L);            // commons.sequence_mask: arange(L) < length
    float* ws = p.ws + (int64_t)b * p.ws_stride;
    const int32_t* tok = p.bio + (int64_t)b * L;
    float* Ks = lds;
    float* Vs = lds + L * KP;
    const int c = t & 31, rg = t >> 5, lane = t & 63, w = t >> 6;
    for (int idx = t; idx < L * BD; idx += NT) {          // model.py:228,232 + transformer.py:42
        const int r = idx >> 5;
        const int tk = min(max(tok[r], 0), p.n_bios - 1);
        ws[idx] = r < len ? p.emb[tk * BD + (idx & 31)] * EMB_SCALE : 0.f;
    }
    __syncthreads();
    for (int l = 0; l < p.n_layers; ++l) {
        float* base = ws + (int64_t)l * O_LAYER * L;
        const float* const* W = p.lw[l];
        const float* xin = base;
        if (t < NR * BK) { Eks[t] = W[I_EK][t]; Evs[t] = W[I_EV][t]; }
        {   // q, k, v = conv_{q,k,v}(x)   (transformer.py:139-141)
            float wq[BD], wk[BD], wv[BD];
#pragma unroll
            for (int k = 0; k < BD; ++k) { wq[k] = W[I_WQ][c * BD + k]; wk[k] = W[I_WK][c * BD + k]; wv[k] = W[I_WV][c * BD + k]; }
            const float bq = W[I_BQ][c], bk = W[I_BK][c], bv = W[I_BV][c];
            for (int r = rg; r < L; r += 8) {
                const float* xr = xin + r * BD;
                const float aq = dot32(xr, wq, bq), ak = dot32(xr, wk, bk), av = dot32(xr, wv, bv);
                base[O_Q * L + r * BD + c] = aq; base[O_K * L + r * BD + c] = ak; base[O_V * L + r * BD + c] = av;
                Ks[r * KP + c] = ak; Vs[r * KP + c] = av;
            }
        }
        __syncthreads();
        for (int i = w; i < L; i += 4) {      // one wave per query row (transformer.py:148-186)
            const bool mi = i < len;
#pragma unroll 1
            for (int h = 0; h < BH; ++h) {
                float qs[BK];
#pragma unroll
                for (int d = 0; d < BK; ++d) qs[d] = base[O_Q * L + i * BD + h * BK + d] * QSCALE;
                float s[MAXJ];
                float mx = -INFINITY;
#pragma unroll
                for (int jt = 0; jt < MAXJ; ++jt) {
                    s[jt] = -INFINITY;
                    if (jt * 64 < L) {
                        const int j = jt * 64 + lane;
                        if (j < L) {
                            const float* kr = Ks + j * KP + h * BK;
                            float dot = 0.f;
#pragma unroll
                            for (int d = 0; d < BK; ++d) dot = fmaf(qs[d], kr[d], dot);
                            const int dj = j - i;
                            if (dj >= -BW && dj <= BW) {
                                const float* er = Eks + (dj + BW) * BK;
#pragma unroll
                                for (int d = 0; d < BK; ++d) dot = fmaf(qs[d], er[d], dot);
                            }
                            s[jt] = (mi && j < len) ? dot : FILL;
                        }
                        mx = fmaxf(mx, s[jt]);
                    }
                }
                mx = wave_max(mx);
                float sum = 0.f;
#pragma unroll
                for (int jt = 0; jt < MAXJ; ++jt)
                    if (jt * 64 < L) { s[jt] = (jt * 64 + lane < L) ? expf(s[jt] - mx) : 0.f; sum += s[jt]; }
                sum = wave_sum(sum);
                const float inv = 1.f / sum;
                float o[BK];
#pragma unroll
                for (int d = 0; d < BK; ++d) o[d] = 0.f;
#pragma unroll
                for (int jt = 0; jt < MAXJ; ++jt)
                    if (jt * 64 < L) {
                        const int j = jt * 64 + lane;
                        if (j < L) {
                            const float pj = s[jt] * inv;
                            const float* vr = Vs + j * KP + h * BK;
#pragma unroll
                            for (int d = 0; d < BK; ++d) o[d] = fmaf(pj, vr[d], o[d]);
                            const int dj = j - i;
                            if (dj >= -BW && dj <= BW) {
                                const float* er = Evs + (dj + BW) * BK;
#pragma unroll
                                for (int d = 0; d < BK; ++d) o[d] = fmaf(pj, er[d], o[d]);
                            }
                        }
                    }
#pragma unroll
                for (int d = 0; d < BK; ++d) o[d] = wave_sum(o[d]);
                if (lane == 0) {
#pragma unroll
                    for (int d = 0; d < BK; ++d) base[O_A * L + i * BD + h * BK + d] = o[d];
                    base[O_LSE * L + h * L + i] = mx + logf(sum);
                }
            }
        }
        __syncthreads();
        {   // x = LayerNorm(x + conv_o(attention))   (transformer.py:44-46)
            float wo[BD];
#pragma unroll
            for (int k = 0; k < BD; ++k) wo[k] = W[I_WO][c * BD + k];
            const float bo = W[I_BO][c], g1 = W[I_G1][c], b1 = W[I_B1][c];
            for (int r0 = 0; r0 < L; r0 += 8) {
                const int r = r0 + rg;
                const bool ok = r < L;
                const int rr = ok ? r : L - 1;
                const float sv = xin[rr * BD + c] + dot32(base + O_A * L + rr * BD, wo, bo);
                const float mean = half_sum(sv) * (1.f / BD);
                const float dv = sv - mean;
                const float rs = 1.f / sqrtf(half_sum(dv * dv) * (1.f / BD) + EPS);
                if (ok) {
                    base[O_S1 * L + r * BD + c] = sv;
                    base[O_X1 * L + r * BD + c] = dv * rs * g1 + b1;
                    if (c == 0) { base[O_ST * L + r * 4] = mean; base[O_ST * L + r * 4 + 1] = rs; }
                }
            }
        }
        __syncthreads();
        {   // h = relu(conv_1(x * mask))   (transformer.py:283-288)
            const int f = t & 127;
            float w1[BD];
#pragma unroll
            for (int k = 0; k < BD; ++k) w1[k] = W[I_W1][f * BD + k];
            const float c1 = W[I_C1][f];
            for (int r = t >> 7; r < L; r += 2) {
                const float hv = r < len ? dot32(base + O_X1 * L + r * BD, w1, c1) : c1;
                base[O_H * L + r * BF + f] = fmaxf(hv, 0.f);
            }
        }
        __syncthreads();
        {   // x = LayerNorm(x + conv_2(h * mask) * mask)   (transformer.py:290-291,48-50)
            float w2[BF];
#pragma unroll
            for (int f = 0; f < BF; ++f) w2[f] = W[I_W2][c * BF + f];
            const float c2 = W[I_C2][c], g2 = W[I_G2][c], b2 = W[I_B2][c];
            float* xnext = base + (int64_t)O_LAYER * L;
            for (int r0 = 0; r0 < L; r0 += 8) {
                const int r = r0 + rg;
                const bool ok = r < L;
                const int rr = ok ? r : L - 1;
                float y = 0.f;
                if (rr < len) {
                    const float4* h4 = (const float4*)(base + O_H * L + rr * BF);
                    y = c2;
#pragma unroll
                    for (int f4 = 0; f4 < BF / 4; ++f4) {
                        const float4 x = h4[f4];
                        y = fmaf(x.x, w2[4 * f4], y); y = fmaf(x.y, w2[4 * f4 + 1], y); y = fmaf(x.z, w2[4 * f4 + 2], y); y = fmaf(x.w, w2[4 * f4 + 3], y);
                    }
                }
                const float sv = base[O_X1 * L + rr * BD + c] + y;
                const float mean = half_sum(sv) * (1.f / BD);
                const float dv = sv - mean;
                const float rs = 1.f / sqrtf(half_sum(dv * dv) * (1.f / BD) + EPS);
                if (ok) {
                    base[O_S2 * L + r * BD + c] = sv;
                    xnext[r * BD + c] = dv * rs * g2 + b2;
                    if (c == 0) { base[O_ST * L + r * 4 + 2] = mean; base[O_ST * L + r * 4 + 3] = rs; }
                }
            }
        }
        __syncthreads();
    }
    float* xfin = ws + (int64_t)p.n_layers * O_LAYER * L;
    for (int idx = len * BD + t; idx < L * BD; idx += NT) xfin[idx] = 0.f;          // transformer.py:51
    __syncthreads();
    const bool mlast = L - 1 < len;                                                   // model.py:234-236
    for (int o = t; o < p.bio_out; o += NT) {
        float y = 0.f;
        if (mlast) {
            y = p.bs[o];
#pragma unroll 8
            for (int k = 0; k < BD; ++k) y = fmaf(xfin[(L - 1) * BD + k], p.Ws[o * BD + k], y);
        }
        p.out[(int64_t)b * p.out_ld + o] = y;
    }
}

// LayerNorm backward over the (row group, channel) mapping: dy rows in `gin`, pre-norm rows in `pre`, (mean, rstd) at st[4 r + so];
// writes ds to `gout`, returns this thread's partial sums of dgamma / dbeta.
__device__ __forceinline__ void ln_bwd_rows(const float* gin, const float* pre, const float* st, int so, float gamma, float* gout, int L, int rg, int c,
                                            float& dg, float& db) {
    dg = 0.f; db = 0.f;
    for (int r0 = 0; r0 < L; r0 += 8) {
        const int r = r0 + rg;
        const bool ok = r < L;
        const int rr = ok ? r : L - 1;
        const float mu = st[rr * 4 + so], rs = st[rr * 4 + so + 1];
        const float xh = (pre[rr * BD + c] - mu) * rs;
        const float dy = ok ? gin[rr * BD + c] : 0.f;
        dg = fmaf(dy, xh, dg); db += dy;
        const float dxh = dy * gamma;
        const float m1 = half_sum(dxh) * (1.f / BD), m2 = half_sum(dxh * xh) * (1.f / BD);
        if (ok) gout[r * BD + c] = rs * (dxh - m1 - xh * m2);
    }
}

__global__ __launch_bounds__(NT) void btse_bio_bwd_kernel(const SclBtseBio p) {
    extern __shared__ float lds[];
    __shared__ float Eks[NR * BK], Evs[NR * BK], red[8 * BD], red2[2 * BF], bandS[4][NR + 1], bandP[4][NR + 1], epart[4][2][NR * BK];
    const int b = blockIdx.x, t = threadIdx.x, L = p.L;
    const int len = min(max(p.lens[b], 0), L);
    float* slab = p.slab + (int64_t)b * p.slab_ld;
    if (len < L) {      // the read-out position is padding: zero score, zero gradient (model.py:234-236)
        for (int64_t idx = t; idx < p.slab_ld; idx += NT) slab[idx] = 0.f;
        return;
    }
    float* ws = p.ws + (int64_t)b * p.ws_stride;
    const int32_t* tok = p.bio + (int64_t)b * L;
    const float* dsc = p.d_out + (int64_t)b * p.dout_ld;
    float* A0 = lds;                 // K, then Q * scale
    float* A1 = lds + L * KP;        // V, then dA
    const int c = t & 31, rg = t >> 5, lane = t & 63, w = t >> 6;
    float* G = ws + ((int64_t)p.n_layers * O_LAYER + BD) * L;
    float* gdx = G + G_DX * L;
    {   // scoring conv at the last position
        const float* xfin = ws + (int64_t)p.n_layers * O_LAYER * L + (L - 1) * BD;
        for (int o = t; o < p.bio_out; o += NT) slab[p.go_bs + o] = dsc[o];
        for (int idx = t; idx < p.bio_out * BD; idx += NT) slab[p.go_Ws + idx] = dsc[idx >> 5] * xfin[idx & 31];
        for (int idx = t; idx < (L - 1) * BD; idx += NT) gdx[idx] = 0.f;
        if (t < BD) {
            float a = 0.f;
            for (int o = 0; o < p.bio_out; ++o) a = fmaf(dsc[o], p.Ws[o * BD + t], a);
            gdx[(L - 1) * BD + t] = a;
        }
    }
    __syncthreads();
    for (int l = p.n_layers - 1; l >= 0; --l) {
        float* base = ws + (int64_t)l * O_LAYER * L;
        const float* const* W = p.lw[l];
        const int32_t* go = p.go[l];
        const float* xin = base;
        const float* st = base + O_ST * L;
        if (t < NR * BK) { Eks[t] = W[I_EK][t]; Evs[t] = W[I_EV][t]; }
        {   // LayerNorm 2
            float dg, db;
            ln_bwd_rows(gdx, base + O_S2 * L, st, 2, W[I_G2][c], G + G_DS2 * L, L, rg, c, dg, db);
            dg = rg_total(dg, red, rg, c); db = rg_total(db, red, rg, c);
            if (rg == 0) { slab[go[I_G2] + c] = dg; slab[go[I_B2] + c] = db; }
        }
        const float* ds2 = G + G_DS2 * L;
        {   // conv_2: bias, weight, input gradient (x ReLU')
            float a = 0.f;
            for (int r = rg; r < L; r += 8) a += ds2[r * BD + c];
            a = rg_total(a, red, rg, c);
            if (rg == 0) slab[go[I_C2] + c] = a;
            const int f = t & 127, cg = t >> 7;
            float acc[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[u] = 0.f;
            for (int r = 0; r < L; ++r) {
                const float hv = base[O_H * L + r * BF + f];
#pragma unroll
                for (int u = 0; u < 16; ++u) acc[u] = fmaf(ds2[r * BD + cg + 2 * u], hv, acc[u]);
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) slab[go[I_W2] + (cg + 2 * u) * BF + f] = acc[u];
            float w2c[BD];
#pragma unroll
            for (int k = 0; k < BD; ++k) w2c[k] = W[I_W2][k * BF + f];
            float dc1 = 0.f;
            for (int r = cg; r < L; r += 2) {
                float v = dot32(ds2 + r * BD, w2c, 0.f);
                v = base[O_H * L + r * BF + f] > 0.f ? v : 0.f;
                G[G_DH * L + r * BF + f] = v;
                dc1 += v;
            }
            red2[cg * BF + f] = dc1;
            __syncthreads();
            if (t < BF) slab[go[I_C1] + t] = red2[t] + red2[BF + t];
        }
        __syncthreads();
        const float* gdh = G + G_DH * L;
        {   // conv_1: weight and input gradient; + the residual branch
            const int fg = rg;
            float acc[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[u] = 0.f;
            for (int r = 0; r < L; ++r) {
                const float xk = base[O_X1 * L + r * BD + c];
#pragma unroll
                for (int u = 0; u < 16; ++u) acc[u] = fmaf(gdh[r * BF + fg + 8 * u], xk, acc[u]);
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) slab[go[I_W1] + (fg + 8 * u) * BD + c] = acc[u];
            float w1c[BF];
#pragma unroll
            for (int f = 0; f < BF; ++f) w1c[f] = W[I_W1][f * BD + c];
            for (int r = rg; r < L; r += 8) {
                const float4* g4 = (const float4*)(gdh + r * BF);
                float a = ds2[r * BD + c];
#pragma unroll
                for (int f4 = 0; f4 < BF / 4; ++f4) {
                    const float4 x = g4[f4];
                    a = fmaf(x.x, w1c[4 * f4], a); a = fmaf(x.y, w1c[4 * f4 + 1], a); a = fmaf(x.z, w1c[4 * f4 + 2], a); a = fmaf(x.w, w1c[4 * f4 + 3], a);
                }
                gdx[r * BD + c] = a;
            }
        }
        __syncthreads();
        {   // LayerNorm 1
            float dg, db;
            ln_bwd_rows(gdx, base + O_S1 * L, st, 0, W[I_G1][c], G + G_DS1 * L, L, rg, c, dg, db);
            dg = rg_total(dg, red, rg, c); db = rg_total(db, red, rg, c);
            if (rg == 0) { slab[go[I_G1] + c] = dg; slab[go[I_B1] + c] = db; }
        }
        const float* ds1 = G + G_DS1 * L;
        {   // conv_o
            float a = 0.f;
            for (int r = rg; r < L; r += 8) a += ds1[r * BD + c];
            a = rg_total(a, red, rg, c);
            if (rg == 0) slab[go[I_BO] + c] = a;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int r = 0; r < L; ++r) {
                const float ak = base[O_A * L + r * BD + c];
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] = fmaf(ds1[r * BD + rg + 8 * u], ak, acc[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) slab[go[I_WO] + (rg + 8 * u) * BD + c] = acc[u];
            float woc[BD];
#pragma unroll
            for (int k = 0; k < BD; ++k) woc[k] = W[I_WO][k * BD + c];
            for (int r = rg; r < L; r += 8) G[G_DA * L + r * BD + c] = dot32(ds1 + r * BD, woc, 0.f);
            for (int r = rg; r < L; r += 8) { A0[r * KP + c] = base[O_K * L + r * BD + c]; A1[r * KP + c] = base[O_V * L + r * BD + c]; }
        }
        __syncthreads();
        const float* gda = G + G_DA * L;
        {   // attention, pass 1: one wave per query row -> dq, delta, the relative-embedding gradients
            float aEk0 = 0.f, aEk1 = 0.f, aEv0 = 0.f, aEv1 = 0.f;
            for (int i = w; i < L; i += 4) {
#pragma unroll 1
                for (int h = 0; h < BH; ++h) {
                    float qs[BK], da[BK];
#pragma unroll
                    for (int d = 0; d < BK; ++d) { qs[d] = base[O_Q * L + i * BD + h * BK + d] * QSCALE; da[d] = gda[i * BD + h * BK + d]; }
                    const float lse = base[O_LSE * L + h * L + i];
                    float pv[MAXJ], dP[MAXJ];
                    float part = 0.f;
#pragma unroll
                    for (int jt = 0; jt < MAXJ; ++jt) {
                        pv[jt] = 0.f; dP[jt] = 0.f;
                        if (jt * 64 < L) {
                            const int j = jt * 64 + lane;
                            if (j < L) {
                                const float* kr = A0 + j * KP + h * BK;
                                const float* vr = A1 + j * KP + h * BK;
                                float dot = 0.f, g = 0.f;
#pragma unroll
                                for (int d = 0; d < BK; ++d) { dot = fmaf(qs[d], kr[d], dot); g = fmaf(da[d], vr[d], g); }
                                const int dj = j - i;
                                if (dj >= -BW && dj <= BW) {
                                    const float* ek = Eks + (dj + BW) * BK;
                                    const float* ev = Evs + (dj + BW) * BK;
#pragma unroll
                                    for (int d = 0; d < BK; ++d) { dot = fmaf(qs[d], ek[d], dot); g = fmaf(da[d], ev[d], g); }
                                }
                                pv[jt] = expf(dot - lse);
                                dP[jt] = g;
                                part = fmaf(pv[jt], g, part);
                            }
                        }
                    }
                    const float delta = wave_sum(part);
                    float dq[BK];
#pragma unroll
                    for (int d = 0; d < BK; ++d) dq[d] = 0.f;
                    if (lane <= NR) { bandS[w][lane] = 0.f; bandP[w][lane] = 0.f; }
#pragma unroll
                    for (int jt = 0; jt < MAXJ; ++jt)
                        if (jt * 64 < L) {
                            const int j = jt * 64 + lane;
                            if (j < L) {
                                const float dS = pv[jt] * (dP[jt] - delta);
                                const float* kr = A0 + j * KP + h * BK;
#pragma unroll
                                for (int d = 0; d < BK; ++d) dq[d] = fmaf(dS, kr[d], dq[d]);
                                const int dj = j - i;
                                if (dj >= -BW && dj <= BW) {
                                    const float* ek = Eks + (dj + BW) * BK;
#pragma unroll
                                    for (int d = 0; d < BK; ++d) dq[d] = fmaf(dS, ek[d], dq[d]);
                                    bandS[w][dj + BW] = dS; bandP[w][dj + BW] = pv[jt];
                                }
                            }
                        }
#pragma unroll
                    for (int d = 0; d < BK; ++d) dq[d] = wave_sum(dq[d]);
                    if (lane == 0) {
#pragma unroll
                        for (int d = 0; d < BK; ++d) G[G_Q * L + i * BD + h * BK + d] = dq[d] * QSCALE;
                        G[G_DELTA * L + h * L + i] = delta;
                    }
                    wave_sync_lds();
                    {   // d emb_rel_k[r][d] += dS[i][i + r - 4] qs[d];  d emb_rel_v[r][d] += p[i][i + r - 4] da[d]   (lane = 8 r + d; r = 8 on lanes 0-7 again)
                        const int d = lane & 7;
                        const float qd = base[O_Q * L + i * BD + h * BK + d] * QSCALE, dd = gda[i * BD + h * BK + d];
                        aEk0 = fmaf(bandS[w][lane >> 3], qd, aEk0); aEv0 = fmaf(bandP[w][lane >> 3], dd, aEv0);
                        if (lane < BK) { aEk1 = fmaf(bandS[w][NR - 1], qd, aEk1); aEv1 = fmaf(bandP[w][NR - 1], dd, aEv1); }
                    }
                    wave_sync_lds();
                }
            }
            epart[w][0][lane] = aEk0; epart[w][1][lane] = aEv0;
            if (lane < BK) { epart[w][0][64 + lane] = aEk1; epart[w][1][64 + lane] = aEv1; }
        }
        __syncthreads();
        if (t < 2 * NR * BK) {
            const int which = t / (NR * BK), e = t % (NR * BK);
            slab[go[which ? I_EV : I_EK] + e] = epart[0][which][e] + epart[1][which][e] + epart[2][which][e] + epart[3][which][e];
        }
        for (int r = rg; r < L; r += 8) { A0[r * KP + c] = base[O_Q * L + r * BD + c] * QSCALE; A1[r * KP + c] = gda[r * BD + c]; }
        __syncthreads();
        for (int j = w; j < L; j += 4) {      // attention, pass 2: one wave per key row -> dk, dv
#pragma unroll 1
            for (int h = 0; h < BH; ++h) {
                float kj[BK], vj[BK], dk[BK], dv[BK];
#pragma unroll
                for (int d = 0; d < BK; ++d) { kj[d] = base[O_K * L + j * BD + h * BK + d]; vj[d] = base[O_V * L + j * BD + h * BK + d]; dk[d] = 0.f; dv[d] = 0.f; }
#pragma unroll
                for (int it = 0; it < MAXJ; ++it)
                    if (it * 64 < L) {
                        const int i = it * 64 + lane;
                        if (i < L) {
                            const float* qr = A0 + i * KP + h * BK;
                            const float* ar = A1 + i * KP + h * BK;
                            float dot = 0.f, g = 0.f;
#pragma unroll
                            for (int d = 0; d < BK; ++d) { dot = fmaf(qr[d], kj[d], dot); g = fmaf(ar[d], vj[d], g); }
                            const int dj = j - i;
                            if (dj >= -BW && dj <= BW) {
                                const float* ek = Eks + (dj + BW) * BK;
                                const float* ev = Evs + (dj + BW) * BK;
#pragma unroll
                                for (int d = 0; d < BK; ++d) { dot = fmaf(qr[d], ek[d], dot); g = fmaf(ar[d], ev[d], g); }
                            }
                            const float pij = expf(dot - base[O_LSE * L + h * L + i]);
                            const float dS = pij * (g - G[G_DELTA * L + h * L + i]);
#pragma unroll
                            for (int d = 0; d < BK; ++d) { dk[d] = fmaf(dS, qr[d], dk[d]); dv[d] = fmaf(pij, ar[d], dv[d]); }
                        }
                    }
#pragma unroll
                for (int d = 0; d < BK; ++d) { dk[d] = wave_sum(dk[d]); dv[d] = wave_sum(dv[d]); }
                if (lane == 0) {
#pragma unroll
                    for (int d = 0; d < BK; ++d) { G[G_K * L + j * BD + h * BK + d] = dk[d]; G[G_V * L + j * BD + h * BK + d] = dv[d]; }
                }
            }
        }
        __syncthreads();
        {   // conv_q / conv_k / conv_v: biases, weights, input gradient + the residual branch
            const float* gq = G + G_Q * L;
            const float* gk = G + G_K * L;
            const float* gv = G + G_V * L;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
            for (int r = rg; r < L; r += 8) { a0 += gq[r * BD + c]; a1 += gk[r * BD + c]; a2 += gv[r * BD + c]; }
            a0 = rg_total(a0, red, rg, c); a1 = rg_total(a1, red, rg, c); a2 = rg_total(a2, red, rg, c);
            if (rg == 0) { slab[go[I_BQ] + c] = a0; slab[go[I_BK] + c] = a1; slab[go[I_BV] + c] = a2; }
            float acc[12];
#pragma unroll
            for (int u = 0; u < 12; ++u) acc[u] = 0.f;
            for (int r = 0; r < L; ++r) {
                const float xk = xin[r * BD + c];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc[u] = fmaf(gq[r * BD + rg + 8 * u], xk, acc[u]);
                    acc[4 + u] = fmaf(gk[r * BD + rg + 8 * u], xk, acc[4 + u]);
                    acc[8 + u] = fmaf(gv[r * BD + rg + 8 * u], xk, acc[8 + u]);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                slab[go[I_WQ] + (rg + 8 * u) * BD + c] = acc[u];
                slab[go[I_WK] + (rg + 8 * u) * BD + c] = acc[4 + u];
                slab[go[I_WV] + (rg + 8 * u) * BD + c] = acc[8 + u];
            }
            float wqc[BD], wkc[BD], wvc[BD];
#pragma unroll
            for (int k = 0; k < BD; ++k) { wqc[k] = W[I_WQ][k * BD + c]; wkc[k] = W[I_WK][k * BD + c]; wvc[k] = W[I_WV][k * BD + c]; }
            for (int r = rg; r < L; r += 8) {
                float a = ds1[r * BD + c];
                a = dot32(gq + r * BD, wqc, a); a = dot32(gk + r * BD, wkc, a); a = dot32(gv + r * BD, wvc, a);
                gdx[r * BD + c] = a;
            }
        }
        __syncthreads();
    }
    for (int idx = t; idx < p.n_bios * BD; idx += NT) {      // embedding rows (model.py:228)
        const int tk = idx >> 5, cc = idx & 31;
        float a = 0.f;
        for (int r = 0; r < L; ++r)
            if (min(max(tok[r], 0), p.n_bios - 1) == tk) a += gdx[r * BD + cc];
        slab[p.go_emb + idx] = a * EMB_SCALE;
    }
}

__global__ void btse_join_fwd_kernel(const float* __restrict__ emb, const float* __restrict__ s, const float* __restrict__ W1, const float* __restrict__ b1,
                                     float* __restrict__ out, int B, int C, int bo, int is_add) {
    const int r = blockIdx.x;
    if (!is_add) {
        for (int cc = threadIdx.x; cc < C; cc += blockDim.x) out[(int64_t)r * (C + bo) + cc] = emb[(int64_t)r * C + cc];
        return;
    }
    for (int o = threadIdx.x; o < bo; o += blockDim.x) {
        float a = b1[o];
        for (int cc = 0; cc < C; ++cc) a = fmaf(emb[(int64_t)r * C + cc], W1[(int64_t)o * C + cc], a);
        out[(int64_t)r * bo + o] = a + s[(int64_t)r * bo + o];
    }
}

__global__ void btse_join_bwd_kernel(const float* __restrict__ db, const float* __restrict__ emb, const float* __restrict__ W1, float* __restrict__ demb,
                                     float* __restrict__ ds, float* __restrict__ dW1, float* __restrict__ db1, int B, int C, int bo, int is_add) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    if (!is_add) {
        for (int idx = t; idx < B * (C + bo); idx += nth) {
            const int r = idx / (C + bo), cc = idx % (C + bo);
            if (cc < C) demb[(int64_t)r * C + cc] = db[idx];
            else ds[(int64_t)r * bo + cc - C] = db[idx];
        }
        return;
    }
    for (int idx = t; idx < B * bo; idx += nth) ds[idx] = db[idx];
    for (int idx = t; idx < B * C; idx += nth) {
        const int r = idx / C, cc = idx % C;
        float a = 0.f;
        for (int o = 0; o < bo; ++o) a = fmaf(db[(int64_t)r * bo + o], W1[(int64_t)o * C + cc], a);
        demb[idx] = a;
    }
    for (int idx = t; idx < bo * C; idx += nth) {
        const int o = idx / C, cc = idx % C;
        float a = 0.f;
        for (int r = 0; r < B; ++r) a = fmaf(db[(int64_t)r * bo + o], emb[(int64_t)r * C + cc], a);
        dW1[idx] = a;
    }
    for (int o = t; o < bo; o += nth) {
        float a = 0.f;
        for (int r = 0; r < B; ++r) a += db[(int64_t)r * bo + o];
        db1[o] = a;
    }
}

int bio_check(const SclBtseBio* p, const char* what, bool bwd) {
    SCL_REQUIRE(p, "%s: null descriptor", what);
    if (!scl_btse_bio_supported(p->bio_dim, p->n_heads, p->pf_dim, p->n_layers, p->window, p->bio_out, p->L)) {
        scl_set_error("%s: unsupported shape (bio_dim %d, heads %d, pf_dim %d, layers %d, window %d, bio_out %d, L %d): the kernel serves 32 / 4 / 128 / "
                      "1..8 / 4 / <= 256 / 1..%d", what, p->bio_dim, p->n_heads, p->pf_dim, p->n_layers, p->window, p->bio_out, p->L, MAXL);
        return SCL_EUNSUPPORTED;
    }
    SCL_REQUIRE(p->B > 0 && p->n_bios > 0 && p->emb && p->Ws && p->bs && p->bio && p->lens && p->ws, "%s: null pointer or empty batch", what);
    SCL_REQUIRE(p->ws_stride >= scl_btse_bio_ws_floats(p->n_layers, p->L), "%s: ws_stride %lld < %lld floats", what, (long long)p->ws_stride,
                (long long)scl_btse_bio_ws_floats(p->n_layers, p->L));
    for (int l = 0; l < p->n_layers; ++l)
        for (int i = 0; i < 18; ++i) SCL_REQUIRE(p->lw[l][i], "%s: layer %d tensor %d is null", what, l, i);
    if (!bwd) { SCL_REQUIRE(p->out && p->out_ld >= p->bio_out, "%s: bad output", what); }
    else { SCL_REQUIRE(p->d_out && p->slab && p->dout_ld >= p->bio_out && p->slab_ld > 0, "%s: bad gradient buffers", what); }
    return SCL_OK;
}
}  // namespace

extern "C" int scl_btse_bio_supported(int bio_dim, int n_heads, int pf_dim, int n_layers, int window, int bio_out, int L) {
    return bio_dim == BD && n_heads == BH && pf_dim == BF && window == BW && n_layers >= 1 && n_layers <= 8 && bio_out >= 1 && bio_out <= 256 &&
           L >= 1 && L <= MAXL;
}
extern "C" int64_t scl_btse_bio_ws_floats(int n_layers, int L) { return ((int64_t)n_layers * O_LAYER + BD + G_TOTAL) * L; }

extern "C" int scl_btse_bio_fwd(const SclBtseBio* p, void* stream) {
    const int rc = bio_check(p, "btse_bio_fwd", false);
    if (rc != SCL_OK) return rc;
    const size_t lds = (size_t)2 * p->L * KP * sizeof(float);
    if (lds > 65536) (void)hipFuncSetAttribute((const void*)btse_bio_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(btse_bio_fwd_kernel, dim3(p->B), dim3(NT), lds, (hipStream_t)stream, *p);
    return scl_check_launch("scl_btse_bio_fwd");
}
extern "C" int scl_btse_bio_bwd(const SclBtseBio* p, void* stream) {
    const int rc = bio_check(p, "btse_bio_bwd", true);
    if (rc != SCL_OK) return rc;
    const size_t lds = (size_t)2 * p->L * KP * sizeof(float);
    if (lds > 65536) (void)hipFuncSetAttribute((const void*)btse_bio_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(btse_bio_bwd_kernel, dim3(p->B), dim3(NT), lds, (hipStream_t)stream, *p);
    return scl_check_launch("scl_btse_bio_bwd");
}
extern "C" int scl_btse_join_fwd(const float* emb, const float* s, const float* W1, const float* b1, float* b, int B, int C, int bio_out, int is_add,
                                 void* stream) {
    SCL_REQUIRE(emb && b && B > 0 && B <= 4096 && C > 0 && bio_out > 0, "btse_join_fwd: bad args");
    SCL_REQUIRE(!is_add || (s && W1 && b1), "btse_join_fwd: is_add needs s, W1, b1");
    hipLaunchKernelGGL(btse_join_fwd_kernel, dim3(B), dim3(128), 0, (hipStream_t)stream, emb, s, W1, b1, b, B, C, bio_out, is_add);
    return scl_check_launch("scl_btse_join_fwd");
}
extern "C" int scl_btse_join_bwd(const float* db, const float* emb, const float* W1, float* demb, float* ds, float* dW1, float* db1, int B, int C,
                                 int bio_out, int is_add, void* stream) {
    SCL_REQUIRE(db && demb && ds && B > 0 && B <= 4096 && C > 0 && bio_out > 0, "btse_join_bwd: bad args");
    SCL_REQUIRE(!is_add || (emb && W1 && dW1 && db1), "btse_join_bwd: is_add needs emb, W1, dW1, db1");
    hipLaunchKernelGGL(btse_join_bwd_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, db, emb, W1, demb, ds, dW1, db1, B, C, bio_out, is_add);
    return scl_check_launch("scl_btse_join_bwd");
}
