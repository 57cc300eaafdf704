// graph.hip — the graph module of the AASIST back-end (model/wav2vec2_aasist.py:62-155 GraphAttentionLayer, :158-332
// HtrgGraphAttentionLayer, :336-374 GraphPool, :545-604 the two heterogeneous branches and the read-out) as a handful of kernels per
// direction instead of ~1300 small launches: one workgroup per utterance keeps a layer's node matrices (<= 80 nodes x 64 features)
// in LDS and walks the layer's whole arithmetic; only the BatchNorm batch statistics force a kernel boundary.
//
//   drop        xd = x * keep-mask                                              (input dropout of the first two layers)
//   [scl_gat_score_fwd: the pairwise scores  s[i][j] = sum_o tanh(W (x_i * x_j) + b)_o a[o]   — csrc/gat.hip, one launch per layer]
//   post_fwd    A = softmax_j(s / temp); g = A xd; y = g Wa^T + ba + xd Wb^T + bb; master node update (heterogeneous layers);
//               BatchNorm statistics of y (fp64 atomics, finished by the last block: csrc/rs_finish.h)
//   pre_fwd     h = selu(bn(y)) of the layer(s) below; GraphPool (sigmoid score, top-k by rank counting, gate, gather); proj_type1 / 2;
//               input dropout -> xd of the next layer
//   final_fwd   BatchNorm + SELU of the last layers, residual adds, drop_way, branch max, |max| / mean read-out, dropout, out_layer
// and the mirror-image backward kernels; parameter gradients leave every block as one slab row [B][P] that graph_reduce sums over B in
// index order into the parameter gradients.  All fp32; dropout masks are counter hashes (csrc/common.h) recomputed in the backward.
#include "rs_finish.h"

namespace {

constexpr int GT = 256;       // threads per block
constexpr int GP = 4;         // row padding (floats) of the LDS images: 16-byte reads of consecutive rows spread over the banks

struct Bump {
    float* cur;
    __device__ float* take(int n) { float* p = cur; cur += (n + 3) & ~3; return p; }
};

__device__ __forceinline__ void g_load(float* dst, int ldd, const float* __restrict__ src, int rows, int cols) {      // cols % 4 == 0
    const int c4n = cols >> 2;
    for (int f = threadIdx.x; f < rows * c4n; f += GT) {
        const int r = f / c4n, c4 = f - r * c4n;
        *reinterpret_cast<f32x4*>(dst + r * ldd + 4 * c4) = *reinterpret_cast<const f32x4*>(src + (size_t)r * cols + 4 * c4);
    }
}
__device__ __forceinline__ void g_store(float* __restrict__ dst, const float* src, int lds_, int rows, int cols) {
    const int c4n = cols >> 2;
    for (int f = threadIdx.x; f < rows * c4n; f += GT) {
        const int r = f / c4n, c4 = f - r * c4n;
        *reinterpret_cast<f32x4*>(dst + (size_t)r * cols + 4 * c4) = *reinterpret_cast<const f32x4*>(src + r * lds_ + 4 * c4);
    }
}
// C[n][o] = (acc ? C[n][o] : 0) + bias[o] + sum_d A[n][d] * W[o][d]       A, C: LDS; wl: LDS image of W [Do][D + GP]; GT % Do == 0
__device__ __forceinline__ void mm_nt(float* C, int ldc, const float* A, int lda, const float* wl, const float* __restrict__ bias, int N, int D, int Do, bool acc) {
    const int o = threadIdx.x % Do, grp = threadIdx.x / Do, G = GT / Do;
    const float bo = bias ? bias[o] : 0.f;
    const float* wr = wl + o * (D + GP);
    for (int n = grp; n < N; n += G) {
        float s = (acc ? C[n * ldc + o] : 0.f) + bo;
        const float* ar = A + n * lda;
        for (int d = 0; d < D; d += 4) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(ar + d), w4 = *reinterpret_cast<const f32x4*>(wr + d);
            s = fmaf(a4[0], w4[0], s); s = fmaf(a4[1], w4[1], s); s = fmaf(a4[2], w4[2], s); s = fmaf(a4[3], w4[3], s);
        }
        C[n * ldc + o] = s;
    }
}
// X[n][d] = (acc ? X : 0) + sum_o Y[n][o] * W[o][d]          GT % D == 0
__device__ __forceinline__ void mm_nn(float* X, int ldx, const float* Y, int ldy, const float* wl, int N, int D, int Do, bool acc) {
    const int d = threadIdx.x % D, grp = threadIdx.x / D, G = GT / D;
    for (int n = grp; n < N; n += G) {
        float s = acc ? X[n * ldx + d] : 0.f;
        const float* yr = Y + n * ldy;
        for (int o = 0; o < Do; o += 4) {
            const f32x4 y4 = *reinterpret_cast<const f32x4*>(yr + o);
            s = fmaf(y4[0], wl[o * (D + GP) + d], s); s = fmaf(y4[1], wl[(o + 1) * (D + GP) + d], s);
            s = fmaf(y4[2], wl[(o + 2) * (D + GP) + d], s); s = fmaf(y4[3], wl[(o + 3) * (D + GP) + d], s);
        }
        X[n * ldx + d] = s;
    }
}
// dW[o][d] = colscale[d] * sum_n Y[n][o] * X[n][d]  -> global (row-major [Do][D]); GT % Do == 0, D / (GT / Do) <= 16
__device__ __forceinline__ void mm_tn_out(float* __restrict__ dW, const float* Y, int ldy, const float* X, int ldx, int N, int D, int Do, const float* colscale) {
    const int o = threadIdx.x % Do, grp = threadIdx.x / Do, G = GT / Do, nd = D / G;
    float s[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) s[k] = 0.f;
    for (int n = 0; n < N; ++n) {
        const float y = Y[n * ldy + o];
        const float* xr = X + n * ldx + grp;
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (k < nd) s[k] = fmaf(y, xr[G * k], s[k]);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if (k < nd) { const int d = grp + G * k; dW[o * D + d] = colscale ? s[k] * colscale[d] : s[k]; }
}
__device__ __forceinline__ void stage_w(float* wl, const float* __restrict__ W, int Do, int D) { g_load(wl, D + GP, W, Do, D); }

// row softmax of A [N][ld] (N <= 128 columns), one wave per row
__device__ __forceinline__ void softmax_rows(float* A, int N, int ld, float scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = wave; r < N; r += GT / 64) {
        float* row = A + r * ld;
        const float v0 = lane < N ? row[lane] * scale : -INFINITY, v1 = lane + 64 < N ? row[lane + 64] * scale : -INFINITY;
        const float mx = wave_max(fmaxf(v0, v1));
        const float e0 = lane < N ? __expf(v0 - mx) : 0.f, e1 = lane + 64 < N ? __expf(v1 - mx) : 0.f;
        const float inv = 1.0f / wave_sum(e0 + e1);
        if (lane < N) row[lane] = e0 * inv;
        if (lane + 64 < N) row[lane + 64] = e1 * inv;
    }
}
__device__ __forceinline__ float g_tanh(float v) { const float e = __expf(2.0f * v); return 1.0f - 2.0f / (e + 1.0f); }

typedef SclGraphBn BnK;

// block partial sums (sum a, sum b per channel) from an LDS matrix pair via f(n, c, &u, &v); finishes into BatchNorm statistics (fwd) or backward means
template <class F>
__device__ __forceinline__ void bn_block_sums(double* mine, int N, int C, F f, bool accumulate = false) {
    if ((int)threadIdx.x < C) {
        double s0 = accumulate ? mine[threadIdx.x] : 0.0, s1 = accumulate ? mine[C + threadIdx.x] : 0.0;
        for (int n = 0; n < N; ++n) { float u, v; f(n, (int)threadIdx.x, u, v); s0 += (double)u; s1 += (double)v; }
        mine[threadIdx.x] = s0; mine[C + threadIdx.x] = s1;
    }
    __syncthreads();
}
__device__ __forceinline__ void bn_finish_fwd(const BnK& k, int C, const double* tot) {
    const int c = threadIdx.x;
    if (c >= C) return;
    if (!k.training) return;
    const double m = tot[c] / k.nvalid;
    double var = tot[C + c] / k.nvalid - m * m;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)k.eps));
    k.stats[c] = (float)m; k.stats[C + c] = rstd; k.stats[2 * C + c] = k.gamma[c] * rstd; k.stats[3 * C + c] = k.beta[c];
    const double unb = k.nvalid > 1.0 ? var * k.nvalid / (k.nvalid - 1.0) : var;
    k.run_mean[c] = (float)((1.0 - k.momentum) * k.run_mean[c] + k.momentum * m);
    k.run_var[c] = (float)((1.0 - k.momentum) * k.run_var[c] + k.momentum * unb);
    if (c == 0) *k.nbt += 1;
}
__device__ __forceinline__ void bn_finish_bwd(const BnK& k, int C, const double* tot) {
    const int c = threadIdx.x;
    if (c >= C) return;
    k.dbeta[c] += (float)tot[c];
    k.dgamma[c] += (float)tot[C + c];
    k.bstats[c] = k.training ? (float)(tot[c] / k.nvalid) : 0.f;
    k.bstats[C + c] = k.training ? (float)(tot[C + c] / k.nvalid) : 0.f;
}

// ---- input dropout of the first two layers ------------------------------------------------------------------------------------------
struct DropArgs { const float* x[2]; float* y[2]; long long n[2]; unsigned seed[2]; float p; };
__global__ __launch_bounds__(GT) void graph_drop_kernel(const DropArgs a) {
    const int i = blockIdx.y;
    for (long long e = (long long)blockIdx.x * GT + threadIdx.x; e < a.n[i]; e += (long long)gridDim.x * GT)
        a.y[i][e] = a.p > 0.f ? a.x[i][e] * dropout_scale(a.seed[i], (uint64_t)e, a.p) : a.x[i][e];
}

// ---- attention layer, second half ------------------------------------------------------------------------------------------------------
typedef SclGraphLayer PostI;
struct PostArgs { PostI L[2]; };

__global__ __launch_bounds__(GT) void graph_post_fwd_kernel(const PostArgs args) {
    const PostI& L = args.L[blockIdx.y];
    const int b = blockIdx.x, tid = threadIdx.x, N = L.N, D = L.D, Do = L.Do;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Bump bp{lds};
    float* LA = bp.take(N * N);
    float* LX = bp.take(N * (D + GP));
    float* LG = bp.take(N * (D + GP));
    float* LY = bp.take(N * (Do + GP));
    float* WL = bp.take(64 * (64 + GP));
    float* LV = bp.take(256);                       // small vectors: master, scores, ...
    double* mine = reinterpret_cast<double*>(bp.take(4 * 64 * 2 + 8));
    g_load(LX, D + GP, L.xd + (size_t)b * N * D, N, D);
    float* Sg = L.S + (size_t)b * N * N;
    for (int e = tid; e < N * N; e += GT) LA[e] = Sg[e];
    __syncthreads();
    softmax_rows(LA, N, N, L.inv_temp);
    __syncthreads();
    for (int e = tid; e < N * N; e += GT) Sg[e] = LA[e];
    {   // g = A xd
        const int d = tid % D, grp = tid / D, G = GT / D;
        for (int n = grp; n < N; n += G) {
            float s = 0.f;
            const float* ar = LA + n * N;
            for (int j = 0; j < N; ++j) s = fmaf(ar[j], LX[j * (D + GP) + d], s);
            LG[n * (D + GP) + d] = s;
        }
    }
    stage_w(WL, L.Wa, Do, D);
    __syncthreads();
    g_store(L.g + (size_t)b * N * D, LG, D + GP, N, D);
    mm_nt(LY, Do + GP, LG, D + GP, WL, L.ba, N, D, Do, false);
    __syncthreads();
    stage_w(WL, L.Wb, Do, D);
    __syncthreads();
    mm_nt(LY, Do + GP, LX, D + GP, WL, L.bb, N, D, Do, true);
    __syncthreads();
    g_store(L.y + (size_t)b * N * Do, LY, Do + GP, N, Do);
    bn_block_sums(mine, N, Do, [&](int n, int c, float& u, float& v) { u = LY[n * (Do + GP) + c]; v = u * u; });
    if (L.has_master) {
        float* LM = LV;            // master in [D]
        float* LS = LV + 64;       // node scores / probabilities [N]
        if (tid < D) LM[tid] = L.min[(size_t)b * L.min_bs + tid];
        stage_w(WL, L.WM, Do, D);
        __syncthreads();
        for (int e = tid; e < N * D; e += GT) { const int n = e / D, d = e - n * D; LG[n * (D + GP) + d] = LX[n * (D + GP) + d] * LM[d]; }
        __syncthreads();
        mm_nt(LY, Do + GP, LG, D + GP, WL, L.bM, N, D, Do, false);
        __syncthreads();
        for (int e = tid; e < N * Do; e += GT) { const int n = e / Do, o = e - n * Do; LY[n * (Do + GP) + o] = g_tanh(LY[n * (Do + GP) + o]); }
        __syncthreads();
        g_store(L.tM + (size_t)b * N * Do, LY, Do + GP, N, Do);
        if (tid < N) {
            float s = 0.f;
            for (int o = 0; o < Do; ++o) s = fmaf(LY[tid * (Do + GP) + o], L.aM[o], s);
            LS[tid] = s * L.inv_temp;
        }
        __syncthreads();
        if (tid < 64) {      // softmax over the nodes (N <= 128), one wave
            const float v0 = tid < N ? LS[tid] : -INFINITY, v1 = tid + 64 < N ? LS[tid + 64] : -INFINITY;
            const float mx = wave_max(fmaxf(v0, v1));
            const float e0 = tid < N ? __expf(v0 - mx) : 0.f, e1 = tid + 64 < N ? __expf(v1 - mx) : 0.f;
            const float inv = 1.0f / wave_sum(e0 + e1);
            if (tid < N) LS[tid] = e0 * inv;
            if (tid + 64 < N) LS[tid + 64] = e1 * inv;
        }
        __syncthreads();
        if (tid < N) L.am[(size_t)b * N + tid] = LS[tid];
        float* LGM = LV + 192;     // [D]
        if (tid < D) {
            float s = 0.f;
            for (int n = 0; n < N; ++n) s = fmaf(LS[n], LX[n * (D + GP) + tid], s);
            LGM[tid] = s;
            L.gm[(size_t)b * D + tid] = s;
        }
        __syncthreads();
        if (tid < Do) {
            float s = L.baM[tid] + L.bbM[tid];
            const float* wa = L.WaM + (size_t)tid * D; const float* wb = L.WbM + (size_t)tid * D;
            for (int d = 0; d < D; ++d) s = fmaf(LGM[d], wa[d], fmaf(LM[d], wb[d], s));
            L.mout[(size_t)b * Do + tid] = s;
        }
    }
    const BnK& bn = L.bn;
    if (!bn.training) return;
    rs_finish(bn.acc, bn.ticket, 2 * Do, mine, mine + 2 * Do, [&](const double* tot) { bn_finish_fwd(bn, Do, tot); });
}

__global__ __launch_bounds__(GT) void graph_post_bwd_kernel(const PostArgs args) {
    const PostI& L = args.L[blockIdx.y];
    const int b = blockIdx.x, tid = threadIdx.x, N = L.N, D = L.D, Do = L.Do;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Bump bp{lds};
    float* LA = bp.take(N * N);
    float* LX = bp.take(N * (D + GP));
    float* LG = bp.take(N * (D + GP));          // g, then dg, then dq
    float* LDY = bp.take(N * (Do + GP));        // dy, then tM / dpre
    float* LDX = bp.take(N * (D + GP));
    float* WL = bp.take(64 * (64 + GP));
    float* LV = bp.take(512);
    float* slab = L.slab + (size_t)b * L.slab_bs;
    const BnK& bn = L.bn;
    {   // dy = gamma * rstd * (dz - mean(dz) - xhat * mean(dz * xhat))
        const float* yg = L.y + (size_t)b * N * Do; const float* dzg = L.dz + (size_t)b * N * Do;
        for (int e = tid; e < N * Do; e += GT) {
            const int n = e / Do, o = e - n * Do;
            const float xh = (yg[e] - bn.stats[o]) * bn.stats[Do + o];
            LDY[n * (Do + GP) + o] = bn.stats[2 * Do + o] * (dzg[e] - bn.bstats[o] - xh * bn.bstats[Do + o]);
        }
    }
    g_load(LX, D + GP, L.xd + (size_t)b * N * D, N, D);
    g_load(LG, D + GP, L.g + (size_t)b * N * D, N, D);
    {
        const float* Sg = L.S + (size_t)b * N * N;
        for (int e = tid; e < N * N; e += GT) LA[e] = Sg[e];
    }
    __syncthreads();
    if (tid < Do) {
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += LDY[n * (Do + GP) + tid];
        slab[L.o_ba + tid] = s; slab[L.o_bb + tid] = s;
    }
    mm_tn_out(slab + L.o_Wa, LDY, Do + GP, LG, D + GP, N, D, Do, nullptr);
    mm_tn_out(slab + L.o_Wb, LDY, Do + GP, LX, D + GP, N, D, Do, nullptr);
    stage_w(WL, L.Wa, Do, D);
    __syncthreads();
    mm_nn(LG, D + GP, LDY, Do + GP, WL, N, D, Do, false);          // dg (g is no longer needed)
    __syncthreads();
    stage_w(WL, L.Wb, Do, D);
    __syncthreads();
    mm_nn(LDX, D + GP, LDY, Do + GP, WL, N, D, Do, false);         // d xd, first part
    __syncthreads();
    {   // dS[i][j] = A[i][j] * (dA[i][j] - sum_j' dA[i][j'] A[i][j']) / temp,  dA[i][j] = <dg[i], xd[j]>        one wave per row
        const int lane = tid & 63, wave = tid >> 6;
        float* dSg = L.dS + (size_t)b * N * N;
        for (int i = wave; i < N; i += GT / 64) {
            float da[2] = {0.f, 0.f}, av[2] = {0.f, 0.f};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int j = lane + 64 * h;
                if (j < N) {
                    float s = 0.f;
                    for (int d = 0; d < D; d += 4) {
                        const f32x4 p = *reinterpret_cast<const f32x4*>(LG + i * (D + GP) + d), q = *reinterpret_cast<const f32x4*>(LX + j * (D + GP) + d);
                        s = fmaf(p[0], q[0], s); s = fmaf(p[1], q[1], s); s = fmaf(p[2], q[2], s); s = fmaf(p[3], q[3], s);
                    }
                    da[h] = s; av[h] = LA[i * N + j];
                }
            }
            const float dot = wave_sum(da[0] * av[0] + da[1] * av[1]);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int j = lane + 64 * h;
                if (j < N) dSg[i * N + j] = av[h] * (da[h] - dot) * L.inv_temp;
            }
        }
    }
    {   // d xd[j][d] += sum_i A[i][j] dg[i][d]
        const int d = tid % D, grp = tid / D, G = GT / D;
        for (int j = grp; j < N; j += G) {
            float s = LDX[j * (D + GP) + d];
            for (int i = 0; i < N; ++i) s = fmaf(LA[i * N + j], LG[i * (D + GP) + d], s);
            LDX[j * (D + GP) + d] = s;
        }
    }
    __syncthreads();
    if (L.has_master) {
        float* LM = LV;             // master in [D]
        float* LDM = LV + 64;       // d mout [Do]
        float* LGMd = LV + 128;     // d gm [D]
        float* LAM = LV + 192;      // am [N], later dsc [N]
        float* LDMI = LV + 320;     // d min [D]
        float* LGM = LV + 384;      // gm [D]
        if (tid < D) { LM[tid] = L.min[(size_t)b * L.min_bs + tid]; LGM[tid] = L.gm[(size_t)b * D + tid]; }
        if (tid < Do) LDM[tid] = (L.d_mout ? L.d_mout[(size_t)b * Do + tid] : 0.f) + (L.d_mout2 ? L.d_mout2[(size_t)b * Do + tid] : 0.f);
        if (tid < N) LAM[tid] = L.am[(size_t)b * N + tid];
        g_load(LDY, Do + GP, L.tM + (size_t)b * N * Do, N, Do);      // tanh values
        __syncthreads();
        for (int e = tid; e < Do * D; e += GT) {
            const int o = e / D, d = e - o * D;
            slab[L.o_WaM + e] = LDM[o] * LGM[d]; slab[L.o_WbM + e] = LDM[o] * LM[d];
        }
        if (tid < Do) { slab[L.o_baM + tid] = LDM[tid]; slab[L.o_bbM + tid] = LDM[tid]; }
        if (tid < D) {
            float s0 = 0.f, s1 = 0.f;
            for (int o = 0; o < Do; ++o) { s0 = fmaf(LDM[o], L.WaM[(size_t)o * D + tid], s0); s1 = fmaf(LDM[o], L.WbM[(size_t)o * D + tid], s1); }
            LGMd[tid] = s0; LDMI[tid] = s1;
        }
        __syncthreads();
        float* LDAM = LV + 448;     // d am [N] (N <= 64 + ... keep inside LV: 448 + N <= 512 -> N <= 64: heterogeneous layers have <= 64 nodes)
        if (tid < N) {
            float s = 0.f;
            for (int d = 0; d < D; ++d) s = fmaf(LGMd[d], LX[tid * (D + GP) + d], s);
            LDAM[tid] = s;
        }
        for (int e = tid; e < N * D; e += GT) { const int n = e / D, d = e - n * D; LDX[n * (D + GP) + d] += LAM[n] * LGMd[d]; }
        __syncthreads();
        if (tid < 64) {
            const float p = tid < N ? LAM[tid] : 0.f, dv = tid < N ? LDAM[tid] : 0.f;
            const float dot = wave_sum(p * dv);
            if (tid < N) LDAM[tid] = p * (dv - dot) * L.inv_temp;      // d score
        }
        __syncthreads();
        if (tid < Do) {      // d aM[o] = sum_n dsc[n] t[n][o]
            float s = 0.f;
            for (int n = 0; n < N; ++n) s = fmaf(LDAM[n], LDY[n * (Do + GP) + tid], s);
            slab[L.o_aM + tid] = s;
        }
        __syncthreads();
        for (int e = tid; e < N * Do; e += GT) {
            const int n = e / Do, o = e - n * Do;
            const float t = LDY[n * (Do + GP) + o];
            LDY[n * (Do + GP) + o] = LDAM[n] * L.aM[o] * (1.0f - t * t);      // d pre
        }
        stage_w(WL, L.WM, Do, D);
        __syncthreads();
        if (tid < Do) {
            float s = 0.f;
            for (int n = 0; n < N; ++n) s += LDY[n * (Do + GP) + tid];
            slab[L.o_bM + tid] = s;
        }
        mm_tn_out(slab + L.o_WM, LDY, Do + GP, LX, D + GP, N, D, Do, LM);      // d WM[o][d] = m[d] sum_n dpre[n][o] xd[n][d]
        mm_nn(LG, D + GP, LDY, Do + GP, WL, N, D, Do, false);                 // dq
        __syncthreads();
        if (tid < D) {
            float s = LDMI[tid];
            for (int n = 0; n < N; ++n) s = fmaf(LG[n * (D + GP) + tid], LX[n * (D + GP) + tid], s);
            L.d_min[(size_t)b * D + tid] = s;
        }
        for (int e = tid; e < N * D; e += GT) { const int n = e / D, d = e - n * D; LDX[n * (D + GP) + d] += LG[n * (D + GP) + d] * LM[d]; }
        __syncthreads();
    }
    g_store(L.dxd + (size_t)b * N * D, LDX, D + GP, N, D);
}

// ---- BatchNorm + SELU, GraphPool, proj_type, input dropout: what sits between two attention layers -------------------------------------------
typedef SclGraphPoolUnit PoolU;
typedef SclGraphPre PreI;
struct PreArgs { PreI I[2]; int shared_pool; };      // shared_pool: both instances read the SAME pooled nodes (layers ST11 / ST21 share pool_S / pool_T)

// GraphPool of one unit: LH [n_in][Dp + GP] holds h; writes scores, indices and the gated, gathered rows into LP [K][Dp + GP]
__device__ __forceinline__ void pool_fwd(const PoolU& u, int b, int Dp, float* LH, float* LP, float* LS, int* LI, bool store) {
    const int tid = threadIdx.x, n_in = u.n_in;
    if (tid < n_in) {
        float s = u.pb[0];
        for (int d = 0; d < Dp; ++d) {
            float z = LH[tid * (Dp + GP) + d];
            if (u.pool_p > 0.f) z *= dropout_scale(u.pool_seed, (uint64_t)(((size_t)b * n_in + tid) * Dp + d), u.pool_p);
            s = fmaf(z, u.pw[d], s);
        }
        LS[tid] = 1.0f / (1.0f + __expf(-s));
    }
    __syncthreads();
    if (tid < n_in) {      // rank by descending score (ties: lower index first) = position in torch.topk's sorted output
        const float s = LS[tid];
        int rank = 0;
        for (int m = 0; m < n_in; ++m) { const float t = LS[m]; rank += (t > s || (t == s && m < tid)) ? 1 : 0; }
        if (rank < u.K) LI[rank] = tid;
        if (store) u.sc[(size_t)b * n_in + tid] = s;
    }
    __syncthreads();
    if (store && tid < u.K) u.idx[(size_t)b * u.K + tid] = LI[tid];
    for (int e = tid; e < u.K * Dp; e += GT) {
        const int r = e / Dp, d = e - r * Dp, n = LI[r];
        LP[r * (Dp + GP) + d] = LH[n * (Dp + GP) + d] * LS[n];
    }
    __syncthreads();
}

__global__ __launch_bounds__(GT) void graph_pre_fwd_kernel(const PreArgs args) {
    const PreI& I = args.I[blockIdx.y];
    const int b = blockIdx.x, tid = threadIdx.x, Dp = I.Dp, N = I.N;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Bump bp{lds};
    float* LH = bp.take(96 * (Dp + GP));
    float* LP = bp.take(N * (Dp + GP));         // pooled rows of both units, concatenated
    float* LXo = bp.take(N * (Dp + GP));
    float* WL = bp.take(64 * (64 + GP));
    float* LS = bp.take(128);
    int* LI = reinterpret_cast<int*>(bp.take(128));
    const bool store = I.store_common != 0;
    for (int ui = 0; ui < 2; ++ui) {
        const PoolU& u = I.u[ui];
        const float* ys = u.ysrc + ((size_t)b * u.src_n + u.row0) * Dp;
        for (int e = tid; e < u.n_in * Dp; e += GT) {
            const int n = e / Dp, d = e - n * Dp;
            const float v = selu_f((ys[e] - u.stats[d]) * u.stats[2 * Dp + d] + u.stats[3 * Dp + d]);
            LH[n * (Dp + GP) + d] = v;
            if (store) u.h[(size_t)b * u.n_in * Dp + e] = v;
        }
        __syncthreads();
        pool_fwd(u, b, Dp, LH, LP + u.row_out * (Dp + GP), LS, LI, store);
        if (store) g_store(u.pooled + (size_t)b * u.K * Dp, LP + u.row_out * (Dp + GP), Dp + GP, u.K, Dp);
        stage_w(WL, u.Wt, Dp, Dp);
        __syncthreads();
        mm_nt(LXo + u.row_out * (Dp + GP), Dp + GP, LP + u.row_out * (Dp + GP), Dp + GP, WL, u.bt, u.K, Dp, Dp, false);
        __syncthreads();
    }
    float* xg = I.xd + (size_t)b * N * Dp;
    for (int e = tid; e < N * Dp; e += GT) {
        const int n = e / Dp, d = e - n * Dp;
        float v = LXo[n * (Dp + GP) + d];
        if (I.in_p > 0.f) v *= dropout_scale(I.in_seed, (uint64_t)((size_t)b * N * Dp + e), I.in_p);
        xg[e] = v;
    }
}

// backward of the above: one block per utterance and per group of instances that share their pooled nodes — grid (B, shared_pool ? 1 : 2)
__global__ __launch_bounds__(GT) void graph_pre_bwd_kernel(const PreArgs args) {
    const int b = blockIdx.x, tid = threadIdx.x;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int ninst = args.shared_pool ? 2 : 1;
    const int Dp = args.I[0].Dp, N = args.I[0].N;
    Bump bp{lds};
    float* LDXo = bp.take(N * (Dp + GP));        // d x (after the dropout mask)
    float* LP = bp.take(N * (Dp + GP));          // pooled rows (forward values)
    float* LDP = bp.take(N * (Dp + GP));         // d pooled, summed over the instances
    float* LH = bp.take(96 * (Dp + GP));
    float* WL = bp.take(64 * (64 + GP));
    float* LSC = bp.take(128);                   // pool scores, then the gate of the gathered path
    float* LDL = bp.take(128);                   // d logit of the pool score
    int* LR = reinterpret_cast<int*>(bp.take(128));      // output row of a node (-1: dropped)
    double* mine = reinterpret_cast<double*>(bp.take(4 * 64 * 2 + 8));
    for (int e = tid; e < N * (Dp + GP); e += GT) LDP[e] = 0.f;
    for (int ii = 0; ii < ninst; ++ii) {
        const PreI& I = args.I[args.shared_pool ? ii : blockIdx.y];
        float* slab = I.slab + (size_t)b * I.slab_bs;
        const float* da = I.dxd_a + (size_t)b * N * Dp; const float* db = I.dxd_b + (size_t)b * N * Dp;
        __syncthreads();
        for (int e = tid; e < N * Dp; e += GT) {
            const int n = e / Dp, d = e - n * Dp;
            float v = da[e] + db[e];
            if (I.in_p > 0.f) v *= dropout_scale(I.in_seed, (uint64_t)((size_t)b * N * Dp + e), I.in_p);
            LDXo[n * (Dp + GP) + d] = v;
        }
        for (int ui = 0; ui < 2; ++ui) {
            const PoolU& u = I.u[ui];
            g_load(LP + u.row_out * (Dp + GP), Dp + GP, u.pooled + (size_t)b * u.K * Dp, u.K, Dp);
        }
        __syncthreads();
        for (int ui = 0; ui < 2; ++ui) {
            const PoolU& u = I.u[ui];
            const float* dxr = LDXo + u.row_out * (Dp + GP);
            if (tid < Dp) {
                float s = 0.f;
                for (int n = 0; n < u.K; ++n) s += dxr[n * (Dp + GP) + tid];
                slab[u.o_bt + tid] = s;
            }
            mm_tn_out(slab + u.o_Wt, dxr, Dp + GP, LP + u.row_out * (Dp + GP), Dp + GP, u.K, Dp, Dp, nullptr);
            stage_w(WL, u.Wt, Dp, Dp);
            __syncthreads();
            mm_nn(LDP + u.row_out * (Dp + GP), Dp + GP, dxr, Dp + GP, WL, u.K, Dp, Dp, true);
            __syncthreads();
        }
    }
    // pool + SELU backward, once per unit (the instances' pooled-node gradients are summed in LDP)
    const PreI& I0 = args.I[args.shared_pool ? 0 : blockIdx.y];
    float* slab0 = I0.slab + (size_t)b * I0.slab_bs;
    for (int ui = 0; ui < 2; ++ui) {
        const PoolU& u = I0.u[ui];
        const int n_in = u.n_in, K = u.K;
        float* LDPu = LDP + u.row_out * (Dp + GP);
        if (u.d_res) {
            const float* dr = u.d_res + (size_t)b * K * Dp;
            for (int e = tid; e < K * Dp; e += GT) { const int r = e / Dp, d = e - r * Dp; LDPu[r * (Dp + GP) + d] += dr[e]; }
        }
        g_load(LH, Dp + GP, u.h + (size_t)b * n_in * Dp, n_in, Dp);
        __syncthreads();
        const int* idx = u.idx + (size_t)b * K;
        if (tid < n_in) {      // d(h * s)[n] = dpooled[r] where idx[r] == n, else 0
            int r = -1;
            for (int q = 0; q < K; ++q) r = idx[q] == tid ? q : r;
            float ds = 0.f;
            if (r >= 0)
                for (int d = 0; d < Dp; ++d) ds = fmaf(LDPu[r * (Dp + GP) + d], LH[tid * (Dp + GP) + d], ds);
            const float s = u.sc[(size_t)b * n_in + tid];
            LDL[tid] = ds * s * (1.0f - s);
            LSC[tid] = r >= 0 ? s : 0.f;
            LR[tid] = r;
        }
        __syncthreads();
        if (tid < Dp) {      // d pw[d] = sum_n dlogit[n] Z[n][d]
            float s = 0.f;
            for (int n = 0; n < n_in; ++n) {
                float z = LH[n * (Dp + GP) + tid];
                if (u.pool_p > 0.f) z *= dropout_scale(u.pool_seed, (uint64_t)(((size_t)b * n_in + n) * Dp + tid), u.pool_p);
                s = fmaf(LDL[n], z, s);
            }
            slab0[u.o_pw + tid] = s;
        }
        if (tid == 0) { float s = 0.f; for (int n = 0; n < n_in; ++n) s += LDL[n]; slab0[u.o_pb] = s; }
        float* dzg = u.dz + ((size_t)b * u.src_n + u.row0) * Dp;
        __syncthreads();
        for (int e = tid; e < n_in * Dp; e += GT) {      // dh = gate * dpooled[row] + dlogit * pw * poolmask;  dz = dh * selu'(h)
            const int n = e / Dp, d = e - n * Dp;
            const int r = LR[n];
            float dh = r >= 0 ? LSC[n] * LDPu[r * (Dp + GP) + d] : 0.f;
            float pm = u.pw[d];
            if (u.pool_p > 0.f) pm *= dropout_scale(u.pool_seed, (uint64_t)(((size_t)b * n_in + n) * Dp + d), u.pool_p);
            dh = fmaf(LDL[n], pm, dh);
            const float dz = dh * selu_grad_from_y(LH[n * (Dp + GP) + d]);
            dzg[e] = dz;
            LH[n * (Dp + GP) + d] = dz;
        }
        __syncthreads();
        // BatchNorm backward sums of the source layer over this unit's rows: sum dz, sum dz * xhat
        const float* ys = u.ysrc + ((size_t)b * u.src_n + u.row0) * Dp;
        const BnK& bn = u.bn;
        bn_block_sums(mine, n_in, Dp, [&](int n, int c, float& a, float& v) {
            a = LH[n * (Dp + GP) + c];
            v = a * ((ys[n * Dp + c] - bn.stats[c]) * bn.stats[Dp + c]);
        }, I0.same_bn && ui == 1);
        if (!I0.same_bn || ui == 1)
            rs_finish(bn.acc, bn.ticket, 2 * Dp, mine, mine + 2 * Dp, [&](const double* tot) { bn_finish_bwd(bn, Dp, tot); });
        __syncthreads();
    }
}

// ---- first layers: d e = (d xd parts) * keep-mask ----------------------------------------------------------------------------------------
struct DropBwdArgs { const float* da[2]; const float* db[2]; float* de[2]; long long n[2]; unsigned seed[2]; float p; };
__global__ __launch_bounds__(GT) void graph_drop_bwd_kernel(const DropBwdArgs a) {
    const int i = blockIdx.y;
    for (long long e = (long long)blockIdx.x * GT + threadIdx.x; e < a.n[i]; e += (long long)gridDim.x * GT) {
        float v = a.da[i][e] + a.db[i][e];
        if (a.p > 0.f) v *= dropout_scale(a.seed[i], (uint64_t)e, a.p);
        a.de[i][e] = v;
    }
}

// ---- read-out ----------------------------------------------------------------------------------------------------------------------------
typedef SclGraphFinalBranch FinalBr;
typedef SclGraphFinal FinalArgs;

// forward values of one utterance into LDS: T_b, S_b [2][K][D], m_b [2][D] after residual add and drop_way, aug kept for the backward
__device__ __forceinline__ void final_branches(const FinalArgs& a, int b, float* LT, float* LSn, float* LMm, float* LAUG) {
    const int tid = threadIdx.x, KT = a.KT, KS = a.KS, D = a.D, N2 = KT + KS;
    for (int br = 0; br < 2; ++br) {
        const FinalBr& B_ = a.br[br];
        const float* y = B_.y2 + (size_t)b * N2 * D;
        for (int e = tid; e < N2 * D; e += GT) {
            const int n = e / D, d = e - n * D;
            const float aug = selu_f((y[e] - B_.stats2[d]) * B_.stats2[2 * D + d] + B_.stats2[3 * D + d]);
            LAUG[br * N2 * D + e] = aug;
            if (n < KT) {
                float v = B_.Tp[(size_t)b * KT * D + e] + aug;
                if (a.way_p > 0.f) v *= dropout_scale(B_.way_seed[0], (uint64_t)((size_t)b * KT * D + e), a.way_p);
                LT[br * KT * D + e] = v;
            } else {
                const int e2 = e - KT * D;
                float v = B_.Sp[(size_t)b * KS * D + e2] + aug;
                if (a.way_p > 0.f) v *= dropout_scale(B_.way_seed[1], (uint64_t)((size_t)b * KS * D + e2), a.way_p);
                LSn[br * KS * D + e2] = v;
            }
        }
        if (tid < D) {
            float v = B_.m1[(size_t)b * D + tid] + B_.m2[(size_t)b * D + tid];
            if (a.way_p > 0.f) v *= dropout_scale(B_.way_seed[2], (uint64_t)((size_t)b * D + tid), a.way_p);
            LMm[br * D + tid] = v;
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(GT) void graph_final_fwd_kernel(const FinalArgs a) {
    const int b = blockIdx.x, tid = threadIdx.x, KT = a.KT, KS = a.KS, D = a.D, N2 = KT + KS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Bump bp{lds};
    float* LT = bp.take(2 * KT * D); float* LSn = bp.take(2 * KS * D); float* LMm = bp.take(2 * D); float* LAUG = bp.take(2 * N2 * D);
    float* LH = bp.take(5 * D);
    final_branches(a, b, LT, LSn, LMm, LAUG);
    if (tid < D) {
        float mx = 0.f, sm = 0.f;
        for (int n = 0; n < KT; ++n) { const float v = fmaxf(LT[n * D + tid], LT[KT * D + n * D + tid]); mx = fmaxf(mx, fabsf(v)); sm += v; }
        LH[tid] = mx; LH[D + tid] = sm / KT;
        mx = 0.f; sm = 0.f;
        for (int n = 0; n < KS; ++n) { const float v = fmaxf(LSn[n * D + tid], LSn[KS * D + n * D + tid]); mx = fmaxf(mx, fabsf(v)); sm += v; }
        LH[2 * D + tid] = mx; LH[3 * D + tid] = sm / KS;
        LH[4 * D + tid] = fmaxf(LMm[tid], LMm[D + tid]);
    }
    __syncthreads();
    if (tid < 5 * D) {
        float v = LH[tid];
        if (a.drop_p > 0.f) v *= dropout_scale(a.drop_seed, (uint64_t)((size_t)b * 5 * D + tid), a.drop_p);
        LH[tid] = v;
        a.hidden[(size_t)b * 5 * D + tid] = v;
    }
    __syncthreads();
    if (tid < a.NC) {
        float s = a.bout[tid];
        for (int k = 0; k < 5 * D; ++k) s = fmaf(LH[k], a.Wout[(size_t)tid * 5 * D + k], s);
        a.logits[(size_t)b * a.NC + tid] = s;
    }
}

// gradient split of torch.max(a, b): to the larger, halves at ties (torch.maximum's backward)
__device__ __forceinline__ void max_split(float x0, float x1, float g, float& g0, float& g1) {
    g0 = x0 > x1 ? g : (x0 == x1 ? 0.5f * g : 0.f);
    g1 = x1 > x0 ? g : (x0 == x1 ? 0.5f * g : 0.f);
}

__global__ __launch_bounds__(GT) void graph_final_bwd_kernel(const FinalArgs a) {
    const int b = blockIdx.x, tid = threadIdx.x, KT = a.KT, KS = a.KS, D = a.D, N2 = KT + KS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    Bump bp{lds};
    float* LT = bp.take(2 * KT * D); float* LSn = bp.take(2 * KS * D); float* LMm = bp.take(2 * D); float* LAUG = bp.take(2 * N2 * D);
    float* LH = bp.take(5 * D); float* LDH = bp.take(5 * D); float* LDL = bp.take(8);
    int* LIX = reinterpret_cast<int*>(bp.take(2 * D));
    float* LDZ = bp.take(N2 * (D + GP));
    double* mine = reinterpret_cast<double*>(bp.take(4 * 64 * 2 + 8));
    final_branches(a, b, LT, LSn, LMm, LAUG);
    float* slab = a.slab + (size_t)b * a.slab_bs;
    if (tid < a.NC) LDL[tid] = a.d_logits ? a.d_logits[(size_t)b * a.NC + tid] : 0.f;
    if (tid < D) {      // forward read-out again: values and the arg of the |max|
        float mx = -1.f, sm = 0.f; int ix = 0;
        for (int n = 0; n < KT; ++n) { const float v = fmaxf(LT[n * D + tid], LT[KT * D + n * D + tid]); if (fabsf(v) > mx) { mx = fabsf(v); ix = n; } sm += v; }
        LH[tid] = mx; LH[D + tid] = sm / KT; LIX[tid] = ix;
        mx = -1.f; sm = 0.f; ix = 0;
        for (int n = 0; n < KS; ++n) { const float v = fmaxf(LSn[n * D + tid], LSn[KS * D + n * D + tid]); if (fabsf(v) > mx) { mx = fabsf(v); ix = n; } sm += v; }
        LH[2 * D + tid] = mx; LH[3 * D + tid] = sm / KS; LIX[D + tid] = ix;
        LH[4 * D + tid] = fmaxf(LMm[tid], LMm[D + tid]);
    }
    __syncthreads();
    if (tid < 5 * D) {
        const float sc = a.drop_p > 0.f ? dropout_scale(a.drop_seed, (uint64_t)((size_t)b * 5 * D + tid), a.drop_p) : 1.f;
        const float lhd = LH[tid] * sc;
        float g = a.d_hidden ? a.d_hidden[(size_t)b * 5 * D + tid] : 0.f;
        for (int c = 0; c < a.NC; ++c) {
            g = fmaf(LDL[c], a.Wout[(size_t)c * 5 * D + tid], g);
            slab[a.o_Wout + c * 5 * D + tid] = LDL[c] * lhd;
        }
        LDH[tid] = g * sc;
    }
    if (tid < a.NC) slab[a.o_bout + tid] = LDL[tid];
    __syncthreads();
    for (int br = 0; br < 2; ++br) {
        const FinalBr& B_ = a.br[br];
        for (int e = tid; e < N2 * D; e += GT) {
            const int n = e / D, d = e - n * D;
            float g;      // d of the branch-max'ed node value
            float x0, x1;
            unsigned seed; uint64_t idx;
            if (n < KT) {
                x0 = LT[e]; x1 = LT[KT * D + e];
                const float v = fmaxf(x0, x1);
                g = LDH[D + d] / KT + (n == LIX[d] ? (v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f)) * LDH[d] : 0.f);
                seed = B_.way_seed[0]; idx = (uint64_t)((size_t)b * KT * D + e);
            } else {
                const int e2 = e - KT * D, n2 = n - KT;
                x0 = LSn[e2]; x1 = LSn[KS * D + e2];
                const float v = fmaxf(x0, x1);
                g = LDH[3 * D + d] / KS + (n2 == LIX[D + d] ? (v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f)) * LDH[2 * D + d] : 0.f);
                seed = B_.way_seed[1]; idx = (uint64_t)((size_t)b * KS * D + e2);
            }
            float g0, g1;
            max_split(x0, x1, g, g0, g1);
            float gb = br == 0 ? g0 : g1;
            if (a.way_p > 0.f) gb *= dropout_scale(seed, idx, a.way_p);
            if (n < KT) B_.dTp[(size_t)b * KT * D + e] = gb; else B_.dSp[(size_t)b * KS * D + (e - KT * D)] = gb;
            const float dz = gb * selu_grad_from_y(LAUG[br * N2 * D + e]);
            B_.dz2[(size_t)b * N2 * D + e] = dz;
            LDZ[n * (D + GP) + d] = dz;
        }
        if (tid < D) {
            float g0, g1;
            max_split(LMm[tid], LMm[D + tid], LDH[4 * D + tid], g0, g1);
            float gb = br == 0 ? g0 : g1;
            if (a.way_p > 0.f) gb *= dropout_scale(B_.way_seed[2], (uint64_t)((size_t)b * D + tid), a.way_p);
            B_.dm1[(size_t)b * D + tid] = gb; B_.dm2[(size_t)b * D + tid] = gb;
        }
        __syncthreads();
        const float* y = B_.y2 + (size_t)b * N2 * D;
        const BnK& bn = B_.bn;
        bn_block_sums(mine, N2, D, [&](int n, int c, float& u, float& v) {
            u = LDZ[n * (D + GP) + c];
            v = u * ((y[n * D + c] - bn.stats[c]) * bn.stats[D + c]);
        });
        rs_finish(bn.acc, bn.ticket, 2 * D, mine, mine + 2 * D, [&](const double* tot) { bn_finish_bwd(bn, D, tot); });
        __syncthreads();
    }
}

// ---- parameter gradients: dst[i] += sum over parts of src[part * stride + i], parts in a fixed order ---------------------------------------
// A block owns 32 consecutive elements of one job and splits the parts eight ways (the pairwise-score partials of a 66-node layer are
// 1152 parts x 4352 columns: one thread walking them alone took a millisecond); the eight partial sums are combined in lane order.
struct ReduceJobs { SclGraphReduceJob j[SCL_GRAPH_MAX_REDUCE_JOBS]; int chunk0[SCL_GRAPH_MAX_REDUCE_JOBS + 1]; int njobs; };
__global__ __launch_bounds__(GT) void graph_reduce_kernel(const ReduceJobs jobs) {
    __shared__ float red[8][32];
    int ji = 0;
    while (ji + 1 < jobs.njobs && (int)blockIdx.x >= jobs.chunk0[ji + 1]) ++ji;
    const SclGraphReduceJob& q = jobs.j[ji];
    const int el = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int i = ((int)blockIdx.x - jobs.chunk0[ji]) * 32 + el;
    float s = 0.f;
    if (i < q.n) {
        const float* p = q.src + i;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int k = pl;
        for (; k + 24 < q.nparts; k += 32) {
            s0 += p[(size_t)k * q.stride]; s1 += p[(size_t)(k + 8) * q.stride]; s2 += p[(size_t)(k + 16) * q.stride]; s3 += p[(size_t)(k + 24) * q.stride];
        }
        for (; k < q.nparts; k += 8) s0 += p[(size_t)k * q.stride];
        s = (s0 + s1) + (s2 + s3);
    }
    red[pl][el] = s;
    __syncthreads();
    if (pl == 0 && i < q.n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][el];
        q.dst[i] += t;
    }
}

template <class K, class A>
int g_launch(K kern, dim3 grid, size_t lds, hipStream_t s, const A& args, const char* what) {
    if (lds > 160 * 1024) { scl_set_error("%s: %zu bytes of LDS (graph too large: the fused graph module takes up to 80 nodes)", what, lds); return SCL_EINVAL; }
    if (lds > 65536) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, grid, dim3(GT), lds, s, args);
    return scl_check_launch(what);
}
bool layer_ok(const SclGraphLayer& L) {
    return L.N > 0 && L.N <= 96 && (L.D == 32 || L.D == 64) && (L.Do == 32 || L.Do == 64) && (!L.has_master || L.N <= 64);
}
size_t layer_lds(const SclGraphLayer* L, int n, bool bwd) {
    size_t m = 0;
    for (int i = 0; i < n; ++i) {
        const size_t N = L[i].N, D = L[i].D, Do = L[i].Do;
        const size_t f = N * N + (bwd ? 3 : 2) * N * (D + GP) + N * (Do + GP) + 64 * (64 + GP) + 512 + 2 * (4 * 64 * 2 + 8) + 64;
        m = f > m ? f : m;
    }
    return m * 4;
}

}  // namespace

extern "C" int scl_graph_post_fwd(const SclGraphLayer* layers, int nlayers, int B, void* stream) {
    SCL_REQUIRE(layers && (nlayers == 1 || nlayers == 2) && B > 0, "graph_post_fwd: 1 or 2 layers");
    PostArgs a;
    for (int i = 0; i < nlayers; ++i) { SCL_REQUIRE(layer_ok(layers[i]), "graph_post_fwd: layer %d: N <= 96 (64 with a master node), D / Do in {32, 64}", i); a.L[i] = layers[i]; }
    return g_launch(graph_post_fwd_kernel, dim3(B, nlayers), layer_lds(layers, nlayers, false), (hipStream_t)stream, a, "graph_post_fwd");
}
extern "C" int scl_graph_post_bwd(const SclGraphLayer* layers, int nlayers, int B, void* stream) {
    SCL_REQUIRE(layers && (nlayers == 1 || nlayers == 2) && B > 0, "graph_post_bwd: 1 or 2 layers");
    PostArgs a;
    for (int i = 0; i < nlayers; ++i) { SCL_REQUIRE(layer_ok(layers[i]), "graph_post_bwd: layer %d: N <= 96 (64 with a master node), D / Do in {32, 64}", i); a.L[i] = layers[i]; }
    return g_launch(graph_post_bwd_kernel, dim3(B, nlayers), layer_lds(layers, nlayers, true), (hipStream_t)stream, a, "graph_post_bwd");
}
static bool pre_ok(const SclGraphPre* I) {
    for (int i = 0; i < 2; ++i) {
        if (!(I[i].Dp == 32 || I[i].Dp == 64) || I[i].N <= 0 || I[i].N > 96 || I[i].Dp != I[0].Dp || I[i].N != I[0].N) return false;
        for (int u = 0; u < 2; ++u) if (I[i].u[u].n_in <= 0 || I[i].u[u].n_in > 96 || I[i].u[u].K <= 0 || I[i].u[u].K > I[i].u[u].n_in) return false;
        if (I[i].u[0].K + I[i].u[1].K != I[i].N || I[i].u[0].row_out != 0 || I[i].u[1].row_out != I[i].u[0].K) return false;
    }
    return true;
}
extern "C" int scl_graph_pre_fwd(const SclGraphPre* inst, int shared_pool, int B, void* stream) {
    SCL_REQUIRE(inst && B > 0 && pre_ok(inst), "graph_pre_fwd: two instances of equal shape, <= 96 nodes per unit, width 32 or 64");
    PreArgs a; a.I[0] = inst[0]; a.I[1] = inst[1]; a.shared_pool = shared_pool;
    const size_t Dp = inst[0].Dp, N = inst[0].N;
    const size_t lds = (96 * (Dp + GP) + 2 * N * (Dp + GP) + 64 * (64 + GP) + 256 + 64) * 4;
    return g_launch(graph_pre_fwd_kernel, dim3(B, 2), lds, (hipStream_t)stream, a, "graph_pre_fwd");
}
extern "C" int scl_graph_pre_bwd(const SclGraphPre* inst, int shared_pool, int B, void* stream) {
    SCL_REQUIRE(inst && B > 0 && pre_ok(inst), "graph_pre_bwd: two instances of equal shape, <= 96 nodes per unit, width 32 or 64");
    PreArgs a; a.I[0] = inst[0]; a.I[1] = inst[1]; a.shared_pool = shared_pool;
    const size_t Dp = inst[0].Dp, N = inst[0].N;
    const size_t lds = (96 * (Dp + GP) + 3 * N * (Dp + GP) + 64 * (64 + GP) + 3 * 128 + 2 * (4 * 64 * 2 + 8) + 64) * 4;
    return g_launch(graph_pre_bwd_kernel, dim3(B, shared_pool ? 1 : 2), lds, (hipStream_t)stream, a, "graph_pre_bwd");
}
extern "C" int scl_graph_drop(const float* x0, float* y0, int64_t n0, uint32_t seed0, const float* x1, float* y1, int64_t n1, uint32_t seed1, float p, void* stream) {
    SCL_REQUIRE(x0 && y0 && x1 && y1 && n0 > 0 && n1 > 0 && p >= 0.f && p < 1.f, "graph_drop: bad arguments");
    DropArgs a; a.x[0] = x0; a.x[1] = x1; a.y[0] = y0; a.y[1] = y1; a.n[0] = n0; a.n[1] = n1; a.seed[0] = seed0; a.seed[1] = seed1; a.p = p;
    const long long nm = n0 > n1 ? n0 : n1;
    hipLaunchKernelGGL(graph_drop_kernel, dim3((unsigned)((nm + 1023) / 1024), 2), dim3(GT), 0, (hipStream_t)stream, a);
    return scl_check_launch("graph_drop");
}
extern "C" int scl_graph_drop_bwd(const float* da0, const float* db0, float* de0, int64_t n0, uint32_t seed0, const float* da1, const float* db1, float* de1,
                                  int64_t n1, uint32_t seed1, float p, void* stream) {
    SCL_REQUIRE(da0 && db0 && de0 && da1 && db1 && de1 && n0 > 0 && n1 > 0 && p >= 0.f && p < 1.f, "graph_drop_bwd: bad arguments");
    DropBwdArgs a; a.da[0] = da0; a.da[1] = da1; a.db[0] = db0; a.db[1] = db1; a.de[0] = de0; a.de[1] = de1; a.n[0] = n0; a.n[1] = n1;
    a.seed[0] = seed0; a.seed[1] = seed1; a.p = p;
    const long long nm = n0 > n1 ? n0 : n1;
    hipLaunchKernelGGL(graph_drop_bwd_kernel, dim3((unsigned)((nm + 1023) / 1024), 2), dim3(GT), 0, (hipStream_t)stream, a);
    return scl_check_launch("graph_drop_bwd");
}
static size_t final_lds(const SclGraphFinal& f) {
    const size_t N2 = f.KT + f.KS, D = f.D;
    return (2 * N2 * D + 2 * D + 2 * N2 * D + 10 * D + 8 + 2 * D + N2 * (D + GP) + 2 * (4 * 64 * 2 + 8) + 64) * 4;
}
extern "C" int scl_graph_final_fwd(const SclGraphFinal* f, int B, void* stream) {
    SCL_REQUIRE(f && B > 0 && f->D > 0 && 5 * f->D <= GT && f->NC > 0 && f->NC <= 8 && f->KT > 0 && f->KS > 0, "graph_final_fwd: 5 D <= 256, NC <= 8");
    return g_launch(graph_final_fwd_kernel, dim3(B), final_lds(*f), (hipStream_t)stream, *f, "graph_final_fwd");
}
extern "C" int scl_graph_final_bwd(const SclGraphFinal* f, int B, void* stream) {
    SCL_REQUIRE(f && B > 0 && 5 * f->D <= GT && f->NC > 0 && f->NC <= 8 && f->KT > 0 && f->KS > 0, "graph_final_bwd: 5 D <= 256, NC <= 8");
    return g_launch(graph_final_bwd_kernel, dim3(B), final_lds(*f), (hipStream_t)stream, *f, "graph_final_bwd");
}
extern "C" int scl_graph_reduce(const SclGraphReduceJob* jobs, int njobs, void* stream) {
    SCL_REQUIRE(jobs && njobs > 0 && njobs <= SCL_GRAPH_MAX_REDUCE_JOBS, "graph_reduce: 1..%d jobs", SCL_GRAPH_MAX_REDUCE_JOBS);
    ReduceJobs rj;
    int chunks = 0;
    for (int i = 0; i < njobs; ++i) {
        SCL_REQUIRE(jobs[i].src && jobs[i].dst && jobs[i].n > 0 && jobs[i].nparts > 0, "graph_reduce: bad job %d", i);
        rj.j[i] = jobs[i]; rj.chunk0[i] = chunks; chunks += (jobs[i].n + 31) / 32;
    }
    rj.chunk0[njobs] = chunks; rj.njobs = njobs;
    hipLaunchKernelGGL(graph_reduce_kernel, dim3(chunks), dim3(GT), 0, (hipStream_t)stream, rj);
    return scl_check_launch("graph_reduce");
}
