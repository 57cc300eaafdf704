// gat.hip — the pairwise attention score of the AASIST graph-attention layers, forward and backward, fused.
//
//   s[b][i][j] = sum_o tanh( sum_d W[o][d] * x[b][i][d] * x[b][j][d] + bias[o] ) * a_t(i,j)[o]
//
// is what GraphAttentionLayer._derive_att_map / HtrgGraphAttentionLayer._derive_att_map compute before the temperature and
// the softmax (model/wav2vec2_aasist.py:107-135, 259-291): `att_proj` applied to the element-wise product of every node
// pair, tanh, and a dot with `att_weight` — for the heterogeneous layer one of three vectors chosen by the types of the
// two nodes (a11 both < n1, a22 both >= n1, a12 mixed).  The reference materialises [B, N, N, D] and [B, N, N, D'] tensors
// and runs fp32 library GEMMs with M = B*N*N rows on them (15 TFLOP/s, ~6 ms of the 16 ms the torch-composed back-end
// costs per step); here a thread owns one node pair, W / x live in LDS, and nothing of size N*N*D ever reaches HBM in the
// forward.  All arithmetic is fp32 (the back-end's parity bar is the fp32 one).
//
// Backward (given ds): recompute h per pair, then
//   d a_t[o]  = sum_{pairs of type t} ds * h_o                       (wave reductions -> per-block partials)
//   d pre_o   = ds * a_t[o] * (1 - h_o^2);  d bias[o] = sum d pre_o
//   d W[o][d] = sum_pairs d pre_o * x_i[d] x_j[d]                    (d pre tile in LDS, 16 outputs per thread)
//   d p[d]    = sum_o d pre_o W[o][d]  -> dP[b][i][j][d] in HBM (the only N*N*D tensor), gathered by a second kernel into
//   d x_i[d]  = sum_j (dP[i][j][d] + dP[j][i][d]) * x_j[d]           (deterministic: no atomics anywhere)
#include "common.h"

namespace {

constexpr int GAT_MAXN = 128;     // nodes per graph
constexpr int GAT_PAIRS = 256;    // pairs (= threads) per block
constexpr int GAT_DPS = 68;       // row stride (floats) of the backward's d-pre tile [pair][o]: 16-byte aligned, 4-way banked

template <int D>
struct GatSmem {
    // W [Do][D] | bias [Do] | a [3][Do] | x [N][D+1]
    __device__ static float* w(float* s) { return s; }
    __device__ static float* bias(float* s, int Do) { return s + Do * D; }
    __device__ static float* a(float* s, int Do) { return s + Do * D + Do; }
    __device__ static float* x(float* s, int Do) { return s + Do * D + 4 * Do; }
    __host__ __device__ static size_t floats(int Do, int N) { return (size_t)Do * D + 4 * Do + (size_t)N * (D + 1); }
};

template <int D>
__device__ __forceinline__ void gat_stage(float* sm, const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
                                          const float* __restrict__ a, int b, int N, int Do) {
    float* w = GatSmem<D>::w(sm);
    for (int i = threadIdx.x; i < Do * D; i += blockDim.x) w[i] = W[i];
    float* bs = GatSmem<D>::bias(sm, Do);
    float* as = GatSmem<D>::a(sm, Do);
    for (int i = threadIdx.x; i < Do; i += blockDim.x) bs[i] = bias[i];
    for (int i = threadIdx.x; i < 3 * Do; i += blockDim.x) as[i] = a[i];
    float* xs = GatSmem<D>::x(sm, Do);
    const float* xb = x + (int64_t)b * N * D;
    for (int i = threadIdx.x; i < N * D; i += blockDim.x) xs[(i / D) * (D + 1) + (i % D)] = xb[i];
}

__device__ __forceinline__ int gat_type(int i, int j, int n1) { return (i < n1) == (j < n1) ? (i < n1 ? 0 : 1) : 2; }

__device__ __forceinline__ float gat_tanh(float v) {
    // tanh(v) = 1 - 2 / (exp(2v) + 1): |err| < 2 ulp of fp32 over the whole range, saturates cleanly
    const float e = __expf(2.0f * v);
    return 1.0f - 2.0f / (e + 1.0f);
}

template <int D>
__global__ __launch_bounds__(GAT_PAIRS) void gat_score_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                                  const float* __restrict__ bias, const float* __restrict__ a,
                                                                  float* __restrict__ s, int N, int Do, int n1) {
    extern __shared__ float sm[];
    const int b = blockIdx.y;
    gat_stage<D>(sm, x, W, bias, a, b, N, Do);
    __syncthreads();
    const int q = blockIdx.x * GAT_PAIRS + threadIdx.x;
    if (q >= N * N) return;
    const int i = q / N, j = q - i * N;
    const float* w = GatSmem<D>::w(sm);
    const float* bs = GatSmem<D>::bias(sm, Do);
    const float* at = GatSmem<D>::a(sm, Do) + gat_type(i, j, n1) * Do;
    const float* xi = GatSmem<D>::x(sm, Do) + i * (D + 1);
    const float* xj = GatSmem<D>::x(sm, Do) + j * (D + 1);
    float p[D];
#pragma unroll
    for (int d = 0; d < D; ++d) p[d] = xi[d] * xj[d];
    float acc = 0.f;
    for (int o = 0; o < Do; ++o) {
        float pre = bs[o];
        const float4* wr = reinterpret_cast<const float4*>(w + o * D);
#pragma unroll
        for (int d4 = 0; d4 < D / 4; ++d4) {
            const float4 ww = wr[d4];
            pre += ww.x * p[4 * d4] + ww.y * p[4 * d4 + 1] + ww.z * p[4 * d4 + 2] + ww.w * p[4 * d4 + 3];
        }
        acc += gat_tanh(pre) * at[o];
    }
    s[(int64_t)b * N * N + q] = acc;
}

// part[blk] = [ dW (Do*D) | dbias (Do) | da (3*Do) ] ; dP[b][q][d]
template <int D>
__global__ __launch_bounds__(GAT_PAIRS) void gat_score_bwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                                  const float* __restrict__ bias, const float* __restrict__ a,
                                                                  const float* __restrict__ ds, float* __restrict__ dP,
                                                                  float* __restrict__ part, int N, int Do, int n1) {
    extern __shared__ float sm[];
    const int b = blockIdx.y;
    gat_stage<D>(sm, x, W, bias, a, b, N, Do);
    float* dpre_l = sm + ((GatSmem<D>::floats(Do, N) + 3) & ~(size_t)3);   // [GAT_PAIRS][GAT_DPS]
    float* red = dpre_l + (size_t)GAT_PAIRS * GAT_DPS;          // [4 waves][Do][4]  (-, da0, da1, da2)
    unsigned char* pi = reinterpret_cast<unsigned char*>(red + 4 * Do * 4);   // [GAT_PAIRS] i, then [GAT_PAIRS] j
    unsigned char* pj = pi + GAT_PAIRS;
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = blockIdx.x * GAT_PAIRS + tid;
    const bool valid = q < N * N;
    const int i = valid ? q / N : 0, j = valid ? q - i * N : 0;
    pi[tid] = (unsigned char)i; pj[tid] = (unsigned char)j;
    const int ty = gat_type(i, j, n1);
    const float* w = GatSmem<D>::w(sm);
    const float* bs = GatSmem<D>::bias(sm, Do);
    const float* at = GatSmem<D>::a(sm, Do) + ty * Do;
    const float* xs = GatSmem<D>::x(sm, Do);
    const float* xi = xs + i * (D + 1);
    const float* xj = xs + j * (D + 1);
    const float g = valid ? ds[(int64_t)b * N * N + q] : 0.f;
    float p[D], dp[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { p[d] = xi[d] * xj[d]; dp[d] = 0.f; }
    for (int o = 0; o < Do; ++o) {
        float pre = bs[o];
        const float4* wr = reinterpret_cast<const float4*>(w + o * D);
#pragma unroll
        for (int d4 = 0; d4 < D / 4; ++d4) {
            const float4 ww = wr[d4];
            pre += ww.x * p[4 * d4] + ww.y * p[4 * d4 + 1] + ww.z * p[4 * d4 + 2] + ww.w * p[4 * d4 + 3];
        }
        const float h = gat_tanh(pre);
        const float dpre = g * at[o] * (1.0f - h * h);
        dpre_l[tid * GAT_DPS + o] = dpre;
#pragma unroll
        for (int d4 = 0; d4 < D / 4; ++d4) {
            const float4 ww = wr[d4];
            dp[4 * d4] += dpre * ww.x; dp[4 * d4 + 1] += dpre * ww.y; dp[4 * d4 + 2] += dpre * ww.z; dp[4 * d4 + 3] += dpre * ww.w;
        }
        // per-wave sums of ds*h by pair type (fixed order: deterministic); the homogeneous layer has one type
        const float gh = g * h;
        const float r1 = wave_sum(ty == 0 ? gh : 0.f);
        float r2 = 0.f, r3 = 0.f;
        if (n1 < N) { r2 = wave_sum(ty == 1 ? gh : 0.f); r3 = wave_sum(ty == 2 ? gh : 0.f); }
        if (lane == 0) {
            float* r = red + (wave * Do + o) * 4;
            r[1] = r1; r[2] = r2; r[3] = r3;
        }
    }
    if (valid) {
        float* out = dP + ((int64_t)b * N * N + q) * D;
#pragma unroll
        for (int d4 = 0; d4 < D / 4; ++d4)
            *reinterpret_cast<float4*>(out + 4 * d4) = make_float4(dp[4 * d4], dp[4 * d4 + 1], dp[4 * d4 + 2], dp[4 * d4 + 3]);
    }
    __syncthreads();
    float* pb = part + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * ((size_t)Do * D + 4 * Do);
    // ---- dW[o][d] = sum_pairs dpre[pair][o] * x_i[d] x_j[d]: thread -> (d = tid % D, 16 consecutive o); the whole wave shares the
    //      d-pre addresses (LDS broadcast), x reads are conflict-free across d; dbias[o] = sum_pairs dpre[pair][o] from the same pass
    {
        const int d = tid % D, og = tid / D;
        for (int o0 = og * 16; o0 < Do; o0 += (GAT_PAIRS / D) * 16) {
            float accw[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) accw[t] = 0.f;
            for (int pr = 0; pr < GAT_PAIRS; ++pr) {
                const float pv = xs[pi[pr] * (D + 1) + d] * xs[pj[pr] * (D + 1) + d];
                const float4* dr = reinterpret_cast<const float4*>(dpre_l + pr * GAT_DPS + o0);
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
                    const float4 dv = dr[t4];
                    accw[4 * t4] += dv.x * pv; accw[4 * t4 + 1] += dv.y * pv; accw[4 * t4 + 2] += dv.z * pv; accw[4 * t4 + 3] += dv.w * pv;
                }
            }
#pragma unroll
            for (int t = 0; t < 16; ++t)
                if (o0 + t < Do) pb[(o0 + t) * D + d] = accw[t];
        }
    }
    // ---- dbias[o] (row sums of the d-pre tile) and da: combine the 4 waves
    for (int o = tid; o < Do; o += GAT_PAIRS) {
        float sb = 0.f;
        for (int pr = 0; pr < GAT_PAIRS; ++pr) sb += dpre_l[pr * GAT_DPS + o];
        pb[(size_t)Do * D + o] = sb;
    }
    for (int t = tid; t < Do * 3; t += GAT_PAIRS) {
        const int o = t / 3, c = 1 + t % 3;
        const float v = red[(0 * Do + o) * 4 + c] + red[(1 * Do + o) * 4 + c] + red[(2 * Do + o) * 4 + c] + red[(3 * Do + o) * 4 + c];
        pb[(size_t)Do * D + Do + (c - 1) * Do + o] = v;
    }
}

// dx[b][i][d] = sum_j (dP[b][i][j][d] + dP[b][j][i][d]) * x[b][j][d];  one block per (b, i), threads = 4 j-lanes x D
template <int D>
__global__ void gat_dx_kernel(const float* __restrict__ dP, const float* __restrict__ x, float* __restrict__ dx, int N) {
    __shared__ float red[4][D];
    const int b = blockIdx.y, i = blockIdx.x;
    const int d = threadIdx.x % D, jl = threadIdx.x / D;
    const float* xb = x + (int64_t)b * N * D;
    const float* pb = dP + (int64_t)b * N * N * D;
    float s = 0.f;
    for (int j = jl; j < N; j += 4)
        s += (pb[((int64_t)i * N + j) * D + d] + pb[((int64_t)j * N + i) * D + d]) * xb[(int64_t)j * D + d];
    red[jl][d] = s;
    __syncthreads();
    if (jl == 0) dx[((int64_t)b * N + i) * D + d] = red[0][d] + red[1][d] + red[2][d] + red[3][d];
}

// ---- matrix-core forms (D in {32, 64}, Do in {32, 64}) ----------------------------------------------------------------------------------
// The pre-activation of a pair is a [pairs x D] x [D x Do] product whose left operand is formed on the fly (x_i * x_j): on
// v_mfma_f32_16x16x4_f32 (exact fp32) a wave takes one node i and 16 nodes j at a time — rows = the 16 pairs, k = the feature index,
// columns = 16 outputs — with the whole of W in registers (Do * D / 64 per lane, loaded once per block), so the N * N * D * Do
// multiply-adds leave the vector ALU, which keeps the N * N * Do tanh evaluations.  Backward, per 16 pairs: the same pre-activations
// again, then dW += dpre^T p (rows = outputs, k = pairs: the accumulators of the first product ARE the left operand) into Do * D / 64
// accumulators per lane that live for the whole block, and dp^T = W^T dpre^T after one trip of the dpre tile through LDS (the only
// transposition); dP goes to HBM for gat_dx_kernel as before.  Blocks = scl_gat_score_nblocks(N) per graph, nodes i dealt out evenly,
// so the partial-sum layout the callers reduce is unchanged.
template <int D>
__device__ __forceinline__ void gat_stage_x(float* xs, const float* __restrict__ xb, int N) {      // [N + 16][D + 1], rows >= N zero
    for (int e = threadIdx.x; e < (N + 16) * D; e += GAT_PAIRS) { const int r = e / D, d = e - r * D; xs[r * (D + 1) + d] = r < N ? xb[e] : 0.f; }
}

template <int D, int Do>
__global__ __launch_bounds__(GAT_PAIRS, 1) void gat_score_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                                         const float* __restrict__ bias, const float* __restrict__ a,
                                                                         float* __restrict__ s, int N, int n1) {
    constexpr int NS = D / 4, NOB = Do / 16, XP = D + 1;
    extern __shared__ float sm[];
    float* xs = sm;
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
    gat_stage_x<D>(xs, x + (int64_t)b * N * D, N);
    float wr[NOB][NS], bo[NOB], av[3][NOB];
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
        const int o = 16 * ob + li;
        bo[ob] = bias[o];
#pragma unroll
        for (int t = 0; t < 3; ++t) av[t][ob] = a[t * Do + o];
#pragma unroll
        for (int k = 0; k < NS; ++k) wr[ob][k] = W[(size_t)o * D + 4 * k + g];
    }
    __syncthreads();
    const int nblk = gridDim.x, ipb = (N + nblk - 1) / nblk;
    const int i0 = blockIdx.x * ipb, i1 = min(N, i0 + ipb), njb = (N + 15) / 16;
    for (int u = wave; u < (i1 - i0) * njb; u += GAT_PAIRS / 64) {
        const int i = i0 + u / njb, j0 = (u % njb) * 16;
        f32x4 acc[NOB];
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) acc[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const float pv = xs[i * XP + 4 * k + g] * xs[(j0 + li) * XP + 4 * k + g];
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv, wr[ob][k], acc[ob], 0, 0, 0);
        }
        float sc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = j0 + 4 * g + r;
            const int ty = gat_type(i, j < N ? j : 0, n1);
            float v = 0.f;
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) v += gat_tanh(acc[ob][r] + bo[ob]) * (ty == 0 ? av[0][ob] : (ty == 1 ? av[1][ob] : av[2][ob]));
            sc[r] = lanes16_sum(v);
        }
        if (li == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int j = j0 + 4 * g + r; if (j < N) s[((int64_t)b * N + i) * N + j] = sc[r]; }
        }
    }
}

template <int D, int Do>
__global__ __launch_bounds__(GAT_PAIRS, 1) void gat_score_bwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                                         const float* __restrict__ bias, const float* __restrict__ a,
                                                                         const float* __restrict__ ds, float* __restrict__ dP,
                                                                         float* __restrict__ part, int N, int n1) {
    constexpr int NS = D / 4, NOB = Do / 16, NDB = D / 16, NSO = Do / 4, XP = D + 1, TP = Do + 1;
    extern __shared__ float sm[];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
    float* xs = sm;                                        // [N + 16][XP]
    float* Tw = sm + (N + 16) * XP + wave * 16 * TP;       // this wave's dpre tile [16 pairs][TP]
    float* red = sm + (N + 16) * XP + 4 * 16 * TP;         // [4 waves][4 * Do]: db | da0 | da1 | da2
    gat_stage_x<D>(xs, x + (int64_t)b * N * D, N);
    float wr[NOB][NS], wt[NDB][NSO], bo[NOB], av[3][NOB];
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
        const int o = 16 * ob + li;
        bo[ob] = bias[o];
#pragma unroll
        for (int t = 0; t < 3; ++t) av[t][ob] = a[t * Do + o];
#pragma unroll
        for (int k = 0; k < NS; ++k) wr[ob][k] = W[(size_t)o * D + 4 * k + g];
    }
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int k = 0; k < NSO; ++k) wt[db][k] = W[(size_t)(4 * k + g) * D + 16 * db + li];      // W^T: row d = 16 db + li, k <-> o = 4 k + g
    f32x4 dwa[NOB][NDB];
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
        for (int db = 0; db < NDB; ++db) dwa[ob][db] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbs[NOB], das[3][NOB];
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) { dbs[ob] = 0.f; das[0][ob] = 0.f; das[1][ob] = 0.f; das[2][ob] = 0.f; }
    __syncthreads();
    const int nblk = gridDim.x, ipb = (N + nblk - 1) / nblk;
    const int i0 = blockIdx.x * ipb, i1 = min(N, i0 + ipb), njb = (N + 15) / 16;
    for (int u = wave; u < (i1 - i0) * njb; u += GAT_PAIRS / 64) {
        const int i = i0 + u / njb, j0 = (u % njb) * 16;
        f32x4 acc[NOB];
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) acc[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const float pv = xs[i * XP + 4 * k + g] * xs[(j0 + li) * XP + 4 * k + g];
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv, wr[ob][k], acc[ob], 0, 0, 0);
        }
        // this lane: outputs o = 16 ob + li of the pairs (i, j0 + 4 g + r)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = j0 + 4 * g + r;
            const float gd = j < N ? ds[((int64_t)b * N + i) * N + j] : 0.f;
            const int ty = gat_type(i, j < N ? j : 0, n1);
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) {
                const float h = gat_tanh(acc[ob][r] + bo[ob]);
                const float at = ty == 0 ? av[0][ob] : (ty == 1 ? av[1][ob] : av[2][ob]);
                const float gh = gd * h;
                das[0][ob] += ty == 0 ? gh : 0.f; das[1][ob] += ty == 1 ? gh : 0.f; das[2][ob] += ty == 2 ? gh : 0.f;
                const float dpre = gd * at * (1.0f - h * h);
                dbs[ob] += dpre;
                acc[ob][r] = dpre;
                Tw[(4 * g + r) * TP + 16 * ob + li] = dpre;
            }
        }
        // dW[o][d] += sum_pairs dpre[pair][o] p[pair][d]:  rows = o, k = pairs (k index g <-> pair 4 g + r), columns = d
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                const float pv = xs[i * XP + 16 * db + li] * xs[(j0 + 4 * g + r) * XP + 16 * db + li];
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) dwa[ob][db] = __builtin_amdgcn_mfma_f32_16x16x4f32(acc[ob][r], pv, dwa[ob][db], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the tile's writes are this wave's own: in order, complete before the reads below
        // dp^T[d][pair] = sum_o W[o][d] dpre[pair][o]:  rows = d, k = o, columns = pairs (j0 + li)
        f32x4 dpa[NDB];
#pragma unroll
        for (int db = 0; db < NDB; ++db) dpa[db] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < NSO; ++k) {
            const float tv = Tw[li * TP + 4 * k + g];
#pragma unroll
            for (int db = 0; db < NDB; ++db) dpa[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[db][k], tv, dpa[db], 0, 0, 0);
        }
        if (j0 + li < N) {
            float* out = dP + (((int64_t)b * N + i) * N + j0 + li) * D;
#pragma unroll
            for (int db = 0; db < NDB; ++db) *reinterpret_cast<f32x4*>(out + 16 * db + 4 * g) = dpa[db];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the reads are done before the next unit overwrites the tile
    }
    // ---- block partials: dW from the accumulators (rows o = 16 ob + 4 g + r', column d = 16 db + li), four waves summed through LDS in order
    float* pb = part + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * ((size_t)Do * D + 4 * Do);
    __syncthreads();
    float* wsum = sm;                                      // [Do * D] (x is no longer needed)
    for (int w4 = 0; w4 < 4; ++w4) {
        if (wave == w4) {
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
                for (int db = 0; db < NDB; ++db)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int idx = (16 * ob + 4 * g + r) * D + 16 * db + li;
                        wsum[idx] = (w4 == 0 ? 0.f : wsum[idx]) + dwa[ob][db][r];
                    }
        }
        __syncthreads();
    }
    for (int e = tid; e < Do * D; e += GAT_PAIRS) pb[e] = wsum[e];
    __syncthreads();      // small graphs: the sums' staging area below may overlap wsum
    // db / da: sum over the four lane groups g (they hold different pairs of the same output), then over the waves
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
        float v[4] = {dbs[ob], das[0][ob], das[1][ob], das[2][ob]};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v[q] += __shfl_xor(v[q], 16, 64);
            v[q] += __shfl_xor(v[q], 32, 64);
            if (g == 0) red[wave * 4 * Do + q * Do + 16 * ob + li] = v[q];
        }
    }
    __syncthreads();
    for (int e = tid; e < 4 * Do; e += GAT_PAIRS) pb[(size_t)Do * D + e] = (red[e] + red[4 * Do + e]) + (red[8 * Do + e] + red[12 * Do + e]);
}

template <int D, int Do>
size_t gat_mfma_lds(int N, bool bwd) {
    size_t f = (size_t)(N + 16) * (D + 1);
    if (bwd) { f += 4 * 16 * (Do + 1) + 16 * Do; if (f < (size_t)Do * D) f = (size_t)Do * D; }
    return f * sizeof(float);
}

}  // namespace

extern "C" int scl_gat_score_nblocks(int N) { return (N * N + GAT_PAIRS - 1) / GAT_PAIRS; }

extern "C" int scl_gat_score_fwd(const float* x, const float* W, const float* bias, const float* a, float* s, int B, int N, int D, int Do,
                                 int n1, void* stream) {
    SCL_REQUIRE(x && W && bias && a && s && B > 0, "gat_score_fwd: null pointer");
    SCL_REQUIRE((D == 64 || D == 32) && Do >= 1 && Do <= 64 && N >= 1 && N <= GAT_MAXN && n1 >= 0 && n1 <= N,
                "gat_score_fwd: need D in {32, 64}, Do <= 64, N <= 128 (D=%d Do=%d N=%d)", D, Do, N);
    dim3 grid(scl_gat_score_nblocks(N), B), block(GAT_PAIRS);
    hipStream_t st = (hipStream_t)stream;
#define GAT_FWD_MFMA(DD, OO) hipLaunchKernelGGL((gat_score_fwd_mfma_kernel<DD, OO>), grid, block, (gat_mfma_lds<DD, OO>(N, false)), st, x, W, bias, a, s, N, n1)
    if (D == 64 && Do == 64) { GAT_FWD_MFMA(64, 64); return scl_check_launch("scl_gat_score_fwd"); }
    if (D == 64 && Do == 32) { GAT_FWD_MFMA(64, 32); return scl_check_launch("scl_gat_score_fwd"); }
    if (D == 32 && Do == 32) { GAT_FWD_MFMA(32, 32); return scl_check_launch("scl_gat_score_fwd"); }
#undef GAT_FWD_MFMA
    if (D == 64) {
        const size_t lds = GatSmem<64>::floats(Do, N) * sizeof(float);
        hipLaunchKernelGGL((gat_score_fwd_kernel<64>), grid, block, lds, st, x, W, bias, a, s, N, Do, n1);
    } else {
        const size_t lds = GatSmem<32>::floats(Do, N) * sizeof(float);
        hipLaunchKernelGGL((gat_score_fwd_kernel<32>), grid, block, lds, st, x, W, bias, a, s, N, Do, n1);
    }
    return scl_check_launch("scl_gat_score_fwd");
}

// dP: f32 [B, N*N, D] scratch; part: f32 [B * nblocks(N)][Do*D + 4*Do] partial sums (dW | dbias | da11 | da22 | da12), to be
// summed over their first dimension by the caller (scl_colreduce_*); dx: f32 [B, N, D]
extern "C" int scl_gat_score_bwd(const float* x, const float* W, const float* bias, const float* a, const float* ds, float* dP, float* part,
                                 float* dx, int B, int N, int D, int Do, int n1, void* stream) {
    SCL_REQUIRE(x && W && bias && a && ds && dP && part && dx && B > 0, "gat_score_bwd: null pointer");
    SCL_REQUIRE((D == 64 || D == 32) && Do >= 1 && Do <= 64 && N >= 1 && N <= GAT_MAXN && n1 >= 0 && n1 <= N,
                "gat_score_bwd: need D in {32, 64}, Do <= 64, N <= 128 (D=%d Do=%d N=%d)", D, Do, N);
    dim3 grid(scl_gat_score_nblocks(N), B), block(GAT_PAIRS);
    hipStream_t st = (hipStream_t)stream;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)gat_score_bwd_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)gat_score_bwd_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
#define GAT_BWD_MFMA(DD, OO)                                                                                                                             \
    do {                                                                                                                                                  \
        hipLaunchKernelGGL((gat_score_bwd_mfma_kernel<DD, OO>), grid, block, (gat_mfma_lds<DD, OO>(N, true)), st, x, W, bias, a, ds, dP, part, N, n1);      \
        hipLaunchKernelGGL((gat_dx_kernel<DD>), dim3(N, B), dim3(4 * DD), 0, st, dP, x, dx, N);                                                           \
        return scl_check_launch("scl_gat_score_bwd");                                                                                                   \
    } while (0)
    if (D == 64 && Do == 64) GAT_BWD_MFMA(64, 64);
    if (D == 64 && Do == 32) GAT_BWD_MFMA(64, 32);
    if (D == 32 && Do == 32) GAT_BWD_MFMA(32, 32);
#undef GAT_BWD_MFMA
    if (D == 64) {
        const size_t lds = (((GatSmem<64>::floats(Do, N) + 3) & ~(size_t)3) + (size_t)GAT_PAIRS * GAT_DPS + 16 * Do) * sizeof(float) + 2 * GAT_PAIRS;
        SCL_REQUIRE(lds <= 160 * 1024, "gat_score_bwd: LDS tile too large");
        hipLaunchKernelGGL((gat_score_bwd_kernel<64>), grid, block, lds, st, x, W, bias, a, ds, dP, part, N, Do, n1);
        hipLaunchKernelGGL((gat_dx_kernel<64>), dim3(N, B), dim3(256), 0, st, dP, x, dx, N);
    } else {
        const size_t lds = (((GatSmem<32>::floats(Do, N) + 3) & ~(size_t)3) + (size_t)GAT_PAIRS * GAT_DPS + 16 * Do) * sizeof(float) + 2 * GAT_PAIRS;
        SCL_REQUIRE(lds <= 160 * 1024, "gat_score_bwd: LDS tile too large");
        hipLaunchKernelGGL((gat_score_bwd_kernel<32>), grid, block, lds, st, x, W, bias, a, ds, dP, part, N, Do, n1);
        hipLaunchKernelGGL((gat_dx_kernel<32>), dim3(N, B), dim3(128), 0, st, dP, x, dx, N);
    }
    return scl_check_launch("scl_gat_score_bwd");
}
