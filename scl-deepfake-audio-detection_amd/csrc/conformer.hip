// conformer.hip — the operators of a Conformer block (model/conformer.py:25-46, 98-106, 147-174) that the GEMM family, LayerNorm and
// BatchNorm kernels do not already cover: Swish, GLU, the depthwise 1-D convolution of the convolution module (forward; the data
// gradient is the same kernel on flipped taps; weight gradient as per-slab partials + a fixed-order finish) and Shaw's relative
// positions in the attention scores (table gather, skewed add + soft-max, its backward, the table's gradient).  fp32 throughout, every
// map channels-last ([B, n, C] = [B n rows][C]); all of it HBM-bound element-wise work, so the rules are 16-byte accesses and no
// re-reads beyond the convolution window.
#include "common.h"
#include <float.h>

namespace {

__device__ __forceinline__ float cf_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

inline int cf_grid(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

#define CF_LOOP(i, n) for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

// ---- Swish (x * sigmoid(x), conformer.py:25-27), GLU (conformer.py:29-36), out = sa * a + sb * b ---------------------------------------
__global__ void swish_fwd_kernel(const float4* __restrict__ x, float4* __restrict__ y, int64_t n4) {
    CF_LOOP(i, n4) {
        const float4 v = x[i];
        y[i] = make_float4(v.x * cf_sigmoid(v.x), v.y * cf_sigmoid(v.y), v.z * cf_sigmoid(v.z), v.w * cf_sigmoid(v.w));
    }
}
__device__ __forceinline__ float swish_grad(float x) {
    const float s = cf_sigmoid(x);
    return s * (1.0f + x * (1.0f - s));
}
__global__ void swish_bwd_kernel(const float4* __restrict__ dy, const float4* __restrict__ x, float4* __restrict__ dx, int64_t n4) {
    CF_LOOP(i, n4) {
        const float4 v = x[i], g = dy[i];
        dx[i] = make_float4(g.x * swish_grad(v.x), g.y * swish_grad(v.y), g.z * swish_grad(v.z), g.w * swish_grad(v.w));
    }
}
__global__ void glu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t M, int C) {
    const int c4n = C / 4;
    CF_LOOP(i, M * c4n) {
        const int64_t m = i / c4n;
        const int c = (int)(i - m * c4n) * 4;
        const float4 a = *reinterpret_cast<const float4*>(x + m * 2 * C + c), g = *reinterpret_cast<const float4*>(x + m * 2 * C + C + c);
        *reinterpret_cast<float4*>(y + m * C + c) = make_float4(a.x * cf_sigmoid(g.x), a.y * cf_sigmoid(g.y), a.z * cf_sigmoid(g.z), a.w * cf_sigmoid(g.w));
    }
}
__global__ void glu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, int64_t M, int C) {
    const int c4n = C / 4;
    CF_LOOP(i, M * c4n) {
        const int64_t m = i / c4n;
        const int c = (int)(i - m * c4n) * 4;
        const float4 a = *reinterpret_cast<const float4*>(x + m * 2 * C + c), g = *reinterpret_cast<const float4*>(x + m * 2 * C + C + c);
        const float4 d = *reinterpret_cast<const float4*>(dy + m * C + c);
        const float s0 = cf_sigmoid(g.x), s1 = cf_sigmoid(g.y), s2 = cf_sigmoid(g.z), s3 = cf_sigmoid(g.w);
        *reinterpret_cast<float4*>(dx + m * 2 * C + c) = make_float4(d.x * s0, d.y * s1, d.z * s2, d.w * s3);
        *reinterpret_cast<float4*>(dx + m * 2 * C + C + c) =
            make_float4(d.x * a.x * s0 * (1.0f - s0), d.y * a.y * s1 * (1.0f - s1), d.z * a.z * s2 * (1.0f - s2), d.w * a.w * s3 * (1.0f - s3));
    }
}
__global__ void axpby_kernel(const float4* __restrict__ a, const float4* __restrict__ b, float sa, float sb, float4* __restrict__ out, int64_t n4) {
    CF_LOOP(i, n4) {
        const float4 u = a[i];
        float4 r = make_float4(sa * u.x, sa * u.y, sa * u.z, sa * u.w);
        if (b) { const float4 v = b[i]; r.x += sb * v.x; r.y += sb * v.y; r.z += sb * v.z; r.w += sb * v.w; }
        out[i] = r;
    }
}

// ---- depthwise 1-D convolution (conformer.py:38-46: F.pad(x, (pad_l, pad_r)) then nn.Conv1d(C, C, k, groups = C)) ----------------------
// y[b][t][c] = bias[c] + sum_j w[c][j] x[b][t + j - pad_l][c], zero outside 0 <= t' < n; the output has the input's length (k - 1 =
// pad_l + pad_r in every form the block builds).  A block owns 64 channels x 128 positions of one utterance: 16 lanes x 4 channels,
// 16 position groups x 8 outputs; the taps of its channels sit transposed in LDS.  flip: taps read back to front (the data gradient
// dx[t] = sum_j w[j] dy[t - j + pad_l] is this kernel with pad_l' = k - 1 - pad_l and no bias).
constexpr int DW_KMAX = 32, DW_TT = 8, DW_TB = 16 * DW_TT;
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ y, int n, int C, int k, int pad_l, int flip) {
    __shared__ float4 wt[DW_KMAX][16];
    const int c0 = blockIdx.x * 64, b = blockIdx.z;
    for (int i = threadIdx.x; i < k * 64; i += 256) {
        const int j = i >> 6, cc = i & 63;
        const float v = c0 + cc < C ? w[(int64_t)(c0 + cc) * k + (flip ? k - 1 - j : j)] : 0.f;
        reinterpret_cast<float*>(&wt[j][0])[cc] = v;
    }
    __syncthreads();
    const int cq = threadIdx.x & 15, tg = threadIdx.x >> 4;
    const int c = c0 + 4 * cq, t0 = blockIdx.y * DW_TB + tg * DW_TT;
    if (c >= C || t0 >= n) return;
    float4 acc[DW_TT];
    const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int o = 0; o < DW_TT; ++o) acc[o] = bv;
    const float* xb = x + (int64_t)b * n * C + c;
    for (int r = 0; r < k + DW_TT - 1; ++r) {
        const int tin = t0 + r - pad_l;
        if (tin < 0 || tin >= n) continue;
        const float4 xv = *reinterpret_cast<const float4*>(xb + (int64_t)tin * C);
#pragma unroll
        for (int o = 0; o < DW_TT; ++o) {
            const int j = r - o;
            if (j >= 0 && j < k) {
                const float4 wv = wt[j][cq];
                acc[o].x = fmaf(wv.x, xv.x, acc[o].x); acc[o].y = fmaf(wv.y, xv.y, acc[o].y);
                acc[o].z = fmaf(wv.z, xv.z, acc[o].z); acc[o].w = fmaf(wv.w, xv.w, acc[o].w);
            }
        }
    }
    float* yb = y + (int64_t)b * n * C + c;
#pragma unroll
    for (int o = 0; o < DW_TT; ++o)
        if (t0 + o < n) *reinterpret_cast<float4*>(yb + (int64_t)(t0 + o) * C) = acc[o];
}

// Weight / bias gradient: dw[c][j] = sum_{b, t} dy[b][t][c] x[b][t + j - pad_l][c], db[c] = sum dy.  Block = 64 channels x one slab of
// 128 positions of one utterance; 16 position groups accumulate in registers, are summed through LDS in group order and leave as
// part[slab][j][c] (j = k: the bias row); dwconv_wgrad_finish sums the slabs in index order (deterministic) into w's [C][k] layout.
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                           int n, int C, int k, int pad_l) {
    __shared__ float4 red[16][16];
    const int c0 = blockIdx.x * 64, b = blockIdx.z;
    const int cq = threadIdx.x & 15, tg = threadIdx.x >> 4;
    const int c = c0 + 4 * cq;
    const bool live = c < C;
    float4 acc[DW_KMAX + 1];
#pragma unroll
    for (int j = 0; j <= DW_KMAX; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        const float* xb = x + (int64_t)b * n * C + c;
        const float* db = dy + (int64_t)b * n * C + c;
        for (int o = 0; o < DW_TT; ++o) {
            const int t = blockIdx.y * DW_TB + tg + 16 * o;
            if (t >= n) break;
            const float4 g = *reinterpret_cast<const float4*>(db + (int64_t)t * C);
            acc[DW_KMAX].x += g.x; acc[DW_KMAX].y += g.y; acc[DW_KMAX].z += g.z; acc[DW_KMAX].w += g.w;
#pragma unroll
            for (int j = 0; j < DW_KMAX; ++j) {
                const int tin = t + j - pad_l;
                if (j < k && tin >= 0 && tin < n) {
                    const float4 xv = *reinterpret_cast<const float4*>(xb + (int64_t)tin * C);
                    acc[j].x = fmaf(g.x, xv.x, acc[j].x); acc[j].y = fmaf(g.y, xv.y, acc[j].y);
                    acc[j].z = fmaf(g.z, xv.z, acc[j].z); acc[j].w = fmaf(g.w, xv.w, acc[j].w);
                }
            }
        }
    }
    const int64_t slab = (int64_t)b * gridDim.y + blockIdx.y;
    float* pb = part + slab * (int64_t)(k + 1) * C;
#pragma unroll
    for (int j = 0; j <= DW_KMAX; ++j) {
        if (j < k || j == DW_KMAX) {
            red[tg][cq] = acc[j];
            __syncthreads();
            if (tg == 0 && live) {
                float4 s = red[0][cq];
#pragma unroll
                for (int g2 = 1; g2 < 16; ++g2) { const float4 v = red[g2][cq]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
                *reinterpret_cast<float4*>(pb + (int64_t)(j == DW_KMAX ? k : j) * C + c) = s;
            }
            __syncthreads();
        }
    }
}
__global__ void dwconv_wgrad_finish_kernel(const float* __restrict__ part, int nslab, int C, int k, float* __restrict__ dw, float* __restrict__ db) {
    CF_LOOP(i, (int64_t)(k + 1) * C) {
        const int j = (int)(i / C), c = (int)(i - (int64_t)j * C);
        float s = 0.f;
        for (int p = 0; p < nslab; ++p) s += part[(int64_t)p * (k + 1) * C + i];
        if (j < k) dw[(int64_t)c * k + j] = s;
        else if (db) db[c] = s;
    }
}

// ---- Shaw's relative positions (conformer.py:98-106) -------------------------------------------------------------------------------
// The reference gathers rel_pos_emb(clamp(i - j) + max_pos) into an [n, n, d] tensor and contracts it with q.  Only 2n - 1 distances
// occur: Eu[r'] = E[clamp(r' - (n - 1), -max_pos, max_pos) + max_pos], r' = 0 .. 2n - 2 (rows up to Nr zero), R = q Eu^T is an
// ordinary GEMM with N = Nr columns, and pos_attn[i][j] = R[i][i - j + n - 1].
__global__ void relpos_gather_kernel(const float* __restrict__ E, float* __restrict__ Eu, int n, int Nr, int D, int max_pos) {
    CF_LOOP(i, (int64_t)Nr * D) {
        const int rp = (int)(i / D), d = (int)(i - (int64_t)rp * D);
        float v = 0.f;
        if (rp < 2 * n - 1) {
            int r = rp - (n - 1);
            r = r < -max_pos ? -max_pos : (r > max_pos ? max_pos : r);
            v = E[(int64_t)(r + max_pos) * D + d];
        }
        Eu[i] = v;
    }
}
// dE[idx] = sum of dEu[r'] over the distances that clamp to idx (one for interior rows, a run at either end), in r' order; every row of
// dE is written (zero where no distance lands).
__global__ void relpos_scatter_grad_kernel(const float* __restrict__ dEu, float* __restrict__ dE, int n, int D, int max_pos) {
    CF_LOOP(i, (int64_t)(2 * max_pos + 1) * D) {
        const int idx = (int)(i / D), d = (int)(i - (int64_t)idx * D);
        const int r = idx - max_pos;
        int lo = r, hi = r;
        if (idx == 0) lo = -(n - 1);
        if (idx == 2 * max_pos) hi = n - 1;
        lo = lo < -(n - 1) ? -(n - 1) : lo;
        hi = hi > n - 1 ? n - 1 : hi;
        float s = 0.f;
        for (int q = lo; q <= hi; ++q) s += dEu[(int64_t)(q + n - 1) * D + d];
        dE[i] = s;
    }
}

// P[row][j] = softmax_j(S[row][j] * scale + R[row][i - j + n - 1] * scale), row = (b, h, i); one wave per row, n <= 1024.  mask (optional,
// bytes [B][n]): pairs with mask[b][i] & mask[b][j] == 0 take -FLT_MAX before the soft-max (conformer.py:108-113 masked_fill), so a
// fully masked row comes out uniform, as in the reference.  Columns n .. ldP - 1 of P are written as zeros.
constexpr int RP_MAXV = 16;
__global__ __launch_bounds__(256) void relpos_softmax_fwd_kernel(const float* __restrict__ S, const float* __restrict__ R, const uint8_t* __restrict__ mask,
                                                                 float* __restrict__ P, int64_t rows, int H, int n, int ldS, int ldR, int ldP, float scale) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int i = (int)(row % n);
    const int64_t b = row / ((int64_t)H * n);
    const float* s = S + row * ldS;
    const float* r = R + row * ldR + i + n - 1;
    const uint8_t* mk = mask ? mask + b * n : nullptr;
    const bool mi = mk ? mk[i] != 0 : true;
    float v[RP_MAXV];
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < RP_MAXV; ++u) {
        const int j = u * 64 + lane;
        v[u] = -INFINITY;
        if (j < n) {
            v[u] = s[j] * scale + r[-j] * scale;
            if (mk && !(mi && mk[j] != 0)) v[u] = -FLT_MAX;
        }
        mx = fmaxf(mx, v[u]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < RP_MAXV; ++u) {
        const int j = u * 64 + lane;
        v[u] = j < n ? expf(v[u] - mx) : 0.f;
        sum += v[u];
    }
    const float inv = 1.0f / wave_sum(sum);
    float* p = P + row * ldP;
#pragma unroll
    for (int u = 0; u < RP_MAXV; ++u) {
        const int j = u * 64 + lane;
        if (j < ldP) p[j] = v[u] * inv;
    }
}
// dS[row][j] = scale * P[j] (dP[j] - <P, dP>) (0 at masked pairs: masked_fill passes no gradient), dR[row][r'] = dS[row][i + n - 1 - r']
// where that key exists, else 0 — every element of both rows is written.
__global__ __launch_bounds__(256) void relpos_softmax_bwd_kernel(const float* __restrict__ P, const float* __restrict__ dP, const uint8_t* __restrict__ mask,
                                                                 float* __restrict__ dS, float* __restrict__ dR, int64_t rows, int H, int n, int ldP,
                                                                 int ldR, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int i = (int)(row % n);
    const int64_t b = row / ((int64_t)H * n);
    const float* p = P + row * ldP;
    const float* g = dP + row * ldP;
    const uint8_t* mk = mask ? mask + b * n : nullptr;
    const bool mi = mk ? mk[i] != 0 : true;
    float dot = 0.f;
    for (int j = lane; j < n; j += 64) dot += p[j] * g[j];
    dot = wave_sum(dot);
    float* ds = dS + row * ldP;
    for (int j = lane; j < ldP; j += 64) {
        float v = 0.f;
        if (j < n && (!mk || (mi && mk[j] != 0))) v = scale * p[j] * (g[j] - dot);
        ds[j] = v;
    }
    float* dr = dR + row * ldR;
    for (int rp = lane; rp < ldR; rp += 64) {
        const int j = i + n - 1 - rp;
        float v = 0.f;
        if (j >= 0 && j < n && (!mk || (mi && mk[j] != 0))) v = scale * p[j] * (g[j] - dot);
        dr[rp] = v;
    }
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" int scl_swish_fwd(const float* x, float* y, int64_t n, void* stream) {
    SCL_REQUIRE(x && y && n > 0 && (n & 3) == 0 && al16(x) && al16(y), "swish_fwd: n a multiple of 4, 16-byte aligned pointers");
    hipLaunchKernelGGL(swish_fwd_kernel, dim3(cf_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, (const float4*)x, (float4*)y, n / 4);
    return scl_check_launch("scl_swish_fwd");
}
extern "C" int scl_swish_bwd(const float* dy, const float* x, float* dx, int64_t n, void* stream) {
    SCL_REQUIRE(dy && x && dx && n > 0 && (n & 3) == 0 && al16(dy) && al16(x) && al16(dx), "swish_bwd: n a multiple of 4, 16-byte aligned pointers");
    hipLaunchKernelGGL(swish_bwd_kernel, dim3(cf_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, (const float4*)dy, (const float4*)x, (float4*)dx, n / 4);
    return scl_check_launch("scl_swish_bwd");
}
extern "C" int scl_glu_fwd(const float* x, float* y, int64_t M, int C, void* stream) {
    SCL_REQUIRE(x && y && M > 0 && C > 0 && (C & 3) == 0 && al16(x) && al16(y), "glu_fwd: C a multiple of 4, 16-byte aligned pointers");
    hipLaunchKernelGGL(glu_fwd_kernel, dim3(cf_grid(M * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, y, M, C);
    return scl_check_launch("scl_glu_fwd");
}
extern "C" int scl_glu_bwd(const float* dy, const float* x, float* dx, int64_t M, int C, void* stream) {
    SCL_REQUIRE(dy && x && dx && M > 0 && C > 0 && (C & 3) == 0 && al16(dy) && al16(x) && al16(dx), "glu_bwd: C a multiple of 4, 16-byte aligned pointers");
    hipLaunchKernelGGL(glu_bwd_kernel, dim3(cf_grid(M * (C / 4))), dim3(256), 0, (hipStream_t)stream, dy, x, dx, M, C);
    return scl_check_launch("scl_glu_bwd");
}
extern "C" int scl_axpby_f32(const float* a, const float* b, float sa, float sb, float* out, int64_t n, void* stream) {
    SCL_REQUIRE(a && out && n > 0 && (n & 3) == 0 && al16(a) && al16(out) && (!b || al16(b)), "axpby: n a multiple of 4, 16-byte aligned pointers");
    hipLaunchKernelGGL(axpby_kernel, dim3(cf_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, (const float4*)a, (const float4*)b, sa, sb, (float4*)out, n / 4);
    return scl_check_launch("scl_axpby_f32");
}
extern "C" int scl_dwconv1d_fwd(const float* x, const float* w, const float* bias, float* y, int B, int n, int C, int k, int pad_l, int flip,
                                void* stream) {
    SCL_REQUIRE(x && w && y && B > 0 && n > 0 && C > 0 && (C & 3) == 0, "dwconv1d_fwd: C a multiple of 4");
    SCL_REQUIRE(k >= 1 && k <= DW_KMAX && pad_l >= 0 && pad_l < k, "dwconv1d_fwd: 1 <= k <= %d taps, 0 <= pad_l < k", DW_KMAX);
    SCL_REQUIRE(al16(x) && al16(y) && (!bias || al16(bias)), "dwconv1d_fwd: 16-byte aligned maps");
    hipLaunchKernelGGL(dwconv_fwd_kernel, dim3((C + 63) / 64, (n + DW_TB - 1) / DW_TB, B), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, n, C, k, pad_l, flip);
    return scl_check_launch("scl_dwconv1d_fwd");
}
extern "C" int scl_dwconv1d_wgrad_nslabs(int B, int n) { return B * ((n + DW_TB - 1) / DW_TB); }
extern "C" int scl_dwconv1d_wgrad(const float* x, const float* dy, float* part, float* dw, float* db, int B, int n, int C, int k, int pad_l,
                                  void* stream) {
    SCL_REQUIRE(x && dy && part && dw && B > 0 && n > 0 && C > 0 && (C & 3) == 0, "dwconv1d_wgrad: C a multiple of 4");
    SCL_REQUIRE(k >= 1 && k <= DW_KMAX && pad_l >= 0 && pad_l < k, "dwconv1d_wgrad: 1 <= k <= %d taps, 0 <= pad_l < k", DW_KMAX);
    SCL_REQUIRE(al16(x) && al16(dy) && al16(part), "dwconv1d_wgrad: 16-byte aligned maps");
    hipLaunchKernelGGL(dwconv_wgrad_kernel, dim3((C + 63) / 64, (n + DW_TB - 1) / DW_TB, B), dim3(256), 0, (hipStream_t)stream, x, dy, part, n, C, k, pad_l);
    hipLaunchKernelGGL(dwconv_wgrad_finish_kernel, dim3(cf_grid((int64_t)(k + 1) * C)), dim3(256), 0, (hipStream_t)stream, (const float*)part,
                       scl_dwconv1d_wgrad_nslabs(B, n), C, k, dw, db);
    return scl_check_launch("scl_dwconv1d_wgrad");
}
extern "C" int scl_relpos_gather(const float* E, float* Eu, int n, int Nr, int D, int max_pos, void* stream) {
    SCL_REQUIRE(E && Eu && n > 0 && Nr >= 2 * n - 1 && D > 0 && max_pos >= 0, "relpos_gather: bad args");
    hipLaunchKernelGGL(relpos_gather_kernel, dim3(cf_grid((int64_t)Nr * D)), dim3(256), 0, (hipStream_t)stream, E, Eu, n, Nr, D, max_pos);
    return scl_check_launch("scl_relpos_gather");
}
extern "C" int scl_relpos_scatter_grad(const float* dEu, float* dE, int n, int D, int max_pos, void* stream) {
    SCL_REQUIRE(dEu && dE && n > 0 && D > 0 && max_pos >= 0, "relpos_scatter_grad: bad args");
    hipLaunchKernelGGL(relpos_scatter_grad_kernel, dim3(cf_grid((int64_t)(2 * max_pos + 1) * D)), dim3(256), 0, (hipStream_t)stream, dEu, dE, n, D, max_pos);
    return scl_check_launch("scl_relpos_scatter_grad");
}
extern "C" int scl_relpos_softmax_fwd(const float* S, const float* R, const uint8_t* mask, float* P, int B, int H, int n, int ldS, int ldR, int ldP,
                                      float scale, void* stream) {
    SCL_REQUIRE(S && R && P && B > 0 && H > 0 && n > 0 && n <= 64 * RP_MAXV, "relpos_softmax_fwd: 1 <= n <= %d", 64 * RP_MAXV);
    SCL_REQUIRE(ldS >= n && ldP >= n && ldP <= 64 * RP_MAXV && ldR >= 2 * n - 1, "relpos_softmax_fwd: row pitches (ldS, ldP >= n; ldR >= 2n - 1)");
    const int64_t rows = (int64_t)B * H * n;
    hipLaunchKernelGGL(relpos_softmax_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, S, R, mask, P, rows, H, n, ldS, ldR, ldP, scale);
    return scl_check_launch("scl_relpos_softmax_fwd");
}
extern "C" int scl_relpos_softmax_bwd(const float* P, const float* dP, const uint8_t* mask, float* dS, float* dR, int B, int H, int n, int ldP, int ldR,
                                      float scale, void* stream) {
    SCL_REQUIRE(P && dP && dS && dR && B > 0 && H > 0 && n > 0 && ldP >= n && ldR >= 2 * n - 1, "relpos_softmax_bwd: bad args");
    const int64_t rows = (int64_t)B * H * n;
    hipLaunchKernelGGL(relpos_softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, P, dP, mask, dS, dR, rows, H, n, ldP, ldR, scale);
    return scl_check_launch("scl_relpos_softmax_bwd");
}
