// nn.hip — the non-GEMM pieces of the AASIST / ResNet back-ends over channels-last fp32 maps [rows = (b, h, w)][C]:
// BatchNorm (batch or running statistics) fused with its activation (ReLU / SELU), forward and backward; copies into the zero-padded
// (and, for strided transposed convolutions, zero-dilated) maps the implicit-GEMM convolutions read; the 3x3 max pool in front of the
// AASIST encoder; global average pooling.  Replaces nn.BatchNorm2d / nn.BatchNorm1d / F.relu / nn.SELU / F.max_pool2d /
// F.adaptive_avg_pool2d of model/wav2vec2_aasist.py:377-433, 436-604, model/resnet.py:47-191, model/wav2vec2_resnet_nll.py:51-74 and
// their autograd backward (main.py:79).  HBM-bound streaming kernels: 16-byte accesses when C % 4 == 0, fp32 throughout; statistics
// are accumulated per 512-row slab in fp32 and combined over slabs in fp64 (the reference's CPU path accumulates in double).
#include "common.h"

namespace {

constexpr int BN_SLAB = 128;          // smallest statistics slab, in rows (a multiple of 256 / C for every legal C keeps a thread on one channel)
// rows per slab: 128 for small maps, else the multiple of 128 that leaves ~512 slabs — the finishing kernels walk the slab partials
// with 16 lanes per channel (2112 slabs of a [32, 66, 128] map took them 24 us each, 19 + 19 times a ResNet step)
static inline int bn_slab_rows(int N) {
    if (N <= BN_SLAB * 512) return BN_SLAB;
    const int r = (N + 511) / 512;
    return (r + BN_SLAB - 1) / BN_SLAB * BN_SLAB;
}
constexpr float SELU_ALPHA = 1.6732632423543772848170429916717f;
constexpr float SELU_SCALE = 1.0507009873554804934193349852946f;

__device__ __forceinline__ float nn_act(int act, float v) {
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 2) return v > 0.f ? SELU_SCALE * v : SELU_SCALE * SELU_ALPHA * (__expf(v) - 1.0f);
    return v;
}
// derivative of the activation expressed through its OUTPUT y (ReLU: y > 0; SELU: y > 0 ? scale : y + scale*alpha)
__device__ __forceinline__ float nn_act_grad_from_y(int act, float y) {
    if (act == 1) return y > 0.f ? 1.f : 0.f;
    if (act == 2) return y > 0.f ? SELU_SCALE : y + SELU_SCALE * SELU_ALPHA;
    return 1.f;
}

// partial (sum a, sum a*b) per slab and channel; a = f(row, c), b = g(row, c).  Thread t owns channel (t % C) when C <= 256 (C divides
// 256), channels t and t + 256 when C == 512: consecutive threads read consecutive addresses.
template <class F>
__device__ __forceinline__ void slab_reduce(int N, int C, double* part, int slab_rows, F f) {
    __shared__ double red[2][256];
    const int slab = blockIdx.x, t = threadIdx.x;
    const long long e0 = (long long)slab * slab_rows * C, e1 = min((long long)(slab + 1) * slab_rows, (long long)N) * C;
    const int nacc = C > 256 ? C / 256 : 1;
    for (int a = 0; a < nacc; ++a) {
        double s0 = 0.0, s1 = 0.0;      // fp64 from the first add: a scalar gradient such as first_bn.weight cancels 1e4 : 1
        const int ch = (t + 256 * a) & (C - 1);       // C is a power of two and divides the slab start
        const long long step = 256 * nacc;
        long long e = e0 + t + 256 * a;
        for (; e + 3 * step < e1; e += 4 * step) {    // four independent loads in flight per thread
            float u0, v0, u1, v1, u2, v2, u3, v3;
            f(e, ch, u0, v0); f(e + step, ch, u1, v1); f(e + 2 * step, ch, u2, v2); f(e + 3 * step, ch, u3, v3);
            s0 += ((double)u0 + (double)u1) + ((double)u2 + (double)u3); s1 += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
        }
        for (; e < e1; e += step) {
            float u, v;
            f(e, ch, u, v);
            s0 += u; s1 += v;
        }
        red[0][t] = s0; red[1][t] = s1;
        __syncthreads();
        const int cc = C > 256 ? 256 : C;          // threads t, t + cc, t + 2cc ... share a channel
        if (t < cc) {
            double r0 = 0.0, r1 = 0.0;
            for (int k = t; k < 256; k += cc) { r0 += red[0][k]; r1 += red[1][k]; }
            part[((long long)slab * 2 + 0) * C + t + 256 * a] = r0;
            part[((long long)slab * 2 + 1) * C + t + 256 * a] = r1;
        }
        __syncthreads();
    }
}

// The same reduction with 16-byte loads (C a power of two in 4 .. 512): a thread owns 4 consecutive channels, the block covers 1024 elements
// (1024 / C rows) per trip, four trips in flight.  Round 5: the scalar form read three maps with 4-byte loads and ran the ResNet back-end's
// BatchNorm backward statistics at 2.5 TB/s (19 launches x 82 us per step at batch 32).
template <class F>
__device__ __forceinline__ void slab_reduce4(int N, int C, double* part, int slab_rows, F f) {
    __shared__ double red4[2][4][256];
    const int slab = blockIdx.x, t = threadIdx.x;
    const long long e0 = (long long)slab * slab_rows * C, e1 = min((long long)(slab + 1) * slab_rows, (long long)N) * C;
    const int ch = (4 * t) & (C - 1);
    const long long step = 1024;
    double s0[4] = {0.0, 0.0, 0.0, 0.0}, s1[4] = {0.0, 0.0, 0.0, 0.0};
    long long e = e0 + 4 * t;
    for (; e + 3 * step < e1; e += 4 * step) {
        float4 u0, v0, u1, v1, u2, v2, u3, v3;
        f(e, ch, u0, v0); f(e + step, ch, u1, v1); f(e + 2 * step, ch, u2, v2); f(e + 3 * step, ch, u3, v3);
        s0[0] += ((double)u0.x + (double)u1.x) + ((double)u2.x + (double)u3.x); s1[0] += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
        s0[1] += ((double)u0.y + (double)u1.y) + ((double)u2.y + (double)u3.y); s1[1] += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
        s0[2] += ((double)u0.z + (double)u1.z) + ((double)u2.z + (double)u3.z); s1[2] += ((double)v0.z + (double)v1.z) + ((double)v2.z + (double)v3.z);
        s0[3] += ((double)u0.w + (double)u1.w) + ((double)u2.w + (double)u3.w); s1[3] += ((double)v0.w + (double)v1.w) + ((double)v2.w + (double)v3.w);
    }
    for (; e < e1; e += step) {
        float4 u, v;
        f(e, ch, u, v);
        s0[0] += u.x; s0[1] += u.y; s0[2] += u.z; s0[3] += u.w; s1[0] += v.x; s1[1] += v.y; s1[2] += v.z; s1[3] += v.w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red4[0][j][t] = s0[j]; red4[1][j][t] = s1[j]; }
    __syncthreads();
    const int groups = C / 4;                     // threads t, t + groups, ... share their 4 channels
    for (int o = t; o < C; o += 256) {
        const int g = o >> 2, j = o & 3;
        double r0 = 0.0, r1 = 0.0;
        for (int k = g; k < 256; k += groups) { r0 += red4[0][j][k]; r1 += red4[1][j][k]; }
        part[((long long)slab * 2 + 0) * C + o] = r0;
        part[((long long)slab * 2 + 1) * C + o] = r1;
    }
}

__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, int N, int C, double* __restrict__ part, int slab_rows) {
    if (C >= 4) {
        slab_reduce4(N, C, part, slab_rows, [&](long long e, int, float4& u, float4& v) {
            u = *reinterpret_cast<const float4*>(x + e);
            v = make_float4(u.x * u.x, u.y * u.y, u.z * u.z, u.w * u.w);
        });
        return;
    }
    slab_reduce(N, C, part, slab_rows, [&](long long e, int, float& u, float& v) { const float a = x[e]; u = a; v = a * a; });
}

// mean / rstd from the slab partials (training) or from the running statistics (eval); training also updates the running
// statistics with momentum (unbiased variance, as torch) and num_batches_tracked
// 64 channels per block x 16 slab lanes: the slab loop is split sixteen ways and combined through LDS (fp64)
__global__ __launch_bounds__(1024) void bn_finish_kernel(const double* __restrict__ part, int nslab, int N, int C, float eps, float momentum, int training,
                                 float* __restrict__ running_mean, float* __restrict__ running_var, long long* __restrict__ nbt,
                                 float* __restrict__ mean, float* __restrict__ rstd) {
    __shared__ double red[2][16][64];
    const int cl = threadIdx.x & 63, lane4 = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double s = 0.0, q = 0.0;
    if (training && c < C)
        for (int k = lane4; k < nslab; k += 16) { s += part[((long long)k * 2) * C + c]; q += part[((long long)k * 2 + 1) * C + c]; }
    red[0][lane4][cl] = s; red[1][lane4][cl] = q;
    __syncthreads();
    if (lane4 != 0 || c >= C) return;
    if (!training) {
        mean[c] = running_mean[c];
        rstd[c] = (float)(1.0 / sqrt((double)running_var[c] + (double)eps));
        return;
    }
    s = 0.0; q = 0.0;
    for (int k = 0; k < 16; ++k) { s += red[0][k][cl]; q += red[1][k][cl]; }
    const double m = s / N;
    double var = q / N - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unb = N > 1 ? var * N / (N - 1) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
    }
    if (nbt && c == 0) *nbt += 1;
}

struct RowMap { int W, HW; long long bs, rs, cs, base; };    // row r = (b, i, j): element offset base + b*bs + i*rs + j*cs (+ c)
__device__ __forceinline__ long long map_row(const RowMap& m, long long r) {
    const long long b = r / m.HW; const int ij = (int)(r - b * m.HW);
    const int i = ij / m.W, j = ij - i * m.W;
    return m.base + b * m.bs + (long long)i * m.rs + (long long)j * m.cs;
}

// y = act((x - mean) * rstd * gamma + beta) -> contiguous f32 (kept for the backward) and / or a mapped (padded) f32 / bf16 map
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, long long n, int C, int act,
                                                       float* __restrict__ y, void* __restrict__ y2, int y2_bf16, RowMap map) {
    const int cshift = __ffs(C) - 1;
    if ((C & 3) == 0 && !y2) {       // 16-byte path (every map but the single-channel input)
        for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < (n >> 2); q += (long long)gridDim.x * 256) {
            const long long e = q << 2;
            const int c = (int)(e & (C - 1));
            const float4 xv = *reinterpret_cast<const float4*>(x + e), mv = *reinterpret_cast<const float4*>(mean + c), rv = *reinterpret_cast<const float4*>(rstd + c);
            float4 gv = make_float4(1.f, 1.f, 1.f, 1.f), bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gamma) gv = *reinterpret_cast<const float4*>(gamma + c);
            if (beta) bv = *reinterpret_cast<const float4*>(beta + c);
            float4 o;
            o.x = nn_act(act, (xv.x - mv.x) * rv.x * gv.x + bv.x); o.y = nn_act(act, (xv.y - mv.y) * rv.y * gv.y + bv.y);
            o.z = nn_act(act, (xv.z - mv.z) * rv.z * gv.z + bv.z); o.w = nn_act(act, (xv.w - mv.w) * rv.w * gv.w + bv.w);
            *reinterpret_cast<float4*>(y + e) = o;
        }
        return;
    }
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const int c = (int)(e & (C - 1));
        const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
        const float v = nn_act(act, (x[e] - mean[c]) * rstd[c] * g + b);
        if (y) y[e] = v;
        if (y2) {
            const long long o = map_row(map, e >> cshift) + c;
            if (y2_bf16) reinterpret_cast<bf16_t*>(y2)[o] = f2bf(v); else reinterpret_cast<float*>(y2)[o] = v;
        }
    }
}

// backward, pass 1: per slab and channel (sum dz, sum dz * xhat) with dz = dy * act'(y)
__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ x,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd, int N, int C, int act,
                                                           double* __restrict__ part, int slab_rows) {
    if (C >= 4) {
        slab_reduce4(N, C, part, slab_rows, [&](long long e, int c, float4& u, float4& v) {
            float4 dz = *reinterpret_cast<const float4*>(dy + e);
            if (act) {
                const float4 yv = *reinterpret_cast<const float4*>(y + e);
                dz.x *= nn_act_grad_from_y(act, yv.x); dz.y *= nn_act_grad_from_y(act, yv.y); dz.z *= nn_act_grad_from_y(act, yv.z); dz.w *= nn_act_grad_from_y(act, yv.w);
            }
            const float4 xv = *reinterpret_cast<const float4*>(x + e), mv = *reinterpret_cast<const float4*>(mean + c), rv = *reinterpret_cast<const float4*>(rstd + c);
            u = dz;
            v = make_float4(dz.x * (xv.x - mv.x) * rv.x, dz.y * (xv.y - mv.y) * rv.y, dz.z * (xv.z - mv.z) * rv.z, dz.w * (xv.w - mv.w) * rv.w);
        });
        return;
    }
    slab_reduce(N, C, part, slab_rows, [&](long long e, int c, float& u, float& v) {
        const float dz = dy[e] * (act ? nn_act_grad_from_y(act, y[e]) : 1.f);
        u = dz; v = dz * (x[e] - mean[c]) * rstd[c];
    });
}
__global__ __launch_bounds__(1024) void bn_bwd_finish_kernel(const double* __restrict__ part, int nslab, int C, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, float* __restrict__ sums, int accumulate) {
    __shared__ double red[2][16][64];
    const int cl = threadIdx.x & 63, lane4 = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double s = 0.0, q = 0.0;
    if (c < C)
        for (int k = lane4; k < nslab; k += 16) { s += part[((long long)k * 2) * C + c]; q += part[((long long)k * 2 + 1) * C + c]; }
    red[0][lane4][cl] = s; red[1][lane4][cl] = q;
    __syncthreads();
    if (lane4 != 0 || c >= C) return;
    s = 0.0; q = 0.0;
    for (int k = 0; k < 16; ++k) { s += red[0][k][cl]; q += red[1][k][cl]; }
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)s : (float)s;
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)q : (float)q;
    sums[c] = (float)s; sums[C + c] = (float)q;
}
// pass 2: dx = gamma * rstd * (dz - [training] (sum dz + xhat * sum dz*xhat) / N)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ x,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ sums, long long n, int N, int C, int act, int training,
                                                           float* __restrict__ dx) {
    const float invN = 1.0f / (float)N;
    if ((C & 3) == 0) {
        for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < (n >> 2); q += (long long)gridDim.x * 256) {
            const long long e = q << 2;
            const int c = (int)(e & (C - 1));
            const float4 dv = *reinterpret_cast<const float4*>(dy + e), xv = *reinterpret_cast<const float4*>(x + e);
            const float4 mv = *reinterpret_cast<const float4*>(mean + c), rv = *reinterpret_cast<const float4*>(rstd + c);
            float4 yv = make_float4(0.f, 0.f, 0.f, 0.f), gv = make_float4(1.f, 1.f, 1.f, 1.f);
            if (act) yv = *reinterpret_cast<const float4*>(y + e);
            if (gamma) gv = *reinterpret_cast<const float4*>(gamma + c);
            const float d4[4] = {dv.x, dv.y, dv.z, dv.w}, x4[4] = {xv.x, xv.y, xv.z, xv.w}, m4[4] = {mv.x, mv.y, mv.z, mv.w};
            const float r4[4] = {rv.x, rv.y, rv.z, rv.w}, y4[4] = {yv.x, yv.y, yv.z, yv.w}, g4[4] = {gv.x, gv.y, gv.z, gv.w};
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float dz = d4[i] * (act ? nn_act_grad_from_y(act, y4[i]) : 1.f);
                const float xh = (x4[i] - m4[i]) * r4[i];
                float v = dz;
                if (training) v -= (sums[c + i] + xh * sums[C + c + i]) * invN;
                o[i] = g4[i] * r4[i] * v;
            }
            *reinterpret_cast<float4*>(dx + e) = make_float4(o[0], o[1], o[2], o[3]);
        }
        return;
    }
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const int c = (int)(e & (C - 1));
        const float dz = dy[e] * (act ? nn_act_grad_from_y(act, y[e]) : 1.f);
        const float xh = (x[e] - mean[c]) * rstd[c];
        const float g = gamma ? gamma[c] : 1.f;
        float v = dz;
        if (training) v -= (sums[c] + xh * sums[C + c]) * invN;
        dx[e] = g * rstd[c] * v;
    }
}

// src [B, H, W, C] contiguous f32 -> interior of a padded / dilated map: dst[b][ph + i*dh][pw + j*dw][c]  (f32 or bf16)
__global__ __launch_bounds__(256) void pad_nhwc_kernel(const float* __restrict__ src, long long n, int C, void* __restrict__ dst, int dst_bf16, RowMap map) {
    if ((C & 3) == 0 && !dst_bf16 && n < (1ll << 33) && ((map.bs | map.rs | map.cs | map.base) & 3) == 0) {
        // four channels per thread: one 16-byte load, one row-map evaluation (32-bit divisions) and one 16-byte store.  The element-wise loop
        // below spends ~100 instructions of 64-bit division per element: 2.2 TB/s on the ResNet's mid-size maps.
        const unsigned C4 = (unsigned)C >> 2, n4 = (unsigned)(n >> 2), HW = (unsigned)map.HW, W = (unsigned)map.W;
        float* d = reinterpret_cast<float*>(dst);
        for (unsigned q = blockIdx.x * 256u + threadIdx.x; q < n4; q += gridDim.x * 256u) {
            const unsigned row = q / C4, c4 = q - row * C4;
            const unsigned b = row / HW, ij = row - b * HW, i = ij / W, j = ij - i * W;
            const long long o = map.base + (long long)b * map.bs + (long long)i * map.rs + (long long)j * map.cs + 4 * c4;
            *reinterpret_cast<float4*>(d + o) = reinterpret_cast<const float4*>(src)[q];
        }
        return;
    }
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const long long o = map_row(map, e / C) + (e % C);       // any C here (the encoder input map has C = 1)
        if (dst_bf16) reinterpret_cast<bf16_t*>(dst)[o] = f2bf(src[e]); else reinterpret_cast<float*>(dst)[o] = src[e];
    }
}

// 3x3 / stride 3 max pool of a single-channel map x[b][H][W] (floor mode): y[b][H/3][W/3], idx = flat argmax inside x[b] (first maximum in
// row-major window order, as torch)
__global__ __launch_bounds__(256) void maxpool3_fwd_kernel(const float* __restrict__ x, long long xs_h, long long xs_w, long long xs_b, int H, int W, int B,
                                                           float* __restrict__ y, int* __restrict__ idx) {
    const int OH = H / 3, OW = W / 3;
    const long long n = (long long)B * OH * OW;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const int b = (int)(e / (OH * OW)), r = (int)(e - (long long)b * OH * OW), oh = r / OW, ow = r - oh * OW;
        float best = -INFINITY; int bi = 0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                const int h = 3 * oh + i, w = 3 * ow + j;
                const float v = x[b * xs_b + h * xs_h + w * xs_w];
                if (v > best || (v != v && best == best)) { best = v; bi = h * W + w; }
            }
        y[e] = best; idx[e] = bi;
    }
}
__global__ __launch_bounds__(256) void maxpool3_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ idx, int H, int W, int B,
                                                           float* __restrict__ dx, long long xs_h, long long xs_w, long long xs_b) {
    // windows do not overlap (stride = kernel): every input element belongs to at most one window -> plain stores into a zeroed dx
    const int OH = H / 3, OW = W / 3;
    const long long n = (long long)B * OH * OW;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const int b = (int)(e / (OH * OW));
        const int h = idx[e] / W, w = idx[e] - h * W;
        dx[b * xs_b + h * xs_h + w * xs_w] = dy[e];
    }
}

// y[b][c] = mean_r x[b][r][c]   /   dx[b][r][c] = dy[b][c] / R
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const float* __restrict__ x, int R, int C, float* __restrict__ y) {
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int r = 0; r < R; ++r) s += x[((long long)b * R + r) * C + c];
        y[(long long)b * C + c] = s / (float)R;
    }
}
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dy, int R, int C, long long n, float* __restrict__ dx) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const long long b = e / ((long long)R * C);
        dx[e] = dy[b * C + (e % C)] / (float)R;
    }
}

int grid_for(long long n) { long long b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b)); }
bool bn_channels_ok(int C) { return C >= 1 && (C == 512 || (C <= 256 && 256 % C == 0)); }

}  // namespace

extern "C" int scl_bn_nslabs(int N) { const int r = bn_slab_rows(N); return (N + r - 1) / r; }

extern "C" int scl_bn_fwd(const float* x, int N, int C, const float* gamma, const float* beta, float* running_mean, float* running_var,
                          long long* num_batches_tracked, int training, float momentum, float eps, int act, float* part, float* mean,
                          float* rstd, float* y, void* y2, int y2_bf16, int m_W, int m_HW, int64_t m_bs, int64_t m_rs, int64_t m_cs,
                          int64_t m_base, void* stream) {
    SCL_REQUIRE(x && mean && rstd && (y || y2) && N >= 1 && bn_channels_ok(C), "bn_fwd: bad args (C must divide 256 or be 512)");
    SCL_REQUIRE(training ? part != nullptr : (running_mean && running_var), "bn_fwd: training needs `part`, eval needs running statistics");
    SCL_REQUIRE(act >= 0 && act <= 2, "bn_fwd: act");
    hipStream_t s = (hipStream_t)stream;
    const int nslab = scl_bn_nslabs(N);
    if (training) hipLaunchKernelGGL(bn_stats_kernel, dim3(nslab), dim3(256), 0, s, x, N, C, (double*)part, bn_slab_rows(N));
    hipLaunchKernelGGL(bn_finish_kernel, dim3((C + 63) / 64), dim3(1024), 0, s, (const double*)part, nslab, N, C, eps, momentum, training, running_mean, running_var,
                       num_batches_tracked, mean, rstd);
    const RowMap map = {m_W > 0 ? m_W : 1, m_HW > 0 ? m_HW : 1, m_bs, m_rs, m_cs, m_base};
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for((long long)N * C)), dim3(256), 0, s, x, mean, rstd, gamma, beta, (long long)N * C, C, act, y, y2,
                       y2_bf16, map);
    return scl_check_launch("scl_bn_fwd");
}

extern "C" int scl_bn_bwd(const float* dy, const float* y, const float* x, const float* mean, const float* rstd, const float* gamma, int N, int C,
                          int act, int training, float* part, float* sums, float* dgamma, float* dbeta, float* dx, int accumulate, void* stream) {
    SCL_REQUIRE(dy && x && mean && rstd && part && sums && dx && N >= 1 && bn_channels_ok(C), "bn_bwd: bad args");
    SCL_REQUIRE(act == 0 || y, "bn_bwd: the activation gradient needs the forward output y");
    hipStream_t s = (hipStream_t)stream;
    const int nslab = scl_bn_nslabs(N);
    hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3(nslab), dim3(256), 0, s, dy, y, x, mean, rstd, N, C, act, (double*)part, bn_slab_rows(N));
    hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3((C + 63) / 64), dim3(1024), 0, s, (const double*)part, nslab, C, dgamma, dbeta, sums, accumulate);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for((long long)N * C)), dim3(256), 0, s, dy, y, x, mean, rstd, gamma, sums, (long long)N * C, N, C,
                       act, training, dx);
    return scl_check_launch("scl_bn_bwd");
}

extern "C" int scl_pad_nhwc_f32(const float* src, int64_t rows, int C, void* dst, int dst_bf16, int m_W, int m_HW, int64_t m_bs, int64_t m_rs,
                                int64_t m_cs, int64_t m_base, void* stream) {
    SCL_REQUIRE(src && dst && rows >= 1 && C >= 1 && m_W >= 1 && m_HW >= 1, "pad_nhwc: bad args");
    const RowMap map = {m_W, m_HW, m_bs, m_rs, m_cs, m_base};
    const bool vec = (C & 3) == 0 && !dst_bf16;      // the kernel's 16-byte path: a quarter of the threads
    hipLaunchKernelGGL(pad_nhwc_kernel, dim3(grid_for(vec ? rows * C / 4 : rows * C)), dim3(256), 0, (hipStream_t)stream, src, rows * C, C, dst, dst_bf16, map);
    return scl_check_launch("scl_pad_nhwc_f32");
}

// Re-layout of one convolution's weights for the implicit-GEMM kernels, both copies in one launch (hipnn.py's _packed cache):
//   fwd [Co][kh][kw][Cp]  = w[co][c][r][s]            (forward and weight-gradient operand; channels c >= Ci are zero)
//   bwd [Ci][kh][kw][Cop] = w[co][c][kh-1-r][kw-1-s]  (data-gradient operand: flipped taps, in/out swapped; channels co >= Co are zero)
// Replaces five torch kernels per convolution and step (two fills, flip, two strided copies: 0.75 ms of the ResNet step at batch 32).
__global__ __launch_bounds__(256) void conv_pack_kernel(const float* __restrict__ w, float* __restrict__ fwd, float* __restrict__ bwd, int Co, int Ci, int kh, int kw,
                                                        int Cp, int Cop) {
    const long long nf = (long long)Co * kh * kw * Cp, nb = (long long)Ci * kh * kw * Cop;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < nf + nb; i += gridDim.x * 256ll) {
        if (i < nf) {
            const int c = (int)(i % Cp); long long t = i / Cp;
            const int s_ = (int)(t % kw); t /= kw;
            const int r = (int)(t % kh); const int co = (int)(t / kh);
            fwd[i] = c < Ci ? w[(((long long)co * Ci + c) * kh + r) * kw + s_] : 0.f;
        } else {
            const long long j = i - nf;
            const int co = (int)(j % Cop); long long t = j / Cop;
            const int s_ = (int)(t % kw); t /= kw;
            const int r = (int)(t % kh); const int c = (int)(t / kh);
            bwd[j] = co < Co ? w[(((long long)co * Ci + c) * kh + (kh - 1 - r)) * kw + (kw - 1 - s_)] : 0.f;
        }
    }
}

extern "C" int scl_conv_pack_weights(const float* w, float* fwd, float* bwd, int Co, int Ci, int kh, int kw, int Cp, int Cop, void* stream) {
    SCL_REQUIRE(w && fwd && bwd && Co >= 1 && Ci >= 1 && kh >= 1 && kw >= 1 && Cp >= Ci && Cop >= Co, "conv_pack_weights: bad args");
    const long long n = (long long)Co * kh * kw * Cp + (long long)Ci * kh * kw * Cop;
    hipLaunchKernelGGL(conv_pack_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, w, fwd, bwd, Co, Ci, kh, kw, Cp, Cop);
    return scl_check_launch("scl_conv_pack_weights");
}

// Weight gradient of a convolution out of its per-utterance slabs: grad[co][c][r][s] += sum_z slabs[z][co][(r*kw+s)*Cp + c] — the fixed-order
// slab sum, the [Co][kh][kw][Cp] -> torch-layout permute and autograd's accumulation into the parameter's .grad in one pass.
__global__ __launch_bounds__(256) void conv_wgrad_finish_kernel(const float* __restrict__ slabs, float* __restrict__ grad, int nslab, int Co, int Ci, int kh, int kw,
                                                                int Cp, int accumulate) {
    // threads run over the SLAB layout, four channels each (Cp % 4 == 0: hipnn pads to the 16-byte vector) and FOUR ADJACENT LANES per
    // channel quad: lane s sums the slabs z = s (mod 4) with coalesced 16-byte loads, two shuffles combine them (a fixed order), lane 0
    // does the four strided writes.  With up to 113 slabs behind a small [Co, K] image, one thread per quad walked them as one latency chain.
    const unsigned C4 = (unsigned)Cp >> 2, taps = (unsigned)(kh * kw), n4 = (unsigned)Co * taps * C4;
    const long long slab = (long long)Co * taps * Cp;
    const unsigned total = (n4 + 63u) / 64u * 256u;          // whole 4-lane groups
    for (unsigned tq = blockIdx.x * 256u + threadIdx.x; tq < total; tq += gridDim.x * 256u) {
        const unsigned q = tq >> 2, zs = tq & 3u;
        const bool live = q < n4;
        const unsigned c = 4 * (q % C4), t = q / C4, rs = t % taps, co = t / taps;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live && c < (unsigned)Ci) {
            const float* src = slabs + 4ll * q;
            for (int z = (int)zs; z < nslab; z += 4) {
                const float4 v = *reinterpret_cast<const float4*>(src + z * slab);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        acc.x += __shfl_xor(acc.x, 1); acc.y += __shfl_xor(acc.y, 1); acc.z += __shfl_xor(acc.z, 1); acc.w += __shfl_xor(acc.w, 1);
        acc.x += __shfl_xor(acc.x, 2); acc.y += __shfl_xor(acc.y, 2); acc.z += __shfl_xor(acc.z, 2); acc.w += __shfl_xor(acc.w, 2);
        if (zs != 0 || !live || c >= (unsigned)Ci) continue;
        float* g = grad + ((long long)co * Ci + c) * taps + rs;
        const float a4[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c + e < (unsigned)Ci) g[(long long)e * taps] = accumulate ? g[(long long)e * taps] + a4[e] : a4[e];
    }
}

extern "C" int scl_conv_wgrad_finish(const float* slabs, float* grad, int nslab, int Co, int Ci, int kh, int kw, int Cp, int accumulate, void* stream) {
    SCL_REQUIRE(slabs && grad && nslab >= 1 && Co >= 1 && Ci >= 1 && kh >= 1 && kw >= 1 && Cp >= Ci && (Cp & 3) == 0 && ((uintptr_t)slabs & 15) == 0 &&
                (long long)Co * kh * kw * Cp < (1ll << 31), "conv_wgrad_finish: bad args (Cp a multiple of 4, 16-byte aligned slabs of < 2^31 elements)");
    hipLaunchKernelGGL(conv_wgrad_finish_kernel, dim3(grid_for((long long)Co * kh * kw * Cp)), dim3(256), 0, (hipStream_t)stream, slabs, grad, nslab, Co, Ci, kh, kw,
                       Cp, accumulate);
    return scl_check_launch("scl_conv_wgrad_finish");
}

extern "C" int scl_maxpool3_fwd(const float* x, int64_t xs_h, int64_t xs_w, int64_t xs_b, int H, int W, int B, float* y, int* idx, void* stream) {
    SCL_REQUIRE(x && y && idx && H >= 3 && W >= 3 && B >= 1, "maxpool3_fwd: bad args");
    hipLaunchKernelGGL(maxpool3_fwd_kernel, dim3(grid_for((long long)B * (H / 3) * (W / 3))), dim3(256), 0, (hipStream_t)stream, x, (long long)xs_h,
                       (long long)xs_w, (long long)xs_b, H, W, B, y, idx);
    return scl_check_launch("scl_maxpool3_fwd");
}
extern "C" int scl_maxpool3_bwd(const float* dy, const int* idx, int H, int W, int B, float* dx, int64_t xs_h, int64_t xs_w, int64_t xs_b, void* stream) {
    SCL_REQUIRE(dy && idx && dx && H >= 3 && W >= 3 && B >= 1, "maxpool3_bwd: bad args");
    hipLaunchKernelGGL(maxpool3_bwd_kernel, dim3(grid_for((long long)B * (H / 3) * (W / 3))), dim3(256), 0, (hipStream_t)stream, dy, idx, H, W, B, dx,
                       (long long)xs_h, (long long)xs_w, (long long)xs_b);
    return scl_check_launch("scl_maxpool3_bwd");
}

extern "C" int scl_avgpool_fwd(const float* x, int B, int R, int C, float* y, void* stream) {
    SCL_REQUIRE(x && y && B >= 1 && R >= 1 && C >= 1, "avgpool_fwd: bad args");
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, R, C, y);
    return scl_check_launch("scl_avgpool_fwd");
}
extern "C" int scl_avgpool_bwd(const float* dy, int B, int R, int C, float* dx, void* stream) {
    SCL_REQUIRE(dy && dx && B >= 1 && R >= 1 && C >= 1, "avgpool_bwd: bad args");
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(grid_for((long long)B * R * C)), dim3(256), 0, (hipStream_t)stream, dy, R, C, (long long)B * R * C, dx);
    return scl_check_launch("scl_avgpool_bwd");
}
