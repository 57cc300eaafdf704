// conv0.hip — first layer of the wav2vec2 feature extractor, forward and backward, fully fused:
//   z[b][t][c] = GELU( LayerNorm_c( bias[c] + sum_j w[c][j] * x[b][stride*t + j] ) )
// (fairseq ConvFeatureExtractionModel layer 0: Conv1d(1, C, k=10, stride=5, bias) -> Fp32LayerNorm
// over channels -> GELU, reached from model/xlsr.py:41).  C_in = 1 makes this layer HBM-bound
// (output [B, 12799, 512] bf16 = 13 MB per utterance), so it is not a GEMM: one wave owns one frame,
// 8 channels per lane, taps and the waveform slab staged in LDS, LayerNorm statistics by wave
// shuffles, 16-byte coalesced bf16 stores.  The backward recomputes the pre-norm activation from the
// waveform instead of saving it and reduces dW / dbias / dgamma / dbeta per block, deterministically.
#include "common.h"

namespace {

constexpr int MAXK = 16;
constexpr int MAXCH0 = 2;  // C <= 1024

struct Conv0Smem {
    float* wT;   // [k][C]
    float* xs;   // [rows*stride + k]
    float* red;  // [C*(k+3)]  (backward only)
};

__device__ __forceinline__ void conv0_stage(float* wT, float* xs, const float* __restrict__ w, const float* __restrict__ x,
                                            int C, int k, int L, int x0, int nx) {
    for (int i = threadIdx.x; i < C * k; i += blockDim.x) {
        const int c = i / k, j = i % k;
        wT[j * C + c] = w[i];
    }
    for (int i = threadIdx.x; i < nx; i += blockDim.x) xs[i] = (x0 + i) < L ? x[x0 + i] : 0.f;
}

// KC: the kernel width as a compile-time constant (10 = wav2vec 2.0's layer 0; 0 = run time).  With a run-time width the tap loop stays
// rolled — ten trips of (3 LDS reads, 8 FMAs, a branch) per frame.
template <bool ZF32, int KC>     // ZF32: fp32 output (the scoring path keeps activations in fp32), else bf16
__global__ __launch_bounds__(256) void conv0_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, void* __restrict__ zv, float* __restrict__ stats,
                                                        int L, int T0, int C, int k, int stride, int rows_per_block, float eps) {
    extern __shared__ __attribute__((aligned(16))) float sm0[];
    float* wT = sm0;
    float* xs = sm0 + k * C;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * rows_per_block;
    const int nrows = min(rows_per_block, T0 - t0);
    const int nx = nrows * stride + k;
    conv0_stage(wT, xs, w, x + (int64_t)b * L, C, k, L, t0 * stride, nx);
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float bia[MAXCH0][8], gam[MAXCH0][8], bet[MAXCH0][8];      // this lane's channels of bias / gamma / beta: loop-invariant over the wave's frames
#pragma unroll
    for (int ch = 0; ch < MAXCH0; ++ch) {
        const int c = ch * 512 + lane * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            bia[ch][i] = c < C ? bias[c + i] : 0.f; gam[ch][i] = c < C ? gamma[c + i] : 0.f; bet[ch][i] = c < C ? beta[c + i] : 0.f;
        }
    }
    for (int r = wv; r < nrows; r += 4) {
        float y[MAXCH0][8];
        float s = 0.f;
#pragma unroll
        for (int ch = 0; ch < MAXCH0; ++ch) {
            const int c = ch * 512 + lane * 8;
            if (c < C) {
#pragma unroll
                for (int i = 0; i < 8; ++i) y[ch][i] = bia[ch][i];
#pragma unroll
                for (int j = 0; j < (KC ? KC : k); ++j) {
                    const float xv = xs[r * stride + j];
                    const float4 w0 = *reinterpret_cast<const float4*>(wT + j * C + c);
                    const float4 w1 = *reinterpret_cast<const float4*>(wT + j * C + c + 4);
                    y[ch][0] += w0.x * xv; y[ch][1] += w0.y * xv; y[ch][2] += w0.z * xv; y[ch][3] += w0.w * xv;
                    y[ch][4] += w1.x * xv; y[ch][5] += w1.y * xv; y[ch][6] += w1.z * xv; y[ch][7] += w1.w * xv;
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) s += y[ch][i];
            }
        }
        const float mean = wave_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int ch = 0; ch < MAXCH0; ++ch) {
            const int c = ch * 512 + lane * 8;
            if (c < C) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { const float d = y[ch][i] - mean; q += d * d; }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
        if (stats && lane == 0) {
            stats[2 * ((int64_t)b * T0 + t0 + r)] = mean;
            stats[2 * ((int64_t)b * T0 + t0 + r) + 1] = rstd;
        }
#pragma unroll
        for (int ch = 0; ch < MAXCH0; ++ch) {
            const int c = ch * 512 + lane * 8;
            if (c < C) {
                float o[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = (y[ch][i] - mean) * rstd * gam[ch][i] + bet[ch][i];
#pragma unroll
                for (int i = 0; i < 8; i += 2) gelu2(o[i], o[i + 1]);      // two elements per packed-f32 instruction, same bits as gelu_f
                const int64_t off = ((int64_t)b * T0 + t0 + r) * C + c;
                if (ZF32) {
                    float* z = reinterpret_cast<float*>(zv);
                    *reinterpret_cast<float4*>(z + off) = make_float4(o[0], o[1], o[2], o[3]);
                    *reinterpret_cast<float4*>(z + off + 4) = make_float4(o[4], o[5], o[6], o[7]);
                } else {
                    uint4 u;
                    u.x = pack_bf2(o[0], o[1]); u.y = pack_bf2(o[2], o[3]); u.z = pack_bf2(o[4], o[5]); u.w = pack_bf2(o[6], o[7]);
                    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(zv) + off) = u;
                }
            }
        }
    }
}

// part[blk][c*(k+3) + q]: q < k -> dW[c][q]; q == k -> dbias; k+1 -> dgamma; k+2 -> dbeta
// STATS: the forward's per-frame (mean, rstd) are read back (8 B per frame) instead of being recomputed with two more
// wave reductions; the 4-wave combine buffer re-uses the tap / waveform LDS once the frame loop is done (30 KiB per block
// instead of 57 KiB; the 104 accumulators per lane keep it at 2 waves per SIMD).
// KT: compile-time bound of the tap loops (10 for wav2vec2's layer 0: exactly 13 accumulators per channel; 16 = generic)
// FULLK: k == KT, so the per-tap `j < k` tests (30 uniform branches per frame in the ISA) are compiled out
template <bool STATS, int KT, bool FULLK>
__global__ __launch_bounds__(256, 2) void conv0_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const bf16_t* __restrict__ dz,
                                                           const float* __restrict__ stats, float* __restrict__ part, int L, int T0, int C,
                                                           int k, int stride, int rows_per_block, float eps) {
    extern __shared__ __attribute__((aligned(16))) float sm0[];
    float* wT = sm0;
    float* xs = sm0 + k * C;
    float* red = sm0;                 // valid only after the frame loop (barrier below)
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * rows_per_block;
    const int nrows = min(rows_per_block, T0 - t0);
    const int nx = nrows * stride + k;
    conv0_stage(wT, xs, w, x + (int64_t)b * L, C, k, L, t0 * stride, nx);
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float aw[8][KT + 3];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int q = 0; q < KT + 3; ++q) aw[i][q] = 0.f;
    const int c = lane * 8;
    const bool act = c < C;
    // the frame's dz vector and (mean, rstd) are requested one frame ahead: with two waves per SIMD (216 registers) an un-prefetched
    // HBM round trip per frame was fully exposed
    float gam[8], bet[8], bia[8];      // this lane's 8 channels of the affine / bias vectors: loop-invariant, kept in registers
#pragma unroll
    for (int i = 0; i < 8; ++i) { gam[i] = act ? gamma[c + i] : 0.f; bet[i] = act ? beta[c + i] : 0.f; bia[i] = act ? bias[c + i] : 0.f; }
    uint4 u_next = make_uint4(0, 0, 0, 0);
    float mean_next = 0.f, rstd_next = 0.f;
    if (wv < nrows) {
        const int64_t row0 = (int64_t)b * T0 + t0 + wv;
        if (act) u_next = *reinterpret_cast<const uint4*>(dz + row0 * C + c);
        if (STATS) { mean_next = stats[2 * row0]; rstd_next = stats[2 * row0 + 1]; }
    }
    for (int r = wv; r < nrows; r += 4) {
        float y[8], g8[8], dzv[8], xr[KT];
        const int64_t row = (int64_t)b * T0 + t0 + r;
        const uint4 u = u_next;
        const float mean_cur = mean_next, rstd_cur = rstd_next;
        if (r + 4 < nrows) {
            if (act) u_next = *reinterpret_cast<const uint4*>(dz + (row + 4) * C + c);
            if (STATS) { mean_next = stats[2 * (row + 4)]; rstd_next = stats[2 * (row + 4) + 1]; }
        }
#pragma unroll
        for (int j = 0; j < KT; ++j) xr[j] = (FULLK || j < k) ? xs[r * stride + j] : 0.f;
        if (act) {
#pragma unroll
            for (int i = 0; i < 8; ++i) y[i] = bia[i];
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                if (j < k) {      // kept as a run-time test even with FULLK: as straight-line code the 20 weight vectors are all requested
                                  // first (80 registers on top of the 104 accumulators: spills and a private segment)
                    const float4 w0 = *reinterpret_cast<const float4*>(wT + j * C + c);
                    const float4 w1 = *reinterpret_cast<const float4*>(wT + j * C + c + 4);
                    y[0] += w0.x * xr[j]; y[1] += w0.y * xr[j]; y[2] += w0.z * xr[j]; y[3] += w0.w * xr[j];
                    y[4] += w1.x * xr[j]; y[5] += w1.y * xr[j]; y[6] += w1.z * xr[j]; y[7] += w1.w * xr[j];
                }

            }
            const uint32_t uw[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) { dzv[2 * i] = __uint_as_float(uw[i] << 16); dzv[2 * i + 1] = __uint_as_float(uw[i] & 0xFFFF0000u); }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) { y[i] = 0.f; dzv[i] = 0.f; }
        }
        float mean, rstd;
        if (STATS) {
            mean = mean_cur; rstd = rstd_cur;
        } else {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) s += y[i];
            mean = wave_sum(s) / (float)C;
            float q = 0.f;
            if (act) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { const float dd = y[i] - mean; q += dd * dd; }
            }
            rstd = rsqrtf(wave_sum(q) / (float)C + eps);
        }
        float s1 = 0.f, s2 = 0.f;
        if (act) {
            float gg[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) gg[i] = ((y[i] - mean) * rstd) * gam[i] + bet[i];
#pragma unroll
            for (int i = 0; i < 8; i += 2) gelu_grad2(gg[i], gg[i + 1]);      // packed-f32 form, same bits as gelu_grad_f
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float h = (y[i] - mean) * rstd;
                const float gmi = gam[i];
                const float dyn = dzv[i] * gg[i];
                aw[i][KT + 1] += dyn * h;  // dgamma
                aw[i][KT + 2] += dyn;      // dbeta
                const float dh = dyn * gmi;
                y[i] = h;      // xhat
                g8[i] = dh;    // dxhat
                s1 += dh; s2 += dh * h;
            }
        }
        s1 = wave_sum(s1) / (float)C;
        s2 = wave_sum(s2) / (float)C;
        if (act) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float dy = rstd * (g8[i] - s1 - y[i] * s2);
                aw[i][KT] += dy;  // dbias
#pragma unroll
                for (int j = 0; j < KT; ++j)
                    if (FULLK || j < k) aw[i][j] += dy * xr[j];
            }
        }
    }
    __syncthreads();   // every wave is done with wT / xs: the combine buffer takes their place
    for (int i = threadIdx.x; i < C * (k + 3); i += blockDim.x) red[i] = 0.f;
    __syncthreads();
    // deterministic combine of the 4 waves
    for (int ww = 0; ww < 4; ++ww) {
        if (wv == ww && act) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < KT; ++j)
                    if (FULLK || j < k) red[(c + i) * (k + 3) + j] += aw[i][j];
                red[(c + i) * (k + 3) + k] += aw[i][KT];
                red[(c + i) * (k + 3) + k + 1] += aw[i][KT + 1];
                red[(c + i) * (k + 3) + k + 2] += aw[i][KT + 2];
            }
        }
        __syncthreads();
    }
    const int64_t blk = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    for (int i = threadIdx.x; i < C * (k + 3); i += blockDim.x) part[blk * C * (k + 3) + i] = red[i];
}

// First level of the partial-sum tree: segment g adds rows g, g + S, g + 2S, ... into row g (in place: no other segment touches
// those rows).  At batch 64 there are 832 partial rows of 6656 values; a single level walks them serially from 26 workgroups (250 us).
__global__ void conv0_reduce_seg_kernel(float* __restrict__ part, int nparts, int n, int S) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int g = blockIdx.y;
    if (i >= n || g >= nparts) return;
    float s = 0.f;
    for (int p = g; p < nparts; p += S) s += part[(int64_t)p * n + i];
    part[(int64_t)g * n + i] = s;
}

__global__ void conv0_reduce_kernel(const float* __restrict__ part, int nparts, int C, int k, float* __restrict__ dW,
                                    float* __restrict__ db, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = C * (k + 3);
    if (i >= n) return;
    float s = 0.f;
    for (int p = 0; p < nparts; ++p) s += part[(int64_t)p * n + i];
    const int c = i / (k + 3), q = i % (k + 3);
    if (q < k) dW[c * k + q] = s;
    else if (q == k) db[c] = s;
    else if (q == k + 1) dgamma[c] = s;
    else dbeta[c] = s;
}

}  // namespace

// frames per backward block.  Two blocks are resident per CU (216 registers): 512 slots, every block of a launch runs equally long, so the
// launch costs ceil(blocks / 512) rounds x (rows + a fixed part: staging the taps, the 4-wave combine, one partial row).  The first form
// aimed at "~512 blocks" with rows <= 1024 and got 13 chunks x 64 utterances = 832 blocks: two rounds of 1024 frames where 16 chunks are two
// rounds of 800.  Pick the chunk count with the smallest cost (rows 64 ... 1024).
static int conv0_bwd_rows(int B, int T0) {
    int best_rows = 1024; long long best = -1;
    for (int chunks = (T0 + 1023) / 1024; chunks <= (T0 + 63) / 64; ++chunks) {
        const int rows = (T0 + chunks - 1) / chunks;
        const long long blocks = (long long)B * ((T0 + rows - 1) / rows);
        const long long cost = ((blocks + 511) / 512) * (rows + 48);
        if (best < 0 || cost < best) { best = cost; best_rows = rows; }
        if (blocks > 8192) break;      // the partial-sum tree is sized for a few thousand rows
    }
    return best_rows < 64 ? 64 : best_rows;
}

extern "C" int scl_conv0_fwd(const float* x, const float* w, const float* bias, const float* gamma, const float* beta,
                             void* z, float* stats, int B, int L, int C, int k, int stride, float eps, void* stream) {
    SCL_REQUIRE(x && w && bias && gamma && beta && z, "conv0_fwd: null pointer");
    SCL_REQUIRE(B > 0 && L >= k && C >= 8 && C <= 1024 && (C & 7) == 0 && k >= 1 && k <= MAXK && stride >= 1, "conv0_fwd: bad dims");
    const int T0 = (L - k) / stride + 1;
    const int rows = 128;
    dim3 grid((T0 + rows - 1) / rows, B), block(256);
    const size_t lds = (size_t)(k * C + rows * stride + k) * sizeof(float);
    if (k == 10) hipLaunchKernelGGL((conv0_fwd_kernel<false, 10>), grid, block, lds, (hipStream_t)stream, x, w, bias, gamma, beta, z, stats, L, T0, C, k, stride, rows, eps);
    else hipLaunchKernelGGL((conv0_fwd_kernel<false, 0>), grid, block, lds, (hipStream_t)stream, x, w, bias, gamma, beta, z, stats, L, T0, C, k, stride, rows, eps);
    return scl_check_launch("scl_conv0_fwd");
}

extern "C" int scl_conv0_fwd_f32(const float* x, const float* w, const float* bias, const float* gamma, const float* beta,
                                 float* z, int B, int L, int C, int k, int stride, float eps, void* stream) {
    SCL_REQUIRE(x && w && bias && gamma && beta && z, "conv0_fwd_f32: null pointer");
    SCL_REQUIRE(B > 0 && L >= k && C >= 8 && C <= 1024 && (C & 7) == 0 && k >= 1 && k <= MAXK && stride >= 1, "conv0_fwd_f32: bad dims");
    const int T0 = (L - k) / stride + 1;
    const int rows = 128;
    dim3 grid((T0 + rows - 1) / rows, B), block(256);
    const size_t lds = (size_t)(k * C + rows * stride + k) * sizeof(float);
    if (k == 10) hipLaunchKernelGGL((conv0_fwd_kernel<true, 10>), grid, block, lds, (hipStream_t)stream, x, w, bias, gamma, beta, (void*)z, nullptr, L, T0, C, k, stride, rows, eps);
    else hipLaunchKernelGGL((conv0_fwd_kernel<true, 0>), grid, block, lds, (hipStream_t)stream, x, w, bias, gamma, beta, (void*)z, nullptr, L, T0, C, k, stride, rows, eps);
    return scl_check_launch("scl_conv0_fwd_f32");
}

extern "C" int scl_conv0_bwd_nparts(int B, int L, int k, int stride) {
    const int T0 = (L - k) / stride + 1;
    const int rows = conv0_bwd_rows(B, T0);
    return B * ((T0 + rows - 1) / rows);
}

extern "C" int scl_conv0_bwd(const float* x, const float* w, const float* bias, const float* gamma, const float* beta,
                             const void* dz, const float* stats, float* part_ws, float* dW, float* db, float* dgamma, float* dbeta,
                             int B, int L, int C, int k, int stride, float eps, void* stream) {
    SCL_REQUIRE(x && w && bias && gamma && beta && dz && part_ws && dW && db && dgamma && dbeta, "conv0_bwd: null pointer");
    SCL_REQUIRE(B > 0 && L >= k && C >= 8 && C <= 512 && (C & 7) == 0 && k >= 1 && k <= MAXK && stride >= 1, "conv0_bwd: bad dims (C <= 512)");
    const int T0 = (L - k) / stride + 1;
    const int rows = conv0_bwd_rows(B, T0);
    dim3 grid((T0 + rows - 1) / rows, B), block(256);
    const size_t a = (size_t)(k * C + rows * stride + k), r = (size_t)C * (k + 3);
    const size_t lds = (a > r ? a : r) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
#define CONV0_BWD(ST, KT, FK) hipLaunchKernelGGL((conv0_bwd_kernel<ST, KT, FK>), grid, block, lds, s, x, w, bias, gamma, beta, (const bf16_t*)dz, stats, \
                                                 part_ws, L, T0, C, k, stride, rows, eps)
    if (k == 10) { if (stats) CONV0_BWD(true, 10, true); else CONV0_BWD(false, 10, true); }
    else if (k < 10) { if (stats) CONV0_BWD(true, 10, false); else CONV0_BWD(false, 10, false); }
    else { if (stats) CONV0_BWD(true, MAXK, false); else CONV0_BWD(false, MAXK, false); }
#undef CONV0_BWD
    int nparts = grid.x * grid.y;
    const int n = C * (k + 3);
    if (nparts > 64) {
        const int S = 32;
        hipLaunchKernelGGL(conv0_reduce_seg_kernel, dim3((n + 255) / 256, S), dim3(256), 0, s, part_ws, nparts, n, S);
        nparts = S;
    }
    hipLaunchKernelGGL(conv0_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, s, part_ws, nparts, C, k, dW, db, dgamma, dbeta);
    return scl_check_launch("scl_conv0_bwd");
}
