// resstack.hip — the RawNet2-style encoder of the AASIST back-end (six Residual_blocks, model/wav2vec2_aasist.py:377-433, stacked at
// :470-476) as hand-scheduled gfx950 kernels, forward and backward.
//
// Layout.  Every map of the stack lives ZERO-BORDERED and FLAT: utterance b owns RPU = H + 2 rows of Wp = W + 2 positions of C
// channels (fp32, channels last); position g = (b * RPU + r) * Wp + c.  With the borders in the buffer a (kh, kw) tap of a stride-1
// convolution is ONE flat shift: out[g] = sum_t sum_c in[g + s_t][c] * W[t][c][n] for every g, garbage only where g is a border
// position — and those are written as zeros (they ARE the next convolution's padding).  No im2col, no padded copy per call, and
// the data gradient of a convolution is the same kernel with the shifts negated and the weights transposed.
//
// Kernels (all exact fp32: v_mfma_f32_16x16x4_f32, 64 FLOP/clk/SIMD = the f32 roof of the chip, 157 TFLOP/s):
//   rs_conv_kernel<CIN, COUT, NT>   persistent blocks; a wave keeps its 16 output channels' weights (NT * CIN values per lane) in
//                                   REGISTERS for the whole launch, 128-position input tiles (+ halo) stream through LDS, each
//                                   ds_read_b128 feeds four MFMAs.  Epilogue through LDS: + bias, + addend (identity / down-sample
//                                   branch), border mask, optional x selu'(a) (BatchNorm backward input), per-channel statistics
//                                   (sum, sum of squares | sum dz, sum dz * xhat) — combined over blocks with fp64 atomics and finished by
//                                   the last block: BatchNorm mean / rstd / running statistics, or dgamma / dbeta and the two means of
//                                   the BatchNorm backward.
//   rs_wgrad_kernel<CIN, COUT, NT>  dW[t][c][n] = sum_g in[g + s_t][c] * dout[g][n]: the whole [NT * CIN, COUT] gradient stays in the
//                                   accumulators (96 registers per lane at 64 -> 64 channels, six taps) while 64-position chunks of
//                                   both maps stream through LDS; one partial slab per block and wave group, summed in a fixed order
//                                   by rs_wgrad_reduce_kernel straight into the torch-layout gradient; bias gradient alongside.
//   rs_bn_act / rs_bn_bwd_apply     the two element-wise passes BatchNorm's batch statistics force between the convolutions.
//   rs_scatter_c1 / rs_gather_c1 / rs_unpad / rs_pad    dense <-> bordered copies at the two ends of the stack.
#include "rs_finish.h"

namespace {

// output positions per tile of the convolution kernel: 64 input channels keep 96 weight registers per lane, so that form runs ONE block
// per CU with all 512 registers (256-position tiles, the next tile's 84 registers of input in flight); narrower inputs run two blocks
// per CU on 128-position tiles
__host__ __device__ constexpr int rs_mt(int cin) { return cin >= 64 ? 256 : 128; }
// positions per chunk of the weight-gradient kernel: the 64 -> 64 form (96 accumulator + 50 fragment registers per lane) runs one block
// per CU on 128-position chunks, the narrower ones two blocks per CU on 64-position chunks
__host__ __device__ constexpr int rs_pc(int cin, int cout) { return cin * cout >= 4096 ? 128 : 64; }
__host__ __device__ constexpr int rs_wgrad_blocks(int cin, int cout) { return cin * cout >= 4096 ? 256 : 512; }
constexpr int RS_MAX_SPAN = 96; // largest halo (W + 3 positions) the register prefetch is sized for: maps up to 93 positions wide

struct RsGeom { int G, Wp, RPU, W, r_lo, r_hi; unsigned m_wp, m_rpu; };      // m_*: ceil(65536 / x), exact quotients for dividends < 65536 / x

__device__ __forceinline__ bool rs_valid(const RsGeom& q, unsigned g) {
    if (g >= (unsigned)q.G) return false;
    const unsigned row = g / (unsigned)q.Wp;
    const int c = (int)(g - row * (unsigned)q.Wp);
    const int r = (int)(row % (unsigned)q.RPU);
    return r >= q.r_lo && r <= q.r_hi && c >= 1 && c <= q.W;
}

// The same test for position g0 + pos of a tile whose first position sits at (row0 % RPU, col0): two 16-bit magic divisions of small
// numbers instead of two 32-bit integer divisions per position (they were most of the epilogue's instructions).
__device__ __forceinline__ bool rs_valid_rel(const RsGeom& q, unsigned rr0, unsigned col0, unsigned pos) {
    const unsigned x = col0 + pos;                        // < Wp + 256
    const unsigned dq = (x * q.m_wp) >> 16;               // x / Wp
    const int c = (int)(x - dq * (unsigned)q.Wp);
    const unsigned y = rr0 + dq;                          // < RPU + 256 / Wp + 1
    const int r = (int)(y - ((y * q.m_rpu) >> 16) * (unsigned)q.RPU);
    return r >= q.r_lo && r <= q.r_hi && c >= 1 && c <= q.W;
}

struct RsConvK {
    const float* in; const float* wpk; const float* bias; const float* addend; float* out;
    const float* act_a; const float* y1; const float* bnstats;     // stat_mode 2: out = conv * selu'(act_a); xhat = (y1 - mean) * rstd
    double* acc; unsigned* ticket;                                   // [2 * COUT] fp64 accumulators (zero between launches) + arrival counter
    const float* gamma; const float* beta; float* run_mean; float* run_var; long long* nbt;
    float* stats_out; float* dgamma; float* dbeta;
    RsGeom q;
    int shift[6];
    int smin, span, stat_mode, training;
    int epi_act;        // 1: the stored value (and its statistics) is selu(conv + bias + addend)   — the attention block's conv -> SELU -> BatchNorm
    int act_a_none;     // statistics mode 2 without the activation factor: out = conv (the gradient of a BatchNorm OUTPUT), xhat from y1
    float eps, momentum;
    double nvalid;
};

template <int CIN, int COUT, int NT, int SM>      // SM: statistics mode of the epilogue (0 none, 1 BatchNorm forward, 2 BatchNorm + SELU backward)
__global__ __launch_bounds__(256, (CIN >= 64 ? 1 : 2)) void rs_conv_kernel(const RsConvK d) {
    constexpr int RS_MT = rs_mt(CIN);
    constexpr int NCB = COUT / 16;            // 16-channel output blocks: one per wave (64), two waves per block (32), four (16)
    constexpr int NRG = 4 / NCB;              // wave groups that split the tile's 16-position row blocks
    constexpr int RPW = (RS_MT / 16) / NRG;   // row blocks per wave
    constexpr int RBG = RPW < 4 ? RPW : 4;    // row blocks in flight (independent accumulators: >= 2 covers the 40-cycle MFMA latency)
    constexpr int NGRP = RPW / RBG;
    constexpr int NJJ = CIN / 16;
    constexpr int PITCH = CIN + 4;            // 16 rows x 16-byte reads hit 64 distinct banks: (CIN / 4 + 1) is odd
    constexpr int OPITCH = COUT + 4;
    constexpr int C4 = CIN / 4, N4 = COUT / 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cb = wave % NCB, rg = wave / NCB, g4 = lane >> 4, li = lane & 15;

    f32x4 w[NT][NJJ];
    {
        const f32x4* wp = reinterpret_cast<const f32x4*>(d.wpk) + (size_t)cb * NT * NJJ * 64 + lane;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int jj = 0; jj < NJJ; ++jj) w[t][jj] = wp[(t * NJJ + jj) * 64];
    }
    const int ch4 = tid % N4;                 // the four output channels this thread owns in every epilogue pass (256 % N4 == 0)
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, mean4 = bias4, rstd4 = bias4;
    if (d.bias) bias4 = *reinterpret_cast<const f32x4*>(d.bias + 4 * ch4);
    if (SM == 2) {
        mean4 = *reinterpret_cast<const f32x4*>(d.bnstats + 4 * ch4);
        rstd4 = *reinterpret_cast<const f32x4*>(d.bnstats + COUT + 4 * ch4);
    }
    double st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int ntiles = (d.q.G + RS_MT - 1) / RS_MT;
    const int nf4 = (RS_MT + d.span) * C4;      // 16-byte pieces of a tile + halo: one contiguous range of the flat map
    // The NEXT tile's input travels to registers while this one is multiplied (a load -> ds_write loop inside the tile pays one memory
    // latency per trip: 13 trips x ~1 us against 12 us of MFMAs); the epilogue's operand reads go out four passes at a time.
    constexpr int NLD = ((RS_MT + RS_MAX_SPAN) * C4 + 255) / 256;
    f32x4 pre[NLD];
    auto issue = [&](int tile) {
        const float* src = d.in + ((long long)tile * RS_MT + d.smin) * CIN;
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int f = tid + 256 * u;
            if (f < nf4) pre[u] = *reinterpret_cast<const f32x4*>(src + (size_t)f * 4);
        }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) issue(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        const int g0 = tile * RS_MT;
        const unsigned row0 = (unsigned)g0 / (unsigned)d.q.Wp, col0 = (unsigned)g0 - row0 * (unsigned)d.q.Wp, rr0 = row0 % (unsigned)d.q.RPU;      // uniform: scalar ALU
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int f = tid + 256 * u;
            if (f < nf4) { const int pos = f / C4, c4 = f - pos * C4; *reinterpret_cast<f32x4*>(lds + pos * PITCH + 4 * c4) = pre[u]; }
        }
        __syncthreads();
#ifndef RS_NO_LOAD
        if (tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x);
#endif
        f32x4 acc[NGRP][RBG];
#pragma unroll
        for (int gp = 0; gp < NGRP; ++gp) {
#pragma unroll
            for (int r = 0; r < RBG; ++r) acc[gp][r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float* tb = lds + ((rg * RPW + gp * RBG) * 16 + li + (d.shift[t] - d.smin)) * PITCH + 4 * g4;
#pragma unroll
                for (int jj = 0; jj < NJJ; ++jj) {
                    f32x4 b4[RBG];
#pragma unroll
                    for (int r = 0; r < RBG; ++r) b4[r] = *reinterpret_cast<const f32x4*>(tb + r * 16 * PITCH + 16 * jj);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < RBG; ++r)
#ifndef RS_NO_MFMA
                            acc[gp][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t][jj][j], b4[r][j], acc[gp][r], 0, 0, 0);
#else
                            acc[gp][r][j] += w[t][jj][j] * b4[r][j];
#endif
                }
            }
        }
        __syncthreads();      // every wave is done with the input tile: the same LDS now stages the outputs
#pragma unroll
        for (int gp = 0; gp < NGRP; ++gp)
#pragma unroll
            for (int r = 0; r < RBG; ++r)
                *reinterpret_cast<f32x4*>(lds + (((rg * RPW + gp * RBG + r) * 16) + li) * OPITCH + 16 * cb + 4 * g4) = acc[gp][r];
        __syncthreads();
        // epilogue operands (addend, a, y1) of EB passes are requested together: one memory latency per batch.  The 64-channel form owns
        // the CU's 512 registers and takes all of a tile's passes in one batch; the two-blocks-per-CU forms four at a time.
        constexpr int NPASS = RS_MT * N4 / 256, EB = CIN >= 64 ? (SM == 2 ? NPASS / 2 : NPASS) : (NPASS < 4 ? NPASS : 4);
#ifdef RS_NO_EPI
        if (d.q.G > 0) { if (tid == 0) d.out[(size_t)g0 * COUT] = lds[tid]; __syncthreads(); continue; }
#endif
        // statistics of the tile in fp32 (16 or fewer values per thread and channel), carried in fp64 across tiles
        f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = t0;
        for (int p0 = 0; p0 < NPASS; p0 += EB) {
            f32x4 ad[EB], aa[EB], yy[EB];
#pragma unroll
            for (int u = 0; u < EB; ++u) {
                const int pos = (tid + 256 * (p0 + u)) / N4;
                const unsigned g = (unsigned)(g0 + pos);
                const size_t off = (size_t)g * COUT + 4 * ch4;
                ad[u] = f32x4{0.f, 0.f, 0.f, 0.f}; aa[u] = ad[u]; yy[u] = ad[u];
                if (g < (unsigned)d.q.G) {
                    if (d.addend) ad[u] = *reinterpret_cast<const f32x4*>(d.addend + off);
                    if (SM == 2) { if (!d.act_a_none) aa[u] = *reinterpret_cast<const f32x4*>(d.act_a + off); yy[u] = *reinterpret_cast<const f32x4*>(d.y1 + off); }
                }
            }
#pragma unroll
            for (int u = 0; u < EB; ++u) {
                const int pos = (tid + 256 * (p0 + u)) / N4;
                const unsigned g = (unsigned)(g0 + pos);
                if (g >= (unsigned)d.q.G) continue;
                const bool ok = rs_valid_rel(d.q, rr0, col0, (unsigned)pos);
                f32x4 v = *reinterpret_cast<const f32x4*>(lds + pos * OPITCH + 4 * ch4);
                v += bias4;
                v += ad[u];
                if (SM == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = ok ? (d.act_a_none ? v[e] : v[e] * selu_grad_from_y(aa[u][e])) : 0.f;
                        t0[e] += v[e];
                        t1[e] += v[e] * ((yy[u][e] - mean4[e]) * rstd4[e]);      // v is 0 off the valid positions (y1 is finite everywhere)
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ok ? (d.epi_act ? selu_f(v[e]) : v[e]) : 0.f;
                    if (SM == 1) { t0 += v; t1 += v * v; }
                }
                *reinterpret_cast<f32x4*>(d.out + (size_t)g * COUT + 4 * ch4) = v;
            }
        }
        if (SM != 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { st[e] += (double)t0[e]; st[4 + e] += (double)t1[e]; }
        }
        __syncthreads();      // the staging area becomes the next input tile
    }
    if (SM == 0) return;
    // block totals: the 256 / N4 threads that share a channel quad, in a fixed order
    double* ld = reinterpret_cast<double*>(lds);
#pragma unroll
    for (int e = 0; e < 8; ++e) ld[tid * 8 + e] = st[e];
    __syncthreads();
    double* mine = ld + 256 * 8;
    if (tid < 2 * COUT) {
        const int stat = tid / COUT, ch = tid - stat * COUT;
        double s = 0.0;
        for (int k = ch >> 2; k < 256; k += N4) s += ld[k * 8 + stat * 4 + (ch & 3)];
        mine[tid] = s;
    }
    __syncthreads();
    rs_finish(d.acc, d.ticket, 2 * COUT, mine, mine + 2 * COUT, [&](const double* tot) {
        if (tid >= COUT) return;
        const int c = tid;
        if (SM == 1) {
            // BatchNorm forward statistics (nn.BatchNorm2d in training: biased variance to normalise, unbiased into the running buffer)
            const double m = tot[c] / d.nvalid;
            double var = tot[COUT + c] / d.nvalid - m * m;
            if (var < 0.0) var = 0.0;
            if (d.stats_out) {
                const float rstd = (float)(1.0 / sqrt(var + (double)d.eps));
                const float sc = (d.gamma ? d.gamma[c] : 1.f) * rstd;
                d.stats_out[c] = (float)m; d.stats_out[COUT + c] = rstd;
                d.stats_out[2 * COUT + c] = sc; d.stats_out[3 * COUT + c] = d.beta ? d.beta[c] : 0.f;
            }
            if (d.run_mean) {
                const double unb = d.nvalid > 1.0 ? var * d.nvalid / (d.nvalid - 1.0) : var;
                d.run_mean[c] = (float)((1.0 - d.momentum) * d.run_mean[c] + d.momentum * m);
                d.run_var[c] = (float)((1.0 - d.momentum) * d.run_var[c] + d.momentum * unb);
            }
            if (d.nbt && c == 0) *d.nbt += 1;
        } else {
            // BatchNorm backward: dbeta = sum dz, dgamma = sum dz * xhat; the two batch means the input gradient subtracts (none in eval mode)
            if (d.dbeta) d.dbeta[c] += (float)tot[c];
            if (d.dgamma) d.dgamma[c] += (float)tot[COUT + c];
            d.stats_out[c] = d.training ? (float)(tot[c] / d.nvalid) : 0.f;
            d.stats_out[COUT + c] = d.training ? (float)(tot[COUT + c] / d.nvalid) : 0.f;
        }
    });
}

// ---- element-wise passes ---------------------------------------------------------------------------------------------------------------
// a = selu((y - mean) * sc + beta) on the valid positions, 0 on the borders (they are conv2's zero padding)
__global__ __launch_bounds__(256) void rs_bn_act_kernel(const float* __restrict__ y, const float* __restrict__ stats, float* __restrict__ a, int C, int act, RsGeom q) {
    const int c4n = C >> 2;
    const long long n4 = (long long)q.G * c4n;
    for (long long f = (long long)blockIdx.x * 256 + threadIdx.x; f < n4; f += (long long)gridDim.x * 256) {
        const unsigned g = (unsigned)(f / c4n);
        const int c = (int)(f - (long long)g * c4n) * 4;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        if (rs_valid(q, g)) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(y + f * 4), mean = *reinterpret_cast<const f32x4*>(stats + c),
                        sc = *reinterpret_cast<const f32x4*>(stats + 2 * C + c), be = *reinterpret_cast<const f32x4*>(stats + 3 * C + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float z = (v[e] - mean[e]) * sc[e] + be[e]; o[e] = act ? selu_f(z) : z; }
        }
        *reinterpret_cast<f32x4*>(a + f * 4) = o;
    }
}
// dy = sc * (dz - m1 - xhat * m2) on the valid positions (m1 = m2 = 0 in eval mode), 0 elsewhere; in place on dz
__global__ __launch_bounds__(256) void rs_bn_bwd_apply_kernel(float* __restrict__ dz, const float* __restrict__ y, const float* __restrict__ stats,
                                                             const float* __restrict__ bstats, int C, int selu_in, RsGeom q) {
    const int c4n = C >> 2;
    const long long n4 = (long long)q.G * c4n;
    for (long long f = (long long)blockIdx.x * 256 + threadIdx.x; f < n4; f += (long long)gridDim.x * 256) {
        const unsigned g = (unsigned)(f / c4n);
        const int c = (int)(f - (long long)g * c4n) * 4;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        if (rs_valid(q, g)) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(dz + f * 4), yy = *reinterpret_cast<const f32x4*>(y + f * 4);
            const f32x4 mean = *reinterpret_cast<const f32x4*>(stats + c), rstd = *reinterpret_cast<const f32x4*>(stats + C + c),
                        sc = *reinterpret_cast<const f32x4*>(stats + 2 * C + c);
            const f32x4 m1 = *reinterpret_cast<const f32x4*>(bstats + c), m2 = *reinterpret_cast<const f32x4*>(bstats + C + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {      // selu_in: y is itself a SELU output (conv -> SELU -> BatchNorm): chain its derivative
                o[e] = sc[e] * (v[e] - m1[e] - (yy[e] - mean[e]) * rstd[e] * m2[e]);
                if (selu_in) o[e] *= selu_grad_from_y(yy[e]);
            }
        }
        *reinterpret_cast<f32x4*>(dz + f * 4) = o;
    }
}
// eval mode: mean / rstd / sc / sh from the running statistics
__global__ void rs_bn_eval_stats_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, int C, float* stats) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float rstd = (float)(1.0 / sqrt((double)rv[c] + (double)eps));
    const float sc = (gamma ? gamma[c] : 1.f) * rstd;
    stats[c] = rm[c]; stats[C + c] = rstd; stats[2 * C + c] = sc; stats[3 * C + c] = beta ? beta[c] : 0.f;
}

// ---- dense <-> bordered copies ------------------------------------------------------------------------------------------------------------
// mode 0: dense [B*H*W, Cs] -> bordered [G, Cd] (channels >= Cs and borders zero);  mode 1: bordered [G, Cs] -> dense [B*H*W, Cd] (first Cd channels)
__global__ __launch_bounds__(256) void rs_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cs, int Cd, int H, int mode, RsGeom q) {
    if (mode == 0) {
        const long long n = (long long)q.G * Cd;
        for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
            const unsigned g = (unsigned)(e / Cd);
            const int c = (int)(e - (long long)g * Cd);
            float v = 0.f;
            if (c < Cs && rs_valid(q, g)) {
                const unsigned row = g / (unsigned)q.Wp;
                const unsigned b = row / (unsigned)q.RPU;
                const int r = (int)(row - b * q.RPU), col = (int)(g - row * (unsigned)q.Wp);
                v = src[(((long long)b * H + (r - q.r_lo)) * q.W + (col - 1)) * Cs + c];
            }
            dst[e] = v;
        }
    } else {
        const long long n = (long long)(q.G / (q.RPU * q.Wp)) * H * q.W * Cd;
        for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
            const long long p = e / Cd;
            const int c = (int)(e - p * Cd);
            const long long bh = p / q.W;
            const int col = (int)(p - bh * q.W);
            const long long b = bh / H;
            const int r = (int)(bh - b * H);
            dst[e] = src[(((b * q.RPU) + r + q.r_lo) * q.Wp + col + 1) * Cs + c];
        }
    }
}

// ---- weights: torch [Co, Ci, KH, KW] -> the register image of rs_conv_kernel --------------------------------------------------------------
// wpk[cb][t][jj][lane][j] = Wt[t][c = 16 jj + 4 (lane >> 4) + j][n = 16 cb + (lane & 15)];  forward: Wt[t][c][n] = W[n][c][t];
// data gradient (transposed): Wt[t][c][n] = W[c][n][t] (c runs over the convolution's OUTPUT channels); zero beyond the real channel counts.
// One launch packs every image of the stack (blockIdx.y = job).
struct RsPackJobs { SclRsPackJob j[SCL_RS_MAX_PACK_JOBS]; };
__global__ __launch_bounds__(256) void rs_pack_kernel(const RsPackJobs jobs) {
    const SclRsPackJob& q = jobs.j[blockIdx.y];
    const int njj = q.CINp / 16;
    const int n_el = (q.COUTp / 16) * q.ntaps * njj * 256;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n_el; e += gridDim.x * 256) {
        const int j = e & 3, lane = (e >> 2) & 63;
        int rest = e >> 8;
        const int jj = rest % njj; rest /= njj;
        const int t = rest % q.ntaps; const int cb = rest / q.ntaps;
        const int c = 16 * jj + 4 * (lane >> 4) + j, n = 16 * cb + (lane & 15);
        float v = 0.f;
        const int ld = q.ld > 0 ? q.ld : q.Ci;      // row pitch of the torch tensor (a column block of a wider filter: Ci < ld)
        if (!q.transposed) { if (n < q.Co && c < q.Ci) v = q.w[((size_t)n * ld + c) * q.ntaps + t]; }
        else { if (c < q.Co && n < q.Ci) v = q.w[((size_t)c * ld + n) * q.ntaps + t]; }
        q.out[e] = v;
    }
}

// ---- weight gradient -------------------------------------------------------------------------------------------------------------------
struct RsWgradK {
    const float* in; const float* dout; float* part;       // part: [gridDim.x * NPART][NT * CIN * COUT] f32 slabs
    double* bacc; unsigned* ticket; float* dbias;          // bias gradient: fp64 accumulators [COUT] + arrival counter; dbias += total (may be NULL)
    int G, smin, span;
    int shift[6];
};

template <int CIN, int COUT, int NT>
__global__ __launch_bounds__(256, (CIN * COUT >= 4096 ? 1 : 2)) void rs_wgrad_kernel(const RsWgradK d) {
    constexpr int RS_PC = rs_pc(CIN, COUT);
    constexpr int NCB = COUT / 16, NPART = 4 / NCB, NCBK = CIN / 16;
    constexpr int PA = CIN == 16 ? 16 : CIN + 16;          // 4 position rows x 16 channels of a 4-byte read hit 64 distinct banks
    constexpr int PD = COUT == 16 ? 16 : COUT + 16;
    constexpr int C4 = CIN / 4, N4 = COUT / 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = wave % NCB, pp = wave / NCB, g4 = lane >> 4, li = lane & 15;
    const int nrows = RS_PC + d.span;
    float* la = lds;
    float* ldo = lds + nrows * PA;
    f32x4 acc[NT][NCBK];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int k = 0; k < NCBK; ++k) acc[t][k] = f32x4{0.f, 0.f, 0.f, 0.f};
    double bs[4] = {0, 0, 0, 0};
    const int nchunks = (d.G + RS_PC - 1) / RS_PC;
    const int nf4 = nrows * C4;
    constexpr int NLA = ((RS_PC + RS_MAX_SPAN) * C4 + 255) / 256, NLO = (RS_PC * N4 + 255) / 256;
    f32x4 pa[NLA], po[NLO];      // the next chunk of both maps, in flight while this one is multiplied
    auto issue = [&](int chunk) {
        const int p0 = chunk * RS_PC;
        const float* src = d.in + ((long long)p0 + d.smin) * CIN;
#pragma unroll
        for (int u = 0; u < NLA; ++u) {
            const int f = tid + 256 * u;
            if (f < nf4) pa[u] = *reinterpret_cast<const f32x4*>(src + (size_t)f * 4);
        }
        const float* sd = d.dout + (long long)p0 * COUT;
#pragma unroll
        for (int u = 0; u < NLO; ++u) {
            const int f = tid + 256 * u;
            po[u] = f32x4{0.f, 0.f, 0.f, 0.f};      // past the map: zeros (garbage x 0 could be NaN)
            if (f < RS_PC * N4 && p0 + f / N4 < d.G) po[u] = *reinterpret_cast<const f32x4*>(sd + (size_t)f * 4);
        }
    };
    int chunk = blockIdx.x;
    if (chunk < nchunks) issue(chunk);
    for (; chunk < nchunks; chunk += gridDim.x) {
#pragma unroll
        for (int u = 0; u < NLA; ++u) {
            const int f = tid + 256 * u;
            if (f < nf4) { const int pos = f / C4, c4 = f - pos * C4; *reinterpret_cast<f32x4*>(la + pos * PA + 4 * c4) = pa[u]; }
        }
#pragma unroll
        for (int u = 0; u < NLO; ++u) {      // f % N4 is the same in every pass (256 % N4 == 0): bs[] stays on one channel quad
            const int f = tid + 256 * u;
            if (f < RS_PC * N4) {
                const int pos = f / N4, c4 = f - pos * N4;
                *reinterpret_cast<f32x4*>(ldo + pos * PD + 4 * c4) = po[u];
                bs[0] += po[u][0]; bs[1] += po[u][1]; bs[2] += po[u][2]; bs[3] += po[u][3];
            }
        }
        __syncthreads();
        if (chunk + (int)gridDim.x < nchunks) issue(chunk + gridDim.x);
        // two k-steps in registers: the fragment reads of one step (NT * NCBK + 1 ds_read_b32) travel while the other step's MFMAs issue
        // (read -> wait -> two MFMAs, as the compiler schedules the plain loop, leaves the matrix pipe idle for an LDS round trip per pair)
        float a0[NT][NCBK], a1[NT][NCBK], b0, b1;
        auto frag = [&](int ks, float (&a)[NT][NCBK], float& b) {
            b = ldo[(4 * ks + g4) * PD + 16 * nb + li];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float* ta = la + (4 * ks + g4 + (d.shift[t] - d.smin)) * PA + li;
#pragma unroll
                for (int k = 0; k < NCBK; ++k) a[t][k] = ta[16 * k];
            }
        };
        auto mma = [&](const float (&a)[NT][NCBK], float b) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int k = 0; k < NCBK; ++k) acc[t][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][k], b, acc[t][k], 0, 0, 0);
        };
        static_assert((RS_PC / 4) % (2 * NPART) == 0, "an even number of k-steps per wave");
        frag(pp, a0, b0);
        for (int ks = pp; ks < RS_PC / 4; ks += 2 * NPART) {
            frag(ks + NPART, a1, b1);
            mma(a0, b0);
            if (ks + 2 * NPART < RS_PC / 4) frag(ks + 2 * NPART, a0, b0);
            mma(a1, b1);
        }
        __syncthreads();
    }
    // slab [t][c][n]: lane holds n = 16 nb + li, c = 16 k + 4 g4 + i
    float* slab = d.part + ((size_t)blockIdx.x * NPART + pp) * (NT * CIN * COUT);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int k = 0; k < NCBK; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) slab[((size_t)t * CIN + 16 * k + 4 * g4 + i) * COUT + 16 * nb + li] = acc[t][k][i];
    // (NPART > 1: the wave groups split the chunk's positions; group pp's waves together fill every column of slab pp)
    if (!d.dbias) return;
    double* ld = reinterpret_cast<double*>(lds);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) ld[tid * 4 + e] = bs[e];
    __syncthreads();
    double* mine = ld + 256 * 4;
    if (tid < COUT) {
        double s = 0.0;
        for (int k = tid >> 2; k < 256; k += N4) s += ld[k * 4 + (tid & 3)];
        mine[tid] = s;
    }
    __syncthreads();
    rs_finish(d.bacc, d.ticket, COUT, mine, mine + COUT, [&](const double* tot) {
        if (tid < COUT) d.dbias[tid] += (float)tot[tid];
    });
}

// dW[n][c][t] (torch layout [Co, Ci, NT]) += sum over slabs of part[slab][t][c][n], slabs in index order.  64 elements x 4 slab lanes per block.
__global__ __launch_bounds__(256) void rs_wgrad_reduce_kernel(const float* __restrict__ part, int nslab, int NT, int CINp, int COUTp, int Co, int Ci, int ld, float* __restrict__ dw) {
    __shared__ float red[4][64];
    const int el = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + el;
    const int n_el = NT * CINp * COUTp;
    float s = 0.f;
    if (e < n_el) {
        const size_t stride = (size_t)n_el;
        const float* p = part + e;
        int k = sl;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (; k + 12 < nslab; k += 16) {
            s0 += p[(size_t)k * stride]; s1 += p[(size_t)(k + 4) * stride]; s2 += p[(size_t)(k + 8) * stride]; s3 += p[(size_t)(k + 12) * stride];
        }
        for (; k < nslab; k += 4) s0 += p[(size_t)k * stride];
        s = (s0 + s1) + (s2 + s3);
    }
    red[sl][el] = s;
    __syncthreads();
    if (sl == 0 && e < n_el) {
        const float tot = (red[0][el] + red[1][el]) + (red[2][el] + red[3][el]);
        const int n = e % COUTp, c = (e / COUTp) % CINp, t = e / (COUTp * CINp);
        if (n < Co && c < Ci) dw[((size_t)n * ld + c) * NT + t] += tot;
    }
}

// ---- attention pooling over the [H, W] map (model/wav2vec2_aasist.py:527-541): e_S[h][c] = sum_w x[h][w][c] softmax_w(l[h][.][c]) + pos_S[h][c],
// e_T[w][c] = sum_h x[h][w][c] softmax_h(l[.][w][c]).  One block per (utterance, four channels): both maps' slices of that channel quad
// sit in LDS ([H][W] float4 each), a thread owns whole rows (softmax over W) and then whole columns (softmax over H).
__global__ __launch_bounds__(256) void rs_attn_pool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ l, const float* __restrict__ pos,
                                                               float* __restrict__ eS, float* __restrict__ eT, int C, int H, int W, RsGeom q) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f32x4* LX = reinterpret_cast<f32x4*>(lds);
    f32x4* LL = LX + H * W;
    const int b = blockIdx.x, c4 = blockIdx.y, tid = threadIdx.x;
    const size_t base = (size_t)b * q.RPU * q.Wp;
    for (int e = tid; e < H * W; e += 256) {
        const int h = e / W, w = e - h * W;
        const size_t g = base + (size_t)(h + 1) * q.Wp + (w + 1);
        LX[e] = *reinterpret_cast<const f32x4*>(x + g * C + 4 * c4);
        LL[e] = *reinterpret_cast<const f32x4*>(l + g * C + 4 * c4);
    }
    __syncthreads();
    for (int r = tid; r < H + W; r += 256) {
        const bool row = r < H;
        const int n = row ? W : H, i0 = row ? r * W : r - H, st = row ? 1 : W;
        f32x4 mx = LL[i0];
        for (int k = 1; k < n; ++k) { const f32x4 v = LL[i0 + k * st]; for (int e = 0; e < 4; ++e) mx[e] = fmaxf(mx[e], v[e]); }
        f32x4 den = {0.f, 0.f, 0.f, 0.f}, num = den;
        for (int k = 0; k < n; ++k) {
            const f32x4 v = LL[i0 + k * st], xv = LX[i0 + k * st];
            for (int e = 0; e < 4; ++e) { const float p = __expf(v[e] - mx[e]); den[e] += p; num[e] = fmaf(p, xv[e], num[e]); }
        }
        f32x4 o;
        for (int e = 0; e < 4; ++e) o[e] = num[e] / den[e];
        if (row) {
            if (pos) o += *reinterpret_cast<const f32x4*>(pos + (size_t)r * C + 4 * c4);
            *reinterpret_cast<f32x4*>(eS + ((size_t)b * H + r) * C + 4 * c4) = o;
        } else {
            *reinterpret_cast<f32x4*>(eT + ((size_t)b * W + (r - H)) * C + 4 * c4) = o;
        }
    }
}
// backward: dx = deS[h] P1 + deT[w] P2;  dl = P1 (deS[h] x - <deS[h] x, P1>_w) + P2 (deT[w] x - <deT[w] x, P2>_h)     (bordered outputs, zero borders kept)
// Four [H][W] slices per channel in LDS: a block takes TWO channels (42 x 66 x 2 x 4 arrays = 88 KiB).
typedef __attribute__((ext_vector_type(2))) float rs_f32x2;
__global__ __launch_bounds__(256) void rs_attn_pool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ l, const float* __restrict__ deS,
                                                               const float* __restrict__ deT, float* __restrict__ dx, float* __restrict__ dl, int C, int H, int W, RsGeom q) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    rs_f32x2* LX = reinterpret_cast<rs_f32x2*>(lds);
    rs_f32x2* LL = LX + H * W;
    rs_f32x2* LDX = LL + H * W;
    rs_f32x2* LDL = LDX + H * W;
    const int b = blockIdx.x, c2 = blockIdx.y, tid = threadIdx.x;
    const size_t base = (size_t)b * q.RPU * q.Wp;
    for (int e = tid; e < H * W; e += 256) {
        const int h = e / W, w = e - h * W;
        const size_t g = base + (size_t)(h + 1) * q.Wp + (w + 1);
        LX[e] = *reinterpret_cast<const rs_f32x2*>(x + g * C + 2 * c2);
        LL[e] = *reinterpret_cast<const rs_f32x2*>(l + g * C + 2 * c2);
    }
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {      // rows first (they initialise LDX / LDL), then columns (they accumulate)
        const int cnt = pass == 0 ? H : W;
        for (int r = tid; r < cnt; r += 256) {
            const bool row = pass == 0;
            const int n = row ? W : H, i0 = row ? r * W : r, st = row ? 1 : W;
            const rs_f32x2 de = row ? *reinterpret_cast<const rs_f32x2*>(deS + ((size_t)b * H + r) * C + 2 * c2)
                                    : *reinterpret_cast<const rs_f32x2*>(deT + ((size_t)b * W + r) * C + 2 * c2);
            rs_f32x2 mx = LL[i0];
            for (int k = 1; k < n; ++k) { const rs_f32x2 v = LL[i0 + k * st]; mx[0] = fmaxf(mx[0], v[0]); mx[1] = fmaxf(mx[1], v[1]); }
            rs_f32x2 den = {0.f, 0.f}, dot = den;
            for (int k = 0; k < n; ++k) {
                const rs_f32x2 v = LL[i0 + k * st], xv = LX[i0 + k * st];
                for (int e = 0; e < 2; ++e) { const float p = __expf(v[e] - mx[e]); den[e] += p; dot[e] = fmaf(p, xv[e], dot[e]); }
            }
            rs_f32x2 inv, ex;      // ex = <x, P> (the forward value without pos_S)
            for (int e = 0; e < 2; ++e) { inv[e] = 1.0f / den[e]; ex[e] = dot[e] * inv[e]; }
            for (int k = 0; k < n; ++k) {
                const int i = i0 + k * st;
                const rs_f32x2 v = LL[i], xv = LX[i];
                rs_f32x2 gx, gl;
                for (int e = 0; e < 2; ++e) {
                    const float p = __expf(v[e] - mx[e]) * inv[e];
                    gx[e] = de[e] * p;
                    gl[e] = p * de[e] * (xv[e] - ex[e]);
                }
                if (row) { LDX[i] = gx; LDL[i] = gl; } else { LDX[i] += gx; LDL[i] += gl; }
            }
        }
        __syncthreads();
    }
    for (int e = tid; e < H * W; e += 256) {
        const int h = e / W, w = e - h * W;
        const size_t g = base + (size_t)(h + 1) * q.Wp + (w + 1);
        *reinterpret_cast<rs_f32x2*>(dx + g * C + 2 * c2) = LDX[e];
        *reinterpret_cast<rs_f32x2*>(dl + g * C + 2 * c2) = LDL[e];
    }
}

template <int CIN, int COUT, int NT, int SM>
int rs_conv_launch_sm(const RsConvK& k, int grid, size_t lds, hipStream_t s) {
    if (lds > 65536) (void)hipFuncSetAttribute((const void*)rs_conv_kernel<CIN, COUT, NT, SM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    SCL_LAUNCH((rs_conv_kernel<CIN, COUT, NT, SM>), dim3(grid), dim3(256), lds, s, k);
    return scl_check_launch("rs_conv");
}
// which statistics modes an instantiation serves: 1 (BatchNorm forward) the forward shapes, 2 (BatchNorm backward) conv2's data gradient
template <int CIN, int COUT, int NT>
int rs_conv_launch_t(const RsConvK& k, int grid, hipStream_t s) {
    constexpr int RS_MT = rs_mt(CIN);
    const int ntiles = (k.q.G + RS_MT - 1) / RS_MT;
    const int cap = CIN >= 64 ? 256 : 512;          // resident blocks: one / two per CU
    grid = ntiles < cap ? ntiles : cap;
    const size_t lds_in = (size_t)(RS_MT + k.span) * (CIN + 4) * 4, lds_out = (size_t)RS_MT * (COUT + 4) * 4, lds_st = (size_t)(256 * 8 + 4 * COUT) * 8;
    size_t lds = lds_in > lds_out ? lds_in : lds_out;
    if (lds_st > lds) lds = lds_st;
    if (lds > 160 * 1024) { scl_set_error("rs_conv: tile + halo of %zu bytes exceeds the LDS", lds); return SCL_EINVAL; }
    constexpr bool FWD = (NT == 6 && COUT >= CIN) || NT == 1, C2T = (NT == 6 || NT == 1) && CIN == COUT;
    if (k.stat_mode == 0) return rs_conv_launch_sm<CIN, COUT, NT, 0>(k, grid, lds, s);
    if constexpr (FWD) { if (k.stat_mode == 1) return rs_conv_launch_sm<CIN, COUT, NT, 1>(k, grid, lds, s); }
    if constexpr (C2T) { if (k.stat_mode == 2) return rs_conv_launch_sm<CIN, COUT, NT, 2>(k, grid, lds, s); }
    scl_set_error("rs_conv: statistics mode %d is not instantiated for %d -> %d channels, %d taps", k.stat_mode, CIN, COUT, NT);
    return SCL_EINVAL;
}
template <int CIN, int COUT, int NT>
int rs_wgrad_launch_t(const RsWgradK& k, int grid, hipStream_t s) {
    constexpr int PA = CIN == 16 ? 16 : CIN + 16, PD = COUT == 16 ? 16 : COUT + 16, RS_PC = rs_pc(CIN, COUT);
    grid = rs_wgrad_blocks(CIN, COUT);
    size_t lds = ((size_t)(RS_PC + k.span) * PA + (size_t)RS_PC * PD) * 4;
    const size_t lds_st = (size_t)(256 * 4 + 2 * COUT) * 8;
    if (lds_st > lds) lds = lds_st;
    if (lds > 160 * 1024) { scl_set_error("rs_wgrad: chunk + halo of %zu bytes exceeds the LDS", lds); return SCL_EINVAL; }
    if (lds > 65536) (void)hipFuncSetAttribute((const void*)rs_wgrad_kernel<CIN, COUT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    SCL_LAUNCH((rs_wgrad_kernel<CIN, COUT, NT>), dim3(grid), dim3(256), lds, s, k);
    return scl_check_launch("rs_wgrad");
}

bool rs_fill_geom(const SclRsGeom& g, RsGeom* q) {
    if (g.B <= 0 || g.H <= 0 || g.W <= 0 || g.r_lo < 0 || g.r_hi > g.H + 1 || g.r_lo > g.r_hi) { scl_set_error("rs: bad geometry"); return false; }
    q->Wp = g.W + 2; q->RPU = g.H + 2; q->W = g.W; q->r_lo = g.r_lo; q->r_hi = g.r_hi;
    const long long G = (long long)g.B * q->RPU * q->Wp;
    if (G >= (1LL << 31) / 64) { scl_set_error("rs: map too large for 32-bit element offsets"); return false; }
    q->G = (int)G;
    q->m_wp = (65536u + q->Wp - 1) / q->Wp; q->m_rpu = (65536u + q->RPU - 1) / q->RPU;
    return true;
}
bool rs_shifts(const int* shift, int ntaps, int* dst, int* smin, int* span) {
    if (ntaps != 1 && ntaps != 3 && ntaps != 6) { scl_set_error("rs: ntaps must be 1, 3 or 6"); return false; }
    int lo = shift[0], hi = shift[0];
    for (int t = 0; t < 6; ++t) { dst[t] = t < ntaps ? shift[t] : 0; if (t < ntaps) { lo = shift[t] < lo ? shift[t] : lo; hi = shift[t] > hi ? shift[t] : hi; } }
    *smin = lo; *span = hi - lo;
    if (*span > RS_MAX_SPAN) { scl_set_error("rs: halo of %d positions exceeds %d (map wider than %d)", *span, RS_MAX_SPAN, RS_MAX_SPAN - 3); return false; }
    return true;
}

}  // namespace

extern "C" int scl_rs_pack_weights(const SclRsPackJob* jobs, int njobs, void* stream) {
    SCL_REQUIRE(jobs && njobs > 0 && njobs <= SCL_RS_MAX_PACK_JOBS, "rs_pack_weights: 1..%d jobs", SCL_RS_MAX_PACK_JOBS);
    RsPackJobs pj;
    int max_el = 0;
    for (int i = 0; i < njobs; ++i) {
        const SclRsPackJob& q = jobs[i];
        SCL_REQUIRE(q.w && q.out && q.Co > 0 && q.Ci > 0 && (q.ntaps == 1 || q.ntaps == 3 || q.ntaps == 6) && q.CINp % 16 == 0 && q.COUTp % 16 == 0 && (q.ld == 0 || q.ld >= q.Ci), "rs_pack_weights: bad job %d", i);
        SCL_REQUIRE(q.transposed ? (q.Co <= q.CINp && q.Ci <= q.COUTp) : (q.Co <= q.COUTp && q.Ci <= q.CINp), "rs_pack_weights: padded sizes below the real ones (job %d)", i);
        pj.j[i] = q;
        const int n_el = (q.COUTp / 16) * q.ntaps * (q.CINp / 16) * 256;
        max_el = n_el > max_el ? n_el : max_el;
    }
    hipLaunchKernelGGL(rs_pack_kernel, dim3((max_el + 1023) / 1024, njobs), dim3(256), 0, (hipStream_t)stream, pj);
    return scl_check_launch("rs_pack_weights");
}

extern "C" int scl_rs_conv(const SclRsConv* c, void* stream) {
    SCL_REQUIRE(c && c->in && c->wpk && c->out, "rs_conv: null pointers");
    RsConvK k;
    if (!rs_fill_geom(c->geom, &k.q)) return SCL_EINVAL;
    if (!rs_shifts(c->shift, c->ntaps, k.shift, &k.smin, &k.span)) return SCL_EINVAL;
    SCL_REQUIRE(c->stat_mode >= 0 && c->stat_mode <= 2, "rs_conv: stat_mode %d", c->stat_mode);
    SCL_REQUIRE(c->stat_mode == 0 || (c->acc && c->ticket && c->nvalid > 0), "rs_conv: statistics need accumulators, a ticket and the valid count");
    SCL_REQUIRE(c->stat_mode != 2 || (c->y1 && c->bnstats && c->stats_out), "rs_conv: stat_mode 2 needs y1, the forward statistics and stats_out");
    SCL_REQUIRE(!c->epi_act || c->stat_mode != 2, "rs_conv: the SELU epilogue belongs to the forward modes");
    k.in = c->in; k.wpk = c->wpk; k.bias = c->bias; k.addend = c->addend; k.out = c->out; k.act_a = c->act_a; k.y1 = c->y1; k.bnstats = c->bnstats;
    k.acc = c->acc; k.ticket = c->ticket; k.gamma = c->gamma; k.beta = c->beta; k.run_mean = c->run_mean; k.run_var = c->run_var;
    k.nbt = (long long*)c->nbt; k.stats_out = c->stats_out; k.dgamma = c->dgamma; k.dbeta = c->dbeta;
    k.stat_mode = c->stat_mode; k.training = c->training; k.eps = c->eps; k.momentum = c->momentum; k.nvalid = c->nvalid;
    k.epi_act = c->epi_act; k.act_a_none = (c->stat_mode == 2 && !c->act_a) ? 1 : 0;
    int grid = 0;          // chosen per instantiation (tile size, blocks per CU)
    hipStream_t s = (hipStream_t)stream;
    const int key = c->cin * 10000 + c->cout * 10 + c->ntaps;
    switch (key) {
        case 160326: return rs_conv_launch_t<16, 32, 6>(k, grid, s);
        case 320326: return rs_conv_launch_t<32, 32, 6>(k, grid, s);
        case 320646: return rs_conv_launch_t<32, 64, 6>(k, grid, s);
        case 640646: return rs_conv_launch_t<64, 64, 6>(k, grid, s);
        case 160323: return rs_conv_launch_t<16, 32, 3>(k, grid, s);
        case 320643: return rs_conv_launch_t<32, 64, 3>(k, grid, s);
        case 320166: return rs_conv_launch_t<32, 16, 6>(k, grid, s);
        case 640326: return rs_conv_launch_t<64, 32, 6>(k, grid, s);
        case 320163: return rs_conv_launch_t<32, 16, 3>(k, grid, s);
        case 640323: return rs_conv_launch_t<64, 32, 3>(k, grid, s);
        case 640641: return rs_conv_launch_t<64, 64, 1>(k, grid, s);
        default: break;
    }
    scl_set_error("rs_conv: no instantiation for %d -> %d channels, %d taps", c->cin, c->cout, c->ntaps);
    return SCL_EINVAL;
}

extern "C" int scl_rs_wgrad_nslabs(int cin, int cout) { return rs_wgrad_blocks(cin, cout) * (cout >= 64 ? 1 : (cout == 32 ? 2 : 4)); }

extern "C" int scl_rs_wgrad(const float* in, const float* dout, int cin, int cout, int ntaps, const int* shift, const SclRsGeom* geom, float* part,
                            double* bacc, unsigned* ticket, float* dbias, void* stream) {
    SCL_REQUIRE(in && dout && part && geom && shift, "rs_wgrad: null pointers");
    SCL_REQUIRE(!dbias || (bacc && ticket), "rs_wgrad: the bias gradient needs accumulators and a ticket");
    RsGeom q;
    if (!rs_fill_geom(*geom, &q)) return SCL_EINVAL;
    RsWgradK k;
    if (!rs_shifts(shift, ntaps, k.shift, &k.smin, &k.span)) return SCL_EINVAL;
    k.in = in; k.dout = dout; k.part = part; k.bacc = bacc; k.ticket = ticket; k.dbias = dbias; k.G = q.G;
    hipStream_t s = (hipStream_t)stream;
    int grid = 0;      // fixed per shape: the slab count the reduction walks (blocks past the last chunk write zero slabs)
    const int key = cin * 10000 + cout * 10 + ntaps;
    switch (key) {
        case 160326: return rs_wgrad_launch_t<16, 32, 6>(k, grid, s);
        case 320326: return rs_wgrad_launch_t<32, 32, 6>(k, grid, s);
        case 320646: return rs_wgrad_launch_t<32, 64, 6>(k, grid, s);
        case 640646: return rs_wgrad_launch_t<64, 64, 6>(k, grid, s);
        case 160323: return rs_wgrad_launch_t<16, 32, 3>(k, grid, s);
        case 320643: return rs_wgrad_launch_t<32, 64, 3>(k, grid, s);
        case 640641: return rs_wgrad_launch_t<64, 64, 1>(k, grid, s);
        default: break;
    }
    scl_set_error("rs_wgrad: no instantiation for %d -> %d channels, %d taps", cin, cout, ntaps);
    return SCL_EINVAL;
}

extern "C" int scl_rs_wgrad_reduce(const float* part, int nslab, int ntaps, int CINp, int COUTp, int Co, int Ci, int ld, float* dw, void* stream) {
    SCL_REQUIRE(part && dw && nslab > 0 && Co <= COUTp && Ci <= CINp && (ld == 0 || ld >= Ci), "rs_wgrad_reduce: bad arguments");
    if (ld == 0) ld = Ci;
    const int n_el = ntaps * CINp * COUTp;
    hipLaunchKernelGGL(rs_wgrad_reduce_kernel, dim3((n_el + 63) / 64), dim3(256), 0, (hipStream_t)stream, part, nslab, ntaps, CINp, COUTp, Co, Ci, ld, dw);
    return scl_check_launch("rs_wgrad_reduce");
}

extern "C" int scl_rs_bn_act(const float* y, const float* stats, float* a, int C, int act, const SclRsGeom* geom, void* stream) {
    SCL_REQUIRE(y && stats && a && geom && C % 4 == 0, "rs_bn_act: bad arguments");
    RsGeom q;
    if (!rs_fill_geom(*geom, &q)) return SCL_EINVAL;
    const long long n4 = (long long)q.G * (C / 4);
    hipLaunchKernelGGL(rs_bn_act_kernel, dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, y, stats, a, C, act, q);
    return scl_check_launch("rs_bn_act");
}

extern "C" int scl_rs_bn_bwd_apply(float* dz, const float* y, const float* stats, const float* bstats, int C, int selu_in, const SclRsGeom* geom, void* stream) {
    SCL_REQUIRE(dz && y && stats && bstats && geom && C % 4 == 0, "rs_bn_bwd_apply: bad arguments");
    RsGeom q;
    if (!rs_fill_geom(*geom, &q)) return SCL_EINVAL;
    const long long n4 = (long long)q.G * (C / 4);
    hipLaunchKernelGGL(rs_bn_bwd_apply_kernel, dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, dz, y, stats, bstats, C, selu_in, q);
    return scl_check_launch("rs_bn_bwd_apply");
}

extern "C" int scl_rs_bn_eval_stats(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps, int C,
                                    float* stats, void* stream) {
    SCL_REQUIRE(running_mean && running_var && stats && C > 0, "rs_bn_eval_stats: bad arguments");
    hipLaunchKernelGGL(rs_bn_eval_stats_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, gamma, beta, running_mean, running_var, eps, C, stats);
    return scl_check_launch("rs_bn_eval_stats");
}

extern "C" int scl_rs_copy(const float* src, float* dst, int Cs, int Cd, int to_dense, const SclRsGeom* geom, void* stream) {
    SCL_REQUIRE(src && dst && geom && Cs > 0 && Cd > 0, "rs_copy: bad arguments");
    SCL_REQUIRE(to_dense ? Cd <= Cs : Cs <= Cd, "rs_copy: channel counts");
    RsGeom q;
    if (!rs_fill_geom(*geom, &q)) return SCL_EINVAL;
    const int H = geom->r_hi - geom->r_lo + 1;
    const long long n = to_dense ? (long long)geom->B * H * q.W * Cd : (long long)q.G * Cd;
    hipLaunchKernelGGL(rs_copy_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, src, dst, Cs, Cd, H, to_dense ? 1 : 0, q);
    return scl_check_launch("rs_copy");
}

extern "C" int scl_rs_attn_pool_fwd(const float* x, const float* l, const float* pos, float* eS, float* eT, int C, const SclRsGeom* geom, void* stream) {
    SCL_REQUIRE(x && l && eS && eT && geom && C % 4 == 0, "rs_attn_pool_fwd: bad arguments");
    RsGeom q;
    if (!rs_fill_geom(*geom, &q)) return SCL_EINVAL;
    const size_t lds = (size_t)2 * geom->H * geom->W * 16;
    SCL_REQUIRE(lds <= 160 * 1024, "rs_attn_pool_fwd: map of %d x %d positions exceeds the LDS", geom->H, geom->W);
    if (lds > 65536) (void)hipFuncSetAttribute((const void*)rs_attn_pool_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(rs_attn_pool_fwd_kernel, dim3(geom->B, C / 4), dim3(256), lds, (hipStream_t)stream, x, l, pos, eS, eT, C, geom->H, geom->W, q);
    return scl_check_launch("rs_attn_pool_fwd");
}
extern "C" int scl_rs_attn_pool_bwd(const float* x, const float* l, const float* deS, const float* deT, float* dx, float* dl, int C, const SclRsGeom* geom, void* stream) {
    SCL_REQUIRE(x && l && deS && deT && dx && dl && geom && C % 4 == 0, "rs_attn_pool_bwd: bad arguments");
    RsGeom q;
    if (!rs_fill_geom(*geom, &q)) return SCL_EINVAL;
    const size_t lds = (size_t)4 * geom->H * geom->W * 8;
    SCL_REQUIRE(lds <= 160 * 1024, "rs_attn_pool_bwd: map of %d x %d positions exceeds the LDS", geom->H, geom->W);
    if (lds > 65536) (void)hipFuncSetAttribute((const void*)rs_attn_pool_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(rs_attn_pool_bwd_kernel, dim3(geom->B, C / 2), dim3(256), lds, (hipStream_t)stream, x, l, deS, deT, dx, dl, C, geom->H, geom->W, q);
    return scl_check_launch("rs_attn_pool_bwd");
}
