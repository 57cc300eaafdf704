// norm.hip — LayerNorm over the channel dimension, forward and backward, with the optional fused
// GELU of the wav2vec2 conv stack, plus the column reductions used for bias / affine gradients.
//
// Reference arithmetic: fairseq Fp32LayerNorm / nn.LayerNorm (eps 1e-5, biased variance, affine)
// reached from model/xlsr.py:41 — conv layers `TransposeLast -> Fp32LayerNorm -> TransposeLast ->
// GELU`, `layer_norm` before post_extract_proj, the two pre-LN norms of each encoder layer and
// `encoder.layer_norm`.  Statistics are always fp32, whatever the storage dtype.
//
// One 64-lane wave owns one row: C/64 (<= 32) values per lane live in registers, mean / variance
// are two wave-shuffle reductions, loads and stores are 16-byte (bf16x8) or 32-byte (f32x8) per lane.
#include "common.h"

namespace {

constexpr int MAXCH = 4;  // chunks of 8 elements per lane -> C <= 64*8*4 = 2048

template <bool F32>
__device__ __forceinline__ void load8(const void* base, int64_t off, float (&v)[8]) {
    if (F32) {
        const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off);
        const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
        const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(base) + off);
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
        }
    }
}
__device__ __forceinline__ void store8_bf16(bf16_t* base, int64_t off, const float (&v)[8]) {
    uint4 u;
    u.x = pack_bf2(v[0], v[1]); u.y = pack_bf2(v[2], v[3]); u.z = pack_bf2(v[4], v[5]); u.w = pack_bf2(v[6], v[7]);
    *reinterpret_cast<uint4*>(base + off) = u;
}
__device__ __forceinline__ void store8_f32(float* base, int64_t off, const float (&v)[8]) {
    *reinterpret_cast<float4*>(base + off) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(base + off + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

// One wave per NR rows (NR = 2: both rows' loads are requested before either reduction — a wave that handles one [1024] row is a
// single dependent chain load -> sum -> sum -> store, and 12736 such waves reached 3.2 TB/s; see tools/ln_probe.py).
template <bool XF32, int NR>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const void* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16_t* __restrict__ y_bf,
                                                     float* __restrict__ y_f32, float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out, int M, int C, int64_t ldx,
                                                     int64_t ldy, float eps, int act) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * NR;
    if (row0 >= M) return;
    constexpr int NCHR = NR == 1 ? MAXCH : 2;      // the multi-row form serves C <= 1024
    float v[NR][NCHR][8];
    float s[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        s[r] = 0.f;
#pragma unroll
        for (int ch = 0; ch < NCHR; ++ch) {
            const int c = ch * 512 + lane * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[r][ch][i] = 0.f;
            if (c < C && row0 + r < M) load8<XF32>(x, (int64_t)(row0 + r) * ldx + c, v[r][ch]);
        }
    }
    float mean[NR], rstd[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
#pragma unroll
        for (int ch = 0; ch < NCHR; ++ch) {
            const int c = ch * 512 + lane * 8;
            if (c < C) {
#pragma unroll
                for (int i = 0; i < 8; ++i) s[r] += v[r][ch][i];
            }
        }
        mean[r] = wave_sum(s[r]) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int ch = 0; ch < NCHR; ++ch) {
            const int c = ch * 512 + lane * 8;
            if (c < C) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { const float d = v[r][ch][i] - mean[r]; q += d * d; }
            }
        }
        rstd[r] = rsqrtf(wave_sum(q) / (float)C + eps);
    }
#pragma unroll
    for (int ch = 0; ch < NCHR; ++ch) {
        const int c = ch * 512 + lane * 8;
        if (c < C) {
            float g[8], b[8];
            load8<true>(gamma, c, g);
            load8<true>(beta, c, b);
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int row = row0 + r;
                if (row < M) {
                    float o[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) o[i] = (v[r][ch][i] - mean[r]) * rstd[r] * g[i] + b[i];
                    if ((act & 0xFF) == 1) {
#pragma unroll
                        for (int i = 0; i < 8; i += 2) gelu2(o[i], o[i + 1]);      // packed-f32 form, same bits as gelu_f
                    }
                    if (y_bf && (act & 0x100)) {
                        // triple-plane output (round 6, scoring path): row = [hi | hi | lo], hi = bf16(o), lo = bf16(o - hi), pitch 3 C —
                        // the left operand of the bf16 GEMM over 3 K that stands in for the f32 product (elementwise.hip split3_kernel)
                        float lo[8];
                        uint4 h;
                        h.x = pack_bf2(o[0], o[1]); h.y = pack_bf2(o[2], o[3]); h.z = pack_bf2(o[4], o[5]); h.w = pack_bf2(o[6], o[7]);
                        const unsigned hw[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            lo[2 * i] = o[2 * i] - __uint_as_float(hw[i] << 16);
                            lo[2 * i + 1] = o[2 * i + 1] - __uint_as_float(hw[i] & 0xFFFF0000u);
                        }
                        bf16_t* dst = y_bf + (int64_t)row * 3 * C + c;
                        *reinterpret_cast<uint4*>(dst) = h;
                        *reinterpret_cast<uint4*>(dst + C) = h;
                        store8_bf16(dst, 2 * (int64_t)C, lo);
                    } else if (y_bf) store8_bf16(y_bf, (int64_t)row * ldy + c, o);
                    if (y_f32) store8_f32(y_f32, (int64_t)row * ldy + c, o);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        if (lane == 0 && row0 + r < M) {
            if (mean_out) mean_out[row0 + r] = mean[r];
            if (rstd_out) rstd_out[row0 + r] = rstd[r];
        }
    }
}

// Backward.  dy -> dx (optionally + residual gradient), per-block partial sums of dgamma / dbeta.
// Each wave walks rows row0 + w, row0 + w + 4, ... of its block's slab and keeps the per-column
// sums in registers; the four waves are combined through LDS.
// NCH = chunks of 512 channels a wave covers (1: C <= 512, 2: C <= 1024, 4: C <= 2048): sized exactly, the accumulators take
// 32*NCH + ... registers instead of 227 for every C, i.e. 4-7 waves per SIMD instead of 2 on this latency-bound kernel.
template <bool XF32, bool DYF32, int NCH>
__global__ __launch_bounds__(256, (NCH <= 2 ? 3 : 2)) void ln_bwd_kernel(const void* __restrict__ dy, const void* __restrict__ x,
                                                     const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ dres, float* __restrict__ dx_f32,
                                                     bf16_t* __restrict__ dx_bf, float* __restrict__ part, int M, int C, int64_t ldx,
                                                     int64_t lddy, int64_t lddx, int rows_per_block, int act, int sum_dres,
                                                     int out_rpb, int64_t out_rbstride, int64_t out_off,
                                                     uint32_t din_seed, float din_p, uint32_t dout_seed, float dout_p) {
    // Element dropout of the encoder (fairseq dropout1 / dropout3, p = cfg.dropout, on the output of out_proj / fc2 BEFORE the residual
    // add): the residual gradient `dres` passes this LayerNorm unmasked, but (a) its column sum = the bias gradient of the linear that
    // fed the residual must be taken of dres x keep-mask of THAT linear's dropout (din), and (b) the bf16 copy of the output, which
    // is only ever the dY operand of the next linear's weight / data gradient GEMMs, carries that linear's mask (dout).  Masks are
    // recomputed from (seed, element index = row * C + column), exactly as the forward GEMM epilogue drew them.
    // Tried in round 5 (tools/ln_probe.py, 12736 x 1024, x f32 / dy bf16 / dres f32 -> dx f32 + bf16: 48 - 51 us = 4.1 - 4.3 TB/s): requesting
    // `dres` together with x and dy — one memory round trip per row instead of two — costs 16 registers, i.e. the third resident block per
    // CU: 61 us; rows per block 8 ... 40 instead of 20: 48 - 59 us, the present choice is at the optimum.
    // sum_dres: also emit the column sums of `dres` (third partial row).  In a pre-LN transformer block the residual gradient
    // that enters this LayerNorm's backward IS the gradient of the preceding linear's output (fc2 / out_proj), so its column
    // sum is that linear's bias gradient — read here anyway, summed for free instead of by a separate pass over [M, C].
    extern __shared__ float red[];  // [4][2 + sum_dres][C]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    float ag[NCH][8], ab[NCH][8], ar[NCH][8];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int i = 0; i < 8; ++i) { ag[ch][i] = 0.f; ab[ch][i] = 0.f; ar[ch][i] = 0.f; }
    const int np = 2 + (sum_dres ? 1 : 0);

    for (int row = r0 + w; row < r1; row += 4) {
        const float mean = mean_in[row], rstd = rstd_in[row];
        float xh[NCH][8], dyv[NCH][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int c = ch * 512 + lane * 8;
            if (c < C) {
                float xv[8], g[8];
                load8<XF32>(x, (int64_t)row * ldx + c, xv);
                load8<DYF32>(dy, (int64_t)row * lddy + c, dyv[ch]);
                load8<true>(gamma, c, g);
                if (act == 1) {
                    float b[8];
                    load8<true>(beta, c, b);
                    float gg[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) gg[i] = ((xv[i] - mean) * rstd) * g[i] + b[i];
#pragma unroll
                    for (int i = 0; i < 8; i += 2) gelu_grad2(gg[i], gg[i + 1]);      // packed-f32 form, same bits as gelu_grad_f
#pragma unroll
                    for (int i = 0; i < 8; ++i) dyv[ch][i] *= gg[i];
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float h = (xv[i] - mean) * rstd;
                    xh[ch][i] = h;
                    ag[ch][i] += dyv[ch][i] * h;
                    ab[ch][i] += dyv[ch][i];
                    const float dh = dyv[ch][i] * g[i];
                    dyv[ch][i] = dh;  // now dxhat
                    s1 += dh;
                    s2 += dh * h;
                }
            }
        }
        s1 = wave_sum(s1) / (float)C;
        s2 = wave_sum(s2) / (float)C;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int c = ch * 512 + lane * 8;
            if (c < C) {
                float o[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = rstd * (dyv[ch][i] - s1 - xh[ch][i] * s2);
                if (dres) {
                    float r[8];
                    load8<true>(dres, (int64_t)row * lddx + c, r);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        o[i] += r[i];
                        ar[ch][i] += din_p > 0.f ? r[i] * dropout_scale(din_seed, (uint64_t)((int64_t)row * C + c + i), din_p) : r[i];
                    }
                }
                if (sum_dres == 2) {      // column sums of the OUTPUT: dx of a conv layer's LayerNorm is d(conv output) => conv bias gradient
#pragma unroll
                    for (int i = 0; i < 8; ++i) ar[ch][i] += o[i];
                }
                if (dx_f32) store8_f32(dx_f32, (int64_t)row * lddx + c, o);
                if (dx_bf) {
                    // optional per-utterance padding of the bf16 output (zero rows around each utterance's frames, pre-zeroed
                    // by the caller): the phase-split conv dgrad reads [dy[u-1], dy[u]] as ONE overlapping GEMM row
                    const int64_t orow = out_rpb > 0 ? (int64_t)(row / out_rpb) * out_rbstride + (int64_t)(row % out_rpb) * lddx + out_off
                                                     : (int64_t)row * lddx;
                    if (dout_p > 0.f) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) o[i] *= dropout_scale(dout_seed, (uint64_t)((int64_t)row * C + c + i), dout_p);
                    }
                    store8_bf16(dx_bf, orow + c, o);
                }
            }
        }
    }
    // combine the 4 waves
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int c = ch * 512 + lane * 8;
        if (c < C) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                red[(w * np + 0) * C + c + i] = ag[ch][i];
                red[(w * np + 1) * C + c + i] = ab[ch][i];
                if (sum_dres) red[(w * np + 2) * C + c + i] = ar[ch][i];
            }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float g = 0.f, b = 0.f, r = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) {
            g += red[(ww * np + 0) * C + c]; b += red[(ww * np + 1) * C + c];
            if (sum_dres) r += red[(ww * np + 2) * C + c];
        }
        part[(int64_t)blockIdx.x * np * C + c] = g;          // [block][dgamma (C) | dbeta (C) | colsum(dres) (C, optional)]
        part[(int64_t)blockIdx.x * np * C + C + c] = b;
        if (sum_dres) part[(int64_t)blockIdx.x * np * C + 2 * C + c] = r;
    }
}

// out[c] (+)= sum_p part[p][c]; 32 columns x 8 part-lanes per block (fixed summation order: deterministic)
__global__ __launch_bounds__(256) void colreduce_kernel(const float* __restrict__ part, float* __restrict__ out, int nparts, int C,
                                                        int64_t pstride, int accumulate) {
    __shared__ float red[8][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + tx;
    float s = 0.f;
    if (c < C)
        for (int p = ty; p < nparts; p += 8) s += part[(int64_t)p * pstride + c];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][tx];
        out[c] = accumulate ? out[c] + t : t;
    }
}

// The same sum with the partial rows split over gridDim.y segments (8x the blocks of the kernel above, which is latency-bound
// on 64 blocks): each block reduces its segment into scratch[seg][c]; the last block of a column group (ticket, agent-scope
// sc1 write-through hand-off) adds the segments in order.  Deterministic; the counters return to zero.
__global__ __launch_bounds__(256) void colreduce_seg_kernel(const float* __restrict__ part, float* __restrict__ out, int nparts, int C,
                                                            int64_t pstride, int accumulate, float* __restrict__ scratch,
                                                            int* __restrict__ counters, float* __restrict__ out2, int split) {
    __shared__ float red[8][33];
    __shared__ int last_flag;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + tx;
    const int nseg = gridDim.y, seg = blockIdx.y;
    const int per = (nparts + nseg - 1) / nseg;
    const int p0 = seg * per, p1 = min(nparts, p0 + per);
    float s = 0.f;
    if (c < C)
        for (int p = p0 + ty; p < p1; p += 8) s += part[(int64_t)p * pstride + c];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][tx];
        __hip_atomic_store(&scratch[(int64_t)seg * C + c], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through (sc1)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = __hip_atomic_fetch_add(&counters[blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int is_last = t == nseg - 1;
        if (is_last) __hip_atomic_store(&counters[blockIdx.x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = is_last;
    }
    __syncthreads();
    if (last_flag && ty == 0 && c < C) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float t = 0.f;
        for (int k = 0; k < nseg; ++k) t += __hip_atomic_load(&scratch[(int64_t)k * C + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float* o = (out2 && c >= split) ? out2 + (c - split) : out + c;     // columns >= split go to the second destination
        *o = accumulate ? *o + t : t;
    }
}

// Several column reductions in ONE launch: the four small reductions that close an encoder layer's backward (LayerNorm parameter
// gradients + the residual linear's bias, twice; fc1 bias; q/k/v bias) were four ~10 us launches of 64-800 blocks each, latency-bound.
// Block = 32 columns x 32 row lanes of one job; fixed summation order (row lane r adds rows r, r + 32, ...; the lanes are combined
// 0..31) => deterministic.
// Several split-K slab combines in one launch (scl_reduce_slabs_multi): every job keeps the grid, the element-to-thread map and the
// summation order (slab 0, 1, 2, ...) it has alone in scl_reduce_slabs_f32 => the same bits; what goes away is three kernel boundaries
// per encoder layer (~5 us each between dependent kernels on this GPU, tools/graph_gap_probe.py).
struct SlabJobs { SclSlabJob job[SCL_SLAB_MAX_JOBS]; int first_block[SCL_SLAB_MAX_JOBS + 1]; int njobs; };
__global__ __launch_bounds__(256) void reduce_slabs_multi_kernel(const SlabJobs J) {
    int j = 0;
    while (j + 1 < J.njobs && (int)blockIdx.x >= J.first_block[j + 1]) ++j;
    const SclSlabJob jb = J.job[j];
    const int nb = J.first_block[j + 1] - J.first_block[j];
    const int64_t i0 = ((int64_t)((int)blockIdx.x - J.first_block[j]) * blockDim.x + threadIdx.x) * 4;
    const int64_t step = (int64_t)nb * blockDim.x * 4;
    for (int64_t i = i0; i < jb.n; i += step) {
        if (i + 4 <= jb.n) {
            float4 s = *reinterpret_cast<const float4*>(jb.slabs + i);
            for (int k = 1; k < jb.nslabs; ++k) {
                const float4 t = *reinterpret_cast<const float4*>(jb.slabs + k * jb.stride + i);
                s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
            }
            *reinterpret_cast<float4*>(jb.out + i) = s;
        } else {
            for (int64_t q = i; q < jb.n; ++q) {
                float s = jb.slabs[q];
                for (int k = 1; k < jb.nslabs; ++k) s += jb.slabs[k * jb.stride + q];
                jb.out[q] = s;
            }
        }
    }
}

struct ReduceJobs { SclReduceJob job[SCL_REDUCE_MAX_JOBS]; int first_block[SCL_REDUCE_MAX_JOBS + 1]; int njobs; };
__global__ __launch_bounds__(1024) void colreduce_multi_kernel(const ReduceJobs J) {
    __shared__ float red[32][33];
    int j = 0;
    while (j + 1 < J.njobs && (int)blockIdx.x >= J.first_block[j + 1]) ++j;
    const SclReduceJob jb = J.job[j];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = ((int)blockIdx.x - J.first_block[j]) * 32 + tx;
    float s = 0.f;
    if (c < jb.C) {
        int p = ty;
        for (; p + 96 < jb.nparts; p += 128) {      // four independent loads in flight
            const float a0 = jb.part[(int64_t)p * jb.pstride + c], a1 = jb.part[(int64_t)(p + 32) * jb.pstride + c];
            const float a2 = jb.part[(int64_t)(p + 64) * jb.pstride + c], a3 = jb.part[(int64_t)(p + 96) * jb.pstride + c];
            s += a0; s += a1; s += a2; s += a3;
        }
        for (; p < jb.nparts; p += 32) s += jb.part[(int64_t)p * jb.pstride + c];
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < jb.C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += red[k][tx];
        float* o = (jb.out2 && c >= jb.split) ? jb.out2 + (c - jb.split) : jb.out + c;
        *o = t;
    }
}

// column sums of a [M, N] matrix (bias gradients): part[rowchunk][n].  With `out` != null the launch also finishes the sum:
// every block stores its partial row write-through (sc1), drains, and draws a ticket of its column group; the block that
// draws the last one adds the group's partials (sc1 loads) in row-chunk order (fixed order => deterministic) and re-arms the
// counter.  That replaces a separate 10 us, 64-block reduce launch per bias gradient (108 per XLS-R train step).
template <bool F32>
__global__ __launch_bounds__(256) void colsum_kernel(const void* __restrict__ x, float* __restrict__ part, int M, int N,
                                                     int64_t ld, int rows_per_block, int* __restrict__ counters, float* __restrict__ out) {
    __shared__ float red[16][128 + 1];
    __shared__ int last_flag;
    const int cg = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c0 = blockIdx.x * 128 + cg * 8;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c0 < N) {
        for (int r = r0 + ty; r < r1; r += 16) {
            float v[8];
            load8<F32>(x, (int64_t)r * ld + c0, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += v[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[ty][cg * 8 + i] = acc[i];
    __syncthreads();
    const int c = blockIdx.x * 128 + threadIdx.x;
    if (threadIdx.x < 128 && c < N) {
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) s += red[t][threadIdx.x];
        // self-finishing launches hand the partials over write-through (sc1): no agent-scope release fence, i.e. no L2
        // write-back per block (with the fence every block paid ~10 us under load: measured, rocprofv3)
        if (out) __hip_atomic_store(&part[(int64_t)blockIdx.y * N + c], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else part[(int64_t)blockIdx.y * N + c] = s;
    }
    if (out == nullptr) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = __hip_atomic_fetch_add(&counters[blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int is_last = t == (int)gridDim.y - 1;
        if (is_last) __hip_atomic_store(&counters[blockIdx.x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for the next launch
        last_flag = is_last;
    }
    __syncthreads();
    if (last_flag) {    // block-uniform
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    // compiler ordering only; every load below is sc1 (bypasses this CU's L1)
        // even / odd row chunks on the two thread halves, 8 loads in flight per thread, fixed association => deterministic
        const int cc = threadIdx.x & 127, half = threadIdx.x >> 7, ny = gridDim.y;
        const int col = blockIdx.x * 128 + cc;
        float s = 0.f;
        if (col < N) {
            for (int p0 = half; p0 < ny; p0 += 16) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int p = p0 + 2 * j;
                    v[j] = p < ny ? __hip_atomic_load(&part[(int64_t)p * N + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) s += v[j];
            }
        }
        float* flat = &red[0][0];
        flat[threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x < 128 && col < N) out[col] = flat[threadIdx.x] + flat[threadIdx.x + 128];
    }
}

}  // namespace

extern "C" int scl_layernorm_fwd(const void* x, int x_f32, const float* gamma, const float* beta, void* y_bf16,
                                 float* y_f32, float* mean, float* rstd, int M, int C, int64_t ldx, int64_t ldy,
                                 float eps, int act, void* stream) {
    SCL_REQUIRE(x && gamma && beta && (y_bf16 || y_f32), "layernorm_fwd: null pointer");
    SCL_REQUIRE(M > 0 && C >= 8 && C <= 2048 && (C & 7) == 0 && (ldx & 7) == 0 && (ldy & 7) == 0,
                "layernorm_fwd: need 8 <= C <= 2048, C, ldx, ldy multiples of 8 (C=%d)", C);
    dim3 block(256);
    hipStream_t s = (hipStream_t)stream;
    const int nr_env = 2;
    const int nr = (nr_env == 2 && C <= 1024 && M >= 4096) ? 2 : 1;
    dim3 grid((M + 4 * nr - 1) / (4 * nr));
#define LN_FWD(XF, NR) hipLaunchKernelGGL((ln_fwd_kernel<XF, NR>), grid, block, 0, s, x, gamma, beta, (bf16_t*)y_bf16, y_f32, mean, rstd, M, C, ldx, ldy, eps, act)
    if (x_f32) { if (nr == 2) LN_FWD(true, 2); else LN_FWD(true, 1); }
    else { if (nr == 2) LN_FWD(false, 2); else LN_FWD(false, 1); }
#undef LN_FWD
    return scl_check_launch("scl_layernorm_fwd");
}

static int ln_bwd_rows_per_block(int M) {
    // ~768 blocks keep all 256 CUs busy (3 blocks each); at least one row per wave, at most 1024 partial rows to reduce
    int rpb = (M + 767) / 768;
    rpb = (rpb + 3) / 4 * 4;
    if (rpb < 4) rpb = 4;
    while ((M + rpb - 1) / rpb > 1024) rpb *= 2;
    return rpb;
}
extern "C" int scl_layernorm_bwd_nparts(int M) {
    const int rpb = ln_bwd_rows_per_block(M);
    return (M + rpb - 1) / rpb;
}

extern "C" int scl_layernorm_bwd(const void* dy, int dy_f32, const void* x, int x_f32, const float* mean,
                                 const float* rstd, const float* gamma, const float* beta, const float* dres,
                                 float* dx_f32, void* dx_bf16, float* part, int M, int C,
                                 int64_t ldx, int64_t lddy, int64_t lddx, int act, int sum_dres, int out_rpb, int64_t out_rbstride,
                                 int64_t out_off, uint32_t din_seed, float din_p, uint32_t dout_seed, float dout_p, void* stream) {
    SCL_REQUIRE(din_p >= 0.f && din_p < 1.f && dout_p >= 0.f && dout_p < 1.f, "layernorm_bwd: dropout probabilities in [0, 1)");
    SCL_REQUIRE(din_p == 0.f || sum_dres == 1, "layernorm_bwd: din mask applies to the column sums of dres (sum_dres = 1)");
    SCL_REQUIRE(dout_p == 0.f || (dx_bf16 && out_rpb == 0 && lddx == C), "layernorm_bwd: dout mask needs a plain [M, C] bf16 output");
    SCL_REQUIRE(out_rpb == 0 || (dx_bf16 && out_rpb > 0 && (out_rbstride & 7) == 0 && (out_off & 7) == 0), "layernorm_bwd: padded output needs dx_bf16");
    SCL_REQUIRE(sum_dres == 0 || (sum_dres == 1 && dres) || (sum_dres == 2 && !dres), "layernorm_bwd: sum_dres 1 needs dres, 2 excludes it");
    SCL_REQUIRE(dy && x && mean && rstd && gamma && part && (dx_f32 || dx_bf16), "layernorm_bwd: null pointer");
    SCL_REQUIRE(act == 0 || beta, "layernorm_bwd: gelu variant needs beta");
    SCL_REQUIRE(M > 0 && C >= 8 && C <= 2048 && (C & 7) == 0 && (ldx & 7) == 0 && (lddy & 7) == 0 && (lddx & 7) == 0,
                "layernorm_bwd: need 8 <= C <= 2048 and multiples of 8");
    const int rows_per_block = ln_bwd_rows_per_block(M);
    const int nblk = (M + rows_per_block - 1) / rows_per_block;
    const size_t lds = (size_t)4 * (2 + (sum_dres ? 1 : 0)) * C * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(nblk), block(256);
#define LN_BWD(XF, DF, NC) hipLaunchKernelGGL((ln_bwd_kernel<XF, DF, NC>), grid, block, lds, s, dy, x, mean, rstd, gamma, beta, dres, \
                                              dx_f32, (bf16_t*)dx_bf16, part, M, C, ldx, lddy, lddx, rows_per_block, act, sum_dres, \
                                              out_rpb, out_rbstride, out_off, din_seed, din_p, dout_seed, dout_p)
#define LN_BWD_C(XF, DF) do { if (C <= 512) LN_BWD(XF, DF, 1); else if (C <= 1024) LN_BWD(XF, DF, 2); else LN_BWD(XF, DF, 4); } while (0)
    if (x_f32 && dy_f32) LN_BWD_C(true, true);
    else if (x_f32) LN_BWD_C(true, false);
    else if (dy_f32) LN_BWD_C(false, true);
    else LN_BWD_C(false, false);
#undef LN_BWD_C
#undef LN_BWD
    return scl_check_launch("scl_layernorm_bwd");
}

extern "C" int scl_colreduce_f32(const float* part, float* out, int nparts, int C, int64_t pstride, int accumulate, void* stream) {
    SCL_REQUIRE(part && out && nparts >= 1 && C >= 1, "colreduce: bad args");
    hipLaunchKernelGGL(colreduce_kernel, dim3((C + 31) / 32), dim3(256), 0, (hipStream_t)stream, part, out, nparts, C, pstride, accumulate);
    return scl_check_launch("scl_colreduce_f32");
}

extern "C" int scl_colreduce_seg_f32(const float* part, float* out, int nparts, int C, int64_t pstride, int accumulate, float* scratch,
                                     int* counters, float* out2, int split, void* stream) {
    SCL_REQUIRE(part && out && scratch && counters && nparts >= 1 && C >= 1, "colreduce_seg: bad args");
    SCL_REQUIRE(!out2 || (split > 0 && split < C), "colreduce_seg: split must lie inside (0, C) when out2 is given");
    SCL_REQUIRE((C + 31) / 32 <= SCL_COLSUM_MAX_GROUPS, "colreduce_seg: C too large for the counter array (%d)", C);
    const int nseg = nparts >= 64 ? SCL_COLREDUCE_SEGMENTS : 1;
    hipLaunchKernelGGL(colreduce_seg_kernel, dim3((C + 31) / 32, nseg), dim3(256), 0, (hipStream_t)stream, part, out, nparts, C, pstride,
                       accumulate, scratch, counters, out2, split);
    return scl_check_launch("scl_colreduce_seg_f32");
}

extern "C" int scl_reduce_slabs_multi(const SclSlabJob* jobs, int njobs, void* stream) {
    SCL_REQUIRE(jobs && njobs >= 1 && njobs <= SCL_SLAB_MAX_JOBS, "reduce_slabs_multi: 1 .. %d jobs", SCL_SLAB_MAX_JOBS);
    SlabJobs J;
    int blocks = 0;
    for (int i = 0; i < njobs; ++i) {
        const SclSlabJob& b = jobs[i];
        SCL_REQUIRE(b.slabs && b.out && b.n > 0 && b.nslabs >= 1, "reduce_slabs_multi: job %d: bad arguments", i);
        SCL_REQUIRE(((uintptr_t)b.slabs & 15) == 0 && ((uintptr_t)b.out & 15) == 0 && (b.stride & 3) == 0, "reduce_slabs_multi: job %d: alignment", i);
        J.job[i] = b;
        J.first_block[i] = blocks;
        int nb = (int)((b.n / 4 + 255) / 256); if (nb > 2048) nb = 2048; if (nb < 1) nb = 1;      // the grid scl_reduce_slabs_f32 gives the job
        blocks += nb;
    }
    J.first_block[njobs] = blocks;
    J.njobs = njobs;
    hipLaunchKernelGGL(reduce_slabs_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, J);
    return scl_check_launch("scl_reduce_slabs_multi");
}

extern "C" int scl_colreduce_multi(const SclReduceJob* jobs, int njobs, void* stream) {
    SCL_REQUIRE(jobs && njobs >= 1 && njobs <= SCL_REDUCE_MAX_JOBS, "colreduce_multi: 1 .. %d jobs", SCL_REDUCE_MAX_JOBS);
    ReduceJobs J;
    int blocks = 0;
    for (int i = 0; i < njobs; ++i) {
        const SclReduceJob& b = jobs[i];
        SCL_REQUIRE(b.part && b.out && b.nparts >= 1 && b.C >= 1 && b.pstride >= b.C, "colreduce_multi: job %d: bad arguments", i);
        SCL_REQUIRE(!b.out2 || (b.split > 0 && b.split < b.C), "colreduce_multi: job %d: split must lie inside (0, C) when out2 is given", i);
        J.job[i] = b;
        J.first_block[i] = blocks;
        blocks += (b.C + 31) / 32;
    }
    J.first_block[njobs] = blocks;
    J.njobs = njobs;
    hipLaunchKernelGGL(colreduce_multi_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, J);
    return scl_check_launch("scl_colreduce_multi");
}

// row slab per block: 256 rows, doubled until at most 64 slabs remain (finer slabs were measured slower: the finishing block
// reads one partial row per slab)
static int colsum_rows(int M) {
    int rows_per_block = 256;
    while ((M + rows_per_block - 1) / rows_per_block > 64) rows_per_block *= 2;
    return rows_per_block;
}

// The self-finishing form over a NARROW matrix (N <= 128: one column group, so the grid is its row slabs only — 44 blocks for the
// [177408, 64] convolution outputs of the AASIST back-end, 1.6 TB/s): up to 512 slabs there; the finishing block reads 512 x N floats.
static int colsum_reduce_rows(int M, int N) {
    if (N > 128) return colsum_rows(M);
    int rows_per_block = 256;
    while ((M + rows_per_block - 1) / rows_per_block > 512) rows_per_block *= 2;
    return rows_per_block;
}

extern "C" int scl_colsum_reduce_nparts(int M, int N) {
    const int rows_per_block = colsum_reduce_rows(M, N);
    return (M + rows_per_block - 1) / rows_per_block;
}

extern "C" int scl_colsum_nparts(int M) {
    const int rows_per_block = colsum_rows(M);
    return (M + rows_per_block - 1) / rows_per_block;
}

extern "C" int scl_colsum(const void* x, int x_f32, float* part, int M, int N, int64_t ld, void* stream) {
    SCL_REQUIRE(x && part && M > 0 && N > 0 && (N & 7) == 0 && (ld & 7) == 0, "colsum: bad args (N, ld multiples of 8)");
    const int rows_per_block = colsum_rows(M);
    dim3 grid((N + 127) / 128, (M + rows_per_block - 1) / rows_per_block), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (x_f32) hipLaunchKernelGGL((colsum_kernel<true>), grid, block, 0, s, x, part, M, N, ld, rows_per_block, (int*)nullptr, (float*)nullptr);
    else hipLaunchKernelGGL((colsum_kernel<false>), grid, block, 0, s, x, part, M, N, ld, rows_per_block, (int*)nullptr, (float*)nullptr);
    return scl_check_launch("scl_colsum");
}

extern "C" int scl_colsum_reduce(const void* x, int x_f32, float* part, int* counters, float* out, int M, int N, int64_t ld, void* stream) {
    SCL_REQUIRE(x && part && counters && out && M > 0 && N > 0 && (N & 7) == 0 && (ld & 7) == 0, "colsum_reduce: bad args (N, ld multiples of 8)");
    SCL_REQUIRE((N + 127) / 128 <= SCL_COLSUM_MAX_GROUPS, "colsum_reduce: N too large for the counter array (%d)", N);
    const int rows_per_block = colsum_reduce_rows(M, N);
    dim3 grid((N + 127) / 128, (M + rows_per_block - 1) / rows_per_block), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (x_f32) hipLaunchKernelGGL((colsum_kernel<true>), grid, block, 0, s, x, part, M, N, ld, rows_per_block, counters, out);
    else hipLaunchKernelGGL((colsum_kernel<false>), grid, block, 0, s, x, part, M, N, ld, rows_per_block, counters, out);
    return scl_check_launch("scl_colsum_reduce");
}
