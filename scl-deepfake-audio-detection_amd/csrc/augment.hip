// augment.hip — on-the-fly waveform augmentation on the GPU: RawBoost LnL / ISD / SSI filter chains,
// RIR (reverb) convolution, MUSAN overlay in pydub's int16 arithmetic, multi-view crop.
//
// Reference (host numpy / scipy / pydub code run inside 8 DataLoader workers):
//   datautils/RawBoost.py:51-56   filterFIR     y[n] = sum_k b[k] x[n + N//2 - k], N = len(b)+1
//   datautils/RawBoost.py:59-69   LnL           sum_{i<N_f} filterFIR(x^(i+1), b_i); -mean; normWav(.,0)
//   datautils/RawBoost.py:73-84   ISD           y[p] = x[p] (1 + g_sd u1 u2); normWav(.,0)
//   datautils/RawBoost.py:89-97   SSI           x + noise_f * ||x|| / ||noise_f|| / 10^(SNR/20)
//   datautils/audio_augmentor/reverb.py:33-44            np.convolve full, / max|.|, -> int16
//   datautils/audio_augmentor/background_noise.py:40-56  gain (floor+clip) then saturating add, int16
//   datautils/audio_augmentor/utils.py:24-30             float -> int16 by C cast (wraps at +1.0)
//   core_scripts/data_io/wav_augmentation.py:209-282     batch_pad_for_multiview
// Random draws (taps, positions, gains, noise) are sampled on the host with the reference's
// distributions and passed in; the kernels are deterministic functions of their inputs.
//
// FIR kernel: direct form in fp32 with a register sliding window, 4096 outputs per workgroup (16 consecutive outputs per lane: one
// LDS read of the entering sample feeds 16 FMAs), taps staged through LDS in chunks of 256 and read four at a time, the x^p window of
// the chunk staged in LDS with a +1-per-16 pad against bank conflicts; all N_f power branches are accumulated in registers, so a clip
// is read once and written once (8 B/sample of HBM traffic).  The same kernel is the RIR convolution (one branch, 8000-16000 taps).
#include "common.h"

namespace {

constexpr int FIR_TILE = 4096, FIR_TC = 256;
__device__ __forceinline__ int padi(int i) { return i + (i >> 4); }

__device__ __forceinline__ float ipow(float x, int p) {
    float r = x;
    for (int i = 1; i < p; ++i) r *= x;
    return r;
}

// block-level reduction of (sum, min, max, sumsq) -> part[4]
__device__ __forceinline__ void block_stats_store(float s, float mn, float mx, float sq, float* __restrict__ dst) {
    __shared__ float red[4][4];
    s = wave_sum(s); sq = wave_sum(sq); mn = wave_min(mn); mx = wave_max(mx);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wv][0] = s; red[wv][1] = mn; red[wv][2] = mx; red[wv][3] = sq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        dst[0] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
        dst[1] = fminf(fminf(red[0][1], red[1][1]), fminf(red[2][1], red[3][1]));
        dst[2] = fmaxf(fmaxf(red[0][2], red[1][2]), fmaxf(red[2][2], red[3][2]));
        dst[3] = red[0][3] + red[1][3] + red[2][3] + red[3][3];
    }
}

// Register sliding window, 16 consecutive outputs per lane: for tap k the lane's 16 outputs read x[n .. n + 15 + h - k], so moving to
// the next tap shifts the window by ONE sample — one LDS read (the entering sample) feeds 16 FMAs, the taps come four at a time by a
// broadcast ds_read_b128.  The round-2 form (8 outputs per lane, one tap read per tap) issued 2 LDS reads per 8 FMAs and ran at the
// LDS rate (0.18 of the fp32 vector peak).  4096 outputs per workgroup, taps in chunks of 256 (the x^p window of a chunk is staged
// once: 4352 samples per 1 Mi FMAs); a partial last chunk runs only its own 16-tap groups.  The window index is padded by one word
// per 16 (lane stride 17 words: the 32 lanes of a ds_read_b32 group hit 32 different banks).
// MODE 0: scalar source, the compiler packs pairs of outputs into v_pk_fma_f32 (128 per 16 taps + ~40 v_mov for pairs that start at an
// odd window offset); MODE 2 (SCL_FIR_MODE=2, A/B only): one v_fma_f32 per output and tap through inline asm.  Measured on MI355X
// (tools/fir_probe.py, 64 clips x 64000 x 5 branches): 55 TFLOP/s either way, and an explicit aligned-pair form (two window copies, one
// shifted by a sample) the same — the inner loop costs ~8 cycles per v_pk_fma_f32 and ~4 per other vector instruction per SIMD
// whatever the packing, i.e. ~79 TFLOP/s is the fp32 FMA rate of the vector pipe; beyond that only the f32 matrix cores go.
template <int MODE>
__global__ __launch_bounds__(256) void fir_kernel(const float* __restrict__ x, int64_t ldx, int Lin, const float* __restrict__ taps,
                                                  const int* __restrict__ tap_off, const int* __restrict__ tap_len,
                                                  const int* __restrict__ tap_h, int nf, int use_pow, float* __restrict__ y,
                                                  int64_t ldy, int Lout, float* __restrict__ part) {
    __shared__ float win[FIR_TILE + FIR_TC + (FIR_TILE + FIR_TC) / 16 + 24];
    __shared__ __attribute__((aligned(16))) float bt[FIR_TC];
    const int clip = blockIdx.y;
    const int n0 = blockIdx.x * FIR_TILE;
    const int t = threadIdx.x;
    const float* xc = x + (int64_t)clip * ldx;
    float out[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) out[i] = 0.f;
    for (int f = 0; f < nf; ++f) {
        const int len = tap_len[clip * nf + f], off = tap_off[clip * nf + f], h = tap_h[clip * nf + f];
        const int p = use_pow ? f + 1 : 1;
        for (int k0 = 0; k0 < len; k0 += FIR_TC) {
            // window sample idx holds x^p[ws + idx]; output i of lane t, chunk tap j = k0 + FIR_TC - 1 - jj reads window 16 t + i + jj
            const int ws = n0 + h - (k0 + FIR_TC - 1);
            const int nvalid = min(FIR_TC, len - k0);
            const int jstart = (FIR_TC - nvalid) & ~15;              // taps of a partial chunk sit at the END of the reversed order
            __syncthreads();
            for (int idx = t + jstart; idx < FIR_TILE + FIR_TC + 1; idx += 256) {
                const int xi = ws + idx;
                win[padi(idx)] = (xi >= 0 && xi < Lin) ? ipow(xc[xi], p) : 0.f;
            }
            bt[t] = (k0 + FIR_TC - 1 - t) < len ? taps[off + k0 + FIR_TC - 1 - t] : 0.f;      // reversed: bt[jj] multiplies window offset jj
            __syncthreads();
            {
                float r[32];
#pragma unroll
                for (int i = 0; i < 16; ++i) r[i] = win[padi(16 * t + jstart + i)];
                for (int j0 = jstart; j0 < FIR_TC; j0 += 16) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) r[16 + i] = win[padi(16 * t + j0 + 16 + i)];
#pragma unroll
                    for (int u4 = 0; u4 < 16; u4 += 4) {
                        const float4 c4 = *reinterpret_cast<const float4*>(bt + j0 + u4);
                        const float c[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
                        for (int v = 0; v < 4; ++v)
#pragma unroll
                            for (int i = 0; i < 16; ++i) {
                                if (MODE == 2) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(out[i]) : "v"(c[v]), "v"(r[i + u4 + v]));
                                else out[i] = __builtin_fmaf(c[v], r[i + u4 + v], out[i]);
                            }
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) r[i] = r[16 + i];
                }
            }
        }
    }
    float s = 0.f, sq = 0.f, mn = INFINITY, mx = -INFINITY;
    float* yc = y + (int64_t)clip * ldy;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int n = n0 + 16 * t + i;
        if (n < Lout) {
            yc[n] = out[i];
            s += out[i]; sq += out[i] * out[i]; mn = fminf(mn, out[i]); mx = fmaxf(mx, out[i]);
        }
    }
    __syncthreads();
    if (part) block_stats_store(s, mn, mx, sq, part + ((int64_t)clip * gridDim.x + blockIdx.x) * 4);
}

__global__ __launch_bounds__(256) void clip_stats_kernel(const float* __restrict__ x, int64_t ldx, int L, float* __restrict__ part) {
    const int clip = blockIdx.y;
    const int n0 = blockIdx.x * FIR_TILE;
    const float* xc = x + (int64_t)clip * ldx;
    float s = 0.f, sq = 0.f, mn = INFINITY, mx = -INFINITY;
    for (int i = threadIdx.x; i < FIR_TILE; i += 256) {
        const int n = n0 + i;
        if (n < L) { const float v = xc[n]; s += v; sq += v * v; mn = fminf(mn, v); mx = fmaxf(mx, v); }
    }
    block_stats_store(s, mn, mx, sq, part + ((int64_t)clip * gridDim.x + blockIdx.x) * 4);
}

// y[p[i]] *= 1 + g_sd * f_r[i]   (positions of one clip are distinct: a permutation prefix)
__global__ void isd_scatter_kernel(float* __restrict__ y, int64_t ldy, const int* __restrict__ pos, const float* __restrict__ fr,
                                   const int* __restrict__ clip_off, float g_sd) {
    const int clip = blockIdx.y;
    const int b = clip_off[clip], e = clip_off[clip + 1];
    for (int i = b + blockIdx.x * blockDim.x + threadIdx.x; i < e; i += gridDim.x * blockDim.x) {
        float* yy = y + (int64_t)clip * ldy + pos[i];
        const float v = *yy;
        *yy = v + g_sd * v * fr[i];
    }
}

// modes of clip_affine
enum { AFF_CENTER_PEAK_COND = 0, AFF_PEAK_COND = 1, AFF_PEAK_ALWAYS = 2, AFF_SSI_MIX = 3, AFF_PEAK_QUANT_I16 = 4 };

__device__ __forceinline__ void reduce_parts(const float* __restrict__ part, int nblk, float& s, float& mn, float& mx, float& sq) {
    s = 0.f; sq = 0.f; mn = INFINITY; mx = -INFINITY;
    for (int i = 0; i < nblk; ++i) {
        s += part[i * 4 + 0]; mn = fminf(mn, part[i * 4 + 1]); mx = fmaxf(mx, part[i * 4 + 2]); sq += part[i * 4 + 3];
    }
}

// out = f(x [, z]) per mode, with per-clip statistics taken from the block partials
__global__ __launch_bounds__(256) void clip_affine_kernel(int mode, const float* __restrict__ x, int64_t ldx, const float* __restrict__ z,
                                                          int64_t ldz, const float* __restrict__ partx, const float* __restrict__ partz,
                                                          int nblk, const float* __restrict__ snr_db, float* __restrict__ out,
                                                          int64_t ldo, int L) {
    const int clip = blockIdx.y;
    float s, mn, mx, sq;
    reduce_parts(partx + (int64_t)clip * nblk * 4, nblk, s, mn, mx, sq);
    float mu = 0.f, a = 1.f, b = 0.f;
    if (mode == AFF_CENTER_PEAK_COND) {
        mu = s / (float)L;
        const float peak = fmaxf(mx - mu, mu - mn);
        a = peak > 1.f ? 1.f / peak : 1.f;
    } else if (mode == AFF_PEAK_COND) {
        const float peak = fmaxf(mx, -mn);
        a = peak > 1.f ? 1.f / peak : 1.f;
    } else if (mode == AFF_PEAK_ALWAYS || mode == AFF_PEAK_QUANT_I16) {
        a = 1.f / fmaxf(mx, -mn);
    } else if (mode == AFF_SSI_MIX) {
        float s2, mn2, mx2, sq2;
        reduce_parts(partz + (int64_t)clip * nblk * 4, nblk, s2, mn2, mx2, sq2);
        a = sqrtf(sq2 / sq) / powf(10.f, 0.05f * snr_db[clip]);  // x = filtered noise, z = clean signal
        b = 1.f;
    }
    const float* xc = x + (int64_t)clip * ldx;
    const float* zc = z ? z + (int64_t)clip * ldz : nullptr;
    float* oc = out + (int64_t)clip * ldo;
    const int n0 = blockIdx.x * FIR_TILE;
    for (int i = threadIdx.x; i < FIR_TILE; i += 256) {
        const int n = n0 + i;
        if (n >= L) break;
        float v = a * (xc[n] - mu);
        if (mode == AFF_SSI_MIX) v += b * zc[n];
        if (mode == AFF_PEAK_QUANT_I16) {
            // librosa_to_pydub: int16(v * 32768) by C cast: truncate toward zero, wrap modulo 2^16
            const float q = truncf(v * 32768.f);
            int iq = (int)q;
            iq = ((iq + 32768) & 0xFFFF) - 32768;
            v = (float)iq;
        }
        oc[n] = v;
    }
}

// ---- int16 (pydub / audioop) arithmetic ------------------------------------------------------
__global__ void f32_to_i16_wrap_kernel(const float* __restrict__ x, short* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double q = trunc((double)x[i] * 32768.0);
        long long iq = (long long)q;
        iq = ((iq + 32768) & 0xFFFF) - 32768;
        out[i] = (short)iq;
    }
}
__global__ __launch_bounds__(256) void i16_sumsq_kernel(const short* __restrict__ x, int64_t n, unsigned long long* __restrict__ part) {
    __shared__ unsigned long long red[256];
    unsigned long long s = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const long long v = x[i];
        s += (unsigned long long)(v * v);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
// out = sat_add( floor(clip(speech * factor)), noise[0:min] ), result as float of the int16 value
__global__ void i16_gain_overlay_kernel(const short* __restrict__ speech, int64_t n, const short* __restrict__ noise, int64_t nn,
                                        double factor, float* __restrict__ out_f32, short* __restrict__ out_i16) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double v = (double)speech[i] * factor;
        v = v > 32767.0 ? 32767.0 : (v < -32768.0 ? -32768.0 : v);
        int a = (int)floor(v);
        if (i < nn) {
            a += (int)noise[i];
            a = a > 32767 ? 32767 : (a < -32768 ? -32768 : a);
        }
        if (out_f32) out_f32[i] = (float)a;
        if (out_i16) out_i16[i] = (short)a;
    }
}

// ---- multi-view crop (batch_pad_for_multiview) -------------------------------------------------
// view v (length len[v], at src + off[v]) is first cut / tiled / zero-padded to firstlen, the set is
// tiled again when firstlen < length (repeat_pad), then all views share the crop [start, start+length)
__global__ void multiview_crop_kernel(const float* __restrict__ src, const int64_t* __restrict__ off, const int* __restrict__ len,
                                      int V, int firstlen, int start, int out_len, int repeat_pad, float* __restrict__ out,
                                      int64_t ldo) {
    const int v = blockIdx.y;
    const float* sv = src + off[v];
    const int lv = len[v];
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < out_len; n += gridDim.x * blockDim.x) {
        int m = start + n;
        if (repeat_pad && firstlen > 0) m = m % firstlen;   // second-level tiling (new_len < length)
        float val = 0.f;
        if (m < firstlen) {
            if (m < lv) val = sv[m];
            else if (repeat_pad) val = sv[m % lv];
        }
        out[(int64_t)v * ldo + n] = val;
    }
}

inline int blocks_for(int64_t n) { int64_t b = (n + 255) / 256; return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b)); }

}  // namespace

extern "C" int scl_fir_nblocks(int Lout) { return (Lout + FIR_TILE - 1) / FIR_TILE; }

extern "C" int scl_fir_multi_f32(const float* x, int64_t ldx, int Lin, const float* taps, const int* tap_off, const int* tap_len,
                                 const int* tap_h, int nclip, int nf, int use_pow, float* y, int64_t ldy, int Lout, float* part,
                                 void* stream) {
    SCL_REQUIRE(x && taps && tap_off && tap_len && tap_h && y, "fir: null pointer");
    SCL_REQUIRE(nclip >= 1 && nclip <= 65535 && nf >= 1 && Lin >= 1 && Lout >= 1, "fir: bad dims");
    dim3 grid(scl_fir_nblocks(Lout), nclip), block(256);
    SclProfScope prof(SCL_KID_AUG, (hipStream_t)stream, 0.0);
    const int mode = 0;      // the three instruction forms measured equal (fir_kernel); the switch is gone
    if (mode == 2) hipLaunchKernelGGL(fir_kernel<2>, grid, block, 0, (hipStream_t)stream, x, ldx, Lin, taps, tap_off, tap_len, tap_h, nf, use_pow, y, ldy, Lout, part);
    else hipLaunchKernelGGL(fir_kernel<0>, grid, block, 0, (hipStream_t)stream, x, ldx, Lin, taps, tap_off, tap_len, tap_h, nf, use_pow, y, ldy, Lout, part);
    return scl_check_launch("scl_fir_multi_f32");
}

extern "C" int scl_clip_stats_f32(const float* x, int64_t ldx, int L, int nclip, float* part, void* stream) {
    SCL_REQUIRE(x && part && L >= 1 && nclip >= 1 && nclip <= 65535, "clip_stats: bad args");
    dim3 grid(scl_fir_nblocks(L), nclip), block(256);
    SclProfScope prof(SCL_KID_AUG, (hipStream_t)stream, 0.0);
    hipLaunchKernelGGL(clip_stats_kernel, grid, block, 0, (hipStream_t)stream, x, ldx, L, part);
    return scl_check_launch("scl_clip_stats_f32");
}

extern "C" int scl_isd_scatter_f32(float* y, int64_t ldy, const int* pos, const float* fr, const int* clip_off, int nclip, int max_per_clip,
                                   float g_sd, void* stream) {
    SCL_REQUIRE(y && pos && fr && clip_off && nclip >= 1 && nclip <= 65535, "isd_scatter: bad args");
    if (max_per_clip <= 0) return SCL_OK;
    dim3 grid(blocks_for(max_per_clip), nclip), block(256);
    SclProfScope prof(SCL_KID_AUG, (hipStream_t)stream, 0.0);
    hipLaunchKernelGGL(isd_scatter_kernel, grid, block, 0, (hipStream_t)stream, y, ldy, pos, fr, clip_off, g_sd);
    return scl_check_launch("scl_isd_scatter_f32");
}

extern "C" int scl_clip_affine_f32(int mode, const float* x, int64_t ldx, const float* z, int64_t ldz, const float* partx,
                                   const float* partz, const float* snr_db, float* out, int64_t ldo, int L, int nclip, void* stream) {
    SCL_REQUIRE(x && partx && out && L >= 1 && nclip >= 1 && nclip <= 65535 && mode >= 0 && mode <= 4, "clip_affine: bad args");
    SCL_REQUIRE(mode != AFF_SSI_MIX || (z && partz && snr_db), "clip_affine: SSI mode needs z, partz, snr_db");
    const int nblk = scl_fir_nblocks(L);
    dim3 grid(nblk, nclip), block(256);
    SclProfScope prof(SCL_KID_AUG, (hipStream_t)stream, 0.0);
    hipLaunchKernelGGL(clip_affine_kernel, grid, block, 0, (hipStream_t)stream, mode, x, ldx, z, ldz, partx, partz, nblk, snr_db, out, ldo, L);
    return scl_check_launch("scl_clip_affine_f32");
}

extern "C" int scl_f32_to_i16_wrap(const float* x, void* out_i16, int64_t n, void* stream) {
    SCL_REQUIRE(x && out_i16 && n > 0, "f32_to_i16: bad args");
    hipLaunchKernelGGL(f32_to_i16_wrap_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, x, (short*)out_i16, n);
    return scl_check_launch("scl_f32_to_i16_wrap");
}

extern "C" int scl_i16_sumsq(const void* x_i16, int64_t n, uint64_t* part64, int nparts, void* stream) {
    SCL_REQUIRE(x_i16 && part64 && n > 0 && nparts >= 1 && nparts <= 1024, "i16_sumsq: bad args");
    hipLaunchKernelGGL(i16_sumsq_kernel, dim3(nparts), dim3(256), 0, (hipStream_t)stream, (const short*)x_i16, n, (unsigned long long*)part64);
    return scl_check_launch("scl_i16_sumsq");
}

extern "C" int scl_i16_gain_overlay(const void* speech_i16, int64_t n, const void* noise_i16, int64_t nn, double factor, float* out_f32,
                                    void* out_i16, void* stream) {
    SCL_REQUIRE(speech_i16 && n > 0 && (out_f32 || out_i16) && (nn == 0 || noise_i16), "i16_gain_overlay: bad args");
    hipLaunchKernelGGL(i16_gain_overlay_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, (const short*)speech_i16, n,
                       (const short*)noise_i16, nn, factor, out_f32, (short*)out_i16);
    return scl_check_launch("scl_i16_gain_overlay");
}

extern "C" int scl_multiview_crop_f32(const float* src, const int64_t* off, const int* len, int V, int firstlen, int start, int out_len,
                                      int repeat_pad, float* out, int64_t ldo, void* stream) {
    SCL_REQUIRE(src && off && len && out && V >= 1 && V <= 65535 && firstlen >= 1 && out_len >= 1 && start >= 0, "multiview_crop: bad args");
    dim3 grid(blocks_for(out_len), V), block(256);
    hipLaunchKernelGGL(multiview_crop_kernel, grid, block, 0, (hipStream_t)stream, src, off, len, V, firstlen, start, out_len, repeat_pad, out, ldo);
    return scl_check_launch("scl_multiview_crop_f32");
}
