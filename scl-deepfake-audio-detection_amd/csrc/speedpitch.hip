// speedpitch.hip — the two conf-5 augmenters on the GPU.
//
//   speed   datautils/audio_augmentor/speed.py:29-33  -> pydub 0.25.1 AudioSegment.speedup: 16-bit integer arithmetic.  The host walks
//           the chunk list (scl_amd/augment.py::speed); one kernel performs AudioSegment.append(chunk, crossfade) IN PLACE on the
//           running output: fade-out of the overlap (audioop.mul with a per-millisecond or per-frame gain ramp), looped fade-in of the
//           incoming chunk, saturating audioop.add, then the untouched tail of the chunk.  Gains are evaluated in fp64 exactly as
//           CPython evaluates them (no fused multiply-add), so the int16 result is bit-exact against the restatement in
//           oracle/audio_speed_pitch.py.
//   pitch   datautils/audio_augmentor/pitch.py:31-38  -> librosa 0.10.0 effects.pitch_shift: STFT (2048 / 512, periodic Hann, centred,
//           zero padded) -> phase vocoder -> inverse STFT -> band-limited resampling -> fix_length.  A 2048-point Stockham radix-2
//           FFT per frame in LDS, one thread per frequency bin for the (sequential in time) phase accumulation, a gather-form
//           overlap-add, and a Kaiser-windowed sinc interpolator evaluated tap by tap (Bessel I0 by its power series in fp64).
#include "common.h"

#pragma clang fp contract(off)

namespace {

// ---- speed ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int mul_i16(int v, double factor) {          // audioop.mul: floor(clip(v * factor))
    double f = (double)v * factor;
    f = f > 32767.0 ? 32767.0 : (f < -32768.0 ? -32768.0 : f);
    return (int)floor(f);
}

struct FadeP {
    double from_power, scale_step;      // gain(i) = from_power + scale_step * i
    int per_ms;                         // i = frame / frames_per_ms (fades longer than 100 ms) or i = frame
    int frames_per_ms;
};

__device__ __forceinline__ double fade_gain(const FadeP& p, int frame) {
    const int i = p.per_ms ? frame / p.frames_per_ms : frame;
    return p.from_power + (p.scale_step * (double)i);
}

// out[a0 + j] = sat( mul(out[a0 + j], g1(j)) + mul(chunk[j % m2], g2(j % m2)) )   j < R;      out[n1 + t] = chunk[tail_off + t]   t < tail_n
__global__ __launch_bounds__(256) void i16_append_xfade_kernel(short* __restrict__ out, int n1, const short* __restrict__ chunk, int a0, int R,
                                                               FadeP f1, int m2, FadeP f2, int tail_off, int tail_n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < R) {
        const int jj = i % m2;
        const int v1 = mul_i16(out[a0 + i], fade_gain(f1, i));
        const int v2 = mul_i16(chunk[jj], fade_gain(f2, jj));
        int s = v1 + v2;
        s = s > 32767 ? 32767 : (s < -32768 ? -32768 : s);
        out[a0 + i] = (short)s;
    } else if (i - R < tail_n) {
        out[n1 + (i - R)] = chunk[tail_off + (i - R)];
    }
}

// ---- pitch: FFT --------------------------------------------------------------------------------------------------------------------------
constexpr int NFFT = 2048, HOP = 512, NBIN = NFFT / 2 + 1;

__device__ __forceinline__ float hann(int n) { return (float)(0.5 - 0.5 * cospi(2.0 * (double)n / (double)NFFT)); }

// 2048-point complex FFT of a[] (LDS), 256 threads, Stockham radix 2; the result is left in a[] or b[] — the pointer is returned.
// tw[k] = exp(-2 pi i k / 2048), k < 1024; `inverse` conjugates it (no 1/N scaling here).
__device__ float2* fft2048(float2* a, float2* b, const float2* tw, int tid, bool inverse) {
    for (int ns = 1; ns < NFFT; ns <<= 1) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = tid + 256 * r;                  // butterfly index, 0..1023
            const int k = j & (ns - 1);
            float2 w = tw[k * (NFFT / 2 / ns)];
            if (inverse) w.y = -w.y;
            const float2 u = a[j], v = a[j + NFFT / 2];
            const float2 t = make_float2(v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x);
            const int d = ((j - k) << 1) + k;
            b[d] = make_float2(u.x + t.x, u.y + t.y);
            b[d + ns] = make_float2(u.x - t.x, u.y - t.y);
        }
        float2* s = a; a = b; b = s;
    }
    __syncthreads();
    return a;
}

__device__ __forceinline__ void fill_twiddles(float2* tw, int tid) {
    for (int k = tid; k < NFFT / 2; k += 256) {
        double s, c;
        sincospi(-2.0 * (double)k / (double)NFFT, &s, &c);
        tw[k] = make_float2((float)c, (float)s);
    }
}

// D[t][k] = sum_n w[n] ypad[t * hop + n] e^{-2 pi i k n / N},  ypad = y with N/2 zeros either side
__global__ __launch_bounds__(256) void stft_kernel(const float* __restrict__ y, int L, float2* __restrict__ D) {
    __shared__ float2 A[NFFT], B[NFFT], tw[NFFT / 2];
    const int t = blockIdx.x, tid = threadIdx.x;
    fill_twiddles(tw, tid);
    for (int n = tid; n < NFFT; n += 256) {
        const int p = t * HOP + n - NFFT / 2;
        A[n] = make_float2(p >= 0 && p < L ? hann(n) * y[p] : 0.f, 0.f);
    }
    const float2* X = fft2048(A, B, tw, tid, false);
    for (int k = tid; k < NBIN; k += 256) D[(size_t)t * NBIN + k] = X[k];
}

// librosa.phase_vocoder: one thread per bin, sequential over the output frames
__global__ __launch_bounds__(256) void phase_vocoder_kernel(const float2* __restrict__ D, int nfr, double rate, float2* __restrict__ out, int nsteps) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= NBIN) return;
    const double two_pi = 6.283185307179586476925286766559;
    const double phi = (double)k * ((3.14159265358979323846 * (double)HOP) / (double)(NBIN - 1));
    const float2 d0 = D[k];
    float phase_acc = atan2f(d0.y, d0.x);
    for (int t = 0; t < nsteps; ++t) {
        const double step = (double)t * rate;
        const int i0 = (int)step;
        const double alpha = step - floor(step);
        const float2 c0 = i0 < nfr ? D[(size_t)i0 * NBIN + k] : make_float2(0.f, 0.f);
        const float2 c1 = i0 + 1 < nfr ? D[(size_t)(i0 + 1) * NBIN + k] : make_float2(0.f, 0.f);
        const double mag = (1.0 - alpha) * (double)hypotf(c0.x, c0.y) + alpha * (double)hypotf(c1.x, c1.y);
        float s, c;
        sincosf(phase_acc, &s, &c);
        out[(size_t)t * NBIN + k] = make_float2((float)(mag * (double)c), (float)(mag * (double)s));
        const float dang = atan2f(c1.y, c1.x) - atan2f(c0.y, c0.x);
        double dphase = (double)dang - phi;
        dphase = dphase - two_pi * rint(dphase / two_pi);
        phase_acc = (float)((double)phase_acc + (phi + dphase));
    }
}

// F[t][n] = w[n] * irfft(Ds[t])[n]
__global__ __launch_bounds__(256) void istft_frames_kernel(const float2* __restrict__ Ds, float* __restrict__ F) {
    __shared__ float2 A[NFFT], B[NFFT], tw[NFFT / 2];
    const int t = blockIdx.x, tid = threadIdx.x;
    fill_twiddles(tw, tid);
    for (int k = tid; k < NBIN; k += 256) {
        const float2 v = Ds[(size_t)t * NBIN + k];
        A[k] = v;
        if (k > 0 && k < NFFT / 2) A[NFFT - k] = make_float2(v.x, -v.y);
    }
    const float2* X = fft2048(A, B, tw, tid, true);
    for (int n = tid; n < NFFT; n += 256) F[(size_t)t * NFFT + n] = hann(n) * (X[n].x * (1.0f / NFFT));
}

// y[n] = (sum_t F[t][p - t hop]) / (sum_t w[p - t hop]^2),  p = n + N/2, over the frames that cover p, in increasing t
__global__ __launch_bounds__(256) void istft_ola_kernel(const float* __restrict__ F, int nfr, float* __restrict__ y, int length) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= length) return;
    const int p = n + NFFT / 2;
    int t0 = (p - (NFFT - 1) + HOP - 1) / HOP;
    if (t0 < 0) t0 = 0;
    int t1 = p / HOP;
    if (t1 > nfr - 1) t1 = nfr - 1;
    float acc = 0.f, wss = 0.f;
    for (int t = t0; t <= t1; ++t) {
        const int m = p - t * HOP;
        acc += F[(size_t)t * NFFT + m];
        const float w = hann(m);
        wss += w * w;
    }
    y[n] = wss > 1.17549435e-38f ? acc / wss : acc;
}

// ---- pitch: resampling -----------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double bessel_i0(double x) {
    const double q = 0.25 * x * x;
    double term = 1.0, sum = 1.0;
    for (int k = 1; k < 64; ++k) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < 1e-17 * sum) break;
    }
    return sum;
}

__global__ __launch_bounds__(256) void resample_sinc_kernel(const float* __restrict__ x, int n_in, double ratio, float* __restrict__ out, int n_out,
                                                            double fc, double W, double beta, double inv_i0b) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= n_out) return;
    const double t = (double)n / ratio;
    int k0 = (int)ceil(t - W), k1 = (int)floor(t + W);
    if (k0 < 0) k0 = 0;
    if (k1 > n_in - 1) k1 = n_in - 1;
    double acc = 0.0;
    for (int k = k0; k <= k1; ++k) {
        const double u = t - (double)k;
        const double r = u / W;
        const double win = bessel_i0(beta * sqrt(fmax(0.0, 1.0 - r * r))) * inv_i0b;
        const double a = 2.0 * fc * u;
        const double sinc = a == 0.0 ? 1.0 : sinpi(a) / (3.14159265358979323846 * a);
        acc += (double)x[k] * (2.0 * fc * sinc * win);
    }
    out[n] = (float)acc;
}

double host_i0(double x) {
    const double q = 0.25 * x * x;
    double term = 1.0, sum = 1.0;
    for (int k = 1; k < 200; ++k) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < 1e-17 * sum) break;
    }
    return sum;
}

inline int blocks_for(int64_t n) { return (int)((n + 255) / 256); }

}  // namespace

extern "C" int scl_i16_append_xfade(void* out_i16, int n1, const void* chunk_i16, int n2, int a0, int R, int per_ms1, double from1, double step1,
                                    int m2, int per_ms2, double from2, double step2, int tail_off, int tail_n, int frames_per_ms, void* stream) {
    SCL_REQUIRE(out_i16 && chunk_i16 && n1 >= 0 && n2 >= 0 && a0 >= 0 && R >= 0 && tail_n >= 0 && tail_off >= 0, "append_xfade: bad args");
    SCL_REQUIRE(a0 + R <= n1 && tail_off + tail_n <= n2 && (R == 0 || (m2 >= 1 && m2 <= n2)) && frames_per_ms >= 1, "append_xfade: ranges");
    if (R + tail_n == 0) return SCL_OK;
    FadeP f1{from1, step1, per_ms1, frames_per_ms}, f2{from2, step2, per_ms2, frames_per_ms};
    hipLaunchKernelGGL(i16_append_xfade_kernel, dim3(blocks_for(R + tail_n)), dim3(256), 0, (hipStream_t)stream, (short*)out_i16, n1,
                       (const short*)chunk_i16, a0, R, f1, m2 > 0 ? m2 : 1, f2, tail_off, tail_n);
    return scl_check_launch("scl_i16_append_xfade");
}

extern "C" int scl_stft_nframes(int L) { return 1 + L / HOP; }

extern "C" int scl_stft_f32(const float* y, int L, void* D_c64, int nframes, void* stream) {
    SCL_REQUIRE(y && D_c64 && L >= 1 && nframes == scl_stft_nframes(L), "stft: bad args");
    hipLaunchKernelGGL(stft_kernel, dim3(nframes), dim3(256), 0, (hipStream_t)stream, y, L, (float2*)D_c64);
    return scl_check_launch("scl_stft_f32");
}

extern "C" int scl_phase_vocoder_c64(const void* D_c64, int nframes, double rate, void* out_c64, int nsteps, void* stream) {
    SCL_REQUIRE(D_c64 && out_c64 && nframes >= 1 && nsteps >= 1 && rate > 0.0, "phase_vocoder: bad args");
    hipLaunchKernelGGL(phase_vocoder_kernel, dim3(blocks_for(NBIN)), dim3(256), 0, (hipStream_t)stream, (const float2*)D_c64, nframes, rate,
                       (float2*)out_c64, nsteps);
    return scl_check_launch("scl_phase_vocoder_c64");
}

extern "C" int scl_istft_f32(const void* D_c64, int nframes, float* frames_ws, float* y, int length, void* stream) {
    SCL_REQUIRE(D_c64 && frames_ws && y && nframes >= 1 && length >= 1, "istft: bad args");
    hipLaunchKernelGGL(istft_frames_kernel, dim3(nframes), dim3(256), 0, (hipStream_t)stream, (const float2*)D_c64, frames_ws);
    hipLaunchKernelGGL(istft_ola_kernel, dim3(blocks_for(length)), dim3(256), 0, (hipStream_t)stream, frames_ws, nframes, y, length);
    return scl_check_launch("scl_istft_f32");
}

extern "C" int scl_resample_sinc_f32(const float* x, int n_in, double ratio, float* out, int n_out, void* stream) {
    SCL_REQUIRE(x && out && n_in >= 1 && n_out >= 1 && ratio > 0.0, "resample_sinc: bad args");
    const double zeros = 32.0, beta = 14.769656459379492, rolloff = 0.95;
    const double fc = 0.5 * rolloff * (ratio < 1.0 ? ratio : 1.0);
    hipLaunchKernelGGL(resample_sinc_kernel, dim3(blocks_for(n_out)), dim3(256), 0, (hipStream_t)stream, x, n_in, ratio, out, n_out, fc,
                       zeros / (2.0 * fc), beta, 1.0 / host_i0(beta));
    return scl_check_launch("scl_resample_sinc_f32");
}
