// attention.hip — row softmax (fp32 statistics) between the QK^T and PV contractions of
// fairseq MultiheadAttention (reached from model/xlsr.py:41), forward and backward.
//
//   fwd:  P[r][j]  = softmax_j S[r][j]                       S f32 [R, T] (ld = lds), P bf16 [R, Tp]
//   bwd:  dS[r][j] = P[r][j] * (dP[r][j] - sum_j dP[r][j] P[r][j])
//
// One wave per row (T <= 4*64 per pass, looped for longer rows); columns T..Tp-1 of P / dS are
// written as zero so the following transposed-operand GEMMs can read whole 16-byte vectors.
#include "common.h"

namespace {

constexpr int MAXV = 8;  // up to 512 columns kept in registers

__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ S, bf16_t* __restrict__ P, int64_t R,
                                                          int T, int ldS, int Tp) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const float* s = S + row * ldS;
    float v[MAXV];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 64 + lane;
        v[i] = c < T ? s[c] : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 64 + lane;
        v[i] = c < T ? __expf(v[i] - mx) : 0.f;
        sum += v[i];
    }
    const float inv = 1.0f / wave_sum(sum);
    bf16_t* p = P + row * Tp;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < Tp) p[c] = f2bf(v[i] * inv);
    }
}

__global__ __launch_bounds__(256) void softmax_bwd_kernel(const bf16_t* __restrict__ P, const float* __restrict__ dP,
                                                          bf16_t* __restrict__ dS, int64_t R, int T, int lddP, int Tp) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const bf16_t* p = P + row * Tp;
    const float* dp = dP + row * lddP;
    float pv[MAXV], dv[MAXV];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 64 + lane;
        pv[i] = c < T ? bf2f(p[c]) : 0.f;
        dv[i] = c < T ? dp[c] : 0.f;
        dot += pv[i] * dv[i];
    }
    dot = wave_sum(dot);
    bf16_t* o = dS + row * Tp;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < Tp) o[c] = f2bf(pv[i] * (dv[i] - dot));
    }
}

}  // namespace

extern "C" int scl_softmax_fwd(const float* S, void* P, int64_t R, int T, int ldS, int Tp, void* stream) {
    SCL_REQUIRE(S && P && R > 0 && T > 0 && T <= 512 && Tp >= T && Tp <= 512 && (Tp & 7) == 0, "softmax_fwd: need T <= Tp <= 512, Tp %% 8 == 0");
    hipLaunchKernelGGL(softmax_fwd_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, S, (bf16_t*)P, R, T, ldS, Tp);
    return scl_check_launch("scl_softmax_fwd");
}

extern "C" int scl_softmax_bwd(const void* P, const float* dP, void* dS, int64_t R, int T, int lddP, int Tp, void* stream) {
    SCL_REQUIRE(P && dP && dS && R > 0 && T > 0 && T <= 512 && Tp >= T && Tp <= 512 && (Tp & 7) == 0, "softmax_bwd: need T <= Tp <= 512, Tp %% 8 == 0");
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)P, dP, (bf16_t*)dS, R, T, lddP, Tp);
    return scl_check_launch("scl_softmax_bwd");
}
