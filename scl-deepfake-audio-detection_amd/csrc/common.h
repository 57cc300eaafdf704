// common.h — shared device helpers for the gfx950 kernels (wave = 64 lanes, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/scl_hip.h"

typedef unsigned short bf16_t;  // raw bfloat16 storage
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

#define SCL_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
    return __builtin_bit_cast(unsigned short, b);
}
// ONE v_cvt_pk_bf16_f32 for the pair: converting the halves separately and merging them cost four instructions (two conversions, a
// shift and an or) in every bf16 store of every epilogue; same round-to-nearest-even bits.
typedef __attribute__((ext_vector_type(2))) __bf16 scl_bf16x2;
typedef __attribute__((ext_vector_type(2))) float scl_f32x2;
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    const scl_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, scl_bf16x2));
}

// erf-GELU (torch.nn.functional.gelu default / fairseq "gelu") with erf by Abramowitz-Stegun 7.1.26:
// |erf error| <= 1.5e-7 absolute — below fp32 round-off of the surrounding arithmetic — at ~12 VALU ops instead of
// libm erff's ~45; GELU sits in the epilogue of every FC1 / conv-stack / pos-conv tile, where erff cost ~30 % of a tile.
// The same exp(-x^2/2) serves the erf tail and the Gaussian pdf of the gradient.
// Every fused multiply-add below is written out and contraction is switched off inside these functions: the scalar form (128x128
// kernels, LayerNorm+GELU, element-wise edge paths) and the two-at-a-time packed form (wide GEMM epilogue: v_pk_fma_f32 /
// v_pk_mul_f32, the reciprocal and the exponential stay scalar) perform the same operations in the same order whatever code surrounds
// them, so the kernels that share a result agree to the last bit.
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& e) {
#pragma clang fp contract(off)
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
    e = __expf(-(z * z));   // = exp(-x^2 / 2)
    float p = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
    p = __builtin_fmaf(t, p, 1.421413741f);
    p = __builtin_fmaf(t, p, -0.284496736f);
    p = __builtin_fmaf(t, p, 0.254829592f);
    const float half_tail = (0.5f * (t * p)) * e;          // = 0.5 * erfc(|x| / sqrt 2)
    cdf = x >= 0.f ? 1.0f - half_tail : half_tail;
}
__device__ __forceinline__ float gelu_f(float x) {
    float cdf, e;
    gelu_parts(x, cdf, e);
    return x * cdf;
}
__device__ __forceinline__ float gelu_grad_f(float x) {
#pragma clang fp contract(off)
    float cdf, e;
    gelu_parts(x, cdf, e);
    return __builtin_fmaf(x * 0.39894228040143268f, e, cdf);
}
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ void gelu_parts2(f32x2 x, f32x2& cdf, f32x2& e) {
#pragma clang fp contract(off)
    const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
    const f32x2 z = ax * 0.70710678118654752f;
    const f32x2 den = __builtin_elementwise_fma(f32x2{0.3275911f, 0.3275911f}, z, f32x2{1.0f, 1.0f});
    const f32x2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
    const f32x2 z2 = z * z;
    e = f32x2{__expf(-z2[0]), __expf(-z2[1])};
    f32x2 p = __builtin_elementwise_fma(t, f32x2{1.061405429f, 1.061405429f}, f32x2{-1.453152027f, -1.453152027f});
    p = __builtin_elementwise_fma(t, p, f32x2{1.421413741f, 1.421413741f});
    p = __builtin_elementwise_fma(t, p, f32x2{-0.284496736f, -0.284496736f});
    p = __builtin_elementwise_fma(t, p, f32x2{0.254829592f, 0.254829592f});
    const f32x2 half_tail = (0.5f * (t * p)) * e;
    cdf = f32x2{x[0] >= 0.f ? 1.0f - half_tail[0] : half_tail[0], x[1] >= 0.f ? 1.0f - half_tail[1] : half_tail[1]};
}
__device__ __forceinline__ void gelu2(float& a, float& b) {
    f32x2 cdf, e;
    gelu_parts2(f32x2{a, b}, cdf, e);
    a = a * cdf[0]; b = b * cdf[1];      // outside the packed part, as gelu_f: a following "+ residual" contracts the same way
}
__device__ __forceinline__ void gelu_grad2(float& a, float& b) {      // a, b := gelu'(a), gelu'(b)
#pragma clang fp contract(off)
    f32x2 cdf, e;
    const f32x2 x = {a, b};
    gelu_parts2(x, cdf, e);
    const f32x2 y = __builtin_elementwise_fma(x * 0.39894228040143268f, e, cdf);
    a = y[0]; b = y[1];
}
// GELU and its derivative from ONE evaluation of the shared parts (activation id 5: the forward epilogue stores gelu'(pre-activation) as
// its second output, so the data-gradient epilogue that consumes it multiplies by a stored number instead of re-evaluating erf and exp
// per element): value = gelu_f's x * cdf, derivative = gelu_grad_f's fma — scalar and packed forms agree to the last bit.
__device__ __forceinline__ float gelu_grad_from_parts(float x, float cdf, float e) {
#pragma clang fp contract(off)
    return __builtin_fmaf(x * 0.39894228040143268f, e, cdf);
}
__device__ __forceinline__ void gelu_both_f(float x, float& y, float& dy) {
    float cdf, e;
    gelu_parts(x, cdf, e);
    dy = gelu_grad_from_parts(x, cdf, e);
    y = x * cdf;
}
__device__ __forceinline__ f32x2 gelu_grad_from_parts2(f32x2 x, f32x2 cdf, f32x2 e) {
#pragma clang fp contract(off)
    return __builtin_elementwise_fma(x * 0.39894228040143268f, e, cdf);
}
__device__ __forceinline__ void gelu_both2(float& a, float& b, float& da, float& db) {      // a, b := gelu; da, db := gelu'
    f32x2 cdf, e;
    const f32x2 x = {a, b};
    gelu_parts2(x, cdf, e);
    const f32x2 g = gelu_grad_from_parts2(x, cdf, e);
    da = g[0]; db = g[1];
    a = a * cdf[0]; b = b * cdf[1];
}
// activation ids shared with SCL_GEMM_ACT_SHIFT: 0 none, 1 gelu, 2 relu, 3 leaky_relu(0.01), 5 gelu with C2 = gelu'(pre-activation)
__device__ __forceinline__ float act_f(int id, float x) {
    switch (id) {
        case 1: case 5: return gelu_f(x);
        case 2: return x > 0.f ? x : 0.f;
        case 3: return x > 0.f ? x : 0.01f * x;
        default: return x;
    }
}
__device__ __forceinline__ float act_grad_f(int id, float x) {
    switch (id) {
        case 1: return gelu_grad_f(x);
        case 2: return x > 0.f ? 1.f : 0.f;
        case 3: return x > 0.f ? 1.f : 0.01f;
        case 4: return x;      // RACT 4: R already holds the derivative (written by an ACT 5 forward epilogue)
        default: return 1.f;
    }
}

// counter-based dropout keep-mask: one 32-bit hash of (seed, element index); keep iff u >= p.
__device__ __forceinline__ uint32_t hash_u32(uint32_t seed, uint64_t idx) {
    uint32_t x = (uint32_t)idx * 0x9E3779B1u ^ (uint32_t)(idx >> 32) * 0x85EBCA77u ^ seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    x += seed * 0xC2B2AE3Du; x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15;
    return x;
}
__device__ __forceinline__ float dropout_scale(uint32_t seed, uint64_t idx, float p) {
    const float u = (float)(hash_u32(seed, idx) >> 8) * (1.0f / 16777216.0f);
    return u >= p ? 1.0f / (1.0f - p) : 0.0f;
}

// Wave-wide reductions on the DPP data path (no LDS crossbar): __shfl_xor lowers to ds_bpermute_b32 on gfx950 — six DEPENDENT
// LDS round trips (~100 cycles each) per reduction, which a kernel with two waves per SIMD (conv0_bwd: two reductions per frame)
// cannot hide.  Here: two quad permutes, row_half_mirror, row_mirror (every lane of a 16-lane row then holds the row's value),
// row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3 (lane 63 holds the total) and one v_readlane.  Fixed order:
// deterministic; the association differs from the butterfly's, so sums may differ from round 2's in the last bits.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float scl_dpp(float v, float old) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
#define SCL_WAVE_REDUCE(OP, IDENT)                                                           \
    v = OP(v, scl_dpp<0xB1, 0xF>(v, v));   /* quad_perm [1,0,3,2] */                          \
    v = OP(v, scl_dpp<0x4E, 0xF>(v, v));   /* quad_perm [2,3,0,1] */                          \
    v = OP(v, scl_dpp<0x141, 0xF>(v, v));  /* row_half_mirror */                              \
    v = OP(v, scl_dpp<0x140, 0xF>(v, v));  /* row_mirror */                                   \
    v = OP(v, scl_dpp<0x142, 0xA>(v, IDENT));  /* row_bcast:15 -> rows 1, 3 */                \
    v = OP(v, scl_dpp<0x143, 0xC>(v, IDENT));  /* row_bcast:31 -> rows 2, 3 */                \
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
__device__ __forceinline__ float scl_addf(float a, float b) { return a + b; }
// sums over aligned groups of 8 / 16 lanes, every lane of the group ends up with the group's total; same pairing tree as the
// __shfl_xor 1, 2, 4(, 8) butterfly (bitwise equal), on the DPP path
__device__ __forceinline__ float lanes8_sum(float v) {
    v += scl_dpp<0xB1, 0xF>(v, v);
    v += scl_dpp<0x4E, 0xF>(v, v);
    v += scl_dpp<0x141, 0xF>(v, v);
    return v;
}
__device__ __forceinline__ float lanes16_sum(float v) {
    v = lanes8_sum(v);
    v += scl_dpp<0x140, 0xF>(v, v);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) { SCL_WAVE_REDUCE(scl_addf, 0.f) }
__device__ __forceinline__ float wave_max(float v) { SCL_WAVE_REDUCE(fmaxf, -INFINITY) }
__device__ __forceinline__ float wave_min(float v) { SCL_WAVE_REDUCE(fminf, INFINITY) }
#undef SCL_WAVE_REDUCE

// host side ---------------------------------------------------------------------------------
void scl_set_error(const char* fmt, ...);
int  scl_check_launch(const char* what);

struct SclProfScope {  // brackets a launch with two events when profiling of `kid` is on
    int kid; hipStream_t s; void* slot;
    hipEvent_t ea, eb;      // dispatch mode: handed to the ONE kernel launch of the scope (SCL_LAUNCH) as its start / stop events
    bool dispatch, taken;
    SclProfScope(int kid, hipStream_t s, double flops, bool dispatch = false);
    ~SclProfScope();
    void note(int M, int N, int K, int flags, int z, int variant);      // what this launch is (scl_prof_read_launches)
};
// The scope a launch on this thread belongs to (dispatch mode only).  The kernel's own dispatch packet then carries the two time
// stamps (hipExtLaunchKernelGGL) instead of two hipEventRecord barrier packets around it, which cost ~20 us of queue bubble each
// — 13 ms over the 330 GEMM launches of bench.py's profiled step.
extern thread_local SclProfScope* scl_prof_active;
#define SCL_LAUNCH(kern, grid, block, lds, stream, ...)                                                                     \
    do {                                                                                                                    \
        SclProfScope* ps_ = scl_prof_active;                                                                                \
        if (ps_ && ps_->slot && !ps_->taken) {                                                                              \
            ps_->taken = true;                                                                                              \
            hipExtLaunchKernelGGL(kern, grid, block, lds, stream, ps_->ea, ps_->eb, 0, __VA_ARGS__);                        \
        } else {                                                                                                            \
            hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);                                                \
        }                                                                                                                   \
    } while (0)

#define SCL_REQUIRE(cond, ...)                     \
    do {                                           \
        if (!(cond)) {                             \
            scl_set_error(__VA_ARGS__);            \
            return SCL_EINVAL;                     \
        }                                          \
    } while (0)
