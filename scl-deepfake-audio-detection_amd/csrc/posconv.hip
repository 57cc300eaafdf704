// posconv.hip — the grouped positional convolution of the wav2vec 2.0 encoder (fairseq ConvPositionalEmbedding: Conv1d(E, E, kernel
// 128, padding 64, groups 16) + SamePad + GELU, reached from model/xlsr.py:41) as an implicit GEMM whose input never re-enters LDS.
//
// As a GEMM (round 1-2: scl_gemm_bf16 with a 2-level contiguous index) a group is [B*T rows] x [64 out] x [K = 128 taps * 64 in]: every
// tap step re-fetches a [rows x 64] slice that overlaps the previous one in all but one row, the 128 x 128 tiles compute 128 columns
// where 64 exist, and the launch reached 383 TFLOP/s (558 us forward, the same again for the data gradient).  Here a workgroup owns
// one (utterance, group): the utterance's padded [T + K rows][64 channels] slab is staged into LDS ONCE (42 KiB), tap t of output row r
// reads slab row r + t, and only the weights stream (one [64 out][64 in] tile of 8 KiB per tap through a ring of four LDS-DMA stages,
// two taps per barrier).  8 waves as 4 (row tiles i, i+4, i+8, i+12 of the 13) x 2 (two 16-column tiles): 16 MFMAs per wave and tap.
// 74 KiB of LDS => two workgroups per CU, 4 waves per SIMD.  Accumulation order per output element is the GEMM's (k = tap * 64 + in,
// ascending, 32 per MFMA) and the epilogue arithmetic is gemm_w8_epi.h's (alpha = 1): results are bit-identical to the GEMM path
// (tests/test_kernels_gpu.py).  The data gradient is the same kernel on (dY x gelu' padded, flipped / transposed weights).
#include "gemm_common.h"

using namespace sclg;

namespace {

constexpr int PC_CG = 64;                       // channels per group (in = out)
constexpr int PC_TILES = 13;                    // 16-row output tiles: T <= 208
constexpr int PC_SLAB_ROWS = 16 * PC_TILES + 128;   // rows a fragment read can touch (row tile 12, lane 15, tap 127)
constexpr int PC_SLAB = PC_SLAB_ROWS * 128;     // 43008 B
constexpr int PC_STAGE = PC_CG * 128;           // 8 KiB: [64 out][64 in] of one tap
constexpr int PC_LDS = PC_SLAB + 4 * PC_STAGE;  // 75776 B (the epilogue's f32 [208][64] tile, 53248 B, reuses it)

template <int NI>
__device__ __forceinline__ void pc_tap(f32x4 (&acc)[4][2], const char* slab, const char* stage, int row0, int tap, int wn, int lane) {
    const int lc = lane & 15, g = lane >> 4;
    const int row = row0 + lc + tap;                       // slab row of this lane's output row for this tap
    const int off0 = row * 128 + ((g ^ ((row >> 1) & 7)) << 4);      // k 0..31: chunk g ^ swizzle; k 32..63: chunk (g + 4) ^ swizzle = offset ^ 64
    const int off1 = off0 ^ 64;
    bf16x8 fa[NI][2], fb[2][2];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        fa[i][0] = *reinterpret_cast<const bf16x8*>(slab + off0 + i * (64 * 128));
        fa[i][1] = *reinterpret_cast<const bf16x8*>(slab + off1 + i * (64 * 128));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fb[j][ks] = frag_k(stage, 2 * wn + j, ks, lane);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][ks], fa[i][ks], acc[i][j], 0, 0, 0);
}

template <int NI>
__device__ __forceinline__ void pc_loop(f32x4 (&acc)[4][2], char* smem, __amdgpu_buffer_rsrc_t rw, unsigned voffw, int K, int row0, int wn,
                                        int lane, int wave) {
    char* ring = smem + PC_SLAB;
    for (int t = 0; t < K; t += 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of taps t, t+1 (and, at t = 0, of the slab) have landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                          // everyone's have; everyone is past its reads of taps t-2, t-1
        if (t + 2 < K) {
#pragma unroll
            for (int u = 2; u < 4; ++u)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)(ring + ((t + u) & 3) * PC_STAGE + wave * 1024), 16, voffw,
                                                         (unsigned)(t + u) * 128u, 0, 0);
        }
        pc_tap<NI>(acc, smem, ring + (t & 3) * PC_STAGE, row0, t, wn, lane);
        pc_tap<NI>(acc, smem, ring + ((t + 1) & 3) * PC_STAGE, row0, t + 1, wn, lane);
    }
}

// FWD: C = gelu(acc + bias) + R, C2 = bf16(acc + bias);  !FWD: C = acc + R
template <bool FWD>
__global__ __launch_bounds__(512, 4) void posconv_mfma_kernel(const bf16_t* __restrict__ xpad, const bf16_t* __restrict__ w, float* __restrict__ C,
                                                              const float* __restrict__ bias, bf16_t* __restrict__ c2,
                                                              const float* __restrict__ R, int Bn, int T, int K, int G) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int b = blockIdx.x % Bn, g = blockIdx.x / Bn;      // consecutive blocks share a group's weights (1 MiB, L2-resident)
    const int E = G * PC_CG, Tp = T + K;

    // ---- the utterance's padded slab -> LDS (rows past T + K arrive as zeros: they only feed output rows >= T, which are not stored)
    {
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(reinterpret_cast<const char*>(xpad + ((long long)b * Tp * E + g * PC_CG)));
#pragma unroll
        for (int i = 0; i < (PC_SLAB_ROWS / 8 + 7) / 8; ++i) {
            const int p = wave + 8 * i;                        // 1-KiB piece = slab rows 8p .. 8p+7
            if (p < PC_SLAB_ROWS / 8) {
                const int r = 8 * p + (lane >> 3);
                const int ch = (lane & 7) ^ ((r >> 1) & 7);
                const unsigned voff = r < Tp ? (unsigned)r * (unsigned)(E * 2) + (unsigned)(ch << 4) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)(smem + p * 1024), 16, voff, 0, 0, 0);
            }
        }
    }
    // ---- weights: stage = tap; wave w carries out-channel rows 8w .. 8w+7 of every stage
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(reinterpret_cast<const char*>(w + (long long)g * PC_CG * K * PC_CG));
    unsigned voffw;
    {
        const int r = 8 * wave + (lane >> 3);
        const int ch = (lane & 7) ^ ((r >> 1) & 7);
        voffw = (unsigned)r * (unsigned)(K * PC_CG * 2) + (unsigned)(ch << 4);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)(smem + PC_SLAB + u * PC_STAGE + wave * 1024), 16, voffw, (unsigned)u * 128u, 0, 0);

    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (wm == 0) pc_loop<4>(acc, smem, rw, voffw, K, 16 * wm, wn, lane, wave);      // row tiles 0, 4, 8, 12
    else pc_loop<3>(acc, smem, rw, voffw, K, 16 * wm, wn, lane, wave);              // row tiles wm, wm + 4, wm + 8

    // ---- epilogue through LDS: f32 [208 rows][64 columns], 16-byte chunks XOR-swizzled with row & 15 (as gemm_w8_epi.h), then whole
    // 256-byte rows per 16 lanes.  The residual values of all seven passes are requested first: their round trip runs under the
    // barriers and the LDS transposition instead of once per pass.
    constexpr int PC_PASSES = (16 * PC_TILES) / 32 + 1;
    const int ch = tid & 15;
    const int col = g * PC_CG + 4 * ch;
    float4 rres[PC_PASSES];
#pragma unroll
    for (int ps = 0; ps < PC_PASSES; ++ps) {
        const int row = 32 * ps + (tid >> 4);
        rres[ps] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < T) rres[ps] = *reinterpret_cast<const float4*>(R + ((long long)b * T + row) * E + col);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
        const int lc = lane & 15, gq = lane >> 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (wm + 4 * i < PC_TILES) {
                const int row = 16 * (wm + 4 * i) + lc;
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    *reinterpret_cast<f32x4*>(smem + row * 256 + ((((2 * wn + j) * 4 + gq) ^ lc) << 4)) = acc[i][j];
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float bb[4] = {0.f, 0.f, 0.f, 0.f};
    if (FWD) { const float4 t4 = *reinterpret_cast<const float4*>(bias + col); bb[0] = t4.x; bb[1] = t4.y; bb[2] = t4.z; bb[3] = t4.w; }
#pragma unroll
    for (int ps = 0; ps < PC_PASSES; ++ps) {
        const int row = 32 * ps + (tid >> 4);
        if (row < T) {
            // the GEMM epilogues evaluate gelu(x) = x * cdf and the residual add in separate basic blocks (run-time flags): no FMA
            // forms across them.  Same here, or the last bit differs.
#pragma clang fp contract(off)
            const f32x4 a = *reinterpret_cast<const f32x4*>(smem + row * 256 + ((ch ^ (row & 15)) << 4));
            const long long off = ((long long)b * T + row) * E + col;
            const float4 rr = rres[ps];
            float v[4] = {1.0f * a[0], 1.0f * a[1], 1.0f * a[2], 1.0f * a[3]};
            if (FWD) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] += bb[q];
                *reinterpret_cast<uint2*>(c2 + off) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                gelu2(v[0], v[1]); gelu2(v[2], v[3]);
            }
            v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
            *reinterpret_cast<float4*>(C + off) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

}  // namespace

extern "C" int scl_posconv_supported(int T, int K, int G, int Cg) {
    return (Cg == PC_CG && T >= 1 && T <= 16 * PC_TILES && K >= 2 && K <= 128 && (K & 1) == 0 && G >= 1) ? 1 : 0;
}

extern "C" int scl_posconv_mfma(const void* xpad, const void* w, float* C, const float* bias, void* c2, const float* R, int B, int T, int K,
                                int G, int Cg, int fwd, void* stream) {
    SCL_REQUIRE(xpad && w && C && R && B > 0, "posconv_mfma: null pointer");
    SCL_REQUIRE(scl_posconv_supported(T, K, G, Cg), "posconv_mfma: needs 64 channels per group, T <= 208, even K <= 128 (got T=%d K=%d Cg=%d)", T, K, Cg);
    SCL_REQUIRE(!fwd || (bias && c2), "posconv_mfma: the forward form needs bias and the pre-activation output");
    auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    SCL_REQUIRE(al(xpad) && al(w) && al(C) && al(R) && al(bias) && (!c2 || ((uintptr_t)c2 & 7) == 0), "posconv_mfma: operands must be 16-byte aligned");
    SCL_REQUIRE((long long)(T + K) * G * Cg * 2 < 0x7FFFFFFFll, "posconv_mfma: utterance slab too large for 32-bit offsets");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)posconv_mfma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS);
        (void)hipFuncSetAttribute((const void*)posconv_mfma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS);
        attr_set = true;
    }
    const dim3 grid((unsigned)(B * G)), block(512);
    hipStream_t s = (hipStream_t)stream;
    // counted with the bf16 MFMA GEMM family (bench.py's roofline): the contraction scl_gemm_bf16 performed for this layer before
    SclProfScope prof(SCL_KID_GEMM, s, 2.0 * B * T * (double)Cg * K * Cg * G, true);
    if (fwd) SCL_LAUNCH((posconv_mfma_kernel<true>), grid, block, PC_LDS, s, (const bf16_t*)xpad, (const bf16_t*)w, C, bias, (bf16_t*)c2, R, B, T, K, G);
    else SCL_LAUNCH((posconv_mfma_kernel<false>), grid, block, PC_LDS, s, (const bf16_t*)xpad, (const bf16_t*)w, C, bias, (bf16_t*)c2, R, B, T, K, G);
    return scl_check_launch("scl_posconv_mfma");
}
