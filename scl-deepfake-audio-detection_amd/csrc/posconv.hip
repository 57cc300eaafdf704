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


// ---- weight gradient ------------------------------------------------------------------------------------------------------------
// dW[g][o][tap * 64 + c] = sum_b sum_t dY[b][t][g*64 + o] * X[b][t + tap][g*64 + c].  As a GEMM (both operands transposed, M = 64,
// N = 8192, K = B*T rows, 16 groups) it ran on the 128 x 128 kernel at ~540 TFLOP/s (400 us), re-fetching the X rows once per tap.
// Here a workgroup owns (group, block of 8 taps), one wave per tap, and walks the utterances: dY [T][64] and the 8-tap window of
// X [T + 7][64] are staged per utterance (double-buffered LDS-DMA, one barrier per utterance), every wave keeps its [64 o][64 c] block
// in 64 accumulator registers across all utterances and writes it once.  Both operands are read with ds_read_b64_tr_b16 from
// [t][channel] images (32-byte chunk index ^= bits 1 and 3 of the row: the 8 rows of a half-wave read land in distinct bank groups).
constexpr int PW_TAPS = 8;
constexpr int PW_TROWS = 224;                        // 7 k steps of 32 rows: T <= 224, rows >= T are zero-filled
constexpr int PW_XROWS = PW_TROWS + PW_TAPS;         // 232
constexpr int PW_BUF = (PW_TROWS + PW_XROWS) * 128;  // 58368 B per utterance
constexpr int PW_LDS = 2 * PW_BUF;                   // 116736 B

__device__ __forceinline__ int pw_f(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }
// the two 64-bit halves of the fragment [16 channels (block cb)][32 rows t0 .. t0+31] of a [row][64 ch] image as an MFMA operand
// (lane: channel lane & 15, 8 consecutive rows).  Inline asm (see frag_t_raw): the builtin would be fenced with s_waitcnt vmcnt(0)
// against the LDS-DMA of the next utterance.  The reads are asynchronous and the compiler does not know it: the caller passes every
// half through PW_WAIT_FRAGS (s_waitcnt lgkmcnt(0) with the halves as read-write operands) before anything may copy or use them.
__device__ __forceinline__ void pw_read(const char* img, int t0, int cb, int lane, s16x4& lo, s16x4& hi) {
    const int i = lane & 15, g = lane >> 4;
    const int ra = t0 + 8 * g + (i >> 2), rb = ra + 4;
    const unsigned aa = (unsigned)(uintptr_t)(lds_void*)(img + ra * 128 + ((cb ^ pw_f(ra)) << 5) + ((i & 3) << 3));
    const unsigned ab = (unsigned)(uintptr_t)(lds_void*)(img + rb * 128 + ((cb ^ pw_f(rb)) << 5) + ((i & 3) << 3));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(aa) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(ab) : "memory");
}
#define PW_WAIT_FRAGS(A, B, C, D)                                                                                             \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(B[0]), "+v"(B[1]), "+v"(B[2]), "+v"(B[3]), \
                                          "+v"(C[0]), "+v"(C[1]), "+v"(C[2]), "+v"(C[3]), "+v"(D[0]), "+v"(D[1]), "+v"(D[2]), "+v"(D[3])  \
                 :: "memory")
__device__ __forceinline__ bf16x8 pw_join(s16x4 lo, s16x4 hi) {
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(512, 2) void posconv_wgrad_kernel(const bf16_t* __restrict__ dypad, const bf16_t* __restrict__ xpad, float* __restrict__ dw,
                                                               int Bn, int T, int K, int G, int dy_row0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntb = K / PW_TAPS;
    const int g = blockIdx.x / ntb, tb = blockIdx.x % ntb;
    const int E = G * PC_CG, Tp = T + K;
    const int tap = tb * PW_TAPS + wave;
    // per-lane source offsets of the staging pieces (8 rows x 128 B each): dest chunk q = lane & 7 of row r holds the logical 16-byte
    // chunk (((q >> 1) ^ f(r)) * 2 + (q & 1))
    const int q = lane & 7, rl = lane >> 3;
    constexpr int NPIECE = (PW_TROWS + PW_XROWS) / 8;      // 57
    unsigned voff[(NPIECE + 7) / 8];
    bool isdy[(NPIECE + 7) / 8];
#pragma unroll
    for (int i = 0; i < (NPIECE + 7) / 8; ++i) {
        const int p = wave + 8 * i;
        voff[i] = OOB; isdy[i] = p < PW_TROWS / 8;
        if (p < NPIECE) {
            const int r = (p < PW_TROWS / 8 ? 8 * p : 8 * (p - PW_TROWS / 8)) + rl;      // row inside its image
            const int ch = (((q >> 1) ^ pw_f(r)) << 1) | (q & 1);
            if (p < PW_TROWS / 8) { if (r < T) voff[i] = (unsigned)(dy_row0 + r) * (unsigned)(E * 2) + (unsigned)(ch << 4); }
            else { if (tb * PW_TAPS + r < Tp) voff[i] = (unsigned)(tb * PW_TAPS + r) * (unsigned)(E * 2) + (unsigned)(ch << 4); }
        }
    }
    const long long ubytes = (long long)Tp * E * 2;      // one utterance of either padded tensor
    const char* dyb = reinterpret_cast<const char*>(dypad + g * PC_CG);
    const char* xb = reinterpret_cast<const char*>(xpad + g * PC_CG);
    auto stage = [&](int b, char* buf) {
        const __amdgpu_buffer_rsrc_t rdy = make_rsrc(dyb + b * ubytes), rx = make_rsrc(xb + b * ubytes);
#pragma unroll
        for (int i = 0; i < (NPIECE + 7) / 8; ++i) {
            const int p = wave + 8 * i;
            if (p < NPIECE) __builtin_amdgcn_raw_ptr_buffer_load_lds(isdy[i] ? rdy : rx, (lds_void*)(buf + p * 1024), 16, voff[i], 0, 0, 0);
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    stage(0, smem);
    for (int b = 0; b < Bn; ++b) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of utterance b
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                          // everyone's; everyone is past its reads of utterance b-1 (the other buffer)
        char* buf = smem + (b & 1) * PW_BUF;
        if (b + 1 < Bn) stage(b + 1, smem + ((b + 1) & 1) * PW_BUF);
        const char* dyi = buf;
        const char* xi = buf + PW_TROWS * 128;      // X rows tb*8 ..: this wave's tap reads row t + wave
#pragma unroll
        for (int ks = 0; ks < PW_TROWS / 32; ++ks) {
            s16x4 alo[4], ahi[4], blo[4], bhi[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) pw_read(dyi, 32 * ks, i, lane, alo[i], ahi[i]);
#pragma unroll
            for (int j = 0; j < 4; ++j) pw_read(xi, 32 * ks + wave, j, lane, blo[j], bhi[j]);
            PW_WAIT_FRAGS(alo, ahi, blo, bhi);
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { fa[i] = pw_join(alo[i], ahi[i]); fb[i] = pw_join(blo[i], bhi[i]); }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
    }
    // lane holds dW[o = 16 i + (lane & 15)][c = 16 j + 4 (lane >> 4) .. + 3] of its tap
    const int lc = lane & 15, gq = lane >> 4;
    float* dst = dw + ((long long)g * PC_CG) * ((long long)K * PC_CG) + (long long)tap * PC_CG;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<float4*>(dst + (long long)(16 * i + lc) * (K * PC_CG) + 16 * j + 4 * gq) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
}

}  // namespace

extern "C" int scl_posconv_supported(int T, int K, int G, int Cg) {
    return (Cg == PC_CG && T >= 1 && T <= 16 * PC_TILES && K >= 2 && K <= 128 && (K & 1) == 0 && G >= 1) ? 1 : 0;
}

extern "C" int scl_posconv_wgrad_supported(int T, int K, int G, int Cg) {
    return (Cg == PC_CG && T >= 1 && T <= PW_TROWS && K >= PW_TAPS && (K % PW_TAPS) == 0 && G >= 1) ? 1 : 0;
}

extern "C" int scl_posconv_wgrad(const void* dypad, int dy_row0, const void* xpad, float* dw, int B, int T, int K, int G, int Cg, void* stream) {
    SCL_REQUIRE(dypad && xpad && dw && B > 0 && dy_row0 >= 0 && dy_row0 <= K, "posconv_wgrad: bad arguments");
    SCL_REQUIRE(scl_posconv_wgrad_supported(T, K, G, Cg), "posconv_wgrad: needs 64 channels per group, T <= 224, K a multiple of 8 (got T=%d K=%d Cg=%d)", T, K, Cg);
    SCL_REQUIRE(((uintptr_t)dypad & 15) == 0 && ((uintptr_t)xpad & 15) == 0 && ((uintptr_t)dw & 15) == 0, "posconv_wgrad: operands must be 16-byte aligned");
    SCL_REQUIRE((long long)(T + K) * G * Cg * 2 < 0x7FFFFFFFll, "posconv_wgrad: utterance slab too large for 32-bit offsets");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)posconv_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PW_LDS);
        attr_set = true;
    }
    hipStream_t s = (hipStream_t)stream;
    SclProfScope prof(SCL_KID_GEMM, s, 2.0 * B * T * (double)Cg * K * Cg * G, true);
    prof.note(Cg * K, Cg * G, B * T, 2, G, 5);
    SCL_LAUNCH(posconv_wgrad_kernel, dim3((unsigned)(G * (K / PW_TAPS))), dim3(512), PW_LDS, s, (const bf16_t*)dypad, (const bf16_t*)xpad, dw, B, T, K, G,
               dy_row0);
    return scl_check_launch("scl_posconv_wgrad");
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
    prof.note(B * T, Cg * G, Cg * K, fwd ? 1 : 0, G, 5);
    if (fwd) SCL_LAUNCH((posconv_mfma_kernel<true>), grid, block, PC_LDS, s, (const bf16_t*)xpad, (const bf16_t*)w, C, bias, (bf16_t*)c2, R, B, T, K, G);
    else SCL_LAUNCH((posconv_mfma_kernel<false>), grid, block, PC_LDS, s, (const bf16_t*)xpad, (const bf16_t*)w, C, bias, (bf16_t*)c2, R, B, T, K, G);
    return scl_check_launch("scl_posconv_mfma");
}
