// gemm_x2.hip — the two-blocks-per-CU member of the bf16 GEMM family: C = epilogue(alpha * A * B^T), 4-wave workgroups, 208 x 128 tiles.
//
// Why it exists (round-2 profile of the encoder GEMMs reached from model/xlsr.py:41 and their backward, profiles/r2_gemm_*): the wide
// kernel of gemm_w8.hip owns a whole CU (8 waves, all 160 KiB of LDS), so nothing runs under its prologue (1.8 us) and its epilogue
// (5-15 us of a 30-39 us block: GELU + two outputs, x gelu'(R), f32 residual read + write are HBM- and VALU-bound, the matrix pipe
// idles).  On the K = 1024 shapes with the heavy epilogues (fc1 forward, fc2 data gradient: 557 TFLOP/s in the step) that is a
// third of the launch.  Here a workgroup is 4 waves with 80 KiB of LDS, so TWO are resident per CU, each on its own tile and its own
// barriers: while one stores its tile the other multiplies, and the two waves that share a SIMD are never in lockstep.
//   * Tile [tile_m <= 208 rows] x 128 columns, tile_m a runtime row pitch as in gemm_w8.hip (12736 rows -> 62 row tiles of 206 rows;
//     N = 1024 / 3072 / 4096 -> 496 / 1488 / 1984 tiles for the 512 workgroup slots).  Waves as 2 (M) x 2 (N): wave row 0 owns 7
//     16-row blocks, wave row 1 six; 64 columns each — the per-wave register image of the wide kernel (112 accumulator registers,
//     fragments double-buffered).
//   * K step 32 (one MFMA depth).  LDS rings of 3 A images (16 KiB slots) + 3 B images (8 KiB): stage t+3 is requested right
//     after the barrier of step t, so a piece has two steps to land; every wave issues exactly 4 + 2 LDS-DMA pieces per step
//     (pieces past the operand are out-of-range requests: zeros into the unused part of the slot), which keeps the vmcnt count
//     uniform.  Per step t:   MFMA(F_t) | vmcnt(6): stage t+1 landed | BARRIER | DMA stage t+3 -> slot of stage t | read F_{t+1}
//     with F double-buffered in registers (the reads of F_{t+1} fly under the MFMAs of F_t issued after them).
//   * K-contiguous images: piece = 16 rows x 32 k (64-byte rows); 16-byte chunk c of row r sits at position c ^ f(r >> 2),
//     f = {0, 3, 2, 1}: the ds_read_b128 lane groups of gfx950 ({0-3, 12-15, 20-27}, ...) then touch every bank once.
//     Transposed images: sub-tiles of [32 k][128 columns], the layout and ds_read_b64_tr_b16 reader of gemm_common.h.
//   * Epilogue: gemm_w8_epi.h (whole lines through a 16 KiB LDS block per wave; 4 x (16 + 4) KiB = the workgroup's 80 KiB).
//   * Accumulation order per output element is k ascending in steps of 32, as in every other kernel of the family: bit-identical.
#ifdef SCL_EXPERIMENTS      // not part of the shipped library (slower than the wide tiles on every encoder shape): see gemm.hip
#include "gemm_common.h"
#include "gemm_w8_epi.h"

using namespace sclg;

namespace {

constexpr int X2_BN = 128;
constexpr int X2_ASLOT = 16384, X2_BSLOT = 8192;
constexpr int X2_LDS = 81920;                 // 3 x 16 + 3 x 8 = 72 KiB of rings; the epilogue uses all 80
constexpr int X2_BOFF = 3 * X2_ASLOT;

__device__ __forceinline__ int x2_swz(int row) { return (4 - ((row >> 2) & 3)) & 3; }      // {0, 3, 2, 1}[(row >> 2) & 3]

// K-contiguous operand, up to 256 rows: piece p = rows 16p .. 16p+15 (64 B each); this wave stages pieces wave*NP + i
template <int NP>
struct X2K {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned voff[NP];
    __device__ __forceinline__ void init(const OpK& o, const char* base, int row0, int rowlimit, int lane, int wave) {
        rsrc = make_rsrc(base);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int row = 16 * (wave * NP + i) + (lane >> 2);
            const int r = row0 + row;
            const int kc = (lane & 3) ^ x2_swz(row);
            voff[i] = r < rowlimit ? row_off(o, (unsigned)r) + (unsigned)(kc << 4) : OOB;
        }
    }
    __device__ __forceinline__ void issue(char* img, int wave, unsigned soff, bool live) const {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const unsigned off = live ? voff[i] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(img + (wave * NP + i) * 1024), 16, off, soff, 0, 0);
        }
    }
    __device__ __forceinline__ static unsigned kstep(const OpK&) { return 64u; }
};
// transposed operand, NP*4/8 sub-tiles of [32 k rows][128 contiguous]: piece p -> sub-tile p >> 3, k rows 4*(p & 7) .. +3
template <int NP>
struct X2T {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned voff[NP];
    __device__ __forceinline__ void init(const OpK& o, const char* base, int col0, int collimit, int lane, int wave) {
        rsrc = make_rsrc(base);
        const int s16 = lane & 15;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int p = wave * NP + i;
            const int kr = 4 * (p & 7) + (lane >> 4);
            const int sw = (kr & 3) | (((kr >> 3) & 1) << 2);
            const int col = col0 + (p >> 3) * 128 + 8 * ((((s16 >> 1) ^ sw) << 1) | (s16 & 1));
            voff[i] = col < collimit ? (unsigned)kr * o.ld_bytes + col_off(o, (unsigned)col) : OOB;
        }
    }
    __device__ __forceinline__ void issue(char* img, int wave, unsigned soff, bool live) const {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const unsigned off = live ? voff[i] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(img + (wave * NP + i) * 1024), 16, off, soff, 0, 0);
        }
    }
    __device__ __forceinline__ static unsigned kstep(const OpK& o) { return 32u * o.ld_bytes; }
};
template <bool T, int NP> struct X2Sel { typedef X2K<NP> type; };
template <int NP> struct X2Sel<true, NP> { typedef X2T<NP> type; };

// fragment of a K-contiguous BK = 32 image: 16 rows x 32 k; lane l holds row (l & 15), k = 8 * (l >> 4) + 0..7
__device__ __forceinline__ bf16x8 x2_frag_k(const char* img, int rowblk, int lane) {
    const int row = rowblk * 16 + (lane & 15);
    return *reinterpret_cast<const bf16x8*>(img + row * 64 + (((lane >> 4) ^ x2_swz(row)) << 4));
}
template <bool T>
__device__ __forceinline__ bf16x8 x2_frag(const char* img, int gb, int lane) {   // gb: 16-row (column) block of the image
    if (T) return frag_t_raw(img + (gb >> 3) * 8192, gb & 7, 0, lane);
    return x2_frag_k(img, gb, lane);
}

template <bool AT, bool BT, int RBW>
__device__ __forceinline__ void x2_body(const GemmK& d, char* smem, const typename X2Sel<AT, 4>::type& la, const typename X2Sel<BT, 2>::type& lb,
                                        int nk, unsigned soffA, unsigned soffB, int ab, int m0, int n0, int mlimit,
                                        int z1, int z2, int ksplit, int lane, int wave, int wc) {
    const unsigned stepA = X2Sel<AT, 4>::type::kstep(d.A), stepB = X2Sel<BT, 2>::type::kstep(d.B);
    f32x4 acc[RBW][4];
#pragma unroll
    for (int i = 0; i < RBW; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fa0[RBW], fa1[RBW], fb0[4], fb1[4];
    const int nbk = wc * 4;
#define X2_READ(FA, FB, SLOT)                                                                                   \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) FB[j] = x2_frag<BT>(smem + X2_BOFF + (SLOT) * X2_BSLOT, nbk + j, lane); \
    _Pragma("unroll") for (int i = 0; i < RBW; ++i) FA[i] = x2_frag<AT>(smem + (SLOT) * X2_ASLOT, ab + i, lane);
#define X2_MFMA(FA, FB)                                                                     \
    _Pragma("unroll") for (int i = 0; i < RBW; ++i)                                         \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                       \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FB[j], FA[i], acc[i][j], 0, 0, 0);
    // one step: F_t is in FA/FB (reads possibly in flight); stage t+1 becomes visible at the barrier; stage t+3 is requested into
    // the slot of stage t, which every wave has read by then (its lgkmcnt(0) precedes the barrier)
#define X2_STEP(FA, FB, FAN, FBN, T)                                                        \
    {                                                                                       \
        const int t_ = (T);                                                                 \
        const int slot_ = t_ % 3;                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                  \
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                    \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        __builtin_amdgcn_s_barrier();                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        const bool live_ = t_ + 3 < nk;                                                     \
        la.issue(smem + slot_ * X2_ASLOT, wave, soffA, live_);                              \
        lb.issue(smem + X2_BOFF + slot_ * X2_BSLOT, wave, soffB, live_);                    \
        soffA += stepA; soffB += stepB;                                                     \
        if (t_ + 1 < nk) { const int sn_ = (t_ + 1) % 3; X2_READ(FAN, FBN, sn_) }           \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        X2_MFMA(FA, FB)                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                  \
    }
    // stage 0 is visible (the caller's barrier); stages 0-2 were requested, soffA / soffB point at stage 3
    X2_READ(fa0, fb0, 0)
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        X2_STEP(fa0, fb0, fa1, fb1, kt)
        X2_STEP(fa1, fb1, fa0, fb0, kt + 1)
    }
    if (kt < nk) X2_STEP(fa0, fb0, fa1, fb1, kt)
#undef X2_STEP
#undef X2_READ
#undef X2_MFMA
    // the tail steps requested out-of-range pieces (zeros) into the rings: none may land in another wave's epilogue block
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
        const long long cbase = z1 * d.c_bs1 + z2 * d.c_bs2 + (long long)ksplit * d.c_split_stride;
        const float* bias = (d.flags & SCL_GEMM_HAS_BIAS) ? d.bias + z2 * d.bias_bs2 : nullptr;
        char* wlds = smem + wave * 16384;
        char* wextra = smem + 4 * 16384 + wave * 4096;
        f32x4 (&alo)[4][4] = *reinterpret_cast<f32x4 (*)[4][4]>(&acc[0]);
        w8_epilogue_pass<4, 0>(d, alo, 4, wlds, wextra, m0 + ab * 16, n0 + wc * 64, mlimit, cbase, bias, lane);
        if (RBW > 4) {
            f32x4 hi[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) hi[i][j] = (4 + i < RBW) ? acc[(4 + i < RBW) ? 4 + i : 0][j] : f32x4{0.f, 0.f, 0.f, 0.f};
            w8_epilogue_pass<4, 0>(d, hi, RBW - 4, wlds, wextra, m0 + (ab + 4) * 16, n0 + wc * 64, mlimit, cbase, bias, lane);
        }
    }
}

template <bool AT, bool BT>
__global__ __launch_bounds__(256, 2) void scl_gemm_x2_kernel(const GemmK d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..3
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_m = (d.M + d.tile_m - 1) / d.tile_m, tiles_n = (d.N + X2_BN - 1) / X2_BN;
    int tm, tn;
    tile_coords(blockIdx.x, gridDim.x, tiles_m, tiles_n, tm, tn, d.group_m);
    const int m0 = tm * d.tile_m, n0 = tn * X2_BN;
    const int mlimit = min(d.M, m0 + d.tile_m);
    int z = blockIdx.z;
    const int ksplit = __builtin_amdgcn_readfirstlane(z % d.splitk); z /= d.splitk;      // uniform, but integer division runs on the vector ALU: back to an SGPR
    const int z1 = __builtin_amdgcn_readfirstlane(z / d.nb2), z2 = z - z1 * d.nb2;
    const int nk_total = d.K / 32;                                  // K % 32 == 0 (checked on the host)
    // split-K slabs begin where the 64-deep kernels' slabs begin, so that a slab holds the same partial sum whichever kernel wrote it
    const int nk_per = 2 * (((d.K + 63) / 64 + d.splitk - 1) / d.splitk);
    const int kt0 = ksplit * nk_per;
    const int nk = max(0, min(nk_per, nk_total - kt0));
    const char* Ab = reinterpret_cast<const char*>(d.A.ptr) + z1 * d.A.bs1 + z2 * d.A.bs2;
    const char* Bb = reinterpret_cast<const char*>(d.B.ptr) + z1 * d.B.bs1 + z2 * d.B.bs2;

    typename X2Sel<AT, 4>::type la;
    typename X2Sel<BT, 2>::type lb;
    la.init(d.A, Ab, m0, mlimit, lane, wave);
    lb.init(d.B, Bb, n0, d.N, lane, wave);
    const unsigned stepA = X2Sel<AT, 4>::type::kstep(d.A), stepB = X2Sel<BT, 2>::type::kstep(d.B);
    unsigned soffA = (unsigned)kt0 * stepA, soffB = (unsigned)kt0 * stepB;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        la.issue(smem + s * X2_ASLOT, wave, soffA, s < nk);
        lb.issue(smem + X2_BOFF + s * X2_BSLOT, wave, soffB, s < nk);
        soffA += stepA; soffB += stepB;
    }
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 0) x2_body<AT, BT, 7>(d, smem, la, lb, nk, soffA, soffB, 0, m0, n0, mlimit, z1, z2, ksplit, lane, wave, wc);
    else x2_body<AT, BT, 6>(d, smem, la, lb, nk, soffA, soffB, 7, m0, n0, mlimit, z1, z2, ksplit, lane, wave, wc);
}

}  // namespace

namespace sclg {

// Tile plan: row tiles of at most 208 rows with the pitch that spreads M evenly; false when the kernel cannot address the operands
// (K-contiguous operands must advance linearly along K, transposed ones must be flat in their reduction rows).
bool scl_gemm_x2_plan(const GemmK& k, bool at, bool bt, const SclGemmDesc& d, long long zdim, int ncu, W8Plan* plan) {
    auto flat_rows = [](const SclOperand& o, long long rows) { return (long long)o.rpb >= rows; };
    const bool a_ok = at ? flat_rows(d.A, d.K) : (d.A.cin == 0x7fffffff || d.A.cin >= d.K);
    const bool b_ok = bt ? flat_rows(d.B, d.K) : (d.B.cin == 0x7fffffff || d.B.cin >= d.K);
    if (!a_ok || !b_ok || (d.K % 32) != 0) return false;
    const long long tiles_n = (d.N + X2_BN - 1) / X2_BN;
    const long long ntm = (d.M + 207) / 208;
    plan->variant = 2; plan->tiles_m = (int)ntm; plan->tile_m = (int)((d.M + ntm - 1) / ntm);
    plan->tiles = ntm * tiles_n;
    plan->cost = ((plan->tiles * zdim + 2 * ncu - 1) / (2 * ncu)) * (208 + X2_BN);
    return true;
}

int scl_gemm_x2_launch(GemmK& k, bool at, bool bt, const W8Plan& plan, long long zdim, hipStream_t s) {
    k.tile_m = plan.tile_m;
    k.debug = 0;
    {   // the epilogue stages R through buffer loads with 32-bit byte offsets: fall back to plain loads for an R extent >= 4 GiB
        const long long nb1 = zdim / ((long long)k.nb2 * k.splitk);
        const long long rows = k.M - 1, rpb = (long long)k.c_rpb;
        const long long maxoff = (nb1 - 1) * k.c_bs1 + ((long long)k.nb2 - 1) * k.c_bs2 + (rows / rpb) * k.c_rbstride + (rows % rpb) * k.ldc + k.N;
        if (maxoff * ((k.flags & SCL_GEMM_R_F32) ? 4 : 2) >= 0xFFFFFF00ll) k.debug |= 2;
    }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)scl_gemm_x2_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, X2_LDS);
        (void)hipFuncSetAttribute((const void*)scl_gemm_x2_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, X2_LDS);
        (void)hipFuncSetAttribute((const void*)scl_gemm_x2_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, X2_LDS);
        (void)hipFuncSetAttribute((const void*)scl_gemm_x2_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, X2_LDS);
        attr_set = true;
    }
    const dim3 grid((unsigned)plan.tiles, 1, (unsigned)zdim), block(256);
    if (!at && !bt) SCL_LAUNCH((scl_gemm_x2_kernel<false, false>), grid, block, X2_LDS, s, k);
    else if (!at && bt) SCL_LAUNCH((scl_gemm_x2_kernel<false, true>), grid, block, X2_LDS, s, k);
    else if (at && !bt) SCL_LAUNCH((scl_gemm_x2_kernel<true, false>), grid, block, X2_LDS, s, k);
    else SCL_LAUNCH((scl_gemm_x2_kernel<true, true>), grid, block, X2_LDS, s, k);
    return 0;
}

}  // namespace sclg
#endif  // SCL_EXPERIMENTS
