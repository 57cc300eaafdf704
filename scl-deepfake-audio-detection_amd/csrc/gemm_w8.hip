// gemm_w8.hip — the wide-tile member of the bf16 GEMM family: C = epilogue(alpha * A * B^T), one 8-wave workgroup per CU.
//
// Why it exists (round-1 profile of the kernels in gemm.hip, the encoder GEMMs reached from model/xlsr.py:41 and their
// backward): at M = B*T = 64*199 = 12736 rows the 128x128 tiles give 800 / 2400 / 3200 tiles for 512 resident blocks — 22 % of
// the block slots idle in the last round — and every K step pays ~10 VALU address instructions per LDS-DMA load.
//   * The tile is [tile_m <= 16*(RB0+RB1) rows] x 256 columns, with tile_m a RUNTIME row pitch: the host picks the number of
//     row tiles first and then the pitch, so that rows x columns tiles fill whole rounds of the 256 CUs (12736 rows -> 62 row
//     tiles of 206 rows; N = 1024 / 3072 / 4096 -> 248 / 744 / 992 tiles = 1 / 3 / 4 rounds).  Rows past the pitch are
//     fetched as zeros by the hardware range check (no memory traffic) and not stored.
//   * 8 waves as 2 (M) x 4 (N).  Wave row 0 owns RB0 16-row blocks, wave row 1 owns RB1 (7 + 6 for the 208-row tile: the
//     two waves that share a SIMD are one of each row, so every SIMD issues the same 13 x 4 MFMAs per 32-deep step; 8 + 8
//     for the 256-row tile).  The two wave rows run half a phase apart (ping-pong): while one issues its MFMA burst under
//     s_setprio(1), the other does its LDS fragment reads and LDS-DMA issues.
//   * Staging is global -> LDS directly (buffer_load_dwordx4 ... lds), swizzles applied to the per-lane source address as in
//     gemm.hip.  The per-lane byte offset is computed ONCE per tile; the K advance is the instruction's scalar offset (one
//     s_add per K step; soffset is not part of the range check, so out-of-range lanes stay out of range).
//   * LDS: 2 buffers x {A image 32 KiB, B image 32 KiB} = 128 KiB.  Hazards (raw s_barrier, counted vmcnt), per K tile t:
//       p0: read B(blocks 0,1) A(blocks 0-3) | DMA A(t+1) -> other buffer        | MFMA A0-3 x B0-1
//       p1: read B(blocks 2,3)               |                                    | MFMA A0-3 x B2-3
//       p2: read A(blocks 4..)               |                                    | MFMA A4.. x B2-3
//       p3:                                  | DMA B(t+2) -> this buffer; vmcnt(4): all of tile t+1 landed | MFMA A4.. x B0-1
//     RAW: tile t+1 is waited for (vmcnt) before the first barrier of p3 and first read in the next phase; WAR: a region is
//     re-staged >= 2 phases after its last ds_read, which covers the wave row that runs half a phase behind.
//   * Accumulation order per output element is k ascending, as in the 128x128 kernels: results are bit-identical to theirs.
#include "gemm_common.h"

using namespace sclg;

namespace {

constexpr int W8_BN = 256;
constexpr int W8_OPB = 2 * TILE_BYTES;      // one operand image of one K step: 256 rows x 64 k (or 2 sub-tiles of [64 k][128])
constexpr int W8_BUF = 2 * W8_OPB;
constexpr int W8_LDS = 2 * W8_BUF;          // 131072 B

// K-contiguous operand, 256 rows: this wave stages the 1-KiB pieces wave*4 + i (8 rows x 64 k each)
struct W8K {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned voff[4];
    __device__ __forceinline__ void init(const OpK& o, const char* base, int row0, int rowlimit, int lane, int wave) {
        rsrc = make_rsrc(base);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 8 * (wave * 4 + i) + (lane >> 3);
            const int r = row0 + row;
            const int kc = (lane & 7) ^ ((row >> 1) & 7);
            voff[i] = r < rowlimit ? row_off(o, (unsigned)r) + (unsigned)(kc << 4) : OOB;
        }
    }
    __device__ __forceinline__ void issue(char* img, int wave, unsigned soff, bool live) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned off = live ? voff[i] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(img + (wave * 4 + i) * 1024), 16, off, soff, 0, 0);
        }
    }
    __device__ __forceinline__ static unsigned kstep(const OpK& o) { return o.cin_shift == 6 ? o.cout_bytes : 128u; }
};
// transposed operand, 2 sub-tiles of [64 k rows][128 contiguous]: piece p -> sub-tile p >> 4, k rows 4*(p & 15) .. +3
struct W8T {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned voff[4];
    __device__ __forceinline__ void init(const OpK& o, const char* base, int col0, int collimit, int lane, int wave) {
        rsrc = make_rsrc(base);
        const int s16 = lane & 15;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = wave * 4 + i;
            const int kr = 4 * (p & 15) + (lane >> 4);
            const int sw = (kr & 3) | (((kr >> 3) & 1) << 2);
            const int col = col0 + (p >> 4) * 128 + 8 * ((((s16 >> 1) ^ sw) << 1) | (s16 & 1));
            voff[i] = col < collimit ? (unsigned)kr * o.ld_bytes + col_off(o, (unsigned)col) : OOB;
        }
    }
    __device__ __forceinline__ void issue(char* img, int wave, unsigned soff, bool live) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned off = live ? voff[i] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(img + (wave * 4 + i) * 1024), 16, off, soff, 0, 0);
        }
    }
    __device__ __forceinline__ static unsigned kstep(const OpK& o) { return 64u * o.ld_bytes; }
};
template <bool T> struct W8Sel { typedef W8K type; };
template <> struct W8Sel<true> { typedef W8T type; };

template <bool T>
__device__ __forceinline__ bf16x8 w8_frag(const char* img, int gb, int ks, int lane) {   // gb: 16-row (column) block 0..15 of the image
    if (T) return frag_t_raw(img + (gb >> 3) * TILE_BYTES, gb & 7, ks, lane);
    return frag_k(img, gb, ks, lane);
}

#define W8_MFMA_PHASE(MH, NH, FBSEL, NI)                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    __builtin_amdgcn_s_barrier();                                                                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    __builtin_amdgcn_s_setprio(1);                                                                         \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                       \
        _Pragma("unroll") for (int i = 0; i < (NI); ++i)                                                   \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                  \
                acc[MH][i][2 * (NH) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FBSEL[j][ks], fa[i][ks], acc[MH][i][2 * (NH) + j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    __builtin_amdgcn_s_barrier();

// main loop + epilogue of one wave whose row blocks are ab .. ab + 4 + NH - 1 of the A image
template <bool AT, bool BT, int NH>
__device__ __forceinline__ void w8_body(const GemmK& d, char* smem, const typename W8Sel<AT>::type& la, const typename W8Sel<BT>::type& lb,
                                        int nk, unsigned soffA, unsigned soffB, int ab, int m0, int n0, int mlimit,
                                        int z1, int z2, int ksplit, int lane, int wave, int wc) {
    const unsigned stepA = W8Sel<AT>::type::kstep(d.A), stepB = W8Sel<BT>::type::kstep(d.B);
    f32x4 acc[2][4][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
    const int nbk = wc * 4;
    for (int kt = 0; kt < nk; ++kt) {
        char* buf = smem + (kt & 1) * W8_BUF;
        char* obuf = smem + ((kt & 1) ^ 1) * W8_BUF;
        const char* tA = buf;
        const char* tB = buf + W8_OPB;
        // ---- p0
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int j = 0; j < 2; ++j) fb0[j][ks] = w8_frag<BT>(tB, nbk + j, ks, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i][ks] = w8_frag<AT>(tA, ab + i, ks, lane);
        }
        soffA += stepA;
        la.issue(obuf, wave, soffA, kt + 1 < nk);
        W8_MFMA_PHASE(0, 0, fb0, 4)
        // ---- p1
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int j = 0; j < 2; ++j) fb1[j][ks] = w8_frag<BT>(tB, nbk + 2 + j, ks, lane);
        }
        W8_MFMA_PHASE(0, 1, fb1, 4)
        // ---- p2
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < NH; ++i) fa[i][ks] = w8_frag<AT>(tA, ab + 4 + i, ks, lane);
        }
        W8_MFMA_PHASE(1, 1, fb1, NH)
        // ---- p3
        soffB += stepB;
        lb.issue(buf + W8_OPB, wave, soffB, kt + 2 < nk);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // everything but B(kt+2): tile kt+1 has landed (this wave's pieces)
        W8_MFMA_PHASE(1, 0, fb0, NH)
    }
    if ((wave >> 2) == 0) __builtin_amdgcn_s_barrier();      // balance the stagger
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // no DMA may still target this block's LDS when it retires
    gemm_epilogue_blk<4>(d, acc[0], m0 + ab * 16, n0 + wc * 64, mlimit, 4, z1, z2, ksplit, lane);
    if (NH > 0) gemm_epilogue_blk<4>(d, acc[1], m0 + (ab + 4) * 16, n0 + wc * 64, mlimit, NH, z1, z2, ksplit, lane);
}

template <bool AT, bool BT, int RB0, int RB1>
__global__ __launch_bounds__(512, 2) void scl_gemm_w8_kernel(const GemmK d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7
    const int wr = wave >> 2, wc = wave & 3;
    const int tiles_m = (d.M + d.tile_m - 1) / d.tile_m, tiles_n = (d.N + W8_BN - 1) / W8_BN;
    int tm, tn;
    tile_coords(blockIdx.x, gridDim.x, tiles_m, tiles_n, tm, tn, d.group_m);
    const int m0 = tm * d.tile_m, n0 = tn * W8_BN;
    const int mlimit = min(d.M, m0 + d.tile_m);
    int z = blockIdx.z;
    const int ksplit = z % d.splitk; z /= d.splitk;
    const int z1 = z / d.nb2, z2 = z - z1 * d.nb2;
    const int nk_total = d.K / BK;                                  // K % 64 == 0 (checked on the host)
    const int nk_per = (nk_total + d.splitk - 1) / d.splitk;
    const int kt0 = ksplit * nk_per;
    const int nk = max(0, min(nk_per, nk_total - kt0));
    const char* Ab = reinterpret_cast<const char*>(d.A.ptr) + z1 * d.A.bs1 + z2 * d.A.bs2;
    const char* Bb = reinterpret_cast<const char*>(d.B.ptr) + z1 * d.B.bs1 + z2 * d.B.bs2;

    typename W8Sel<AT>::type la;
    typename W8Sel<BT>::type lb;
    la.init(d.A, Ab, m0, mlimit, lane, wave);
    lb.init(d.B, Bb, n0, d.N, lane, wave);
    const unsigned stepA = W8Sel<AT>::type::kstep(d.A), stepB = W8Sel<BT>::type::kstep(d.B);
    unsigned soffA = (unsigned)kt0 * stepA, soffB = (unsigned)kt0 * stepB;

    // prologue: all of tile 0, B of tile 1
    la.issue(smem, wave, soffA, nk > 0);
    lb.issue(smem + W8_OPB, wave, soffB, nk > 0);
    soffB += stepB;
    lb.issue(smem + W8_BUF + W8_OPB, wave, soffB, nk > 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();          // wave row 1 runs half a phase behind wave row 0

    if (RB0 == RB1) {
        w8_body<AT, BT, RB0 - 4>(d, smem, la, lb, nk, soffA, soffB, wr * RB0, m0, n0, mlimit, z1, z2, ksplit, lane, wave, wc);
    } else if (wr == 0) {
        w8_body<AT, BT, RB0 - 4>(d, smem, la, lb, nk, soffA, soffB, 0, m0, n0, mlimit, z1, z2, ksplit, lane, wave, wc);
    } else {
        w8_body<AT, BT, RB1 - 4>(d, smem, la, lb, nk, soffA, soffB, RB0, m0, n0, mlimit, z1, z2, ksplit, lane, wave, wc);
    }
}

template <int RB0, int RB1>
void w8_launch_rb(const GemmK& k, bool at, bool bt, dim3 grid, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)scl_gemm_w8_kernel<false, false, RB0, RB1>, hipFuncAttributeMaxDynamicSharedMemorySize, W8_LDS);
        (void)hipFuncSetAttribute((const void*)scl_gemm_w8_kernel<false, true, RB0, RB1>, hipFuncAttributeMaxDynamicSharedMemorySize, W8_LDS);
        (void)hipFuncSetAttribute((const void*)scl_gemm_w8_kernel<true, false, RB0, RB1>, hipFuncAttributeMaxDynamicSharedMemorySize, W8_LDS);
        (void)hipFuncSetAttribute((const void*)scl_gemm_w8_kernel<true, true, RB0, RB1>, hipFuncAttributeMaxDynamicSharedMemorySize, W8_LDS);
        attr_set = true;
    }
    const dim3 block(512);
    if (!at && !bt) hipLaunchKernelGGL((scl_gemm_w8_kernel<false, false, RB0, RB1>), grid, block, W8_LDS, s, k);
    else if (!at && bt) hipLaunchKernelGGL((scl_gemm_w8_kernel<false, true, RB0, RB1>), grid, block, W8_LDS, s, k);
    else if (at && !bt) hipLaunchKernelGGL((scl_gemm_w8_kernel<true, false, RB0, RB1>), grid, block, W8_LDS, s, k);
    else hipLaunchKernelGGL((scl_gemm_w8_kernel<true, true, RB0, RB1>), grid, block, W8_LDS, s, k);
}

}  // namespace

namespace sclg {

// Tile plan: the row-block split (7+6 = 208 rows or 8+8 = 256 rows) and the row pitch that need the fewest MFMA-rounds on
// `ncu` CUs; returns 0 when the wide kernel cannot address the operands (caller falls back to gemm.hip's kernels).
bool scl_gemm_w8_plan(const GemmK& k, bool at, bool bt, const SclGemmDesc& d, long long zdim, int ncu, W8Plan* plan) {
    auto flat_rows = [](const SclOperand& o, long long rows) { return (long long)o.rpb >= rows; };
    const bool a_ok = at ? flat_rows(d.A, d.K) : (d.A.cin == 0x7fffffff || d.A.cin == 64 || d.A.cin >= d.K);
    const bool b_ok = bt ? flat_rows(d.B, d.K) : (d.B.cin == 0x7fffffff || d.B.cin == 64 || d.B.cin >= d.K);
    if (!a_ok || !b_ok || (d.K % BK) != 0) return false;
    const long long tiles_n = (d.N + W8_BN - 1) / W8_BN;
    long long best = -1;
    for (int v = 0; v < 2; ++v) {
        const int bm = v == 0 ? 208 : 256, blocks = v == 0 ? 13 : 16;
        const long long ntm = (d.M + bm - 1) / bm;
        const long long rounds = (ntm * tiles_n * zdim + ncu - 1) / ncu;
        const long long cost = rounds * blocks;
        if (best < 0 || cost < best) {
            best = cost;
            plan->variant = v; plan->tiles_m = (int)ntm; plan->tile_m = (int)((d.M + ntm - 1) / ntm);
            plan->tiles = ntm * tiles_n; plan->cost = cost;
        }
    }
    return true;
}

int scl_gemm_w8_launch(GemmK& k, bool at, bool bt, const W8Plan& plan, long long zdim, hipStream_t s) {
    k.tile_m = plan.tile_m;
    const dim3 grid((unsigned)plan.tiles, 1, (unsigned)zdim);
    if (plan.variant == 0) w8_launch_rb<7, 6>(k, at, bt, grid, s);
    else w8_launch_rb<8, 8>(k, at, bt, grid, s);
    return 0;
}

}  // namespace sclg
