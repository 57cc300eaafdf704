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
//     for the 256-row tile).
//   * Two K loops over the same LDS images, bit-identical results.  The DEFAULT for every layout is the single-barrier loop
//     (w8s_body, further down: one barrier per 64-deep step, MFMA bursts in quarters with the other sub-step's fragment reads and
//     the LDS-DMA pieces between them, counted LDS waits).  The two-barrier ping-pong (w8_body, described here) serves the
//     utterance-batched K rows of the conv-stack weight gradients and SCL_W8_MODE=0: its two wave rows run half a phase apart —
//     while one issues its MFMA burst under s_setprio(1), the other does its LDS fragment reads and LDS-DMA issues.
//   * Staging is global -> LDS directly (buffer_load_dwordx4 ... lds), swizzles applied to the per-lane source address as in
//     gemm.hip.  The per-lane byte offset is computed ONCE per tile; the K advance is the instruction's scalar offset (one
//     s_add per K step; soffset is not part of the range check, so out-of-range lanes stay out of range).
//   * LDS (all 160 KiB): A images in a ring of 3, B images in a ring of 2, 32 KiB each; two DMA pieces per wave and phase (the
//     round-1 ping-pong kernel issued 4 in p0 and 4 in p3: its load segments, not its MFMA bursts, set the phase time), eight
//     pieces (8 KiB per wave) stay in flight across the wait.  Per K tile t (raw s_barrier, counted vmcnt):
//       p0: read B(blocks 0,1) A(blocks 0-3) | DMA A(t+2) pieces 0,1 -> A ring    | MFMA A0-3 x B0-1
//       p1: read B(blocks 2,3)               | DMA A(t+2) pieces 2,3              | MFMA A0-3 x B2-3
//       p2: read A(blocks 4..)               | DMA B(t+2) pieces 0,1 -> this B    | MFMA A4.. x B2-3
//       p3:                                  | DMA B(t+2) pieces 2,3; vmcnt(8): all of tile t+1 landed | MFMA A4.. x B0-1
//     RAW: tile t+1 is waited for (vmcnt) before the first barrier of p3 and first read in the next phase.  WAR: every phase
//     retires its LDS reads (lgkmcnt(0)) BEFORE its first barrier, so a region may be re-staged one phase after its last read
//     even by the wave row that runs half a phase ahead (B image: read p0/p1, staged p2/p3; A image t-1: read until p2 of the
//     previous tile, staged p0/p1).
//   * The epilogue goes through LDS (w8_epilogue_pass): whole 128-B lines per store instead of 16 x 32-byte fragments.
//   * Accumulation order per output element is k ascending, as in the 128x128 kernels: results are bit-identical to theirs.
#include "gemm_common.h"
#include "gemm_w8_epi.h"

using namespace sclg;

// in-kernel stamps of the wide kernels (diagnostic: flag SCL_GEMM_STAMPS; wave 0 of every block writes {realtime, shader clock}
// at kernel entry, after the prologue wait, after the K loop and after the epilogue)
__device__ unsigned long long scl_gemm_stamps[4096 * 8];

namespace {

__device__ __forceinline__ void w8_stamp(const GemmK& d, int slot, int lane, int wave) {
    if ((d.flags & SCL_GEMM_STAMPS) && wave == 0 && lane == 0 && blockIdx.x < 4096 && blockIdx.z == 0) {
        scl_gemm_stamps[blockIdx.x * 8 + 2 * slot] = __builtin_amdgcn_s_memrealtime();
        scl_gemm_stamps[blockIdx.x * 8 + 2 * slot + 1] = __builtin_amdgcn_s_memtime();
    }
}

constexpr int W8_BN = 256;
constexpr int W8_OPB = 2 * TILE_BYTES;      // one operand image of one K step: 256 rows x 64 k (or 2 sub-tiles of [64 k][128])
constexpr int W8_NA = 3, W8_NB = 2;         // A images form a ring of 3 (two K steps in flight), B images a ring of 2
constexpr int W8_LDS = (W8_NA + W8_NB) * W8_OPB;   // 163840 B = all of the CU's LDS

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
    template <int I0>
    __device__ __forceinline__ void issue2(char* img, int wave, unsigned soff, bool live) const {   // pieces I0, I0 + 1
#pragma unroll
        for (int i = I0; i < I0 + 2; ++i) {
            const unsigned off = live ? voff[i] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(img + (wave * 4 + i) * 1024), 16, off, soff, 0, 0);
        }
    }
    __device__ __forceinline__ void issue1(char* img, int wave, unsigned soff, bool live, int i) const {   // piece i
        const unsigned off = live ? voff[i] : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(img + (wave * 4 + i) * 1024), 16, off, soff, 0, 0);
    }
    __device__ __forceinline__ static unsigned kstep(const OpK& o) { return o.cin == 64 ? o.cout_bytes : 128u; }
    __device__ __forceinline__ void rows(const OpK&, int, int, int) {}
    __device__ __forceinline__ void set_batched(bool) {}
    __device__ __forceinline__ unsigned step(const OpK& o) const { return kstep(o); }
};
// transposed operand, 2 sub-tiles of [64 k rows][128 contiguous]: piece p -> sub-tile p >> 4, k rows 4*(p & 15) .. +3
struct W8T {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned voff[4];
    int col0_, collimit_;
    bool batched;      // K rows are utterance-batched (rpb < K: conv weight gradients): no scalar K advance, offsets per K step (rows())
    __device__ __forceinline__ void set_batched(bool b) { batched = b; }
    __device__ __forceinline__ void init(const OpK& o, const char* base, int col0, int collimit, int lane, int wave) {
        rsrc = make_rsrc(base);
        col0_ = col0; collimit_ = collimit;
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
    // batched rows: the offsets of the K step whose first row is krow (row_off divides by the rows per utterance); nothing otherwise
    __device__ __forceinline__ void rows(const OpK& o, int krow, int lane, int wave) {
        if (!batched) return;
        const int s16 = lane & 15;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = wave * 4 + i;
            const int kr = 4 * (p & 15) + (lane >> 4);
            const int sw = (kr & 3) | (((kr >> 3) & 1) << 2);
            const int col = col0_ + (p >> 4) * 128 + 8 * ((((s16 >> 1) ^ sw) << 1) | (s16 & 1));
            voff[i] = col < collimit_ ? row_off(o, (unsigned)(krow + kr)) + col_off(o, (unsigned)col) : OOB;
        }
    }
    template <int I0>
    __device__ __forceinline__ void issue2(char* img, int wave, unsigned soff, bool live) const {
#pragma unroll
        for (int i = I0; i < I0 + 2; ++i) {
            const unsigned off = live ? voff[i] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(img + (wave * 4 + i) * 1024), 16, off, soff, 0, 0);
        }
    }
    __device__ __forceinline__ void issue1(char* img, int wave, unsigned soff, bool live, int i) const {
        const unsigned off = live ? voff[i] : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(img + (wave * 4 + i) * 1024), 16, off, soff, 0, 0);
    }
    __device__ __forceinline__ static unsigned kstep(const OpK& o) { return 64u * o.ld_bytes; }
    __device__ __forceinline__ unsigned step(const OpK& o) const { return batched ? 0u : kstep(o); }
};
template <bool T> struct W8Sel { typedef W8K type; };
template <> struct W8Sel<true> { typedef W8T type; };

template <bool T>
__device__ __forceinline__ bf16x8 w8_frag(const char* img, int gb, int ks, int lane) {   // gb: 16-row (column) block 0..15 of the image
    if (T) return frag_t_raw(img + (gb >> 3) * TILE_BYTES, gb & 7, ks, lane);
    return frag_k(img, gb, ks, lane);
}

// one phase: [the caller's LDS reads / DMA issues]  wait reads -> barrier -> MFMA burst -> barrier.  The LDS reads are retired
// BEFORE the first barrier: a region may then be re-staged one phase after its last read (the B image at p2).
#define W8_MFMA_PHASE(MH, NH, FBSEL, NI)                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    __builtin_amdgcn_s_barrier();                                                                          \
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
__device__ __forceinline__ void w8_body(const GemmK& d, char* smem, typename W8Sel<AT>::type& la, typename W8Sel<BT>::type& lb,
                                        int nk, unsigned soffA, unsigned soffB, int ab, int m0, int n0, int mlimit,
                                        int z1, int z2, int ksplit, int lane, int wave, int wc, int krow0) {
    const unsigned stepA = la.step(d.A), stepB = lb.step(d.B);
    f32x4 acc[2][4][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
    const int nbk = wc * 4;
    char* const bimg = smem + W8_NA * W8_OPB;
    // A image of tile kt / kt + 1 / kt + 2 (= kt - 1, free since p2 of the previous tile)
    char *a_cur = smem, *a_nxt = smem + W8_OPB, *a_fill = smem + 2 * W8_OPB;
    // on entry soffA / soffB are the scalar offsets of tile 1 (the prologue staged tiles 0 and 1)
    for (int kt = 0; kt < nk; ++kt) {
        const char* tA = a_cur;
        char* tB = bimg + (kt & 1) * W8_OPB;
#ifdef W8_FAKE_AB      // timing experiment only (wrong results): the loop's pieces go out of range = no memory latency
        const bool live2 = false;
#else
        const bool live2 = kt + 2 < nk;
#endif
        soffA += stepA; soffB += stepB;          // tile kt + 2
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
        la.rows(d.A, krow0 + (kt + 2) * BK, lane, wave);      // utterance-batched K rows: this step's offsets (no-op otherwise)
        la.template issue2<0>(a_fill, wave, soffA, live2);
        W8_MFMA_PHASE(0, 0, fb0, 4)
        // ---- p1
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int j = 0; j < 2; ++j) fb1[j][ks] = w8_frag<BT>(tB, nbk + 2 + j, ks, lane);
        }
        la.template issue2<2>(a_fill, wave, soffA, live2);
        W8_MFMA_PHASE(0, 1, fb1, 4)
        // ---- p2   (every wave retired its B reads of this tile before the first barrier of its p1)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < NH; ++i) fa[i][ks] = w8_frag<AT>(tA, ab + 4 + i, ks, lane);
        }
        lb.rows(d.B, krow0 + (kt + 2) * BK, lane, wave);
        lb.template issue2<0>(tB, wave, soffB, live2);
        W8_MFMA_PHASE(1, 1, fb1, NH)
        // ---- p3
        lb.template issue2<2>(tB, wave, soffB, live2);      // (both B pairs at p2: 3-6 % slower on the weight-gradient shapes, same box)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // all but tile kt+2's 8 pieces: tile kt+1 has landed (this wave's pieces)
        W8_MFMA_PHASE(1, 0, fb0, NH)
        char* t = a_cur; a_cur = a_nxt; a_nxt = a_fill; a_fill = t;
    }
    if ((wave >> 2) == 0) __builtin_amdgcn_s_barrier();      // balance the stagger: every wave is past its last LDS read
    // the tail iterations issued out-of-range pieces (zeros) into the rings: none may land in another wave's epilogue block
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    w8_stamp(d, 2, lane, wave);
    {
        const long long cbase = z1 * d.c_bs1 + z2 * d.c_bs2 + (long long)ksplit * d.c_split_stride;
        const float* bias = (d.flags & SCL_GEMM_HAS_BIAS) ? d.bias + z2 * d.bias_bs2 : nullptr;
        char* wlds = smem + wave * 16384;
        char* wextra = smem + 8 * 16384 + wave * 4096;      // the 32 KiB above the eight transposition blocks: R staging
        // optional column sums of the stored values: partial row (tile row * 4 + wave row * 2 + pass)
        float* cs0 = d.colsum ? d.colsum + ((long long)(m0 / d.tile_m) * 4 + (wave >> 2) * 2) * d.N : nullptr;
        w8_epilogue_pass<4, 0>(d, acc[0], 4, wlds, wextra, m0 + ab * 16, n0 + wc * 64, mlimit, cbase, bias, lane, cs0);
        w8_epilogue_pass<4, 0>(d, acc[1], NH, wlds, wextra, m0 + (ab + 4) * 16, n0 + wc * 64, mlimit, cbase, bias, lane, cs0 ? cs0 + d.N : nullptr);
    }
    if (d.flags & SCL_GEMM_STAMPS) {      // only the diagnostic stamp needs the stores drained: a block retires with them in flight,
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // and the next block's prologue on this CU overlaps the drain
        w8_stamp(d, 3, lane, wave);
    }
}

template <bool AT, bool BT, int RB0, int RB1>
__global__ __launch_bounds__(512, 2) void scl_gemm_w8_kernel(const GemmK d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7
    const int wr = wave >> 2, wc = wave & 3;
    w8_stamp(d, 0, lane, wave);
    const int tiles_m = (d.M + d.tile_m - 1) / d.tile_m, tiles_n = (d.N + W8_BN - 1) / W8_BN;
    int tm, tn;
    tile_coords(blockIdx.x, gridDim.x, tiles_m, tiles_n, tm, tn, d.group_m);
    const int m0 = tm * d.tile_m, n0 = tn * W8_BN;
    const int mlimit = min(d.M, m0 + d.tile_m);
    int z = blockIdx.z;
    const int ksplit = __builtin_amdgcn_readfirstlane(z % d.splitk); z /= d.splitk;      // uniform, but integer division runs on the vector ALU: back to an SGPR
    const int z1 = __builtin_amdgcn_readfirstlane(z / d.nb2), z2 = z - z1 * d.nb2;
    const int nk_total = d.K / BK;                                  // K % 64 == 0 (checked on the host)
    const int nk_per = __builtin_amdgcn_readfirstlane((nk_total + d.splitk - 1) / d.splitk);
    const int kt0 = ksplit * nk_per;
    const int nk = max(0, min(nk_per, nk_total - kt0));
    const char* Ab = reinterpret_cast<const char*>(d.A.ptr) + z1 * d.A.bs1 + z2 * d.A.bs2;
    const char* Bb = reinterpret_cast<const char*>(d.B.ptr) + z1 * d.B.bs1 + z2 * d.B.bs2;

    typename W8Sel<AT>::type la;
    typename W8Sel<BT>::type lb;
    la.init(d.A, Ab, m0, mlimit, lane, wave);
    lb.init(d.B, Bb, n0, d.N, lane, wave);
    la.set_batched(AT && (d.debug & 4));      // conv weight gradients: K rows batched per utterance (host: scl_gemm_w8_launch)
    lb.set_batched(BT && (d.debug & 8));
    const unsigned stepA = la.step(d.A), stepB = lb.step(d.B);
    unsigned soffA = (unsigned)kt0 * stepA, soffB = (unsigned)kt0 * stepB;
    const int krow0 = kt0 * BK;

    // prologue: tiles 0 and 1 (16 pieces per wave)
    la.rows(d.A, krow0, lane, wave); lb.rows(d.B, krow0, lane, wave);
    la.template issue2<0>(smem, wave, soffA, nk > 0); la.template issue2<2>(smem, wave, soffA, nk > 0);
    lb.template issue2<0>(smem + W8_NA * W8_OPB, wave, soffB, nk > 0); lb.template issue2<2>(smem + W8_NA * W8_OPB, wave, soffB, nk > 0);
    soffA += stepA; soffB += stepB;
    la.rows(d.A, krow0 + BK, lane, wave); lb.rows(d.B, krow0 + BK, lane, wave);
    la.template issue2<0>(smem + W8_OPB, wave, soffA, nk > 1); la.template issue2<2>(smem + W8_OPB, wave, soffA, nk > 1);
    lb.template issue2<0>(smem + (W8_NA + 1) * W8_OPB, wave, soffB, nk > 1); lb.template issue2<2>(smem + (W8_NA + 1) * W8_OPB, wave, soffB, nk > 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    w8_stamp(d, 1, lane, wave);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();          // wave row 1 runs half a phase behind wave row 0

    if (RB0 == RB1) {
        w8_body<AT, BT, RB0 - 4>(d, smem, la, lb, nk, soffA, soffB, wr * RB0, m0, n0, mlimit, z1, z2, ksplit, lane, wave, wc, krow0);
    } else if (wr == 0) {
        w8_body<AT, BT, RB0 - 4>(d, smem, la, lb, nk, soffA, soffB, 0, m0, n0, mlimit, z1, z2, ksplit, lane, wave, wc, krow0);
    } else {
        w8_body<AT, BT, RB1 - 4>(d, smem, la, lb, nk, soffA, soffB, RB0, m0, n0, mlimit, z1, z2, ksplit, lane, wave, wc, krow0);
    }
}

// ---- single-barrier variant ("w8s") ---------------------------------------------------------------------------------------
// Ablation of the ping-pong loop above on MI355X (tools/abl.sh, profiles/r2_gemm_ablation.txt): with DMA, LDS reads and MFMAs
// all removed the 16 K steps of a block still took 11 us — 16 s_barrier per wave and K step cost ~0.7 us of a 1.44-us step, and
// the MFMA bursts do not overlap them.  This variant keeps the tile, the LDS images, the rings and the epilogue but has ONE
// barrier per K step: each wave double-buffers its fragments in registers per 32-deep sub-step, so the MFMAs of one sub-step
// run while the LDS reads of the next are in flight — also across the barrier, where the second sub-step of tile t multiplies
// while the first fragments of tile t+1 are read.  Ablation of THIS loop (16 K steps, 208 x 256 tile, profiles/r2_gemm_ablation.txt):
// all 22.9 us, MFMA only 17.6, LDS reads only 8.8, DMA only 11.1 (84 GB/s per CU), nothing 4.6 — the loop is MFMA-paced at the
// 1.8-2.0 GHz the chip holds under this load; running wave row 1 half a step out of phase (read | MFMA swapped) measured +-2 %.
// Per K step t (round 3; measurements of every step in profiles/r3_gemm_kloop_feed.txt):
//     4 x { 4 / 4 / 3 / 0 reads of set 1 <- tile t, k 32..63 | counted wait: B fragment j of set 0 (j = 0: and the A fragments) |
//           1/4 of MFMA set 0 | 1 of 4 DMA pieces A(t+2) -> A ring }
//     wait(set 1) | vmcnt(4): tile t+1 landed | BARRIER (tile t+1 visible to all; every wave has retired its reads of tile t)
//     4 x { 4 / 4 / 3 / 0 reads of set 0 <- tile t+1, k 0..31 | 1/4 of MFMA set 1 | 2 / 2 / 0 / 0 DMA pieces B(t+2) -> B image of tile t }
// A pieces have 1.5 K steps to land, B pieces one (the B image retires at this step's barrier and is needed at the next).  With the
// loop's pieces issued out of range (no memory access, -DW8S_FAKE_A / _B) a 4-round launch takes 95 instead of 112 us: the L2 -> LDS
// latency that 2 A tiles + 1 B tile in flight cannot cover is what separates this loop from its MFMA pace.
// Same K order per output element: bit-identical to the other kernels.
//
// Fragment reads of the single-barrier loop with compile-time indices.  Kernels with a transposed operand read it with inline-asm
// ds_read_b64_tr_b16 (frag_t_raw: the builtin would drain the LDS-DMA queue) which the compiler's lgkmcnt bookkeeping does not see; if
// their K-contiguous operand used plain loads, the compiler would count only those and wait far too early (it believes fewer
// operations are in flight than there are).  So in those kernels EVERY fragment read is inline asm and the loop's counted waits
// (lds_wait_upto) are the only ones; the kernel without transposed operands keeps plain loads and the compiler's own (exact) waits.
template <int OFF>
__device__ __forceinline__ bf16x8 lds_b128_raw(unsigned addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <bool AT, bool BT, int RBW>
struct W8SRead {
    static constexpr bool RAW = AT || BT;
    // LDS byte address of lane's 16 bytes of row block gb, 32-deep sub-step ks (frag_k's formula; + 2048 per further row block)
    static __device__ __forceinline__ unsigned kaddr(const char* img, int gb, int ks, int lane) {
        const int row = gb * 16 + (lane & 15);
        const int c = 4 * ks + (lane >> 4);
        return (unsigned)(uintptr_t)(lds_void*)(img + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
    }
    template <int R, int REND>
    static __device__ __forceinline__ void step(bf16x8 (&FA)[RBW], bf16x8 (&FB)[4], const char* TA, const char* TB, unsigned ka, unsigned kb,
                                                int ab, int nbk, int ks, int lane) {
        if constexpr (R < REND) {
            if constexpr (R == 0 || (R > RBW)) {
                constexpr int j = R == 0 ? 0 : R - RBW;
                if constexpr (BT) FB[j] = frag_t_raw(TB + ((nbk + j) >> 3) * TILE_BYTES, (nbk + j) & 7, ks, lane);
                else if constexpr (RAW) FB[j] = lds_b128_raw<j * 2048>(kb);
                else FB[j] = frag_k(TB, nbk + j, ks, lane);
            } else {
                constexpr int i = R - 1;
                if constexpr (AT) FA[i] = frag_t_raw(TA + ((ab + i) >> 3) * TILE_BYTES, (ab + i) & 7, ks, lane);
                else if constexpr (RAW) FA[i] = lds_b128_raw<i * 2048>(ka);
                else FA[i] = frag_k(TA, ab + i, ks, lane);
            }
            if (RAW) __builtin_amdgcn_sched_barrier(0);
            step<R + 1, REND>(FA, FB, TA, TB, ka, kb, ab, nbk, ks, lane);
        }
    }
    // fragments R0 .. R1-1 of the set of sub-step ks of images TA / TB.  ORDERED: also the plain loads keep this order (prologue only)
    template <int R0, int R1, bool ORDERED>
    static __device__ __forceinline__ void go(bf16x8 (&FA)[RBW], bf16x8 (&FB)[4], const char* TA, const char* TB, int ab, int nbk, int ks, int lane) {
        if constexpr (R0 < R1) {
            unsigned ka = 0, kb = 0;
            if constexpr (RAW && !AT) ka = kaddr(TA, ab, ks, lane);
            if constexpr (RAW && !BT) kb = kaddr(TB, nbk, ks, lane);
            if constexpr (ORDERED && !RAW) {
                step<R0, R0 + 1>(FA, FB, TA, TB, ka, kb, ab, nbk, ks, lane);
                __builtin_amdgcn_sched_barrier(0);
                go<R0 + 1, R1, true>(FA, FB, TA, TB, ab, nbk, ks, lane);
            } else {
                step<R0, R1>(FA, FB, TA, TB, ka, kb, ab, nbk, ks, lane);
            }
        }
    }
};

template <bool AT, bool BT, int RBW>
__device__ __forceinline__ void w8s_body(const GemmK& d, char* smem, const typename W8Sel<AT>::type& la, const typename W8Sel<BT>::type& lb,
                                         int nk, unsigned soffA, unsigned soffB, int ab, int m0, int n0, int mlimit,
                                         int z1, int z2, int ksplit, int lane, int wave, int wc) {
    const unsigned stepA = W8Sel<AT>::type::kstep(d.A), stepB = W8Sel<BT>::type::kstep(d.B);
    f32x4 acc[RBW][4];
#pragma unroll
    for (int i = 0; i < RBW; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fa0[RBW], fa1[RBW], fb0[4], fb1[4];
    const int nbk = wc * 4;
    char* const bimg = smem + W8_NA * W8_OPB;
    char *a_cur = smem, *a_nxt = smem + W8_OPB, *a_fill = smem + 2 * W8_OPB;
    // Fragment r of a sub-step's set, in the order the MFMA quarters need them: B fragment 0, the RBW A fragments, B fragments 1..3
    // (W8SRead).  A K-contiguous fragment is one ds_read_b128, a transposed one two ds_read_b64_tr_b16 (OPA / OPB LDS operations:
    // lgkmcnt counts operations).
    constexpr int NRD = RBW + 4, OPA = AT ? 2 : 1, OPB = BT ? 2 : 1;
    // reads issued before quarter 0 / 1 / 2 (cumulative), none before quarter 3: the last read of a set has a quarter of MFMAs to
    // complete before anything waits for it (3 / 3 / 3 / 2 measured equal)
    constexpr int RQ0 = NRD < 4 ? NRD : 4, RQ1 = NRD < 8 ? NRD : 8, RQ2 = NRD;      // (RBW = 3: seven fragments per set)
    typedef W8SRead<AT, BT, RBW> RD;
    // LDS operations of the first n fragments of a set
    auto ops_upto = [](int n) constexpr { return (n > 0 ? OPB : 0) + (n > 1 ? (n - 1 < RBW ? n - 1 : RBW) * OPA : 0) + (n > RBW + 1 ? (n - RBW - 1) * OPB : 0); };
    // tile 0 is visible (the caller's barrier); tiles 0 and 1 were staged, soffA / soffB point at tile 1
    lds_wait_upto(0);      // no scalar load may be pending when the loop starts: the compiler would answer with lgkmcnt(0) inside it
    __builtin_amdgcn_sched_barrier(0);
    RD::template go<0, NRD, true>(fa0, fb0, a_cur, bimg, ab, nbk, 0, lane);
    for (int kt = 0; kt < nk; ++kt) {
        char* tB = bimg + (kt & 1) * W8_OPB;
        const bool live2 = kt + 2 < nk;
#ifdef W8S_FAKE_A      // timing experiments only (results are wrong): the loop's pieces of one operand are out of range = no memory latency
        const bool live2a = false;
#else
        const bool live2a = live2;
#endif
#ifdef W8S_FAKE_B
        const bool live2b = false;
#else
        const bool live2b = live2;
#endif
        soffA += stepA; soffB += stepB;          // tile kt + 2
        // The sub-step's 28 (24) MFMAs go out in four quarters (one B fragment each); ahead of each quarter the wave issues part of the
        // OTHER sub-step's fragment reads, behind it one LDS-DMA piece.  Issued as clusters outside the burst (round 2) the 8 DMA
        // pieces and 22 reads of a K step cost both waves of a SIMD ~10 % of the step with the matrix pipe idle (same box, one call:
        // plain store 122 -> 116 us per 4-round launch, roofline.frac +0.012).  Accumulators are independent: the order of the MFMAs
        // does not change any sum.
        // Waits are COUNTED (LDS operations retire in order): quarter j of this half needs B fragment j of set 0 (quarter 0 also the
        // A fragments), i.e. everything but the 3 - j last B fragments of the set read during the previous half, and none of the
        // reads of set 1 issued since.
#define W8S_Q_FIRST(j, R0, R1)                                                                                               \
        RD::template go<R0, R1, false>(fa1, fb1, a_cur, tB, ab, nbk, 1, lane);                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        lds_wait_upto((3 - (j)) * OPB + ops_upto(R1));                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        _Pragma("unroll") for (int i = 0; i < RBW; ++i)                                                                      \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0[j], fa0[i], acc[i][j], 0, 0, 0);                         \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        la.issue1(a_fill, wave, soffA, live2a, j);                                                                           \
        __builtin_amdgcn_sched_barrier(0);
        W8S_Q_FIRST(0, 0, RQ0) W8S_Q_FIRST(1, RQ0, RQ1) W8S_Q_FIRST(2, RQ1, RQ2) W8S_Q_FIRST(3, RQ2, NRD)
#undef W8S_Q_FIRST
        lds_wait_upto(0);                                   // set 1 complete, and with it this wave's last read of tile kt
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");    // all but A(kt+2): tile kt+1 has landed (this wave's pieces)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const char* tBn = bimg + ((kt & 1) ^ 1) * W8_OPB;
        // (on the last K step these reads fetch whatever the next images hold and nothing uses them: cheaper than a branch per quarter)
        // B(kt+2) -> the B image the barrier just retired, two pieces behind each of quarters 0 and 1: it has to land by the NEXT barrier
        // (one K step; A pieces have 1.5) and a piece per quarter, the last one 0.6 step before that barrier, cost 4 % (same box, one
        // call: plain store 120.6 -> 116.2 us, roofline.frac 0.355 -> 0.366).  All four behind quarter 0, two of them ahead of it, or
        // A(kt+3) moved into quarters 2 and 3 of this half (two steps of lead): equal or slower.
#define W8S_B_ISSUE(j) if ((j) < 2) { lb.issue1(tB, wave, soffB, live2b, 2 * (j)); lb.issue1(tB, wave, soffB, live2b, 2 * (j) + 1); }
#define W8S_Q_SECOND(j, R0, R1)                                                                                              \
        RD::template go<R0, R1, false>(fa0, fb0, a_nxt, tBn, ab, nbk, 0, lane);                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        _Pragma("unroll") for (int i = 0; i < RBW; ++i)                                                                      \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1[j], fa1[i], acc[i][j], 0, 0, 0);                         \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        W8S_B_ISSUE(j)                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);
        W8S_Q_SECOND(0, 0, RQ0) W8S_Q_SECOND(1, RQ0, RQ1) W8S_Q_SECOND(2, RQ1, RQ2) W8S_Q_SECOND(3, RQ2, NRD)
#undef W8S_Q_SECOND
#undef W8S_B_ISSUE
        char* t = a_cur; a_cur = a_nxt; a_nxt = a_fill; a_fill = t;
    }
    // the tail iterations issued out-of-range pieces (zeros) into the rings: none may land in another wave's epilogue block
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    w8_stamp(d, 2, lane, wave);
    {
        const long long cbase = z1 * d.c_bs1 + z2 * d.c_bs2 + (long long)ksplit * d.c_split_stride;
        const float* bias = (d.flags & SCL_GEMM_HAS_BIAS) ? d.bias + z2 * d.bias_bs2 : nullptr;
        char* wlds = smem + wave * 16384;
        char* wextra = smem + 8 * 16384 + wave * 4096;      // the 32 KiB above the eight transposition blocks: R staging
        float* cs0 = d.colsum ? d.colsum + ((long long)(m0 / d.tile_m) * 4 + (wave >> 2) * 2) * d.N : nullptr;
        if constexpr (RBW >= 4) {
            f32x4 (&alo)[4][4] = *reinterpret_cast<f32x4 (*)[4][4]>(&acc[0]);
            w8_epilogue_pass<4, ((RBW < 8 && !AT) ? 0x1E : 0)>(d, alo, 4, wlds, wextra, m0 + ab * 16, n0 + wc * 64, mlimit, cbase, bias, lane, cs0);
        } else {      // the 112-row tile's second wave row: three row blocks
            f32x4 lo[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) lo[i][j] = (i < RBW) ? acc[(i < RBW) ? i : 0][j] : f32x4{0.f, 0.f, 0.f, 0.f};
            w8_epilogue_pass<4, ((RBW < 8 && !AT) ? 0x1E : 0)>(d, lo, RBW, wlds, wextra, m0 + ab * 16, n0 + wc * 64, mlimit, cbase, bias, lane, cs0);
        }
        if (RBW > 4) {
            f32x4 hi[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) hi[i][j] = (4 + i < RBW) ? acc[(4 + i < RBW) ? 4 + i : 0][j] : f32x4{0.f, 0.f, 0.f, 0.f};
            w8_epilogue_pass<4, ((RBW < 8 && !AT) ? 0x1E : 0)>(d, hi, RBW - 4, wlds, wextra, m0 + (ab + 4) * 16, n0 + wc * 64, mlimit, cbase, bias, lane, cs0 ? cs0 + d.N : nullptr);
        }
    }
    if (d.flags & SCL_GEMM_STAMPS) {      // only the diagnostic stamp needs the stores drained: a block retires with them in flight,
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // and the next block's prologue on this CU overlaps the drain
        w8_stamp(d, 3, lane, wave);
    }
}

// One output tile of the single-barrier loop: `bid` of `nblk` blocks in x (the tile id within ITS problem), `bz` the batch / split-K index.
// The plain kernel passes its own block indices; the grouped kernel (several problems in one launch) passes the position inside the member.
template <bool AT, bool BT, int RB0, int RB1, bool SEQ = false>
__device__ __forceinline__ void w8s_tile(const GemmK& d, char* smem, const int bid, const int nblk, const int bz) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7
    const int wr = wave >> 2, wc = wave & 3;
    w8_stamp(d, 0, lane, wave);
#ifdef SCL_EXPERIMENTS
    // experiment (SCL_W8_STAGGER, units of s_sleep 127 ~ 3.9 us): the first-round blocks of every other XCD start late, so that the
    // epilogues of the two halves of the chip do not hit HBM at the same moment for the rest of the launch
    if ((d.debug >> 8) && (bid & 1) && bid < 256 && bz == 0) {
        for (int i = 0; i < (d.debug >> 8); ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif
    const int tiles_m = (d.M + d.tile_m - 1) / d.tile_m, tiles_n = (d.N + W8_BN - 1) / W8_BN;
    int tm, tn;
    tile_coords<SEQ>(bid, nblk, tiles_m, tiles_n, tm, tn, d.group_m);
    const int m0 = tm * d.tile_m, n0 = tn * W8_BN;
    const int mlimit = min(d.M, m0 + d.tile_m);
    int z = bz;
    const int ksplit = __builtin_amdgcn_readfirstlane(z % d.splitk); z /= d.splitk;      // uniform, but integer division runs on the vector ALU: back to an SGPR
    const int z1 = __builtin_amdgcn_readfirstlane(z / d.nb2), z2 = z - z1 * d.nb2;
    const int nk_total = d.K / BK;
    const int nk_per = __builtin_amdgcn_readfirstlane((nk_total + d.splitk - 1) / d.splitk);
    const int kt0 = ksplit * nk_per;
    const int nk = max(0, min(nk_per, nk_total - kt0));
    const char* Ab = reinterpret_cast<const char*>(d.A.ptr) + z1 * d.A.bs1 + z2 * d.A.bs2;
    const char* Bb = reinterpret_cast<const char*>(d.B.ptr) + z1 * d.B.bs1 + z2 * d.B.bs2;

    typename W8Sel<AT>::type la;
    typename W8Sel<BT>::type lb;
    la.init(d.A, Ab, m0, mlimit, lane, wave);
    lb.init(d.B, Bb, n0, d.N, lane, wave);
    la.set_batched(false); lb.set_batched(false);
    const unsigned stepA = W8Sel<AT>::type::kstep(d.A), stepB = W8Sel<BT>::type::kstep(d.B);
    unsigned soffA = (unsigned)kt0 * stepA, soffB = (unsigned)kt0 * stepB;
    la.template issue2<0>(smem, wave, soffA, nk > 0); la.template issue2<2>(smem, wave, soffA, nk > 0);
    lb.template issue2<0>(smem + W8_NA * W8_OPB, wave, soffB, nk > 0); lb.template issue2<2>(smem + W8_NA * W8_OPB, wave, soffB, nk > 0);
    soffA += stepA; soffB += stepB;
    la.template issue2<0>(smem + W8_OPB, wave, soffA, nk > 1); la.template issue2<2>(smem + W8_OPB, wave, soffA, nk > 1);
    lb.template issue2<0>(smem + (W8_NA + 1) * W8_OPB, wave, soffB, nk > 1); lb.template issue2<2>(smem + (W8_NA + 1) * W8_OPB, wave, soffB, nk > 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    w8_stamp(d, 1, lane, wave);
    __builtin_amdgcn_s_barrier();
    if (RB0 == RB1) {
        w8s_body<AT, BT, RB0>(d, smem, la, lb, nk, soffA, soffB, wr * RB0, m0, n0, mlimit, z1, z2, ksplit, lane, wave, wc);
    } else if (wr == 0) {
        w8s_body<AT, BT, RB0>(d, smem, la, lb, nk, soffA, soffB, 0, m0, n0, mlimit, z1, z2, ksplit, lane, wave, wc);
    } else {
        w8s_body<AT, BT, RB1>(d, smem, la, lb, nk, soffA, soffB, RB0, m0, n0, mlimit, z1, z2, ksplit, lane, wave, wc);
    }
}

template <bool AT, bool BT, int RB0, int RB1>
__global__ __launch_bounds__(512, 2) void scl_gemm_w8s_kernel(const GemmK d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    w8s_tile<AT, BT, RB0, RB1>(d, smem, blockIdx.x, gridDim.x, blockIdx.z);
}

// ---- grouped launch: up to W8_GROUP_MAX independent problems, one block per tile, ONE kernel -----------------------------------------
// Why: the four weight gradients of a transformer layer (out-proj 16 tiles, QKV 48, fc1 64, fc2 64 of 256 x 256 at E = 1024, F = 4096) each
// fill the 256 CUs only through split-K: 4 - 16 slabs of f32 partial sums per gradient (written, read back and summed by a fifth launch:
// 1.5 ms and 6 GB of HBM traffic per step at batch 64).  Together they are 192 tiles: one launch, every block walks the WHOLE reduction
// (M = 12736 rows = 199 K steps instead of 12 - 50, so prologue, epilogue and the launch boundary are paid once per 199 steps) and stores
// the finished gradient tile straight into the flat gradient buffer — no slabs, no reduction pass, five launches become one.
// A block finds its problem from blockIdx.x and the members' first-tile table, and reads that member's descriptor from the
// kernel-argument segment with scalar loads (uniform index): nothing of the descriptor is copied, nothing goes to scratch.
struct GemmGroupK {
    GemmK k[W8_GROUP_MAX];
    int first[W8_GROUP_MAX + 1];      // launch-wide block index of each member's first block
    int tile0[W8_GROUP_MAX];          // the member's first tile inside ITS problem (a member may be a RANGE of a problem's tiles: carry-over)
    int ntiles[W8_GROUP_MAX];         // tiles of the whole problem (tile_coords needs the full count)
    int n;
    int xcd_major;                    // 1: blocks laid out XCD-major over the launch's whole tile sequence (default); 0: round 5's per-member runs
};
typedef const __attribute__((address_space(4))) GemmGroupK* W8GArg;
// Block -> tile (round 6).  Every block of this launch streams its two operand panels ([K rows] x 256 columns each: 6.5 MB at K = 12736)
// through its XCD's L2 once, and the 32 blocks of an XCD walk K in step, so what the XCD fetches over the fabric is the number of DISTINCT
// panels among its 32 tiles.  Rounds 5's mapping gave every member's tiles to the XCDs eight at a time (the per-member XCD run of
// tile_coords): 8 x 1 or 4 x 2 patches, ~28 distinct panels per XCD — 1.6 - 1.8 GB of FETCH_SIZE per launch (profiles/r6_gemm_classes.txt),
// 5.3 TB/s of fabric traffic under a 320-us launch: the launch was paced by that, not by its MFMAs (271 / 336 / 348 us at 128 / 192 / 256
// tiles, tools/group_fill_probe.py).  Now the launch's block sequence is XCD-major over ALL members: XCD x owns positions
// [x * total / 8, (x + 1) * total / 8) of the concatenated tile sequence, i.e. 32 consecutive tiles of (mostly) one problem in the grouped
// order — an 8 x 4 or 4 x 8 patch with 12 distinct panels.
template <bool AT, bool BT, int RB0, int RB1>
__global__ __launch_bounds__(512, 2) void scl_gemm_w8s_group_kernel(const GemmGroupK g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    W8GArg gp = (W8GArg)__builtin_amdgcn_kernarg_segment_ptr();
    const int total = g.first[W8_GROUP_MAX];
    const int x = blockIdx.x & 7, idx = blockIdx.x >> 3, q = total >> 3, r = total & 7;
#ifdef SCL_EXPERIMENTS      // the experiment build keeps round 5's per-member XCD runs selectable (SCL_WGRAD_XCD_MAJOR=0: profiles/r6_group_xcd_major.txt)
    const int pos = (g.xcd_major ? ((x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx) : (int)blockIdx.x);
#else
    const int pos = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
#endif
    int mem = 0;
#pragma unroll
    for (int i = 1; i < W8_GROUP_MAX; ++i) mem += (i < g.n && pos >= g.first[i]) ? 1 : 0;
    mem = __builtin_amdgcn_readfirstlane(mem);
    const int t = __builtin_amdgcn_readfirstlane(pos - gp->first[mem] + gp->tile0[mem]);
#ifdef SCL_EXPERIMENTS
    if (!g.xcd_major) { w8s_tile<AT, BT, RB0, RB1, false>(*(const GemmK*)&gp->k[mem], smem, t, gp->ntiles[mem], 0); return; }
#endif
    w8s_tile<AT, BT, RB0, RB1, true>(*(const GemmK*)&gp->k[mem], smem, t, gp->ntiles[mem], 0);      // (one instantiation in the shipped library: the tile body is 120 KB of code)
}

#ifdef SCL_EXPERIMENTS      // opt-in experiment, not part of the shipped library (see gemm.hip)
// ---- persistent single-barrier variant ("w8p") -----------------------------------------------------------------------------
// One block per CU stays resident and walks the tiles v = blockIdx.x, + gridDim.x, ... (gridDim.x a multiple of 8: v & 7, the XCD of
// tile_coords, is the block's own).  Why: with one 160-KiB block per CU nothing overlaps a block's retirement, the dispatch of the
// next one, its argument loads and the first LDS-DMA round trip — at K = 1024 a 208 x 256 tile spends 23 us in its 16 K steps and
// 35 us on the CU (fc1 forward, plain stores: 141 us for 4 rounds; the vendor library's persistent kernel: 117 us).  Here
//   * K step nk-2 of a tile stages the NEXT tile's K-stage 0 instead of out-of-range pieces (A into the ring slot that is free,
//     B into the image just consumed): it lands under the last two K steps;
//   * the epilogue runs in the 96 KiB that do not hold that stage: a wave's transposition block is two 4-KiB halves (two 16-row
//     blocks per pass) at ITS OWN piece offsets of the two free A slots, its R staging the same 4 KiB of the free B image — so
//     after its last pass a wave stages K-stage 1 of the next tile straight into those regions without a block-wide barrier;
//   * the next K loop starts on stage 0 (landed before the epilogue), its first wait covers stage 1 and the epilogue's stores.
// Same tile, same K order, same epilogue arithmetic and column-sum partial rows as w8s: bit-identical results.
// The descriptor is NOT kept in scalar registers across the K loops (w8s already spills some; here the epilogue's and the loaders'
// fields would be live through every K step of every tile): the hand-over step and the epilogue re-read it from the kernel-argument
// segment through a pointer the compiler cannot trace back (scalar loads, a few hundred cycles once per tile).
typedef const __attribute__((address_space(4))) GemmK* W8KArg;
__device__ __forceinline__ const GemmK* w8p_args(W8KArg p) {
    asm volatile("" : "+s"(p));
    return (const GemmK*)p;
}
template <bool AT, bool BT, int RBW>
__device__ __forceinline__ void w8p_body(const GemmK& d, W8KArg kp, char* smem, const char* Ab, const char* Bb, int ab, int tiles_m, int tiles_n,
                                         int lane, int wave, int wc) {
    typedef typename W8Sel<AT>::type LA;
    typedef typename W8Sel<BT>::type LB;
    const unsigned stepA = LA::kstep(d.A), stepB = LB::kstep(d.B);
    const int nk = d.K / BK;                                       // >= 3, no split-K (host)
    const int ntiles = tiles_m * tiles_n, G = gridDim.x;
    const int nbk = wc * 4;
    char* const bimg = smem + W8_NA * W8_OPB;
    char *a_cur = smem, *a_nxt = smem + W8_OPB, *a_fill = smem + 2 * W8_OPB;
    int bpar = 0;                                                  // B image that holds K-stage 0 of the current tile
    int v = blockIdx.x, tm, tn;
    tile_coords(v, ntiles, tiles_m, tiles_n, tm, tn, d.group_m);
    const int tile_m = d.tile_m, Mrows = d.M, group_m = d.group_m;
    int m0 = tm * tile_m, n0 = tn * W8_BN, mlimit = min(Mrows, m0 + tile_m);
    LA la;
    LB lb;
    la.init(d.A, Ab, m0, mlimit, lane, wave);
    lb.init(d.B, Bb, n0, d.N, lane, wave);
    la.set_batched(false); lb.set_batched(false);
    la.template issue2<0>(a_cur, wave, 0u, true); la.template issue2<2>(a_cur, wave, 0u, true);
    lb.template issue2<0>(bimg, wave, 0u, true); lb.template issue2<2>(bimg, wave, 0u, true);
    la.template issue2<0>(a_nxt, wave, stepA, true); la.template issue2<2>(a_nxt, wave, stepA, true);
    lb.template issue2<0>(bimg + W8_OPB, wave, stepB, true); lb.template issue2<2>(bimg + W8_OPB, wave, stepB, true);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#define W8S_READ(FA, FB, TA, TB, KS)                                                        \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) FB[j] = w8_frag<BT>(TB, nbk + j, KS, lane); \
    _Pragma("unroll") for (int i = 0; i < RBW; ++i) FA[i] = w8_frag<AT>(TA, ab + i, KS, lane);
#define W8S_MFMA(FA, FB)                                                                    \
    _Pragma("unroll") for (int i = 0; i < RBW; ++i)                                         \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                       \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FB[j], FA[i], acc[i][j], 0, 0, 0);
    for (;;) {
        f32x4 acc[RBW][4];
#pragma unroll
        for (int i = 0; i < RBW; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 fa0[RBW], fa1[RBW], fb0[4], fb1[4];
        const int vn = v + G;
        const bool has_next = vn < ntiles;
        int m0n = 0, n0n = 0, mlimn = 0;
        if (has_next) {
            tile_coords(vn, ntiles, tiles_m, tiles_n, tm, tn, group_m);
            m0n = tm * tile_m; n0n = tn * W8_BN; mlimn = min(Mrows, m0n + tile_m);
        }
        unsigned soffA = stepA, soffB = stepB;                     // K-stage 1 is in flight, stage 0 visible
        W8S_READ(fa0, fb0, a_cur, bimg + bpar * W8_OPB, 0)
        for (int kt = 0; kt < nk; ++kt) {
            char* tB = bimg + ((kt + bpar) & 1) * W8_OPB;
            const bool hand = kt + 2 == nk;                        // this step stages the next tile's K-stage 0
            const bool live2 = kt + 2 < nk || (hand && has_next);
            soffA += stepA; soffB += stepB;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            W8S_READ(fa1, fb1, a_cur, tB, 1)
            __builtin_amdgcn_sched_barrier(0);
            W8S_MFMA(fa0, fb0)
            __builtin_amdgcn_sched_barrier(0);
            if (hand && has_next) {      // the current tile's last A stage was requested at kt-1
                const GemmK* dh = w8p_args(kp);
                la.init(dh->A, reinterpret_cast<const char*>(dh->A.ptr), m0n, mlimn, lane, wave);
            }
            {
                const unsigned sa = hand ? 0u : soffA;
                la.template issue2<0>(a_fill, wave, sa, live2); la.template issue2<2>(a_fill, wave, sa, live2);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (hand && has_next) {
                const GemmK* dh = w8p_args(kp);
                lb.init(dh->B, reinterpret_cast<const char*>(dh->B.ptr), n0n, dh->N, lane, wave);
            }
            {
                const unsigned sb = hand ? 0u : soffB;
                lb.template issue2<0>(tB, wave, sb, live2); lb.template issue2<2>(tB, wave, sb, live2);
            }
            if (kt + 1 < nk) { W8S_READ(fa0, fb0, a_nxt, bimg + (((kt + bpar) & 1) ^ 1) * W8_OPB, 0) }
            __builtin_amdgcn_sched_barrier(0);
            W8S_MFMA(fa1, fb1)
            __builtin_amdgcn_sched_barrier(0);
            char* t = a_cur; a_cur = a_nxt; a_nxt = a_fill; a_fill = t;
        }
        // a_cur: stage 0 of the next tile (or zeros); a_nxt: the zero pieces of step nk-1; a_fill: K tile nk-1; B image
        // (nk-1+bpar)&1: zero pieces.  Everything has landed and every wave is past its last fragment read after this barrier.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        char* const bfree = bimg + ((nk - 1 + bpar) & 1) * W8_OPB;
        {
            const GemmK& d = *w8p_args(kp);      // shadows the kernel's by-value copy on purpose
            const float* bias = (d.flags & SCL_GEMM_HAS_BIAS) ? d.bias : nullptr;
            char* wlds = a_fill + wave * 4096;
            const int bstride = (int)(a_nxt - a_fill);
            char* wextra = bfree + wave * 4096;
            float* cs0 = d.colsum ? d.colsum + ((long long)(m0 / tile_m) * 4 + (wave >> 2) * 2) * d.N : nullptr;
            float cs[8];
#pragma unroll
            for (int pr = 0; pr < (RBW + 1) / 2; ++pr) {
                if (pr == 0 || pr == 2) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) cs[j] = 0.f;
                }
                if (2 * pr + 1 < RBW) {
                    f32x4 (&a2)[2][4] = *reinterpret_cast<f32x4 (*)[2][4]>(&acc[2 * pr]);
                    w8_epilogue_pass<2, 0>(d, a2, 2, wlds, wextra, m0 + (ab + 2 * pr) * 16, n0 + wc * 64, mlimit, 0ll, bias, lane, nullptr,
                                        cs0 ? cs : nullptr, bstride);
                } else {
                    f32x4 (&a1)[1][4] = *reinterpret_cast<f32x4 (*)[1][4]>(&acc[2 * pr]);
                    w8_epilogue_pass<1, 0>(d, a1, 1, wlds, wextra, m0 + (ab + 2 * pr) * 16, n0 + wc * 64, mlimit, 0ll, bias, lane, nullptr,
                                        cs0 ? cs : nullptr, bstride);
                }
                if (cs0 && (pr == 1 || pr == (RBW + 1) / 2 - 1)) w8_colsum_store(d, cs, cs0 + (pr >> 1) * d.N, n0 + wc * 64, lane);
            }
        }
        if (!has_next) break;
        // K-stage 1 of the next tile into this wave's own epilogue regions (its LDS reads have returned: their values were stored)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        la.template issue2<0>(a_nxt, wave, stepA, true); la.template issue2<2>(a_nxt, wave, stepA, true);
        lb.template issue2<0>(bfree, wave, stepB, true); lb.template issue2<2>(bfree, wave, stepB, true);
        bpar = (bpar + nk) & 1;
        v = vn; m0 = m0n; n0 = n0n; mlimit = mlimn;
    }
#undef W8S_READ
#undef W8S_MFMA
}

template <bool AT, bool BT, int RB0, int RB1>
__global__ __launch_bounds__(512, 2) void scl_gemm_w8p_kernel(const GemmK d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7
    const int wr = wave >> 2, wc = wave & 3;
    const int tiles_m = (d.M + d.tile_m - 1) / d.tile_m, tiles_n = (d.N + W8_BN - 1) / W8_BN;
    const char* Ab = reinterpret_cast<const char*>(d.A.ptr);
    const char* Bb = reinterpret_cast<const char*>(d.B.ptr);
    W8KArg kp = (W8KArg)__builtin_amdgcn_kernarg_segment_ptr();
    if (RB0 == RB1) {
        w8p_body<AT, BT, RB0>(d, kp, smem, Ab, Bb, wr * RB0, tiles_m, tiles_n, lane, wave, wc);
    } else if (wr == 0) {
        w8p_body<AT, BT, RB0>(d, kp, smem, Ab, Bb, 0, tiles_m, tiles_n, lane, wave, wc);
    } else {
        w8p_body<AT, BT, RB1>(d, kp, smem, Ab, Bb, RB0, tiles_m, tiles_n, lane, wave, wc);
    }
}

#endif  // SCL_EXPERIMENTS

template <int RB0, int RB1>
void w8_launch_rb(const GemmK& k, bool at, bool bt, dim3 grid, hipStream_t s, int mode) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)scl_gemm_w8_kernel<false, false, RB0, RB1>, hipFuncAttributeMaxDynamicSharedMemorySize, W8_LDS);
        (void)hipFuncSetAttribute((const void*)scl_gemm_w8_kernel<false, true, RB0, RB1>, hipFuncAttributeMaxDynamicSharedMemorySize, W8_LDS);
        (void)hipFuncSetAttribute((const void*)scl_gemm_w8_kernel<true, false, RB0, RB1>, hipFuncAttributeMaxDynamicSharedMemorySize, W8_LDS);
        (void)hipFuncSetAttribute((const void*)scl_gemm_w8_kernel<true, true, RB0, RB1>, hipFuncAttributeMaxDynamicSharedMemorySize, W8_LDS);
        (void)hipFuncSetAttribute((const void*)scl_gemm_w8s_kernel<false, false, RB0, RB1>, hipFuncAttributeMaxDynamicSharedMemorySize, W8_LDS);
        (void)hipFuncSetAttribute((const void*)scl_gemm_w8s_kernel<false, true, RB0, RB1>, hipFuncAttributeMaxDynamicSharedMemorySize, W8_LDS);
        (void)hipFuncSetAttribute((const void*)scl_gemm_w8s_kernel<true, false, RB0, RB1>, hipFuncAttributeMaxDynamicSharedMemorySize, W8_LDS);
        (void)hipFuncSetAttribute((const void*)scl_gemm_w8s_kernel<true, true, RB0, RB1>, hipFuncAttributeMaxDynamicSharedMemorySize, W8_LDS);
#ifdef SCL_EXPERIMENTS
        if constexpr (RB0 == 7) {
            (void)hipFuncSetAttribute((const void*)scl_gemm_w8p_kernel<false, false, RB0, RB1>, hipFuncAttributeMaxDynamicSharedMemorySize, W8_LDS);
            (void)hipFuncSetAttribute((const void*)scl_gemm_w8p_kernel<false, true, RB0, RB1>, hipFuncAttributeMaxDynamicSharedMemorySize, W8_LDS);
            (void)hipFuncSetAttribute((const void*)scl_gemm_w8p_kernel<true, false, RB0, RB1>, hipFuncAttributeMaxDynamicSharedMemorySize, W8_LDS);
        }
#endif
        attr_set = true;
    }
    const dim3 block(512);
#ifdef SCL_EXPERIMENTS
    if constexpr (RB0 == 7) {      // the persistent loop exists for the 208-row tile only (scl_gemm_w8_launch never asks for it with 256 rows)
        if (mode == 2) {      // grid = resident blocks (never both operands transposed: those launches take the ping-pong loop)
            if (!at && !bt) SCL_LAUNCH((scl_gemm_w8p_kernel<false, false, RB0, RB1>), grid, block, W8_LDS, s, k);
            else if (!at && bt) SCL_LAUNCH((scl_gemm_w8p_kernel<false, true, RB0, RB1>), grid, block, W8_LDS, s, k);
            else SCL_LAUNCH((scl_gemm_w8p_kernel<true, false, RB0, RB1>), grid, block, W8_LDS, s, k);
            return;
        }
    }
#endif
    if (mode == 1) {
        if (!at && !bt) SCL_LAUNCH((scl_gemm_w8s_kernel<false, false, RB0, RB1>), grid, block, W8_LDS, s, k);
        else if (!at && bt) SCL_LAUNCH((scl_gemm_w8s_kernel<false, true, RB0, RB1>), grid, block, W8_LDS, s, k);
        else if (at && !bt) SCL_LAUNCH((scl_gemm_w8s_kernel<true, false, RB0, RB1>), grid, block, W8_LDS, s, k);
        else SCL_LAUNCH((scl_gemm_w8s_kernel<true, true, RB0, RB1>), grid, block, W8_LDS, s, k);
        return;
    }
    if (!at && !bt) SCL_LAUNCH((scl_gemm_w8_kernel<false, false, RB0, RB1>), grid, block, W8_LDS, s, k);
    else if (!at && bt) SCL_LAUNCH((scl_gemm_w8_kernel<false, true, RB0, RB1>), grid, block, W8_LDS, s, k);
    else if (at && !bt) SCL_LAUNCH((scl_gemm_w8_kernel<true, false, RB0, RB1>), grid, block, W8_LDS, s, k);
    else SCL_LAUNCH((scl_gemm_w8_kernel<true, true, RB0, RB1>), grid, block, W8_LDS, s, k);
}

// 112-row tiles (4 + 3 row blocks; round 6): the single-barrier loop with K-contiguous A only (forward and data-gradient launches).
// Why: at M = 32 x 199 = 6368 rows an N = 1024 linear is 31 x 4 = 124 tiles of 208 rows — half the CUs — and ran on the 128 x 128 kernel
// instead (600 - 735 TFLOP/s); 57 x 4 = 228 tiles of 112 rows fill one round.  Same LDS images, rings and epilogue; rows 112.. of the A
// image are out-of-range pieces (no traffic).
void w8_launch_43(const GemmK& k, bool bt, dim3 grid, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)scl_gemm_w8s_kernel<false, false, 4, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, W8_LDS);
        (void)hipFuncSetAttribute((const void*)scl_gemm_w8s_kernel<false, true, 4, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, W8_LDS);
        attr_set = true;
    }
    if (!bt) SCL_LAUNCH((scl_gemm_w8s_kernel<false, false, 4, 3>), grid, dim3(512), W8_LDS, s, k);
    else SCL_LAUNCH((scl_gemm_w8s_kernel<false, true, 4, 3>), grid, dim3(512), W8_LDS, s, k);
}

}  // namespace

namespace sclg {

// Tile plan: the row-block split (7+6 = 208 rows or 8+8 = 256 rows) and the row pitch that need the fewest MFMA-rounds on
// `ncu` CUs; returns 0 when the wide kernel cannot address the operands (caller falls back to gemm.hip's kernels).
bool scl_gemm_w8_plan(const GemmK& k, bool at, bool bt, const SclGemmDesc& d, long long zdim, int ncu, W8Plan* plan) {
    auto flat_rows = [](const SclOperand& o, long long rows) { return (long long)o.rpb >= rows; };
    // both operands transposed (weight gradients, ping-pong loop): K rows may be utterance-batched (conv layers: rpb = frames per
    // utterance) — the loaders then compute their offsets per K step instead of advancing a scalar offset
    const bool tt = at && bt;
    const bool a_ok = at ? (flat_rows(d.A, d.K) || (tt && d.A.rpb >= 1)) : (d.A.cin == 0x7fffffff || d.A.cin == 64 || d.A.cin >= d.K);
    const bool b_ok = bt ? (flat_rows(d.B, d.K) || (tt && d.B.rpb >= 1)) : (d.B.cin == 0x7fffffff || d.B.cin == 64 || d.B.cin >= d.K);
    if (!a_ok || !b_ok || (d.K % BK) != 0) return false;
    const long long tiles_n = (d.N + W8_BN - 1) / W8_BN;
    long long best = -1;
    // variant 2 = 112-row tiles: K-contiguous A, one un-batched problem, no fused column sums (their partial rows assume two epilogue passes
    // per wave row), the single-barrier loop (SCL_W8_MODE=0 keeps the ping-pong, which has no 112-row form)
    // (read per call, like SCL_W8_MODE: the bit-identity tests run the same shapes on both tilings in one process)
    const char* e43 = getenv("SCL_W8_TILE112");
    const char* epp = getenv("SCL_W8_MODE");
    const bool v43_env = !e43 || atoi(e43) != 0, pingpong_env = epp && *epp && atoi(epp) == 0;
    const bool v43_ok = v43_env && !pingpong_env && !at && zdim == 1 && !d.colsum_part && !(d.flags & SCL_GEMM_STAMPS);
    for (int v = 0; v < (v43_ok ? 3 : 2); ++v) {
        // the K loop of either tile is bound by the per-CU L2 -> LDS feed (stamps + ablation, profiles/r2_gemm_*): a round costs
        // ~ (tile rows + 256 columns) x K bytes per CU, not the MFMA count
        const int bm = v == 0 ? 208 : (v == 1 ? 256 : 112);
        const long long ntm = (d.M + bm - 1) / bm;      // (round 5: 64 tiles of 199 rows instead of 62 of 206 at M = 12736 — a full last round, 3.4 % fewer rows per tile — measured 113.7 vs 112.5 us per launch: no gain, not kept)
        const long long rounds = (ntm * tiles_n * zdim + ncu - 1) / ncu;
        const long long cost = rounds * (bm + W8_BN);
        if (best < 0 || cost < best) {
            best = cost;
            plan->variant = v; plan->tiles_m = (int)ntm; plan->tile_m = (int)((d.M + ntm - 1) / ntm);
            plan->tiles = ntm * tiles_n; plan->cost = cost; plan->ncu = ncu;
        }
    }
    return true;
}

// Grouped launch of weight-gradient problems (both operands transposed, whole reduction per block, plain f32 store): see
// scl_gemm_w8s_group_kernel.  The caller (gemm.hip: scl_gemm_bf16_group) has checked every member with scl_gemm_w8_group_member_ok.
bool scl_gemm_w8_group_member_ok(const GemmK& k, bool at, bool bt, const SclGemmDesc& d) {
    if (!at || !bt || d.nb1 != 1 || d.nb2 != 1 || d.splitk != 1 || (d.K % BK) != 0 || d.K / BK < 3) return false;
    if ((long long)d.A.rpb < (long long)d.K || (long long)d.B.rpb < (long long)d.K) return false;      // flat K rows only (no utterance batching)
    if ((d.M % 8) || (d.N % 8) || !k.vec_ok) return false;
    const int plain = SCL_GEMM_A_T | SCL_GEMM_B_T | SCL_GEMM_C_F32;
    return (d.flags & ~(SCL_GEMM_NO_DMA | SCL_GEMM_NO_W8)) == plain && !d.colsum_part;
}

int scl_gemm_w8_group_tiles(const GemmK& k) { return ((k.M + 255) / 256) * ((k.N + W8_BN - 1) / W8_BN); }

// tile0 / ntile (optional): member i covers tiles [tile0[i], tile0[i] + ntile[i]) of its problem — the caller carries the rest over into a
// later launch, so that every launch is a full round of the 256 CUs (the encoder's backward: 3 launches of 256 tiles per 4 layers instead of
// 4 of 192; the launch time barely depends on the tile count: 336 us at 192 tiles, 348 at 256, tools/group_fill_probe.py).
int scl_gemm_w8_group_launch(GemmK* ks, int n, const int* tile0, const int* ntile, hipStream_t s) {
    // 256-row tiles.  Measured against 240 tiles of 205 rows for a layer's four weight gradients (94 % instead of 75 % of the CUs busy, the
    // choice scl_gemm_w8_plan's cost model would make): 365 vs 275 us per launch, the step 45.1 vs 42.5 ms (round 5, three interleaved
    // pairs): the smaller tile moves more operand bytes per MFMA through the L2 -> LDS path that paces the loop.
    GemmGroupK g = GemmGroupK();
    int total = 0;
    for (int i = 0; i < n; ++i) {
        GemmK& k = ks[i];
        const int ntm = (k.M + 255) / 256;
        k.tile_m = (k.M + ntm - 1) / ntm;
        k.debug = 0;
        g.k[i] = k;
        g.first[i] = total;
        g.ntiles[i] = scl_gemm_w8_group_tiles(k);
        g.tile0[i] = tile0 ? tile0[i] : 0;
        total += ntile ? ntile[i] : g.ntiles[i];
    }
    for (int i = n; i <= W8_GROUP_MAX; ++i) g.first[i] = total;
    g.n = n;
#ifdef SCL_EXPERIMENTS
    static const int xcd_major = [] { const char* e = getenv("SCL_WGRAD_XCD_MAJOR"); return e ? atoi(e) : 1; }();      // 0: round 5's per-member XCD runs (A/B)
    g.xcd_major = xcd_major;
#else
    g.xcd_major = 1;
#endif
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)scl_gemm_w8s_group_kernel<true, true, 8, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, W8_LDS);
        attr_set = true;
    }
    SCL_LAUNCH((scl_gemm_w8s_group_kernel<true, true, 8, 8>), dim3((unsigned)total), dim3(512), W8_LDS, s, g);
    return 0;
}

static long long w8p_launches = 0;
long long scl_gemm_w8p_launches() { return w8p_launches; }

int scl_gemm_read_stamps(unsigned long long* out, int nblocks) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(scl_gemm_stamps), sizeof(unsigned long long) * 8 * (size_t)nblocks) == hipSuccess ? 0 : -1;
}

int scl_gemm_w8_launch(GemmK& k, bool at, bool bt, const W8Plan& plan, long long zdim, hipStream_t s) {
    k.tile_m = plan.tile_m;
    k.debug = 0;
    {   // the epilogue stages R through buffer loads with 32-bit byte offsets: fall back to plain loads for an R extent >= 4 GiB
        const long long nb1 = zdim / ((long long)k.nb2 * k.splitk);
        const long long rows = k.M - 1, rpb = (long long)k.c_rpb;
        const long long maxoff = (nb1 - 1) * k.c_bs1 + ((long long)k.nb2 - 1) * k.c_bs2 + (rows / rpb) * k.c_rbstride + (rows % rpb) * k.ldc + k.N;
        if (maxoff * ((k.flags & SCL_GEMM_R_F32) ? 4 : 2) >= 0xFFFFFF00ll) k.debug |= 2;
    }
    const dim3 grid((unsigned)plan.tiles, 1, (unsigned)zdim);
    // 1: single barrier per K step (default), 0: two-barrier ping-pong.  Round 2 measured the ping-pong 2-9 % ahead when both operands
    // are transposed (weight gradients: twice the LDS read instructions per fragment); with the single-barrier loop's MFMA quarters,
    // counted waits and early B pieces (round 3) it is the other way round — one call, tools/gemm_bench "wgrad sk4", ping-pong ->
    // single barrier: qkv 110.4 -> 109.0, out 81.2 -> 72.2, fc1 111.9 -> 109.0, fc2 109.1 -> 102.5 us; whole step 49.2 -> 47.1 ms.
    // The ping-pong loop stays for utterance-batched K rows (below) and as SCL_W8_MODE=0.
    const char* me = getenv("SCL_W8_MODE");
    int mode = (me && *me) ? atoi(me) : 1;
    if (at && bt) {      // utterance-batched K rows (see scl_gemm_w8_plan): only the ping-pong loop computes per-step offsets
        if ((long long)k.A.rpb < (long long)k.K) { k.debug |= 4; mode = 0; }
        if ((long long)k.B.rpb < (long long)k.K) { k.debug |= 8; mode = 0; }
    }
    {
        static const bool epi_generic = [] { const char* e = getenv("SCL_W8_EPI_GENERIC"); return e && atoi(e) != 0; }();
        if (epi_generic) k.debug |= 16;      // A/B: the generic (run-time flag) epilogue loop for every launch
    }
    dim3 g = grid;
#ifdef SCL_EXPERIMENTS
    {
        const char* sg = getenv("SCL_W8_STAGGER");
        const int stg = sg ? atoi(sg) : 0;
        if (stg > 0 && plan.tiles > 256 && zdim == 1) k.debug |= (stg & 0xFF) << 8;
    }
    // persistent blocks (w8p): SCL_GEMM_PERSIST = 0 (default) never, 1 when the launch has more than one round of tiles, 8 .. 256 =
    // that many resident blocks whenever the kernel is legal (tests: several tiles per block on small problems).  Opt-in: measured
    // on MI355X against one-tile blocks (tools/persist_probe.py, profiles/r3_gemm_persistent_vs_one_tile.txt) it is equal within
    // +-2 % on the encoder's 3- and 4-round launches — the first K step of the next tile still waits (in-order vmcnt) for the
    // epilogue's stores, whose drain into HBM, with every CU storing at once, is what a block switch already overlapped — and the
    // 256-row variant spills.  In the full train step SCL_GEMM_PERSIST=1 is much SLOWER (84.8 vs 47.4 ms: +0.5 ms per persistent
    // launch): the w8p kernels are the only ones in the step with a private segment (48-96 B of scratch from their register
    // pressure), and a kernel that needs scratch between kernels that do not costs a queue-side scratch set-up per dispatch.  Kept
    // as a tested, documented experiment; do not switch it on for training.
    const char* pe = getenv("SCL_GEMM_PERSIST");
    const int pv = pe ? atoi(pe) : 0;
    if (mode == 1 && pv > 0 && !(at && bt) && zdim == 1 && k.splitk == 1 && k.K / BK >= 3 && !(k.flags & SCL_GEMM_STAMPS)) {
        const long long ncu = plan.ncu >= 8 ? plan.ncu : 256;
        const long long rounds = (plan.tiles + ncu - 1) / ncu;
        long long G = pv >= 8 ? (pv & ~7) : ((((plan.tiles + rounds - 1) / rounds) + 7) & ~7ll);      // equal rounds on every block, a multiple of 8
        if (G > (ncu & ~7ll) && pv < 8) G = ncu & ~7ll;
        // the 208-row variant only — the 256-row one spilled under the persistent loop's register pressure (2 x slower) and is not built
        if (plan.variant == 0 && (pv >= 8 || rounds >= 2) && G >= 8 && G < plan.tiles) { mode = 2; g = dim3((unsigned)G, 1, 1); ++w8p_launches; }
    }
#endif
    if (plan.variant == 2) w8_launch_43(k, bt, g, s);
    else if (plan.variant == 0) w8_launch_rb<7, 6>(k, at, bt, g, s, mode);
    else w8_launch_rb<8, 8>(k, at, bt, g, s, mode);
    return 0;
}

}  // namespace sclg
