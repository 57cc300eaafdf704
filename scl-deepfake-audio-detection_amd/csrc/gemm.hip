// gemm.hip — the generic batched bf16 contraction of the hot path, on CDNA4 matrix cores.
//
//   C[z][m][n] = epilogue( alpha * sum_k A[z][m][k] * B[z][n][k] )
//
// Every GEMM-shaped piece of the reference's forward and backward goes through this one kernel
// family: the wav2vec2 conv layers 1..6 (im2col is free in channels-last layout: ld = stride*C),
// the grouped positional conv (2-level contiguous index), post_extract_proj / q,k,v,out_proj /
// fc1 / fc2 (fairseq Wav2Vec2Model.forward reached from model/xlsr.py:41), LL and the BackEnd
// linears (model/wav2vec2_linear_nll.py:107,49-67), the attention QK^T / PV products, and all of
// their dgrad / wgrad contractions (autograd backward triggered at main.py:79).
//
// Design (gfx950):
//   * 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 4x4 MFMA
//     tiles of v_mfma_f32_16x16x32_bf16), BK = 64 per barrier, fp32 accumulation.
//   * LDS double buffer (2 x 32 KiB): global -> registers -> LDS, the loads of tile k+1 in
//     flight while tile k is multiplied; one barrier per K step.
//   * Loads are `buffer_load_dwordx4` through a raw buffer descriptor with 32-bit byte offsets:
//     rows / reduction indices outside the problem get the offset 0xFFFFFFFF, which the hardware
//     range check turns into zeros — the main loop has no divergent branch.  Row offsets of the
//     utterance-batched operands ((r / rpb) * rbstride + (r % rpb) * ld) use a multiply-high
//     "magic" division prepared on the host.
//   * Operand layouts: K-contiguous operands are staged as [rows][64 k] with a 16-byte-slot XOR
//     swizzle (slot ^= (row>>1)&7) so ds_read_b128 fragment reads are bank-conflict free;
//     transposed operands ([K rows][128 contiguous]) are staged as they lie in memory (coalesced
//     16-byte loads, 32-byte-chunk XOR swizzle) and delivered to the MFMA through
//     ds_read_b64_tr_b16 — no transposed copies of weights or activations exist anywhere.
//   * The MFMA is issued with the operands swapped (D = B-frag x A-frag), so each lane ends up
//     with 4 CONSECUTIVE output columns of one row: the epilogue (alpha, bias, activation +
//     pre-activation second output, residual add or activation-gradient multiply, dropout mask)
//     is vectorised 4-wide and stores 8 B (bf16) / 16 B (f32) per lane.
//   * Workgroup ids are remapped so that each XCD (private L2) owns a contiguous run of tiles.
#include "gemm_common.h"

using namespace sclg;

namespace {

constexpr int BM = 128, BN = 128;

// ---- staging of a K-contiguous operand tile: [128 rows][64 k] -------------------------------
struct StageK {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned rowoff[4];   // OOB for rows outside the problem
    unsigned lds_off[4];
    int kcur, kend;

    __device__ __forceinline__ void init(const OpK& o, const char* base, int row0, int rowlimit, int kbegin, int kend_, int tid) {
        rsrc = make_rsrc(base);
        const int c = tid & 7;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (tid >> 3) + 32 * i;
            const int r = row0 + row;
            rowoff[i] = r < rowlimit ? row_off(o, (unsigned)r) : OOB;
            lds_off[i] = (unsigned)(row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
        }
        kcur = kbegin + 8 * c; kend = kend_;
    }
    __device__ __forceinline__ void load(const OpK& o, u32x4 (&r)[4]) const {
        const unsigned koff = col_off(o, (unsigned)kcur);
        const bool kok = kcur < kend;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned off = (kok && rowoff[i] != OOB) ? rowoff[i] + koff : OOB;
            r[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        }
    }
    __device__ __forceinline__ void fix_tail(u32x4 (&r)[4]) const {   // only called on a partial last K step
        const int nvalid = kend - kcur;
        if (nvalid < 8) {
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = mask_tail(r[i], nvalid);
        }
    }
    __device__ __forceinline__ void advance() { kcur += BK; }
};

// ---- staging of a transposed operand tile: [64 k rows][128 contiguous] ----------------------
struct StageT {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned coloff;      // OOB when this thread's 8 columns are all outside the problem
    unsigned lds_off[4];
    int rcur, kend, ncolvalid;

    __device__ __forceinline__ void init(const OpK& o, const char* base, int col0, int collimit, int kbegin, int kend_, int tid) {
        rsrc = make_rsrc(base);
        const int c16 = tid & 15;
        const int col = col0 + 8 * c16;
        int nv = collimit - col; nv = nv < 0 ? 0 : (nv > 8 ? 8 : nv);
        ncolvalid = nv;
        coloff = nv > 0 ? col_off(o, (unsigned)col) : OOB;
        rcur = kbegin + (tid >> 4); kend = kend_;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int krow = (tid >> 4) + 16 * i;
            const int sw = (krow & 3) | (((krow >> 3) & 1) << 2);
            lds_off[i] = (unsigned)(krow * 256 + (((c16 >> 1) ^ sw) << 5) + ((c16 & 1) << 4));
        }
    }
    __device__ __forceinline__ void load(const OpK& o, u32x4 (&r)[4]) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rr = rcur + 16 * i;
            const unsigned off = (rr < kend && coloff != OOB) ? row_off(o, (unsigned)rr) + coloff : OOB;
            r[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        }
    }
    __device__ __forceinline__ void fix_tail(u32x4 (&r)[4]) const {   // only called on column-edge tiles
        if (ncolvalid < 8) {
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = mask_tail(r[i], ncolvalid);
        }
    }
    __device__ __forceinline__ void advance() { rcur += BK; }
};

template <bool T> struct StageSel { typedef StageK type; };
template <> struct StageSel<true> { typedef StageT type; };



template <bool AT, bool BT>
__global__ __launch_bounds__(256, 2) void scl_gemm_kernel(const GemmK d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    const int tiles_n = (d.N + BN - 1) / BN;
    int tm, tn;
    tile_coords(blockIdx.x, gridDim.x, (d.M + BM - 1) / BM, tiles_n, tm, tn, d.group_m);
    const int m0 = tm * BM, n0 = tn * BN;

    int z = blockIdx.z;
    const int ksplit = __builtin_amdgcn_readfirstlane(z % d.splitk); z /= d.splitk;      // uniform, but integer division runs on the vector ALU: back to an SGPR
    const int z1 = __builtin_amdgcn_readfirstlane(z / d.nb2), z2 = z - z1 * d.nb2;

    const int nk_total = (d.K + BK - 1) / BK;
    const int nk_per = __builtin_amdgcn_readfirstlane((nk_total + d.splitk - 1) / d.splitk);
    const int kbegin = ksplit * nk_per * BK;
    int kend = kbegin + nk_per * BK; if (kend > d.K) kend = d.K;
    const int nk = kend > kbegin ? (kend - kbegin + BK - 1) / BK : 0;

    const char* Ab = reinterpret_cast<const char*>(d.A.ptr) + z1 * d.A.bs1 + z2 * d.A.bs2;
    const char* Bb = reinterpret_cast<const char*>(d.B.ptr) + z1 * d.B.bs1 + z2 * d.B.bs2;

    typename StageSel<AT>::type sa;
    typename StageSel<BT>::type sb;
    sa.init(d.A, Ab, m0, d.M, kbegin, kend, tid);
    sb.init(d.B, Bb, n0, d.N, kbegin, kend, tid);
    // block-uniform conditions under which loaded vectors can be partially valid
    const bool a_edge = AT ? (m0 + BM > d.M && (d.M & 7)) : false;
    const bool b_edge = BT ? (n0 + BN > d.N && (d.N & 7)) : false;
    const bool k_tail = (kend & 7) != 0;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 ra[4], rb[4];
    if (nk > 0) {
        sa.load(d.A, ra); sb.load(d.B, rb);
        if (AT ? a_edge : (k_tail && nk == 1)) sa.fix_tail(ra);
        if (BT ? b_edge : (k_tail && nk == 1)) sb.fix_tail(rb);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<u32x4*>(smem + sa.lds_off[i]) = ra[i];
            *reinterpret_cast<u32x4*>(smem + TILE_BYTES + sb.lds_off[i]) = rb[i];
        }
    }
    __syncthreads();

    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = (kt + 1) < nk;
        if (more) {
            sa.advance(); sb.advance();
            sa.load(d.A, ra); sb.load(d.B, rb);
        }
        const char* tA = smem + cur * (2 * TILE_BYTES);
        const char* tB = tA + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = AT ? frag_t(tA, wr * 4 + i, ks, lane) : frag_k(tA, wr * 4 + i, ks, lane);
                fb[i] = BT ? frag_t(tB, wc * 4 + i, ks, lane) : frag_k(tB, wc * 4 + i, ks, lane);
            }
            // swapped operands: D[i = n within tile][j = m within tile]
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        if (more) {
            const bool last = (kt + 2) == nk;
            if (AT ? a_edge : (k_tail && last)) sa.fix_tail(ra);
            if (BT ? b_edge : (k_tail && last)) sb.fix_tail(rb);
            char* nA = smem + (cur ^ 1) * (2 * TILE_BYTES);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<u32x4*>(nA + sa.lds_off[i]) = ra[i];
                *reinterpret_cast<u32x4*>(nA + TILE_BYTES + sb.lds_off[i]) = rb[i];
            }
        }
        __syncthreads();
        cur ^= 1;
    }

    gemm_epilogue(d, acc, m0, n0, z1, z2, ksplit, lane, wr, wc);
}


// ---- LDS-DMA variant ---------------------------------------------------------------------------
// Same tile, same LDS images, same fragment reads and epilogue; the staging goes global -> LDS
// directly (`buffer_load_dwordx4 ... lds`, 1 KiB per wave-instruction, LDS address = uniform base +
// lane*16) so no staging VGPRs and no ds_write pass exist.  Because the DMA writes lane-linearly, the
// XOR swizzles are applied to the per-lane SOURCE address: lane l of piece p fetches the logical
// chunk that belongs at physical position (p, l).  Out-of-range rows / reduction indices fetch at
// offset 0xFFFFFFFF (hardware range check -> zeros land in LDS).  Used whenever no 16-byte vector can
// be partially valid (K % 8 == 0 for K-contiguous operands, M / N % 8 == 0 for transposed ones).

struct DmaK {   // K-contiguous operand: piece p = rows 8p..8p+7; this wave owns pieces wave*4 + i
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned rowoff[4];
    int kc[4];            // logical k-chunk (x8 elements) this lane fetches for piece i
    int kcur, kend;
    __device__ __forceinline__ void init(const OpK& o, const char* base, int row0, int rowlimit, int kbegin, int kend_, int lane, int wave) {
        rsrc = make_rsrc(base);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = wave * 4 + i;
            const int row = 8 * p + (lane >> 3);
            const int r = row0 + row;
            rowoff[i] = r < rowlimit ? row_off(o, (unsigned)r) : OOB;
            kc[i] = (lane & 7) ^ ((row >> 1) & 7);
        }
        kcur = kbegin; kend = kend_;
    }
    __device__ __forceinline__ void issue(const OpK& o, char* tile, int wave) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = kcur + 8 * kc[i];
            const unsigned off = (k < kend && rowoff[i] != OOB) ? rowoff[i] + col_off(o, (unsigned)k) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(tile + (wave * 4 + i) * 1024), 16, off, 0, 0, 0);
        }
    }
    __device__ __forceinline__ void advance() { kcur += BK; }
};

struct DmaT {   // transposed operand: piece p = k-rows 4p..4p+3 (256 B each)
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned coloff[4];
    int krow[4];
    int kcur, kend;
    __device__ __forceinline__ void init(const OpK& o, const char* base, int col0, int collimit, int kbegin, int kend_, int lane, int wave) {
        rsrc = make_rsrc(base);
        const int s16 = lane & 15;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = wave * 4 + i;
            const int kr = 4 * p + (lane >> 4);
            const int sw = (kr & 3) | (((kr >> 3) & 1) << 2);
            const int col = col0 + 8 * ((((s16 >> 1) ^ sw) << 1) | (s16 & 1));
            krow[i] = kr;
            coloff[i] = col < collimit ? col_off(o, (unsigned)col) : OOB;
        }
        kcur = kbegin; kend = kend_;
    }
    __device__ __forceinline__ void issue(const OpK& o, char* tile, int wave) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rr = kcur + krow[i];
            const unsigned off = (rr < kend && coloff[i] != OOB) ? row_off(o, (unsigned)rr) + coloff[i] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(tile + (wave * 4 + i) * 1024), 16, off, 0, 0, 0);
        }
    }
    __device__ __forceinline__ void advance() { kcur += BK; }
};

template <bool T> struct DmaSel { typedef DmaK type; };
template <> struct DmaSel<true> { typedef DmaT type; };

template <bool AT, bool BT>
__global__ __launch_bounds__(256, 2) void scl_gemm_dma_kernel(const GemmK d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_n = (d.N + BN - 1) / BN;
    int tm, tn;
    tile_coords(blockIdx.x, gridDim.x, (d.M + BM - 1) / BM, tiles_n, tm, tn, d.group_m);
    const int m0 = tm * BM, n0 = tn * BN;
    int z = blockIdx.z;
    const int ksplit = __builtin_amdgcn_readfirstlane(z % d.splitk); z /= d.splitk;      // uniform, but integer division runs on the vector ALU: back to an SGPR
    const int z1 = __builtin_amdgcn_readfirstlane(z / d.nb2), z2 = z - z1 * d.nb2;
    const int nk_total = (d.K + BK - 1) / BK;
    const int nk_per = __builtin_amdgcn_readfirstlane((nk_total + d.splitk - 1) / d.splitk);
    const int kbegin = ksplit * nk_per * BK;
    int kend = kbegin + nk_per * BK; if (kend > d.K) kend = d.K;
    const int nk = kend > kbegin ? (kend - kbegin + BK - 1) / BK : 0;
    const char* Ab = reinterpret_cast<const char*>(d.A.ptr) + z1 * d.A.bs1 + z2 * d.A.bs2;
    const char* Bb = reinterpret_cast<const char*>(d.B.ptr) + z1 * d.B.bs1 + z2 * d.B.bs2;

    typename DmaSel<AT>::type sa;
    typename DmaSel<BT>::type sb;
    sa.init(d.A, Ab, m0, d.M, kbegin, kend, lane, wave);
    sb.init(d.B, Bb, n0, d.N, kbegin, kend, lane, wave);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (nk > 0) { sa.issue(d.A, smem, wave); sb.issue(d.B, smem + TILE_BYTES, wave); }
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile kt have landed
        __syncthreads();                                    // everyone's have; everyone is done reading the other buffer
        if (kt + 1 < nk) {
            sa.advance(); sb.advance();
            char* nb = smem + (cur ^ 1) * (2 * TILE_BYTES);
            sa.issue(d.A, nb, wave); sb.issue(d.B, nb + TILE_BYTES, wave);
        }
        const char* tA = smem + cur * (2 * TILE_BYTES);
        const char* tB = tA + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = AT ? frag_t_raw(tA, wr * 4 + i, ks, lane) : frag_k(tA, wr * 4 + i, ks, lane);
                fb[i] = BT ? frag_t_raw(tB, wc * 4 + i, ks, lane) : frag_k(tB, wc * 4 + i, ks, lane);
            }
            if (AT || BT) lds_wait_frags(fa, fb);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
    }
    gemm_epilogue(d, acc, m0, n0, z1, z2, ksplit, lane, wr, wc);
}


#ifdef SCL_EXPERIMENTS
// ---- deep-ring variant of the LDS-DMA kernel: the same tile, images, fragment reads and epilogue behind a ring of S stages ------------
// Round 6 (profiles/r6_pack11_gemm_classes.txt): at M = 11 x 199 rows an N = 1024 linear is 144 tiles — at most one 4-wave workgroup per
// CU on 144 of the 256 CUs — and the two-stage loop above then runs at the LATENCY of one stage per K step (~1.5 us: 21 - 29 us for a
// 4.6-GFLOP product, 160 - 220 TFLOP/s).  When the whole grid fits one workgroup per CU nothing else wants the CU's LDS, so this variant
// spends it on bytes in flight: S - 1 stages of 32 KiB outstanding behind a counted vmcnt and raw barriers (a __syncthreads() would drain
// the queue).  Loads past the last K tile are issued out of range (no fetch) so that the count is the same in every iteration.
// Accumulation order is k ascending as above: results are bit-identical to the two-stage kernel's.
// MEASURED (profiles/r6_small_m_probe.txt, rings of 3 / 5 stages, kernel time from the dispatch stamps): no gain — 17.6 vs 17.9 us at
// K = 1024, 48.6 vs 52.8 us at K = 4096 (where split-K 3 takes 31 + 8), equal under split-K.  The launch is 6 us of fixed cost + 0.73 us per
// K step: a lone 4-wave workgroup is not waiting for memory, with ONE wave per SIMD its 16 fragment reads (64 KiB of LDS traffic per step
// and CU = 512 cycles) and its 32 MFMAs (512 cycles) run one after the other, whatever is in flight.  Two co-resident workgroups (split-K 2:
// 1.02 us per step each) reach 1.43 x that throughput — which also bounds what an 8-wave K-split tile could gain (2.6 us at K = 1024,
// nothing against split-K 3 at K >= 3072).  Kept for the record behind SCL_EXPERIMENTS (SCL_GEMM_DEEP=1 selects it).
#ifndef SCL_DEEP_STAGES
#define SCL_DEEP_STAGES 5
#endif
constexpr int DEEP_S = SCL_DEEP_STAGES;
constexpr int DEEP_LDS = DEEP_S * 2 * TILE_BYTES;      // 5 stages = all 160 KiB

template <bool AT, bool BT, int S>
__global__ __launch_bounds__(256, 1) void scl_gemm_deep_kernel(const GemmK d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_n = (d.N + BN - 1) / BN;
    int tm, tn;
    tile_coords(blockIdx.x, gridDim.x, (d.M + BM - 1) / BM, tiles_n, tm, tn, d.group_m);
    const int m0 = tm * BM, n0 = tn * BN;
    int z = blockIdx.z;
    const int ksplit = __builtin_amdgcn_readfirstlane(z % d.splitk); z /= d.splitk;
    const int z1 = __builtin_amdgcn_readfirstlane(z / d.nb2), z2 = z - z1 * d.nb2;
    const int nk_total = (d.K + BK - 1) / BK;
    const int nk_per = __builtin_amdgcn_readfirstlane((nk_total + d.splitk - 1) / d.splitk);
    const int kbegin = ksplit * nk_per * BK;
    int kend = kbegin + nk_per * BK; if (kend > d.K) kend = d.K;
    const int nk = kend > kbegin ? (kend - kbegin + BK - 1) / BK : 0;
    const char* Ab = reinterpret_cast<const char*>(d.A.ptr) + z1 * d.A.bs1 + z2 * d.A.bs2;
    const char* Bb = reinterpret_cast<const char*>(d.B.ptr) + z1 * d.B.bs1 + z2 * d.B.bs2;

    typename DmaSel<AT>::type sa;
    typename DmaSel<BT>::type sb;
    sa.init(d.A, Ab, m0, d.M, kbegin, kend, lane, wave);
    sb.init(d.B, Bb, n0, d.N, kbegin, kend, lane, wave);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int STAGE = 2 * TILE_BYTES;
    // prologue: tiles 0 .. S-2 (out-of-range no-ops past the last one)
#pragma unroll
    for (int s = 0; s < S - 1; ++s) {
        if (s) { sa.advance(); sb.advance(); }
        sa.issue(d.A, smem + s * STAGE, wave); sb.issue(d.B, smem + s * STAGE + TILE_BYTES, wave);
    }
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // this wave's 8 pieces of tile kt have landed; the S - 2 younger tiles stay in flight
        if (S == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (S == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // ... for every wave, and every wave is done reading the stage of tile kt - 1
        {
            sa.advance(); sb.advance();
            int ns = stage + (S - 1); if (ns >= S) ns -= S;
            char* nb = smem + ns * STAGE;
            sa.issue(d.A, nb, wave); sb.issue(d.B, nb + TILE_BYTES, wave);
        }
        const char* tA = smem + stage * STAGE;
        const char* tB = tA + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = AT ? frag_t_raw(tA, wr * 4 + i, ks, lane) : frag_k(tA, wr * 4 + i, ks, lane);
                fb[i] = BT ? frag_t_raw(tB, wc * 4 + i, ks, lane) : frag_k(tB, wc * 4 + i, ks, lane);
            }
            if (AT || BT) lds_wait_frags(fa, fb);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        stage = stage + 1 == S ? 0 : stage + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no DMA may still target this block's LDS when it retires
    gemm_epilogue(d, acc, m0, n0, z1, z2, ksplit, lane, wr, wc);
}
#endif


// (A 128x128 / BK = 32 / 5-stage variant that doubles the bytes in flight per CU was measured 4-19 % SLOWER than the kernel
// above on every encoder shape — the extra barrier per 16 MFMAs costs more than the deeper prefetch returns — and removed.)

// The 256x128 ("big") and 256x256 ping-pong ("p8") kernels and the two-workgroups-per-CU / persistent experiments of gemm_x2.hip /
// gemm_w8.hip lost every A/B against the default family and are NOT part of the shipped library: they are compiled
// only with -DSCL_EXPERIMENTS (SCL_BUILD_DEFINES=-DSCL_EXPERIMENTS python scl-deepfake-audio-detection_amd/build.py), for the record and
// for their bit-identity tests.  Unused instantiations cost the default path 5-15 % through the instruction cache.
#ifdef SCL_EXPERIMENTS
// ---- "big" variant: 256x128 tile, 8 waves, 3-stage LDS-DMA ring --------------------------------------
// The 128x128 kernels above are latency-bound on the K loop (rocprofv3: 48 % of wave cycles in
// s_waitcnt/barrier, MFMA pipe 22 % busy, ~3300 cycles per 64-deep K step with 64 KiB in flight per CU).
// This variant keeps TWO 48 KiB stages (A 256x64 + B 128x64) in flight per CU behind a counted
// `s_waitcnt vmcnt(6)` and raw `s_barrier`s (a __syncthreads() would drain the DMA queue), raises the
// arithmetic intensity of a stage by a third, and still runs 2 waves per SIMD (512 threads, 1 block/CU,
// 144 KiB of the 160 KiB LDS).  Per K step: wait(tile kt) -> barrier -> issue(tile kt+2) -> 32 MFMA/wave.
// Loads past the last tile are issued at offset 0xFFFFFFFF (no fetch, zeros) so the vmcnt arithmetic is
// the same in every iteration.
constexpr int BIG_BM = 256;
constexpr int BIG_STAGE = 3 * TILE_BYTES;   // A(2 x 16 KiB) + B(16 KiB)
constexpr int BIG_LDS = 3 * BIG_STAGE;      // 147456 B

template <int NSUB>   // K-contiguous operand with NSUB*128 rows; wave owns pieces wave*(2*NSUB) + i
struct BigK {
    static constexpr int NP = 2 * NSUB;
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned rowoff[NP];
    int kc[NP];
    int kcur, kend;
    __device__ __forceinline__ void init(const OpK& o, const char* base, int row0, int rowlimit, int kbegin, int kend_, int lane, int wave) {
        rsrc = make_rsrc(base);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int p = wave * NP + i;
            const int row = 8 * p + (lane >> 3);
            const int r = row0 + row;
            rowoff[i] = r < rowlimit ? row_off(o, (unsigned)r) : OOB;
            kc[i] = (lane & 7) ^ ((row >> 1) & 7);
        }
        kcur = kbegin; kend = kend_;
    }
    __device__ __forceinline__ void issue(const OpK& o, char* tile, int wave) const {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int k = kcur + 8 * kc[i];
            const unsigned off = (k < kend && rowoff[i] != OOB) ? rowoff[i] + col_off(o, (unsigned)k) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(tile + (wave * NP + i) * 1024), 16, off, 0, 0, 0);
        }
    }
    __device__ __forceinline__ void advance() { kcur += BK; }
};

template <int NSUB>   // transposed operand with NSUB sub-tiles of [64 k][128 cols]
struct BigT {
    static constexpr int NP = 2 * NSUB;
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned coloff[NP];
    int krow[NP];
    int kcur, kend;
    __device__ __forceinline__ void init(const OpK& o, const char* base, int col0, int collimit, int kbegin, int kend_, int lane, int wave) {
        rsrc = make_rsrc(base);
        const int s16 = lane & 15;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int p = wave * NP + i;
            const int sub = p >> 4, pp = p & 15;
            const int kr = 4 * pp + (lane >> 4);
            const int sw = (kr & 3) | (((kr >> 3) & 1) << 2);
            const int col = col0 + sub * 128 + 8 * ((((s16 >> 1) ^ sw) << 1) | (s16 & 1));
            krow[i] = kr;
            coloff[i] = col < collimit ? col_off(o, (unsigned)col) : OOB;
        }
        kcur = kbegin; kend = kend_;
    }
    __device__ __forceinline__ void issue(const OpK& o, char* tile, int wave) const {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int rr = kcur + krow[i];
            const unsigned off = (rr < kend && coloff[i] != OOB) ? row_off(o, (unsigned)rr) + coloff[i] : OOB;
            // pieces of sub-tile `sub` start at sub*16 KiB; within it piece pp is at pp*1 KiB: (wave*NP+i)*1024 covers both
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(tile + (wave * NP + i) * 1024), 16, off, 0, 0, 0);
        }
    }
    __device__ __forceinline__ void advance() { kcur += BK; }
};

template <bool T, int NSUB> struct BigSel { typedef BigK<NSUB> type; };
template <int NSUB> struct BigSel<true, NSUB> { typedef BigT<NSUB> type; };

template <bool AT, bool BT>
__global__ __launch_bounds__(512, 2) void scl_gemm_big_kernel(const GemmK d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7
    const int wr = wave >> 1, wc = wave & 1;                      // 4 x 2 waves of 64x64
    const int tiles_n = (d.N + BN - 1) / BN;
    int tm, tn;
    tile_coords(blockIdx.x, gridDim.x, (d.M + BIG_BM - 1) / BIG_BM, tiles_n, tm, tn);
    const int m0 = tm * BIG_BM, n0 = tn * BN;
    int z = blockIdx.z;
    const int ksplit = __builtin_amdgcn_readfirstlane(z % d.splitk); z /= d.splitk;      // uniform, but integer division runs on the vector ALU: back to an SGPR
    const int z1 = __builtin_amdgcn_readfirstlane(z / d.nb2), z2 = z - z1 * d.nb2;
    const int nk_total = (d.K + BK - 1) / BK;
    const int nk_per = __builtin_amdgcn_readfirstlane((nk_total + d.splitk - 1) / d.splitk);
    const int kbegin = ksplit * nk_per * BK;
    int kend = kbegin + nk_per * BK; if (kend > d.K) kend = d.K;
    const int nk = kend > kbegin ? (kend - kbegin + BK - 1) / BK : 0;
    const char* Ab = reinterpret_cast<const char*>(d.A.ptr) + z1 * d.A.bs1 + z2 * d.A.bs2;
    const char* Bb = reinterpret_cast<const char*>(d.B.ptr) + z1 * d.B.bs1 + z2 * d.B.bs2;

    typename BigSel<AT, 2>::type sa;
    typename BigSel<BT, 1>::type sb;
    sa.init(d.A, Ab, m0, d.M, kbegin, kend, lane, wave);
    sb.init(d.B, Bb, n0, d.N, kbegin, kend, lane, wave);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // prologue: tiles 0 and 1 (OOB no-ops when nk < 2)
    sa.issue(d.A, smem, wave); sb.issue(d.B, smem + 2 * TILE_BYTES, wave);
    sa.advance(); sb.advance();
    sa.issue(d.A, smem + BIG_STAGE, wave); sb.issue(d.B, smem + BIG_STAGE + 2 * TILE_BYTES, wave);
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // tile kt landed (this wave's pieces); tile kt+1 stays in flight
        __builtin_amdgcn_s_barrier();                       // ... for every wave; all waves are done with stage (kt+2)%3
        {
            sa.advance(); sb.advance();
            int ns = stage + 2; if (ns >= 3) ns -= 3;
            char* nb = smem + ns * BIG_STAGE;
            sa.issue(d.A, nb, wave); sb.issue(d.B, nb + 2 * TILE_BYTES, wave);
        }
        const char* tA = smem + stage * BIG_STAGE;
        const char* tB = tA + 2 * TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = AT ? frag_t_raw(tA + (wr >> 1) * TILE_BYTES, (wr & 1) * 4 + i, ks, lane) : frag_k(tA, wr * 4 + i, ks, lane);
                fb[i] = BT ? frag_t_raw(tB, wc * 4 + i, ks, lane) : frag_k(tB, wc * 4 + i, ks, lane);
            }
            if (AT || BT) lds_wait_frags(fa, fb);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        stage = stage + 1 == 3 ? 0 : stage + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no DMA may still target this block's LDS when it retires
    gemm_epilogue(d, acc, m0, n0, z1, z2, ksplit, lane, wr, wc);
}

// ---- "p8" variant: 256x256 tile, 8 waves as 2 (M) x 4 (N), 128x64 per wave, two wave groups in ping-pong ------
// In the kernels above every wave of a block reaches the K-step barrier together, then all of them read LDS (MFMA pipe
// idle), then all of them issue MFMAs (LDS idle).  Here the block's two wave groups (wr = 0 / 1, one wave of each per
// SIMD) run half a phase apart: while one group issues its 16 MFMAs of a phase (one 64x32 quadrant x K=64) under
// s_setprio(1), the other group does its ds_reads and LDS-DMA issues for the next phase.  Four phases per K tile:
//   p0: read A(rows 0-63) + B(cols 0-31)   | DMA: A halves of tile t+1 -> other buffer   | MFMA quadrant (0,0)
//   p1: read B(cols 32-63)                  |                                              | MFMA quadrant (0,1)
//   p2: read A(rows 64-127)                 |                                              | MFMA quadrant (1,1)
//   p3: (B cols 0-31 still in registers)    | vmcnt(0): tile t+1 landed; DMA: B halves of tile t+2 -> this buffer | (1,0)
// LDS: 2 buffers x {A0, A1, B0, B1} half-tiles of 16 KiB (128 KiB).  Hazards (raw s_barrier, counted waits):
//   RAW — a tile is waited for (vmcnt) before the FIRST barrier of phase p3 and first read in the next phase, so every
//         wave's wait precedes a barrier every reader has passed, also across the half-phase stagger;
//   WAR — a half-tile is re-staged >= 2 phases after its last ds_read (B: read p0/p1 -> staged p3; A: read p0/p2 ->
//         staged p0 of the next tile), which covers the group that runs half a phase behind.
// Accumulation order per output element is the same as in the 128x128 kernels (k ascending), so results are bit-identical.
constexpr int P8_BM = 256, P8_BN = 256;
constexpr int P8_BUF = 4 * TILE_BYTES;
constexpr int P8_LDS = 2 * P8_BUF;          // 131072 B

#define P8_PHASE(MH, NH, FBSEL)                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    __builtin_amdgcn_s_barrier();                                                                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    __builtin_amdgcn_s_setprio(1);                                                                         \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                       \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                      \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                  \
                acc[MH][i][2 * (NH) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FBSEL[j][ks], fa[i][ks], acc[MH][i][2 * (NH) + j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    __builtin_amdgcn_s_barrier();

template <bool AT, bool BT>
__global__ __launch_bounds__(512, 2) void scl_gemm_p8_kernel(const GemmK d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7
    const int wr = wave >> 2, wc = wave & 3;
    const int tiles_n = (d.N + P8_BN - 1) / P8_BN;
    int tm, tn;
    tile_coords(blockIdx.x, gridDim.x, (d.M + P8_BM - 1) / P8_BM, tiles_n, tm, tn);
    const int m0 = tm * P8_BM, n0 = tn * P8_BN;
    int z = blockIdx.z;
    const int ksplit = __builtin_amdgcn_readfirstlane(z % d.splitk); z /= d.splitk;      // uniform, but integer division runs on the vector ALU: back to an SGPR
    const int z1 = __builtin_amdgcn_readfirstlane(z / d.nb2), z2 = z - z1 * d.nb2;
    const int nk_total = (d.K + BK - 1) / BK;
    const int nk_per = __builtin_amdgcn_readfirstlane((nk_total + d.splitk - 1) / d.splitk);
    const int kbegin = ksplit * nk_per * BK;
    int kend = kbegin + nk_per * BK; if (kend > d.K) kend = d.K;
    const int nk = kend > kbegin ? (kend - kbegin + BK - 1) / BK : 0;
    const char* Ab = reinterpret_cast<const char*>(d.A.ptr) + z1 * d.A.bs1 + z2 * d.A.bs2;
    const char* Bb = reinterpret_cast<const char*>(d.B.ptr) + z1 * d.B.bs1 + z2 * d.B.bs2;

    typename BigSel<AT, 1>::type la0, la1;
    typename BigSel<BT, 1>::type lb0, lb1;
    la0.init(d.A, Ab, m0, d.M, kbegin, kend, lane, wave);
    la1.init(d.A, Ab, m0 + 128, d.M, kbegin, kend, lane, wave);
    lb0.init(d.B, Bb, n0, d.N, kbegin, kend, lane, wave);
    lb1.init(d.B, Bb, n0 + 128, d.N, kbegin, kend, lane, wave);

    f32x4 acc[2][4][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#define P8_ISSUE_A(BUF) { la0.issue(d.A, (BUF), wave); la1.issue(d.A, (BUF) + TILE_BYTES, wave); la0.advance(); la1.advance(); }
#define P8_ISSUE_B(BUF) { lb0.issue(d.B, (BUF) + 2 * TILE_BYTES, wave); lb1.issue(d.B, (BUF) + 3 * TILE_BYTES, wave); lb0.advance(); lb1.advance(); }

    // prologue: all of tile 0, B halves of tile 1 (loads past the last tile are offset-0xFFFFFFFF no-ops: counts stay uniform)
    P8_ISSUE_A(smem) P8_ISSUE_B(smem) P8_ISSUE_B(smem + P8_BUF)
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();          // group 1 runs half a phase behind group 0

    bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
    for (int kt = 0; kt < nk; ++kt) {
        char* buf = smem + (kt & 1) * P8_BUF;
        char* obuf = smem + ((kt & 1) ^ 1) * P8_BUF;
        const char* tA = buf + wr * TILE_BYTES;
        const char* tB = buf + (2 + (wc >> 1)) * TILE_BYTES;
        const int nb = (wc & 1) * 4;
        // ---- p0
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int j = 0; j < 2; ++j) fb0[j][ks] = BT ? frag_t_raw(tB, nb + j, ks, lane) : frag_k(tB, nb + j, ks, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i][ks] = AT ? frag_t_raw(tA, i, ks, lane) : frag_k(tA, i, ks, lane);
        }
        P8_ISSUE_A(obuf)
        P8_PHASE(0, 0, fb0)
        // ---- p1
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int j = 0; j < 2; ++j) fb1[j][ks] = BT ? frag_t_raw(tB, nb + 2 + j, ks, lane) : frag_k(tB, nb + 2 + j, ks, lane);
        }
        P8_PHASE(0, 1, fb1)
        // ---- p2
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i][ks] = AT ? frag_t_raw(tA, 4 + i, ks, lane) : frag_k(tA, 4 + i, ks, lane);
        }
        P8_PHASE(1, 1, fb1)
        // ---- p3
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile kt+1 has landed (this wave's pieces); read from the next phase on
        P8_ISSUE_B(buf)
        P8_PHASE(1, 0, fb0)
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();           // balance the stagger
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // no DMA may still target this block's LDS when it retires
    gemm_epilogue(d, acc[0], m0 + wr * 128, n0 + wc * 64, z1, z2, ksplit, lane, 0, 0);
    gemm_epilogue(d, acc[1], m0 + wr * 128 + 64, n0 + wc * 64, z1, z2, ksplit, lane, 0, 0);
}
#undef P8_PHASE
#undef P8_ISSUE_A
#undef P8_ISSUE_B

#endif  // SCL_EXPERIMENTS

__global__ void scl_reduce_slabs_kernel(const float* __restrict__ slabs, float* __restrict__ out, int64_t n,
                                        int nslabs, int64_t stride) {
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int64_t step = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t i = i0; i < n; i += step) {
        if (i + 4 <= n) {
            float4 s = *reinterpret_cast<const float4*>(slabs + i);
            for (int k = 1; k < nslabs; ++k) {
                const float4 t = *reinterpret_cast<const float4*>(slabs + k * stride + i);
                s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
            }
            *reinterpret_cast<float4*>(out + i) = s;
        } else {
            for (int64_t j = i; j < n; ++j) {
                float s = slabs[j];
                for (int k = 1; k < nslabs; ++k) s += slabs[k * stride + j];
                out[j] = s;
            }
        }
    }
}

}  // namespace

namespace sclg {
// Granlund-Montgomery round-up magic: q = (mulhi(magic, n) + n) >> shift for all n < 2^31, 1 <= d < 2^31
void make_magic(unsigned dv, unsigned* magic, unsigned* shift) {
    unsigned l = 0;
    while ((1ull << l) < dv) ++l;
    *shift = l;
    *magic = (unsigned)((((1ull << l) - dv) << 32) / dv + 1);
}

// host SclOperand -> device OpK for elements of `esz` bytes (2: bf16, 4: f32); 16-byte vectors = 16 / esz elements
bool fill_operand(const SclOperand& o, const char* name, long long rows, long long contig, OpK* k, int esz) {
    const int vec = 16 / esz;
    if (!o.ptr || ((uintptr_t)o.ptr & 15)) { scl_set_error("gemm: %s ptr null or not 16B aligned", name); return false; }
    if (o.rpb < 1 || o.cin < vec) { scl_set_error("gemm: %s rpb/cin invalid", name); return false; }
    if ((o.ld % vec) || (o.rbstride % vec) || (o.cout % vec) || (o.bs1 % vec) || (o.bs2 % vec) || (o.cin != 0x7fffffff && (o.cin % vec))) {
        scl_set_error("gemm: %s strides must be multiples of %d elements", name, vec); return false;
    }
    // largest element offset this launch can touch must stay below 2^32 bytes (32-bit byte offsets)
    const long long rq = (rows - 1) / o.rpb, rr = o.rpb == 0x7fffffff ? rows - 1 : (long long)o.rpb - 1;
    const long long cq = o.cin == 0x7fffffff ? 0 : (contig - 1) / o.cin;
    const long long cr = o.cin == 0x7fffffff ? contig - 1 : (long long)o.cin - 1;
    const long long maxoff = rq * o.rbstride + (rr < rows - 1 ? rr : rows - 1) * (long long)o.ld + cq * o.cout + cr + vec;
    if (maxoff < 0 || maxoff * esz >= (1ll << 32) - 32 || rows >= (1ll << 31)) {
        scl_set_error("gemm: %s extent exceeds 32-bit byte offsets (%lld elements)", name, maxoff); return false;
    }
    k->ptr = o.ptr; k->bs1 = o.bs1 * esz; k->bs2 = o.bs2 * esz;
    k->rb_bytes = (unsigned)(o.rbstride * esz); k->ld_bytes = (unsigned)o.ld * (unsigned)esz; k->cout_bytes = (unsigned)(o.cout * esz);
    k->rpb = (unsigned)o.rpb; make_magic((unsigned)o.rpb, &k->rpb_magic, &k->rpb_shift);
    k->cin = (unsigned)o.cin; make_magic((unsigned)o.cin, &k->cin_magic, &k->cin_mshift);
    k->esz_shift = esz == 4 ? 2u : 1u;
    return true;
}
}  // namespace sclg

namespace {

// wide tiles with a runtime row pitch (gemm_w8.hip): whole rounds of the 256 CUs at M = 64 x 199 rows; picked when the problem fills at
// least half a round of them and the operands advance linearly along K
bool gemm_pick_w8(const GemmK& k, bool at, bool bt, const SclGemmDesc& d, long long zdim, W8Plan* plan) {
    const int w8_env = 1;
    if (d.flags & (SCL_GEMM_NO_W8 | SCL_GEMM_FORCE_P8 | SCL_GEMM_FORCE_BIG | SCL_GEMM_NO_DMA)) return false;
    const bool a_whole = at ? (d.M % 8 == 0) : (d.K % 8 == 0);
    const bool b_whole = bt ? (d.N % 8 == 0) : (d.K % 8 == 0);
    // CUs the tile plan may count on: all 256, unless the caller says a communication library's persistent kernels hold some of
    // them during the backward (data parallel over RCCL): a 248-block grid on 240 free CUs would take two rounds
    static const int ncu_env = [] { const char* e = getenv("SCL_GEMM_CUS"); const int v = e ? atoi(e) : 256; return v >= 32 && v <= 256 ? v : 256; }();
    if (!a_whole || !b_whole || !scl_gemm_w8_plan(k, at, bt, d, zdim, ncu_env, plan)) return false;
    // row-major A with 128-159 wide tiles (qkv forward of a pack-sized step: 11 x 12 tiles): the 432 tiles of the 128x128 kernel win
    // 24.4 vs 31.8 us; from 176 wide tiles on (fc1 forward, 11 x 16) the wide kernel does (32.5 vs 38.9) — tools/splitk_probe.py
    const long long need = at ? 128 : 160;
    return (d.flags & SCL_GEMM_FORCE_W8) || (w8_env && d.N >= 192 && plan->tiles * zdim >= need);
}

// 208 x 128 tiles, two 4-wave workgroups per CU (gemm_x2.hip): the epilogue of one tile runs under the K loop of the other.
// SCL_GEMM_X2: 0 never, 1 automatic (below), 2 whenever the operands can be addressed.
#ifdef SCL_EXPERIMENTS
bool gemm_pick_x2(const GemmK& k, bool at, bool bt, const SclGemmDesc& d, long long zdim, W8Plan* plan) {
    static const int x2_env = [] { const char* e = getenv("SCL_GEMM_X2"); return e ? atoi(e) : 0; }();
    if (d.flags & (SCL_GEMM_NO_X2 | SCL_GEMM_FORCE_W8 | SCL_GEMM_FORCE_P8 | SCL_GEMM_FORCE_BIG | SCL_GEMM_NO_DMA)) return false;
    if (!(d.flags & SCL_GEMM_FORCE_X2) && x2_env == 0) return false;
    const bool a_whole = at ? (d.M % 8 == 0) : (d.K % 8 == 0);
    const bool b_whole = bt ? (d.N % 8 == 0) : (d.K % 8 == 0);
    static const int ncu_env = [] { const char* e = getenv("SCL_GEMM_CUS"); const int v = e ? atoi(e) : 256; return v >= 32 && v <= 256 ? v : 256; }();
    if (!a_whole || !b_whole || !scl_gemm_x2_plan(k, at, bt, d, zdim, ncu_env, plan)) return false;
    if ((d.flags & SCL_GEMM_FORCE_X2) || x2_env == 2) return true;
    // automatic: forward / data-gradient shapes (A K-contiguous) that fill at least the 256 CUs once
    return !at && d.N >= 128 && plan->tiles * zdim >= 248;
}

#else
bool gemm_pick_x2(const GemmK&, bool, bool, const SclGemmDesc&, long long, W8Plan*) { return false; }
#endif
}  // namespace

extern "C" int scl_gemm_uses_wide_tiles(const SclGemmDesc* dp) {
    if (!dp || dp->M <= 0 || dp->N <= 0 || dp->K <= 0 || dp->nb1 < 1 || dp->nb2 < 1 || dp->splitk < 1) return 0;
    GemmK k; W8Plan plan;
    const bool at = dp->flags & SCL_GEMM_A_T, bt = dp->flags & SCL_GEMM_B_T;
    const long long zdim = (long long)dp->nb1 * dp->nb2 * dp->splitk;
    if (gemm_pick_x2(k, at, bt, *dp, zdim, &plan)) return 3;
    return gemm_pick_w8(k, at, bt, *dp, zdim, &plan) ? (plan.variant == 2 ? 4 : 1 + plan.variant) : 0;      // 4: the 112-row wide tile
}

// partial rows of the fused column sums (SclGemmDesc.colsum_part): wide tiles of gemm_w8.hip only, one un-batched problem, whole 8-column
// vectors in every row
static int gemm_colsum_rows(const SclGemmDesc& d) {
    if (d.M <= 0 || d.N <= 0 || d.K <= 0 || d.nb1 != 1 || d.nb2 != 1 || d.splitk != 1 || (d.N & 7) || (d.ldc & 7) || (d.flags & SCL_GEMM_AB_F32)) return 0;
    GemmK k; W8Plan plan;
    const bool at = d.flags & SCL_GEMM_A_T, bt = d.flags & SCL_GEMM_B_T;
    const bool a_whole = at ? (d.M % 8 == 0) : (d.K % 8 == 0), b_whole = bt ? (d.N % 8 == 0) : (d.K % 8 == 0);
    SclGemmDesc dq = d;      // plan as the launch WITH the partial rows will be planned (the 112-row tile does not write them: scl_gemm_w8_plan)
    if (!dq.colsum_part) dq.colsum_part = reinterpret_cast<float*>(16);
    if (!a_whole || !b_whole || (d.flags & SCL_GEMM_NO_DMA) || gemm_pick_x2(k, at, bt, dq, 1, &plan) || !gemm_pick_w8(k, at, bt, dq, 1, &plan)) return 0;
    return plan.tiles_m * 4;
}
extern "C" int scl_gemm_colsum_rows(const SclGemmDesc* dp) { return dp ? gemm_colsum_rows(*dp) : 0; }

// descriptor -> kernel argument block (validation included): shared by scl_gemm_bf16 and scl_gemm_bf16_group
static int gemm_fill_k(const SclGemmDesc& d, GemmK& k) {
    SCL_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0, "gemm: M,N,K must be positive (%d,%d,%d)", d.M, d.N, d.K);
    SCL_REQUIRE(d.nb1 >= 1 && d.nb2 >= 1 && d.splitk >= 1, "gemm: nb1/nb2/splitk must be >= 1");
    const bool at = d.flags & SCL_GEMM_A_T, bt = d.flags & SCL_GEMM_B_T;
    const bool f32ab = d.flags & SCL_GEMM_AB_F32;
    if (!f32ab) {
        if (!fill_operand(d.A, "A", at ? d.K : d.M, at ? d.M : d.K, &k.A, 2)) return SCL_EINVAL;
        if (!fill_operand(d.B, "B", bt ? d.K : d.N, bt ? d.N : d.K, &k.B, 2)) return SCL_EINVAL;
    }
    SCL_REQUIRE(d.C && d.c_rpb >= 1, "gemm: C null or c_rpb < 1");
    const int rmode = (d.flags >> SCL_GEMM_RMODE_SHIFT) & 0xF;
    SCL_REQUIRE(rmode == 0 || d.R, "gemm: RMODE set but R is null");
    SCL_REQUIRE(!(d.flags & SCL_GEMM_HAS_C2) || d.C2, "gemm: HAS_C2 set but C2 is null");
    SCL_REQUIRE(!(d.flags & SCL_GEMM_HAS_BIAS) || d.bias, "gemm: HAS_BIAS set but bias is null");
    SCL_REQUIRE(d.splitk == 1 || ((d.flags & SCL_GEMM_C_F32) && rmode == 0 && !(d.flags & (SCL_GEMM_HAS_BIAS | SCL_GEMM_HAS_C2))
                                  && ((d.flags >> SCL_GEMM_ACT_SHIFT) & 0xF) == 0),
                "gemm: split-K needs a plain f32 slab output");
    const long long zdim = (long long)d.nb1 * d.nb2 * d.splitk;
    SCL_REQUIRE(zdim <= 65535, "gemm: batch*splitk too large (%lld)", zdim);
    k.C = d.C; k.C2 = d.C2; k.R = d.R; k.bias = d.bias; k.colsum = d.colsum_part;
    SCL_REQUIRE(!d.colsum_part || (gemm_colsum_rows(d) > 0 && ((uintptr_t)d.colsum_part & 15) == 0),
                "gemm: colsum_part needs a launch on the wide tiles (scl_gemm_colsum_rows() > 0) and a 16-byte aligned buffer");
    k.c_bs1 = d.c_bs1; k.c_bs2 = d.c_bs2; k.c_rbstride = d.c_rbstride; k.c_split_stride = d.c_split_stride; k.bias_bs2 = d.bias_bs2;
    k.c_rpb = (unsigned)d.c_rpb; make_magic((unsigned)d.c_rpb, &k.c_magic, &k.c_shift);
    k.ldc = d.ldc; k.M = d.M; k.N = d.N; k.K = d.K; k.nb2 = d.nb2; k.splitk = d.splitk; k.flags = d.flags;
    k.alpha = d.alpha; k.drop_p = d.drop_p; k.drop_seed = d.drop_seed;
#ifdef SCL_EXPERIMENTS
    // round-6 A/B of the tile order (tools/xcd_order_probe.sh): SCL_GEMM_GROUP_M = rows of a group (1 .. 255), SCL_GEMM_XCD_ROWS = R of the
    // R x 8/R region grid (1, 2, 4, 8; 0 = shipped order)
    static const int group_m_env = [] {
        const char* e = getenv("SCL_GEMM_GROUP_M"); const char* x = getenv("SCL_GEMM_XCD_ROWS");
        const int g = e ? atoi(e) : 8, r = x ? atoi(x) : 0;
        return (g >= 1 && g <= 255 ? g : 8) | ((r == 1 || r == 2 || r == 4 || r == 8 ? r : 0) << 8);
    }();
#else
    const int group_m_env = 8;      // 2 .. 62 swept in round 3: 8 and 16 equal, the extremes slower; round 6 added the XCD region grid to the sweep (profiles/r6_xcd_order_ab.txt)
#endif
    static const bool epi_generic = [] { const char* e = getenv("SCL_W8_EPI_GENERIC"); return e && atoi(e) != 0; }();
    k.group_m = group_m_env; k.tile_m = 0; k.debug = epi_generic ? 16 : 0;      // bit 4: generic epilogue loops (A/B); the wide launch sets its own bits
    // 4-wide vector epilogue needs every 4-column group 16-byte (f32) / 8-byte (bf16) aligned in C, C2, R and bias
    auto al = [](const void* p, int bytes) { return p == nullptr || ((uintptr_t)p & (bytes - 1)) == 0; };
    const bool strides4 = !(d.ldc & 3) && !(d.c_bs1 & 3) && !(d.c_bs2 & 3) && !(d.c_rbstride & 3) && !(d.c_split_stride & 3) && !(d.bias_bs2 & 3);
    k.vec_ok = strides4 && al(d.C, (d.flags & SCL_GEMM_C_F32) ? 16 : 8) && al(d.C2, (d.flags & SCL_GEMM_C2_F32) ? 16 : 8) &&
               al(d.R, (d.flags & SCL_GEMM_R_F32) ? 16 : 8) && al(d.bias, 16);
    // the wide epilogue writes its column-sum partial rows from the 8-column vector path only: without it the rows would stay unwritten
    SCL_REQUIRE(!d.colsum_part || k.vec_ok, "gemm: colsum_part needs 16-byte aligned C / C2 / R / bias and strides that are multiples of 4");
    SCL_REQUIRE(!((unsigned)d.flags & SCL_GEMM_C_SPLIT3) || (!(d.flags & (SCL_GEMM_C_F32 | SCL_GEMM_HAS_C2 | SCL_GEMM_AB_F32)) && d.splitk == 1 && d.nb1 == 1 && d.nb2 == 1 &&
                                                             (d.N & 7) == 0 && d.ldc == 3 * d.N && d.c_rpb >= d.M && k.vec_ok && ((uintptr_t)d.C & 15) == 0 && !d.colsum_part),
                "gemm: C_SPLIT3 needs one un-batched problem, bf16 C with ldc = 3 N, N %% 8 == 0, 16-byte aligned C and no second output");
    return SCL_OK;
}

// Several independent weight-gradient problems in ONE launch (gemm_w8.hip: scl_gemm_w8s_group_kernel): every member has both operands
// transposed (C = A^T B over the K rows), flat K rows, K % 64 == 0, no batching, no split-K and a plain f32 output.  _ok answers whether
// a descriptor list qualifies (1) without launching; scl_gemm_bf16_group returns SCL_EUNSUPPORTED for a list that does not.
static int gemm_group_prepare(const SclGemmDesc* descs, int n, GemmK* ks) {
    if (!descs || n < 1 || n > W8_GROUP_MAX) return SCL_EUNSUPPORTED;
    for (int i = 0; i < n; ++i) {
        const SclGemmDesc& d = descs[i];
        if (d.flags & SCL_GEMM_AB_F32) return SCL_EUNSUPPORTED;
        const int rc = gemm_fill_k(d, ks[i]);
        if (rc != SCL_OK) return rc;
        if (!scl_gemm_w8_group_member_ok(ks[i], d.flags & SCL_GEMM_A_T, d.flags & SCL_GEMM_B_T, d)) return SCL_EUNSUPPORTED;
    }
    return SCL_OK;
}
extern "C" int scl_gemm_bf16_group_ok(const SclGemmDesc* descs, int n) {
    GemmK ks[W8_GROUP_MAX];
    return gemm_group_prepare(descs, n, ks) == SCL_OK ? 1 : 0;
}
static int gemm_group_run(const SclGemmDesc* descs, const int32_t* tile0, const int32_t* ntile, int n, void* stream, const char* what) {
    GemmK ks[W8_GROUP_MAX];
    const int rc = gemm_group_prepare(descs, n, ks);
    if (rc == SCL_EUNSUPPORTED) scl_set_error("%s: 1..%d members, each A^T B with flat K rows, K %% 64 == 0, no batch / split-K, plain f32 output", what, W8_GROUP_MAX);
    if (rc != SCL_OK) return rc;
    double flops = 0.0;
    for (int i = 0; i < n; ++i) {
        const int all = scl_gemm_w8_group_tiles(ks[i]);
        if (tile0 || ntile) {
            SCL_REQUIRE(tile0 && ntile && tile0[i] >= 0 && ntile[i] >= 1 && tile0[i] + ntile[i] <= all, "%s: member %d covers tiles [%d, %d) of %d", what, i,
                        tile0 ? tile0[i] : 0, (tile0 ? tile0[i] : 0) + (ntile ? ntile[i] : 0), all);
        }
        flops += 2.0 * descs[i].M * descs[i].N * (double)descs[i].K * (ntile ? (double)ntile[i] / all : 1.0);
    }
    {
        SclProfScope prof(SCL_KID_GEMM, (hipStream_t)stream, flops, true);
        int tiles = 0;
        for (int i = 0; i < n; ++i) tiles += ntile ? ntile[i] : scl_gemm_w8_group_tiles(ks[i]);
        prof.note(descs[0].M, descs[0].N, descs[0].K, tiles, n, 4);
        scl_gemm_w8_group_launch(ks, n, tile0, ntile, (hipStream_t)stream);
    }
    return scl_check_launch(what);
}
extern "C" int scl_gemm_bf16_group(const SclGemmDesc* descs, int n, void* stream) {
    return gemm_group_run(descs, nullptr, nullptr, n, stream, "scl_gemm_bf16_group");
}
extern "C" int scl_gemm_bf16_group_tiles(const SclGemmDesc* desc) {
    GemmK k;
    if (gemm_group_prepare(desc, 1, &k) != SCL_OK) return 0;
    return scl_gemm_w8_group_tiles(k);
}
extern "C" int scl_gemm_bf16_group_part(const SclGemmDesc* descs, const int32_t* tile0, const int32_t* ntile, int n, void* stream) {
    SCL_REQUIRE(tile0 && ntile, "gemm group part: tile ranges missing");
    return gemm_group_run(descs, tile0, ntile, n, stream, "scl_gemm_bf16_group_part");
}

extern "C" int scl_gemm_bf16(const SclGemmDesc* dp, void* stream) {
    SCL_REQUIRE(dp, "gemm: null desc");
    const SclGemmDesc& d = *dp;
    GemmK k;
    {
        const int rc = gemm_fill_k(d, k);
        if (rc != SCL_OK) return rc;
    }
    const bool at = d.flags & SCL_GEMM_A_T, bt = d.flags & SCL_GEMM_B_T;
    const bool f32ab = d.flags & SCL_GEMM_AB_F32;
    const int tiles = ((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN);
    const long long zdim = (long long)d.nb1 * d.nb2 * d.splitk;
    dim3 grid(tiles, 1, (unsigned)zdim), block(256);
    const size_t lds = 4 * TILE_BYTES;
    hipStream_t s = (hipStream_t)stream;
    if (f32ab) {
        SclProfScope prof(SCL_KID_GEMM_F32, s, 2.0 * d.M * d.N * (double)d.K * d.nb1 * d.nb2, true);
        const int rc = scl_gemm_f32_launch(d, k, s);
        if (rc != SCL_OK) return rc;
        return scl_check_launch("scl_gemm_bf16(f32 operands)");
    }
    {
        SclProfScope prof(SCL_KID_GEMM, s, 2.0 * d.M * d.N * (double)d.K * d.nb1 * d.nb2, true);
        // LDS-DMA staging cannot mask a partially valid 16-byte vector: use it only when none can occur
        const bool a_whole = at ? (d.M % 8 == 0) : (d.K % 8 == 0);
        const bool b_whole = bt ? (d.N % 8 == 0) : (d.K % 8 == 0);
        const bool dma = a_whole && b_whole && !(d.flags & SCL_GEMM_NO_DMA);
        // 256x128 tiles, one 8-wave block per CU, 3-stage ring: opt-in (SCL_GEMM_FORCE_BIG).  A/B in one process on MI355X: equal
        // to the 128x128 kernel on the conv-stack shapes (+-2 %), 2.5 % slower end to end at batch 64 where it used to be picked
        // for the encoder linears as well; bit-identical results either way.
#ifdef SCL_EXPERIMENTS
        const long long big_tiles = (long long)((d.M + BIG_BM - 1) / BIG_BM) * ((d.N + BN - 1) / BN) * zdim;
        const bool big = dma && (d.flags & SCL_GEMM_FORCE_BIG) && !(d.flags & SCL_GEMM_NO_BIG) && d.M >= 256 && d.K >= 192 && big_tiles >= 1;
        // 256x256 ping-pong tiles: opt-in (SCL_GEMM_FORCE_P8).  Measured on MI355X (tools/gemm_square.py, tools/gemm_p8_ab.py):
        // 1093 / 1308 TFLOP/s at 4096^3 / 8192^3 (128x128 kernel: 811 / 1041), but on the encoder's M = 6368, K = 1024 shapes
        // 400 tiles = 1.56 rounds of 256 CUs and a 16-step K loop leave it behind the 128x128 kernel (fc1 fwd 576 vs 694).
        const long long p8_tiles = (long long)((d.M + P8_BM - 1) / P8_BM) * ((d.N + P8_BN - 1) / P8_BN) * zdim;
        const bool p8 = dma && (d.flags & SCL_GEMM_FORCE_P8) && !(d.flags & SCL_GEMM_NO_P8);
#else
        const bool big = false, p8 = false;
#endif
        // wide tiles with a runtime row pitch (gemm_w8.hip): whole rounds of the 256 CUs at M = 64 x 199 rows; picked when the
        // problem fills at least half a round of them and the operands advance linearly along K
        W8Plan plan;
#ifdef SCL_EXPERIMENTS
        const bool x2 = dma && gemm_pick_x2(k, at, bt, d, zdim, &plan);
#else
        const bool x2 = false;
#endif
        const bool w8 = !x2 && dma && gemm_pick_w8(k, at, bt, d, zdim, &plan);
        if (((unsigned)d.flags & SCL_GEMM_C_SPLIT3) && !w8) {
            scl_set_error("gemm: C_SPLIT3 is served by the wide-tile kernel only, and this launch (M %d, N %d, K %d) does not qualify for it", d.M, d.N, d.K);
            return SCL_EUNSUPPORTED;
        }
#ifdef SCL_EXPERIMENTS
        // one workgroup per CU at most and a K loop long enough to fill the ring: the deep-ring variant of the 128 x 128 kernel (opt-in)
        const char* deep_env = getenv("SCL_GEMM_DEEP");
        const bool deep = !x2 && !w8 && dma && deep_env && atoi(deep_env) != 0 && (long long)tiles * zdim <= 256 && (d.K + BK - 1) / BK / d.splitk >= 4;
#else
        const bool deep = false;
#endif
        prof.note(d.M, d.N, d.K, d.flags, (int)zdim, x2 ? 3 : (w8 ? (plan.variant == 2 ? 6 : 1 + plan.variant) : (deep ? 7 : 0)));
        if (x2) {
#ifdef SCL_EXPERIMENTS
            scl_gemm_x2_launch(k, at, bt, plan, zdim, s);
#endif
        } else if (w8) {
            scl_gemm_w8_launch(k, at, bt, plan, zdim, s);
#ifdef SCL_EXPERIMENTS
        } else if (p8) {
            static bool p8_attr_set = false;
            if (!p8_attr_set) {
                hipFuncSetAttribute((const void*)scl_gemm_p8_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS);
                hipFuncSetAttribute((const void*)scl_gemm_p8_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS);
                hipFuncSetAttribute((const void*)scl_gemm_p8_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS);
                hipFuncSetAttribute((const void*)scl_gemm_p8_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS);
                p8_attr_set = true;
            }
            dim3 pgrid((unsigned)(p8_tiles / zdim), 1, (unsigned)zdim), pblock(512);
            if (!at && !bt) SCL_LAUNCH((scl_gemm_p8_kernel<false, false>), pgrid, pblock, P8_LDS, s, k);
            else if (!at && bt) SCL_LAUNCH((scl_gemm_p8_kernel<false, true>), pgrid, pblock, P8_LDS, s, k);
            else if (at && !bt) SCL_LAUNCH((scl_gemm_p8_kernel<true, false>), pgrid, pblock, P8_LDS, s, k);
            else SCL_LAUNCH((scl_gemm_p8_kernel<true, true>), pgrid, pblock, P8_LDS, s, k);
        } else if (big) {
            static bool attr_set = false;
            if (!attr_set) {
                hipFuncSetAttribute((const void*)scl_gemm_big_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS);
                hipFuncSetAttribute((const void*)scl_gemm_big_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS);
                hipFuncSetAttribute((const void*)scl_gemm_big_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS);
                hipFuncSetAttribute((const void*)scl_gemm_big_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS);
                attr_set = true;
            }
            dim3 bgrid(((d.M + BIG_BM - 1) / BIG_BM) * ((d.N + BN - 1) / BN), 1, (unsigned)zdim), bblock(512);
            if (!at && !bt) SCL_LAUNCH((scl_gemm_big_kernel<false, false>), bgrid, bblock, BIG_LDS, s, k);
            else if (!at && bt) SCL_LAUNCH((scl_gemm_big_kernel<false, true>), bgrid, bblock, BIG_LDS, s, k);
            else if (at && !bt) SCL_LAUNCH((scl_gemm_big_kernel<true, false>), bgrid, bblock, BIG_LDS, s, k);
            else SCL_LAUNCH((scl_gemm_big_kernel<true, true>), bgrid, bblock, BIG_LDS, s, k);
#endif
#ifdef SCL_EXPERIMENTS
        } else if (deep) {
            static bool deep_attr_set = false;
            if (!deep_attr_set) {
                hipFuncSetAttribute((const void*)scl_gemm_deep_kernel<false, false, DEEP_S>, hipFuncAttributeMaxDynamicSharedMemorySize, DEEP_LDS);
                hipFuncSetAttribute((const void*)scl_gemm_deep_kernel<false, true, DEEP_S>, hipFuncAttributeMaxDynamicSharedMemorySize, DEEP_LDS);
                hipFuncSetAttribute((const void*)scl_gemm_deep_kernel<true, false, DEEP_S>, hipFuncAttributeMaxDynamicSharedMemorySize, DEEP_LDS);
                hipFuncSetAttribute((const void*)scl_gemm_deep_kernel<true, true, DEEP_S>, hipFuncAttributeMaxDynamicSharedMemorySize, DEEP_LDS);
                deep_attr_set = true;
            }
            if (!at && !bt) SCL_LAUNCH((scl_gemm_deep_kernel<false, false, DEEP_S>), grid, block, DEEP_LDS, s, k);
            else if (!at && bt) SCL_LAUNCH((scl_gemm_deep_kernel<false, true, DEEP_S>), grid, block, DEEP_LDS, s, k);
            else if (at && !bt) SCL_LAUNCH((scl_gemm_deep_kernel<true, false, DEEP_S>), grid, block, DEEP_LDS, s, k);
            else SCL_LAUNCH((scl_gemm_deep_kernel<true, true, DEEP_S>), grid, block, DEEP_LDS, s, k);
#endif
        } else if (dma) {
            if (!at && !bt) SCL_LAUNCH((scl_gemm_dma_kernel<false, false>), grid, block, lds, s, k);
            else if (!at && bt) SCL_LAUNCH((scl_gemm_dma_kernel<false, true>), grid, block, lds, s, k);
            else if (at && !bt) SCL_LAUNCH((scl_gemm_dma_kernel<true, false>), grid, block, lds, s, k);
            else SCL_LAUNCH((scl_gemm_dma_kernel<true, true>), grid, block, lds, s, k);
        } else {
            if (!at && !bt) SCL_LAUNCH((scl_gemm_kernel<false, false>), grid, block, lds, s, k);
            else if (!at && bt) SCL_LAUNCH((scl_gemm_kernel<false, true>), grid, block, lds, s, k);
            else if (at && !bt) SCL_LAUNCH((scl_gemm_kernel<true, false>), grid, block, lds, s, k);
            else SCL_LAUNCH((scl_gemm_kernel<true, true>), grid, block, lds, s, k);
        }
    }
    return scl_check_launch("scl_gemm_bf16");
}

extern "C" long long scl_debug_gemm_persistent_launches(void) { return scl_gemm_w8p_launches(); }
extern "C" int scl_build_flags(void) {
#ifdef SCL_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

extern "C" int scl_debug_gemm_stamps(unsigned long long* out, int nblocks) {
    SCL_REQUIRE(out && nblocks > 0 && nblocks <= 4096, "gemm stamps: bad args");
    return scl_gemm_read_stamps(out, nblocks);
}

// ---- split-K finish: C = epilogue(sum of the f32 slabs) ------------------------------------------------------------------------------
// A GEMM with few output tiles and a long reduction (the N = 1024 linears of a pack-sized step, M = 11 x 199 rows, K = 3072 / 4096:
// 144 tiles of 128x128, one 4-wave block on 144 of the 256 CUs walking 48-64 K steps at HBM latency) runs faster as `nslabs` partial
// GEMMs into f32 slabs followed by this pass, which applies the ORIGINAL descriptor's epilogue (bias, second output, activation,
// residual / activation-gradient, dropout, bf16 or f32 store) to the sum — the same arithmetic, in the same order, as the in-kernel
// epilogue applies to its accumulator.  One thread per 4 columns of a row.
__global__ __launch_bounds__(256) void scl_gemm_finish_kernel(const GemmK d, const float* __restrict__ slabs, int nslabs, long long stride) {
    const int ncg = (d.N + 3) >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)d.M * ncg) return;
    const int row = (int)(idx / ncg), col = (int)(idx - (long long)row * ncg) * 4;
    const int flags = d.flags;
    const float* p = slabs + (long long)row * d.N + col;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const bool full = col + 4 <= d.N && !(d.N & 3) && !(stride & 3);
    for (int s = 0; s < nslabs; ++s, p += stride) {
        if (full) {
            const float4 t = *reinterpret_cast<const float4*>(p);
            v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
        } else {
            for (int i = 0; i < 4; ++i) if (col + i < d.N) v[i] += p[i];
        }
    }
    const unsigned q = udiv_magic((unsigned)row, d.c_magic, d.c_shift);
    const long long off = (long long)q * d.c_rbstride + (long long)((unsigned)row - q * d.c_rpb) * d.ldc + col;
    const float* bias = (flags & SCL_GEMM_HAS_BIAS) ? d.bias : nullptr;
    if (full && d.vec_ok) {
        const bool c_f32 = flags & SCL_GEMM_C_F32, c2_f32 = flags & SCL_GEMM_C2_F32, r_f32 = flags & SCL_GEMM_R_F32;
        const int act = (flags >> SCL_GEMM_ACT_SHIFT) & 0xF, rmode = (flags >> SCL_GEMM_RMODE_SHIFT) & 0xF, ract = (flags >> SCL_GEMM_RACT_SHIFT) & 0xF;
        if (bias) {
            const float4 bb = *reinterpret_cast<const float4*>(bias + col);
            v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
        }
        float c2v[4] = {v[0], v[1], v[2], v[3]};
        if (act == 5) { gelu_both_f(v[0], v[0], c2v[0]); gelu_both_f(v[1], v[1], c2v[1]); gelu_both_f(v[2], v[2], c2v[2]); gelu_both_f(v[3], v[3], c2v[3]); }
        if (flags & SCL_GEMM_HAS_C2) {
            if (c2_f32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(d.C2) + off) = make_float4(c2v[0], c2v[1], c2v[2], c2v[3]);
            else *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(d.C2) + off) = make_uint2(pack_bf2(c2v[0], c2v[1]), pack_bf2(c2v[2], c2v[3]));
        }
        if (act && act != 5) { v[0] = act_f(act, v[0]); v[1] = act_f(act, v[1]); v[2] = act_f(act, v[2]); v[3] = act_f(act, v[3]); }
        float r[4] = {0.f, 0.f, 0.f, 0.f};
        if (rmode) {
            if (r_f32) {
                const float4 t = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(d.R) + off);
                r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
            } else {
                const uint2 t = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(d.R) + off);
                r[0] = __uint_as_float(t.x << 16); r[1] = __uint_as_float(t.x & 0xFFFF0000u);
                r[2] = __uint_as_float(t.y << 16); r[3] = __uint_as_float(t.y & 0xFFFF0000u);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (rmode == 2) v[i] *= act_grad_f(ract, r[i]);
            if (flags & SCL_GEMM_DROPOUT) v[i] *= dropout_scale(d.drop_seed, (uint64_t)(off + i), d.drop_p);
            if (rmode == 1) v[i] += r[i];
        }
        if (c_f32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(d.C) + off) = make_float4(v[0], v[1], v[2], v[3]);
        else *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(d.C) + off) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
    } else {
        const EpiArgs ea = {d.C, d.C2, d.R, d.N, d.flags, d.drop_seed, d.drop_p};
        for (int i = 0; i < 4; ++i) epi_scalar(ea, v[i], off + i, col + i, bias);
    }
}

extern "C" int scl_gemm_splitk_finish(const SclGemmDesc* dp, const float* slabs, int nslabs, int64_t stride, void* stream) {
    SCL_REQUIRE(dp && slabs && nslabs >= 1 && stride >= 0, "gemm finish: bad args");
    const SclGemmDesc& d = *dp;
    SCL_REQUIRE(d.M > 0 && d.N > 0 && d.nb1 == 1 && d.nb2 == 1, "gemm finish: one un-batched [M, N] problem");
    SCL_REQUIRE(d.C && d.c_rpb >= 1, "gemm finish: C null or c_rpb < 1");
    const int rmode = (d.flags >> SCL_GEMM_RMODE_SHIFT) & 0xF;
    SCL_REQUIRE(rmode == 0 || d.R, "gemm finish: RMODE set but R is null");
    SCL_REQUIRE(!(d.flags & SCL_GEMM_HAS_C2) || d.C2, "gemm finish: HAS_C2 set but C2 is null");
    SCL_REQUIRE(!(d.flags & SCL_GEMM_HAS_BIAS) || d.bias, "gemm finish: HAS_BIAS set but bias is null");
    SCL_REQUIRE(((uintptr_t)slabs & 15) == 0, "gemm finish: slabs must be 16-byte aligned");
    GemmK k;
    k.C = d.C; k.C2 = d.C2; k.R = d.R; k.bias = d.bias; k.colsum = nullptr;
    k.c_bs1 = 0; k.c_bs2 = 0; k.c_rbstride = d.c_rbstride; k.c_split_stride = 0; k.bias_bs2 = 0;
    k.c_rpb = (unsigned)d.c_rpb; make_magic((unsigned)d.c_rpb, &k.c_magic, &k.c_shift);
    k.ldc = d.ldc; k.M = d.M; k.N = d.N; k.K = d.K; k.nb2 = 1; k.splitk = 1; k.flags = d.flags;
    k.alpha = 1.0f; k.drop_p = d.drop_p; k.drop_seed = d.drop_seed; k.group_m = 8; k.tile_m = 0; k.debug = 0;
    auto al = [](const void* p, int bytes) { return p == nullptr || ((uintptr_t)p & (bytes - 1)) == 0; };
    k.vec_ok = !(d.ldc & 3) && !(d.c_rbstride & 3) && al(d.C, (d.flags & SCL_GEMM_C_F32) ? 16 : 8) && al(d.C2, (d.flags & SCL_GEMM_C2_F32) ? 16 : 8) &&
               al(d.R, (d.flags & SCL_GEMM_R_F32) ? 16 : 8) && al(d.bias, 16);
    const long long n = (long long)d.M * ((d.N + 3) / 4);
    hipLaunchKernelGGL(scl_gemm_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, k, slabs, nslabs, (long long)stride);
    return scl_check_launch("scl_gemm_splitk_finish");
}

extern "C" int scl_reduce_slabs_f32(const float* slabs, float* out, int64_t n, int nslabs, int64_t stride, void* stream) {
    SCL_REQUIRE(slabs && out && n > 0 && nslabs >= 1, "reduce_slabs: bad args");
    SCL_REQUIRE(((uintptr_t)slabs & 15) == 0 && ((uintptr_t)out & 15) == 0 && (stride & 3) == 0, "reduce_slabs: alignment");
    int blocks = (int)((n / 4 + 255) / 256); if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(scl_reduce_slabs_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, slabs, out, n, nslabs, stride);
    return scl_check_launch("scl_reduce_slabs_f32");
}
