// gemm_f32.hip — the fp32 member of the GEMM family: C = epilogue(alpha * A * B^T) with f32 operands on the f32-input matrix cores
// (v_mfma_f32_16x16x4_f32: exact fp32 products and fp32 accumulation — a k-ordered fmaf chain, no reduced-precision pass).
//
// Two users, both of which need fp32 end to end where the bf16 kernels cannot give it:
//   * the scoring path (main.py --eval / --predict / --emb; reference main.py:161-214 runs fp32, north_star asks for scores within
//     1e-3 of it): every encoder / head contraction with f32 activations and the f32 master weights;
//   * the AASIST / ResNet back-ends (model/wav2vec2_aasist.py:377-604, model/resnet.py:47-191): their 2-D convolutions as implicit
//     GEMMs over channels-last maps (rows = output positions with overlapping windows, K = (kh; kw*C + c) through the operand's
//     2-level contiguous index, one batch entry per utterance) and their small linears, forward / dgrad / wgrad.
// Same descriptor, operand addressing (rpb / rbstride / ld / cin / cout / batch strides), split-K and fused epilogue as scl_gemm_bf16.
//
// Design: 128x128 (or 64x64 for small problems) tile per 256-thread block, 4 waves as 2x2, BK = 32, LDS double buffer, register-staged
// 16-byte global loads.  The K order inside a 32-deep step is permuted so that a lane's 8 operands of one row are contiguous in LDS
// (lane group q = lane>>4 owns k = 8q .. 8q+7; MFMA j of the step multiplies k = {j, 8+j, 16+j, 24+j}): two ds_read_b128 per
// fragment instead of eight ds_read_b32.  Transposed operands are staged as they lie in memory ([k][128 contiguous]) and read
// element-wise.  MFMA operands are swapped (D = B-frag x A-frag) so the epilogue of gemm_common.h applies unchanged.
#include "gemm_common.h"

using namespace sclg;

namespace {

constexpr int FBK = 32;
constexpr int LDK = 36;      // words per row of a K-contiguous image ([rows][32 k] + 4 pad: 16-byte aligned rows, spread banks)
constexpr int LDT = 132;     // words per k-row of a transposed image ([32 k][128 contiguous] + 4 pad), 68 for the 64-wide tile

constexpr int PK = 32;       // bf16 per row of a pair-form image ([rows][32 k], 64-byte rows, no padding: the four 16-byte chunks of a row are XOR-swizzled with
                             // (row >> 2) & 3, so the 16 rows of a fragment read hit 16 distinct bank quads).  Padding to 80-byte rows instead made a 128-row block exactly 80 KiB (half the CU's LDS);
                             // same speed either way (352 vs 355 us on the QKV-shaped scoring GEMM), the swizzle keeps 32 KiB per CU free

// four f32 -> four (hi, lo) bf16 pairs: hi = bf16(a) (round to nearest even), lo = bf16(a - hi)
__device__ __forceinline__ void f32_split4(const u32x4& raw, uint2& hi, uint2& lo) {
    typedef __attribute__((ext_vector_type(4))) float f4;
    typedef __attribute__((ext_vector_type(4))) __bf16 b4;
    const f4 x = __builtin_bit_cast(f4, raw);
    const b4 h = __builtin_convertvector(x, b4);
    const f4 back = __builtin_convertvector(h, f4);
    const b4 l = __builtin_convertvector(x - back, b4);
    hi = __builtin_bit_cast(uint2, h);
    lo = __builtin_bit_cast(uint2, l);
}

template <int TM> struct F32Geom {
    static constexpr int ROWS = TM;                       // tile rows (= columns)
    static constexpr int WT = TM / 2;                     // per-wave rows
    static constexpr int NB = WT / 16;                    // 16-row blocks per wave (4 or 2)
    static constexpr int LDT_ = TM + 4;
    static constexpr int IMG = (TM * LDK > FBK * (TM + 4) ? TM * LDK : FBK * (TM + 4));   // words per operand image
};

// ---- staging: global -> registers -> LDS -----------------------------------------------------------------------------------------
template <int TM, bool T> struct F32Stage;

template <int TM> struct F32Stage<TM, false> {      // K-contiguous: TM rows x 32 k; thread loads NV float4 (row = t/8 + 32 i, k chunk t%8)
    static constexpr int NV = TM / 32;
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned rowoff[NV];
    int kcur, kend;
    __device__ __forceinline__ void init(const OpK& o, const char* base, int row0, int rowlimit, int kbegin, int kend_, int tid) {
        rsrc = make_rsrc(base);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int r = row0 + (tid >> 3) + 32 * i;
            rowoff[i] = r < rowlimit ? row_off(o, (unsigned)r) : OOB;
        }
        kcur = kbegin + 4 * (tid & 7); kend = kend_;
    }
    // The fetches of a step go out back to back; what a partial vector at the K tail needs (K % 4 != 0 is legal for f32 operands whose ld
    // keeps rows 16-byte aligned) is done by fix() when the registers are consumed.  With the zeroing inside this loop every load sat
    // between two exec-masked branches and the compiler put a vmcnt wait at each join: half a step's fetches waited for the other half.
    __device__ __forceinline__ int load(const OpK& o, u32x4 (&r)[NV]) const {      // returns the k still valid at this thread's vector (>= 4: whole)
        const unsigned koff = col_off(o, (unsigned)kcur);
        const int nv_ld = kend - kcur;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const unsigned off = (nv_ld > 0 && rowoff[i] != OOB) ? rowoff[i] + koff : OOB;
            r[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        }
        return nv_ld;
    }
    __device__ __forceinline__ void fix(u32x4 (&r)[NV], int nv_ld) const {
        if (nv_ld < 4) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) if (e >= nv_ld) r[i][e] = 0u;
        }
    }
    __device__ __forceinline__ void store(float* img, const u32x4 (&r)[NV], int tid) const {
        const int c = tid & 7;                       // logical k chunk (4 k); lane group q = c>>1 owns k 8q..8q+7
#pragma unroll
        for (int i = 0; i < NV; ++i) *reinterpret_cast<u32x4*>(img + ((tid >> 3) + 32 * i) * LDK + 4 * c) = r[i];
    }
    // pair form: the element as hi = bf16(a) and lo = bf16(a - hi), each into its own [row][32 k] bf16 image (pitch PK)
    __device__ __forceinline__ void store_pair(char* hi_img, char* lo_img, const u32x4 (&r)[NV], int tid) const {
        const int c = tid & 7;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            uint2 h, l;
            f32_split4(r[i], h, l);
            const int row = (tid >> 3) + 32 * i;
            const int off = row * (PK * 2) + ((((c >> 1) ^ (row >> 2)) & 3) << 4) + 8 * (c & 1);
            *reinterpret_cast<uint2*>(hi_img + off) = h;
            *reinterpret_cast<uint2*>(lo_img + off) = l;
        }
    }
    __device__ __forceinline__ void advance() { kcur += FBK; }
};

template <int TM> struct F32Stage<TM, true> {       // transposed: 32 k rows x TM contiguous; thread loads NV float4 (k = t/(TM/4) + step i)
    static constexpr int CPR = TM / 4;               // float4 chunks per k row
    static constexpr int KSTEP = 256 / CPR;          // k rows covered per pass (8 for TM = 128, 16 for 64)
    static constexpr int NV = FBK / KSTEP;
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned coloff;
    int kcur, kend, ncolvalid;
    __device__ __forceinline__ void init(const OpK& o, const char* base, int col0, int collimit, int kbegin, int kend_, int tid) {
        rsrc = make_rsrc(base);
        const int col = col0 + 4 * (tid % CPR);
        int nv = collimit - col; nv = nv < 0 ? 0 : (nv > 4 ? 4 : nv);
        ncolvalid = nv;
        coloff = nv > 0 ? col_off(o, (unsigned)col) : OOB;
        kcur = kbegin + tid / CPR; kend = kend_;
    }
    __device__ __forceinline__ int load(const OpK& o, u32x4 (&r)[NV]) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int rr = kcur + KSTEP * i;
            const unsigned off = (rr < kend && coloff != OOB) ? row_off(o, (unsigned)rr) + coloff : OOB;
            r[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        }
        return 4;
    }
    __device__ __forceinline__ void fix(u32x4 (&r)[NV], int) const {      // columns past the operand's edge (a partial 4-column vector)
        if (ncolvalid < 4) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) if (e >= ncolvalid) r[i][e] = 0u;
        }
    }
    __device__ __forceinline__ void store(float* img, const u32x4 (&r)[NV], int tid) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) *reinterpret_cast<u32x4*>(img + (tid / CPR + KSTEP * i) * (TM + 4) + 4 * (tid % CPR)) = r[i];
    }
    __device__ __forceinline__ void advance() { kcur += FBK; }
};

// fragment of one 16-row block for the whole 32-deep step: v[j] = operand[row = blk*16 + (lane&15)][k = 8*(lane>>4) + j]
template <int TM, bool T>
__device__ __forceinline__ void f32_frag(const float* img, int blk, int lane, float (&v)[8]) {
    const int row = blk * 16 + (lane & 15), q = lane >> 4;
    if (!T) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(img + row * LDK + 8 * q);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(img + row * LDK + 8 * q + 4);
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = img[(8 * q + j) * (TM + 4) + row];
    }
}

// ---- "f32 x3": fp32 operands as bf16 pairs on the bf16 matrix cores ------------------------------------------------------------------
// a = a_hi + a_lo + O(2^-17 |a|) with a_hi = bf16(a), a_lo = bf16(a - a_hi).  a b = a_hi b_hi + a_hi b_lo + a_lo b_hi + O(2^-16 |a b|): three
// v_mfma_f32_16x16x32_bf16 (fp32 accumulation) replace the eight v_mfma_f32_16x16x4_f32 of a 32-deep step — 48 instead of 256 matrix-core
// cycles per 16 x 16 block, i.e. the 2.5 PFLOP/s pipe at a third of its rate (~830 TFLOP/s) against the 157 TFLOP/s of the f32 pipe — for
// products accurate to ~2e-5 relative (random signs: ~1e-6 of a long dot product) instead of 6e-8.  Same staging, same LDS images, same
// fragments (a lane's eight k of one row ARE the bf16 instruction's operand layout), same epilogue; the split runs on the VALU per step.
// Opt-in per launch (SCL_GEMM_F32X3): the back-end convolutions / linears and the scoring path ask for it, the exact kernel stays the
// meaning of SCL_GEMM_AB_F32 alone.
__device__ __forceinline__ void f32_split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
    typedef __attribute__((ext_vector_type(8))) float f32x8;
    f32x8 x;
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = v[j];
    hi = __builtin_convertvector(x, bf16x8);
    const f32x8 back = __builtin_convertvector(hi, f32x8);
    lo = __builtin_convertvector(x - back, bf16x8);
}

// the 64x64 tile needs 60-90 VGPRs and 36 KiB of LDS: four blocks per CU hide its single-stage global prefetch (measured on the AASIST
// convolution wgrads: 192 blocks at one per CU ran at the HBM latency, 1.6 us per 32-deep step)
template <int TM, bool AT, bool BT, bool X3>
__global__ __launch_bounds__(256, (TM == 64 ? 4 : 2)) void scl_gemm_f32_kernel(const GemmK d) {
    typedef F32Geom<TM> G;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* smem = reinterpret_cast<float*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_n = (d.N + TM - 1) / TM;
    int tm, tn;
    tile_coords(blockIdx.x, gridDim.x, (d.M + TM - 1) / TM, tiles_n, tm, tn, d.group_m);
    const int m0 = tm * TM, n0 = tn * TM;
    int z = blockIdx.z;
    const int ksplit = __builtin_amdgcn_readfirstlane(z % d.splitk); z /= d.splitk;      // uniform, but integer division runs on the vector ALU: back to an SGPR
    const int z1 = __builtin_amdgcn_readfirstlane(z / d.nb2), z2 = z - z1 * d.nb2;
    const int nk_total = (d.K + FBK - 1) / FBK;
    const int nk_per = __builtin_amdgcn_readfirstlane((nk_total + d.splitk - 1) / d.splitk);
    const int kbegin = ksplit * nk_per * FBK;
    int kend = kbegin + nk_per * FBK; if (kend > d.K) kend = d.K;
    const int nk = kend > kbegin ? (kend - kbegin + FBK - 1) / FBK : 0;
    const char* Ab = reinterpret_cast<const char*>(d.A.ptr) + z1 * d.A.bs1 + z2 * d.A.bs2;
    const char* Bb = reinterpret_cast<const char*>(d.B.ptr) + z1 * d.B.bs1 + z2 * d.B.bs2;

    F32Stage<TM, AT> sa;
    F32Stage<TM, BT> sb;
    sa.init(d.A, Ab, m0, d.M, kbegin, kend, tid);
    sb.init(d.B, Bb, n0, d.N, kbegin, kend, tid);

    f32x4 acc[G::NB][G::NB];
#pragma unroll
    for (int i = 0; i < G::NB; ++i)
#pragma unroll
        for (int j = 0; j < G::NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 ra[F32Stage<TM, AT>::NV], rb[F32Stage<TM, BT>::NV];
    {   // no branch around a fetch or around the store that consumes it (see scl_gemm_f32p_kernel): past the last step the offsets are out of
        // range, the loads return zeros without traffic and the images written are never read
        const int na = sa.load(d.A, ra), nb = sb.load(d.B, rb);
        sa.fix(ra, na); sb.fix(rb, nb);
        sa.store(smem, ra, tid); sb.store(smem + G::IMG, rb, tid);
    }
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        sa.advance(); sb.advance();
        const int na = sa.load(d.A, ra), nb = sb.load(d.B, rb);
        const float* tA = smem + cur * 2 * G::IMG;
        const float* tB = tA + G::IMG;
        float fa[G::NB][8], fb[G::NB][8];
#pragma unroll
        for (int i = 0; i < G::NB; ++i) {
            f32_frag<TM, AT>(tA, wr * G::NB + i, lane, fa[i]);
            f32_frag<TM, BT>(tB, wc * G::NB + i, lane, fb[i]);
        }
        if constexpr (X3) {
            bf16x8 ah[G::NB], al[G::NB], bh[G::NB], bl[G::NB];
#pragma unroll
            for (int i = 0; i < G::NB; ++i) { f32_split8(fa[i], ah[i], al[i]); f32_split8(fb[i], bh[i], bl[i]); }
#pragma unroll
            for (int i = 0; i < G::NB; ++i)
#pragma unroll
                for (int n = 0; n < G::NB; ++n) {      // the two small terms first, the leading term last
                    acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[n], ah[i], acc[i][n], 0, 0, 0);
                    acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[n], al[i], acc[i][n], 0, 0, 0);
                    acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[n], ah[i], acc[i][n], 0, 0, 0);
                }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int i = 0; i < G::NB; ++i)
#pragma unroll
                    for (int n = 0; n < G::NB; ++n)
                        acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[n][j], fa[i][j], acc[i][n], 0, 0, 0);
        }
        {
            float* nA = smem + (cur ^ 1) * 2 * G::IMG;
            sa.fix(ra, na); sb.fix(rb, nb);
            sa.store(nA, ra, tid); sb.store(nA + G::IMG, rb, tid);
        }
        __syncthreads();
        cur ^= 1;
    }
    if constexpr (G::NB == 4) {
        f32x4 (&a4)[4][4] = *reinterpret_cast<f32x4 (*)[4][4]>(&acc[0][0]);
        gemm_epilogue_blk<4>(d, a4, m0 + wr * 64, n0 + wc * 64, d.M, 4, z1, z2, ksplit, lane);
    } else {
        // 32 x 32 per wave: the shared epilogue walks 4 column blocks per row block — hand it 2 x 2 padded to [2][4] with the unused
        // columns masked by an N limit (nt >= 2 columns fall outside [nbase, nbase + 32))
        f32x4 a24[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) { a24[i][0] = acc[i][0]; a24[i][1] = acc[i][1]; a24[i][2] = f32x4{0.f, 0.f, 0.f, 0.f}; a24[i][3] = a24[i][2]; }
        GemmK dd = d;
        const int nlim = n0 + wc * 32 + 32;
        dd.N = nlim < d.N ? nlim : d.N;
        gemm_epilogue_blk<2>(dd, a24, m0 + wr * 32, n0 + wc * 32, d.M, 2, z1, z2, ksplit, lane);
    }
}

// ---- pair form with the split at STAGING time (round 5, second step) ---------------------------------------------------------------------
// The X3 path above converts every fragment in every wave (each element twice, ~190 VALU instructions per 32-deep step and wave) and reads
// f32 from LDS.  Here a thread converts the float4 it just fetched ONCE and stores hi and lo into two bf16 images; a fragment is one
// ds_read_b128 per image and the loop body is 3 NB^2 MFMAs + 4 NB reads.  Same tiles, same global staging, same epilogue.
// Measured (tools/f32x3_probe.py, MI355X): scoring GEMMs 449-605 -> 355-500 us (216-235 TFLOP/s of f32-equivalent work), ResNet data-gradient
// convolutions 171-226 -> 150-172 us.  A transposed operand would need 2-byte scatter stores into the images (95.8 vs 68.2 us on the
// ResNet TT weight gradient) and keeps the fragment-split form.
// Tried and dropped in the same round (QKV-shaped scoring GEMM 12864 x 3072 x 1024 / ResNet 64->64 conv at 32 utterances):
//  * a THREE-part form (a = p0 + p1 + p2 carries all 24 mantissa bits; the six terms of weight >= 2^-16) as a replacement for the exact
//    v_mfma_f32_16x16x4_f32 kernel: same error against fp64 (1.0e-6 vs 1.3e-6) at 96 instead of 256 matrix-core cycles per 16x16x32
//    block, but 745 vs 759 us / 286 vs 254 us — three images per operand are 120 KB of LDS (one block per CU) and 1.5x the LDS traffic;
//  * a second register set (fetch distance 2) on the loop as first written: no change (354 vs 355 us) — the fetch and the conversion sat behind
//    step-index branches and the compiler's vmcnt placement drained the queue anyway; the branch-free form below is what runs now.
//  Ablations on the QKV shape: full 354 us; without the global fetches 257; without the MFMAs 237; without the split + image stores 297 —
//  no single resource bounds it: fetch issue, fragment reads, MFMAs and image stores run back to back inside a wave and two blocks
//  per CU overlap them only partly (LDS traffic ~= MFMA time at this tile shape).
template <int TM>
__global__ __launch_bounds__(256, (TM == 64 ? 4 : 2)) void scl_gemm_f32p_kernel(const GemmK d) {
    constexpr bool AT = false, BT = false;
    typedef F32Geom<TM> G;
    constexpr int IMGB = TM * PK * 2;      // bytes per image; a buffer = A hi, A lo, B hi, B lo
    extern __shared__ __attribute__((aligned(16))) char psm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_n = (d.N + TM - 1) / TM;
    int tm, tn;
    tile_coords(blockIdx.x, gridDim.x, (d.M + TM - 1) / TM, tiles_n, tm, tn, d.group_m);
    const int m0 = tm * TM, n0 = tn * TM;
    int z = blockIdx.z;
    const int ksplit = __builtin_amdgcn_readfirstlane(z % d.splitk); z /= d.splitk;
    const int z1 = __builtin_amdgcn_readfirstlane(z / d.nb2), z2 = z - z1 * d.nb2;
    const int nk_total = (d.K + FBK - 1) / FBK;
    const int nk_per = __builtin_amdgcn_readfirstlane((nk_total + d.splitk - 1) / d.splitk);
    const int kbegin = ksplit * nk_per * FBK;
    int kend = kbegin + nk_per * FBK; if (kend > d.K) kend = d.K;
    const int nk = kend > kbegin ? (kend - kbegin + FBK - 1) / FBK : 0;
    const char* Ab = reinterpret_cast<const char*>(d.A.ptr) + z1 * d.A.bs1 + z2 * d.A.bs2;
    const char* Bb = reinterpret_cast<const char*>(d.B.ptr) + z1 * d.B.bs1 + z2 * d.B.bs2;
    F32Stage<TM, AT> sa;
    F32Stage<TM, BT> sb;
    sa.init(d.A, Ab, m0, d.M, kbegin, kend, tid);
    sb.init(d.B, Bb, n0, d.N, kbegin, kend, tid);
    f32x4 acc[G::NB][G::NB];
#pragma unroll
    for (int i = 0; i < G::NB; ++i)
#pragma unroll
        for (int j = 0; j < G::NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Two register sets: the fetches of step kt + 2 go out before the MFMAs of step kt, and the set split + stored behind those MFMAs (step
    // kt + 1) landed a step ago — its conversion has no memory wait in front of it and can issue in the shadow of the matrix instructions.
    constexpr int NV = F32Stage<TM, false>::NV;
    u32x4 ra0[NV], rb0[NV], ra1[NV], rb1[NV];
    int na0 = 4, nb0 = 4, na1 = 4, nb1 = 4;
    // Nothing in the pipeline below is conditional on the step index: a fetch past the last step has every offset out of range (the buffer
    // loads return zeros without traffic), its images are zeros, and an odd step count runs one extra step on them that adds exact zeros.
    // Every branch around a fetch or around the conversion that consumes one makes the compiler's vmcnt placement assume the path that did
    // NOT wait — it then drains the whole fetch queue at the loop header.  Measured on one box against the one-register-set loop of the round's
    // first form (QKV-shaped scoring GEMM / ResNet 128 -> 128 data-gradient convolution, TFLOP/s of f32-equivalent work): 224 - 232 / 152 ->
    // 234 - 237 / 164 - 166; cutting the conversion into pieces between the MFMA row blocks: 226 / 157 (dropped).
    na0 = sa.load(d.A, ra0); nb0 = sb.load(d.B, rb0);
    sa.advance(); sb.advance(); na1 = sa.load(d.A, ra1); nb1 = sb.load(d.B, rb1);
    sa.fix(ra0, na0); sb.fix(rb0, nb0);
    sa.store_pair(psm, psm + IMGB, ra0, tid); sb.store_pair(psm + 2 * IMGB, psm + 3 * IMGB, rb0, tid);
    __syncthreads();
    const int frow = (lane & 15) * (PK * 2) + ((((lane >> 4) ^ (lane >> 2)) & 3) << 4);      // a lane's row and its (swizzled) chunk of 8 consecutive k inside a 16-row block
    // one step: LDS buffer `cur` holds step kt, (rn_a, rn_b) hold step kt + 1, (rf_a, rf_b) are free for step kt + 2
    auto step = [&](int cur, u32x4 (&rf_a)[NV], u32x4 (&rf_b)[NV], int& nf_a, int& nf_b, u32x4 (&rn_a)[NV], u32x4 (&rn_b)[NV], int nn_a, int nn_b) {
        // unconditional: past the last step every offset is out of range and the buffer loads return zeros without traffic.  Behind a branch the
        // compiler must place the vmcnt waits of the conversion below for the path that issued nothing, i.e. wait for THESE fetches too.
        sa.advance(); sb.advance(); nf_a = sa.load(d.A, rf_a); nf_b = sb.load(d.B, rf_b);
        __builtin_amdgcn_sched_barrier(0);      // the fetches stay FIRST: sunk below the fragment reads, their registers alias the fragments' and the loop header waits vmcnt(0)
        const char* t0 = psm + cur * 4 * IMGB;
        bf16x8 ah[G::NB], al[G::NB], bh[G::NB], bl[G::NB];
#pragma unroll
        for (int i = 0; i < G::NB; ++i) {
            const int ao = (wr * G::NB + i) * 16 * (PK * 2) + frow, bo = (wc * G::NB + i) * 16 * (PK * 2) + frow;
            ah[i] = *reinterpret_cast<const bf16x8*>(t0 + ao);
            al[i] = *reinterpret_cast<const bf16x8*>(t0 + IMGB + ao);
            bh[i] = *reinterpret_cast<const bf16x8*>(t0 + 2 * IMGB + bo);
            bl[i] = *reinterpret_cast<const bf16x8*>(t0 + 3 * IMGB + bo);
        }
#pragma unroll
        for (int i = 0; i < G::NB; ++i)
#pragma unroll
            for (int n = 0; n < G::NB; ++n) {
                acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[n], ah[i], acc[i][n], 0, 0, 0);
                acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[n], al[i], acc[i][n], 0, 0, 0);
                acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[n], ah[i], acc[i][n], 0, 0, 0);
            }
        char* nbuf = psm + (cur ^ 1) * 4 * IMGB;
        sa.fix(rn_a, nn_a); sb.fix(rn_b, nn_b);
        sa.store_pair(nbuf, nbuf + IMGB, rn_a, tid); sb.store_pair(nbuf + 2 * IMGB, nbuf + 3 * IMGB, rn_b, tid);
        __syncthreads();
    };
    for (int kt = 0; kt < nk; kt += 2) {
        step(0, ra0, rb0, na0, nb0, ra1, rb1, na1, nb1);
        step(1, ra1, rb1, na1, nb1, ra0, rb0, na0, nb0);
    }
    if constexpr (G::NB == 4) {
        f32x4 (&a4)[4][4] = *reinterpret_cast<f32x4 (*)[4][4]>(&acc[0][0]);
        gemm_epilogue_blk<4>(d, a4, m0 + wr * 64, n0 + wc * 64, d.M, 4, z1, z2, ksplit, lane);
    } else {
        f32x4 a24[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) { a24[i][0] = acc[i][0]; a24[i][1] = acc[i][1]; a24[i][2] = f32x4{0.f, 0.f, 0.f, 0.f}; a24[i][3] = a24[i][2]; }
        GemmK dd = d;
        const int nlim = n0 + wc * 32 + 32;
        dd.N = nlim < d.N ? nlim : d.N;
        gemm_epilogue_blk<2>(dd, a24, m0 + wr * 32, n0 + wc * 32, d.M, 2, z1, z2, ksplit, lane);
    }
}

template <int TM>
void f32p_launch(const GemmK& k, dim3 grid, hipStream_t s) {
    const size_t lds = (size_t)2 * 4 * TM * PK * 2;
    SCL_LAUNCH((scl_gemm_f32p_kernel<TM>), grid, dim3(256), lds, s, k);
}

template <int TM, bool X3>
void f32_launch(const GemmK& k, bool at, bool bt, dim3 grid, hipStream_t s) {
    const size_t lds = 4 * (size_t)F32Geom<TM>::IMG * sizeof(float);
    static bool attr_set = false;
    if (!attr_set && lds > 65536) {
        (void)hipFuncSetAttribute((const void*)scl_gemm_f32_kernel<TM, false, false, X3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)scl_gemm_f32_kernel<TM, false, true, X3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)scl_gemm_f32_kernel<TM, true, false, X3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)scl_gemm_f32_kernel<TM, true, true, X3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const dim3 block(256);
    if (!at && !bt) SCL_LAUNCH((scl_gemm_f32_kernel<TM, false, false, X3>), grid, block, lds, s, k);
    else if (!at && bt) SCL_LAUNCH((scl_gemm_f32_kernel<TM, false, true, X3>), grid, block, lds, s, k);
    else if (at && !bt) SCL_LAUNCH((scl_gemm_f32_kernel<TM, true, false, X3>), grid, block, lds, s, k);
    else SCL_LAUNCH((scl_gemm_f32_kernel<TM, true, true, X3>), grid, block, lds, s, k);
}

}  // namespace

namespace sclg {

int scl_gemm_f32_launch(const SclGemmDesc& d, GemmK& k, hipStream_t s) {
    const bool at = d.flags & SCL_GEMM_A_T, bt = d.flags & SCL_GEMM_B_T;
    if (!fill_operand(d.A, "A", at ? d.K : d.M, at ? d.M : d.K, &k.A, 4)) return SCL_EINVAL;
    if (!fill_operand(d.B, "B", bt ? d.K : d.N, bt ? d.N : d.K, &k.B, 4)) return SCL_EINVAL;
    const long long zdim = (long long)d.nb1 * d.nb2 * d.splitk;
    // 128x128 tiles once they fill the chip twice over; otherwise 64x64 (4x the blocks: the back-ends' maps are small)
    const long long t128 = (long long)((d.M + 127) / 128) * ((d.N + 127) / 128) * zdim;
    const bool x3 = d.flags & SCL_GEMM_F32X3;
    // K-contiguous operands take the staging-split kernel; a transposed operand would need 2-byte scatter stores into the images
    // (measured 95.8 vs 68.2 us on the ResNet TT weight gradient) and keeps the fragment-split form
    const bool frag_form = at || bt;
    if (t128 >= 512 && d.M >= 128 && d.N >= 128) {
        const dim3 grid((unsigned)(t128 / zdim), 1, (unsigned)zdim);
        if (x3 && !frag_form) f32p_launch<128>(k, grid, s);
        else if (x3) f32_launch<128, true>(k, at, bt, grid, s);
        else f32_launch<128, false>(k, at, bt, grid, s);
    } else {
        const long long t64 = (long long)((d.M + 63) / 64) * ((d.N + 63) / 64);
        const dim3 grid((unsigned)t64, 1, (unsigned)zdim);
        if (x3 && !frag_form) f32p_launch<64>(k, grid, s);
        else if (x3) f32_launch<64, true>(k, at, bt, grid, s);
        else f32_launch<64, false>(k, at, bt, grid, s);
    }
    return SCL_OK;
}

}  // namespace sclg
