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
//   * Operand layouts: K-contiguous operands are staged as [rows][64 k] with a 16-byte-slot XOR
//     swizzle (slot ^= (row>>1)&7) so ds_read_b128 fragment reads are bank-conflict free;
//     transposed operands ([K rows][128 contiguous]) are staged as they lie in memory (coalesced
//     16-byte loads, 32-byte-chunk XOR swizzle) and delivered to the MFMA through
//     ds_read_b64_tr_b16 — no transposed copies of weights or activations exist anywhere.
//   * Epilogue fused: alpha, bias, activation (+ pre-activation second output), residual add or
//     activation-gradient multiply, dropout mask, bf16/f32 store, split-K slabs.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;  // 16 KiB per operand per stage

__device__ __forceinline__ uint4 mask_tail(uint4 v, int nvalid) {  // keep the first nvalid (1..7) bf16
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        if (2 * d >= nvalid) w[d] = 0u;
        else if (2 * d + 1 >= nvalid) w[d] &= 0x0000FFFFu;
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// ---- staging of a K-contiguous operand tile: [128 rows][64 k] -------------------------------
struct StageK {
    const bf16_t* base;
    int64_t rowoff[4];
    int64_t kq_off;
    int krem, kcur, kend, cin;
    int64_t cout;
    uint32_t rowok;  // bit i: row i in range
    uint32_t lds_off[4];

    __device__ __forceinline__ void init(const SclOperand& o, const bf16_t* b, int row0, int rowlimit,
                                         int kbegin, int kend_, int tid) {
        base = b;
        const int c = tid & 7;
        rowok = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (tid >> 3) + 32 * i;
            const int r = row0 + row;
            const bool ok = r < rowlimit;
            const int rr = ok ? r : 0;
            rowoff[i] = (int64_t)(rr / o.rpb) * o.rbstride + (int64_t)(rr % o.rpb) * o.ld;
            rowok |= (ok ? 1u : 0u) << i;
            lds_off[i] = (uint32_t)(row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
        }
        cin = o.cin; cout = o.cout;
        kcur = kbegin + 8 * c; kend = kend_;
        kq_off = (int64_t)(kcur / cin) * cout;
        krem = kcur % cin;
    }
    __device__ __forceinline__ void load(uint4 (&r)[4]) const {
        const int nvalid = kend - kcur;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (((rowok >> i) & 1u) && nvalid > 0) {
                v = *reinterpret_cast<const uint4*>(base + rowoff[i] + kq_off + krem);
                if (nvalid < 8) v = mask_tail(v, nvalid);
            }
            r[i] = v;
        }
    }
    __device__ __forceinline__ void advance() {
        kcur += BK; krem += BK;
        while (krem >= cin) { krem -= cin; kq_off += cout; }
    }
};

// ---- staging of a transposed operand tile: [64 k rows][128 contiguous] ----------------------
struct StageT {
    const bf16_t* base;
    int64_t coloff;
    int64_t rq_off[4];
    int rrem[4];
    int rcur, kend, rpb, ld, ncolvalid;
    int64_t rbstride;
    uint32_t lds_off[4];

    __device__ __forceinline__ void init(const SclOperand& o, const bf16_t* b, int col0, int collimit,
                                         int kbegin, int kend_, int tid) {
        base = b;
        const int c16 = tid & 15;
        const int col = col0 + 8 * c16;
        int nv = collimit - col; nv = nv < 0 ? 0 : (nv > 8 ? 8 : nv);
        ncolvalid = nv;
        const int cc = nv > 0 ? col : 0;
        coloff = (int64_t)(cc / o.cin) * o.cout + (cc % o.cin);
        rpb = o.rpb; ld = o.ld; rbstride = o.rbstride;
        rcur = kbegin + (tid >> 4); kend = kend_;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int krow = (tid >> 4) + 16 * i;
            const int r = kbegin + krow;
            rq_off[i] = (int64_t)(r / rpb) * rbstride;
            rrem[i] = r % rpb;
            const int sw = (krow & 3) | (((krow >> 3) & 1) << 2);
            lds_off[i] = (uint32_t)(krow * 256 + (((c16 >> 1) ^ sw) << 5) + ((c16 & 1) << 4));
        }
    }
    __device__ __forceinline__ void load(uint4 (&r)[4]) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ncolvalid > 0 && (rcur + 16 * i) < kend) {
                v = *reinterpret_cast<const uint4*>(base + rq_off[i] + (int64_t)rrem[i] * ld + coloff);
                if (ncolvalid < 8) v = mask_tail(v, ncolvalid);
            }
            r[i] = v;
        }
    }
    __device__ __forceinline__ void advance() {
        rcur += BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rrem[i] += BK;
            while (rrem[i] >= rpb) { rrem[i] -= rpb; rq_off[i] += rbstride; }
        }
    }
};

template <bool T> struct StageSel { typedef StageK type; };
template <> struct StageSel<true> { typedef StageT type; };

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// fragment of a K-contiguous tile: 16 rows x 32 k; lane l holds row (l&15), k = 8*(l>>4)+0..7
__device__ __forceinline__ bf16x8 frag_k(const char* tile, int rowblk, int ks, int lane) {
    const int row = rowblk * 16 + (lane & 15);
    const int c = 4 * ks + (lane >> 4);
    const char* p = tile + row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
    return *reinterpret_cast<const bf16x8*>(p);
}
// fragment of a transposed tile ([k][col]): the same register image, through ds_read_b64_tr_b16
__device__ __forceinline__ bf16x8 frag_t(const char* tile, int colblk, int ks, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const int krow = 32 * ks + 8 * g + (i >> 2);
    const int sw = (krow & 3) | (((krow >> 3) & 1) << 2);
    const char* p = tile + krow * 256 + ((colblk ^ sw) << 5) + ((i & 3) << 3);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * 256));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

template <bool AT, bool BT>
__global__ __launch_bounds__(256, 2) void scl_gemm_kernel(const SclGemmDesc d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    const int tiles_n = (d.N + BN - 1) / BN;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    int z = blockIdx.z;
    const int ksplit = z % d.splitk; z /= d.splitk;
    const int z1 = z / d.nb2, z2 = z % d.nb2;

    // K range of this split (multiples of BK)
    const int nk_total = (d.K + BK - 1) / BK;
    const int nk_per = (nk_total + d.splitk - 1) / d.splitk;
    const int kbegin = ksplit * nk_per * BK;
    int kend = kbegin + nk_per * BK; if (kend > d.K) kend = d.K;
    const int nk = kend > kbegin ? (kend - kbegin + BK - 1) / BK : 0;

    const bf16_t* Ab = reinterpret_cast<const bf16_t*>(d.A.ptr) + z1 * d.A.bs1 + z2 * d.A.bs2;
    const bf16_t* Bb = reinterpret_cast<const bf16_t*>(d.B.ptr) + z1 * d.B.bs1 + z2 * d.B.bs2;

    typename StageSel<AT>::type sa;
    typename StageSel<BT>::type sb;
    sa.init(d.A, Ab, m0, d.M, kbegin, kend, tid);
    sb.init(d.B, Bb, n0, d.N, kbegin, kend, tid);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint4 ra[4], rb[4];
    if (nk > 0) {
        sa.load(ra); sb.load(rb);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<uint4*>(smem + sa.lds_off[i]) = ra[i];
            *reinterpret_cast<uint4*>(smem + TILE_BYTES + sb.lds_off[i]) = rb[i];
        }
    }
    __syncthreads();

    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = (kt + 1) < nk;
        if (more) {
            sa.advance(); sb.advance();
            sa.load(ra); sb.load(rb);
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
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            char* nA = smem + (cur ^ 1) * (2 * TILE_BYTES);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<uint4*>(nA + sa.lds_off[i]) = ra[i];
                *reinterpret_cast<uint4*>(nA + TILE_BYTES + sb.lds_off[i]) = rb[i];
            }
        }
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue ----------------------------------------------------------------------------
    const int flags = d.flags;
    const bool c_f32 = flags & SCL_GEMM_C_F32, c2_f32 = flags & SCL_GEMM_C2_F32, r_f32 = flags & SCL_GEMM_R_F32;
    const bool has_bias = flags & SCL_GEMM_HAS_BIAS, has_c2 = flags & SCL_GEMM_HAS_C2, drop = flags & SCL_GEMM_DROPOUT;
    const int act = (flags >> SCL_GEMM_ACT_SHIFT) & 0xF;
    const int rmode = (flags >> SCL_GEMM_RMODE_SHIFT) & 0xF;
    const int ract = (flags >> SCL_GEMM_RACT_SHIFT) & 0xF;
    const int64_t cbase = z1 * d.c_bs1 + z2 * d.c_bs2 + (int64_t)ksplit * d.c_split_stride;
    const float* bias = has_bias ? d.bias + z2 * d.bias_bs2 : nullptr;
    const int g = lane >> 4, lc = lane & 15;

#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = m0 + wr * 64 + mt * 16 + 4 * g + reg;
            if (row >= d.M) continue;
            const int64_t roff = cbase + (int64_t)(row / d.c_rpb) * d.c_rbstride + (int64_t)(row % d.c_rpb) * d.ldc;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int col = n0 + wc * 64 + nt * 16 + lc;
                if (col >= d.N) continue;
                float v = d.alpha * acc[mt][nt][reg];
                if (has_bias) v += bias[col];
                const int64_t off = roff + col;
                if (has_c2) {
                    if (c2_f32) reinterpret_cast<float*>(d.C2)[off] = v;
                    else reinterpret_cast<bf16_t*>(d.C2)[off] = f2bf(v);
                }
                v = act_f(act, v);
                if (rmode == 2) {
                    const float h = r_f32 ? reinterpret_cast<const float*>(d.R)[off]
                                          : bf2f(reinterpret_cast<const bf16_t*>(d.R)[off]);
                    v *= act_grad_f(ract, h);
                }
                if (drop) v *= dropout_scale(d.drop_seed, (uint64_t)off, d.drop_p);
                if (rmode == 1) {
                    v += r_f32 ? reinterpret_cast<const float*>(d.R)[off]
                               : bf2f(reinterpret_cast<const bf16_t*>(d.R)[off]);
                }
                if (c_f32) reinterpret_cast<float*>(d.C)[off] = v;
                else reinterpret_cast<bf16_t*>(d.C)[off] = f2bf(v);
            }
        }
    }
}

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

bool operand_ok(const SclOperand& o, const char* name) {
    if (!o.ptr || ((uintptr_t)o.ptr & 15)) { scl_set_error("gemm: %s ptr null or not 16B aligned", name); return false; }
    if (o.rpb < 1 || o.cin < 8 || (o.cin != 0x7fffffff && (o.cin & 7))) { scl_set_error("gemm: %s rpb/cin invalid", name); return false; }
    if ((o.ld & 7) || (o.rbstride & 7) || (o.cout & 7) || (o.bs1 & 7) || (o.bs2 & 7)) {
        scl_set_error("gemm: %s strides must be multiples of 8 elements", name); return false;
    }
    return true;
}

}  // namespace

extern "C" int scl_gemm_bf16(const SclGemmDesc* dp, void* stream) {
    SCL_REQUIRE(dp, "gemm: null desc");
    const SclGemmDesc& d = *dp;
    SCL_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0, "gemm: M,N,K must be positive (%d,%d,%d)", d.M, d.N, d.K);
    SCL_REQUIRE(d.nb1 >= 1 && d.nb2 >= 1 && d.splitk >= 1, "gemm: nb1/nb2/splitk must be >= 1");
    if (!operand_ok(d.A, "A") || !operand_ok(d.B, "B")) return SCL_EINVAL;
    SCL_REQUIRE(d.C && d.c_rpb >= 1, "gemm: C null or c_rpb < 1");
    const int rmode = (d.flags >> SCL_GEMM_RMODE_SHIFT) & 0xF;
    SCL_REQUIRE(rmode == 0 || d.R, "gemm: RMODE set but R is null");
    SCL_REQUIRE(!(d.flags & SCL_GEMM_HAS_C2) || d.C2, "gemm: HAS_C2 set but C2 is null");
    SCL_REQUIRE(!(d.flags & SCL_GEMM_HAS_BIAS) || d.bias, "gemm: HAS_BIAS set but bias is null");
    SCL_REQUIRE(d.splitk == 1 || ((d.flags & SCL_GEMM_C_F32) && rmode == 0 && !(d.flags & (SCL_GEMM_HAS_BIAS | SCL_GEMM_HAS_C2))
                                  && ((d.flags >> SCL_GEMM_ACT_SHIFT) & 0xF) == 0),
                "gemm: split-K needs a plain f32 slab output");
    const int tiles = ((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN);
    const long long zdim = (long long)d.nb1 * d.nb2 * d.splitk;
    SCL_REQUIRE(zdim <= 65535, "gemm: batch*splitk too large (%lld)", zdim);
    dim3 grid(tiles, 1, (unsigned)zdim), block(256);
    const size_t lds = 4 * TILE_BYTES;
    hipStream_t s = (hipStream_t)stream;
    const bool at = d.flags & SCL_GEMM_A_T, bt = d.flags & SCL_GEMM_B_T;
    {
        SclProfScope prof(SCL_KID_GEMM, s, 2.0 * d.M * d.N * (double)d.K * d.nb1 * d.nb2);
        if (!at && !bt) hipLaunchKernelGGL((scl_gemm_kernel<false, false>), grid, block, lds, s, d);
        else if (!at && bt) hipLaunchKernelGGL((scl_gemm_kernel<false, true>), grid, block, lds, s, d);
        else if (at && !bt) hipLaunchKernelGGL((scl_gemm_kernel<true, false>), grid, block, lds, s, d);
        else hipLaunchKernelGGL((scl_gemm_kernel<true, true>), grid, block, lds, s, d);
    }
    return scl_check_launch("scl_gemm_bf16");
}

extern "C" int scl_reduce_slabs_f32(const float* slabs, float* out, int64_t n, int nslabs, int64_t stride, void* stream) {
    SCL_REQUIRE(slabs && out && n > 0 && nslabs >= 1, "reduce_slabs: bad args");
    SCL_REQUIRE(((uintptr_t)slabs & 15) == 0 && ((uintptr_t)out & 15) == 0 && (stride & 3) == 0, "reduce_slabs: alignment");
    int blocks = (int)((n / 4 + 255) / 256); if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(scl_reduce_slabs_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, slabs, out, n, nslabs, stride);
    return scl_check_launch("scl_reduce_slabs_f32");
}
