// loss.hip — supervised-contrastive loss with the reference's sequence similarity, and the NLL term.
//
// Reference: model/loss_metrics.py:85-209 (sim_metric_seq, supcon_loss) as called from
// model/wav2vec2_linear_nll.py:158-192 (Model.loss): n_views = 1, contra_mode 'all', t = 0.07, no
// length normalisation.  With F[i] = feats[i] flattened to K = T'*d values,
//     S[i][j]   = <F[i], F[j]> / (T' * t)                        (bmm over frames, mean over frames, / t)
//     m_i       = max_j S[i][j] * selfmask[i][j]                 (diagonal forced to 0, detached)
//     logp[i][j]= (S[i][j] - m_i) - log sum_{j != i} exp(S[i][j] - m_i)
//     loss      = - mean_i  sum_{j != i, y_j == y_i} logp[i][j] / #{j != i, y_j == y_i}   (0/0 -> NaN kept)
// All of it in fp32 (the logits reach 1e2..1e4 because features are not normalised).
// Three kernels: Gram partials over K-chunks, a single-workgroup loss + dL/dS kernel, and the
// backward contraction dF = (dL/dS + dL/dS^T) F / (T' t).  The [T', bz, bz] temporary of the
// reference's bmm is never materialised.
#include <string.h>
#include "common.h"

namespace {

constexpr int KC = 64;  // K-chunk staged in LDS per step

// part[chunk][i][j] = sum_{k in chunk} F[i][k] F[j][k];   grid.x = number of chunks
__global__ __launch_bounds__(256) void supcon_gram_kernel(const float* __restrict__ F, float* __restrict__ part, int bz,
                                                          int64_t K, int64_t ldF, int64_t kchunk) {
    extern __shared__ float sm[];  // [bz][KC+1]
    const int64_t k0 = (int64_t)blockIdx.x * kchunk;
    const int64_t k1 = min(K, k0 + kchunk);
    const int npairs = bz * bz;
    // each thread owns pairs p = tid, tid+256, ...  (<= 64 pairs per thread for bz <= 128)
    float acc[64];
#pragma unroll
    for (int q = 0; q < 64; ++q) acc[q] = 0.f;
    for (int64_t kb = k0; kb < k1; kb += KC) {
        const int kn = (int)min((int64_t)KC, k1 - kb);
        __syncthreads();
        for (int idx = threadIdx.x; idx < bz * KC; idx += 256) {
            const int i = idx / KC, kk = idx % KC;
            sm[i * (KC + 1) + kk] = kk < kn ? F[(int64_t)i * ldF + kb + kk] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 64; ++q) {
            const int p = threadIdx.x + q * 256;
            if (p < npairs) {
                const int i = p / bz, j = p % bz;
                float s = 0.f;
                for (int kk = 0; kk < KC; ++kk) s += sm[i * (KC + 1) + kk] * sm[j * (KC + 1) + kk];
                acc[q] += s;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 64; ++q) {
        const int p = threadIdx.x + q * 256;
        if (p < npairs) part[(int64_t)blockIdx.x * npairs + p] = acc[q];
    }
}

// single block: S = sum(parts) * scale; loss; G[i][j] = dloss/dS[i][j]
__global__ __launch_bounds__(1024) void supcon_loss_kernel(const float* __restrict__ part, int nparts, const int64_t* __restrict__ labels,
                                                          int bz, float scale, float* __restrict__ loss_out, float* __restrict__ G,
                                                          float* __restrict__ S_out) {
    extern __shared__ float sm[];  // S [bz*bz], rowloss [bz]
    float* S = sm;
    float* rowloss = sm + bz * bz;
    const int npairs = bz * bz;
    for (int p = threadIdx.x; p < npairs; p += blockDim.x) {
        float s = 0.f;
        for (int c = 0; c < nparts; ++c) s += part[(int64_t)c * npairs + p];
        S[p] = s * scale;
        if (S_out) S_out[p] = s * scale;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = wv; i < bz; i += (int)(blockDim.x >> 6)) {
        const int64_t yi = labels[i];
        float mx = -INFINITY;
        for (int j = lane; j < bz; j += 64) mx = fmaxf(mx, j == i ? 0.f : S[i * bz + j]);  // logits * self_mask
        mx = wave_max(mx);
        float se = 0.f, npos = 0.f, spos = 0.f;
        for (int j = lane; j < bz; j += 64) {
            if (j == i) continue;
            const float l = S[i * bz + j] - mx;
            se += expf(l);
            if (labels[j] == yi) { npos += 1.f; spos += l; }
        }
        se = wave_sum(se); npos = wave_sum(npos); spos = wave_sum(spos);
        const float lse = logf(se);
        const float mlpp = (spos - npos * lse) / npos;  // 0/0 -> NaN as the reference
        if (lane == 0) rowloss[i] = -mlpp;
        // d(-mlpp_i)/dS[i][j] = -( pos_ij / npos - exp(l_ij) / se ), j != i ; times 1/bz for the mean
        for (int j = lane; j < bz; j += 64) {
            float g = 0.f;
            if (j != i) {
                const float l = S[i * bz + j] - mx;
                g = -((labels[j] == yi ? 1.f / npos : 0.f) - expf(l) / se) / (float)bz;
            }
            G[i * bz + j] = g;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < bz; ++i) s += rowloss[i];
        *loss_out = s / (float)bz;
    }
}

// ---- batches of more than 128 utterances (the reference has no limit: nn.DataParallel computes Model.loss on the gathered batch of all
// GPUs, 8 x 64 = 512 rows at BASELINE configs[2] / [3] — SCL_GLOBAL_SUPCON=1) --------------------------------------------------------
// Gram fallback for shapes the GEMM form does not take: one thread per pair over its K chunk, straight from global memory.
__global__ __launch_bounds__(256) void supcon_gram_any_kernel(const float* __restrict__ F, float* __restrict__ part, int bz,
                                                              int64_t K, int64_t ldF, int64_t kchunk) {
    const int64_t k0 = (int64_t)blockIdx.y * kchunk, k1 = min(K, k0 + kchunk);
    const int64_t npairs = (int64_t)bz * bz;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= npairs) return;
    const int i = (int)(p / bz), j = (int)(p % bz);
    const float* fi = F + (int64_t)i * ldF;
    const float* fj = F + (int64_t)j * ldF;
    float acc = 0.f;
    for (int64_t kb = k0; kb < k1; kb += KC) {      // the same chunking of the sum as supcon_gram_kernel: KC terms, then the running total
        const int64_t ke = min(k1, kb + KC);
        float s = 0.f;
        for (int64_t k = kb; k < ke; ++k) s += fi[k] * fj[k];
        acc += s;
    }
    part[(int64_t)blockIdx.y * npairs + p] = acc;
}

// The loss for any bz: one WAVE per row i (the rows are independent); S stays in global memory (S_buf, bz x bz), nothing in LDS.
// Same arithmetic per row as supcon_loss_kernel; rowloss[i] to global, summed in index order by supcon_loss_finish_kernel.
__global__ __launch_bounds__(256) void supcon_loss_rows_kernel(const float* __restrict__ part, int nparts, const int64_t* __restrict__ labels,
                                                               int bz, float scale, float* __restrict__ rowloss, float* __restrict__ G,
                                                               float* __restrict__ S_buf) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= bz) return;
    const int64_t npairs = (int64_t)bz * bz;
    float* Si = S_buf + (int64_t)i * bz;
    const int64_t yi = labels[i];
    float mx = -INFINITY;
    for (int j = lane; j < bz; j += 64) {
        float s = 0.f;
        for (int c = 0; c < nparts; ++c) s += part[(int64_t)c * npairs + (int64_t)i * bz + j];
        s *= scale;
        Si[j] = s;
        mx = fmaxf(mx, j == i ? 0.f : s);      // logits * self_mask
    }
    mx = wave_max(mx);
    float se = 0.f, npos = 0.f, spos = 0.f;
    for (int j = lane; j < bz; j += 64) {      // this lane re-reads the values it wrote itself
        if (j == i) continue;
        const float l = Si[j] - mx;
        se += expf(l);
        if (labels[j] == yi) { npos += 1.f; spos += l; }
    }
    se = wave_sum(se); npos = wave_sum(npos); spos = wave_sum(spos);
    const float lse = logf(se);
    const float mlpp = (spos - npos * lse) / npos;  // 0/0 -> NaN as the reference
    if (lane == 0) rowloss[i] = -mlpp;
    for (int j = lane; j < bz; j += 64) {
        float g = 0.f;
        if (j != i) {
            const float l = Si[j] - mx;
            g = -((labels[j] == yi ? 1.f / npos : 0.f) - expf(l) / se) / (float)bz;
        }
        G[(int64_t)i * bz + j] = g;
    }
}
__global__ void supcon_loss_finish_kernel(const float* __restrict__ rowloss, int bz, float* __restrict__ loss_out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < bz; ++i) s += rowloss[i];
        *loss_out = s / (float)bz;
    }
}

// dF[i][k] (+)= upstream * scale * sum_j (G[i][j] + G[j][i]) F[j][k]
__global__ __launch_bounds__(256) void supcon_bwd_kernel(const float* __restrict__ F, const float* __restrict__ G,
                                                         const float* __restrict__ upstream, float coef, float* __restrict__ dF,
                                                         bf16_t* __restrict__ dF_bf, int bz, int64_t K, int64_t ldF, int accumulate) {
    extern __shared__ float sm[];  // Gs [bz*bz] (bz <= 128; larger batches read G + G^T from global memory: block-uniform addresses)
    const bool in_lds = bz <= 128;
    if (in_lds) {
        for (int p = threadIdx.x; p < bz * bz; p += 256) {
            const int i = p / bz, j = p % bz;
            sm[p] = G[i * bz + j] + G[j * bz + i];
        }
        __syncthreads();
    }
    const float c = coef * (upstream ? *upstream : 1.f);
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    // column k of F for all j (coalesced across threads), then the bz outputs
    for (int i0 = 0; i0 < bz; i0 += 8) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int j = 0; j < bz; ++j) {
            const float f = F[(int64_t)j * ldF + k];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (i0 + q < bz)
                    acc[q] += (in_lds ? sm[(i0 + q) * bz + j] : G[(int64_t)(i0 + q) * bz + j] + G[(int64_t)j * bz + i0 + q]) * f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (i0 + q < bz) {
                const int64_t o = (int64_t)(i0 + q) * ldF + k;
                float v = c * acc[q];
                if (accumulate) v += dF[o];
                dF[o] = v;
                if (dF_bf) dF_bf[o] = f2bf(v);
            }
        }
    }
}

// NLL as the reference computes it: CrossEntropyLoss applied to log-probs, mean over the batch, then / bz.
//   loss = (1/bz) * mean_i ( -log_softmax(logp_i)[y_i] );  dlogp = upstream * (softmax(logp_i) - onehot) / bz^2
__global__ void nll_kernel(const float* __restrict__ logp, const int64_t* __restrict__ labels, int bz, int NC,
                           float* __restrict__ loss_out, float* __restrict__ dlogp_coef) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < bz; i += blockDim.x) {
        float mx = -INFINITY;
        for (int k = 0; k < NC; ++k) mx = fmaxf(mx, logp[i * NC + k]);
        float se = 0.f;
        for (int k = 0; k < NC; ++k) se += expf(logp[i * NC + k] - mx);
        const float lse = mx + logf(se);
        const int y = (int)labels[i];
        s += -(logp[i * NC + y] - lse);
        for (int k = 0; k < NC; ++k)
            dlogp_coef[i * NC + k] = (expf(logp[i * NC + k] - lse) - (k == y ? 1.f : 0.f)) / ((float)bz * (float)bz);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < blockDim.x; ++i) t += red[i];
        *loss_out = t / ((float)bz * (float)bz);
    }
}

// Gs[i][j] = c * (G[i][j] + G[j][i]),  c = coef * (*upstream): the A operand of the backward contraction when it runs as a GEMM
__global__ __launch_bounds__(256) void supcon_gsym_kernel(const float* __restrict__ G, const float* __restrict__ upstream, float coef,
                                                          float* __restrict__ Gs, int bz, int ld) {
    const float c = coef * (upstream ? *upstream : 1.f);
    for (int p = blockIdx.x * 256 + threadIdx.x; p < bz * ld; p += gridDim.x * 256) {
        const int i = p / ld, j = p % ld;
        Gs[p] = j < bz ? c * (G[i * bz + j] + G[j * bz + i]) : 0.f;
    }
}

// Both contractions are GEMMs with a tiny M — [bz, K] x [K, bz] (split along K when the rows are long: feats, K = T' x 128 = 25 472)
// and [bz, bz] x [bz, K] — and run on the exact-fp32 matrix-core kernel (gemm_f32.hip) instead of the scalar kernels above, which put
// 25 / 100 workgroups on the 256 CUs for feats (130 / 220 us per call at bz = 64; the GEMMs: ~12 / ~8 us) and ONE for emb (K = 128:
// 210 us in the backward).  The scalar kernels remain for batch sizes / row lengths that are not multiples of 4.
inline bool supcon_as_gemm(int bz, int64_t K, int64_t ldF, const void* F) {      // bz % 4: the backward's reduction runs over bz rows of F
    return K >= 64 && (K & 3) == 0 && (ldF & 3) == 0 && (bz & 3) == 0 && ((uintptr_t)F & 15) == 0;
}
constexpr int SUPCON_GEMM_SPLIT = 32;
inline int supcon_split(int64_t K) { const int64_t s = K / 256; return (int)(s < 1 ? 1 : (s > SUPCON_GEMM_SPLIT ? SUPCON_GEMM_SPLIT : s)); }

inline SclOperand f32_rows(const float* p, int64_t ld) {
    SclOperand o;
    memset(&o, 0, sizeof(o));
    o.ptr = p; o.rpb = 0x7fffffff; o.ld = (int32_t)ld; o.cin = 0x7fffffff;
    return o;
}

}  // namespace

extern "C" int scl_supcon_nchunks(int64_t K) {
    int64_t n = (K + 1023) / 1024;
    if (n > 512) n = 512;
    if (n < SUPCON_GEMM_SPLIT) n = SUPCON_GEMM_SPLIT;      // the GEMM form writes this many slabs
    return (int)n;
}

// floats the forward's workspace needs: the partial Gram matrices + one row of per-utterance losses (batches of more than 128)
extern "C" long long scl_supcon_ws_floats(int bz, int64_t K) {
    return (long long)scl_supcon_nchunks(K) * bz * bz + bz;
}

// F f32 [bz, K] (row stride ldF); labels int64 [bz]; ws: scl_supcon_ws_floats(bz, K) floats; G: 2*bz*bz floats.
// loss_out = supcon_loss(feats) exactly as the reference returns it (before Model.loss's extra 1/bz).
extern "C" int scl_supcon_fwd(const float* F, const int64_t* labels, int bz, int64_t K, int64_t ldF, int Tprime, float temperature,
                              float* ws, float* G, float* loss_out, float* S_out, void* stream) {
    SCL_REQUIRE(F && labels && ws && G && loss_out, "supcon_fwd: null pointer");
    SCL_REQUIRE(bz >= 1 && bz <= SCL_SUPCON_MAX_BZ && K >= 1 && Tprime >= 1 && temperature > 0.f, "supcon_fwd: need 1 <= bz <= %d", SCL_SUPCON_MAX_BZ);
    hipStream_t s = (hipStream_t)stream;
    // bz <= 128: S, the loss and dL/dS in ONE workgroup (S in LDS).  Larger batches: a wave per row, S in global memory — the caller's G
    // buffer (2 bz^2 floats) holds dL/dS in its first half and S in its second (the backward overwrites that half with its own scratch),
    // the row losses go behind the partial sums (ws holds one extra row of bz floats: scl_supcon_ws_floats)
    const bool small = bz <= 128;
    auto finish = [&](int nparts) {
        const float scale = 1.0f / ((float)Tprime * temperature);
        if (small) {
            hipLaunchKernelGGL(supcon_loss_kernel, dim3(1), dim3(1024), (size_t)(bz * bz + bz) * sizeof(float), s, ws, nparts, labels, bz, scale,
                               loss_out, G, S_out);
        } else {
            float* rowloss = ws + (size_t)scl_supcon_nchunks(K) * bz * bz;
            float* S_buf = S_out ? S_out : G + (size_t)bz * bz;
            hipLaunchKernelGGL(supcon_loss_rows_kernel, dim3((bz + 3) / 4), dim3(256), 0, s, ws, nparts, labels, bz, scale, rowloss, G, S_buf);
            hipLaunchKernelGGL(supcon_loss_finish_kernel, dim3(1), dim3(64), 0, s, rowloss, bz, loss_out);
        }
    };
    if (supcon_as_gemm(bz, K, ldF, F)) {
        SclGemmDesc d;
        memset(&d, 0, sizeof(d));
        d.A = f32_rows(F, ldF); d.B = f32_rows(F, ldF);
        d.C = ws; d.ldc = bz; d.c_rpb = 0x7fffffff; d.M = bz; d.N = bz; d.K = (int32_t)K; d.nb1 = 1; d.nb2 = 1;
        const int sk = supcon_split(K);
        d.splitk = sk; d.c_split_stride = sk > 1 ? (int64_t)bz * bz : 0; d.flags = SCL_GEMM_C_F32 | SCL_GEMM_AB_F32; d.alpha = 1.0f;
        const int rc = scl_gemm_bf16(&d, stream);
        if (rc != SCL_OK) return rc;
        finish(sk);
        return scl_check_launch("scl_supcon_fwd");
    }
    const int nch = (int)((K + 1023) / 1024 < 1 ? 1 : ((K + 1023) / 1024 > 512 ? 512 : (K + 1023) / 1024));
    int64_t kchunk = (K + nch - 1) / nch;
    kchunk = (kchunk + KC - 1) / KC * KC;
    const int nch_eff = (int)((K + kchunk - 1) / kchunk);
    if (small) hipLaunchKernelGGL(supcon_gram_kernel, dim3(nch_eff), dim3(256), (size_t)bz * (KC + 1) * sizeof(float), s, F, ws, bz, K, ldF, kchunk);
    else hipLaunchKernelGGL(supcon_gram_any_kernel, dim3((unsigned)(((int64_t)bz * bz + 255) / 256), nch_eff), dim3(256), 0, s, F, ws, bz, K, ldF, kchunk);
    finish(nch_eff);
    return scl_check_launch("scl_supcon_fwd");
}

// dF (+)= (*upstream) * coef / (T' t) * (G + G^T) F      (coef carries Model.loss's 1/bz)
extern "C" int scl_supcon_bwd(const float* F, const float* G, const float* upstream, float coef, int bz, int64_t K, int64_t ldF,
                              int Tprime, float temperature, float* dF, void* dF_bf16, int accumulate, void* stream) {
    SCL_REQUIRE(F && G && dF && bz >= 1 && bz <= SCL_SUPCON_MAX_BZ && K >= 1, "supcon_bwd: bad args (1 <= bz <= %d)", SCL_SUPCON_MAX_BZ);
    if (supcon_as_gemm(bz, K, ldF, F) && ((uintptr_t)dF & 15) == 0 && (dF_bf16 == nullptr || ((uintptr_t)dF_bf16 & 7) == 0)) {
        // dF[i][k] (+)= sum_j Gs[i][j] F[j][k]:  A = Gs [bz, bz4] (reduction padded to a multiple of 4 with zero columns),
        // B = F read as [j rows][k contiguous] (transposed operand), C = dF f32 (+ R = dF when accumulating), C2 = the bf16 copy
        float* gs_buf = const_cast<float*>(G) + (size_t)bz * bz;       // the caller's G holds 2 * bz * bz floats: [bz*bz, 2*bz*bz) is scratch
        const int bz4 = bz;                                             // bz % 4 == 0 on this path
        hipLaunchKernelGGL(supcon_gsym_kernel, dim3(16), dim3(256), 0, (hipStream_t)stream, G, upstream, coef / ((float)Tprime * temperature),
                           gs_buf, bz, bz4);
        SclGemmDesc d;
        memset(&d, 0, sizeof(d));
        d.A = f32_rows(gs_buf, bz4); d.B = f32_rows(F, ldF);
        d.C = dF; d.ldc = (int32_t)ldF; d.c_rpb = 0x7fffffff; d.M = bz; d.N = (int32_t)K; d.K = bz4; d.nb1 = 1; d.nb2 = 1; d.splitk = 1;
        d.flags = SCL_GEMM_C_F32 | SCL_GEMM_AB_F32 | SCL_GEMM_B_T; d.alpha = 1.0f;
        if (accumulate) { d.R = dF; d.flags |= SCL_GEMM_R_F32 | (1 << SCL_GEMM_RMODE_SHIFT); }
        if (dF_bf16) {
            // the epilogue's second output is the value BEFORE the residual add: only usable when nothing is accumulated
            if (accumulate) { scl_set_error("supcon_bwd: bf16 copy with accumulate is not supported on the GEMM path"); return SCL_EUNSUPPORTED; }
            d.C2 = dF_bf16; d.flags |= SCL_GEMM_HAS_C2;
        }
        const int rc = scl_gemm_bf16(&d, stream);
        if (rc != SCL_OK) return rc;
        return scl_check_launch("scl_supcon_bwd");
    }
    hipLaunchKernelGGL(supcon_bwd_kernel, dim3((unsigned)((K + 255) / 256)), dim3(256), bz <= 128 ? (size_t)bz * bz * sizeof(float) : 0, (hipStream_t)stream,
                       F, G, upstream, coef / ((float)Tprime * temperature), dF, (bf16_t*)dF_bf16, bz, K, ldF, accumulate);
    return scl_check_launch("scl_supcon_bwd");
}

extern "C" int scl_nll_fwd(const float* logp, const int64_t* labels, int bz, int NC, float* loss_out, float* dlogp_coef, void* stream) {
    SCL_REQUIRE(logp && labels && loss_out && dlogp_coef && bz >= 1 && NC >= 1 && NC <= 8, "nll_fwd: bad args");
    hipLaunchKernelGGL(nll_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logp, labels, bz, NC, loss_out, dlogp_coef);
    return scl_check_launch("scl_nll_fwd");
}
