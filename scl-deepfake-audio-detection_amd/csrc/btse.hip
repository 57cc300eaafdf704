// btse.hip — the "bio" branch of the reference's wav2vec2_btse plugin (BASELINE.json configs[4]): row kernels and per-head attention
// kernels behind two entry points (scl_btse_bio_fwd / _bwd), and the join in front of fc2.  Reference: model/wav2vec2_btse/model.py:210-238 (bioEncoderTransformersmall), :321-343 (Model.forward);
// model/wav2vec2_btse/transformer.py:17-52 (Encoder), :105-260 (MultiHeadAttention, window_size = 4 relative keys and values shared by
// the heads), :261-306 (FFN); model/wav2vec2_btse/modules.py:27-39 (LayerNorm over channels).
//
// The encoder is 40 752 parameters on a few hundred tokens (embedding -> n_layers x {QKV, relative-position attention, out-projection +
// residual + LayerNorm, FFN + residual + LayerNorm} -> scoring conv at the last position), fp32 throughout, laid out so that nothing waits on
// a chain of dependent memory round trips:
//   * every product with a 32- / 128-wide weight matrix is THREAD-PER-ROW (one workgroup per utterance): a thread holds its token's row in
//     registers, the output channel is a uniform loop index, so the weights arrive through scalar loads and enter the FMAs as scalar
//     operands; LayerNorm and the residual adds are in-thread, all L rows run in parallel;
//   * the attention is one workgroup per (utterance, HEAD), thread per row: the head's K and V slices sit in LDS and every lane reads the
//     SAME key row (a broadcast), two passes over the keys (maximum; exponentials + P V), the relative-position skew (scores[i][j] +=
//     q_i . emb_rel_k[j - i + 4], out[i] += p[i][j] emb_rel_v[j - i + 4] for |j - i| <= 4: what the reference's pad / reshape tricks at
//     transformer.py:189-243 compute) as a band test — not one cross-lane operation;
//   * the backward recomputes the probabilities from the saved row log-sum-exp, per query row (dq, delta, the band of dS / P the
//     relative-embedding gradients need) and per key row (dk, dv, with Q and dA in LDS); parameter gradients are reductions over the
//     rows: 64-row chunks of both operands staged in LDS, a thread owns 4 - 16 outputs in registers; they leave as one slab row per
//     utterance and the host sums the rows in index order.
// History (profiles/r5_btse_bio_probe*.txt, batch 128 x 199 tokens, forward / backward): one-channel-per-thread row loops re-reading the
// scratch row in every trip 1.80 / 4.06 ms (at batch 64); thread-per-row with scalar weights, one workgroup per utterance for the whole encoder
// 1.00 / 2.05 ms (half the CUs idle, one wave per SIMD on the rest, 796 (row, head) items behind 256 threads); 512 threads per utterance
// 0.74 / 1.71 ms; the attention as its own launch per layer with a workgroup per (utterance, head) 0.59 / 1.20 ms (2 n_layers + 1 launches
// forward, 3 n_layers + 1 backward); four lanes per attention row over interleaved keys 0.50 / 1.03 ms; the forward's row launches as 64-thread
// workgroups over (utterance, 64-row chunk) — the same per-thread work on all 256 CUs instead of four waves on each of B — 0.42 / 1.02 ms
// (512 tokens: 3.9 / 8.0 -> 0.59 / 2.40 ms).  What is left is the backward's row + reduction launches (4 x ~170 us: one workgroup per
// utterance, ~1500 scalar weight loads per row phase behind a single wave per SIMD, six weight-gradient reductions in sequence).
// The activations the backward needs go to a per-utterance scratch row in HBM (L2-resident: 392 L floats per layer).
// model.py:236 reads the LAST padded position times its mask: an utterance shorter than L scores exactly zero and contributes no
// gradient — the backward writes a zero row for it and returns.
#include "common.h"

namespace {
constexpr int BD = 32, BH = 4, BK = 8, BF = 128, BW = 4, NR = 2 * BW + 1, NT = 256, NG = NT / 32, NTA = 256, KS = 4, RPB = NTA / KS, NTR = 64, MAXL = 512, RC = 64;
constexpr float QSCALE = 0.35355339059327373f;      // 1 / sqrt(k_channels = 8), transformer.py:155
constexpr float EMB_SCALE = 5.656854249492381f;     // sqrt(bio_dim = 32), model.py:228
constexpr float FILL = -1e4f, EPS = 1e-5f;          // transformer.py:168, modules.py:28
enum { I_WQ, I_BQ, I_WK, I_BK, I_WV, I_BV, I_WO, I_BO, I_EK, I_EV, I_G1, I_B1, I_W1, I_C1, I_W2, I_C2, I_G2, I_B2 };
// per-utterance scratch, in units of L floats: layer l at l * O_LAYER, the encoder output at n_layers * O_LAYER, the backward's own
// buffers behind it
enum : int { O_XIN = 0, O_Q = 32, O_K = 64, O_V = 96, O_A = 128, O_S1 = 160, O_X1 = 192, O_S2 = 224, O_H = 256, O_ST = 384, O_LSE = 388, O_LAYER = 392 };
enum : int { G_DX = 0, G_Q = 32, G_K = 64, G_V = 96, G_DA = 128, G_DS1 = 160, G_DS2 = 192, G_DH = 224, G_DELTA = 352, G_BS = 356, G_BP = 392, G_TOTAL = 428 };

// The parameters are read-only for the lifetime of a launch: addressed through the CONSTANT address space, a load with a uniform index is a
// scalar load (s_load_dwordx8 / x16 into SGPRs, which feed the FMAs as scalar operands).  Through a plain pointer the compiler must assume
// the kernel's own stores to the scratch row could alias the weights and issues one vector load per FMA instead (351 global loads in the
// forward kernel, every one a memory round trip that all 64 lanes wait for: 1.03 ms instead of the time below).
typedef const __attribute__((address_space(4))) float* cptr;
__device__ __forceinline__ cptr as_const(const float* p) { return (cptr)(uintptr_t)p; }

__device__ __forceinline__ void load_row32(const float* __restrict__ p, float (&x)[BD]) {
    const float4* p4 = (const float4*)p;
#pragma unroll
    for (int i = 0; i < BD / 4; ++i) { const float4 v = p4[i]; x[4 * i] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w; }
}
__device__ __forceinline__ void store_row32(float* __restrict__ p, const float (&x)[BD]) {
    float4* p4 = (float4*)p;
#pragma unroll
    for (int i = 0; i < BD / 4; ++i) p4[i] = make_float4(x[4 * i], x[4 * i + 1], x[4 * i + 2], x[4 * i + 3]);
}
// y[c] = b[c] + sum_k x[k] W[c][k] for one row held in registers: the output channel is a uniform loop index, so the weight row
// arrives through scalar loads and feeds the FMAs as scalar operands — no LDS, no per-lane weight traffic
__device__ __forceinline__ void matvec32(const float (&x)[BD], cptr W, cptr b, float (&y)[BD]) {
#pragma unroll 4
    for (int c = 0; c < BD; ++c) {
        float a = b[c];
#pragma unroll
        for (int k = 0; k < BD; ++k) a = fmaf(x[k], W[c * BD + k], a);
        y[c] = a;
    }
}
// y[k] += sum_c d[c] W[c][k]  (the data gradient of the same layer: rows of W again, the output index now the unrolled one)
__device__ __forceinline__ void matvec32_t(const float (&d)[BD], cptr W, float (&y)[BD]) {
#pragma unroll 4
    for (int c = 0; c < BD; ++c) {
        const float dc = d[c];
#pragma unroll
        for (int k = 0; k < BD; ++k) y[k] = fmaf(dc, W[c * BD + k], y[k]);
    }
}
__device__ __forceinline__ void layernorm32(const float (&s)[BD], cptr g, cptr b, float (&y)[BD], float& mean, float& rs) {
    float m = 0.f;
#pragma unroll
    for (int c = 0; c < BD; ++c) m += s[c];
    m *= 1.f / BD;
    float v = 0.f;
#pragma unroll
    for (int c = 0; c < BD; ++c) { const float d = s[c] - m; v = fmaf(d, d, v); }
    rs = 1.f / sqrtf(v * (1.f / BD) + EPS);
    mean = m;
#pragma unroll
    for (int c = 0; c < BD; ++c) y[c] = (s[c] - m) * rs * g[c] + b[c];
}
// ds = LayerNorm backward of one row (dy in, pre-norm row s, its mean / rstd)
__device__ __forceinline__ void layernorm32_bwd(const float (&dy)[BD], const float (&s)[BD], float mean, float rs, cptr g, float (&ds)[BD]) {
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int c = 0; c < BD; ++c) { const float xh = (s[c] - mean) * rs, dxh = dy[c] * g[c]; m1 += dxh; m2 = fmaf(dxh, xh, m2); }
    m1 *= 1.f / BD; m2 *= 1.f / BD;
#pragma unroll
    for (int c = 0; c < BD; ++c) { const float xh = (s[c] - mean) * rs; ds[c] = rs * (dy[c] * g[c] - m1 - xh * m2); }
}

// One attention score of query row i (scaled q in registers) against key row j of head h: q . (k_j + emb_rel_k[j - i + 4]) inside the band
__device__ __forceinline__ float score(const float (&qs)[BK], const float* __restrict__ kr, const float* __restrict__ Eks, int dj) {
    const float4 k0 = ((const float4*)kr)[0], k1 = ((const float4*)kr)[1];
    float s = qs[0] * k0.x;
    s = fmaf(qs[1], k0.y, s); s = fmaf(qs[2], k0.z, s); s = fmaf(qs[3], k0.w, s);
    s = fmaf(qs[4], k1.x, s); s = fmaf(qs[5], k1.y, s); s = fmaf(qs[6], k1.z, s); s = fmaf(qs[7], k1.w, s);
    if ((unsigned)(dj + BW) <= 2u * BW) {
        const float* e = Eks + (dj + BW) * BK;
#pragma unroll
        for (int d = 0; d < BK; ++d) s = fmaf(qs[d], e[d], s);
    }
    return s;
}
__device__ __forceinline__ float dot8(const float* __restrict__ a, const float (&b)[BK]) {
    const float4 a0 = ((const float4*)a)[0], a1 = ((const float4*)a)[1];
    float s = a0.x * b[0];
    s = fmaf(a0.y, b[1], s); s = fmaf(a0.z, b[2], s); s = fmaf(a0.w, b[3], s);
    s = fmaf(a1.x, b[4], s); s = fmaf(a1.y, b[5], s); s = fmaf(a1.z, b[6], s); s = fmaf(a1.w, b[7], s);
    return s;
}
__device__ __forceinline__ float dot8_lds(const float* __restrict__ a, const float* __restrict__ e) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < BK; ++d) s = fmaf(a[d], e[d], s);
    return s;
}

// ---- forward ---------------------------------------------------------------------------------------------------------------------------
// Thread-per-ROW for everything that is a product with a 32- / 128-wide weight matrix (QKV, out-projection + residual + LayerNorm, the
// whole FFN + LayerNorm: a row's 12 k multiply-adds run out of registers against scalar-loaded weights, all L rows in parallel, no
// barrier inside) and thread-per-(row, head) for the attention (K / V in LDS, every lane reads the SAME key row: a broadcast; two passes
// over the keys — maximum, then exponentials + P V — and not one cross-lane operation).
//
// Launch structure (round 5, third form).  One workgroup per utterance for the WHOLE encoder left 128 of 256 CUs idle at batch 128 and
// one or two waves per SIMD on the others, with 796 (row, head) attention items queueing behind 256 - 512 threads: 0.74 - 1.0 ms forward,
// 1.7 - 2.0 ms backward, three quarters of it the attention loops.  The attention is now its own launch per layer with one workgroup
// per (utterance, HEAD): 4 x the workgroups, the head's K / V slices (16 L floats each) in LDS, every lane on the same key row.  The
// row phases between two attentions (the rest of layer l - 1, then Q / K / V of layer l out of the same registers) are one launch per
// utterance as before.  n_layers + 1 row launches and n_layers attention launches forward; the activations were in the per-utterance
// scratch row already, so the split adds no traffic.

// rows kernel `l` (0 .. n_layers): [embedding | the post-attention half of layer l - 1] -> x;  [Q, K, V of layer l | the read-out]
__global__ __launch_bounds__(NTR) void btse_rows_fwd_kernel(const SclBtseBio p, const int l) {
    const int b = blockIdx.x, t = threadIdx.x, L = p.L;
    const int len = min(max(p.lens[b], 0), L);            // commons.sequence_mask: arange(L) < length
    float* ws = p.ws + (int64_t)b * p.ws_stride;
    const int32_t* tok = p.bio + (int64_t)b * L;
    float* cur = ws + (int64_t)l * O_LAYER * L;           // layer l's block: its input rows first
    {
        const int r = blockIdx.y * NTR + t;
        if (r < L) {
        float x[BD];
        if (l == 0) {                                     // model.py:228,232 + transformer.py:42
            const int tk = min(max(tok[r], 0), p.n_bios - 1);
            load_row32(p.emb + tk * BD, x);
#pragma unroll
            for (int c = 0; c < BD; ++c) x[c] = r < len ? x[c] * EMB_SCALE : 0.f;
        } else {                                          // the rest of layer l - 1, one row per thread
            float* base = ws + (int64_t)(l - 1) * O_LAYER * L;
            const float* const* W = p.lw[l - 1];
            const bool valid = r < len;
            float a[BD], x1[BD], s[BD];
            load_row32(base + O_A * L + r * BD, a);
            matvec32(a, as_const(W[I_WO]), as_const(W[I_BO]), s);              // conv_o (transformer.py:146)
            load_row32(base + r * BD, a);
#pragma unroll
            for (int c = 0; c < BD; ++c) s[c] += a[c];      // x + y (transformer.py:46)
            float mean, rs;
            layernorm32(s, as_const(W[I_G1]), as_const(W[I_B1]), x1, mean, rs);
            store_row32(base + O_S1 * L + r * BD, s);
            store_row32(base + O_X1 * L + r * BD, x1);
            base[O_ST * L + r * 4] = mean; base[O_ST * L + r * 4 + 1] = rs;
            // FFN (transformer.py:283-291): 32 hidden units at a time — h = relu(conv_1(x * mask)), y += conv_2(h * mask)
            float y[BD];
#pragma unroll
            for (int c = 0; c < BD; ++c) { y[c] = valid ? as_const(W[I_C2])[c] : 0.f; a[c] = valid ? x1[c] : 0.f; }
#pragma unroll 1
            for (int f0 = 0; f0 < BF; f0 += BD) {
                float hf[BD];
                matvec32(a, as_const(W[I_W1]) + f0 * BD, as_const(W[I_C1]) + f0, hf);
#pragma unroll
                for (int j = 0; j < BD; ++j) hf[j] = fmaxf(hf[j], 0.f);
                store_row32(base + O_H * L + r * BF + f0, hf);
                if (valid) {
#pragma unroll 4
                    for (int c = 0; c < BD; ++c) {
                        float acc = y[c];
                        const cptr w2 = as_const(W[I_W2]) + c * BF + f0;
#pragma unroll
                        for (int j = 0; j < BD; ++j) acc = fmaf(hf[j], w2[j], acc);
                        y[c] = acc;
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < BD; ++c) s[c] = x1[c] + y[c];      // y is already zero on padded rows (... * x_mask, transformer.py:291)
            layernorm32(s, as_const(W[I_G2]), as_const(W[I_B2]), x, mean, rs);
            store_row32(base + O_S2 * L + r * BD, s);
            base[O_ST * L + r * 4 + 2] = mean; base[O_ST * L + r * 4 + 3] = rs;
            if (l == p.n_layers && r >= len) {              // transformer.py:51: the encoder output is masked
#pragma unroll
                for (int c = 0; c < BD; ++c) x[c] = 0.f;
            }
        }
        store_row32(cur + r * BD, x);
        if (l < p.n_layers) {                             // q, k, v = conv_{q,k,v}(x)   (transformer.py:139-141)
            const float* const* W = p.lw[l];
            float y[BD];
            matvec32(x, as_const(W[I_WQ]), as_const(W[I_BQ]), y); store_row32(cur + O_Q * L + r * BD, y);
            matvec32(x, as_const(W[I_WK]), as_const(W[I_BK]), y); store_row32(cur + O_K * L + r * BD, y);
            matvec32(x, as_const(W[I_WV]), as_const(W[I_BV]), y); store_row32(cur + O_V * L + r * BD, y);
        }
        }
    }
    if (l < p.n_layers || (int)blockIdx.y != (L - 1) / NTR) return;       // the read-out is the work of the block that owns row L - 1
    __syncthreads();                                                                  // row L - 1 was written by a thread of this block
    const bool mlast = L - 1 < len;                                                   // model.py:234-236
    for (int o = t; o < p.bio_out; o += NTR) {
        float y = 0.f;
        if (mlast) {
            y = p.bs[o];
#pragma unroll 8
            for (int k = 0; k < BD; ++k) y = fmaf(cur[(L - 1) * BD + k], p.Ws[o * BD + k], y);
        }
        p.out[(int64_t)b * p.out_ld + o] = y;
    }
}

// sum / max over the KS lanes of a row's group (adjacent lanes: two DPP-class shuffles)
__device__ __forceinline__ float quad_sum(float v) { v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); return v; }
__device__ __forceinline__ float quad_max(float v) { v = fmaxf(v, __shfl_xor(v, 1)); v = fmaxf(v, __shfl_xor(v, 2)); return v; }

// attention of layer l, head blockIdx.y of utterance blockIdx.x, RPB rows per block (blockIdx.z): KS = 4 adjacent lanes share a query row, lane s
// takes the keys j = s (mod 4) — the four lanes read four ADJACENT key rows of the head's LDS slice — and the row's maximum / sum / output
// are combined with two shuffles each.  4 x the threads of a thread-per-row launch: eight waves per SIMD hide the LDS and exp latencies that a
// 2 x 199-key loop behind two waves per SIMD exposed (95 -> see profiles/r5_btse_bio_probe_final.txt).   (transformer.py:148-186)
__global__ __launch_bounds__(NTA) void btse_attn_fwd_kernel(const SclBtseBio p, const int l) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float Eks[NR * BK], Evs[NR * BK];
    const int b = blockIdx.x, h = blockIdx.y, t = threadIdx.x, L = p.L;
    const int len = min(max(p.lens[b], 0), L);
    float* base = p.ws + (int64_t)b * p.ws_stride + (int64_t)l * O_LAYER * L;
    const float* const* W = p.lw[l];
    float* Kh = lds;                 // [L][8]
    float* Vh = lds + L * BK;
    if (t < NR * BK) { Eks[t] = W[I_EK][t]; Evs[t] = W[I_EV][t]; }
    for (int idx = t; idx < 2 * L; idx += NTA) {
        const int j = idx >> 1, half = idx & 1;
        ((float4*)(Kh + j * BK))[half] = ((const float4*)(base + O_K * L + j * BD + h * BK))[half];
        ((float4*)(Vh + j * BK))[half] = ((const float4*)(base + O_V * L + j * BD + h * BK))[half];
    }
    __syncthreads();
    const int i = blockIdx.z * RPB + (t >> 2), ks = t & 3;
    const int ir = min(i, L - 1);                         // the lanes of rows past L run the loops on row L - 1 (shuffles need whole groups), store nothing
    const bool mi = ir < len;
    float qs[BK];
    {
        const float4 q0 = ((const float4*)(base + O_Q * L + ir * BD + h * BK))[0], q1 = ((const float4*)(base + O_Q * L + ir * BD + h * BK))[1];
        qs[0] = q0.x * QSCALE; qs[1] = q0.y * QSCALE; qs[2] = q0.z * QSCALE; qs[3] = q0.w * QSCALE;
        qs[4] = q1.x * QSCALE; qs[5] = q1.y * QSCALE; qs[6] = q1.z * QSCALE; qs[7] = q1.w * QSCALE;
    }
    float mx = -INFINITY;
#pragma unroll 2
    for (int j = ks; j < L; j += KS) {
        const float s = (mi && j < len) ? score(qs, Kh + j * BK, Eks, j - ir) : FILL;
        mx = fmaxf(mx, s);
    }
    mx = quad_max(mx);
    float sum = 0.f, o[BK];
#pragma unroll
    for (int d = 0; d < BK; ++d) o[d] = 0.f;
#pragma unroll 2
    for (int j = ks; j < L; j += KS) {
        const int dj = j - ir;
        const float s = (mi && j < len) ? score(qs, Kh + j * BK, Eks, dj) : FILL;
        const float e = __expf(s - mx);
        sum += e;
        const float4 v0 = ((const float4*)(Vh + j * BK))[0], v1 = ((const float4*)(Vh + j * BK))[1];
        o[0] = fmaf(e, v0.x, o[0]); o[1] = fmaf(e, v0.y, o[1]); o[2] = fmaf(e, v0.z, o[2]); o[3] = fmaf(e, v0.w, o[3]);
        o[4] = fmaf(e, v1.x, o[4]); o[5] = fmaf(e, v1.y, o[5]); o[6] = fmaf(e, v1.z, o[6]); o[7] = fmaf(e, v1.w, o[7]);
        if ((unsigned)(dj + BW) <= 2u * BW) {
            const float* ev = Evs + (dj + BW) * BK;
#pragma unroll
            for (int d = 0; d < BK; ++d) o[d] = fmaf(e, ev[d], o[d]);
        }
    }
    sum = quad_sum(sum);
#pragma unroll
    for (int d = 0; d < BK; ++d) o[d] = quad_sum(o[d]);
    if (ks == 0 && i < L) {
        const float inv = 1.f / sum;
        float4* ao = (float4*)(base + O_A * L + i * BD + h * BK);
        ao[0] = make_float4(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
        ao[1] = make_float4(o[4] * inv, o[5] * inv, o[6] * inv, o[7] * inv);
        base[O_LSE * L + h * L + i] = mx + logf(sum);
    }
}

// ---- backward ---------------------------------------------------------------------------------------------------------------------------
// dW[c][k] = sum_r dy[r][c] in[r][k] (and db[c] = sum_r dy[r][c]) for this utterance: 64-row chunks of both operands staged in LDS, a
// thread owns output column kk and C K / 256 rows of dW in registers across the chunks.  Rows are summed in index order.
template <int C, int K>
__device__ __forceinline__ void wgrad_rows(const float* __restrict__ dy, const float* __restrict__ in, int L, float* __restrict__ dW, float* __restrict__ db,
                                           float* stage, int t) {
    constexpr int U = C * K / NT, G = NT / K;
    const int kk = t % K, cg = t / K;
    float acc[U], accb = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = 0.f;
    float* dyS = stage;
    float* inS = stage + RC * C;
    for (int r0 = 0; r0 < L; r0 += RC) {
        const int nr = min(RC, L - r0);
        __syncthreads();
        for (int idx = t; idx < nr * C / 4; idx += NT) ((float4*)dyS)[idx] = ((const float4*)(dy + (int64_t)r0 * C))[idx];
        for (int idx = t; idx < nr * K / 4; idx += NT) ((float4*)inS)[idx] = ((const float4*)(in + (int64_t)r0 * K))[idx];
        __syncthreads();
        for (int r = 0; r < nr; ++r) {
            const float ik = inS[r * K + kk];
#pragma unroll
            for (int u = 0; u < U; ++u) acc[u] = fmaf(dyS[r * C + cg + G * u], ik, acc[u]);
            if (db && t < C) accb += dyS[r * C + t];
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) dW[(cg + G * u) * K + kk] = acc[u];
    if (db && t < C) db[t] = accb;
}

// dgamma[c] = sum_r dy[r][c] xhat[r][c], dbeta[c] = sum_r dy[r][c] over this utterance's rows (NT / 32 row groups x 32 channels, combined in order)
__device__ __forceinline__ void ln_param_grads(const float* __restrict__ dy, const float* __restrict__ pre, const float* __restrict__ st, int so, int L,
                                               float* __restrict__ dg, float* __restrict__ dbeta, float* red, int t) {
    const int c = t & 31, rg = t >> 5;
    float a = 0.f, bsum = 0.f;
    for (int r = rg; r < L; r += NG) {
        const float d = dy[r * BD + c];
        a = fmaf(d, (pre[r * BD + c] - st[r * 4 + so]) * st[r * 4 + so + 1], a);
        bsum += d;
    }
    __syncthreads();
    red[rg * BD + c] = a; red[NG * BD + rg * BD + c] = bsum;
    __syncthreads();
    if (t < 2 * BD) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g) s += red[(t >> 5) * NG * BD + g * BD + (t & 31)];
        (t < BD ? dg : dbeta)[t & 31] = s;
    }
}

// Backward launch structure (the mirror of the forward's): per layer, top down,
//   rows kernel  : [the scoring conv's gradient | steps (4) + (6) of the layer above: relative-embedding and Q / K / V parameter gradients, input
//                  gradient]  then steps (0) - (2) of this layer: LayerNorm / FFN / out-projection data gradients, one row per thread, and their
//                  parameter gradients;
//   attention (3): workgroup per (utterance, head), thread per QUERY row -> dq, delta, the band of dS / P the relative embeddings need;
//   attention (5): workgroup per (utterance, head), thread per KEY row   -> dk, dv;
// and a last rows kernel for (4) + (6) of layer 0 and the embedding gradient.  2 n_layers attention + n_layers + 1 row launches.
__global__ __launch_bounds__(NT) void btse_rows_bwd_kernel(const SclBtseBio p, const int l6, const int l012) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float red[2 * NG * BD], epart[2][3][NR * BK];
    const int b = blockIdx.x, t = threadIdx.x, L = p.L;
    const int len = min(max(p.lens[b], 0), L);
    float* slab = p.slab + (int64_t)b * p.slab_ld;
    if (len < L) {      // the read-out position is padding: zero score, zero gradient (model.py:234-236)
        if (l6 < 0)
            for (int64_t idx = t; idx < p.slab_ld; idx += NT) slab[idx] = 0.f;
        return;
    }
    float* ws = p.ws + (int64_t)b * p.ws_stride;
    const int32_t* tok = p.bio + (int64_t)b * L;
    float* G = ws + ((int64_t)p.n_layers * O_LAYER + BD) * L;
    float* gdx = G + G_DX * L;
    const float* gda = G + G_DA * L;
    if (l6 < 0) {   // scoring conv at the last position
        const float* dsc = p.d_out + (int64_t)b * p.dout_ld;
        const float* xfin = ws + (int64_t)p.n_layers * O_LAYER * L + (L - 1) * BD;
        for (int o = t; o < p.bio_out; o += NT) slab[p.go_bs + o] = dsc[o];
        for (int idx = t; idx < p.bio_out * BD; idx += NT) slab[p.go_Ws + idx] = dsc[idx >> 5] * xfin[idx & 31];
        for (int idx = t; idx < (L - 1) * BD; idx += NT) gdx[idx] = 0.f;
        if (t < BD) {
            float a = 0.f;
            for (int o = 0; o < p.bio_out; ++o) a = fmaf(dsc[o], p.Ws[o * BD + t], a);
            gdx[(L - 1) * BD + t] = a;
        }
        __syncthreads();
    } else {
        float* base = ws + (int64_t)l6 * O_LAYER * L;
        const float* const* W = p.lw[l6];
        const int32_t* go = p.go[l6];
        // (4) relative-embedding gradients: d emb_rel_k[r][d] = sum_(i,h) dS[i][i + r - 4] qs_i[d], d emb_rel_v[r][d] = sum P[i][i + r - 4] dA_i[d];
        //     144 outputs x 3 row ranges, combined in order
        for (int wi = t; wi < 3 * 2 * NR * BK; wi += NT) {
            const int part = wi / (2 * NR * BK), e = wi % (2 * NR * BK), which = e / (NR * BK), rd = e % (NR * BK), r = rd >> 3, d = rd & 7;
            const int i0 = (L * part) / 3, i1 = (L * (part + 1)) / 3;
            const float* bsrc = G + (which ? G_BP : G_BS) * L;
            const float* vsrc = which ? gda : base + O_Q * L;
            const float sc = which ? 1.f : QSCALE;
            float a = 0.f;
            for (int i = i0; i < i1; ++i)
#pragma unroll
                for (int h = 0; h < BH; ++h) a = fmaf(bsrc[(i * BH + h) * NR + r], vsrc[i * BD + h * BK + d] * sc, a);
            epart[which][part][rd] = a;
        }
        __syncthreads();
        if (t < 2 * NR * BK) {
            const int which = t / (NR * BK), rd = t % (NR * BK);
            slab[go[which ? I_EV : I_EK] + rd] = (epart[which][0][rd] + epart[which][1][rd]) + epart[which][2][rd];
        }
        // (6) conv_q / conv_k / conv_v: parameter gradients; input gradient
        const float* xin = base;
        const float* ds1 = G + G_DS1 * L;
        wgrad_rows<BD, BD>(G + G_Q * L, xin, L, slab + go[I_WQ], slab + go[I_BQ], lds, t);
        wgrad_rows<BD, BD>(G + G_K * L, xin, L, slab + go[I_WK], slab + go[I_BK], lds, t);
        wgrad_rows<BD, BD>(G + G_V * L, xin, L, slab + go[I_WV], slab + go[I_BV], lds, t);
        __syncthreads();
        for (int r = t; r < L; r += NT) {
            float d[BD], y[BD];
            load_row32(ds1 + r * BD, y);                       // the residual branch
            load_row32(G + G_Q * L + r * BD, d); matvec32_t(d, as_const(W[I_WQ]), y);
            load_row32(G + G_K * L + r * BD, d); matvec32_t(d, as_const(W[I_WK]), y);
            load_row32(G + G_V * L + r * BD, d); matvec32_t(d, as_const(W[I_WV]), y);
            store_row32(gdx + r * BD, y);
        }
        __syncthreads();
    }
    if (l012 >= 0) {
        float* base = ws + (int64_t)l012 * O_LAYER * L;
        const float* const* W = p.lw[l012];
        const int32_t* go = p.go[l012];
        const float* st = base + O_ST * L;
        // (0) LayerNorm 2's parameter gradients need d x_out itself: before (1) re-uses the buffer
        ln_param_grads(gdx, base + O_S2 * L, st, 2, L, slab + go[I_G2], slab + go[I_B2], red, t);
        __syncthreads();
        // (1) one row per thread: LayerNorm 2, the FFN, LayerNorm 1, the out-projection — data gradients only
        for (int r = t; r < L; r += NT) {
            float dy[BD], s[BD], ds2[BD], dx1[BD];
            load_row32(gdx + r * BD, dy);
            load_row32(base + O_S2 * L + r * BD, s);
            layernorm32_bwd(dy, s, st[r * 4 + 2], st[r * 4 + 3], as_const(W[I_G2]), ds2);
            store_row32(G + G_DS2 * L + r * BD, ds2);
#pragma unroll
            for (int k = 0; k < BD; ++k) dx1[k] = ds2[k];      // the residual branch
#pragma unroll 1
            for (int f0 = 0; f0 < BF; f0 += BD) {
                float dh[BD], hf[BD];
                load_row32(base + O_H * L + r * BF + f0, hf);
#pragma unroll
                for (int j = 0; j < BD; ++j) dh[j] = 0.f;
#pragma unroll 4
                for (int c = 0; c < BD; ++c) {
                    const float dc = ds2[c];
                    const cptr w2 = as_const(W[I_W2]) + c * BF + f0;
#pragma unroll
                    for (int j = 0; j < BD; ++j) dh[j] = fmaf(dc, w2[j], dh[j]);
                }
#pragma unroll
                for (int j = 0; j < BD; ++j) dh[j] = hf[j] > 0.f ? dh[j] : 0.f;
                store_row32(G + G_DH * L + r * BF + f0, dh);
                matvec32_t(dh, as_const(W[I_W1]) + f0 * BD, dx1);
            }
            load_row32(base + O_S1 * L + r * BD, s);
            layernorm32_bwd(dx1, s, st[r * 4], st[r * 4 + 1], as_const(W[I_G1]), dy);      // dy := d s1
            store_row32(G + G_DS1 * L + r * BD, dy);
            store_row32(gdx + r * BD, dx1);                                        // kept for LayerNorm 1's parameter gradients
#pragma unroll
            for (int k = 0; k < BD; ++k) s[k] = 0.f;
            matvec32_t(dy, as_const(W[I_WO]), s);
            store_row32(G + G_DA * L + r * BD, s);
        }
        __syncthreads();
        const float* ds2 = G + G_DS2 * L;
        const float* ds1 = G + G_DS1 * L;
        // (2) parameter gradients of those layers
        wgrad_rows<BD, BF>(ds2, base + O_H * L, L, slab + go[I_W2], slab + go[I_C2], lds, t);
        wgrad_rows<BF, BD>(G + G_DH * L, base + O_X1 * L, L, slab + go[I_W1], slab + go[I_C1], lds, t);
        wgrad_rows<BD, BD>(ds1, base + O_A * L, L, slab + go[I_WO], slab + go[I_BO], lds, t);
        ln_param_grads(gdx, base + O_S1 * L, st, 0, L, slab + go[I_G1], slab + go[I_B1], red, t);
        return;
    }
    for (int idx = t; idx < p.n_bios * BD; idx += NT) {      // embedding rows (model.py:228)
        const int tk = idx >> 5, cc = idx & 31;
        float a = 0.f;
        for (int r = 0; r < L; ++r)
            if (min(max(tok[r], 0), p.n_bios - 1) == tk) a += gdx[r * BD + cc];
        slab[p.go_emb + idx] = a * EMB_SCALE;
    }
}

// (3) attention backward, pass 1 of layer l: head blockIdx.y of utterance blockIdx.x, four lanes per QUERY row (keys j = s mod 4, as the forward)
//     -> dq, delta, the band of dS / P
__global__ __launch_bounds__(NTA) void btse_attn_bwd_q_kernel(const SclBtseBio p, const int l) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float Eks[NR * BK], Evs[NR * BK];
    const int b = blockIdx.x, h = blockIdx.y, t = threadIdx.x, L = p.L;
    if (min(max(p.lens[b], 0), L) < L) return;
    float* ws = p.ws + (int64_t)b * p.ws_stride;
    float* base = ws + (int64_t)l * O_LAYER * L;
    const float* const* W = p.lw[l];
    float* G = ws + ((int64_t)p.n_layers * O_LAYER + BD) * L;
    const float* gda = G + G_DA * L;
    float* Kh = lds;                 // [L][8]
    float* Vh = lds + L * BK;
    if (t < NR * BK) { Eks[t] = W[I_EK][t]; Evs[t] = W[I_EV][t]; }
    for (int idx = t; idx < 2 * L; idx += NTA) {
        const int j = idx >> 1, half = idx & 1;
        ((float4*)(Kh + j * BK))[half] = ((const float4*)(base + O_K * L + j * BD + h * BK))[half];
        ((float4*)(Vh + j * BK))[half] = ((const float4*)(base + O_V * L + j * BD + h * BK))[half];
    }
    __syncthreads();
    const int i = blockIdx.z * RPB + (t >> 2), ks = t & 3;
    const int ir = min(i, L - 1);
    float qs[BK], da[BK], dq[BK];
#pragma unroll
    for (int d = 0; d < BK; ++d) { qs[d] = base[O_Q * L + ir * BD + h * BK + d] * QSCALE; da[d] = gda[ir * BD + h * BK + d]; dq[d] = 0.f; }
    const float lse = base[O_LSE * L + h * L + ir];
    float delta = 0.f;
#pragma unroll 2
    for (int j = ks; j < L; j += KS) {
        const int dj = j - ir;
        const float pij = __expf(score(qs, Kh + j * BK, Eks, dj) - lse);
        delta = fmaf(pij, score(da, Vh + j * BK, Evs, dj), delta);
    }
    delta = quad_sum(delta);
    float* bS = G + G_BS * L + (ir * BH + h) * NR;
    float* bP = G + G_BP * L + (ir * BH + h) * NR;
    if (i < L) {
        // band entry r belongs to key j = i + r - 4 and is written by that key's lane; entries whose key lies outside the sequence are zero
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int j = i + r - BW;
            if (((j & 3) == ks) && (j < 0 || j >= L)) { bS[r] = 0.f; bP[r] = 0.f; }
        }
    }
#pragma unroll 2
    for (int j = ks; j < L; j += KS) {
        const int dj = j - ir;
        const float* kr = Kh + j * BK;
        const float pij = __expf(score(qs, kr, Eks, dj) - lse);
        const float dS = pij * (score(da, Vh + j * BK, Evs, dj) - delta);
#pragma unroll
        for (int d = 0; d < BK; ++d) dq[d] = fmaf(dS, kr[d], dq[d]);
        if ((unsigned)(dj + BW) <= 2u * BW) {
            const float* ek = Eks + (dj + BW) * BK;
#pragma unroll
            for (int d = 0; d < BK; ++d) dq[d] = fmaf(dS, ek[d], dq[d]);
            if (i < L) { bS[dj + BW] = dS; bP[dj + BW] = pij; }
        }
    }
#pragma unroll
    for (int d = 0; d < BK; ++d) dq[d] = quad_sum(dq[d]);
    if (ks == 0 && i < L) {
#pragma unroll
        for (int d = 0; d < BK; ++d) G[G_Q * L + i * BD + h * BK + d] = dq[d] * QSCALE;
        G[G_DELTA * L + h * L + i] = delta;
    }
}

// (5) attention backward, pass 2: four lanes per KEY row (queries i = s mod 4) -> dk, dv (the head's Q * scale, dA, log-sum-exp and delta rows in LDS)
__global__ __launch_bounds__(NTA) void btse_attn_bwd_kv_kernel(const SclBtseBio p, const int l) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float Eks[NR * BK], Evs[NR * BK];
    const int b = blockIdx.x, h = blockIdx.y, t = threadIdx.x, L = p.L;
    if (min(max(p.lens[b], 0), L) < L) return;
    float* ws = p.ws + (int64_t)b * p.ws_stride;
    float* base = ws + (int64_t)l * O_LAYER * L;
    const float* const* W = p.lw[l];
    float* G = ws + ((int64_t)p.n_layers * O_LAYER + BD) * L;
    const float* gda = G + G_DA * L;
    float* Qh = lds;                 // [L][8], scaled
    float* Ah = lds + L * BK;        // [L][8]
    float* lseS = lds + 2 * L * BK;  // [L]
    float* delS = lseS + L;          // [L]
    if (t < NR * BK) { Eks[t] = W[I_EK][t]; Evs[t] = W[I_EV][t]; }
    for (int idx = t; idx < 2 * L; idx += NTA) {
        const int i = idx >> 1, half = idx & 1;
        const float4 q = ((const float4*)(base + O_Q * L + i * BD + h * BK))[half];
        ((float4*)(Qh + i * BK))[half] = make_float4(q.x * QSCALE, q.y * QSCALE, q.z * QSCALE, q.w * QSCALE);
        ((float4*)(Ah + i * BK))[half] = ((const float4*)(gda + i * BD + h * BK))[half];
    }
    for (int i = t; i < L; i += NTA) { lseS[i] = base[O_LSE * L + h * L + i]; delS[i] = G[G_DELTA * L + h * L + i]; }
    __syncthreads();
    const int j = blockIdx.z * RPB + (t >> 2), ks = t & 3;
    const int jr = min(j, L - 1);
    float kj[BK], vj[BK], dk[BK], dv[BK];
#pragma unroll
    for (int d = 0; d < BK; ++d) { kj[d] = base[O_K * L + jr * BD + h * BK + d]; vj[d] = base[O_V * L + jr * BD + h * BK + d]; dk[d] = 0.f; dv[d] = 0.f; }
#pragma unroll 2
    for (int i = ks; i < L; i += KS) {
        const int dj = jr - i;
        const float* qr = Qh + i * BK;
        const float* ar = Ah + i * BK;
        const bool band = (unsigned)(dj + BW) <= 2u * BW;
        const float sij = dot8(qr, kj) + (band ? dot8_lds(qr, Eks + (dj + BW) * BK) : 0.f);      // q_i . (k_j + emb_rel_k[j - i + 4])
        const float g = dot8(ar, vj) + (band ? dot8_lds(ar, Evs + (dj + BW) * BK) : 0.f);         // dA_i . (v_j + emb_rel_v[j - i + 4])
        const float pij = __expf(sij - lseS[i]);
        const float dS = pij * (g - delS[i]);
#pragma unroll
        for (int d = 0; d < BK; ++d) { dk[d] = fmaf(dS, qr[d], dk[d]); dv[d] = fmaf(pij, ar[d], dv[d]); }
    }
#pragma unroll
    for (int d = 0; d < BK; ++d) { dk[d] = quad_sum(dk[d]); dv[d] = quad_sum(dv[d]); }
    if (ks == 0 && j < L) {
#pragma unroll
        for (int d = 0; d < BK; ++d) { G[G_K * L + j * BD + h * BK + d] = dk[d]; G[G_V * L + j * BD + h * BK + d] = dv[d]; }
    }
}

__global__ void btse_join_fwd_kernel(const float* __restrict__ emb, const float* __restrict__ s, const float* __restrict__ W1, const float* __restrict__ b1,
                                     float* __restrict__ out, int B, int C, int bo, int is_add) {
    const int r = blockIdx.x;
    if (!is_add) {
        for (int cc = threadIdx.x; cc < C; cc += blockDim.x) out[(int64_t)r * (C + bo) + cc] = emb[(int64_t)r * C + cc];
        return;
    }
    for (int o = threadIdx.x; o < bo; o += blockDim.x) {
        float a = b1[o];
        for (int cc = 0; cc < C; ++cc) a = fmaf(emb[(int64_t)r * C + cc], W1[(int64_t)o * C + cc], a);
        out[(int64_t)r * bo + o] = a + s[(int64_t)r * bo + o];
    }
}

__global__ void btse_join_bwd_kernel(const float* __restrict__ db, const float* __restrict__ emb, const float* __restrict__ W1, float* __restrict__ demb,
                                     float* __restrict__ ds, float* __restrict__ dW1, float* __restrict__ db1, int B, int C, int bo, int is_add) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    if (!is_add) {
        for (int idx = t; idx < B * (C + bo); idx += nth) {
            const int r = idx / (C + bo), cc = idx % (C + bo);
            if (cc < C) demb[(int64_t)r * C + cc] = db[idx];
            else ds[(int64_t)r * bo + cc - C] = db[idx];
        }
        return;
    }
    for (int idx = t; idx < B * bo; idx += nth) ds[idx] = db[idx];
    for (int idx = t; idx < B * C; idx += nth) {
        const int r = idx / C, cc = idx % C;
        float a = 0.f;
        for (int o = 0; o < bo; ++o) a = fmaf(db[(int64_t)r * bo + o], W1[(int64_t)o * C + cc], a);
        demb[idx] = a;
    }
    for (int idx = t; idx < bo * C; idx += nth) {
        const int o = idx / C, cc = idx % C;
        float a = 0.f;
        for (int r = 0; r < B; ++r) a = fmaf(db[(int64_t)r * bo + o], emb[(int64_t)r * C + cc], a);
        dW1[idx] = a;
    }
    for (int o = t; o < bo; o += nth) {
        float a = 0.f;
        for (int r = 0; r < B; ++r) a += db[(int64_t)r * bo + o];
        db1[o] = a;
    }
}

int bio_check(const SclBtseBio* p, const char* what, bool bwd) {
    SCL_REQUIRE(p, "%s: null descriptor", what);
    if (!scl_btse_bio_supported(p->bio_dim, p->n_heads, p->pf_dim, p->n_layers, p->window, p->bio_out, p->L)) {
        scl_set_error("%s: unsupported shape (bio_dim %d, heads %d, pf_dim %d, layers %d, window %d, bio_out %d, L %d): the kernel serves 32 / 4 / 128 / "
                      "1..8 / 4 / <= 256 / 1..%d", what, p->bio_dim, p->n_heads, p->pf_dim, p->n_layers, p->window, p->bio_out, p->L, MAXL);
        return SCL_EUNSUPPORTED;
    }
    SCL_REQUIRE(p->B > 0 && p->n_bios > 0 && p->emb && p->Ws && p->bs && p->bio && p->lens && p->ws, "%s: null pointer or empty batch", what);
    SCL_REQUIRE(p->ws_stride >= scl_btse_bio_ws_floats(p->n_layers, p->L), "%s: ws_stride %lld < %lld floats", what, (long long)p->ws_stride,
                (long long)scl_btse_bio_ws_floats(p->n_layers, p->L));
    for (int l = 0; l < p->n_layers; ++l)
        for (int i = 0; i < 18; ++i) SCL_REQUIRE(p->lw[l][i], "%s: layer %d tensor %d is null", what, l, i);
    if (!bwd) { SCL_REQUIRE(p->out && p->out_ld >= p->bio_out, "%s: bad output", what); }
    else { SCL_REQUIRE(p->d_out && p->slab && p->dout_ld >= p->bio_out && p->slab_ld > 0, "%s: bad gradient buffers", what); }
    return SCL_OK;
}
}  // namespace

extern "C" int scl_btse_bio_supported(int bio_dim, int n_heads, int pf_dim, int n_layers, int window, int bio_out, int L) {
    return bio_dim == BD && n_heads == BH && pf_dim == BF && window == BW && n_layers >= 1 && n_layers <= 8 && bio_out >= 1 && bio_out <= 256 &&
           L >= 1 && L <= MAXL;
}
extern "C" int64_t scl_btse_bio_ws_floats(int n_layers, int L) { return ((int64_t)n_layers * O_LAYER + BD + G_TOTAL) * L; }

extern "C" int scl_btse_bio_fwd(const SclBtseBio* p, void* stream) {
    const int rc = bio_check(p, "btse_bio_fwd", false);
    if (rc != SCL_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)2 * p->L * BK * sizeof(float);      // one head's K and V slices
    for (int l = 0; l <= p->n_layers; ++l) {
        hipLaunchKernelGGL(btse_rows_fwd_kernel, dim3(p->B, (p->L + NTR - 1) / NTR), dim3(NTR), 0, s, *p, l);
        if (l < p->n_layers) hipLaunchKernelGGL(btse_attn_fwd_kernel, dim3(p->B, BH, (p->L + RPB - 1) / RPB), dim3(NTA), lds, s, *p, l);
    }
    return scl_check_launch("scl_btse_bio_fwd");
}
extern "C" int scl_btse_bio_bwd(const SclBtseBio* p, void* stream) {
    const int rc = bio_check(p, "btse_bio_bwd", true);
    if (rc != SCL_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const size_t stage = (size_t)RC * (BD + BF) * sizeof(float);                 // the widest weight-gradient staging chunk
    const size_t lds_q = (size_t)2 * p->L * BK * sizeof(float), lds_kv = (size_t)(2 * p->L * BK + 2 * p->L) * sizeof(float);
    for (int l = p->n_layers - 1; l >= 0; --l) {
        hipLaunchKernelGGL(btse_rows_bwd_kernel, dim3(p->B), dim3(NT), stage, s, *p, l + 1 < p->n_layers ? l + 1 : -1, l);
        hipLaunchKernelGGL(btse_attn_bwd_q_kernel, dim3(p->B, BH, (p->L + RPB - 1) / RPB), dim3(NTA), lds_q, s, *p, l);
        hipLaunchKernelGGL(btse_attn_bwd_kv_kernel, dim3(p->B, BH, (p->L + RPB - 1) / RPB), dim3(NTA), lds_kv, s, *p, l);
    }
    hipLaunchKernelGGL(btse_rows_bwd_kernel, dim3(p->B), dim3(NT), stage, s, *p, 0, -1);
    return scl_check_launch("scl_btse_bio_bwd");
}
extern "C" int scl_btse_join_fwd(const float* emb, const float* s, const float* W1, const float* b1, float* b, int B, int C, int bio_out, int is_add,
                                 void* stream) {
    SCL_REQUIRE(emb && b && B > 0 && B <= 4096 && C > 0 && bio_out > 0, "btse_join_fwd: bad args");
    SCL_REQUIRE(!is_add || (s && W1 && b1), "btse_join_fwd: is_add needs s, W1, b1");
    hipLaunchKernelGGL(btse_join_fwd_kernel, dim3(B), dim3(128), 0, (hipStream_t)stream, emb, s, W1, b1, b, B, C, bio_out, is_add);
    return scl_check_launch("scl_btse_join_fwd");
}
extern "C" int scl_btse_join_bwd(const float* db, const float* emb, const float* W1, float* demb, float* ds, float* dW1, float* db1, int B, int C,
                                 int bio_out, int is_add, void* stream) {
    SCL_REQUIRE(db && demb && ds && B > 0 && B <= 4096 && C > 0 && bio_out > 0, "btse_join_bwd: bad args");
    SCL_REQUIRE(!is_add || (emb && W1 && dW1 && db1), "btse_join_bwd: is_add needs emb, W1, dW1, db1");
    hipLaunchKernelGGL(btse_join_bwd_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, db, emb, W1, demb, ds, dW1, db1, B, C, bio_out, is_add);
    return scl_check_launch("scl_btse_join_bwd");
}
