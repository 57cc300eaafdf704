/*
 * scl_hip.h — C ABI of libscl_hip.so, the MI355X (gfx950) kernel library behind the
 * SCL-Deepfake-audio-detection training hot path.
 *
 * Every entry point replaces a PyTorch / fairseq / scipy / numpy library call that the reference
 * makes on its hot path; the citation after each declaration is the reference call site
 * (path:line relative to the reference repo root) whose arithmetic the entry point reproduces.
 *
 * Conventions:
 *   - return 0 on success, a negative SCL_E* code on error (never throws / aborts);
 *     scl_last_error() returns a thread-local message for the last failure;
 *   - no hidden allocation: the caller owns every buffer (device pointers unless noted);
 *   - all work is enqueued on the caller's `stream` (a hipStream_t passed as void*), no
 *     implicit device synchronisation;
 *   - re-entrant; one thread per device;
 *   - "bf16" = raw uint16 storage of bfloat16; "f32" = float; sizes in elements unless noted.
 */
#ifndef SCL_HIP_H
#define SCL_HIP_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCL_OK            0
#define SCL_EINVAL       -1   /* bad argument (shape, alignment, flag combination) */
#define SCL_ELAUNCH      -2   /* hipLaunch / runtime error */
#define SCL_EUNSUPPORTED -3

/* ------------------------------------------------------------------------------------------ */
/* library                                                                                     */
/* ------------------------------------------------------------------------------------------ */
int         scl_version(void);
const char* scl_last_error(void);

/* Stream ordering: work submitted to `waiter` after this call starts only when everything submitted to `signaler` before it has
 * completed (one event record + one stream wait, events from an internal ring).  The encoder backward uses it to run the weight-
 * gradient GEMMs and bias-gradient column sums on a second stream beside the data-gradient chain (autograd's engine does the same
 * for independent branches of main.py:79's loss.backward() only when the user asks for streams; here it is explicit). */
int scl_stream_wait_stream(void* waiter, void* signaler);

/* Per-kernel HIP-event profiling used by bench.py's roofline leg: when enabled, every launch of
 * the kernel family `kid` (SCL_KID_*) is bracketed by two hipEvents on the launch stream.
 * scl_prof_read synchronises those events and returns launch count and summed milliseconds. */
#define SCL_KID_GEMM 0
#define SCL_KID_GEMM_F32 2   /* the f32-operand GEMM (scoring path, AASIST / ResNet back-ends) */
#define SCL_KID_AUG  1   /* the RawBoost chain: scl_fir_multi_f32, scl_clip_stats_f32, scl_isd_scatter_f32, scl_clip_affine_f32 */
#define SCL_KID_MAX  8
/* bit 0: the library was built with -DSCL_EXPERIMENTS (the opt-in GEMM experiments: 256x128 / 256x256 ping-pong tiles, two workgroups per
 * CU, persistent blocks, start stagger).  The shipped library is built without; their flags / environment switches are then ignored. */
int scl_build_flags(void);
int scl_prof_enable(int kid, int on);
/* create n_pairs event pairs ahead of time: the first profiled step would otherwise pay hipEventCreate for every launch inside the
 * timed region (bench.py: 335 launches, ~30 ms) */
int scl_prof_reserve(int kid, int n_pairs);
int scl_prof_read(int kid, int64_t* n_launches, double* total_ms, double* total_flops);
/* The same launches one by one, in issue order (call it before scl_prof_read, which recycles the events): ms[i] and eight int32 per
 * launch — M, N, K, flags, batch x split-K, kernel variant (0: 128 x 128, 1 / 2: wide ping-pong / single-barrier, 3: two-per-CU,
 * 4: grouped weight gradients [M, N, K of the first member, flags = tiles], 5: pos-conv, 6: the 112-row wide tile), 0, 0.  *n_launches = how many there are
 * (also when cap is smaller).  tools/gemm_classes.py builds profiles/r6_gemm_classes.txt from it. */
int scl_prof_read_launches(int kid, int cap, float* ms, int32_t* meta8, int64_t* n_launches);

/* ------------------------------------------------------------------------------------------ */
/* generic batched bf16 MFMA contraction                                                       */
/* ------------------------------------------------------------------------------------------ */
/* One 2-D bf16 operand.  `rows` are M (for A) / N (for B) when the operand is K-contiguous, or
 * the reduction index K when the operand is transposed (flag SCL_GEMM_A_T / _B_T); the other
 * index runs along memory.
 *   element offset(row r, contiguous index c) =
 *       (r / rpb) * rbstride + (r % rpb) * ld + (c / cin) * cout + (c % cin)
 * rpb/rbstride express "rows of utterance b start at b*rbstride" (conv im2col without a copy:
 * ld = stride*C < K makes successive rows overlap); cin/cout express a 2-level contiguous index
 * (grouped positional conv: k = (tap j, channel ci) -> j*1024 + ci).                          */
typedef struct SclOperand {
    const void* ptr;
    int64_t bs1, bs2;      /* batch strides for z1 = z / nb2 and z2 = z % nb2 (elements) */
    int64_t rbstride;
    int64_t cout;
    int32_t rpb;           /* >= 1; 0x7fffffff = flat */
    int32_t ld;
    int32_t cin;           /* multiple of 8; 0x7fffffff = flat */
    int32_t _pad;
} SclOperand;

#define SCL_GEMM_A_T      0x00000001  /* A stored [K rows][M contiguous] */
#define SCL_GEMM_B_T      0x00000002  /* B stored [K rows][N contiguous] (e.g. W[N,K] used for dgrad) */
#define SCL_GEMM_C_F32    0x00000004  /* C is f32 (else bf16) */
#define SCL_GEMM_C2_F32   0x00000008
#define SCL_GEMM_R_F32    0x00000010
#define SCL_GEMM_HAS_BIAS 0x00000020
#define SCL_GEMM_HAS_C2   0x00000040  /* also store the pre-activation value */
#define SCL_GEMM_DROPOUT  0x00000080  /* multiply by keep-mask(seed,row,col)/(1-p) after act / grad-mul */
#define SCL_GEMM_NO_DMA   0x00100000  /* force the register-staged kernel (testing / A-B comparison) */
#define SCL_GEMM_NO_BIG   0x00200000  /* never the 256x128 / 3-stage variant */
#define SCL_GEMM_FORCE_BIG 0x04000000 /* pick the 256x128 / 3-stage variant (A-B comparison; not chosen automatically) */
#define SCL_GEMM_NO_P8    0x00400000  /* do not pick the 256x256 ping-pong variant */
#define SCL_GEMM_FORCE_P8 0x00800000  /* pick it whenever it is legal (testing / A-B comparison) */
#define SCL_GEMM_NO_W8    0x01000000  /* never the wide-tile (<=208/256 x 256, runtime row pitch) ping-pong kernel of gemm_w8.hip */
#define SCL_GEMM_F32X3    0x20000000  /* with SCL_GEMM_AB_F32: every f32 operand as a (hi, lo) bf16 pair, three bf16 MFMAs per product term (gemm_f32.hip): ~2e-5 relative per product, ~5x the matrix-core rate of the exact kernel */
#define SCL_GEMM_FORCE_W8 0x02000000  /* pick it whenever it can address the operands (testing / A-B comparison) */
#define SCL_GEMM_FORCE_X2 0x08000000  /* pick the two-blocks-per-CU 208 x 128 kernel of gemm_x2.hip whenever it can address the operands */
#define SCL_GEMM_NO_X2    0x10000000  /* never that kernel (testing / A-B comparison) */
#define SCL_GEMM_AB_F32   0x40000000  /* A and B are f32 (strides in f32 elements, multiples of 4): exact-fp32 MFMA kernel (gemm_f32.hip) */
#define SCL_GEMM_C_SPLIT3 0x80000000u /* bf16 C, wide tiles only: the stored value v leaves as the row image [hi | hi | lo] of three N-wide planes (ldc = 3 N, hi = bf16(v),
                                         lo = bf16(v - hi)) — the left operand of the NEXT triple-plane GEMM (scl_split3_f32_bf16) without an f32 round trip; refused
                                         (SCL_EUNSUPPORTED) when the launch would not run on the wide-tile kernel */
#define SCL_GEMM_STAMPS   0x20000000  /* diagnostic: the wide kernels record per-block time stamps (scl_debug_gemm_stamps) */
#define SCL_GEMM_ACT_SHIFT   8        /* 0 none, 1 gelu(erf), 2 relu, 3 leaky_relu(0.01), 5 gelu(erf) with C2 = gelu'(pre-activation) instead
                                       * of the pre-activation (the backward multiplies by it: RACT 4) */
#define SCL_GEMM_RMODE_SHIFT 12       /* 0 none, 1 C += R, 2 C *= act'(R) with act = RACT (RACT 4: C *= R, R holds the derivative) */
#define SCL_GEMM_RACT_SHIFT  16

typedef struct SclGemmDesc {
    SclOperand A, B;
    void*        C;
    void*        C2;
    const void*  R;
    const float* bias;           /* indexed by column n (+ z2 * bias_bs2) */
    int64_t c_bs1, c_bs2;        /* C / C2 / R batch strides (elements) */
    int64_t c_rbstride;
    int64_t c_split_stride;      /* split-K slab stride (elements); slabs are summed by scl_reduce_slabs */
    int64_t bias_bs2;
    int32_t c_rpb, ldc;
    int32_t M, N, K;
    int32_t nb1, nb2, splitk;
    int32_t flags;
    float    alpha;
    float    drop_p;
    uint32_t drop_seed;
    int32_t  _pad;
    /* optional, wide-tile kernels only (scl_gemm_colsum_rows() > 0): every tile also writes the column sums of the f32 values it
     * stores to C into colsum_part[(tile_row * 4 + s) * N + col], s = 0..3 (two wave rows x two row passes); the caller sums the
     * scl_gemm_colsum_rows() partial rows in a fixed order (scl_colreduce_f32).  The bias gradient of a Linear is the column sum of
     * its output gradient (autograd of F.linear, main.py:79): produced here by the GEMM that writes that gradient, instead of a
     * second pass over it. */
    float*   colsum_part;
} SclGemmDesc;

/* C[z][m][n] = epilogue( alpha * sum_k A[z][m][k] * B[z][n][k] ), fp32 accumulate on MFMA.
 * Replaces: F.linear / nn.Conv1d / torch.bmm calls inside fairseq Wav2Vec2Model.forward
 * (model/xlsr.py:41), nn.Linear in model/wav2vec2_linear_nll.py:107,49-67 and their autograd
 * backward (main.py:79).                                                                       */
int scl_gemm_bf16(const SclGemmDesc* desc, void* stream);
/* Up to 8 independent weight-gradient contractions in ONE launch: every member is C = A^T B (SCL_GEMM_A_T | SCL_GEMM_B_T, plain f32 output,
 * flat K rows, K % 64 == 0 and >= 192, no batching, no split-K, M and N multiples of 8, 16-byte aligned C).  One 8-wave block per 256 x 256
 * output tile walks the WHOLE reduction and stores the finished tile: the four weight gradients of a transformer layer (autograd backward of
 * fairseq's TransformerSentenceEncoderLayer reached from model/xlsr.py:41; 16 + 48 + 64 + 64 tiles at E = 1024, F = 4096) need neither
 * split-K slabs nor a reduction pass.  scl_gemm_bf16_group_ok: 1 if the list qualifies (nothing is launched); scl_gemm_bf16_group returns
 * SCL_EUNSUPPORTED for a list that does not. */
int scl_gemm_bf16_group_ok(const SclGemmDesc* descs, int n);
int scl_gemm_bf16_group(const SclGemmDesc* descs, int n, void* stream);
/* The same with every member restricted to a RANGE of its problem's 256 x 256 output tiles: member i computes tiles [tile0[i], tile0[i] +
 * ntile[i]) of the scl_gemm_bf16_group_tiles(&descs[i]) tiles of its problem (ids in the kernel's own tile order; any partition of [0, tiles)
 * over several launches covers every output element exactly once).  Lets a caller carry tiles over so that each launch is a whole round of
 * the 256 CUs: the encoder backward issues 3 launches of 256 tiles per 4 layers instead of 4 of 192 (scl_amd/encoder.py).
 * scl_gemm_bf16_group_tiles: 0 if the descriptor does not qualify as a member. */
int scl_gemm_bf16_group_tiles(const SclGemmDesc* desc);
int scl_gemm_bf16_group_part(const SclGemmDesc* descs, const int32_t* tile0, const int32_t* ntile, int n, void* stream);

/* 0 when scl_gemm_bf16 would run this descriptor on the 128x128 tiles, 1 / 2 for the wide (<= 208 / 256 rows x 256 columns) tiles of
 * gemm_w8.hip — lets the caller size split-K for the tile that will actually be used (pointers are not dereferenced). */
int scl_gemm_uses_wide_tiles(const SclGemmDesc* desc);
/* number of partial rows a launch of this descriptor writes to colsum_part (tile rows x 4), or 0 when it would not run on a kernel
 * that can (then colsum_part must be null: scl_gemm_bf16 refuses it). */
int scl_gemm_colsum_rows(const SclGemmDesc* desc);

/* Grouped positional convolution of the encoder (fairseq ConvPositionalEmbedding behind model/xlsr.py:41) as an implicit GEMM with the
 * utterance's padded input resident in LDS (csrc/posconv.hip).  xpad: bf16 [B][T + K][G * Cg], output row t of utterance b reads rows
 * t .. t + K - 1; w: bf16 [G][Cg out][K * Cg] with k = tap * Cg + in (scl_posconv_weight_pack's forward or data-gradient image);
 * fwd != 0: C = gelu(conv + bias) + R, c2 = bf16(conv + bias) (the pre-activation the backward needs); fwd == 0: C = conv + R
 * (bias / c2 unused).  C, R: f32 [B * T][G * Cg].  Bit-identical to the same contraction through scl_gemm_bf16.
 * scl_posconv_supported: 1 when the shape can take this kernel (Cg = 64, T <= 208, even K <= 128), else the caller uses the GEMM. */
int scl_posconv_supported(int T, int K, int G, int Cg);
/* Weight gradient of the same convolution: dw[g][o][tap * Cg + c] = sum_b sum_t dypad[b][dy_row0 + t][g * Cg + o] * xpad[b][t + tap][g * Cg + c]
 * (f32, overwritten; both inputs bf16 [B][T + K][G * Cg]; rows dy_row0 .. dy_row0 + T - 1 of dypad hold the output gradient).  One
 * workgroup per (group, 8 taps) walks the utterances with the accumulators resident.  supported: Cg = 64, T <= 224, K % 8 == 0. */
int scl_posconv_wgrad_supported(int T, int K, int G, int Cg);
int scl_posconv_wgrad(const void* dypad, int dy_row0, const void* xpad, float* dw, int B, int T, int K, int G, int Cg, void* stream);
int scl_posconv_mfma(const void* xpad, const void* w, float* C, const float* bias, void* c2, const float* R, int B, int T, int K, int G,
                     int Cg, int fwd, void* stream);

/* diagnostic: copy the per-block stamps of the last SCL_GEMM_STAMPS launch: 8 x u64 per block = {realtime (100 MHz), shader
 * clock} at kernel entry, after the prologue, after the K loop, after the epilogue (blocks 0 .. nblocks-1, nblocks <= 4096). */
int scl_debug_gemm_stamps(unsigned long long* out, int nblocks);
/* diagnostic: launches so far (this process) that took the persistent wide-tile kernel (gemm_w8.hip, w8p: one resident block per CU
 * walks several tiles, next tile's first K stage requested before the epilogue).  Environment SCL_GEMM_PERSIST: 0 (default) never, 1
 * when a launch has more than one round of tiles, 8 .. 256 = that many resident blocks whenever the kernel is legal (tests). */
long long scl_debug_gemm_persistent_launches(void);

/* Split-K with the epilogue kept: run `desc` as a plain split-K GEMM into f32 slabs ([nslabs][M][N], slab stride `stride` elements:
 * same A / B, C = slabs, ldc = N, splitk = nslabs, no epilogue flags, alpha as in desc), then this pass stores
 * desc.C (and C2) = epilogue(sum_s slab_s) exactly as scl_gemm_bf16(desc) would have (alpha already applied by the partial GEMMs).
 * For un-batched problems (nb1 = nb2 = 1) with few output tiles and a long K — the N = 1024 linears of a pack-sized train step. */
int scl_gemm_splitk_finish(const SclGemmDesc* desc, const float* slabs, int nslabs, int64_t stride, void* stream);

/* out[i] = sum_s slabs[s*stride + i]  (deterministic split-K combine). */
int scl_reduce_slabs_f32(const float* slabs, float* out, int64_t n, int nslabs, int64_t stride, void* stream);
/* up to SCL_SLAB_MAX_JOBS such combines in ONE launch, each with the summation order of scl_reduce_slabs_f32 (bit-identical): the four
 * split-K weight gradients of an encoder layer (fc2, fc1, out_proj, q/k/v of the fairseq layer reached from model/xlsr.py:41) are
 * combined by one launch at the end of the layer's backward instead of one launch behind every weight-gradient GEMM. */
#define SCL_SLAB_MAX_JOBS 8
typedef struct SclSlabJob {
    const float* slabs;
    float*       out;
    int64_t      n, stride;
    int32_t      nslabs, _pad;
} SclSlabJob;
int scl_reduce_slabs_multi(const SclSlabJob* jobs, int njobs, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* AASIST / ResNet back-end pieces over channels-last fp32 maps (csrc/nn.hip)                  */
/* ------------------------------------------------------------------------------------------ */
/* BatchNorm over the rows of x [N, C] (C a power of two <= 512) fused with an activation (act: 0 none, 1 ReLU, 2 SELU).
 * training != 0: batch statistics (part: scratch of scl_bn_nslabs(N) * 2 * C DOUBLES = 16 * nslabs * C bytes, 8-byte aligned), running_mean / running_var / num_batches_tracked
 * updated like torch (momentum, unbiased variance); else the running statistics are used.  Saves mean / rstd [C].  Writes
 * y [N, C] f32 (may be NULL) and / or y2: element (row r = (b,i,j), c) at y2[m_base + b*m_bs + i*m_rs + j*m_cs + c] with
 * i = (r % m_HW) / m_W, j = r % m_W — the interior of the zero-padded map the next convolution reads (f32, or bf16 when y2_bf16).
 * Replaces nn.BatchNorm2d / nn.BatchNorm1d (+ F.relu / nn.SELU) of model/resnet.py:47-191, model/wav2vec2_aasist.py:62-155,377-604. */
int scl_bn_nslabs(int N);
int scl_bn_fwd(const float* x, int N, int C, const float* gamma, const float* beta, float* running_mean, float* running_var,
               long long* num_batches_tracked, int training, float momentum, float eps, int act, float* part, float* mean,
               float* rstd, float* y, void* y2, int y2_bf16, int m_W, int m_HW, int64_t m_bs, int64_t m_rs, int64_t m_cs,
               int64_t m_base, void* stream);
/* backward of the above: dz = dy * act'(y); dgamma = sum dz * xhat, dbeta = sum dz (either may be NULL); dx = gamma * rstd *
 * (dz - [training] (sum dz + xhat * sum dz*xhat) / N).  part: as above; sums: f32 [2*C] scratch.  accumulate != 0: dgamma / dbeta are ADDED to
 * (autograd's accumulation into an attached .grad buffer, done by the finishing kernel). */
int scl_bn_bwd(const float* dy, const float* y, const float* x, const float* mean, const float* rstd, const float* gamma, int N, int C,
               int act, int training, float* part, float* sums, float* dgamma, float* dbeta, float* dx, int accumulate, void* stream);
/* src [rows, C] contiguous f32 -> mapped (padded / dilated) destination, f32 or bf16 (row mapping as scl_bn_fwd's y2) */
int scl_pad_nhwc_f32(const float* src, int64_t rows, int C, void* dst, int dst_bf16, int m_W, int m_HW, int64_t m_bs, int64_t m_rs,
                     int64_t m_cs, int64_t m_base, void* stream);
/* Both re-laid-out copies of a Conv2d weight [Co][Ci][kh][kw] (torch layout, model/wav2vec2_resnet_nll.py's nn.Conv2d parameters) for the
 * implicit-GEMM convolution: fwd [Co][kh][kw][Cp] (zero channels Ci..Cp) and bwd [Ci][kh][kw][Cop] with flipped taps (zero channels Co..Cop). */
int scl_conv_pack_weights(const float* w, float* fwd, float* bwd, int Co, int Ci, int kh, int kw, int Cp, int Cop, void* stream);
/* grad [Co][Ci][kh][kw] (+)= sum over nslab slabs [Co][kh][kw][Cp] in slab order (the conv weight gradient's split-K / per-utterance partials,
 * what torch's conv backward + AccumulateGrad produce); accumulate = 0 overwrites.  Cp % 4 == 0, slabs 16-byte aligned. */
int scl_conv_wgrad_finish(const float* slabs, float* grad, int nslab, int Co, int Ci, int kh, int kw, int Cp, int accumulate, void* stream);
/* F.max_pool2d(x, (3, 3)) of a single-channel map given by strides (elements): y [B, H/3, W/3], idx = flat argmax inside x[b]
 * (model/wav2vec2_aasist.py:517); the backward scatters dy into a zeroed dx. */
int scl_maxpool3_fwd(const float* x, int64_t xs_h, int64_t xs_w, int64_t xs_b, int H, int W, int B, float* y, int* idx, void* stream);
int scl_maxpool3_bwd(const float* dy, const int* idx, int H, int W, int B, float* dx, int64_t xs_h, int64_t xs_w, int64_t xs_b, void* stream);
/* y[b][c] = mean_r x[b][r][c] (F.adaptive_avg_pool2d(x, 1) on a channels-last map, model/resnet.py:186) and its backward */
int scl_avgpool_fwd(const float* x, int B, int R, int C, float* y, void* stream);
int scl_avgpool_bwd(const float* dy, int B, int R, int C, float* dx, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* LayerNorm (+GELU), column reductions                                                        */
/* ------------------------------------------------------------------------------------------ */
/* y = act(LN(x) * gamma + beta) per row of [M, C]; x f32 or bf16; y to bf16 and/or f32; saves mean / rstd.
 * act: 0 none, 1 gelu; | 0x100: y_bf16 receives the row as [hi | hi | lo] with pitch 3 C (hi = bf16(y), lo = bf16(y - hi): the left
 * operand of the scoring path's triple-plane GEMMs, see scl_split3_f32_bf16).  Replaces fairseq Fp32LayerNorm / nn.LayerNorm
 * (+ nn.GELU in the conv stack) inside Wav2Vec2Model.forward (model/xlsr.py:41). */
int scl_layernorm_fwd(const void* x, int x_f32, const float* gamma, const float* beta, void* y_bf16, float* y_f32,
                      float* mean, float* rstd, int M, int C, int64_t ldx, int64_t ldy, float eps, int act, void* stream);
/* number of row-slab partials scl_layernorm_bwd writes for M rows */
int scl_layernorm_bwd_nparts(int M);
/* dx = LN'(dy [* gelu'(.)]) (+ dres); per-slab partial sums into part[nparts][2*C] = (dgamma | dbeta) per slab
 * (one scl_colreduce_f32 over 2*C columns finishes both).  Autograd backward of the above (main.py:79). */
int scl_layernorm_bwd(const void* dy, int dy_f32, const void* x, int x_f32, const float* mean, const float* rstd,
                      const float* gamma, const float* beta, const float* dres, float* dx_f32, void* dx_bf16,
                      float* part, int M, int C, int64_t ldx, int64_t lddy, int64_t lddx, int act, int sum_dres, int out_rpb,
                      int64_t out_rbstride, int64_t out_off, uint32_t din_seed, float din_p, uint32_t dout_seed, float dout_p, void* stream);
/* din / dout (p = 0: off): the encoder's element dropout on the output of out_proj / fc2 (fairseq dropout1 / dropout3, p = cfg.dropout,
 * model/xlsr.py:33-41 runs the encoder in train mode).  din: the column sums of dres (sum_dres == 1) are taken of dres x keep-mask(din_seed,
 * row * C + col) — the bias gradient of the linear whose dropped output fed the residual.  dout: the bf16 output is multiplied by
 * keep-mask(dout_seed, row * C + col) — it is the dY operand of the next linear's gradient GEMMs; the f32 output stays unmasked. */
/* out_rpb > 0: bf16 output row r goes to element (r / out_rpb) * out_rbstride + (r % out_rpb) * lddx + out_off (per-utterance
 * zero padding kept by the caller) */
/* sum_dres == 1: `part` rows are [dgamma | dbeta | colsum(dres)] (3*C floats per slab instead of 2*C; sum_dres == 2: third row =
 * colsum of the OUTPUT dx, the bias gradient of the Conv1d in front of a conv-stack LayerNorm) — the residual gradient
 * entering a pre-LN block's LayerNorm backward is the output gradient of the preceding fc2 / out_proj, so its column sum is
 * that layer's bias gradient (fairseq TransformerSentenceEncoderLayer, reached from model/xlsr.py:41) */
int scl_colreduce_f32(const float* part, float* out, int nparts, int C, int64_t pstride, int accumulate, void* stream);
/* the same reduction spread over SCL_COLREDUCE_SEGMENTS x more blocks, finished in-launch by the last block of each 32-column group;
 * scratch: f32 [SCL_COLREDUCE_SEGMENTS][C]; counters: as for scl_colsum_reduce (C <= 32 * SCL_COLSUM_MAX_GROUPS) */
#define SCL_COLREDUCE_SEGMENTS 8
#define SCL_COLSUM_MAX_GROUPS 128
/* Up to SCL_REDUCE_MAX_JOBS column reductions in one launch (out[c] = sum_p part[p * pstride + c]; columns >= split go to out2 when
 * it is given): the bias / LayerNorm-parameter gradient sums that close an encoder layer's backward (main.py:79). */
#define SCL_REDUCE_MAX_JOBS 8
typedef struct SclReduceJob {
    const float* part;
    float*       out;
    float*       out2;
    int64_t      pstride;
    int32_t      nparts, C, split, _pad;
} SclReduceJob;
int scl_colreduce_multi(const SclReduceJob* jobs, int njobs, void* stream);
/* out2 (optional): columns [split, C) are written to out2[0 .. C-split) instead of out[split ..) */
int scl_colreduce_seg_f32(const float* part, float* out, int nparts, int C, int64_t pstride, int accumulate, float* scratch, int* counters,
                          float* out2, int split, void* stream);
/* bias gradients: part[p][n] = sum over row slab p of x[m][n]; nparts = scl_colsum_nparts(M) */
int scl_colsum_nparts(int M);
/* partial rows scl_colsum_reduce needs in `part` for an [M, N] input (>= scl_colsum_nparts(M) only for N <= 128, where it cuts finer) */
int scl_colsum_reduce_nparts(int M, int N);
int scl_colsum(const void* x, int x_f32, float* part, int M, int N, int64_t ld, void* stream);
/* the same plus the final sum in one launch: out[n] = sum_m x[m][n] (the `.sum(0)` autograd runs for every nn.Linear / Conv1d bias,
 * e.g. fairseq fc1/fc2/q,k,v,out_proj reached from model/xlsr.py:41).  `counters`: SCL_COLSUM_MAX_GROUPS int32, zero before the
 * first use, left zero by every launch; launches sharing one counter array must be ordered on one stream. */
int scl_colsum_reduce(const void* x, int x_f32, float* part, int* counters, float* out, int M, int N, int64_t ld, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* element-wise glue                                                                           */
/* ------------------------------------------------------------------------------------------ */
int scl_cast_f32_bf16(const float* src, void* dst, int64_t n, void* stream);
/* f32 [rows][K] (pitch ldx) -> bf16 [rows][3 K]: per row [hi | hi | lo] (order 0, left operand) or [hi | lo | hi] (order 1, right operand),
 * hi = bf16(x), lo = bf16(x - hi): one bf16 GEMM over 3 K then yields hi.hi + hi.lo + lo.hi = the f32 product to ~2^-17 relative.  The
 * scoring path's linears (encoder.forward_f32: main.py --eval / --predict / --emb, reference main.py:161-214) run that way on the wide
 * bf16 kernel. */
int scl_split3_f32_bf16(const float* x, int64_t rows, int K, int64_t ldx, void* out, int order, void* stream);
int scl_add_f32(const float* a, const float* b, float* out, void* out_bf16, int64_t n, void* stream);
/* dst[b][r][:] = src[b][r - pad_before][:] (* act'(pre)), zero outside [0,T): zero-padded operand of the
 * grouped positional conv (fairseq encoder.pos_conv, padding = k/2) and of its dgrad. */
int scl_pad_rows_bf16(const void* src, int src_f32, void* dst, const void* pre, int ract, int B, int T, int C,
                      int rows_out, int pad_before, void* stream);
/* conv-stack dgrad tail: dz[b][r][c] = sum_j dcol[b][(r-j)/s][j*C+c] (Conv1d backward-data, layers 1..6) */
int scl_col2im_bf16(const void* dcol, void* dz, int B, int Tin, int Tout, int C, int k, int s, void* stream);
/* Conv1d weight [co][ci][j] f32 <-> GEMM operand [co][j*Ci+ci] (bf16 forward copy / f32 gradient back).  wd (optional): the
 * backward-data operand of the phase-split transposed convolution, [k tap blocks][co][ci] bf16, blocks ordered by phase
 * p = j mod stride and, inside a phase, by descending tap — dz[stride*u + p] = [dy[u-q_max] .. dy[u]] x wd(phase p) is then one
 * GEMM per phase with overlapping A rows, and no [M, k*C] column buffer / col2im pass exists (Conv1d backward, layers 1..6) */
int scl_conv_weight_pack(const float* w, void* wk, void* wd, int Co, int Ci, int k, int stride, void* stream);
int scl_conv_weight_unpack_grad(const float* dwk, float* dw, int Co, int Ci, int k, void* stream);
/* torch.nn.utils.weight_norm(dim=2) of encoder.pos_conv.0 + GEMM layouts (forward and flipped dgrad).  K must divide 256.
 * sdot_ws: K + E*K floats (per-tap sums, then per-output-channel partials). */
int scl_posconv_weight_pack(const float* v, const float* g, float* norm, void* wf, void* wd, int E, int Cg, int K, void* stream);
int scl_posconv_weight_bwd(const float* dwf, const float* v, const float* g, const float* norm, float* sdot_ws,
                           float* dv, float* dg, int E, int Cg, int K, void* stream);
/* tail of the linear head: mean over frames, m_utt_level, log_softmax (model/wav2vec2_linear_nll.py:88-93,134) */
/* y = x * keep-mask(seed, i) / (1 - p), to f32 and / or bf16 (in place allowed): fairseq TransformerEncoder.extract_features'
 * F.dropout(x, p = cfg.dropout) after the positional-conv residual add, its backward, and the backward of dropout_input */
int scl_dropout_f32(const float* x, float* y_f32, void* y_bf16, int64_t n, uint32_t seed, float p, void* stream);
/* y[r][j] = x[r][j] * keep-mask(seed, r * T + j) / (1 - p) for j < T, 0 for T <= j < ld; x, y: [R, ld] bf16 (is_f32 == 0) or f32, in place
 * allowed.  Attention dropout of fairseq MultiheadAttention (reached from model/xlsr.py:41) on the un-fused path: same mask index as
 * scl_attn_fwd / scl_attn_bwd (row r = (b, h, q)). */
int scl_dropout_rows(const void* x, void* y, int64_t R, int T, int ld, int is_f32, uint32_t seed, float p, void* stream);
int scl_meanpool_fwd(const void* h, float* emb, int B, int T, int C, void* stream);
int scl_meanpool_bwd(const float* demb, const void* pre, void* dpre, int B, int T, int C, int ract, float drop_p,
                     uint32_t seed, void* stream);
/* the same two with f32 frame-level activations: the training path keeps the 128-wide frame-level head (BackEnd.m_frame_level,
 * model/wav2vec2_linear_nll.py:60-93) in f32 — its backward spreads ONE row per utterance over all T frames, and a bf16
 * rounding of that row would be systematic over the T-row sums of the weight gradients instead of averaging out */
int scl_meanpool_fwd_f32(const float* h, float* emb, int B, int T, int C, void* stream);
int scl_meanpool_bwd_f32(const float* demb, const float* pre, float* dpre, int B, int T, int C, int ract, float drop_p,
                         uint32_t seed, void* stream);
int scl_utt_head_fwd(const float* emb, const float* W, const float* bias, float* logp, int B, int C, int NC, void* stream);
int scl_utt_head_bwd(const float* dlogp, const float* logp, const float* emb, const float* W, const float* demb_in,
                     float* demb, float* dW, float* db, float* ws, int B, int C, int NC, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* attention softmax (between the QK^T and PV contractions)                                    */
/* ------------------------------------------------------------------------------------------ */
int scl_softmax_fwd(const float* S, void* P, int64_t R, int T, int ldS, int Tp, void* stream);
int scl_softmax_bwd(const void* P, const float* dP, void* dS, int64_t R, int T, int lddP, int Tp, void* stream);
/* the forward with f32 probabilities (columns T..Tp-1 zero): the fp32 scoring path (main.py:161-214 runs fp32 end to end) */
int scl_softmax_fwd_f32(const float* S, float* P, int64_t R, int T, int ldS, int Tp, void* stream);
/* Fused attention for head dim 64 (scores stay on chip).  qkv / dqkv: bf16 [B, T, 3, H, 64]; ctx / dctx: bf16 [B, T, H*64];
 * lse: f32 [B, H, T] row log-sum-exp of the scaled scores.  fwd: T <= 256; bwd: T <= 224 (LDS budget).
 * Replaces F.multi_head_attention_forward inside fairseq's TransformerSentenceEncoderLayer (model/xlsr.py:41) and its backward.
 * drop_p / drop_seed: attention dropout on the probabilities (fairseq MultiheadAttention.dropout_module, p = cfg.attention_dropout):
 * keep-mask hash(seed, ((b*H + h)*T + query)*T + key), recomputed by the backward from the same seed; 0 = off. */
int scl_attn_fwd(const void* qkv, void* ctx, float* lse, int B, int T, int H, int D, float scale, float drop_p, uint32_t drop_seed,
                 void* stream);
/* bias_part (optional, f32 [B, 3*H*64]): per-utterance column sums of dqkv, from the f32 accumulators — summed over B (scl_colreduce_f32)
 * they are the q/k/v bias gradients, which otherwise cost a pass over dqkv. */
/* fp8 variant of the forward (BASELINE.json configs[4]): K, V, Q and the probabilities as OCP e4m3 operands of v_mfma_f32_16x16x32_fp8_fp8,
 * fp32 accumulation and soft-max statistics; same arguments and outputs as scl_attn_fwd without dropout.  Opt-in (SCL_ATTN_FP8=1: the
 * encoder's no-grad bf16 forward); 6e-2 relative L2 of ctx against the fp32 arithmetic of the reference. */
int scl_attn_fwd_fp8(const void* qkv, void* ctx, float* lse, int B, int T, int H, int D, float scale, void* stream);
int scl_attn_bwd(const void* qkv, const void* ctx, const void* dctx, const float* lse, void* dqkv, float* bias_part, int B, int T,
                 int H, int D, float scale, float drop_p, uint32_t drop_seed, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* feature-extractor layer 0 (Conv1d(1,C,10,5) + LayerNorm + GELU), fused fwd / bwd            */
/* ------------------------------------------------------------------------------------------ */
/* stats (optional, may be NULL): f32 [B*T0][2] = per-frame (mean, rstd) of the LayerNorm, written by the forward and read
 * back by the backward instead of being recomputed */
int scl_conv0_fwd(const float* x, const float* w, const float* bias, const float* gamma, const float* beta, void* z, float* stats,
                  int B, int L, int C, int k, int stride, float eps, void* stream);
/* the same with an fp32 output map (scoring path: activations stay fp32 end to end) */
int scl_conv0_fwd_f32(const float* x, const float* w, const float* bias, const float* gamma, const float* beta, float* z,
                      int B, int L, int C, int k, int stride, float eps, void* stream);
int scl_conv0_bwd_nparts(int B, int L, int k, int stride);
int scl_conv0_bwd(const float* x, const float* w, const float* bias, const float* gamma, const float* beta, const void* dz,
                  const float* stats, float* part_ws, float* dW, float* db, float* dgamma, float* dbeta, int B, int L, int C, int k,
                  int stride, float eps, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* AASIST graph attention: pairwise attention score, fused (model/wav2vec2_aasist.py:107-135,   */
/* 259-291: att_proj(x_i * x_j) -> tanh -> dot with att_weight{,11,22,12})                      */
/* ------------------------------------------------------------------------------------------ */
/* s[b][i][j] = sum_o tanh(sum_d W[o][d] x[b][i][d] x[b][j][d] + bias[o]) * a[t(i,j)][o];  t = 0 (i, j < n1), 1 (i, j >= n1),
 * 2 (mixed); the homogeneous layer passes n1 = N and only a[0] is used.  All f32.  x [B,N,D], W [Do,D], a [3,Do], s [B,N,N];
 * D in {32, 64}, Do <= 64, N <= 128.  Temperature and softmax stay with the caller. */
int scl_gat_score_nblocks(int N);
int scl_gat_score_fwd(const float* x, const float* W, const float* bias, const float* a, float* s, int B, int N, int D, int Do, int n1,
                      void* stream);
/* backward: dP f32 [B, N*N, D] scratch; part f32 [B * scl_gat_score_nblocks(N)][Do*D + 4*Do] per-block partial sums
 * (dW | dbias | da[0] | da[1] | da[2]) for the caller to sum over the first dimension; dx f32 [B, N, D] */
int scl_gat_score_bwd(const float* x, const float* W, const float* bias, const float* a, const float* ds, float* dP, float* part, float* dx,
                      int B, int N, int D, int Do, int n1, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* AASIST RawNet2-style encoder: the six Residual_blocks (model/wav2vec2_aasist.py:377-433,     */
/* stacked at :470-476) on zero-bordered flat maps, exact fp32 on the f32 matrix cores           */
/* ------------------------------------------------------------------------------------------ */
/* A map of the stack: utterance b owns H + 2 rows of W + 2 positions of C channels (f32, channels last); flat position
 * g = (b * (H + 2) + r) * (W + 2) + c.  Rows r_lo..r_hi and columns 1..W hold values, everything else is zero (the convolutions'
 * padding).  Buffers carry >= W + 260 positions of slack on either side of [0, G) (finite values; never part of a valid result). */
typedef struct SclRsGeom { int32_t B, H, W, r_lo, r_hi, _pad; } SclRsGeom;
/* out[g][n] = mask(g) * (bias[n] + addend[g][n] + sum_t sum_c in[g + shift[t]][c] * Wt[t][c][n])   — a (kh, kw) tap of a stride-1
 * convolution over a bordered map is ONE flat shift; the data gradient is the same call with negated shifts and transposed weights.
 * wpk: scl_rs_pack_weights image.  (cin, cout, ntaps) in {(16|32,32,6), (32|64,64,6), (16,32,3), (32,64,3), (32,16,3|6), (64,32,3|6), (64,64,1)}.
 * stat_mode 1: per-channel sum / sum of squares of the stored values; the last block turns them into BatchNorm statistics
 *   stats_out[4][cout] = mean, rstd, gamma * rstd, beta (biased variance, eps) and, when run_mean is given, updates
 *   running_mean / running_var (momentum, unbiased) and num_batches_tracked exactly as nn.BatchNorm2d in training does.
 * stat_mode 2 (BatchNorm + SELU backward, first half): out = mask * conv * selu'(act_a) = dz; sums of dz and dz * xhat with
 *   xhat = (y1 - mean) * rstd from bnstats; the last block adds them to dbeta / dgamma and stores stats_out[2][cout] = their means
 *   (zeros when training == 0) for scl_rs_bn_bwd_apply.
 * acc: SCL_RS_NSLOT * 2 * cout zeroed doubles, ticket: one zeroed uint32 (both are left zeroed).  nvalid = number of unmasked positions.
 * stat_mode 1 exists for the forward shapes (cout >= cin, six taps) and the 1-tap 64 -> 64 form, stat_mode 2 for cin == cout with six taps or one
 * (act_a == NULL: the sums of the plain convolution result — the gradient arriving at a BatchNorm output). */
#define SCL_RS_NSLOT 16
typedef struct SclRsConv {
    const float* in; const float* wpk; const float* bias; const float* addend; float* out;
    const float* act_a; const float* y1; const float* bnstats;
    double* acc; uint32_t* ticket;
    const float* gamma; const float* beta; float* run_mean; float* run_var; int64_t* nbt;
    float* stats_out; float* dgamma; float* dbeta;
    double nvalid;
    SclRsGeom geom;
    int32_t shift[6];
    int32_t cin, cout, ntaps, stat_mode, training, epi_act;      /* epi_act: store (and take the statistics of) selu(conv + bias + addend) */
    float eps, momentum;
} SclRsConv;
int scl_rs_conv(const SclRsConv* c, void* stream);
/* torch [Co, Ci, KH, KW] (ntaps = KH * KW, tap t = kh * KW + kw) -> register image for scl_rs_conv (out: COUTp * ntaps * CINp floats);
 * CINp / COUTp: the kernel's channel counts (multiples of 16, zero-filled); transposed != 0: the data-gradient image (the convolution's
 * Co becomes the contraction).  One launch packs up to SCL_RS_MAX_PACK_JOBS images. */
#define SCL_RS_MAX_PACK_JOBS 32
typedef struct SclRsPackJob { const float* w; float* out; int32_t Co, Ci, ntaps, CINp, COUTp, transposed, ld, _pad; } SclRsPackJob;      /* ld: row pitch (0: Ci) */
int scl_rs_pack_weights(const SclRsPackJob* jobs, int njobs, void* stream);
/* part[scl_rs_wgrad_nslabs(cin, cout)][ntaps * cin * cout] f32 partial slabs of dW[t][c][n] = sum_g in[g + shift[t]][c] * dout[g][n]
 * (dout zero off the valid positions); dbias[n] += sum_g dout[g][n] when dbias != NULL (bacc: SCL_RS_NSLOT * cout zeroed doubles, ticket as above) */
int scl_rs_wgrad_nslabs(int cin, int cout);
int scl_rs_wgrad(const float* in, const float* dout, int cin, int cout, int ntaps, const int* shift, const SclRsGeom* geom, float* part,
                 double* bacc, uint32_t* ticket, float* dbias, void* stream);
/* dw (torch layout [Co, Ci, ntaps]) += the slabs, summed in index order */
int scl_rs_wgrad_reduce(const float* part, int nslab, int ntaps, int CINp, int COUTp, int Co, int Ci, int ld, float* dw, void* stream);      /* ld: row pitch of dw (0: Ci) */
/* a = mask * selu((y - stats[0]) * stats[2] + stats[3]) (BatchNorm + SELU, model/wav2vec2_aasist.py:423-424) */
int scl_rs_bn_act(const float* y, const float* stats, float* a, int C, int act, const SclRsGeom* geom, void* stream);      /* act 0: the affine map alone */
/* in place: dz := mask * stats[2] * (dz - bstats[0] - (y - stats[0]) * stats[1] * bstats[1]) — the input gradient of that BatchNorm */
int scl_rs_bn_bwd_apply(float* dz, const float* y, const float* stats, const float* bstats, int C, int selu_in, const SclRsGeom* geom, void* stream);      /* selu_in: y is a SELU output, chain selu'(y) */
/* attention pooling of the AASIST encoder output (model/wav2vec2_aasist.py:527-541) over bordered maps x, l [G, C]:
 * eS[b][h][c] = sum_w x softmax_w(l) + pos[h][c] (pos may be NULL), eT[b][w][c] = sum_h x softmax_h(l); backward writes the interiors of
 * dx, dl (their borders must already be zero) */
int scl_rs_attn_pool_fwd(const float* x, const float* l, const float* pos, float* eS, float* eT, int C, const SclRsGeom* geom, void* stream);
int scl_rs_attn_pool_bwd(const float* x, const float* l, const float* deS, const float* deT, float* dx, float* dl, int C, const SclRsGeom* geom, void* stream);
/* eval mode: stats[4][C] from the running statistics */
int scl_rs_bn_eval_stats(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps, int C,
                         float* stats, void* stream);
/* to_dense == 0: dense [B * H' * W, Cs] -> bordered [G, Cd] (rows r_lo..r_hi, H' = r_hi - r_lo + 1; other channels and positions zero);
 * to_dense != 0: bordered [G, Cs] -> dense [B * H' * W, Cd] (the first Cd channels) */
int scl_rs_copy(const float* src, float* dst, int Cs, int Cd, int to_dense, const SclRsGeom* geom, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* AASIST graph module: GraphAttentionLayer / HtrgGraphAttentionLayer / GraphPool / read-out    */
/* (model/wav2vec2_aasist.py:62-155, 158-332, 336-374, 545-604), one workgroup per utterance     */
/* ------------------------------------------------------------------------------------------ */
/* All tensors f32 (indices int32), [B][...] contiguous per utterance.  Parameter gradients: every workgroup writes its contribution
 * into row b of a slab [B][slab_bs] at the o_* offsets; scl_graph_reduce sums the rows in index order into the gradients.
 * BatchNorm1d over (utterances x nodes): acc = SCL_RS_NSLOT * 2 * C zeroed doubles + a zeroed uint32 ticket (left zeroed). */
typedef struct SclGraphBn {
    double* acc; uint32_t* ticket;
    const float* gamma; const float* beta; float* run_mean; float* run_var; int64_t* nbt;
    float* stats;          /* [4][C] mean, rstd, gamma * rstd, beta (written in training, given in eval) */
    float* bstats;         /* [2][C] backward: means of dz and dz x xhat (zeros in eval) */
    float* dgamma; float* dbeta;      /* += */
    double nvalid; float eps, momentum; int32_t training, _pad;
} SclGraphBn;
/* One attention layer after its pairwise scores (scl_gat_score_fwd wrote S): A = softmax_j(S / temp) (stored over S), g = A xd,
 * y = g Wa^T + ba + xd Wb^T + bb, BatchNorm statistics of y; heterogeneous layers (has_master) also update the master node:
 * am = softmax_n((tanh((xd * m) WM^T + bM) aM) / temp), gm = am^T xd, mout = gm WaM^T + baM + m WbM^T + bbM.
 * Backward (dz = d selu-output * selu' given, bn.bstats finished): dy, the parameter-gradient slab entries, dS (for scl_gat_score_bwd),
 * dxd (without the score path's part), d_min; d_mout / d_mout2 are the two gradients arriving at mout (either may be NULL). */
typedef struct SclGraphLayer {
    const float* xd; float* S; float* g; float* y;
    const float *Wa, *ba, *Wb, *bb;
    const float* min; int64_t min_bs;
    const float *WM, *bM, *aM, *WaM, *baM, *WbM, *bbM;
    float *am, *gm, *tM, *mout;
    SclGraphBn bn;
    int32_t N, D, Do, has_master; float inv_temp; int32_t _pad;
    const float* dz; const float* d_mout; const float* d_mout2; float* dS; float* dxd; float* d_min;
    float* slab; int64_t slab_bs;
    int32_t o_Wa, o_ba, o_Wb, o_bb, o_WM, o_bM, o_aM, o_WaM, o_baM, o_WbM, o_bbM, _pad2;
} SclGraphLayer;
/* launches 1 or 2 layers (grid.y) of equal batch */
int scl_graph_post_fwd(const SclGraphLayer* layers, int nlayers, int B, void* stream);
int scl_graph_post_bwd(const SclGraphLayer* layers, int nlayers, int B, void* stream);
/* What sits between two attention layers, per unit of nodes: h = selu(bn(y rows)) of the layer below, GraphPool (score = sigmoid(proj(drop h)),
 * top-K by descending score, rows gated by their score), proj_type, then the input dropout of the next layer over the concatenated units. */
typedef struct SclGraphPoolUnit {
    const float* ysrc; int32_t src_n, row0, n_in, K;
    const float* stats;
    const float* pw; const float* pb; uint32_t pool_seed; float pool_p;
    float* h; float* sc; int32_t* idx; float* pooled;
    const float* Wt; const float* bt;
    int32_t row_out, _pad;
    const float* d_res; float* dz;
    int32_t o_pw, o_pb, o_Wt, o_bt;
    SclGraphBn bn;
} SclGraphPoolUnit;
typedef struct SclGraphPre {
    SclGraphPoolUnit u[2];
    float* xd; const float* dxd_a; const float* dxd_b;
    uint32_t in_seed; float in_p;
    int32_t Dp, N, store_common, same_bn;
    float* slab; int64_t slab_bs;
} SclGraphPre;
/* two instances per launch; shared_pool != 0: both read the same pooled nodes (HtrgGAT_layer_ST11 / ST21 share pool_S / pool_T), the
 * backward then sums their pooled-node gradients inside one workgroup */
int scl_graph_pre_fwd(const SclGraphPre* inst, int shared_pool, int B, void* stream);
int scl_graph_pre_bwd(const SclGraphPre* inst, int shared_pool, int B, void* stream);
/* y = x * keep-mask(seed, element) / (1 - p) for the two first layers; backward de = (da + db) * mask */
int scl_graph_drop(const float* x0, float* y0, int64_t n0, uint32_t seed0, const float* x1, float* y1, int64_t n1, uint32_t seed1, float p, void* stream);
int scl_graph_drop_bwd(const float* da0, const float* db0, float* de0, int64_t n0, uint32_t seed0, const float* da1, const float* db1, float* de1,
                       int64_t n1, uint32_t seed1, float p, void* stream);
/* read-out of the two branches: aug = selu(bn(y2)); T = Tp + aug[:KT], S = Sp + aug[KT:], m = m1 + m2; drop_way; branch max;
 * hidden = dropout([max|T|, mean T, max|S|, mean S, m]); logits = hidden Wout^T + bout   (model/wav2vec2_aasist.py:572-604) */
typedef struct SclGraphFinalBranch {
    const float* y2; const float* stats2; const float* Tp; const float* Sp; const float* m1; const float* m2;
    uint32_t way_seed[3]; int32_t _pad;
    float* dz2; float* dTp; float* dSp; float* dm1; float* dm2;
    SclGraphBn bn;
} SclGraphFinalBranch;
typedef struct SclGraphFinal {
    SclGraphFinalBranch br[2];
    const float* Wout; const float* bout; float* logits; float* hidden;
    const float* d_logits; const float* d_hidden;
    float* slab; int64_t slab_bs; int32_t o_Wout, o_bout;
    int32_t KT, KS, D, NC; float way_p, drop_p; uint32_t drop_seed; int32_t _pad;
} SclGraphFinal;
int scl_graph_final_fwd(const SclGraphFinal* f, int B, void* stream);
int scl_graph_final_bwd(const SclGraphFinal* f, int B, void* stream);
/* dst[i] += sum_{k < nparts} src[k * stride + i], k ascending; up to SCL_GRAPH_MAX_REDUCE_JOBS jobs per launch */
#define SCL_GRAPH_MAX_REDUCE_JOBS 96
typedef struct SclGraphReduceJob { const float* src; float* dst; int64_t stride; int32_t n, nparts; } SclGraphReduceJob;
int scl_graph_reduce(const SclGraphReduceJob* jobs, int njobs, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* losses: supervised contrastive (model/loss_metrics.py:85-209) and NLL (linear_nll.py:167)   */
/* ------------------------------------------------------------------------------------------ */
/* Any batch size 1 <= bz <= SCL_SUPCON_MAX_BZ (the reference has no limit: under nn.DataParallel Model.loss sees the gathered batch of all
 * GPUs, main.py:62-66).  Up to 128 utterances the loss and dL/dS come from one workgroup with S in LDS; larger batches take a wave per row.
 * Buffers: ws = scl_supcon_ws_floats(bz, K) floats (= scl_supcon_nchunks(K) * bz * bz partial sums + bz row losses); G = 2 * bz * bz
 * floats — [0, bz*bz) receives dL/dS from the forward and is read by the backward, [bz*bz, 2*bz*bz) is scratch (S of a batch of more than
 * 128 in the forward when S_out is null; the symmetrised, scaled coefficient matrix of the backward's GEMM form). */
#define SCL_SUPCON_MAX_BZ 1024
int scl_supcon_nchunks(int64_t K);
long long scl_supcon_ws_floats(int bz, int64_t K);
int scl_supcon_fwd(const float* F, const int64_t* labels, int bz, int64_t K, int64_t ldF, int Tprime, float temperature,
                   float* ws, float* G, float* loss_out, float* S_out, void* stream);
int scl_supcon_bwd(const float* F, const float* G, const float* upstream, float coef, int bz, int64_t K, int64_t ldF,
                   int Tprime, float temperature, float* dF, void* dF_bf16, int accumulate, void* stream);
int scl_nll_fwd(const float* logp, const int64_t* labels, int bz, int NC, float* loss_out, float* dlogp_coef, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* optimizer: torch.optim.AdamW semantics over a flat buffer (main.py:339,80)                  */
/* ------------------------------------------------------------------------------------------ */
int scl_adamw_flat(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int step, float grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* waveform augmentation (datautils/RawBoost.py, datautils/audio_augmentor/*, wav_augmentation.py) */
/* ------------------------------------------------------------------------------------------ */
/* y[c][n] = sum_{f<nf} sum_k taps_f[k] * x[c][n + h_f - k]^(use_pow ? f+1 : 1), n in [0,Lout), x zero outside [0,Lin).
 * filterFIR: h = (len+1)/2, Lout = Lin; np.convolve full (reverb): h = 0, Lout = Lin+len-1.
 * part (optional): per clip and 2048-sample block (sum, min, max, sumsq) of y. */
int scl_fir_nblocks(int Lout);
int scl_fir_multi_f32(const float* x, int64_t ldx, int Lin, const float* taps, const int* tap_off, const int* tap_len,
                      const int* tap_h, int nclip, int nf, int use_pow, float* y, int64_t ldy, int Lout, float* part, void* stream);
int scl_clip_stats_f32(const float* x, int64_t ldx, int L, int nclip, float* part, void* stream);
int scl_isd_scatter_f32(float* y, int64_t ldy, const int* pos, const float* fr, const int* clip_off, int nclip, int max_per_clip,
                        float g_sd, void* stream);
#define SCL_AFF_CENTER_PEAK_COND 0  /* y - mean(y), then / max|.| iff > 1           (LnL tail, RawBoost.py:67-68)  */
#define SCL_AFF_PEAK_COND        1  /* normWav(x, 0)                                (ISD tail, RawBoost.py:83)     */
#define SCL_AFF_PEAK_ALWAYS      2  /* normWav(x, 1)                                                               */
#define SCL_AFF_SSI_MIX          3  /* z + x * ||z|| / ||x|| / 10^(snr/20)          (RawBoost.py:93-96)            */
#define SCL_AFF_PEAK_QUANT_I16   4  /* int16(x / max|x| * 32768) by C cast, as float (reverb.py:41-44, utils.py:26) */
int scl_clip_affine_f32(int mode, const float* x, int64_t ldx, const float* z, int64_t ldz, const float* partx, const float* partz,
                        const float* snr_db, float* out, int64_t ldo, int L, int nclip, void* stream);
int scl_f32_to_i16_wrap(const float* x, void* out_i16, int64_t n, void* stream);
int scl_i16_sumsq(const void* x_i16, int64_t n, uint64_t* part64, int nparts, void* stream);
int scl_i16_gain_overlay(const void* speech_i16, int64_t n, const void* noise_i16, int64_t nn, double factor, float* out_f32,
                         void* out_i16, void* stream);
int scl_multiview_crop_f32(const float* src, const int64_t* off, const int* len, int V, int firstlen, int start, int out_len,
                           int repeat_pad, float* out, int64_t ldo, void* stream);

/* ---- file reader: FLAC (csrc/flac.hip, host code) ----------------------------------------------------------------------------------
 * Stands where the reference calls librosa.load / AudioSegment.from_file on ASVspoof's .flac utterances
 * (datautils/asvspoof_2019_augall_3.py:97-100, audio_augmentor/background_noise.py:22-28): `data` is the whole file in host memory.
 * scl_flac_info reads STREAMINFO; scl_flac_decode_i32 writes interleaved samples [total][channels] as sign-extended int32 (divide
 * by 2^(bits-1) for librosa's float range), checks every frame's CRC-8 / CRC-16 and, with check_md5 != 0, the MD5 signature of the
 * decoded audio.  total_samples == 0 in STREAMINFO (unknown length): decode with a capacity of the caller's choosing. */
int scl_flac_info(const void* data, int64_t nbytes, int* sample_rate, int* channels, int* bits_per_sample, int64_t* total_samples);
int scl_flac_decode_i32(const void* data, int64_t nbytes, int32_t* out, int64_t capacity_samples, int64_t* decoded_samples, int check_md5);
/* The same stream as what librosa.load(path, sr=None, mono=True) returns: float32 [total], every sample / 2^(bits-1), channels averaged
 * (sequential float32 sum, then the division, as numpy's mean over the channel axis) — one pass, no int32 image (round 6: the
 * pack builder's and the scoring loop's reader). */
int scl_flac_decode_mono_f32(const void* data, int64_t nbytes, float* out, int64_t capacity_samples, int64_t* decoded_samples, int check_md5);

/* ---- conf-5 augmenters (csrc/speedpitch.hip) ----------------------------------------------------------------------------------
 * speed: datautils/audio_augmentor/speed.py:29-33 -> pydub 0.25.1 AudioSegment.speedup(speed_factor).  scl_i16_append_xfade is
 * AudioSegment.append(chunk, crossfade) applied in place to the running 16-bit output `out` (n1 frames, capacity >= n1 + tail_n):
 *   j < R:  out[a0 + j] = sat( floor(clip(out[a0 + j] * g1(j))) + floor(clip(chunk[j % m2] * g2(j % m2))) )      (audioop.mul / add)
 *   t < tail_n:  out[n1 + t] = chunk[tail_off + t]
 * g(i) = from + step * (per_ms ? i / frames_per_ms : i) in fp64, evaluated as CPython does (multiply, then add).  Positive
 * crossfades mix the last R frames with the chunk's first R; pydub's negative crossfades (speed factor < 1) fade the whole output
 * after its first |c| ms under a looped fade-in of the chunk (m2 < R).  The caller walks the chunk list (scl_amd/augment.py).
 * pitch: datautils/audio_augmentor/pitch.py:31-38 -> librosa 0.10.0 effects.pitch_shift = stft (2048 / 512, periodic Hann, centred,
 * zero padding; D is [frames][1025] complex64) -> phase_vocoder(rate) -> istft(length) -> resample(ratio = rate) -> fix_length.
 * scl_resample_sinc_f32 stands where librosa calls soxr_hq: Kaiser-windowed sinc, 32 zero crossings, beta 14.77, cut-off
 * 0.95 x the lower Nyquist, n_out = ceil(n_in * ratio) chosen by the caller.  frames_ws: [nframes][2048] f32 scratch. */
int scl_i16_append_xfade(void* out_i16, int n1, const void* chunk_i16, int n2, int a0, int R, int per_ms1, double from1, double step1,
                         int m2, int per_ms2, double from2, double step2, int tail_off, int tail_n, int frames_per_ms, void* stream);
int scl_stft_nframes(int L);
int scl_stft_f32(const float* y, int L, void* D_c64, int nframes, void* stream);
int scl_phase_vocoder_c64(const void* D_c64, int nframes, double rate, void* out_c64, int nsteps, void* stream);
int scl_istft_f32(const void* D_c64, int nframes, float* frames_ws, float* y, int length, void* stream);
int scl_resample_sinc_f32(const float* x, int n_in, double ratio, float* out, int n_out, void* stream);

/* ---- Conformer block (csrc/conformer.hip; BASELINE.json configs[4]'s head: model/conformer.py:180-216, importable although the
 * BTSE plugin that would call it is not) --------------------------------------------------------------------------------------------
 * Everything fp32, channels-last.  The block's Linear / 1x1-Conv1d / attention contractions run on scl_gemm_bf16 with f32 operands
 * (SCL_GEMM_AB_F32), its LayerNorms on scl_layernorm_*, its BatchNorm1d on scl_bn_*; these are the remaining operators.
 * Swish: y = x sigmoid(x) (conformer.py:25-27).  GLU over the last dimension of [M, 2C] (conformer.py:29-36, dim = channels):
 * y[m][c] = x[m][c] sigmoid(x[m][C + c]).  scl_axpby_f32: out = sa a + sb b (b may be null) — Scale(0.5, .) and the halved biases. */
int scl_swish_fwd(const float* x, float* y, int64_t n, void* stream);
int scl_swish_bwd(const float* dy, const float* x, float* dx, int64_t n, void* stream);
int scl_glu_fwd(const float* x, float* y, int64_t M, int C, void* stream);
int scl_glu_bwd(const float* dy, const float* x, float* dx, int64_t M, int C, void* stream);
int scl_axpby_f32(const float* a, const float* b, float sa, float sb, float* out, int64_t n, void* stream);
/* DepthWiseConv1d (conformer.py:38-46): y[b][t][c] = bias[c] + sum_j w[c][j] x[b][t + j - pad_l][c] on [B, n, C] maps, zero outside the
 * utterance, output length n (pad_l + pad_r = k - 1), k <= 32 taps.  flip = 1 reads the taps back to front: the data gradient is
 * scl_dwconv1d_fwd(dy, w, NULL, dx, ..., pad_l' = k - 1 - pad_l, flip = 1).  scl_dwconv1d_wgrad: part f32
 * [scl_dwconv1d_wgrad_nslabs(B, n)][k + 1][C] scratch; writes dw [C][k] and (optional) db [C], slabs summed in index order. */
int scl_dwconv1d_fwd(const float* x, const float* w, const float* bias, float* y, int B, int n, int C, int k, int pad_l, int flip, void* stream);
int scl_dwconv1d_wgrad_nslabs(int B, int n);
int scl_dwconv1d_wgrad(const float* x, const float* dy, float* part, float* dw, float* db, int B, int n, int C, int k, int pad_l, void* stream);
/* Shaw's relative positions (conformer.py:98-106: dist = clamp(i - j, -max_pos, max_pos) + max_pos; pos_attn = q . rel_pos_emb(dist)).
 * scl_relpos_gather: Eu[r'] = E[clamp(r' - (n - 1)) + max_pos] for the 2n - 1 distances that occur (rows 2n - 1 .. Nr - 1 zero), so that
 * R = q Eu^T is one GEMM and pos_attn[i][j] = R[i][i - j + n - 1].  scl_relpos_softmax_fwd: P[(b,h,i)][j] = softmax_j(scale S + scale R
 * skewed), optional byte mask [B][n] (pairs with mask[b][i] & mask[b][j] == 0 take -FLT_MAX: conformer.py:108-113), columns n .. ldP - 1
 * zero; n <= 1024.  scl_relpos_softmax_bwd: dS (pitch ldP) and the skewed dR (pitch ldR), every element written.
 * scl_relpos_scatter_grad: dE [2 max_pos + 1][D] from dEu [2n - 1][D] (the clamped ends sum their run in order). */
int scl_relpos_gather(const float* E, float* Eu, int n, int Nr, int D, int max_pos, void* stream);
int scl_relpos_scatter_grad(const float* dEu, float* dE, int n, int D, int max_pos, void* stream);
int scl_relpos_softmax_fwd(const float* S, const float* R, const uint8_t* mask, float* P, int B, int H, int n, int ldS, int ldR, int ldP,
                           float scale, void* stream);
int scl_relpos_softmax_bwd(const float* P, const float* dP, const uint8_t* mask, float* dS, float* dR, int B, int H, int n, int ldP, int ldR,
                           float scale, void* stream);

/* ---- BTSE plugin (csrc/btse.hip; BASELINE.json configs[4]: model/wav2vec2_btse/model.py:210-238,321-343) ---------------------------
 * The "bio" branch of the reference's wav2vec2_btse Model: bioEncoderTransformersmall = Embedding(n_bios, 32) * sqrt(32) ->
 * transformer.Encoder (transformer.py:17-52: n_layers post-LN layers of MultiHeadAttention with window_size 4 relative keys AND values
 * shared by the heads, :105-260, and a kernel-1 ReLU FFN, :261-306; LayerNorm over channels modules.py:27-39) -> Conv1d(32, bio_out, 1)
 * -> the LAST padded position times its mask (model.py:234-236).  One workgroup per utterance runs the whole encoder: K / V of the
 * layer in LDS, one wave per query row (the L x L scores never leave registers), weights in registers per phase, everything fp32.
 * Shapes served: bio_dim 32, 4 heads of 8, pf_dim 128, window 4, 1..8 layers, 1 <= L <= 512 tokens, bio_out <= 256; anything else is
 * refused (SCL_EUNSUPPORTED) — scl_btse_bio_supported answers without launching.
 *   lw[l][]: Wq bq Wk bk Wv bv Wo bo emb_rel_k emb_rel_v gamma1 beta1 W1 c1 W2 c2 gamma2 beta2 of layer l, torch layouts ([out][in] rows,
 *            emb_rel_* [9][8]); go[l][] / go_emb / go_Ws / go_bs: element offsets of their gradients inside one slab row.
 *   bio [B][L] int32 tokens (clamped to [0, n_bios)), lens [B] int32; out [B][out_ld] f32: columns 0..bio_out-1 of row b are written
 *   (out may point into the concatenated [emb | bio] row, model.py:333).
 *   ws: f32 scratch, scl_btse_bio_ws_floats(n_layers, L) per utterance (the forward leaves every layer's activations there for the
 *   backward; floats [n_layers * 392 * L, +32 L) of an utterance hold the encoder's output x * mask, transformer.py:51).
 * scl_btse_bio_bwd: d_out [B][dout_ld] -> slab [B][slab_ld]: row b = utterance b's contribution to every parameter gradient at the
 * go_* offsets (zeros where the read-out position is padding: model.py:236 multiplies by the mask); sum the rows in index order
 * (scl_reduce_slabs_f32) for the batch gradient. */
typedef struct SclBtseBio {
    const float* emb; const float* lw[8][18]; const float* Ws; const float* bs;
    const int32_t* bio; const int32_t* lens;
    float* ws; float* out; const float* d_out; float* slab;
    int64_t ws_stride, slab_ld;
    int32_t go[8][18]; int32_t go_emb, go_Ws, go_bs;
    int32_t n_layers, n_bios, bio_out, L, B, out_ld, dout_ld, bio_dim, n_heads, pf_dim, window;
} SclBtseBio;
int scl_btse_bio_supported(int bio_dim, int n_heads, int pf_dim, int n_layers, int window, int bio_out, int L);
int64_t scl_btse_bio_ws_floats(int n_layers, int L);
int scl_btse_bio_fwd(const SclBtseBio* p, void* stream);
int scl_btse_bio_bwd(const SclBtseBio* p, void* stream);
/* model.py:329-335, the join in front of fc2.  is_add == 0: b[r] = [emb[r] (C) | s[r] (bio_out)] — emb is copied, the s columns are
 * expected in place already (scl_btse_bio_fwd writes them), bwd: demb = db[:, :C], ds = db[:, C:].  is_add != 0: b = fc1(emb) + s
 * (fc1: [bio_out][C]); bwd also writes dW1 / db1.  All f32, B <= 4096. */
int scl_btse_join_fwd(const float* emb, const float* s, const float* W1, const float* b1, float* b, int B, int C, int bio_out, int is_add, void* stream);
int scl_btse_join_bwd(const float* db, const float* emb, const float* W1, float* demb, float* ds, float* dW1, float* db1, int B, int C, int bio_out,
                      int is_add, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SCL_HIP_H */
