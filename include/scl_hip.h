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

/* Per-kernel HIP-event profiling used by bench.py's roofline leg: when enabled, every launch of
 * the kernel family `kid` (SCL_KID_*) is bracketed by two hipEvents on the launch stream.
 * scl_prof_read synchronises those events and returns launch count and summed milliseconds. */
#define SCL_KID_GEMM 0
#define SCL_KID_MAX  8
int scl_prof_enable(int kid, int on);
int scl_prof_read(int kid, int64_t* n_launches, double* total_ms, double* total_flops);

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
#define SCL_GEMM_ACT_SHIFT   8        /* 0 none, 1 gelu(erf), 2 relu, 3 leaky_relu(0.01) */
#define SCL_GEMM_RMODE_SHIFT 12       /* 0 none, 1 C += R, 2 C *= act'(R) with act = RACT */
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
} SclGemmDesc;

/* C[z][m][n] = epilogue( alpha * sum_k A[z][m][k] * B[z][n][k] ), fp32 accumulate on MFMA.
 * Replaces: F.linear / nn.Conv1d / torch.bmm calls inside fairseq Wav2Vec2Model.forward
 * (model/xlsr.py:41), nn.Linear in model/wav2vec2_linear_nll.py:107,49-67 and their autograd
 * backward (main.py:79).                                                                       */
int scl_gemm_bf16(const SclGemmDesc* desc, void* stream);

/* out[i] = sum_s slabs[s*stride + i]  (deterministic split-K combine). */
int scl_reduce_slabs_f32(const float* slabs, float* out, int64_t n, int nslabs, int64_t stride, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SCL_HIP_H */
