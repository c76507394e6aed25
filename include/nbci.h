/* nbci.h — C-ABI of the MI355X-native NDT1 hot path (libnbci.so).
 *
 * The reference (colehurwitz/llm_bci) is pure Python: it has no FFI of its own, so every
 * entry point below is new. Each one names the reference call(s) it replaces
 * (file:line relative to the reference repo root). The Python host side
 * (llm_bci_amd/) binds these with ctypes; INTEGRATION.md shows the stub a reference
 * maintainer would add.
 *
 * Conventions
 *  - plain pointers + sizes, no torch types. Device pointers unless noted "host".
 *  - every launch takes an explicit stream (hipStream_t passed as void*); no entry point
 *    synchronises or allocates.
 *  - return 0 on success, negative on error; nbci_last_error() returns a thread-local
 *    message. Nothing aborts the process.
 *  - the library BORROWS every pointer for the duration of the call; the only owned state
 *    is the opaque plan (create/destroy).
 */
#ifndef NBCI_H
#define NBCI_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* nbci_stream_t; /* hipStream_t */

#define NBCI_VERSION 100

/* status codes */
#define NBCI_OK 0
#define NBCI_EINVAL (-1)
#define NBCI_ESHAPE (-2)
#define NBCI_EALIGN (-3)
#define NBCI_EWORKSPACE (-4)
#define NBCI_EHIP (-5)

/* activation ids (transformers ACT2FN names used by configs/ndt1.yaml:46,65) */
#define NBCI_ACT_NONE 0
#define NBCI_ACT_SOFTSIGN 1
#define NBCI_ACT_GELU 2
#define NBCI_ACT_RELU 3
#define NBCI_ACT_TANH 4

/* dtype ids */
#define NBCI_F32 0
#define NBCI_BF16 1

int nbci_version(void);
const char* nbci_last_error(void);

/* ------------------------------------------------------------------------------------
 * GEMM: C[z] = alpha * A[z] (M x K) . B[z]^T (N x K) with fused epilogue.
 * Replaces every nn.Linear / matmul on the path: models/ndt1.py:173 (embed), :180
 * (Unfold + stack_projection, via the row-offset view below), :280-282 (q,k,v), :292
 * (out_proj), :226-227 (MLP), :494 (decoder), the two matmuls inside
 * F.scaled_dot_product_attention (:289), and their autograd transposes.
 *
 * Operand storage: a row-major matrix of "storage rows". kmajor=1: storage row = m (or n)
 * index, columns = k. kmajor=0: storage row = k index, columns = m (or n).
 * Storage row r starts at element offset
 *     rpb ? (r / rpb) * gstride + (r % rpb) * ld : r * ld
 * so overlapping sliding windows (nn.Unfold, ndt1.py:138) are a view, never materialised.
 * Batch z (0 <= z < batch) adds (z / zdiv) * zs1 + (z % zdiv) * zs2.
 * Epilogue order: acc*alpha (+bias[n]) -> [store C2 = pre-activation] -> act -> dropout ->
 * (+residual[m][n]) -> (+beta*C) -> store C.   splitk > 1: C(f32) += partial via atomics;
 * then only alpha is honoured and the caller zero-fills C first.
 */
typedef struct nbci_operand {
    const void* ptr;
    int64_t ld;
    int32_t kmajor;
    int32_t rpb;
    int64_t gstride;
    int64_t zs1;
    int64_t zs2;
} nbci_operand;

typedef struct nbci_gemm_desc {
    int32_t M, N, K;
    int32_t in_dtype; /* NBCI_F32 (exact f32 MFMA) or NBCI_BF16 */
    nbci_operand A, B;
    void* C;
    void* C2; /* optional pre-activation copy (same dtype/ld as C) */
    int64_t ldc;
    int64_t czs1, czs2;
    int32_t c_dtype;
    int32_t batch;
    int32_t zdiv;
    int32_t splitk;
    float alpha;
    float beta;
    const float* bias;     /* [N] or NULL */
    int32_t act;           /* NBCI_ACT_* */
    float drop_p;          /* 0 = off; keep-scale 1/(1-p) */
    uint32_t seed, site;   /* dropout stream id */
    const float* residual; /* f32 [M][ldr] or NULL */
    int64_t ldr;
} nbci_gemm_desc;

int nbci_gemm(const nbci_gemm_desc* d, nbci_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NBCI_H */
