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
 *     rpb > 0 ? (r / rpb) * gstride + (r % rpb) * ld : r * ld
 * (rpb < 0: uniform rows that OVERLAP by design, ld smaller than a row's extent - the windows of a signal at a fixed hop)
 * so overlapping sliding windows (nn.Unfold, ndt1.py:138) are a view, never materialised.
 * Batch z (0 <= z < batch) adds (z / zdiv) * zs1 + (z % zdiv) * zs2.
 * Epilogue order: acc*alpha (+bias[n]) -> [store C2 = pre-activation] -> (+residual if
 * residual_first) -> act -> (*act'(gate)) -> dropout -> (+residual) -> (+beta*C) -> store C.
 * The dropout draw for an element is keyed by (seed, site, element offset inside C): one 32-bit
 * hash per pair of elements, 16 bits each, drop iff draw < floor(p * 65536).
 * splitk > 1: C(f32) += partial via atomics; then only alpha is honoured.
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
    const void* residual;  /* [M][ldr] or NULL: f32, or bf16 when residual_dtype = NBCI_BF16 (last field) */
    int64_t ldr;
    const int64_t* residual_rows; /* optional gather: residual row index per output row (nn.Embedding add, ndt1.py:189);
                                     batched: indexed by the global row (batch offset czs / ldc + m) */
    int32_t residual_first;       /* 1: residual is added BEFORE act/dropout */
    const void* gate;             /* optional [M][ldg] in in_dtype: result *= act'(gate) (GELU/softsign backward) */
    int64_t ldg;
    int32_t gate_act;             /* < 0: gate already holds act' (see c2_grad): plain multiply; 0..63: NBCI_ACT_* of
                                     the PRE-activation stored in gate; 64 + act: gate holds the activation's OUTPUT
                                     (softsign / relu / tanh derivative from the output) */
    int32_t c2_grad;              /* 1: C2 receives act'(pre-activation) instead of the pre-activation */
    float* colsum;                /* optional f32 [N]: colsum[n] += sum_m C[m][n] of the STORED values (bias
                                     gradient fused into the GEMM that produces the activation gradient) */
    int64_t colsum_rep_stride;    /* colsum replicas (to spread same-address atomics): replica r at colsum + r*stride */
    int32_t colsum_nrep;          /* 0/1 = no replication */
    int32_t gate_follows_c;       /* 1: batched GEMM whose gate has C's layout: the batch offset (czs1/czs2) applies to it too */
    int32_t residual_dtype;       /* storage type of `residual`: NBCI_F32 (0, default) or NBCI_BF16 (a bf16 residual stream: widened,
                                     added in f32, the sum rounded once when C is stored) */
} nbci_gemm_desc;

int nbci_gemm(const nbci_gemm_desc* d, nbci_stream_t stream);
/* n <= 6 independent GEMMs of one operand layout in ONE launch (a layer's weight gradients: few output
 * tiles each, K = all tokens). Problems that do not qualify run one launch each. */
int nbci_gemm_grouped(const nbci_gemm_desc* descs, int32_t n, nbci_stream_t stream);


/* ------------------------------------------------------------------------------------
 * Single kernels (also used on their own by the tests). "dtype" arguments take NBCI_F32 /
 * NBCI_BF16 and name the activation dtype feeding the GEMMs.
 */

/* SmoothAndNoise.forward, models/ndt1.py:92-107: depthwise gaussian taps along T ('same',
 * zero padded) + white / per-trial offset noise (sd = 0 disables). spikes f32 (B,T,N). */
int nbci_smooth_noise(const float* spikes, void* out, int32_t out_dtype, int32_t B, int32_t T, int32_t N,
                      const float* taps, int32_t ntaps, float white_sd, float offset_sd, uint32_t seed,
                      nbci_stream_t stream);

/* nn.LayerNorm(H) forward/backward (ndt1.py:309,311,402). backward: dx (+)= LN'(dy), dw/db += */
int nbci_layernorm_fwd(const float* x, const float* w, const float* b, void* y, int32_t y_dtype, float* mean,
                       float* rstd, int32_t M, int32_t H, nbci_stream_t stream);
int nbci_layernorm_bwd(const float* dy, const float* x, const float* w, const float* mean, const float* rstd,
                       float* dx, float* dw, float* db, int32_t M, int32_t H, int32_t accumulate_dx,
                       nbci_stream_t stream);
/* The general forms the NDT1 train step launches. x: f32, or bf16 (a bf16 residual stream, nbci_ndt1_config.residual_dtype: statistics
 * and normalisation in f32 from the widened row; y must then be bf16). Backward: dy f32 or bf16 (dy_dtype); the gradient stream
 * dx_out = dx_in (NULL: nothing) + LN'(dy) in x's dtype (one buffer or two), summed in f32 and rounded once at the store; optionally the
 * dropout-masked operand copy of dx_out for the next GEMMs (cast_out in cast_dtype; mask keyed by (seed, site, element offset), keep
 * scale 1 / (1 - p)) and the column sums of that copy (cast_colsum f32 [H] +=: the bias gradient of the Linear below). dw / db +=. */
int nbci_layernorm_fwd_ex(const void* x, int32_t x_dtype, const float* w, const float* b, void* y, int32_t y_dtype, float* mean,
                          float* rstd, int32_t M, int32_t H, nbci_stream_t stream);
int nbci_layernorm_bwd_ex(const void* dy, int32_t dy_dtype, const void* x, int32_t x_dtype, const float* w, const float* mean,
                          const float* rstd, const void* dx_in, void* dx_out, float* dw, float* db, int32_t M, int32_t H,
                          void* cast_out, int32_t cast_dtype, float drop_p, uint32_t seed, uint32_t site, float* cast_colsum,
                          nbci_stream_t stream);

/* masked softmax + attention-prob dropout of F.scaled_dot_product_attention (ndt1.py:289) with the
 * mask of ndt1.py:435-437 computed from token validity + context span instead of a (B,T',T') tensor */
int nbci_softmax_fwd(const float* S, void* P, void* Pd, int32_t p_dtype, const int32_t* token_mask, int32_t B,
                     int32_t n_heads, int32_t Tp, int32_t ldS, int32_t ldP, int32_t ctx_forward,
                     int32_t ctx_backward, float drop_p, uint32_t seed, uint32_t site, nbci_stream_t stream);
int nbci_softmax_bwd(const float* dPd, const void* P, void* dS, int32_t p_dtype, int32_t B, int32_t n_heads,
                     int32_t Tp, int32_t ldS, int32_t ldP, float drop_p, uint32_t seed, uint32_t site,
                     nbci_stream_t stream);

/* Fused attention (bf16, head size 128, T' <= 160): F.scaled_dot_product_attention(q,k,v,attn_mask,dropout_p)
 * of ndt1.py:289 on a packed (B*T', 3H) qkv buffer, mask of ndt1.py:435-437 built on the fly, output already in the
 * merged (B*T', H) layout with the attention-output dropout of ndt1.py:292 applied; lse: f32 (B, heads, T') row
 * log-sum-exp of the scaled masked scores, kept for the backward (may be NULL when no backward follows).
 * Backward: out / lse = what the forward wrote; d_out = d loss / d(Pd v) (B*T', H), i.e. with the output dropout's
 * backward already applied -> dqkv (B*T', 3H); dS_ws / Pd_ws: bf16 scratch (B, heads, T', ldP), ldP = round_up(T', 8);
 * bias_grad: optional f32 (3H) += column sums of dqkv. */
int nbci_attention_fwd(const void* qkv, const int32_t* token_mask, void* out, float* lse, int32_t B, int32_t n_heads, int32_t Tp,
                       int32_t H, int32_t ctx_forward, int32_t ctx_backward, float drop_p, uint32_t seed, uint32_t site_prob,
                       uint32_t site_out, nbci_stream_t stream);
int nbci_attention_bwd(const void* qkv, const int32_t* token_mask, const void* out, const float* lse, const void* d_out, void* dS_ws,
                       void* Pd_ws, int32_t ldP, void* dqkv, float* bias_grad, int32_t B, int32_t n_heads, int32_t Tp, int32_t H,
                       int32_t ctx_forward, int32_t ctx_backward, float drop_p, uint32_t seed, uint32_t site_prob, nbci_stream_t stream);

/* Unmasked multi-head attention WITHOUT a score tensor (online softmax), over a packed (NS*S, 3H) q|k|v buffer of NS
 * sequences of S tokens; out (NS*S, H) merged heads; lse (NS*n_heads*S) f32 row log-sum-exp kept for the backward;
 * dropout on the probabilities (site). Replaces eager_attention_forward / nn.MultiheadAttention's core under
 * models/patchtst.py:176 and models/itransformer.py:158-173 (no attention mask on those paths).
 *   small : head size 16 / 32 / 64, dtype f32 or bf16, one thread per query / key, K/V rows through scalar loads
 *   flash : head size 32 / 64 / 96 / 128, bf16, one wave per 32 queries / keys on MFMA (32 x 32 score tiles)
 * The backward writes dqkv (NS*S, 3H); dsum (NS*n_heads*S) f32 is scratch. With drop_p > 0 the flash kernels keep library-owned buffers of dropout keep
 * bits, 4 * NS * n_heads * ceil(S / 32)^2 * 32 bytes each: the forward writes one per LAYER, found again by the backward under the same lse pointer (and
 * only used when seed, site, drop_p and the shape match - otherwise the backward draws the bits itself into one buffer per (device, stream)). Allocated on
 * first use, grown on demand (a stream synchronisation then), freed by nbci_release_scratch(). lse must therefore be the same buffer in the forward and the
 * backward of a layer (it has to be anyway: the backward reads it). */
int nbci_attention_small_fwd(const void* qkv, void* out, float* lse, int32_t dtype, int32_t NS, int32_t n_heads, int32_t S, int32_t H,
                             float drop_p, uint32_t seed, uint32_t site, nbci_stream_t stream);
int nbci_attention_small_bwd(const void* qkv, const void* out, const void* d_out, const float* lse, float* dsum, void* dqkv, int32_t dtype,
                             int32_t NS, int32_t n_heads, int32_t S, int32_t H, float drop_p, uint32_t seed, uint32_t site,
                             nbci_stream_t stream);
int nbci_attention_flash_fwd(const void* qkv, void* out, float* lse, int32_t NS, int32_t n_heads, int32_t S, int32_t H, float drop_p,
                             uint32_t seed, uint32_t site, nbci_stream_t stream);
int nbci_attention_flash_bwd(const void* qkv, const void* out, const void* d_out, const float* lse, float* dsum, void* dqkv, int32_t NS,
                             int32_t n_heads, int32_t S, int32_t H, float drop_p, uint32_t seed, uint32_t site, nbci_stream_t stream);

/* BCI coupler splice (models/bci.py:143-166): per example b, out[b] = cat(text[b,:d_b], spikes[b], text[b,d_b:]) for
 * embeddings (B,Lt,H)+(B,Ts,H) -> (B,Lt+Ts,H) in `dtype`, the attention mask (text mask / spike validity) and the
 * targets (-100 over the spike span). int64 masks/targets as the reference collates them. text/targets may be NULL.
 * The backward scatters d_out back to d_text (optional) and d_spikes. */
int nbci_coupler_splice_fwd(const void* text, const void* spikes, void* out, int32_t dtype, const int64_t* text_mask,
                            const int64_t* spikes_valid, int64_t* mask_out, const int64_t* targets, int64_t* targets_out,
                            const int64_t* split, int32_t B, int32_t Lt, int32_t Ts, int32_t H, nbci_stream_t stream);
int nbci_coupler_splice_bwd(const void* d_out, void* d_text, void* d_spikes, int32_t dtype, const int64_t* split, int32_t B,
                            int32_t Lt, int32_t Ts, int32_t H, nbci_stream_t stream);

/* out[n] += sum_m in[m][n] (f32 accumulate; `in` is (M, ld) in in_dtype): the bias gradient of a Linear whose output gradient
 * does not come out of one of this library's GEMMs (the BCI projector's last Linear, models/bci.py:88-96, whose output gradient
 * arrives from the LLM's autograd through the splice backward). */
int nbci_colsum(const void* in, int32_t in_dtype, int64_t ld, int32_t M, int32_t N, float* out, nbci_stream_t stream);

/* MX-scaled fp8 (OCP e4m3) projection, BASELINE configs[4] "fp8 MFMA QKV": replaces the q / k / v nn.Linear of the HF PatchTST
 * attention the reference's encoder runs (models/patchtst.py:176,223-225) in the forward pass.
 * nbci_mx_quantize: rows x K values (f32 or bf16, row stride ldx; K % 32 == 0) -> e4m3 codes (rows x K bytes) + one E8M0 scale byte
 * per 32 consecutive k (rows x K/32 bytes): X = 2^(floor(log2 amax) - 8), codes = RNE(v / X) saturated at +-448.
 * nbci_gemm_fp8: C[M][N] (c_dtype, row stride ldc) = dequant(A8, sA) . dequant(W8, sW)^T + bias on v_mfma_scale_f32_16x16x128_f8f6f4
 * (f32 accumulate; K = 128, 256, 384 or 512; operands 16-byte aligned). */
int nbci_mx_quantize(const void* x, int32_t dtype, int64_t ldx, void* codes, void* scales, int64_t rows, int32_t K, nbci_stream_t stream);
int nbci_gemm_fp8(const void* A8, const void* sA, const void* W8, const void* sW, const float* bias, void* C, int32_t c_dtype, int64_t M, int32_t N,
                  int32_t K, int64_t ldc, nbci_stream_t stream);

/* nn.LogSoftmax(-1) of the decoder (ndt1.py:499) + argmax path (main.py:69) */
int nbci_logsoftmax(const float* logits, int32_t ldl, float* preds, int32_t* argmax, int32_t M, int32_t V,
                    nbci_stream_t stream);

/* nn.CTCLoss(reduction="none", blank, zero_infinity) forward + gradient wrt logits (ndt1.py:517,581).
 * preds (B,Tp,V) f32 log-probs; in_lens int32 (B); targets int64 (B,S); alpha_ws >= nbci_ctc_workspace_floats.
 * dlogits (B*Tp, ldd) in d_dtype, may be NULL (loss only). */
int64_t nbci_ctc_workspace_floats(int32_t B, int32_t Tp, int32_t S);
int nbci_ctc(const float* preds, const int64_t* targets, const int32_t* in_lens, const int64_t* tgt_lens, int32_t B,
             int32_t Tp, int32_t V, int32_t S, int32_t blank, int32_t zero_infinity, float* loss, float* alpha_ws,
             void* dlogits, int32_t d_dtype, int32_t ldd, float grad_scale, nbci_stream_t stream);

/* format_ctc + word_error_count (utils/eval_bci.py:11-48, main.py:68-74), integer exact.
 * decoded (B,Tp) int32 padded -1, dec_lens (B), errors (B,2) = {edit distance, target tokens};
 * scratch int32 B*2*(S+2). */
int nbci_per(const int32_t* argmax, const int64_t* targets, const int64_t* tgt_lens, int32_t B, int32_t Tp, int32_t S,
             int32_t blank, int32_t* decoded, int32_t* dec_lens, int32_t* errors, int32_t* scratch,
             nbci_stream_t stream);

/* Per-step bookkeeping of Trainer.train (models/trainer.py:353-362) in one launch, no host sync: stats (4 x f64 on the
 * device) accumulates {sum of the per-sample losses, n_examples, per-batch PER ratio sum(errors[:,0]) / sum(errors[:,1]),
 * number of batches}; errors = nbci_per's (B,2) output or NULL (no metric this step). */
int nbci_step_stats(double* stats, const float* loss, int32_t B, double n_examples, const int32_t* errors, nbci_stream_t stream);

/* torch.optim.AdamW step over a flat buffer (models/trainer.py:229,340); bc1/bc2 = 1 - beta^t.
 * p_lp: optional bf16 shadow of p refreshed in the same pass. g is multiplied by grad_scale first
 * (1/world_size turns an all-reduce SUM into DDP's mean). */
int nbci_adamw(float* p, const float* g, float* m, float* v, void* p_lp, int64_t n, float lr, float beta1,
               float beta2, float eps, float weight_decay, float bc1, float bc2, float grad_scale,
               nbci_stream_t stream);
/* the same step with the gradient in bf16 (a bucket as a bf16 all-reduce left it: no widening pass back into the f32 gradient buffer) */
int nbci_adamw_lp(float* p, const void* g_bf16, float* m, float* v, void* p_lp, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, float bc1, float bc2, float grad_scale,
                  nbci_stream_t stream);
/* the same step + optimizer.zero_grad() (trainer.py:340-342) in one pass: g is cleared as it is consumed. max_blocks > 0 caps
 * the launch's workgroups (a caller running it on a second stream beside other kernels leaves them wave slots); 0 = default. */
int nbci_adamw_zero(float* p, float* g, float* m, float* v, void* p_lp, int64_t n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, float bc1, float bc2, float grad_scale,
                    int32_t max_blocks, nbci_stream_t stream);

int nbci_cast(const float* in, void* out, int32_t out_dtype, int64_t n, nbci_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Data-parallel exchange for a host that is not PyTorch (the Python host uses torch.distributed, backend "nccl" = RCCL):
 * replaces accelerate -> DDP's bucketed gradient all-reduce (models/trainer.py:260-262,339). One communicator per process /
 * GPU over the node's xGMI mesh. Rank 0 calls nbci_comm_unique_id, ships the 128 bytes to the other ranks by any host channel,
 * then every rank calls nbci_comm_create. nbci_allreduce_bucket all-reduces (SUM, in place) one contiguous gradient range of
 * the flat buffer (nbci_ndt1_segment_range) on `stream` - call it right after nbci_ndt1_backward of that segment so the exchange
 * overlaps the rest of the backward - and DDP's mean is nbci_adamw's grad_scale = 1 / world_size. dtype NBCI_F32 or NBCI_BF16.
 * librccl.so is opened on first use (dlopen): libnbci.so itself does not depend on it. */
typedef void* nbci_comm;
int nbci_comm_unique_id(void* id128);
int nbci_comm_create(nbci_comm* comm, int32_t world_size, int32_t rank, const void* id128);
void nbci_comm_destroy(nbci_comm comm);
int nbci_allreduce_bucket(nbci_comm comm, void* buf, int64_t n, int32_t dtype, nbci_stream_t stream);

/* How many CUs the GEMM tile cost model may count on (default 256). An overlapped RCCL all-reduce parks its channels' workgroups
 * on some CUs; grids sized for exactly one or two rounds of 256 CUs then spill into an extra, nearly empty round. With the
 * number lowered the model prefers tile heights whose rounds fit the CUs that are left (tools/dp_cu_footprint.py measures it). */
int nbci_set_available_cus(int32_t cus);
/* Measurement aid: park n_workgroups workgroups (256 threads, lds_bytes of LDS each) for `microseconds` on `stream`. */
int nbci_debug_occupy_cus(int32_t n_workgroups, int32_t lds_bytes, double microseconds, nbci_stream_t stream);

/* Measurement aid (no reference counterpart): when enabled, every GEMM launched by
 * nbci_ndt1_forward/backward is bracketed by HIP events on its own stream. collect() waits for
 * them and fills out24[kind*3 + {0,1,2}] = {total ms, total FLOPs, launches}, kind =
 * (bf16 ? 4 : 0) + (A.kmajor ? 2 : 0) + (B.kmajor ? 1 : 0). Host-synchronising; never call it in
 * a timed region. */
int nbci_profile_enable(int32_t on);
/* Measurement / test aid: which bf16 GEMM kernel family nbci_gemm picks for eligible shapes (k-major A, K % 64 == 0, N >= 256):
 * 0 = the two-workgroup-per-CU kernels only, 1 = the producer / consumer kernel (144 x 256 tiles, gemm_pc.hip) where the tile
 * cost model prefers it (default), 2 = the producer / consumer kernel whenever eligible. Initial value: NBCI_GEMM_PC. */
int nbci_debug_gemm_pc(int32_t mode);
/* prototype (round 4): the MLP half of an encoder layer - up projection (bias + activation, act' copy) and down projection (bias / dropout /
 * residual), reference models/ndt1.py:224-227,328 - in ONE launch with the row strip resident on its CU; same results as the two nbci_gemm calls.
 * bf16, k-major operands, up.K % 64 == 0, up.N % 128 == 0 (<= 1024), down.N % 128 == 0, at most 40 rows per CU. csrc/mlp_strip.hip */
int nbci_debug_mlp_strip(const nbci_gemm_desc* up, const nbci_gemm_desc* down, nbci_stream_t stream);
/* Measurement / test aid: how nbci_gemm_grouped launches a group of direct-to-LDS problems (K % 64 == 0): 0 = one workgroup per
 * 128 x 128 output tile, 1 = deal the K tiles of all output tiles out evenly over the chip's workgroup slots when one workgroup
 * per tile would leave more than 12 % of the slot-rounds empty (default; gemm_streamk.hip: partial tiles go through a scratch buffer
 * of 64 KB per slot that the library keeps per (device, stream); fixed summation order, no atomics), 2 = always, 3 = always and with the owner / helper ("aligned") scheme wherever it applies, 4 = always and with the block-per-XCD scheme wherever it applies. Initial value:
 * NBCI_STREAMK. The reference has no counterpart (a layer's weight gradients are four cuBLAS calls inside loss.backward(),
 * models/trainer.py:339). */
int nbci_debug_gemm_streamk(int32_t mode);
/* Host-only (no launch, no device access; works without a GPU): how nbci_gemm_grouped would launch this group under the current mode and
 * nbci_set_available_cus. out8 = {1 if the K tiles are dealt out over all workgroup slots / 0 one workgroup per tile / -1 one launch per
 * problem, scheme (0 contiguous runs, 1 owner + helper "aligned", 2 block per XCD), workgroups, owner K tiles, remainder K tiles, scratch
 * slots of 64 KB, output tiles, K tiles per output tile of problem 0}. The pointers in the descriptors are only checked for alignment. */
int nbci_debug_gemm_grouped_plan(const nbci_gemm_desc* descs, int32_t n, int32_t* out8);
/* Everything queued on `before` so far happens-before what is queued on `after` from here on: an event record + stream wait whose event carries
 * no system-scope fence (both streams are on the calling thread's current device; kernel boundaries publish at agent scope). The reference's
 * counterpart is implicit (one stream); torch's Stream.wait_stream records a fenced event, whose cache write-back sits between two kernels
 * of the main stream every time. Host visibility still needs a stream / device synchronisation. */
int nbci_stream_order(nbci_stream_t before, nbci_stream_t after);
/* Frees the scratch buffers the library allocated on its own (the grouped GEMM's partial tiles). Call with the streams idle. */
int nbci_release_scratch(void);
/* The balanced grouped launch's owners wait for other workgroups' partial tiles with a BOUNDED spin; *out = how many gave up since the
 * scratch buffers were created (0 in a healthy run; a non-zero count means wrong weight-gradient tiles were stored instead of a hang).
 * Synchronises the streams that own scratch. */
int nbci_streamk_timeouts(int64_t* out);
int nbci_profile_collect(double* out24);
/* every kernel a profiling scope brackets (GEMMs, LayerNorm, attention, AdamW, ...), aggregated by kernel symbol: one text line per
 * symbol "symbol<TAB>launches<TAB>total ms<TAB>algorithmic flops<TAB>algorithmic bytes". Drains the records (as nbci_profile_collect does). */
int nbci_profile_collect_text(char* buf, int64_t cap);

/* ------------------------------------------------------------------------------------
 * NDT1-CTC model level. Replaces NeuralEncoder.forward + NDT1.forward(ctc) and their autograd
 * graph (models/ndt1.py:408-450, 523-589), i.e. what Trainer.train calls at trainer.py:336-339.
 */
typedef struct nbci_ndt1_config { /* configs/ndt1.yaml, flattened */
    int32_t n_channels, input_dim, stack_size, stack_stride, hidden, n_layers, n_heads, inter, vocab, max_F;
    float smooth_sd;                 /* <= 0: no smoothing */
    int32_t noise;
    float white_noise_sd, constant_offset_sd;
    int32_t embed_act, mlp_act;      /* NBCI_ACT_* */
    float embed_dropout, dropout;
    int32_t use_rope;
    float rope_theta;
    int32_t context_forward, context_backward;
    int32_t pos;
    int32_t blank_id, zero_infinity;
    int32_t dtype;                   /* NBCI_F32: exact-f32 parity path; NBCI_BF16: bf16 MFMA, f32 accumulate */
    /* NeuralFactorsProjection (models/ndt1.py:348-373, configs/ndt1.yaml factors.*): encoder output =
     * act(Linear(hidden -> factors_size)(out_norm(x))) instead of out_norm(x); 0 = inactive (identity). The decoder then
     * reads factors_size inputs, hidden_out / d_hidden are (B,T',factors_size). factors.dropout must be 0. */
    int32_t factors_size, factors_act, factors_bias;
    /* embedder.adapt (models/ndt1.py:124-129,170-171): adapt_days > 0 = one embed_spikes Linear per recording day
     * ("encoder.embedder.embed_spikes.<d>.weight / .bias"), picked per sample by io.day_idx; 0 = one shared layer. */
    int32_t adapt_days;
    /* embedder.day_token / block_token (models/ndt1.py:151-155,192-201,444-448): a learned token per recording day / block put in
     * front of the spike tokens ([day, block, tokens...]), always attendable, dropped again after out_norm. Table sizes (n_days /
     * n_blocks); 0 = off. io.day_idx / io.block_idx pick the rows. Not with use_rope (the reference fails there). */
    int32_t day_token_days, block_token_blocks;
    /* storage of the residual stream x (the saved LayerNorm inputs) and of its gradient stream between kernels. NBCI_F32 (0, default):
     * what bf16 autocast keeps in f32 in the reference (ndt1.py:325,328: x = x + branch). NBCI_BF16 (dtype bf16 only): both streams are
     * stored in bf16, every kernel widens them, adds / normalises in f32 and rounds ONCE at its store: half the bytes of the
     * HBM-bound kernels (LayerNorm forward / backward, the residual epilogues of out_proj and down_proj). */
    int32_t residual_dtype;
} nbci_ndt1_config;

typedef struct nbci_ndt1_io {
    int32_t B, T, S;                    /* batch, padded bins, padded target length (0 if no targets) */
    const float* spikes;                /* (B,T,N) f32, as pad_collate_fn lays it out (datasets.py:236-272) */
    const int64_t* spikes_mask;         /* (B,T) */
    const int64_t* spikes_timestamp;    /* (B,T) */
    const int64_t* spikes_lengths;      /* (B) */
    const int64_t* targets;             /* (B,S) or NULL */
    const int64_t* targets_lengths;     /* (B) */
    const float* rope_cos;              /* (max_F, head) f32 tables when use_rope */
    const float* rope_sin;
    int32_t train;                      /* module.training: enables noise + dropout */
    int32_t want_grad;                  /* also produce d loss / d logits for nbci_ndt1_backward */
    uint32_t seed;                      /* per-step RNG stream */
    float grad_scale;                   /* loss scale (1/gradient_accumulation_steps, trainer.py:339) */
    float* preds;                       /* out (B,T',V) f32 log-probs */
    float* loss;                        /* out (B) per-sample CTC loss (sum it for NDT1Output.loss) */
    int32_t* argmax;                    /* out (B,T') greedy path or NULL */
    void* hidden_out;                   /* out (B,T',H or factors_size) encoder output in cfg.dtype, or NULL */
    int32_t* token_mask_out;            /* out (B,T') stacked validity mask (ndt1.py:182-183), or NULL */
    const float* d_hidden;              /* backward: d loss / d hidden_out, f32 (B,T',H). When set, the head segment
                                           starts from it (encoder used as a feature extractor, models/bci.py:125)
                                           instead of the CTC gradient; decoder gradients are not touched. */
    void* workspace;
    int64_t workspace_bytes;
    const int64_t* day_idx;             /* (B) recording day of each sample; required when adapt_days > 0 or day_token_days > 0 */
    const int64_t* block_idx;           /* (B) block of each sample; required when block_token_blocks > 0 */
    int32_t embed_part;                 /* backward of segment 0 only: 0 = whole segment; 1 = the stack-projection / position /
                                           token-table gradients (everything after embed_spikes.* in the flat layout), 2 = the rest.
                                           Lets a data-parallel caller all-reduce the large first part while part 2 computes. */
    nbci_stream_t aux_stream;           /* backward, optional (bf16 mode): a second stream of the caller's. The weight-gradient GEMMs and
                                           the fold of the bias / LayerNorm gradient sums are queued there, beside the data-gradient
                                           chain on `stream` (ordered by events the plan owns). The gradients of the segments of the
                                           call are then complete on aux_stream, not on `stream`: queue their consumer (optimizer,
                                           all-reduce) behind aux_stream, and make `stream` wait for aux_stream before the next
                                           forward on this workspace. NULL: everything on `stream` (the default). Pays when one
                                           launch does not fill the chip (small batches); the reference has no counterpart
                                           (autograd runs trainer.py:339's backward on one stream). */
} nbci_ndt1_io;

typedef void* nbci_ndt1_plan;

int nbci_ndt1_plan_create(const nbci_ndt1_config* cfg, nbci_ndt1_plan* out);
void nbci_ndt1_plan_destroy(nbci_ndt1_plan plan);
/* flat parameter buffer: total elements, number of tensors / segments, per-tensor placement.
 * Names are the reference's state-dict keys; segment 0 = embedder, 1..L = layers, L+1 = head. */
int64_t nbci_ndt1_param_count(nbci_ndt1_plan plan);
int32_t nbci_ndt1_num_params(nbci_ndt1_plan plan);
int32_t nbci_ndt1_num_segments(nbci_ndt1_plan plan);
int nbci_ndt1_param_info(nbci_ndt1_plan plan, int32_t index, char* name, int32_t name_cap, int64_t* offset,
                         int64_t* numel, int32_t* rows, int32_t* cols, int32_t* segment);
int nbci_ndt1_segment_range(nbci_ndt1_plan plan, int32_t seg, int64_t* begin, int64_t* end);
int64_t nbci_ndt1_workspace_bytes(nbci_ndt1_plan plan, int32_t B, int32_t T, int32_t S);
int32_t nbci_ndt1_tokens(nbci_ndt1_plan plan, int32_t T);
int nbci_ndt1_forward(nbci_ndt1_plan plan, const float* params, const void* params_lp, const nbci_ndt1_io* io,
                      nbci_stream_t stream);
/* backward over segments seg_hi..seg_lo (descending); gradients ACCUMULATE into grads (flat, f32) */
int nbci_ndt1_backward(nbci_ndt1_plan plan, const float* params, const void* params_lp, const nbci_ndt1_io* io,
                       float* grads, int32_t seg_hi, int32_t seg_lo, nbci_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Masker.forward (models/masker.py:44-104, and the three extra modes of "models/masker copy.py":81-104,117,133) on the device.
 * The mask bit of an element comes from the counter RNG (site), keyed by the index the mode makes it constant along:
 * temporal = (b,t) (widened by `timespan`, the reference's expand_timesteps :107-110), neuron = (b,n), random = (b,t,n);
 * TABLE_BN / TABLE_N / TABLE_T take the Bernoulli probability from `probs` ((B,N): `region`, `inter-region`, `intra-region`
 * - the host maps region names and draws the region sample; (N): `co-smooth`; (T): `forward-pred`);
 * GIVEN uses `ext_mask` as is (tests, replay). `target_bn` ((B,N), optional): the RETURNED mask is the bit AND target_bn != 0
 * while the corruption below uses the full bit (`intra-region`: every neuron outside the target regions is masked, the
 * targets are the masked bins of the target regions). Then zero_ratio of the masked elements are zeroed (site+1) and
 * random_ratio of the remaining masked ones become U(0, max(out)) (site+2, site+3). `out` may alias `in` (the
 * reference mutates its input); `mask` is overwritten, or OR-ed into when `accumulate` (itransformer.py:324-326).
 * scratch: 4 bytes of device memory. */
enum { NBCI_MASK_TEMPORAL = 0, NBCI_MASK_NEURON = 1, NBCI_MASK_RANDOM = 2, NBCI_MASK_TABLE_BN = 3, NBCI_MASK_TABLE_N = 4,
       NBCI_MASK_GIVEN = 5, NBCI_MASK_TABLE_T = 6 };
typedef struct nbci_masker_desc {
    int32_t B, T, N;
    int32_t mode;
    float ratio;               /* Bernoulli probability (temporal: already divided by timespan, masker.py:59) */
    int32_t timespan;          /* temporal mode: width of the expansion (1 = none) */
    float zero_ratio, random_ratio;
    const float* probs;
    const int64_t* ext_mask;   /* (B,T,N) for NBCI_MASK_GIVEN */
    uint32_t seed, site;
    const float* in;           /* (B,T,N) f32 */
    float* out;                /* (B,T,N) f32 */
    int64_t* mask;             /* (B,T,N) int64 0/1 */
    int32_t accumulate;
    void* scratch;
    const float* target_bn;    /* optional (B,N): see above */
} nbci_masker_desc;
int nbci_masker(const nbci_masker_desc* desc, nbci_stream_t stream);

/* ------------------------------------------------------------------------------------
 * iTransformer SSL model level: iTransformerEncoder.forward (`mlp` embedder) + iTransformer.forward, method
 * "mlm" (models/itransformer.py:175-210, 312-359) and their autograd graph. Maskers run before it (nbci_masker).
 * Post-norm torch.nn.TransformerEncoderLayer semantics (packed in_proj, dropout on attention probabilities,
 * dropout1/2, inner FFN dropout), no attention mask, CLS token first, mlm decoder on the channel tokens.
 */
enum { NBCI_LOSS_POISSON_LOG = 0, NBCI_LOSS_POISSON_RATE = 1, NBCI_LOSS_MSE = 2 };
typedef struct nbci_itr_config { /* configs/itransformer.yaml, flattened */
    int32_t max_n_bins;              /* T: the embedding MLP's input width (must be a multiple of 4) */
    int32_t hidden, n_heads, n_layers;
    int32_t max_n_channels;          /* 0: no channel embeddings */
    int32_t n_regions;               /* 0: embed_region false */
    int32_t act;                     /* NBCI_ACT_RELU (encoder + embedder activation) */
    int32_t dec_act;
    float embed_dropout, dropout;
    int32_t use_cls, mlp_decoder;
    int32_t loss;                    /* NBCI_LOSS_* (poisson_nll with log_input true/false, mse) */
    int32_t dtype;
    int32_t residual_dtype;          /* storage of the LayerNorm inputs r1 / r2 and of the gradient streams between kernels: NBCI_F32 (default),
                                      * or NBCI_BF16 (dtype bf16 only; f32 arithmetic, one rounding per store - as nbci_ndt1_config.residual_dtype).
                                      * With bf16 the residual a layer adds is the bf16 LayerNorm output its GEMMs read (no f32 copy of it). */
    int32_t embed_depth;             /* 1: depth_embeddings = LayerNorm(Linear(act(Linear(neuron_depths)))) added to the channel tokens
                                      * (itransformer.py:143-150,200-202); io.neuron_depths required */
    int32_t emb_mode;                /* 0: `mlp` embedder (itransformer.py:108-118); 1: `transformer` = UnivariateTransformer (:40-93) + embed_proj
                                      * (:119-124): every (sample, channel) is a sequence [cls | its max_n_bins bins], token = Linear(act(Linear(count)))
                                      * + embed_pos[timestamp], a post-norm encoder of emb_layers layers, the CLS output projected to `hidden` */
    int32_t emb_hidden, emb_heads, emb_layers;   /* configs/itransformer.yaml encoder.embedder.{hidden_size, n_heads, n_layers} (emb_mode 1);
                                                  * its layers' dropout is embed_dropout, its activation `act` (relu) */
} nbci_itr_config;

typedef struct nbci_itr_io {
    int32_t B, N;                       /* batch, channels (tokens = N + use_cls) */
    const float* spikes;                /* (B,T,N) f32 untouched input = the mlm targets (itransformer.py:319) */
    const float* masked;                /* (B,T,N) f32 after the maskers */
    const int64_t* mask;                /* (B,T,N) OR of the maskers' masks */
    const int64_t* spikes_mask;         /* (B,T) padding mask */
    const int64_t* spikes_spacestamp;   /* (B,N) channel ids or NULL = arange(N) */
    const int64_t* region_idx;          /* (B,N) region ids when n_regions > 0 */
    const int64_t* spikes_timestamp;    /* (B,T) bin timestamps, values < max_n_bins (emb_mode 1: rows of embed_pos), or NULL = arange(T) */
    const float* neuron_depths;         /* (B,N) f32 when embed_depth */
    int32_t train, want_grad;
    uint32_t seed;
    float grad_scale;
    float* preds;                       /* out (B,T,N) f32 */
    int64_t* mask_out;                  /* out (B,T,N): mask & spikes_mask (itransformer.py:343) */
    float* loss;                        /* out (1): masked loss sum */
    int64_t* n_examples;                /* out (1): mask_out.sum() */
    void* hidden_out;                   /* out (B,S,H) final-norm output in cfg.dtype, or NULL */
    void* workspace;
    int64_t workspace_bytes;
} nbci_itr_io;

typedef void* nbci_itr_plan;
int nbci_itr_plan_create(const nbci_itr_config* cfg, nbci_itr_plan* out);
void nbci_itr_plan_destroy(nbci_itr_plan plan);
/* flat parameter buffer as for NDT1: segment 0 = embedding side, 1..L = layers, L+1 = final norm + decoder */
int64_t nbci_itr_param_count(nbci_itr_plan plan);
int32_t nbci_itr_num_params(nbci_itr_plan plan);
int32_t nbci_itr_num_segments(nbci_itr_plan plan);
int nbci_itr_param_info(nbci_itr_plan plan, int32_t index, char* name, int32_t name_cap, int64_t* offset, int64_t* numel,
                        int32_t* rows, int32_t* cols, int32_t* segment);
int nbci_itr_segment_range(nbci_itr_plan plan, int32_t seg, int64_t* begin, int64_t* end);
int64_t nbci_itr_workspace_bytes(nbci_itr_plan plan, int32_t B, int32_t N);
int nbci_itr_forward(nbci_itr_plan plan, const float* params, const void* params_lp, const nbci_itr_io* io, nbci_stream_t stream);
int nbci_itr_backward(nbci_itr_plan plan, const float* params, const void* params_lp, const nbci_itr_io* io, float* grads,
                      int32_t seg_hi, int32_t seg_lo, nbci_stream_t stream);

/* ------------------------------------------------------------------------------------
 * PatchTST model level: PatchTSTForSpikingActivity.forward (models/patchtst.py:214-255) with PredictHead (ctc, :70-94)
 * or PretrainHead (mlm, :139-154) over the encoder the reference takes from HF transformers (PatchTSTModel,
 * patchtst.py:8,176): NOP scaler, patchify, random patch masking, shared patch embedding + fixed sincos positions,
 * pre-norm layers [BatchNorm1d over all (b,c,p) rows -> MHA over the patches of one channel -> residual;
 * BatchNorm1d -> Linear, GELU, Dropout, Linear -> residual]. share_embedding / share_projection true,
 * channel_attention false, norm_type batchnorm, pooling mean (the configs/patchtst.yaml settings).
 *
 * Besides the flat trainable parameters there is an `aux` f32 buffer the state dict also carries:
 *   [position_enc (P,D) | per layer: norm1 running_mean, running_var, norm3 running_mean, running_var (D each)]
 * and an int64 buffer nbt[2L] (num_batches_tracked). Train-mode forwards update the running statistics in place.
 */
enum { NBCI_PTST_CTC = 0, NBCI_PTST_MLM = 1 };
typedef struct nbci_ptst_config { /* configs/patchtst.yaml, flattened */
    int32_t num_input_channels, context_length, patch_length, patch_stride, num_hidden_layers, d_model, num_attention_heads, ffn_dim;
    float norm_eps, attention_dropout, positional_dropout, path_dropout, ff_dropout;
    int32_t act;                       /* NBCI_ACT_GELU */
    int32_t do_mask_input;
    double random_mask_ratio;          /* double: len_keep = int(P * (1 - ratio)) must round as the reference's Python float does */
    int32_t channel_consistent_masking;
    float mask_value;
    int32_t method;                    /* NBCI_PTST_CTC | NBCI_PTST_MLM */
    int32_t vocab, blank_id, zero_infinity;
    int32_t mlp_decoder, dec_act;
    int32_t loss;                      /* NBCI_LOSS_* (mlm) */
    int32_t dtype;
    int32_t fp8_qkv;                   /* 1: the q / k / v projections of the FORWARD run on the block-scaled fp8 matrix instruction
                                          (MX e4m3, see nbci_gemm_fp8); needs dtype = NBCI_BF16 and d_model % 128 == 0 */
    int32_t residual_dtype;            /* storage of the residual stream (the saved BatchNorm inputs) and of its gradient stream between kernels:
                                          NBCI_F32 (default), or NBCI_BF16 (dtype bf16 only; f32 arithmetic, one rounding per store; the
                                          data-gradient GEMMs then hand BatchNorm's backward a bf16 gradient as well). hidden_out stays f32. */
} nbci_ptst_config;

typedef struct nbci_ptst_io {
    int32_t B, S;                       /* batch, padded target length (ctc; 0 otherwise) */
    const float* spikes;                /* (B,T,C) f32, T = context_length */
    const int64_t* spikes_mask;         /* (B,T) */
    const int64_t* spikes_lengths;      /* (B) (ctc) */
    const int64_t* targets;             /* (B,S) (ctc) */
    const int64_t* targets_lengths;     /* (B) */
    const uint8_t* ext_mask;            /* (B,C,P) 0/1: replaces the random patch mask (tests / replay), or NULL */
    int32_t train, want_grad;
    uint32_t seed;
    float grad_scale;
    float* aux;                         /* see above; running statistics updated when train */
    int64_t* nbt;
    float* preds;                       /* out: ctc (B,P,V) log-probs; mlm (B,C,P,patch_length) */
    float* patch_input;                 /* out (B,C,P,patch_length): the un-masked patches (PatchTSTModelOutput.patch_input) or NULL */
    uint8_t* mask_out;                  /* out (B,C,P): mlm loss mask (model mask & unpadded patches) or NULL */
    float* loss;                        /* out: ctc (B) per-sample losses; mlm (1) */
    int64_t* n_examples;                /* out (1) (mlm: number of masked unpadded patches) */
    int32_t* argmax;                    /* out (B,P) greedy path (ctc) or NULL */
    void* hidden_out;                   /* out (B,C,P,D) last_hidden_state in f32, or NULL */
    void* workspace;
    int64_t workspace_bytes;
} nbci_ptst_io;

typedef void* nbci_ptst_plan;
int nbci_ptst_plan_create(const nbci_ptst_config* cfg, nbci_ptst_plan* out);
void nbci_ptst_plan_destroy(nbci_ptst_plan plan);
int64_t nbci_ptst_param_count(nbci_ptst_plan plan);
int32_t nbci_ptst_num_params(nbci_ptst_plan plan);
int32_t nbci_ptst_num_segments(nbci_ptst_plan plan);
int nbci_ptst_param_info(nbci_ptst_plan plan, int32_t index, char* name, int32_t name_cap, int64_t* offset, int64_t* numel,
                         int32_t* rows, int32_t* cols, int32_t* segment);
int nbci_ptst_segment_range(nbci_ptst_plan plan, int32_t seg, int64_t* begin, int64_t* end);
int32_t nbci_ptst_num_patches(nbci_ptst_plan plan);
int64_t nbci_ptst_aux_floats(nbci_ptst_plan plan);
int64_t nbci_ptst_workspace_bytes(nbci_ptst_plan plan, int32_t B, int32_t S);
int nbci_ptst_forward(nbci_ptst_plan plan, const float* params, const void* params_lp, const nbci_ptst_io* io, nbci_stream_t stream);
int nbci_ptst_backward(nbci_ptst_plan plan, const float* params, const void* params_lp, const nbci_ptst_io* io, float* grads,
                       int32_t seg_hi, int32_t seg_lo, nbci_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NBCI_H */
