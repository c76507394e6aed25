// kernels.h — internal launch API of the non-GEMM kernels (all HBM-bound, wave64, gfx950).
// "act dtype" = dtype of activations that feed GEMMs: NBCI_F32 (parity path) or NBCI_BF16.
#pragma once
#include "nbci_common.h"
#include "../../include/nbci.h"

namespace nbci {

int gemm_launch(const nbci_gemm_desc& d, hipStream_t stream);
int gemm_launch_timed(const nbci_gemm_desc& d, hipStream_t stream);  // = gemm_launch unless profiling is on
int gemm_grouped_launch(const nbci_gemm_desc* descs, int n, hipStream_t stream);        // <= 6 problems, one launch
int gemm_grouped_launch_timed(const nbci_gemm_desc* descs, int n, hipStream_t stream);
int gemm_grouped_describe(const nbci_gemm_desc* descs, int n, int32_t* out8);   // host-only: the launch scheme gemm_grouped_launch would pick
void gemm_profile_enable(bool on);
void gemm_pc_set_mode(int m);
void gemm_streamk_set_mode(int m);   // gemm_streamk.hip: 0 off, 1 where the tile count leaves slots idle, 2 always (nbci_debug_gemm_streamk)
int gemm_streamk_release();            // frees the per-stream partial-tile scratch (nbci_release_scratch)
int gemm_streamk_timeouts(long long* out);
void set_available_cus(int cus);   // gemm_glds.hip: CUs the tile cost model may count on (nbci_set_available_cus)
int available_cus();   // gemm_pc.hip: kernel-family switch (nbci_debug_gemm_pc)
bool gemm_profile_on();
int gemm_profile_collect(double* out24);

// models/ndt1.py:92-107 — depthwise gaussian smoothing along T ('same', zero padded) + train noise
int smooth_noise_launch(const float* spikes, void* out, int out_dtype, int B, int T, int N, const float* taps,
                        int ntaps, float white_sd, float offset_sd, uint32_t seed, hipStream_t s);

// models/ndt1.py:181-183,207-208 — token mask (prod over window), first-T' timestamps, stacked lens
int token_prep_launch(const int64_t* mask, const int64_t* ts, const int64_t* lens, int B, int T, int Tp,
                      int size, int stride, int32_t* tmask, int64_t* tts, int32_t* tlens, hipStream_t s, int npre = 0);
// learned prefix tokens (day / block, ndt1.py:192-203): assemble [prefix rows | spike tokens] + embedder dropout; table gradients
int prefix_assemble_launch(const float* xtok, const float* tab0, const int64_t* idx0, const float* tab1, const int64_t* idx1, void* x,
                           int B, int Tp, int npre, int H, float drop_p, uint32_t seed, uint32_t site, hipStream_t s, int x_dtype = NBCI_F32);
int prefix_grad_launch(const void* dx, const int64_t* idx, float* dtab, int B, int Tt, int k, int H, float drop_p, uint32_t seed,
                       uint32_t site, hipStream_t s, int dx_dtype = NBCI_F32);

// nn.LayerNorm (eps 1e-5, affine) forward: x f32 (M,H) -> y act dtype, saves mean/rstd. First form: x f32, or bf16 (a bf16 residual
// stream: statistics and the normalisation in f32 from the widened row; y bf16)
int layernorm_fwd_launch(const void* x, int x_dtype, const float* w, const float* b, void* y, int y_dtype, float* mean,
                         float* rstd, int M, int H, hipStream_t s, float* y32 = nullptr);
int layernorm_fwd_launch(const float* x, const float* w, const float* b, void* y, int y_dtype, float* mean,
                         float* rstd, int M, int H, hipStream_t s, float* y32 = nullptr);  // y32: optional f32 copy of y
// backward: dx (f32, M,H) += LN'(dy); dw += sum dy*xhat; db += sum dy
// dy: f32, or (dy_bf16) bf16 as a bf16 data-gradient GEMM wrote it
int layernorm_bwd_launch(const void* dy, const float* x, const float* w, const float* mean, const float* rstd,
                         float* dx, float* dw, float* db, int M, int H, int accumulate_dx, hipStream_t s,
                         RepCfg rc = RepCfg{0, 1}, LnCast cz = LnCast{nullptr, 0, 0u, 1.f, 0u, nullptr}, int dy_bf16 = 0);
// general form: the saved input x and the gradient stream in f32 or (both) bf16; dx_out = dx_in (NULL: nothing) + LN'(dy), two buffers
// or one. All arithmetic in f32; a bf16 stream is rounded once, at the store.
struct LnStreams { int x_bf16; const void* dx_in; void* dx_out; int dx_bf16; };
int layernorm_bwd_launch(const void* dy, int dy_bf16, const void* x, const float* w, const float* mean, const float* rstd, LnStreams st,
                         float* dw, float* db, int M, int H, hipStream_t s, RepCfg rc = RepCfg{0, 1},
                         LnCast cz = LnCast{nullptr, 0, 0u, 1.f, 0u, nullptr});

// masked softmax over scores (B,nh,Tp,ldS f32): mask = eye | (ctx & key_valid) (ndt1.py:435-437; tmask NULL = all valid),
// writes P (pre-dropout) and Pd (post attention-prob dropout, ndt1.py:289) in act dtype, ld = ldP
int softmax_fwd_launch(const float* S, void* P, void* Pd, int p_dtype, const int32_t* tmask, int B, int nh, int Tp,
                       int ldS, int ldP, int ctx_fwd, int ctx_bwd, float drop_p, uint32_t seed, uint32_t site,
                       hipStream_t s);
// dS = P * (dP - sum(dP*P)) with dP = dPd * keepmask; dPd f32 (ld = ldS), dS act dtype (ld = ldP)
int softmax_bwd_launch(const float* dPd, const void* P, void* dS, int p_dtype, int B, int nh, int Tp, int ldS,
                       int ldP, float drop_p, uint32_t seed, uint32_t site, hipStream_t s);

// out(act dtype) = in(f32) * keepmask(site)   (p = 0: plain cast)
int dropcast_launch(const float* in, void* out, int out_dtype, int64_t n, float drop_p, uint32_t seed,
                    uint32_t site, hipStream_t s);
// 2-D form with the bias-gradient column sum fused in: colsum[n] += sum_m out[m][n] (may be NULL)
int dropcast2d_launch(const void* in, void* out, int out_dtype, int M, int N, float drop_p, uint32_t seed,
                      uint32_t site, float* colsum, hipStream_t s, RepCfg rc = RepCfg{0, 1}, int in_dtype = NBCI_F32);   // in: f32, or a bf16 stream
// out[n] += sum_m in[m][n]  (bias gradients)
int colsum_launch(const void* in, int in_dtype, int64_t ld, int M, int N, float* out, hipStream_t s,
                  RepCfg rc = RepCfg{0, 1});
// grads[flat_of[ci]] += sum_r rep[r*stride + ci] for ci in [cbegin, cend); replicas are zeroed
int fold_replicas_launch(float* rep, long long stride, int nrep, const int* flat_of, int cbegin, int cend, float* grads,
                         hipStream_t s);

// stack backward: dy[b,t,c] = sum_j dwin[b,j,(t - stride*j)*D + c]; dpre = dy * act'(y) -> act dtype
int col2im_actgrad_launch(const void* dwin, const void* y, void* dpre, int dtype, int B, int T, int Tp, int D,
                          int size, int stride, int act, hipStream_t s);
// dpos[tts[row]][:] += dx[row][:] * keepmask  (nn.Embedding backward + embed dropout)
int posgrad_launch(const void* dx, const int64_t* tts, float* dpos, int M, int H, float drop_p, uint32_t seed,
                   uint32_t site, hipStream_t s, int Tp = 0, int npre = 0, int dx_dtype = NBCI_F32);

// RoPE on the q and k thirds of a packed (M, 3H) qkv buffer, in place (ndt1.py:62-71); inverse = backward
int rope_launch(void* qkv, int dtype, const int64_t* tts, const float* cos_t, const float* sin_t, int M, int H,
                int nh, int inverse, hipStream_t s);

// log-softmax over V of logits (M, ldl) f32 -> preds (M, V) f32 contiguous, argmax path int32 (M)
int logsoftmax_launch(const float* logits, int ldl, float* preds, int32_t* argmax, int M, int V, hipStream_t s);
// CTC (torch.nn.CTCLoss(reduction="none", blank, zero_infinity), ndt1.py:517,581)
// alpha_ws: f32 workspace B * Tp * (2*S+1). dlogits (B*Tp, ldd) act dtype = grad_scale * (softmax - occupancy)
int ctc_launch(const float* preds, const int64_t* targets, const int32_t* in_lens, const int64_t* tgt_lens,
               int B, int Tp, int V, int S, int blank, int zero_infinity, float* loss, float* alpha_ws,
               void* dlogits, int d_dtype, int ldd, float grad_scale, hipStream_t s);
size_t ctc_alpha_floats(int B, int Tp, int S);

// greedy decode (reference's format_ctc, utils/eval_bci.py:41-48) + Levenshtein vs targets (main.py:68-74)
int adapt_gather_launch(const void* wsrc, long long day_stride, const int64_t* day, void* wsel, int64_t* rows, int dtype, int B, int wn,
                        int T, int ndays, hipStream_t s);
int adapt_grads_launch(const void* dpre, int dtype, const float* wpart, float* bpart, const int64_t* day, float* gw, float* gb,
                       long long day_stride, int B, int T, int D, int wn, int ndays, hipStream_t s);
int gate_cast_launch(const float* src, const void* gate, void* out, int dtype, long long n, hipStream_t s);
int step_stats_launch(double* stats, const float* loss, int B, double n_examples, const int32_t* err, hipStream_t s);
int per_launch(const int32_t* argmax, const int64_t* targets, const int64_t* tgt_lens, int B, int Tp, int S,
               int blank, int32_t* decoded, int32_t* dec_lens, int32_t* errors, int32_t* scratch, hipStream_t s);

// fused AdamW over a flat buffer (torch.optim.AdamW semantics; models/trainer.py:229,340)
int adamw_launch(float* p, void* g, float* m, float* v, void* p_lp, int64_t n, float lr, float beta1,
                 float beta2, float eps, float wd, float bc1, float bc2, float grad_scale, hipStream_t s, bool zero_grad = false, int max_blocks = 0,
                 bool g_bf16 = false);

int cast_launch(const float* in, void* out, int out_dtype, int64_t n, hipStream_t s);

// BCI coupler splice (models/bci.py:143-166)
int splice_fwd_launch(const void* text, const void* spikes, void* out, int dtype, const int64_t* tmask, const int64_t* svalid,
                      int64_t* mask_out, const int64_t* targets, int64_t* targets_out, const int64_t* split, int B, int Lt, int Ts,
                      int H, hipStream_t s);
int splice_bwd_launch(const void* dout, void* dtext, void* dspikes, int dtype, const int64_t* split, int B, int Lt, int Ts, int H,
                      hipStream_t s);

// ---- iTransformer SSL path (itr_kernels.hip) ----
int masker_launch(const nbci_masker_desc& d, hipStream_t s);
int btn_to_bnt_launch(const float* in, float* out, int B, int T, int N, hipStream_t s);
int itr_assemble_fwd_launch(const float* t2, const float* w, const float* b, const float* tab1, const int64_t* idx1, const float* tab2,
                            const int64_t* idx2, const float* cls, float* x32, void* xb, int xb_dtype, float* mean, float* rstd, int B,
                            int N, int H, int use_cls, float drop_p, uint32_t seed, uint32_t site, hipStream_t s,
                            const float* extra = nullptr);   // extra: optional f32 (B*N, H) added to the channel tokens (the depth embedding)
// Linear(1 -> W) + ReLU of a scalar per row and its weight / bias gradients (UnivariateTransformer embed_spikes.0, depth_embeddings.0)
int scalar_lin_fwd_launch(const float* x, const float* w, const float* b, void* out, int out_dtype, long long rows, int W, int period, hipStream_t s);
int scalar_lin_bwd_launch(const void* du, int du_dtype, const float* x, float* dw, float* db, RepCfg rc, long long rows, int W, int period, hipStream_t s);
// UnivariateTransformer (itransformer.py:75-93): input assembly [cls | tokens + embed_pos] and its backward (token-gradient operand copy + cls
// gradient; position-table gradient)
int uni_finish_fwd_launch(const float* t_in, const float* pos, const int64_t* ts, const float* cls, float* y32, void* xb, int xb_dtype, int B, int N,
                          int T, int h, hipStream_t s);
int uni_split_bwd_launch(const void* dseq, int stream_dtype, void* dte, int dte_dtype, float* dcls, RepCfg rc, long long rows, int T, int h, hipStream_t s);
int uni_posgrad_launch(const void* dseq, int stream_dtype, const int64_t* ts, float* dpos, int B, int N, int T, int h, hipStream_t s);
int itr_assemble_bwd_launch(const void* dx0, float* dtok, float* dtab1, const int64_t* idx1, float* dtab2, const int64_t* idx2,
                            float* dcls, RepCfg rc, int B, int N, int H, int use_cls, float drop_p, uint32_t seed, uint32_t site,
                            hipStream_t s, int dx_dtype = NBCI_F32);
int itr_mlm_loss_launch(const float* pred, int ldp, const float* targets, const int64_t* mask, const int64_t* smask, float* preds_out,
                        int64_t* mask_out, void* dpred, int d_dtype, float* loss, int64_t* n_examples, int B, int T, int N, int use_cls,
                        int kind, float grad_scale, hipStream_t s);

// ---- PatchTST path (ptst_kernels.hip) ----
int ptst_mask_launch(uint8_t* mask, int B, int C, int P, double ratio, int channel_consistent, uint32_t seed, uint32_t site, hipStream_t s);
int ptst_patchify_launch(const float* x, float* patch, float* xm, const uint8_t* mask, int B, int T, int C, int P, int pl, int stride,
                         int start, float mask_value, hipStream_t s);
int ptst_embed_launch(const float* xm, const float* W, const float* bias, const float* pos, void* h, long long M, int P, int pl, int D,
                      float drop_p, uint32_t seed, uint32_t site, hipStream_t s, int h_dtype = NBCI_F32);
size_t bn_partial_floats(long long M, int D);
// x (and in the backward the gradient stream dx): f32, or bf16 (nbci_ptst_config.residual_dtype); dy: f32, or bf16 as a bf16 GEMM writes it
int batchnorm_fwd_launch(const void* x, const float* w, const float* b, float* run_mean, float* run_var, int train, float eps, void* y,
                         int y_dtype, float* mean, float* rstd, float* partials, long long M, int D, hipStream_t s, void* q8 = nullptr, void* q8_scales = nullptr,
                         int x_dtype = NBCI_F32);
int batchnorm_bwd_launch(const void* dy, const void* x, const float* mean, const float* rstd, const float* w, void* dx, float* dw, float* db,
                         float* partials, float* sums /* 3 D floats: the second pass's per-column coefficients */, long long M, int D, int train, hipStream_t s,
                         int dy_dtype = NBCI_F32, int stream_dtype = NBCI_F32);
// dW[D][pl] += de^T xm over all M rows (f32; the shared patch embedding's weight gradient as a streaming reduction)
int ptst_embed_wgrad_launch(const float* de, const float* xm, float* dW, long long M, int pl, int D, hipStream_t s);
size_t ptst_pool_partial_floats(int B, int P, int D);   // f32 scratch of the pooling's channel-chunk partial sums
int ptst_pool_fwd_launch(const void* h, void* pooled, int dtype, int B, int C, int P, int D, hipStream_t s, int h_dtype, float* partial);
int ptst_pool_bwd_launch(const float* dpooled, void* dh, int B, int C, int P, int D, hipStream_t s, int dh_dtype = NBCI_F32);
int ptst_lens_launch(const int64_t* lens, int32_t* out, int B, int pl, int stride, hipStream_t s);
int ptst_mlm_loss_launch(const float* pred, int ldp, const float* target, const uint8_t* mask, const int64_t* smask, float* preds_out,
                         uint8_t* mask_out, void* dpred, int d_dtype, float* loss, int64_t* n_examples, int B, int T, int C, int P, int pl,
                         int stride, int kind, float grad_scale, hipStream_t s);

// small-head attention without a score tensor (attn_small.hip): head 16 / 32 / 64, any length, no mask
bool sattn_eligible(int dtype, int S, int H, int nh);
size_t sattn_stat_floats(int NS, int nh, int S);
int sattn_fwd_launch(const void* qkv, void* out, float* L, int dtype, int NS, int nh, int S, int H, float drop_p, uint32_t seed, uint32_t site,
                     hipStream_t s);
int sattn_bwd_launch(const void* qkv, const void* out, const void* dout, const float* L, float* Dsum, void* dqkv, int dtype, int NS, int nh, int S,
                     int H, float drop_p, uint32_t seed, uint32_t site, hipStream_t s);

// streaming MFMA attention (attn_flash.hip): bf16, head 32 / 64 / 96 / 128, any length, no mask
// fp8.hip: MX-scaled e4m3 (OCP) quantisation and the block-scaled-MFMA projection GEMM
int mx_quantize_launch(const void* x, int dtype, long long ldx, void* q, void* scales, long long rows, int K, hipStream_t s);
int gemm_fp8_launch(const void* A8, const void* sA, const void* W8, const void* sW, const float* bias, void* C, int c_dtype, long long M, int N, int K,
                    long long ldc, hipStream_t s);
bool fattn_eligible(int dtype, int S, int H, int nh);
int fattn_release();                   // frees the per-stream keep-bit scratch of the streaming attention backward (nbci_release_scratch)
// NDT1's masked attention (key validity + context span + self, ndt1.py:30-41,435-437) on the streaming kernels, any length;
// the output dropout of ndt1.py:292 is fused into the forward's store (site_out)
int fattn_masked_fwd_launch(const void* qkv, const int32_t* tmask, void* out, float* L, int NS, int nh, int S, int H, int cf, int cb,
                            float drop_p, uint32_t seed, uint32_t site_prob, uint32_t site_out, hipStream_t s);
int fattn_masked_bwd_launch(const void* qkv, const int32_t* tmask, const void* out, const void* dout, const float* L, float* Dsum, void* dqkv,
                            int NS, int nh, int S, int H, int cf, int cb, float drop_p, uint32_t seed, uint32_t site_prob, hipStream_t s);
int fattn_fwd_launch(const void* qkv, void* out, float* L, int NS, int nh, int S, int H, float drop_p, uint32_t seed, uint32_t site, hipStream_t s);
int fattn_bwd_launch(const void* qkv, const void* out, const void* dout, const float* L, float* Dsum, void* dqkv, int NS, int nh, int S, int H,
                     float drop_p, uint32_t seed, uint32_t site, hipStream_t s);

// fused attention (attention.hip): bf16, head 128, T' <= 160
bool attn_fused_eligible(int dtype, int Tp, int H, int nh);
int attn_fwd_launch(const void* qkv, const int32_t* tmask, void* ad, float* lse, int B, int nh, int Tp, int H, int cf, int cb, float drop_p,
                    uint32_t seed, uint32_t site_p, uint32_t site_o, hipStream_t s);
int attn_bwd_launch(const void* qkv, const int32_t* tmask, const void* ad, const float* lse, const void* da, void* dS, void* Pd, int ldP,
                    void* dqkv, float* bias_grad, int B, int nh, int Tp, int H, int cf, int cb, float drop_p, uint32_t seed, uint32_t site_p,
                    hipStream_t s, RepCfg rc);

}  // namespace nbci
