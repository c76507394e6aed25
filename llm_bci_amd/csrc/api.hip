// api.hip — extern "C" surface of libnbci.so (see include/nbci.h).
#include "kernels.h"

namespace nbci {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
int fail(int code, const std::string& msg) { g_last_error = msg; return code; }
}  // namespace nbci

#include <map>
#include <mutex>
namespace nbci {
int ensure_dyn_lds(const void* kernel, int bytes, const char* what) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(NBCI_EHIP, std::string(what) + ": hipGetDevice: " + hipGetErrorString(e));
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, int> granted;
    std::lock_guard<std::mutex> l(mu);
    int& g = granted[{dev, kernel}];
    if (g >= bytes) return NBCI_OK;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return fail(NBCI_EHIP, std::string(what) + ": LDS attribute: " + hipGetErrorString(e));
    g = bytes;
    return NBCI_OK;
}
}  // namespace nbci

extern "C" {

int nbci_version(void) { return NBCI_VERSION; }
const char* nbci_last_error(void) { return nbci::g_last_error.c_str(); }

int nbci_gemm(const nbci_gemm_desc* d, nbci_stream_t stream) {
    if (!d) return nbci::fail(NBCI_EINVAL, "nbci_gemm: null desc");
    return nbci::gemm_launch_timed(*d, (hipStream_t)stream);
}

int nbci_gemm_grouped(const nbci_gemm_desc* descs, int32_t n, nbci_stream_t stream) {
    return nbci::gemm_grouped_launch_timed(descs, n, (hipStream_t)stream);
}
int nbci_smooth_noise(const float* spikes, void* out, int32_t out_dtype, int32_t B, int32_t T, int32_t N, const float* taps,
                      int32_t ntaps, float white_sd, float offset_sd, uint32_t seed, nbci_stream_t stream) {
    return nbci::smooth_noise_launch(spikes, out, out_dtype, B, T, N, taps, ntaps, white_sd, offset_sd, seed, (hipStream_t)stream);
}
int nbci_layernorm_fwd(const float* x, const float* w, const float* b, void* y, int32_t y_dtype, float* mean, float* rstd,
                       int32_t M, int32_t H, nbci_stream_t stream) {
    return nbci::layernorm_fwd_launch(x, w, b, y, y_dtype, mean, rstd, M, H, (hipStream_t)stream);
}
int nbci_layernorm_bwd(const float* dy, const float* x, const float* w, const float* mean, const float* rstd, float* dx,
                       float* dw, float* db, int32_t M, int32_t H, int32_t accumulate_dx, nbci_stream_t stream) {
    return nbci::layernorm_bwd_launch(dy, x, w, mean, rstd, dx, dw, db, M, H, accumulate_dx, (hipStream_t)stream);
}
int nbci_layernorm_fwd_ex(const void* x, int32_t x_dtype, const float* w, const float* b, void* y, int32_t y_dtype, float* mean, float* rstd,
                          int32_t M, int32_t H, nbci_stream_t stream) {
    return nbci::layernorm_fwd_launch(x, x_dtype, w, b, y, y_dtype, mean, rstd, M, H, (hipStream_t)stream);
}
int nbci_layernorm_bwd_ex(const void* dy, int32_t dy_dtype, const void* x, int32_t x_dtype, const float* w, const float* mean, const float* rstd,
                          const void* dx_in, void* dx_out, float* dw, float* db, int32_t M, int32_t H, void* cast_out, int32_t cast_dtype,
                          float drop_p, uint32_t seed, uint32_t site, float* cast_colsum, nbci_stream_t stream) {
    if (!cast_out && drop_p > 0.f && !cast_colsum) return nbci::fail(NBCI_EINVAL, "layernorm_bwd_ex: dropout without a cast output or its column sums");
    const nbci::LnCast cz{cast_out, cast_dtype == NBCI_BF16 ? 1 : 0, nbci::drop_threshold(drop_p), drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f,
                          nbci::drop_key(seed, site), cast_colsum, 0, 0, 0, 0};
    return nbci::layernorm_bwd_launch(dy, dy_dtype == NBCI_BF16 ? 1 : 0, x, w, mean, rstd,
                                      nbci::LnStreams{x_dtype == NBCI_BF16 ? 1 : 0, dx_in, dx_out, x_dtype == NBCI_BF16 ? 1 : 0}, dw, db, M, H,
                                      (hipStream_t)stream, nbci::RepCfg{0, 1}, cz);
}
int nbci_softmax_fwd(const float* S, void* P, void* Pd, int32_t p_dtype, const int32_t* token_mask, int32_t B, int32_t n_heads,
                     int32_t Tp, int32_t ldS, int32_t ldP, int32_t ctx_forward, int32_t ctx_backward, float drop_p,
                     uint32_t seed, uint32_t site, nbci_stream_t stream) {
    return nbci::softmax_fwd_launch(S, P, Pd, p_dtype, token_mask, B, n_heads, Tp, ldS, ldP, ctx_forward, ctx_backward, drop_p,
                                    seed, site, (hipStream_t)stream);
}
int nbci_softmax_bwd(const float* dPd, const void* P, void* dS, int32_t p_dtype, int32_t B, int32_t n_heads, int32_t Tp,
                     int32_t ldS, int32_t ldP, float drop_p, uint32_t seed, uint32_t site, nbci_stream_t stream) {
    return nbci::softmax_bwd_launch(dPd, P, dS, p_dtype, B, n_heads, Tp, ldS, ldP, drop_p, seed, site, (hipStream_t)stream);
}
int nbci_logsoftmax(const float* logits, int32_t ldl, float* preds, int32_t* argmax, int32_t M, int32_t V, nbci_stream_t stream) {
    return nbci::logsoftmax_launch(logits, ldl, preds, argmax, M, V, (hipStream_t)stream);
}
int64_t nbci_ctc_workspace_floats(int32_t B, int32_t Tp, int32_t S) { return (int64_t)nbci::ctc_alpha_floats(B, Tp, S); }
int nbci_ctc(const float* preds, const int64_t* targets, const int32_t* in_lens, const int64_t* tgt_lens, int32_t B, int32_t Tp,
             int32_t V, int32_t S, int32_t blank, int32_t zero_infinity, float* loss, float* alpha_ws, void* dlogits,
             int32_t d_dtype, int32_t ldd, float grad_scale, nbci_stream_t stream) {
    return nbci::ctc_launch(preds, targets, in_lens, tgt_lens, B, Tp, V, S, blank, zero_infinity, loss, alpha_ws, dlogits, d_dtype,
                            ldd, grad_scale, (hipStream_t)stream);
}
int nbci_step_stats(double* stats, const float* loss, int32_t B, double n_examples, const int32_t* errors, nbci_stream_t stream) {
    return nbci::step_stats_launch(stats, loss, B, n_examples, errors, (hipStream_t)stream);
}
int nbci_per(const int32_t* argmax, const int64_t* targets, const int64_t* tgt_lens, int32_t B, int32_t Tp, int32_t S, int32_t blank,
             int32_t* decoded, int32_t* dec_lens, int32_t* errors, int32_t* scratch, nbci_stream_t stream) {
    return nbci::per_launch(argmax, targets, tgt_lens, B, Tp, S, blank, decoded, dec_lens, errors, scratch, (hipStream_t)stream);
}
int nbci_adamw(float* p, const float* g, float* m, float* v, void* p_lp, int64_t n, float lr, float beta1, float beta2, float eps,
               float weight_decay, float bc1, float bc2, float grad_scale, nbci_stream_t stream) {
    nbci::ProfScope ps("adamw_kernel<false, float>", 0.0, (double)n * (p_lp ? 30.0 : 28.0), (hipStream_t)stream);
    return nbci::adamw_launch(p, const_cast<float*>(g), m, v, p_lp, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale, (hipStream_t)stream);
}
int nbci_adamw_lp(float* p, const void* g_bf16, float* m, float* v, void* p_lp, int64_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, float bc1, float bc2, float grad_scale, nbci_stream_t stream) {
    nbci::ProfScope ps("adamw_kernel<false, __bf16>", 0.0, (double)n * (p_lp ? 28.0 : 26.0), (hipStream_t)stream);
    return nbci::adamw_launch(p, const_cast<void*>(g_bf16), m, v, p_lp, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale, (hipStream_t)stream, false, 0, true);
}
int nbci_adamw_zero(float* p, float* g, float* m, float* v, void* p_lp, int64_t n, float lr, float beta1, float beta2, float eps,
                    float weight_decay, float bc1, float bc2, float grad_scale, int32_t max_blocks, nbci_stream_t stream) {
    nbci::ProfScope ps("adamw_kernel<true, float>", 0.0, (double)n * (p_lp ? 34.0 : 32.0), (hipStream_t)stream);
    return nbci::adamw_launch(p, g, m, v, p_lp, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale, (hipStream_t)stream, true, max_blocks);
}
int nbci_cast(const float* in, void* out, int32_t out_dtype, int64_t n, nbci_stream_t stream) {
    return nbci::cast_launch(in, out, out_dtype, n, (hipStream_t)stream);
}

int nbci_attention_fwd(const void* qkv, const int32_t* token_mask, void* out, float* lse, int32_t B, int32_t n_heads, int32_t Tp, int32_t H,
                       int32_t ctx_forward, int32_t ctx_backward, float drop_p, uint32_t seed, uint32_t site_prob, uint32_t site_out,
                       nbci_stream_t stream) {
    return nbci::attn_fwd_launch(qkv, token_mask, out, lse, B, n_heads, Tp, H, ctx_forward, ctx_backward, drop_p, seed, site_prob, site_out,
                                 (hipStream_t)stream);
}
int nbci_attention_bwd(const void* qkv, const int32_t* token_mask, const void* out, const float* lse, const void* d_out, void* dS_ws,
                       void* Pd_ws, int32_t ldP, void* dqkv, float* bias_grad, int32_t B, int32_t n_heads, int32_t Tp, int32_t H,
                       int32_t ctx_forward, int32_t ctx_backward, float drop_p, uint32_t seed, uint32_t site_prob, nbci_stream_t stream) {
    return nbci::attn_bwd_launch(qkv, token_mask, out, lse, d_out, dS_ws, Pd_ws, ldP, dqkv, bias_grad, B, n_heads, Tp, H, ctx_forward,
                                 ctx_backward, drop_p, seed, site_prob, (hipStream_t)stream, nbci::RepCfg{0, 1});
}
int nbci_coupler_splice_fwd(const void* text, const void* spikes, void* out, int32_t dtype, const int64_t* text_mask,
                            const int64_t* spikes_valid, int64_t* mask_out, const int64_t* targets, int64_t* targets_out,
                            const int64_t* split, int32_t B, int32_t Lt, int32_t Ts, int32_t H, nbci_stream_t stream) {
    return nbci::splice_fwd_launch(text, spikes, out, dtype, text_mask, spikes_valid, mask_out, targets, targets_out, split, B, Lt, Ts, H,
                                   (hipStream_t)stream);
}
int nbci_coupler_splice_bwd(const void* d_out, void* d_text, void* d_spikes, int32_t dtype, const int64_t* split, int32_t B, int32_t Lt,
                            int32_t Ts, int32_t H, nbci_stream_t stream) {
    return nbci::splice_bwd_launch(d_out, d_text, d_spikes, dtype, split, B, Lt, Ts, H, (hipStream_t)stream);
}
int nbci_colsum(const void* in, int32_t in_dtype, int64_t ld, int32_t M, int32_t N, float* out, nbci_stream_t stream) {
    if (!in || !out || M <= 0 || N <= 0 || ld < N) return nbci::fail(NBCI_EINVAL, "colsum: bad arguments");
    return nbci::colsum_launch(in, in_dtype, ld, M, N, out, (hipStream_t)stream);
}
int nbci_mx_quantize(const void* x, int32_t dtype, int64_t ldx, void* codes, void* scales, int64_t rows, int32_t K, nbci_stream_t stream) {
    return nbci::mx_quantize_launch(x, dtype, ldx, codes, scales, rows, K, (hipStream_t)stream);
}
int nbci_gemm_fp8(const void* A8, const void* sA, const void* W8, const void* sW, const float* bias, void* C, int32_t c_dtype, int64_t M, int32_t N,
                  int32_t K, int64_t ldc, nbci_stream_t stream) {
    return nbci::gemm_fp8_launch(A8, sA, W8, sW, bias, C, c_dtype, M, N, K, ldc, (hipStream_t)stream);
}
int nbci_debug_gemm_pc(int32_t mode) { nbci::gemm_pc_set_mode(mode); return NBCI_OK; }
int nbci_debug_gemm_streamk(int32_t mode) {
    if (mode < 0 || mode > 4) return nbci::fail(NBCI_EINVAL, "stream-K mode must be 0 .. 4");
    nbci::gemm_streamk_set_mode(mode);
    return NBCI_OK;
}
int nbci_stream_order(nbci_stream_t before, nbci_stream_t after) {
    static thread_local hipEvent_t ev[16] = {};
    int dev = 0;
    NBCI_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) return nbci::fail(NBCI_EINVAL, "stream_order: device index out of range");
    if (!ev[dev]) NBCI_CHECK_HIP(hipEventCreateWithFlags(&ev[dev], hipEventDisableTiming | hipEventDisableSystemFence));
    NBCI_CHECK_HIP(hipEventRecord(ev[dev], (hipStream_t)before));
    NBCI_CHECK_HIP(hipStreamWaitEvent((hipStream_t)after, ev[dev], 0));   // (the wait captures this record; the event may be re-recorded at once)
    return NBCI_OK;
}
int nbci_release_scratch(void) {
    const int rc = nbci::gemm_streamk_release();
    const int rc2 = nbci::fattn_release();
    return rc != NBCI_OK ? rc : rc2;
}
int nbci_streamk_timeouts(int64_t* out) {
    if (!out) return nbci::fail(NBCI_EINVAL, "streamk_timeouts: null output");
    long long n = 0;
    const int rc = nbci::gemm_streamk_timeouts(&n);
    *out = (int64_t)n;
    return rc;
}
int nbci_debug_gemm_grouped_plan(const nbci_gemm_desc* descs, int32_t n, int32_t* out8) { return nbci::gemm_grouped_describe(descs, n, out8); }
int nbci_profile_enable(int32_t on) { nbci::gemm_profile_enable(on != 0); return NBCI_OK; }
int nbci_profile_collect_text(char* buf, int64_t cap) {
    if (!buf || cap < 1) return nbci::fail(NBCI_EINVAL, "profile_collect_text: null output");
    return nbci::prof_collect_text(buf, cap);
}
int nbci_profile_collect(double* out24) {
    if (!out24) return nbci::fail(NBCI_EINVAL, "profile_collect: null output");
    return nbci::gemm_profile_collect(out24);
}


int nbci_attention_small_fwd(const void* qkv, void* out, float* lse, int32_t dtype, int32_t NS, int32_t n_heads, int32_t S, int32_t H,
                             float drop_p, uint32_t seed, uint32_t site, nbci_stream_t stream) {
    return nbci::sattn_fwd_launch(qkv, out, lse, dtype, NS, n_heads, S, H, drop_p, seed, site, (hipStream_t)stream);
}
int nbci_attention_small_bwd(const void* qkv, const void* out, const void* d_out, const float* lse, float* dsum, void* dqkv, int32_t dtype,
                             int32_t NS, int32_t n_heads, int32_t S, int32_t H, float drop_p, uint32_t seed, uint32_t site,
                             nbci_stream_t stream) {
    return nbci::sattn_bwd_launch(qkv, out, d_out, lse, dsum, dqkv, dtype, NS, n_heads, S, H, drop_p, seed, site, (hipStream_t)stream);
}
int nbci_attention_flash_fwd(const void* qkv, void* out, float* lse, int32_t NS, int32_t n_heads, int32_t S, int32_t H, float drop_p,
                             uint32_t seed, uint32_t site, nbci_stream_t stream) {
    return nbci::fattn_fwd_launch(qkv, out, lse, NS, n_heads, S, H, drop_p, seed, site, (hipStream_t)stream);
}
int nbci_attention_flash_bwd(const void* qkv, const void* out, const void* d_out, const float* lse, float* dsum, void* dqkv, int32_t NS,
                             int32_t n_heads, int32_t S, int32_t H, float drop_p, uint32_t seed, uint32_t site, nbci_stream_t stream) {
    return nbci::fattn_bwd_launch(qkv, out, d_out, lse, dsum, dqkv, NS, n_heads, S, H, drop_p, seed, site, (hipStream_t)stream);
}

}  // extern "C"
