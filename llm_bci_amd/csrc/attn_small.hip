// attn_small.hip — unmasked multi-head attention for SMALL heads (head size 16 / 32 / 64), any sequence length,
// bf16 or f32 operands, f32 math: the PatchTST shape (d_model 256 / 8 heads = 32, 205 patches, 8 k - 65 k
// (batch, channel, head) units per step) — transformers' eager_attention_forward + its autograd under
// models/patchtst.py:176 — and the small-head iTransformer configs (itransformer.py:158-173).
//
// At head size 32 a 205 x 205 score block is a poor fit for MFMA tiles and the unfused path (batched GEMMs with
// K = 32 + an f32 score tensor written and re-read) was ~45 % of the PatchTST step. Here one THREAD owns one query
// (forward, dq) or one key (dk, dv) and streams over the other axis with an online softmax; the streamed rows are
// wave-uniform, so they arrive through scalar loads (s_load_dwordx4) and feed v_fma as SGPR operands: no LDS, no
// score tensor, only the row log-sum-exp (f32, M x heads) is kept for the backward.
//   fwd      : O_i = sum_j dropout(softmax_j(q_i k_j / sqrt(d))) v_j ; L_i = logsumexp_j
//   bwd_q    : D_i = dO_i . O_i ; dq_i = sum_j dS_ij k_j / sqrt(d),  dS_ij = P_ij (dP_ij - D_i), dP_ij = (dO_i . v_j) keep_ij
//   bwd_kv   : dk_j = sum_i dS_ij q_i / sqrt(d) ; dv_j = sum_i Pd_ij dO_i
// Dropout bits = the same counter stream as the unfused softmax kernel (index ((unit*S + i)*S + j)), so both paths
// and the oracle draw identical masks.
#include "kernels.h"

namespace nbci {

static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NBCI_EHIP, std::string(what) + ": " + hipGetErrorString(e));
    return NBCI_OK;
}

struct SAArgs {
    const void* qkv;   // (NS*S, 3H) packed q | k | v
    void* out;         // fwd: (NS*S, H) merged heads
    float* L;          // (NS, nh, S) row log-sum-exp
    float* Dsum;       // (NS, nh, S) dO . O (backward scratch)
    const void* dout;  // bwd in (NS*S, H)
    void* dqkv;        // bwd out (NS*S, 3H)
    int NS, nh, S, H;
    float scale;
    unsigned thr; float dscale; uint32_t key;
};

// row of HD elements starting at element offset `off` (a multiple of 8) -> f32 registers / SGPRs when `off` is uniform
template <int HD, typename T> struct RowLoad;
template <int HD> struct RowLoad<HD, float> {
    static __device__ __forceinline__ void load(const void* base, long long off, float (&o)[HD]) {
        const float4* p = (const float4*)((const float*)base + off);
#pragma unroll
        for (int c = 0; c < HD / 4; ++c) { const float4 v = p[c]; o[4 * c] = v.x; o[4 * c + 1] = v.y; o[4 * c + 2] = v.z; o[4 * c + 3] = v.w; }
    }
    static __device__ __forceinline__ void store(void* base, long long off, const float (&v)[HD]) {
        float4* p = (float4*)((float*)base + off);
#pragma unroll
        for (int c = 0; c < HD / 4; ++c) p[c] = make_float4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
    }
};
template <int HD> struct RowLoad<HD, bf16_t> {
    static __device__ __forceinline__ void load(const void* base, long long off, float (&o)[HD]) {
        const uint4* p = (const uint4*)((const bf16_t*)base + off);
#pragma unroll
        for (int c = 0; c < HD / 8; ++c) {
            const uint4 v = p[c];
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[8 * c + 2 * e] = __uint_as_float(w[e] << 16); o[8 * c + 2 * e + 1] = __uint_as_float(w[e] & 0xFFFF0000u); }
        }
    }
    static __device__ __forceinline__ void store(void* base, long long off, const float (&v)[HD]) {
        bf16_t* p = (bf16_t*)base + off;
#pragma unroll
        for (int c = 0; c < HD / 8; ++c) {
            bf16x8 o = {f2bf(v[8 * c]), f2bf(v[8 * c + 1]), f2bf(v[8 * c + 2]), f2bf(v[8 * c + 3]),
                        f2bf(v[8 * c + 4]), f2bf(v[8 * c + 5]), f2bf(v[8 * c + 6]), f2bf(v[8 * c + 7])};
            *(bf16x8*)(p + 8 * c) = o;
        }
    }
};

template <int HD> __device__ __forceinline__ float dot(const float (&a)[HD], const float (&b)[HD]) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) s += a[d] * b[d];
    return s;
}

constexpr int SA_KC = 8;   // keys per online-softmax group (one rescale of the accumulator per group)

template <int HD, typename T>
__global__ __launch_bounds__(64) void sattn_fwd_kernel(SAArgs a) {
    const int unit = blockIdx.x, s = unit / a.nh, h = unit % a.nh;
    const int i = blockIdx.y * 64 + threadIdx.x;
    const bool live = i < a.S;
    const int iq = live ? i : a.S - 1;
    const long long row0 = (long long)s * a.S;
    float q[HD], o[HD];
    RowLoad<HD, T>::load(a.qkv, (row0 + iq) * 3 * a.H + h * HD, q);
#pragma unroll
    for (int d = 0; d < HD; ++d) { q[d] *= a.scale; o[d] = 0.f; }
    float m = -INFINITY, l = 0.f;
    const unsigned rbase = (unsigned)(((long long)unit * a.S + iq) * a.S);
    for (int j0 = 0; j0 < a.S; j0 += SA_KC) {
        float sc[SA_KC];
        float cm = -INFINITY;
#pragma unroll
        for (int jj = 0; jj < SA_KC; ++jj) {
            const int j = j0 + jj;
            if (j < a.S) {   // uniform
                float k[HD];
                RowLoad<HD, T>::load(a.qkv, (row0 + j) * 3 * a.H + a.H + h * HD, k);
                sc[jj] = dot<HD>(q, k);
                cm = fmaxf(cm, sc[jj]);
            } else {
                sc[jj] = -INFINITY;
            }
        }
        const float mn = fmaxf(m, cm);
        const float corr = __expf(m - mn);
        l *= corr;
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] *= corr;
        m = mn;
#pragma unroll
        for (int jj = 0; jj < SA_KC; ++jj) {
            const int j = j0 + jj;
            if (j < a.S) {
                float p = __expf(sc[jj] - mn);
                l += p;
                if (a.thr) p = drop_keep(a.key, a.thr, rbase + (unsigned)j) ? p * a.dscale : 0.f;
                float v[HD];
                RowLoad<HD, T>::load(a.qkv, (row0 + j) * 3 * a.H + 2 * a.H + h * HD, v);
#pragma unroll
                for (int d = 0; d < HD; ++d) o[d] += p * v[d];
            }
        }
    }
    if (!live) return;
    const float inv = 1.0f / l;
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] *= inv;
    RowLoad<HD, T>::store(a.out, (row0 + i) * a.H + h * HD, o);
    a.L[(long long)unit * a.S + i] = m + __logf(l);
}

template <int HD, typename T>
__global__ __launch_bounds__(64) void sattn_bwd_q_kernel(SAArgs a) {
    const int unit = blockIdx.x, s = unit / a.nh, h = unit % a.nh;
    const int i = blockIdx.y * 64 + threadIdx.x;
    const bool live = i < a.S;
    const int iq = live ? i : a.S - 1;
    const long long row0 = (long long)s * a.S;
    float q[HD], go[HD], dq[HD];
    RowLoad<HD, T>::load(a.qkv, (row0 + iq) * 3 * a.H + h * HD, q);
    RowLoad<HD, T>::load(a.dout, (row0 + iq) * a.H + h * HD, go);
    float D;
    {
        float ov[HD];
        RowLoad<HD, T>::load(a.out, (row0 + iq) * a.H + h * HD, ov);
        D = dot<HD>(go, ov);
    }
    const float Li = a.L[(long long)unit * a.S + iq];
#pragma unroll
    for (int d = 0; d < HD; ++d) { q[d] *= a.scale; dq[d] = 0.f; }
    const unsigned rbase = (unsigned)(((long long)unit * a.S + iq) * a.S);
    for (int j = 0; j < a.S; ++j) {
        float k[HD], v[HD];
        RowLoad<HD, T>::load(a.qkv, (row0 + j) * 3 * a.H + a.H + h * HD, k);
        RowLoad<HD, T>::load(a.qkv, (row0 + j) * 3 * a.H + 2 * a.H + h * HD, v);
        const float p = __expf(dot<HD>(q, k) - Li);
        float dp = dot<HD>(go, v);
        if (a.thr) dp = drop_keep(a.key, a.thr, rbase + (unsigned)j) ? dp * a.dscale : 0.f;
        const float ds = p * (dp - D);
#pragma unroll
        for (int d = 0; d < HD; ++d) dq[d] += ds * k[d];
    }
    if (!live) return;
#pragma unroll
    for (int d = 0; d < HD; ++d) dq[d] *= a.scale;
    RowLoad<HD, T>::store(a.dqkv, (row0 + i) * 3 * a.H + h * HD, dq);
    a.Dsum[(long long)unit * a.S + i] = D;
}

template <int HD, typename T>
__global__ __launch_bounds__(64) void sattn_bwd_kv_kernel(SAArgs a) {
    const int unit = blockIdx.x, s = unit / a.nh, h = unit % a.nh;
    const int j = blockIdx.y * 64 + threadIdx.x;
    const bool live = j < a.S;
    const int jk = live ? j : a.S - 1;
    const long long row0 = (long long)s * a.S;
    float k[HD], v[HD], dk[HD], dv[HD];
    RowLoad<HD, T>::load(a.qkv, (row0 + jk) * 3 * a.H + a.H + h * HD, k);
    RowLoad<HD, T>::load(a.qkv, (row0 + jk) * 3 * a.H + 2 * a.H + h * HD, v);
#pragma unroll
    for (int d = 0; d < HD; ++d) { k[d] *= a.scale; dk[d] = 0.f; dv[d] = 0.f; }
    for (int i = 0; i < a.S; ++i) {
        float q[HD], go[HD];
        RowLoad<HD, T>::load(a.qkv, (row0 + i) * 3 * a.H + h * HD, q);
        RowLoad<HD, T>::load(a.dout, (row0 + i) * a.H + h * HD, go);
        const float Li = a.L[(long long)unit * a.S + i], D = a.Dsum[(long long)unit * a.S + i];
        const float p = __expf(dot<HD>(q, k) - Li);
        float keep = 1.f;
        if (a.thr) keep = drop_keep(a.key, a.thr, (unsigned)(((long long)unit * a.S + i) * a.S) + (unsigned)jk) ? a.dscale : 0.f;
        const float pd = p * keep;
        const float ds = p * (dot<HD>(go, v) * keep - D);
#pragma unroll
        for (int d = 0; d < HD; ++d) { dv[d] += pd * go[d]; dk[d] += ds * q[d]; }
    }
    if (!live) return;
#pragma unroll
    for (int d = 0; d < HD; ++d) dk[d] *= a.scale;
    RowLoad<HD, T>::store(a.dqkv, (row0 + j) * 3 * a.H + a.H + h * HD, dk);
    RowLoad<HD, T>::store(a.dqkv, (row0 + j) * 3 * a.H + 2 * a.H + h * HD, dv);
}

bool sattn_eligible(int dtype, int S, int H, int nh) {
    static const bool off = measure_env("NBCI_SMALL_ATTN", 1) == 0;
    if (off || nh <= 0 || H % nh) return false;
    const int hd = H / nh;
    (void)dtype;
    return (hd == 16 || hd == 32 || hd == 64) && S >= 1;
}

size_t sattn_stat_floats(int NS, int nh, int S) { return (size_t)NS * nh * S; }

template <typename T>
static int sattn_dispatch(int which, int hd, const SAArgs& a, hipStream_t s) {
    dim3 g(a.NS * a.nh, (a.S + 63) / 64);
#define SA_LAUNCH(K, HDV) hipLaunchKernelGGL((K<HDV, T>), g, dim3(64), 0, s, a)
#define SA_HD(K) do { if (hd == 16) SA_LAUNCH(K, 16); else if (hd == 32) SA_LAUNCH(K, 32); else SA_LAUNCH(K, 64); } while (0)
    if (which == 0) SA_HD(sattn_fwd_kernel);
    else if (which == 1) SA_HD(sattn_bwd_q_kernel);
    else SA_HD(sattn_bwd_kv_kernel);
#undef SA_HD
#undef SA_LAUNCH
    return check_launch("small attention");
}

static int sattn_args(SAArgs& a, int NS, int nh, int S, int H, float drop_p, uint32_t seed, uint32_t site) {
    NBCI_REQUIRE((long long)NS * nh * S * (long long)S < (1ll << 32), NBCI_ESHAPE, "small attention: too large for the 32-bit dropout counter");
    a.NS = NS; a.nh = nh; a.S = S; a.H = H;
    a.scale = 1.0f / sqrtf((float)(H / nh));
    a.thr = drop_threshold(drop_p);
    a.dscale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    a.key = drop_key(seed, site);
    return NBCI_OK;
}

int sattn_fwd_launch(const void* qkv, void* out, float* L, int dtype, int NS, int nh, int S, int H, float drop_p, uint32_t seed, uint32_t site,
                     hipStream_t s) {
    NBCI_REQUIRE(sattn_eligible(dtype, S, H, nh), NBCI_ESHAPE, "small attention: head size must be 16, 32 or 64");
    SAArgs a{};
    int rc = sattn_args(a, NS, nh, S, H, drop_p, seed, site);
    if (rc != NBCI_OK) return rc;
    a.qkv = qkv; a.out = out; a.L = L;
    return dtype == NBCI_BF16 ? sattn_dispatch<bf16_t>(0, H / nh, a, s) : sattn_dispatch<float>(0, H / nh, a, s);
}

int sattn_bwd_launch(const void* qkv, const void* out, const void* dout, const float* L, float* Dsum, void* dqkv, int dtype, int NS, int nh, int S,
                     int H, float drop_p, uint32_t seed, uint32_t site, hipStream_t s) {
    NBCI_REQUIRE(sattn_eligible(dtype, S, H, nh), NBCI_ESHAPE, "small attention: head size must be 16, 32 or 64");
    SAArgs a{};
    int rc = sattn_args(a, NS, nh, S, H, drop_p, seed, site);
    if (rc != NBCI_OK) return rc;
    a.qkv = qkv; a.out = (void*)out; a.dout = dout; a.L = (float*)L; a.Dsum = Dsum; a.dqkv = dqkv;
    rc = dtype == NBCI_BF16 ? sattn_dispatch<bf16_t>(1, H / nh, a, s) : sattn_dispatch<float>(1, H / nh, a, s);
    if (rc != NBCI_OK) return rc;
    return dtype == NBCI_BF16 ? sattn_dispatch<bf16_t>(2, H / nh, a, s) : sattn_dispatch<float>(2, H / nh, a, s);
}

}  // namespace nbci
