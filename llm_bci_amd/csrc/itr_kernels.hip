// itr_kernels.hip — the non-GEMM kernels that only the iTransformer SSL path needs (models/masker.py,
// models/itransformer.py). All HBM-bound: one pass over the (B,T,N) spike tensor each, transposes go
// through a padded LDS tile so both sides stay coalesced. See kernels.h for the launch API.
#include <algorithm>

#include "kernels.h"

namespace nbci {

template <typename T> __device__ __forceinline__ void stv(T* p, long long i, float v);
template <> __device__ __forceinline__ void stv<float>(float* p, long long i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void stv<bf16_t>(bf16_t* p, long long i, float v) { p[i] = f2bf(v); }

static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NBCI_EHIP, std::string(what) + ": " + hipGetErrorString(e));
    return NBCI_OK;
}

// 24-bit uniform in [0,1) from the counter RNG (mirrored by oracle/itransformer.py uniform())
__device__ __forceinline__ float uni24(uint32_t seed, uint32_t site, uint32_t idx) {
    return (float)(rng_u32(seed, site, idx) >> 8) * (1.0f / 16777216.0f);
}
// order-preserving float <-> uint map, so the tensor maximum can use atomicMax
__device__ __forceinline__ unsigned f2ord(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

// ------------------------------------------------------------------------------------------
// Masker.forward (models/masker.py:44-104)
// ------------------------------------------------------------------------------------------
// the mask bit of element i = (b, t, c): constant along the axes the mode says (masker.py:54-93, masker copy.py:65-117)
__device__ __forceinline__ bool masker_bit(const nbci_masker_desc& d, long long i, int b, int t, int c) {
    switch (d.mode) {
        case NBCI_MASK_TEMPORAL: {   // bernoulli over (B,T), optionally widened: conv1d(..., ones(span), 'same') >= 1
            bool m = false;
            const int left = (d.timespan - 1) / 2;
            for (int j = 0; j < d.timespan; ++j) {
                const int tt = t - left + j;
                if (tt >= 0 && tt < d.T) m = m || (uni24(d.seed, d.site, (uint32_t)(b * d.T + tt)) < d.ratio);
            }
            return m;
        }
        case NBCI_MASK_NEURON: return uni24(d.seed, d.site, (uint32_t)(b * d.N + c)) < d.ratio;
        case NBCI_MASK_RANDOM: return uni24(d.seed, d.site, (uint32_t)i) < d.ratio;
        case NBCI_MASK_TABLE_BN: return uni24(d.seed, d.site, (uint32_t)(b * d.N + c)) < d.probs[b * d.N + c];
        case NBCI_MASK_TABLE_N: return uni24(d.seed, d.site, (uint32_t)c) < d.probs[c];
        case NBCI_MASK_TABLE_T: return uni24(d.seed, d.site, (uint32_t)t) < d.probs[t];   // forward-pred (masker copy.py:81-85)
        default: return d.ext_mask[i] != 0;   // NBCI_MASK_GIVEN
    }
}

// pass 1: mask bit per element (constant along the axes the mode says), zero `zero_ratio` of the masked
// elements, tensor maximum of the result (needed by the random replacement, masker.py:101).
__global__ __launch_bounds__(256) void masker_zero_kernel(nbci_masker_desc d, unsigned* __restrict__ maxbits) {
    const long long n = (long long)d.B * d.T * d.N;
    float mx = -INFINITY;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % d.N);
        const long long bt = i / d.N;
        const int t = (int)(bt % d.T), b = (int)(bt / d.T);
        const bool m = masker_bit(d, i, b, t, c);
        float v = d.in[i];
        if (m && uni24(d.seed, d.site + 1, (uint32_t)i) < d.zero_ratio) v = 0.f;
        d.out[i] = v;
        const bool mt = m && (!d.target_bn || d.target_bn[b * d.N + c] != 0.f);   // intra-region: targets = masked bins of the target regions only (masker copy.py:133)
        d.mask[i] = d.accumulate ? (d.mask[i] | (long long)mt) : (long long)mt;
        mx = fmaxf(mx, v);
    }
    // ONE atomic per block: every atomic of the launch goes to the same word and they serialise in L2 at ~10 ns each - one per wave of
    // 4096 blocks was 160 of this kernel's 192 us (iTransformer, 1500 channels)
    __shared__ float wmax[4];
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(maxbits, f2ord(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
}

// pass 2: of the masked elements that were not zeroed, `random_ratio` become U(0, max) (masker.py:100-102)
__global__ __launch_bounds__(256) void masker_random_kernel(nbci_masker_desc d, const unsigned* __restrict__ maxbits) {
    const long long n = (long long)d.B * d.T * d.N;
    const float mx = ord2f(*maxbits);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % d.N);
        const long long bt = i / d.N;
        const int t = (int)(bt % d.T), b = (int)(bt / d.T);
        const bool m = masker_bit(d, i, b, t, c);
        if (!m) continue;
        if (uni24(d.seed, d.site + 1, (uint32_t)i) < d.zero_ratio) continue;   // zeroed in pass 1
        if (uni24(d.seed, d.site + 2, (uint32_t)i) < d.random_ratio) d.out[i] = mx * uni24(d.seed, d.site + 3, (uint32_t)i);
    }
}

int masker_launch(const nbci_masker_desc& d, hipStream_t s) {
    NBCI_REQUIRE(d.B > 0 && d.T > 0 && d.N > 0, NBCI_ESHAPE, "masker: B, T, N must be positive");
    NBCI_REQUIRE(d.in && d.out && d.mask && d.scratch, NBCI_EINVAL, "masker: in, out, mask and scratch are required");
    NBCI_REQUIRE(d.mode >= NBCI_MASK_TEMPORAL && d.mode <= NBCI_MASK_TABLE_T, NBCI_EINVAL, "masker: unknown mode");
    NBCI_REQUIRE(!((d.mode == NBCI_MASK_TABLE_BN || d.mode == NBCI_MASK_TABLE_N || d.mode == NBCI_MASK_TABLE_T) && !d.probs), NBCI_EINVAL,
                 "masker: probs table required");
    NBCI_REQUIRE(!(d.mode == NBCI_MASK_GIVEN && !d.ext_mask), NBCI_EINVAL, "masker: ext_mask required");
    NBCI_REQUIRE(d.mode != NBCI_MASK_TEMPORAL || (d.timespan >= 1 && d.timespan <= 64), NBCI_EINVAL, "masker: timespan must be in 1..64");
    NBCI_REQUIRE((long long)d.B * d.T * d.N < (1ll << 32), NBCI_ESHAPE, "masker: tensor too large for the 32-bit RNG counter");
    const long long n = (long long)d.B * d.T * d.N;
    const int blocks = (int)std::min<long long>((n + 255) / 256, 256 * 4);   // (grid-stride loops; few blocks = few same-word atomics)
    NBCI_CHECK_HIP(hipMemsetAsync(d.scratch, 0, 4, s));   // ordered-uint encoding: 0 < every float
    hipLaunchKernelGGL(masker_zero_kernel, dim3(blocks), dim3(256), 0, s, d, (unsigned*)d.scratch);
    int rc = check_launch("masker_zero");
    if (rc != NBCI_OK) return rc;
    if (d.random_ratio > 0.f && d.zero_ratio < 1.f) {
        hipLaunchKernelGGL(masker_random_kernel, dim3(blocks), dim3(256), 0, s, d, (const unsigned*)d.scratch);
        rc = check_launch("masker_random");
    }
    return rc;
}

// ------------------------------------------------------------------------------------------
// (B,T,N) -> (B*N, T) f32: the channel-as-token view the embedding MLP reads (itransformer.py:187)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void btn_to_bnt_kernel(const float* __restrict__ in, float* __restrict__ out, int T, int N) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int t = t0 + r, n = n0 + tx;
        tile[r][tx] = (t < T && n < N) ? in[((long long)b * T + t) * N + n] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, t = t0 + tx;
        if (n < N && t < T) out[((long long)b * N + n) * T + t] = tile[tx][r];
    }
}

int btn_to_bnt_launch(const float* in, float* out, int B, int T, int N, hipStream_t s) {
    hipLaunchKernelGGL(btn_to_bnt_kernel, dim3((N + 31) / 32, (T + 31) / 32, B), dim3(256), 0, s, in, out, T, N);
    return check_launch("btn_to_bnt");
}

// ------------------------------------------------------------------------------------------
// token assembly (itransformer.py:187-209): row (b,0) = cls; row (b,1+n) = LN_e(t2[b,n]) + LN_c(E_c)[ss[b,n]]
// (+ region table), then embed dropout over the whole (B,S,H) tensor. One wave per row.
// ------------------------------------------------------------------------------------------
template <int NV, typename TO>
__global__ __launch_bounds__(256) void itr_assemble_fwd_kernel(const float* __restrict__ t2, const float* __restrict__ w,
                                                               const float* __restrict__ bia, const float* __restrict__ tab1,
                                                               const long long* __restrict__ idx1, const float* __restrict__ tab2,
                                                               const long long* __restrict__ idx2, const float* __restrict__ cls,
                                                               float* __restrict__ x32, TO* __restrict__ xb, float* __restrict__ mean,
                                                               float* __restrict__ rstd, int B, int N, int H, int use_cls,
                                                               unsigned thr, float dscale, uint32_t key, const float* __restrict__ extra) {
    const int lane = threadIdx.x & 63;
    const int S = N + use_cls;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)B * S) return;
    const int b = (int)(row / S), sidx = (int)(row % S);
    float4 v[NV];
    if (use_cls && sidx == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = (k * 64 + lane) * 4;
            v[k] = (c < H) ? *(const float4*)(cls + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else {
        const long long r0 = (long long)b * N + (sidx - use_cls);
        const float* xr = t2 + r0 * H;
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = (k * 64 + lane) * 4;
            v[k] = (c < H) ? *(const float4*)(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            sum += v[k].x + v[k].y + v[k].z + v[k].w;
        }
        const float mu = wave_sum(sum) / (float)H;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = (k * 64 + lane) * 4;
            if (c < H) {
                const float a = v[k].x - mu, b2 = v[k].y - mu, c2 = v[k].z - mu, d2 = v[k].w - mu;
                q += a * a + b2 * b2 + c2 * c2 + d2 * d2;
            }
        }
        const float rs = 1.0f / sqrtf(wave_sum(q) / (float)H + 1e-5f);
        if (lane == 0) { mean[r0] = mu; rstd[r0] = rs; }
        const float* e1 = tab1 ? tab1 + idx1[r0] * H : nullptr;
        const float* e2 = tab2 ? tab2 + idx2[r0] * H : nullptr;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = (k * 64 + lane) * 4;
            if (c < H) {
                const float4 ww = *(const float4*)(w + c), bv = *(const float4*)(bia + c);
                float4 r = make_float4((v[k].x - mu) * rs * ww.x + bv.x, (v[k].y - mu) * rs * ww.y + bv.y,
                                       (v[k].z - mu) * rs * ww.z + bv.z, (v[k].w - mu) * rs * ww.w + bv.w);
                if (e1) { const float4 e = *(const float4*)(e1 + c); r.x += e.x; r.y += e.y; r.z += e.z; r.w += e.w; }
                if (e2) { const float4 e = *(const float4*)(e2 + c); r.x += e.x; r.y += e.y; r.z += e.z; r.w += e.w; }
                if (extra) { const float4 e = *(const float4*)(extra + r0 * H + c); r.x += e.x; r.y += e.y; r.z += e.z; r.w += e.w; }
                v[k] = r;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = (k * 64 + lane) * 4;
        if (c < H) {
            const long long o = row * H + c;
            float r[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
            if (thr) drop4(key, thr, (unsigned)o, dscale, r);
            *(float4*)(x32 + o) = make_float4(r[0], r[1], r[2], r[3]);
            stv<TO>(xb, o + 0, r[0]); stv<TO>(xb, o + 1, r[1]); stv<TO>(xb, o + 2, r[2]); stv<TO>(xb, o + 3, r[3]);
        }
    }
}

int itr_assemble_fwd_launch(const float* t2, const float* w, const float* b, const float* tab1, const int64_t* idx1, const float* tab2,
                            const int64_t* idx2, const float* cls, float* x32, void* xb, int xb_dtype, float* mean, float* rstd, int B,
                            int N, int H, int use_cls, float drop_p, uint32_t seed, uint32_t site, hipStream_t s, const float* extra) {
    NBCI_REQUIRE(H % 4 == 0 && H <= 4096, NBCI_ESHAPE, "itr assemble: hidden must be a multiple of 4 and <= 4096");
    NBCI_REQUIRE((long long)B * (N + use_cls) * H < (1ll << 32), NBCI_ESHAPE, "itr assemble: tensor too large for the dropout counter");
    const unsigned thr = drop_threshold(drop_p);
    const float dscale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    const long long rows = (long long)B * (N + use_cls);
    dim3 g((unsigned)((rows + 3) / 4));
    const int nv = (H + 255) / 256;
#define ASM(NVV, TO)                                                                                                        \
    hipLaunchKernelGGL((itr_assemble_fwd_kernel<NVV, TO>), g, dim3(256), 0, s, t2, w, b, tab1, (const long long*)idx1, tab2,    \
                       (const long long*)idx2, cls, x32, (TO*)xb, mean, rstd, B, N, H, use_cls, thr, dscale, drop_key(seed, site), extra)
    if (xb_dtype == NBCI_BF16) { if (nv <= 4) ASM(4, bf16_t); else ASM(16, bf16_t); }
    else { if (nv <= 4) ASM(4, float); else ASM(16, float); }
#undef ASM
    return check_launch("itr_assemble_fwd");
}

// backward of the assembly: d = dx0 * keepmask; cls grad += sum_b d[b,0]; dtok[(b,n)] = d[b,1+n]; dtab[idx[b,n]] += d[b,1+n]
template <typename TG>   // TG: storage of the incoming gradient stream (f32, or bf16)
__global__ __launch_bounds__(256) void itr_assemble_bwd_kernel(const TG* __restrict__ dx0, float* __restrict__ dtok,
                                                               float* __restrict__ dtab1, const long long* __restrict__ idx1,
                                                               float* __restrict__ dtab2, const long long* __restrict__ idx2,
                                                               float* __restrict__ dcls, RepCfg rc, int B, int N, int H, int use_cls,
                                                               unsigned thr, float dscale, uint32_t key) {
    // a thread owns columns c0 + 64 e (e = 0..3) of its wave's 256-column span: every atomic wave-instruction then covers 64 CONSECUTIVE
    // floats (f32 atomics run at full rate only in that shape; with four adjacent columns per thread each instruction touched 64 words
    // spread over 1 KB: 232 us for the 18 M adds of the channel table at 1500 channels)
    const int S = N + use_cls;
    const long long row = (long long)blockIdx.y;
    const int c0 = blockIdx.x * 1024 + (threadIdx.x >> 6) * 256 + (threadIdx.x & 63);
    const int b = (int)(row / S), sidx = (int)(row % S);
    float r[4];
    bool ok[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = c0 + 64 * e;
        ok[e] = c < H;
        const long long o = row * H + c;
        r[e] = ok[e] ? (float)dx0[o] : 0.f;
        if (thr && ok[e]) r[e] = drop_keep(key, thr, (unsigned)o) ? r[e] * dscale : 0.f;
    }
    if (use_cls && sidx == 0) {
        float* dc = rep_ptr(dcls, rc, (unsigned)b) + c0;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (ok[e]) atomicAdd(dc + 64 * e, r[e]);
        return;
    }
    const long long r0 = (long long)b * N + (sidx - use_cls);
#pragma unroll
    for (int e = 0; e < 4; ++e) if (ok[e]) dtok[r0 * H + c0 + 64 * e] = r[e];
    if (dtab1) {
        float* p = dtab1 + idx1[r0] * H + c0;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (ok[e]) atomicAdd(p + 64 * e, r[e]);
    }
    if (dtab2) {
        float* p = dtab2 + idx2[r0] * H + c0;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (ok[e]) atomicAdd(p + 64 * e, r[e]);
    }
}

int itr_assemble_bwd_launch(const void* dx0, float* dtok, float* dtab1, const int64_t* idx1, float* dtab2, const int64_t* idx2,
                            float* dcls, RepCfg rc, int B, int N, int H, int use_cls, float drop_p, uint32_t seed, uint32_t site,
                            hipStream_t s, int dx_dtype) {
    const unsigned thr = drop_threshold(drop_p);
    const float dscale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    dim3 g((H / 4 + 255) / 256, (unsigned)((long long)B * (N + use_cls)));
    if (dx_dtype == NBCI_BF16)
        hipLaunchKernelGGL((itr_assemble_bwd_kernel<bf16_t>), g, dim3(256), 0, s, (const bf16_t*)dx0, dtok, dtab1, (const long long*)idx1, dtab2,
                           (const long long*)idx2, dcls, rc, B, N, H, use_cls, thr, dscale, drop_key(seed, site));
    else
        hipLaunchKernelGGL((itr_assemble_bwd_kernel<float>), g, dim3(256), 0, s, (const float*)dx0, dtok, dtab1, (const long long*)idx1, dtab2,
                           (const long long*)idx2, dcls, rc, B, N, H, use_cls, thr, dscale, drop_key(seed, site));
    return check_launch("itr_assemble_bwd");
}

// ------------------------------------------------------------------------------------------
// mlm head tail (itransformer.py:339-352): pred[(b,1+n), t] -> preds (B,T,N); masked Poisson-NLL / MSE sum;
// n_examples; d(loss)/d(pred) back in the token-row layout (operand dtype, zero in CLS rows and pad columns).
// ------------------------------------------------------------------------------------------
template <typename TO>
__global__ __launch_bounds__(256) void itr_mlm_loss_kernel(const float* __restrict__ pred, int ldp, const float* __restrict__ targets,
                                                           const long long* __restrict__ mask, const long long* __restrict__ smask,
                                                           float* __restrict__ preds_out, long long* __restrict__ mask_out,
                                                           TO* __restrict__ dpred, float* __restrict__ loss,
                                                           unsigned long long* __restrict__ nex, int T, int N, int use_cls, int kind,
                                                           float gscale) {
    __shared__ float tp[32][33];   // [n][t] raw prediction
    __shared__ float tg[32][33];   // [n][t] gradient
    __shared__ float red[4];
    __shared__ unsigned cnt[4];
    const int b = blockIdx.z, t0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int S = N + use_cls;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {   // read rows n, lanes along t (contiguous in the GEMM output)
        const int n = n0 + r, t = t0 + tx;
        tp[r][tx] = (n < N && t < T) ? pred[((long long)b * S + use_cls + n) * ldp + t] : 0.f;
    }
    __syncthreads();
    float lsum = 0.f;
    unsigned lcnt = 0;
    for (int r = ty; r < 32; r += 8) {   // rows t, lanes along n (contiguous in (B,T,N))
        const int t = t0 + r, n = n0 + tx;
        float g = 0.f;
        if (t < T && n < N) {
            const long long i = ((long long)b * T + t) * N + n;
            const float raw = tp[tx][r];
            const float y = targets[i];
            const long long m = mask[i] & smask[(long long)b * T + t];
            float p = raw, el, dl;
            if (kind == NBCI_LOSS_POISSON_LOG) { const float e = expf(p); el = e - y * p; dl = e - y; }
            else if (kind == NBCI_LOSS_POISSON_RATE) { p = fmaxf(raw, 0.f); el = p - y * logf(p + 1e-8f); dl = (raw > 0.f) ? 1.f - y / (p + 1e-8f) : 0.f; }
            else { const float df = p - y; el = df * df; dl = 2.f * df; }
            preds_out[i] = p;
            mask_out[i] = m;
            if (m) { lsum += el; ++lcnt; g = dl * gscale; }
        }
        tg[tx][r] = g;
    }
    __syncthreads();
    if (dpred) {
        for (int r = ty; r < 32; r += 8) {
            const int n = n0 + r, t = t0 + tx;
            if (n < N && t < T) stv<TO>(dpred, ((long long)b * S + use_cls + n) * ldp + t, tg[r][tx]);
        }
    }
    lsum = wave_sum(lsum);
    float fc = wave_sum((float)lcnt);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = lsum; cnt[threadIdx.x >> 6] = (unsigned)fc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(loss, red[0] + red[1] + red[2] + red[3]);
        atomicAdd(nex, (unsigned long long)(cnt[0] + cnt[1] + cnt[2] + cnt[3]));
    }
}

int itr_mlm_loss_launch(const float* pred, int ldp, const float* targets, const int64_t* mask, const int64_t* smask, float* preds_out,
                        int64_t* mask_out, void* dpred, int d_dtype, float* loss, int64_t* n_examples, int B, int T, int N, int use_cls,
                        int kind, float grad_scale, hipStream_t s) {
    NBCI_REQUIRE(kind >= NBCI_LOSS_POISSON_LOG && kind <= NBCI_LOSS_MSE, NBCI_EINVAL, "mlm loss: unknown loss kind");
    NBCI_CHECK_HIP(hipMemsetAsync(loss, 0, 4, s));
    NBCI_CHECK_HIP(hipMemsetAsync(n_examples, 0, 8, s));
    if (dpred)   // CLS rows and the pad columns (ldp > T) must read as zeros in the backward GEMMs
        NBCI_CHECK_HIP(hipMemsetAsync(dpred, 0, (size_t)B * (N + use_cls) * ldp * (d_dtype == NBCI_BF16 ? 2 : 4), s));
    dim3 g((N + 31) / 32, (T + 31) / 32, B);
    if (d_dtype == NBCI_BF16)
        hipLaunchKernelGGL((itr_mlm_loss_kernel<bf16_t>), g, dim3(256), 0, s, pred, ldp, targets, (const long long*)mask,
                           (const long long*)smask, preds_out, (long long*)mask_out, (bf16_t*)dpred, loss,
                           (unsigned long long*)n_examples, T, N, use_cls, kind, grad_scale);
    else
        hipLaunchKernelGGL((itr_mlm_loss_kernel<float>), g, dim3(256), 0, s, pred, ldp, targets, (const long long*)mask,
                           (const long long*)smask, preds_out, (long long*)mask_out, (float*)dpred, loss,
                           (unsigned long long*)n_examples, T, N, use_cls, kind, grad_scale);
    return check_launch("itr_mlm_loss");
}

// ------------------------------------------------------------------------------------------
// Linear(1 -> W) + ReLU on a scalar per row: the first layer of the UnivariateTransformer's embed_spikes (models/itransformer.py:48-52,81:
// the scalar is one bin's spike count) and of depth_embeddings (:145-150,201: the scalar is the neuron's depth).
//   out[r][j] = relu(x(r) * w[j] + b[j])
// period > 0: rows come in sequences of `period` = 1 + T rows whose first row is the CLS slot (written as zeros, so it adds nothing to
// the weight gradients downstream) and x(r) = x[(r / period) * (period - 1) + r % period - 1]; period == 0: x(r) = x[r].
// ------------------------------------------------------------------------------------------
template <typename TO>
__global__ __launch_bounds__(256) void scalar_lin_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                             TO* __restrict__ out, long long rows, int W, int period) {
    const long long n = rows * (long long)(W / 4);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long r = i / (W / 4);
        const int c = (int)(i % (W / 4)) * 4;
        float xv = 0.f;
        bool cls = false;
        if (period > 0) {
            const long long z = r / period;
            const int j = (int)(r % period);
            cls = j == 0;
            if (!cls) xv = x[z * (period - 1) + j - 1];
        } else {
            xv = x[r];
        }
        const float4 ww = *(const float4*)(w + c), bb = *(const float4*)(b + c);
        float v[4] = {fmaxf(xv * ww.x + bb.x, 0.f), fmaxf(xv * ww.y + bb.y, 0.f), fmaxf(xv * ww.z + bb.z, 0.f), fmaxf(xv * ww.w + bb.w, 0.f)};
#pragma unroll
        for (int e = 0; e < 4; ++e) stv<TO>(out, r * W + c + e, cls ? 0.f : v[e]);
    }
}

int scalar_lin_fwd_launch(const float* x, const float* w, const float* b, void* out, int out_dtype, long long rows, int W, int period, hipStream_t s) {
    NBCI_REQUIRE(W % 4 == 0 && rows > 0, NBCI_ESHAPE, "scalar linear: width must be a multiple of 4");
    const long long n = rows * (W / 4);
    dim3 g((unsigned)std::min<long long>((n + 255) / 256, 1 << 16));
    if (out_dtype == NBCI_BF16) hipLaunchKernelGGL((scalar_lin_fwd_kernel<bf16_t>), g, dim3(256), 0, s, x, w, b, (bf16_t*)out, rows, W, period);
    else hipLaunchKernelGGL((scalar_lin_fwd_kernel<float>), g, dim3(256), 0, s, x, w, b, (float*)out, rows, W, period);
    return check_launch("scalar_lin_fwd");
}

// its backward: dw[j] += sum_r du[r][j] x(r), db[j] += sum_r du[r][j] (du already carries relu' - the gated data-gradient GEMM wrote it).
// A block walks SL_ROWS rows; a thread owns one column of a 256-column pass; replicated atomics at the end (one per thread and pass).
constexpr int SL_ROWS = 1024;
template <typename TI>
__global__ __launch_bounds__(256) void scalar_lin_bwd_kernel(const TI* __restrict__ du, const float* __restrict__ x, float* __restrict__ dw,
                                                             float* __restrict__ db, RepCfg rc, long long rows, int W, int period) {
    const int cpp = W < 256 ? W : 256;            // columns per pass
    const int rpar = 256 / cpp;                   // rows walked side by side
    const int tc = threadIdx.x % cpp, tr = threadIdx.x / cpp;
    const long long r0 = (long long)blockIdx.x * SL_ROWS, r1 = r0 + SL_ROWS < rows ? r0 + SL_ROWS : rows;
    if (tr >= rpar) return;
    for (int c0 = 0; c0 < W; c0 += cpp) {
        const int c = c0 + tc;
        if (c >= W) continue;
        float sw = 0.f, sb = 0.f;
        for (long long r = r0 + tr; r < r1; r += rpar) {
            float xv;
            if (period > 0) {
                const int j = (int)(r % period);
                if (j == 0) continue;              // CLS slot: no scalar input
                xv = x[(r / period) * (period - 1) + j - 1];
            } else {
                xv = x[r];
            }
            const float g = (float)du[r * W + c];
            sw += g * xv; sb += g;
        }
        atomicAdd(rep_ptr(dw, rc, blockIdx.x) + c, sw);
        atomicAdd(rep_ptr(db, rc, blockIdx.x) + c, sb);
    }
}

int scalar_lin_bwd_launch(const void* du, int du_dtype, const float* x, float* dw, float* db, RepCfg rc, long long rows, int W, int period, hipStream_t s) {
    dim3 g((unsigned)((rows + SL_ROWS - 1) / SL_ROWS));
    if (du_dtype == NBCI_BF16) hipLaunchKernelGGL((scalar_lin_bwd_kernel<bf16_t>), g, dim3(256), 0, s, (const bf16_t*)du, x, dw, db, rc, rows, W, period);
    else hipLaunchKernelGGL((scalar_lin_bwd_kernel<float>), g, dim3(256), 0, s, (const float*)du, x, dw, db, rc, rows, W, period);
    return check_launch("scalar_lin_bwd");
}

// ------------------------------------------------------------------------------------------
// UnivariateTransformer input assembly (itransformer.py:83-87): per (sample, channel) sequence of 1 + T rows,
//   row 0 = cls_embed, row 1 + t = t_in[row] + embed_pos[ts[b][t]]   (t_in = the second embed_spikes Linear's output, bias included)
// written as the operand copy xb and (y32 != NULL) the f32 residual copy.
// ------------------------------------------------------------------------------------------
template <typename TO>
__global__ __launch_bounds__(256) void uni_finish_fwd_kernel(const float* __restrict__ t_in, const float* __restrict__ pos, const long long* __restrict__ ts,
                                                             const float* __restrict__ cls, float* __restrict__ y32, TO* __restrict__ xb, long long rows,
                                                             int N, int T, int h) {
    const long long n = rows * (long long)(h / 4);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long r = i / (h / 4);
        const int c = (int)(i % (h / 4)) * 4;
        const long long z = r / (T + 1);
        const int j = (int)(r % (T + 1));
        float4 v;
        if (j == 0) {
            v = *(const float4*)(cls + c);
        } else {
            const long long b = z / N;
            const long long tpos = ts ? ts[b * T + j - 1] : (long long)(j - 1);
            const float4 a = *(const float4*)(t_in + r * h + c), p = *(const float4*)(pos + tpos * h + c);
            v = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
        }
        if (y32) *(float4*)(y32 + r * h + c) = v;
        stv<TO>(xb, r * h + c, v.x); stv<TO>(xb, r * h + c + 1, v.y); stv<TO>(xb, r * h + c + 2, v.z); stv<TO>(xb, r * h + c + 3, v.w);
    }
}

int uni_finish_fwd_launch(const float* t_in, const float* pos, const int64_t* ts, const float* cls, float* y32, void* xb, int xb_dtype, int B, int N,
                          int T, int h, hipStream_t s) {
    NBCI_REQUIRE(h % 4 == 0, NBCI_ESHAPE, "uni finish: embedder hidden size must be a multiple of 4");
    const long long rows = (long long)B * N * (T + 1), n = rows * (h / 4);
    dim3 g((unsigned)std::min<long long>((n + 255) / 256, 1 << 16));
    if (xb_dtype == NBCI_BF16)
        hipLaunchKernelGGL((uni_finish_fwd_kernel<bf16_t>), g, dim3(256), 0, s, t_in, pos, (const long long*)ts, cls, y32, (bf16_t*)xb, rows, N, T, h);
    else
        hipLaunchKernelGGL((uni_finish_fwd_kernel<float>), g, dim3(256), 0, s, t_in, pos, (const long long*)ts, cls, y32, (float*)xb, rows, N, T, h);
    return check_launch("uni_finish_fwd");
}

// its backward, part 1: dseq (the gradient stream at the embedder stack's input) -> dte = the operand copy of the token rows' gradient with the
// CLS rows zeroed (they have no embed_spikes input), and cls_embed's gradient = the sum of the CLS rows.
template <typename TG, typename TO>
__global__ __launch_bounds__(256) void uni_split_bwd_kernel(const TG* __restrict__ dseq, TO* __restrict__ dte, float* __restrict__ dcls, RepCfg rc,
                                                            long long rows, int T, int h) {
    const long long n = rows * (long long)h;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long r = i / h;
        const int c = (int)(i % h);
        const float v = (float)dseq[i];
        if (r % (T + 1) == 0) {
            atomicAdd(rep_ptr(dcls, rc, (unsigned)(r / (T + 1))) + c, v);
            stv<TO>(dte, i, 0.f);
        } else {
            stv<TO>(dte, i, v);
        }
    }
}

int uni_split_bwd_launch(const void* dseq, int stream_dtype, void* dte, int dte_dtype, float* dcls, RepCfg rc, long long rows, int T, int h, hipStream_t s) {
    const long long n = rows * h;
    dim3 g((unsigned)std::min<long long>((n + 255) / 256, 1 << 16));
#define USB(TG, TO) hipLaunchKernelGGL((uni_split_bwd_kernel<TG, TO>), g, dim3(256), 0, s, (const TG*)dseq, (TO*)dte, dcls, rc, rows, T, h)
    if (stream_dtype == NBCI_BF16) { if (dte_dtype == NBCI_BF16) USB(bf16_t, bf16_t); else USB(bf16_t, float); }
    else { if (dte_dtype == NBCI_BF16) USB(float, bf16_t); else USB(float, float); }
#undef USB
    return check_launch("uni_split_bwd");
}

// part 2: embed_pos gradient: dpos[ts[b][t]][:] += sum over the sample's N channels of dseq[(b, n), 1 + t][:] (itransformer.py:85: the position
// embedding is broadcast over channels). One wave per (b, t): the channel sum stays in registers, then one atomic per column.
template <typename TG>
__global__ __launch_bounds__(256) void uni_posgrad_kernel(const TG* __restrict__ dseq, const long long* __restrict__ ts, float* __restrict__ dpos, int B,
                                                          int N, int T, int h) {
    const int lane = threadIdx.x & 63;
    const long long bt = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bt >= (long long)B * T) return;
    const int b = (int)(bt / T), t = (int)(bt % T);
    const long long tpos = ts ? ts[bt] : (long long)t;
    for (int c0 = 0; c0 < h; c0 += 64) {
        const int c = c0 + lane;
        if (c >= h) continue;
        float acc = 0.f;
        const TG* p = dseq + (((long long)b * N) * (T + 1) + 1 + t) * h + c;
#pragma unroll 8
        for (int n = 0; n < N; ++n) acc += (float)p[(long long)n * (T + 1) * h];
        atomicAdd(dpos + tpos * h + c, acc);
    }
}

int uni_posgrad_launch(const void* dseq, int stream_dtype, const int64_t* ts, float* dpos, int B, int N, int T, int h, hipStream_t s) {
    dim3 g((unsigned)(((long long)B * T + 3) / 4));
    if (stream_dtype == NBCI_BF16) hipLaunchKernelGGL((uni_posgrad_kernel<bf16_t>), g, dim3(256), 0, s, (const bf16_t*)dseq, (const long long*)ts, dpos, B, N, T, h);
    else hipLaunchKernelGGL((uni_posgrad_kernel<float>), g, dim3(256), 0, s, (const float*)dseq, (const long long*)ts, dpos, B, N, T, h);
    return check_launch("uni_posgrad");
}

}  // namespace nbci
