// kernels.hip — every non-GEMM kernel of the NDT1-CTC train step. All are HBM/latency-bound:
// wave64 shuffles for row reductions, 16-byte accesses where layout allows, no LDS staging
// beyond small cross-wave reductions. See kernels.h for the reference lines each replaces.
#include "kernels.h"
#include <algorithm>
#include <type_traits>

namespace nbci {

template <typename T> __device__ __forceinline__ float ldf(const T* p, long long i);
template <> __device__ __forceinline__ float ldf<float>(const float* p, long long i) { return p[i]; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p, long long i) { return bf2f(p[i]); }
template <typename T> __device__ __forceinline__ void stf(T* p, long long i, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, long long i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t* p, long long i, float v) { p[i] = f2bf(v); }

static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NBCI_EHIP, std::string(what) + ": " + hipGetErrorString(e));
    return NBCI_OK;
}

#define DISPATCH_DTYPE(dt, T, ...)                                   \
    do {                                                             \
        if ((dt) == NBCI_BF16) { using T = bf16_t; __VA_ARGS__; }    \
        else { using T = float; __VA_ARGS__; }                       \
    } while (0)

// ------------------------------------------------------------------------------------------
// smoothing + noise
// ------------------------------------------------------------------------------------------
constexpr int SM_TCHUNK = 30;   // outputs per thread (even: white noise is drawn in (cos, sin) pairs)
constexpr int SM_MAXTAPS = 64;

// two standard normals from one Box-Muller transform (pair index = element index >> 1 along time)
__device__ __forceinline__ void normal_pair(uint32_t seed, uint32_t site, uint32_t pidx, float& z0, float& z1) {
    const uint32_t a = rng_u32(seed, site, pidx);
    const uint32_t b = rng_u32(seed ^ 0x5bd1e995U, site + 0x1000193U, pidx);
    const float rad = sqrtf(-2.0f * __logf(rng_uniform01(a)));
    const float ang = 6.283185307179586f * rng_uniform01(b);
    z0 = rad * __cosf(ang);
    z1 = rad * __sinf(ang);
}

// One thread = one (batch, channel) column x SM_TCHUNK consecutive bins; lanes run along channels
// (coalesced). The taps slide over a register window, so every input element is loaded once per thread.
template <typename TO, int NT>
__global__ __launch_bounds__(256) void smooth_noise_kernel(const float* __restrict__ in, TO* __restrict__ out, int B,
                                                           int T, int N, const float* __restrict__ taps, int ntaps,
                                                           float white_sd, float offset_sd, uint32_t seed) {
    __shared__ float stap[SM_MAXTAPS];
    if (threadIdx.x < ntaps) stap[threadIdx.x] = taps[threadIdx.x];
    __syncthreads();
    const long long col = (long long)blockIdx.x * 256 + threadIdx.x;
    if (col >= (long long)B * N) return;
    const int b = (int)(col / N), n = (int)(col % N);
    const int half = (ntaps - 1) / 2;
    const int t0 = blockIdx.y * SM_TCHUNK;
    const float* src = in + (long long)b * T * N + n;
    float off = 0.f;
    if (offset_sd != 0.f) off = offset_sd * rng_normal(seed, 2u, (uint32_t)(b * N + n));
    if constexpr (NT > 0) {
        float win[NT + SM_TCHUNK - 1];
#pragma unroll
        for (int i = 0; i < NT + SM_TCHUNK - 1; ++i) {
            const int tt = t0 + i - (NT - 1) / 2;
            win[i] = (tt >= 0 && tt < T) ? src[(long long)tt * N] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < SM_TCHUNK; j += 2) {
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int i = 0; i < NT; ++i) { a0 += stap[i] * win[j + i]; a1 += stap[i] * win[j + 1 + i]; }
            const int t = t0 + j;
            const long long o = ((long long)b * T + t) * N + n;
            if (white_sd != 0.f) {
                // pair (t, t+1) of one channel shares a Box-Muller draw; index = position of the even bin
                float z0, z1;
                normal_pair(seed, 1u, (uint32_t)(((long long)b * N + n) * ((T + 1) / 2) + (t >> 1)), z0, z1);
                a0 += white_sd * z0; a1 += white_sd * z1;
            }
            if (t < T) stf<TO>(out, o, a0 + off);
            if (t + 1 < T) stf<TO>(out, o + N, a1 + off);
        }
    } else {
        for (int t = t0; t < min(T, t0 + SM_TCHUNK); t += 2) {
            float a0 = 0.f, a1 = 0.f;
            if (ntaps > 0) {
                for (int i = 0; i < ntaps; ++i) {
                    const int tt = t + i - half;
                    if (tt >= 0 && tt < T) a0 += stap[i] * src[(long long)tt * N];
                    if (tt + 1 >= 0 && tt + 1 < T) a1 += stap[i] * src[(long long)(tt + 1) * N];
                }
            } else {
                a0 = src[(long long)t * N];
                a1 = (t + 1 < T) ? src[(long long)(t + 1) * N] : 0.f;
            }
            const long long o = ((long long)b * T + t) * N + n;
            if (white_sd != 0.f) {
                float z0, z1;
                normal_pair(seed, 1u, (uint32_t)(((long long)b * N + n) * ((T + 1) / 2) + (t >> 1)), z0, z1);
                a0 += white_sd * z0; a1 += white_sd * z1;
            }
            stf<TO>(out, o, a0 + off);
            if (t + 1 < T) stf<TO>(out, o + N, a1 + off);
        }
    }
}

int smooth_noise_launch(const float* spikes, void* out, int out_dtype, int B, int T, int N, const float* taps,
                        int ntaps, float white_sd, float offset_sd, uint32_t seed, hipStream_t s) {
    NBCI_REQUIRE(ntaps <= SM_MAXTAPS, NBCI_ESHAPE, "smooth: too many taps (max 64)");
    dim3 grid((unsigned)(((long long)B * N + 255) / 256), (T + SM_TCHUNK - 1) / SM_TCHUNK);
    DISPATCH_DTYPE(out_dtype, TO, {
        if (ntaps == 13)  // the recipe's smooth_sd = 2: taps unrolled over a register window
            hipLaunchKernelGGL((smooth_noise_kernel<TO, 13>), grid, dim3(256), 0, s, spikes, (TO*)out, B, T, N, taps, ntaps,
                               white_sd, offset_sd, seed);
        else
            hipLaunchKernelGGL((smooth_noise_kernel<TO, 0>), grid, dim3(256), 0, s, spikes, (TO*)out, B, T, N, taps, ntaps,
                               white_sd, offset_sd, seed);
    });
    return check_launch("smooth_noise");
}

// ------------------------------------------------------------------------------------------
// token prep
// ------------------------------------------------------------------------------------------
// tmask is (B, npre + Tp): the npre learned prefix tokens (day / block, ndt1.py:192-201) are always valid keys
__global__ void token_prep_kernel(const int64_t* mask, const int64_t* ts, const int64_t* lens, int B, int T, int Tp,
                                  int size, int stride, int32_t* tmask, int64_t* tts, int32_t* tlens, int npre) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * Tp) {
        const int b = i / Tp, j = i % Tp;
        int64_t prod = 1;
        for (int r = 0; r < size; ++r) prod *= mask[(long long)b * T + j * stride + r];
        tmask[(long long)b * (Tp + npre) + npre + j] = prod != 0 ? 1 : 0;
        tts[i] = ts[(long long)b * T + j];
    }
    if (i < B * npre) tmask[(long long)(i / npre) * (Tp + npre) + i % npre] = 1;
    if (i < B) {
        // (1 + (len - size) / stride) in floating point, truncating cast (ndt1.py:208)
        const double v = 1.0 + ((double)lens[i] - (double)size) / (double)stride;
        tlens[i] = (int32_t)v;
    }
}

int token_prep_launch(const int64_t* mask, const int64_t* ts, const int64_t* lens, int B, int T, int Tp, int size,
                      int stride, int32_t* tmask, int64_t* tts, int32_t* tlens, hipStream_t s, int npre) {
    const int n = max(max(B * Tp, B), B * npre);
    hipLaunchKernelGGL(token_prep_kernel, dim3((n + 255) / 256), dim3(256), 0, s, mask, ts, lens, B, T, Tp, size, stride,
                       tmask, tts, tlens, npre);
    return check_launch("token_prep");
}

// prefix tokens, forward: x (B, npre + Tp, H) = [table0[idx0[b]], (table1[idx1[b]]), xtok[b, :]] then the embedder dropout over ALL of
// it (ndt1.py:192-203), dropout stream index = element offset in x
template <typename T>   // T: storage type of x (f32, or a bf16 residual stream)
__global__ __launch_bounds__(256) void prefix_assemble_kernel(const float* __restrict__ xtok, const float* __restrict__ tab0,
                                                              const int64_t* __restrict__ idx0, const float* __restrict__ tab1,
                                                              const int64_t* __restrict__ idx1, T* __restrict__ x, int B, int Tp, int npre,
                                                              int H, unsigned thr, float dscale, uint32_t key) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    const long long total = (long long)B * (Tp + npre) * H;
    if (i >= total) return;
    const int c = (int)(i % H);
    const long long row = i / H;
    const int b = (int)(row / (Tp + npre)), j = (int)(row % (Tp + npre));
    const float* src;
    if (j >= npre) src = xtok + ((long long)b * Tp + (j - npre)) * H + c;
    else if (j == 0) src = tab0 + idx0[b] * H + c;
    else src = tab1 + idx1[b] * H + c;
    const float4 a = *(const float4*)src;
    float v[4] = {a.x, a.y, a.z, a.w};
    if (thr) drop4(key, thr, (unsigned)i, dscale, v);
    st4f(x + i, make_float4(v[0], v[1], v[2], v[3]));
}

int prefix_assemble_launch(const float* xtok, const float* tab0, const int64_t* idx0, const float* tab1, const int64_t* idx1, void* x,
                           int B, int Tp, int npre, int H, float drop_p, uint32_t seed, uint32_t site, hipStream_t s, int x_dtype) {
    NBCI_REQUIRE(xtok && tab0 && idx0 && x && npre >= 1 && npre <= 2 && H % 4 == 0 && (npre == 1 || (tab1 && idx1)), NBCI_EINVAL,
                 "prefix_assemble: bad argument");
    const long long n4 = (long long)B * (Tp + npre) * H / 4;
    DISPATCH_DTYPE(x_dtype, T, hipLaunchKernelGGL((prefix_assemble_kernel<T>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, xtok, tab0, idx0, tab1, idx1,
                                                  (T*)x, B, Tp, npre, H, drop_threshold(drop_p), drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f, drop_key(seed, site)));
    return check_launch("prefix_assemble");
}

// prefix tokens, backward: table[idx[b]] += dropout-masked dx[b, k, :] for the k-th prefix position
template <typename T>
__global__ __launch_bounds__(256) void prefix_grad_kernel(const T* __restrict__ dx, const int64_t* __restrict__ idx, float* __restrict__ dtab,
                                                          int B, int Tt, int k, int H, unsigned thr, float dscale, uint32_t key) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, c = i % H;
    const long long o = ((long long)b * Tt + k) * H + c;
    float v = ldf<T>(dx, o);
    if (thr) v = drop_keep(key, thr, (unsigned)o) ? v * dscale : 0.f;
    if (v != 0.f) atomicAdd(dtab + idx[b] * H + c, v);
}

int prefix_grad_launch(const void* dx, const int64_t* idx, float* dtab, int B, int Tt, int k, int H, float drop_p, uint32_t seed,
                       uint32_t site, hipStream_t s, int dx_dtype) {
    DISPATCH_DTYPE(dx_dtype, T, hipLaunchKernelGGL((prefix_grad_kernel<T>), dim3((unsigned)((B * H + 255) / 256)), dim3(256), 0, s, (const T*)dx, idx, dtab, B, Tt, k, H,
                                                   drop_threshold(drop_p), drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f, drop_key(seed, site)));
    return check_launch("prefix_grad");
}

// ------------------------------------------------------------------------------------------
// LayerNorm
// ------------------------------------------------------------------------------------------
// One wave per row. (Two rows per wave, so that M = 9152 rows fit one round of 8192 resident waves, measured slower: 12.7 vs 11.3 us alone;
// the kernel is a latency chain - load, two wave reductions, store - not a bandwidth one: 1144 rows take 8.3 us, 9152 rows 11.3.)
template <int NV, typename TO, typename TX>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const TX* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ b, TO* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int M, int H,
                                                     float* __restrict__ y32) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const TX* xr = x + (long long)row * H;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = (k * 64 + lane) * 4;
        v[k] = (c < H) ? ld4f(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += v[k].x + v[k].y + v[k].z + v[k].w;
    }
    const float mu = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = (k * 64 + lane) * 4;
        if (c < H) {
            const float a = v[k].x - mu, bb = v[k].y - mu, cc = v[k].z - mu, dd = v[k].w - mu;
            q += a * a + bb * bb + cc * cc + dd * dd;
        }
    }
    const float rs = 1.0f / sqrtf(wave_sum(q) / (float)H + 1e-5f);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = (k * 64 + lane) * 4;
        if (c < H) {
            const float4 ww = *(const float4*)(w + c), bv = *(const float4*)(b + c);
            const long long o = (long long)row * H + c;
            const float4 r = make_float4((v[k].x - mu) * rs * ww.x + bv.x, (v[k].y - mu) * rs * ww.y + bv.y,
                                         (v[k].z - mu) * rs * ww.z + bv.z, (v[k].w - mu) * rs * ww.w + bv.w);
            st4f(y + o, r);
            if (y32) *(float4*)(y32 + o) = r;   // f32 copy for a following residual add (post-norm layers)
        }
    }
}

template <typename TO, typename TX>
static void ln_fwd_dispatch(int nv, hipStream_t s, const TX* x, const float* w, const float* b, TO* y,
                            float* mean, float* rstd, int M, int H, float* y32) {
    const dim3 g((unsigned)((M + 3) / 4));
    if (nv <= 1) hipLaunchKernelGGL((ln_fwd_kernel<1, TO, TX>), g, dim3(256), 0, s, x, w, b, y, mean, rstd, M, H, y32);
    else if (nv == 3) hipLaunchKernelGGL((ln_fwd_kernel<3, TO, TX>), g, dim3(256), 0, s, x, w, b, y, mean, rstd, M, H, y32);   // (hidden 768: no idle quarter)
    else if (nv <= 4) hipLaunchKernelGGL((ln_fwd_kernel<4, TO, TX>), g, dim3(256), 0, s, x, w, b, y, mean, rstd, M, H, y32);
    else if (nv <= 8) hipLaunchKernelGGL((ln_fwd_kernel<8, TO, TX>), g, dim3(256), 0, s, x, w, b, y, mean, rstd, M, H, y32);
    else hipLaunchKernelGGL((ln_fwd_kernel<16, TO, TX>), g, dim3(256), 0, s, x, w, b, y, mean, rstd, M, H, y32);
}

int layernorm_fwd_launch(const void* x, int x_dtype, const float* w, const float* b, void* y, int y_dtype, float* mean, float* rstd,
                         int M, int H, hipStream_t s, float* y32) {
    NBCI_REQUIRE(H % 4 == 0 && H <= 4096, NBCI_ESHAPE, "layernorm: hidden must be a multiple of 4 and <= 4096");
    NBCI_REQUIRE(x_dtype == NBCI_F32 || y_dtype == NBCI_BF16, NBCI_EINVAL, "layernorm: a bf16 input row goes with a bf16 output");
    ProfScope ps("ln_fwd_kernel", 0.0, (double)M * H * ((x_dtype == NBCI_BF16 ? 2 : 4) + (y_dtype == NBCI_BF16 ? 2 : 4) + (y32 ? 4 : 0)), s);
    const int nv = (H + 255) / 256;
    if (x_dtype == NBCI_BF16) {
        ln_fwd_dispatch<bf16_t, bf16_t>(nv, s, (const bf16_t*)x, w, b, (bf16_t*)y, mean, rstd, M, H, y32);
    } else {
        DISPATCH_DTYPE(y_dtype, TO, (ln_fwd_dispatch<TO, float>(nv, s, (const float*)x, w, b, (TO*)y, mean, rstd, M, H, y32)));
    }
    return check_launch("layernorm_fwd");
}
int layernorm_fwd_launch(const float* x, const float* w, const float* b, void* y, int y_dtype, float* mean, float* rstd,
                         int M, int H, hipStream_t s, float* y32) {
    return layernorm_fwd_launch(x, NBCI_F32, w, b, y, y_dtype, mean, rstd, M, H, s, y32);
}

// rows per wave (`iters`; the column partial sums stay in registers across them, so more rows = fewer LDS / atomic tails, and the next
// row's loads run under the current row's arithmetic). A block is 4 waves x `iters` rows; two blocks fit a CU (182 - 240 VGPRs). What
// matters is the most loaded CU: ceil(blocks / (2 CUs)) x iters rows per wave. M = 9152 (B = 64): iters = 5 gives 458 blocks, one round;
// B = 8 (1144 rows): one row per wave, 286 blocks, every CU busy.
static inline int lnb_iters(int M) {
    const int slots = 2 * std::max(1, available_cus());
    int best = 1, best_cost = 1 << 30;
    for (int it = 1; it <= 12; ++it) {
        const int blocks = (M + 4 * it - 1) / (4 * it);
        const int cost = ((blocks + slots - 1) / slots) * it;
        if (cost <= best_cost) { best_cost = cost; best = it; }   // ties: the larger block (fewer tails)
    }
    return best;
}

// DY = float, or bf16_t: the incoming gradient as a bf16 GEMM wrote it (half the bytes). TX: the saved LayerNorm input (f32, or a bf16
// residual stream). TD: the gradient stream, read from dx_in (NULL = nothing to add to) and written to dx_out (the two may be one buffer).
// FULL: H == NV * 256, no column predicates.
// A wave owns rows row0, row0 + 4, ... of its block's 4 * iters rows and keeps TWO rows in registers: the loads of the next row are
// issued (as stored: 8 bytes per bf16x4) before the arithmetic of the current one, so the ~1 us of VALU work per row (statistics,
// dropout hash, packing) runs under the next row's memory latency instead of after it (measured alone, M = 9152, H = 1024, bf16
// streams: tools/time_ln.py).
template <typename T> struct Raw4 { using type = float4; };
template <> struct Raw4<bf16_t> { using type = bf16x4; };
template <typename T>
__device__ __forceinline__ typename Raw4<T>::type ldraw(const T* p, bool ok) {
    using R = typename Raw4<T>::type;
    if constexpr (sizeof(T) == 4) {
        return ok ? *(const R*)p : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        const R z = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
        return ok ? *(const R*)p : z;
    }
}
__device__ __forceinline__ float4 widen4(const float4& v) { return v; }
__device__ __forceinline__ float4 widen4(const bf16x4& t) { return make_float4(bf2f(t[0]), bf2f(t[1]), bf2f(t[2]), bf2f(t[3])); }

template <int NV, typename DY, typename TX, typename TD, bool FULL>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const DY* __restrict__ dy, const TX* __restrict__ x,
                                                     const float* __restrict__ w, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const TD* dx_in, TD* dx_out,
                                                     float* __restrict__ dw, float* __restrict__ db, int M, int H,
                                                     RepCfg rc, LnCast cz, int iters, int dbg_notail) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [4 waves][3][NV*256]
    using RX = typename Raw4<TX>::type;
    using RY = typename Raw4<DY>::type;
    using RD = typename Raw4<TD>::type;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float invH = 1.0f / (float)H;
    const int W = NV * 256;
    const bool want_cast = cz.out != nullptr || cz.colsum != nullptr;
    float4 gw[NV], gb[NV], gc[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) { gw[k] = make_float4(0.f, 0.f, 0.f, 0.f); gb[k] = gw[k]; gc[k] = gw[k]; }
    const int row0 = blockIdx.x * iters * 4 + wv;

    auto load_row = [&](int r, RX (&xa)[NV], RY (&da)[NV], RD (&oa)[NV], float& mu, float& rs) {
        const bool rok = r < M;
        mu = rok ? mean[r] : 0.f; rs = rok ? rstd[r] : 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = (k * 64 + lane) * 4;
            const bool ok = rok && (FULL || c < H);
            const long long o = (long long)r * H + c;
            xa[k] = ldraw(x + o, ok);
            da[k] = ldraw(dy + o, ok);
            oa[k] = ldraw(dx_in + o, ok && dx_in != nullptr);
        }
    };
    auto compute_row = [&](int r, const RX (&xa)[NV], const RY (&da)[NV], const RD (&oa)[NV], float mu, float rs) {
        if (r >= M) return;   // (wave-uniform)
        float4 xh[NV], dh[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = (k * 64 + lane) * 4;
            const float4 ww = (FULL || c < H) ? *(const float4*)(w + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 xv = widen4(xa[k]), dv = widen4(da[k]);   // (columns past H: dv = 0, so nothing below accumulates there)
            xh[k] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
            dh[k] = make_float4(dv.x * ww.x, dv.y * ww.y, dv.z * ww.z, dv.w * ww.w);
            s1 += dh[k].x + dh[k].y + dh[k].z + dh[k].w;
            s2 += dh[k].x * xh[k].x + dh[k].y * xh[k].y + dh[k].z * xh[k].z + dh[k].w * xh[k].w;
            gw[k].x += dv.x * xh[k].x; gw[k].y += dv.y * xh[k].y; gw[k].z += dv.z * xh[k].z; gw[k].w += dv.w * xh[k].w;
            gb[k].x += dv.x; gb[k].y += dv.y; gb[k].z += dv.z; gb[k].w += dv.w;
        }
        s1 = wave_sum(s1) * invH;
        s2 = wave_sum(s2) * invH;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = (k * 64 + lane) * 4;
            if (FULL || c < H) {
                const float4 od = widen4(oa[k]);
                const float4 o = make_float4(od.x + rs * (dh[k].x - s1 - xh[k].x * s2), od.y + rs * (dh[k].y - s1 - xh[k].y * s2),
                                             od.z + rs * (dh[k].z - s1 - xh[k].z * s2), od.w + rs * (dh[k].w - s1 - xh[k].w * s2));
                st4f(dx_out + (long long)r * H + c, o);
                if (want_cast) {  // fused "dropcast" of the updated gradient stream for the next GEMMs (+ bias-grad sums)
                    const long long i = (long long)r * H + c;   // dropout stream index: the unpadded position
                    float v[4] = {o.x, o.y, o.z, o.w};
                    if (cz.thr) drop4(cz.key, cz.thr, (unsigned)i, cz.scale, v);
                    if (cz.out) {
                        // optional row remap: groups of rpg rows land gpitch rows apart, goff rows in (zero-padded sample blocks)
                        const long long io = cz.rpg > 0 ? ((long long)(r / cz.rpg) * cz.gpitch + cz.goff + r % cz.rpg) * H + c : i;
                        if (cz.bf16) { bf16x4 ov = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])}; *(bf16x4*)((bf16_t*)cz.out + io) = ov; }
                        else *(float4*)((float*)cz.out + io) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                    if (cz.nskip == 0 || (r % cz.rpg) >= cz.nskip) {   // (prefix-token rows do not feed the consumer's bias gradient)
                        gc[k].x += v[0]; gc[k].y += v[1]; gc[k].z += v[2]; gc[k].w += v[3];
                    }
                }
            }
        }
    };

    {
        RX xa[NV], xb[NV]; RY da[NV], db2[NV]; RD oa[NV], ob[NV];
        float mua, rsa, mub, rsb;
        load_row(row0, xa, da, oa, mua, rsa);
#pragma unroll 1
        for (int it = 0; it < iters; it += 2) {
            const int r = row0 + it * 4;
            const bool more = it + 1 < iters;
            if (more) load_row(r + 4, xb, db2, ob, mub, rsb);
            compute_row(r, xa, da, oa, mua, rsa);
            if (more) {
                if (it + 2 < iters) load_row(r + 8, xa, da, oa, mua, rsa);
                compute_row(r + 4, xb, db2, ob, mub, rsb);
            }
        }
    }
    // cross-wave reduce through LDS, then thread t owns columns t, t+256, ...: one CONTIGUOUS 256-B
    // atomic wave-instruction per 64 columns (float atomics run at full rate only in that shape)
    if (dbg_notail) return;   // (measurement builds only: what the tail costs; wrong column sums)
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        *(float4*)(red + (wv * 3 + 0) * W + (k * 64 + lane) * 4) = gw[k];
        *(float4*)(red + (wv * 3 + 1) * W + (k * 64 + lane) * 4) = gb[k];
        *(float4*)(red + (wv * 3 + 2) * W + (k * 64 + lane) * 4) = gc[k];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += 256) {
        float a = 0.f, b2 = 0.f, c2 = 0.f;
#pragma unroll
        for (int o = 0; o < 4; ++o) { a += red[(o * 3 + 0) * W + c]; b2 += red[(o * 3 + 1) * W + c]; c2 += red[(o * 3 + 2) * W + c]; }
        atomicAdd(rep_ptr(dw, rc, blockIdx.x) + c, a);
        atomicAdd(rep_ptr(db, rc, blockIdx.x) + c, b2);
        if (cz.colsum) atomicAdd(rep_ptr(cz.colsum, rc, blockIdx.x) + c, c2);
    }
}

template <typename DYT, typename TX, typename TD>
static void ln_bwd_dispatch(int nv, dim3 g, hipStream_t s, const void* dy, const void* x, const float* w, const float* mean, const float* rstd,
                            const void* dx_in, void* dx_out, float* dw, float* db, int M, int H, RepCfg rc, LnCast cz, int iters, int notail) {
#define LNB(NVV)                                                                                                                        \
    do {                                                                                                                                \
        if (H == NVV * 256)                                                                                                             \
            hipLaunchKernelGGL((ln_bwd_kernel<NVV, DYT, TX, TD, true>), g, dim3(256), 4 * 3 * NVV * 256 * sizeof(float), s, (const DYT*)dy, \
                               (const TX*)x, w, mean, rstd, (const TD*)dx_in, (TD*)dx_out, dw, db, M, H, rc, cz, iters, notail);        \
        else                                                                                                                            \
            hipLaunchKernelGGL((ln_bwd_kernel<NVV, DYT, TX, TD, false>), g, dim3(256), 4 * 3 * NVV * 256 * sizeof(float), s, (const DYT*)dy, \
                               (const TX*)x, w, mean, rstd, (const TD*)dx_in, (TD*)dx_out, dw, db, M, H, rc, cz, iters, notail);        \
    } while (0)
    if (nv <= 1) LNB(1);
    else if (nv == 3) LNB(3);   // (hidden 768, the iTransformer's: three exact column groups instead of four with one masked off)
    else if (nv <= 4) LNB(4);
    else LNB(8);
#undef LNB
}

int layernorm_bwd_launch(const void* dy, int dy_bf16, const void* x, const float* w, const float* mean, const float* rstd, LnStreams st,
                         float* dw, float* db, int M, int H, hipStream_t s, RepCfg rc, LnCast cz) {
    NBCI_REQUIRE(H % 4 == 0 && H <= 2048, NBCI_ESHAPE, "layernorm backward: hidden must be a multiple of 4 and <= 2048");
    NBCI_REQUIRE(st.dx_out, NBCI_EINVAL, "layernorm backward: null dx");
    NBCI_REQUIRE((st.x_bf16 != 0) == (st.dx_bf16 != 0), NBCI_EINVAL, "layernorm backward: a bf16 residual stream goes with a bf16 gradient stream");
    const double xs = st.x_bf16 ? 2 : 4, ds = st.dx_bf16 ? 2 : 4;
    ProfScope ps("ln_bwd_kernel", 0.0, (double)M * H * ((dy_bf16 ? 2 : 4) + xs + ds + (st.dx_in ? ds : 0) + (cz.out ? (cz.bf16 ? 2 : 4) : 0)), s);
    const int nv = (H + 255) / 256;
    static const int notail = measure_env("NBCI_LNB_NOTAIL", 0);
    static const int force_it = measure_env("NBCI_LNB_ITERS", 0);
    const int iters = force_it > 0 ? force_it : lnb_iters(M), rows = 4 * iters;
    dim3 g((M + rows - 1) / rows);
    if (st.x_bf16) {
        if (dy_bf16) ln_bwd_dispatch<bf16_t, bf16_t, bf16_t>(nv, g, s, dy, x, w, mean, rstd, st.dx_in, st.dx_out, dw, db, M, H, rc, cz, iters, notail);
        else ln_bwd_dispatch<float, bf16_t, bf16_t>(nv, g, s, dy, x, w, mean, rstd, st.dx_in, st.dx_out, dw, db, M, H, rc, cz, iters, notail);
    } else {
        if (dy_bf16) ln_bwd_dispatch<bf16_t, float, float>(nv, g, s, dy, x, w, mean, rstd, st.dx_in, st.dx_out, dw, db, M, H, rc, cz, iters, notail);
        else ln_bwd_dispatch<float, float, float>(nv, g, s, dy, x, w, mean, rstd, st.dx_in, st.dx_out, dw, db, M, H, rc, cz, iters, notail);
    }
    return check_launch("layernorm_bwd");
}
int layernorm_bwd_launch(const void* dy, const float* x, const float* w, const float* mean, const float* rstd,
                         float* dx, float* dw, float* db, int M, int H, int accumulate_dx, hipStream_t s, RepCfg rc, LnCast cz, int dy_bf16) {
    return layernorm_bwd_launch(dy, dy_bf16, x, w, mean, rstd, LnStreams{0, accumulate_dx ? dx : nullptr, dx, 0}, dw, db, M, H, s, rc, cz);
}

// ------------------------------------------------------------------------------------------
// masked softmax (+ attention-prob dropout) forward / backward, one wave per (b, head, query) row
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool ctx_allowed(int i, int j, int f, int bk) {
    // create_context_mask (ndt1.py:30-41): allowed iff j - i <= f' and i - j <= b' (-2 => unbounded)
    if (f >= -1 && j - i > f) return false;
    if (bk >= -1 && i - j > bk) return false;
    return true;
}

template <int NV, typename TP>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ S, TP* __restrict__ P,
                                                          TP* __restrict__ Pd, const int32_t* __restrict__ tmask,
                                                          int rows, int nh, int Tp, int ldS, int ldP, int cf, int cb,
                                                          unsigned thr, float dscale, uint32_t key) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int i = row % Tp, bh = row / Tp, b = bh / nh;
    const float* sr = S + (long long)row * ldS;
    float v[NV];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int j = k * 64 + lane;
        float val = -INFINITY;
        if (j < Tp) {
            const bool ok = (j == i) || (ctx_allowed(i, j, cf, cb) && (!tmask || tmask[b * Tp + j] != 0));
            if (ok) val = sr[j];
        }
        v[k] = val;
        mx = fmaxf(mx, val);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        v[k] = (v[k] == -INFINITY) ? 0.f : expf(v[k] - mx);
        sum += v[k];
    }
    const float inv = 1.0f / wave_sum(sum);
    TP* pr = P + (long long)row * ldP;
    TP* pdr = Pd + (long long)row * ldP;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int j = k * 64 + lane;
        if (j < ldP) {
            float p = (j < Tp) ? v[k] * inv : 0.f;
            stf<TP>(pr, j, p);
            if (thr && j < Tp) {
                const unsigned idx = (unsigned)((long long)row * Tp + j);
                p = drop_keep(key, thr, idx) ? p * dscale : 0.f;
            }
            stf<TP>(pdr, j, p);
        }
    }
}

int softmax_fwd_launch(const float* S, void* P, void* Pd, int p_dtype, const int32_t* tmask, int B, int nh, int Tp,
                       int ldS, int ldP, int ctx_fwd, int ctx_bwd, float drop_p, uint32_t seed, uint32_t site,
                       hipStream_t s) {
    NBCI_REQUIRE(ldP <= 2048 && Tp <= ldP, NBCI_ESHAPE, "softmax: at most 2048 tokens");
    const int rows = B * nh * Tp;
    const unsigned thr = drop_threshold(drop_p);
    const float dscale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    dim3 g((rows + 3) / 4);
    DISPATCH_DTYPE(p_dtype, TP, {
        if (ldP <= 256)
            hipLaunchKernelGGL((softmax_fwd_kernel<4, TP>), g, dim3(256), 0, s, S, (TP*)P, (TP*)Pd, tmask, rows, nh, Tp,
                               ldS, ldP, ctx_fwd, ctx_bwd, thr, dscale, drop_key(seed, site));
        else if (ldP <= 1024)
            hipLaunchKernelGGL((softmax_fwd_kernel<16, TP>), g, dim3(256), 0, s, S, (TP*)P, (TP*)Pd, tmask, rows, nh, Tp,
                               ldS, ldP, ctx_fwd, ctx_bwd, thr, dscale, drop_key(seed, site));
        else
            hipLaunchKernelGGL((softmax_fwd_kernel<32, TP>), g, dim3(256), 0, s, S, (TP*)P, (TP*)Pd, tmask, rows, nh, Tp,
                               ldS, ldP, ctx_fwd, ctx_bwd, thr, dscale, drop_key(seed, site));
    });
    return check_launch("softmax_fwd");
}

template <int NV, typename TP>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ dPd, const TP* __restrict__ P,
                                                          TP* __restrict__ dS, int rows, int Tp, int ldS, int ldP,
                                                          unsigned thr, float dscale, uint32_t key) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float p[NV], dp[NV];
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int j = k * 64 + lane;
        p[k] = 0.f; dp[k] = 0.f;
        if (j < Tp) {
            p[k] = ldf<TP>(P, (long long)row * ldP + j);
            float g = dPd[(long long)row * ldS + j];
            if (thr) {
                const unsigned idx = (unsigned)((long long)row * Tp + j);
                g = drop_keep(key, thr, idx) ? g * dscale : 0.f;
            }
            dp[k] = g;
            dot += g * p[k];
        }
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int j = k * 64 + lane;
        if (j < ldP) stf<TP>(dS, (long long)row * ldP + j, (j < Tp) ? p[k] * (dp[k] - dot) : 0.f);
    }
}

int softmax_bwd_launch(const float* dPd, const void* P, void* dS, int p_dtype, int B, int nh, int Tp, int ldS, int ldP,
                       float drop_p, uint32_t seed, uint32_t site, hipStream_t s) {
    NBCI_REQUIRE(ldP <= 2048 && Tp <= ldP, NBCI_ESHAPE, "softmax: at most 2048 tokens");
    const int rows = B * nh * Tp;
    const unsigned thr = drop_threshold(drop_p);
    const float dscale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    dim3 g((rows + 3) / 4);
    DISPATCH_DTYPE(p_dtype, TP, {
        if (ldP <= 256)
            hipLaunchKernelGGL((softmax_bwd_kernel<4, TP>), g, dim3(256), 0, s, dPd, (const TP*)P, (TP*)dS, rows, Tp, ldS,
                               ldP, thr, dscale, drop_key(seed, site));
        else if (ldP <= 1024)
            hipLaunchKernelGGL((softmax_bwd_kernel<16, TP>), g, dim3(256), 0, s, dPd, (const TP*)P, (TP*)dS, rows, Tp, ldS,
                               ldP, thr, dscale, drop_key(seed, site));
        else
            hipLaunchKernelGGL((softmax_bwd_kernel<32, TP>), g, dim3(256), 0, s, dPd, (const TP*)P, (TP*)dS, rows, Tp, ldS,
                               ldP, thr, dscale, drop_key(seed, site));
    });
    return check_launch("softmax_bwd");
}

// ------------------------------------------------------------------------------------------
// dropout-cast, cast, colsum
// ------------------------------------------------------------------------------------------
constexpr int DC_ROWS = 8;  // rows per block: all 8 row loads of a thread are in flight together

// out = in * keepmask (act dtype), optionally colsum[n] += sum_m out[m][n]. One thread owns 4
// consecutive columns of DC_ROWS rows, so the bias-gradient column sums cost one atomic per column per block.
template <typename TO, typename TI>
__global__ __launch_bounds__(256) void dropcast_kernel(const TI* __restrict__ in, TO* __restrict__ out, int M, int N,
                                                       unsigned thr, float dscale, uint32_t key, float* __restrict__ colsum, RepCfg rc,
                                                       int cpt, int gpb) {
    // cpt = column-threads per row (N / 4, at most 256 per block), gpb = row groups per block (narrow matrices keep all
    // 256 threads busy); a block walks row groups with a grid stride so the column sums cost one atomic per thread.
    __shared__ float csum[1024];   // [gpb][cpt * 4]: the row groups' column sums (colsum only)
    const int cg = threadIdx.x % cpt, rg = threadIdx.x / cpt;
    const int c = (blockIdx.x * cpt + cg) * 4;
    const bool live = c < N && rg < gpb;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (live)
    for (long long r0 = ((long long)blockIdx.y * gpb + rg) * DC_ROWS; r0 < M; r0 += (long long)gridDim.y * gpb * DC_ROWS) {
        float4 q[DC_ROWS];
#pragma unroll
        for (int j = 0; j < DC_ROWS; ++j)
            q[j] = (r0 + j < M) ? ld4f(in + (r0 + j) * N + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < DC_ROWS; ++j) {
            if (r0 + j >= M) break;
            const long long i = (r0 + j) * N + c;
            float v[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
            if (thr) drop4(key, thr, (unsigned)i, dscale, v);
            if constexpr (sizeof(TO) == 2) {
                bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
                *(bf16x4*)((bf16_t*)out + i) = o;
            } else {
                *(float4*)((float*)out + i) = make_float4(v[0], v[1], v[2], v[3]);
            }
            s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
        }
    }
    if (colsum) {
        // the block's row groups are summed through LDS and ONE lane-contiguous atomic per column leaves the block (a thread adding its
        // four adjacent columns made every atomic wave-instruction touch 64 words spread over 1 KB, and every row group added its own)
        const int ncol = cpt * 4;
        if (live) *(float4*)(csum + rg * ncol + cg * 4) = make_float4(s[0], s[1], s[2], s[3]);
        __syncthreads();
        for (int i = threadIdx.x; i < ncol; i += 256) {
            const int col = blockIdx.x * ncol + i;
            if (col >= N) break;
            float t = 0.f;
            for (int g2 = 0; g2 < gpb; ++g2) t += csum[g2 * ncol + i];
            atomicAdd(rep_ptr(colsum, rc, blockIdx.y) + col, t);
        }
    }
}

int dropcast2d_launch(const void* in, void* out, int out_dtype, int M, int N, float drop_p, uint32_t seed, uint32_t site,
                      float* colsum, hipStream_t s, RepCfg rc, int in_dtype) {
    NBCI_REQUIRE(N % 4 == 0, NBCI_ESHAPE, "dropcast: N must be a multiple of 4");
    const unsigned thr = drop_threshold(drop_p);
    const float dscale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    const int cpt = std::min(256, N / 4), gpb = std::max(1, 256 / cpt);
    const long long groups = ((long long)M + DC_ROWS - 1) / DC_ROWS;
    dim3 g((N / 4 + cpt - 1) / cpt, (unsigned)std::min<long long>((groups + gpb - 1) / gpb, 4096));
    if (in_dtype == NBCI_BF16) {
        DISPATCH_DTYPE(out_dtype, TO,
                       hipLaunchKernelGGL((dropcast_kernel<TO, bf16_t>), g, dim3(256), 0, s, (const bf16_t*)in, (TO*)out, M, N, thr, dscale,
                                          drop_key(seed, site), colsum, rc, cpt, gpb));
    } else {
        DISPATCH_DTYPE(out_dtype, TO,
                       hipLaunchKernelGGL((dropcast_kernel<TO, float>), g, dim3(256), 0, s, (const float*)in, (TO*)out, M, N, thr, dscale,
                                          drop_key(seed, site), colsum, rc, cpt, gpb));
    }
    return check_launch("dropcast");
}

int dropcast_launch(const float* in, void* out, int out_dtype, int64_t n, float drop_p, uint32_t seed, uint32_t site,
                    hipStream_t s) {
    // flat form: treat as rows of 1024 (or one row when short / not divisible)
    const int N = (n % 1024 == 0) ? 1024 : (int)n;
    NBCI_REQUIRE(n % 4 == 0 && n / N < 2147483647LL, NBCI_ESHAPE, "dropcast: length must be a multiple of 4");
    return dropcast2d_launch(in, out, out_dtype, (int)(n / N), N, drop_p, seed, site, nullptr, s, RepCfg{0, 1});
}

int cast_launch(const float* in, void* out, int out_dtype, int64_t n, hipStream_t s) {
    return dropcast_launch(in, out, out_dtype, n, 0.f, 0, 0, s);
}

constexpr int CS_ROWS = 128;  // rows per block (4 row-threads x 32 rows each, 16-byte loads)

// out[n] += sum_m in[m][n]. Block = 64 column-threads x 4 row-threads; a thread owns E = 16 B / sizeof(T)
// consecutive columns. Falls back to scalar columns when the row pitch is not 16-byte aligned.
template <typename TI, bool VEC>
__global__ __launch_bounds__(256) void colsum_kernel(const TI* __restrict__ in, long long ld, int M, int N,
                                                     float* __restrict__ out, RepCfg rc) {
    constexpr int E = VEC ? (int)(16 / sizeof(TI)) : 1;
    __shared__ float red[4][64 * E];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + cx) * E;
    const int r0 = blockIdx.y * CS_ROWS;
    float acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
    if (c < N) {
        const int rend = min(M, r0 + CS_ROWS);
        if constexpr (VEC) {
#pragma unroll 4
            for (int r = r0 + ry; r < rend; r += 4) {
                const uint4 q = *(const uint4*)(in + (long long)r * ld + c);
                const TI* qe = (const TI*)&q;
#pragma unroll
                for (int e = 0; e < E; ++e) acc[e] += (c + e < N) ? (float)qe[e] : 0.f;
            }
        } else {
            for (int r = r0 + ry; r < rend; r += 4) acc[0] += ldf<TI>(in, (long long)r * ld + c);
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) red[ry][cx * E + e] = acc[e];
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * E; i += 256) {
        const int col = blockIdx.x * 64 * E + i;
        if (col < N) atomicAdd(rep_ptr(out, rc, blockIdx.y) + col, red[0][i] + red[1][i] + red[2][i] + red[3][i]);
    }
}

int colsum_launch(const void* in, int in_dtype, int64_t ld, int M, int N, float* out, hipStream_t s, RepCfg rc) {
    const int E = in_dtype == NBCI_BF16 ? 8 : 4;
    const bool vec = (ld % E == 0) && (((uintptr_t)in) % 16 == 0);
    dim3 g(vec ? (N + 64 * E - 1) / (64 * E) : (N + 63) / 64, (M + CS_ROWS - 1) / CS_ROWS);
    DISPATCH_DTYPE(in_dtype, TI, {
        if (vec) hipLaunchKernelGGL((colsum_kernel<TI, true>), g, dim3(256), 0, s, (const TI*)in, (long long)ld, M, N, out, rc);
        else hipLaunchKernelGGL((colsum_kernel<TI, false>), g, dim3(256), 0, s, (const TI*)in, (long long)ld, M, N, out, rc);
    });
    return check_launch("colsum");
}

// ------------------------------------------------------------------------------------------
// stack backward: col2im over the overlapping windows fused with the embed activation gradient
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void col2im_actgrad_kernel(const T* __restrict__ dwin, const T* __restrict__ y,
                                                             T* __restrict__ dpre, int B, int Tt, int Tp, int D, int size,
                                                             int stride, int act) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)B * Tt * D;
    if (i >= total) return;
    const int c = (int)(i % D);
    const int t = (int)((i / D) % Tt);
    const int b = (int)(i / ((long long)D * Tt));
    int jmax = t / stride;
    if (jmax > Tp - 1) jmax = Tp - 1;
    int jmin = (t - size + 1 + stride - 1);
    jmin = jmin > 0 ? jmin / stride : 0;
    float acc = 0.f;
    for (int j = jmin; j <= jmax; ++j) {
        const int r = t - stride * j;  // 0 <= r < size
        acc += ldf<T>(dwin, ((long long)b * Tp + j) * ((long long)size * D) + (long long)r * D + c);
    }
    stf<T>(dpre, i, acc * act_bwd_from_output(act, ldf<T>(y, i)));
}

int col2im_actgrad_launch(const void* dwin, const void* y, void* dpre, int dtype, int B, int T, int Tp, int D, int size,
                          int stride, int act, hipStream_t s) {
    NBCI_REQUIRE(act != ACT_GELU, NBCI_EINVAL, "embed activation gelu is not supported (needs the pre-activation)");
    const long long total = (long long)B * T * D;
    DISPATCH_DTYPE(dtype, TT,
                   hipLaunchKernelGGL((col2im_actgrad_kernel<TT>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                                      (const TT*)dwin, (const TT*)y, (TT*)dpre, B, T, Tp, D, size, stride, act));
    return check_launch("col2im_actgrad");
}

// nn.Embedding backward of the position table + embed dropout: dpos[tts[b][j]] += keepmask * dx[b][npre + j].
// One thread per (token position j, column) walks the batch: with the usual arange timestamps every sample hits the SAME table
// row at position j, so the run is summed in a register and leaves as ONE atomic (f32 atomics run at ~1.3 TB/s chip-wide: one per
// element was 30 us of a 32 us kernel); a differing timestamp just flushes the run.
template <typename T>   // T: the gradient stream's storage type (f32, or bf16)
__global__ __launch_bounds__(256) void posgrad_kernel(const T* __restrict__ dx, const int64_t* __restrict__ tts,
                                                      float* __restrict__ dpos, int B, int H, unsigned thr, float dscale,
                                                      uint32_t key, int Tp, int npre) {
    const int c = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (c >= H) return;
    long long cur = -1;
    float acc = 0.f;
    const int bper = (B + (int)gridDim.z - 1) / (int)gridDim.z, bend = min(B, ((int)blockIdx.z + 1) * bper);   // this block's slice of the batch
    for (int b0 = blockIdx.z * bper; b0 < bend; b0 += 8) {   // eight samples' loads in flight together (the run logic below is a serial chain)
        float v[8];
        long long t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int b = b0 + k < bend ? b0 + k : bend - 1;
            const long long o = ((long long)b * (Tp + npre) + npre + j) * H + c;
            v[k] = ldf<T>(dx, o);
            if (thr) v[k] = drop_keep(key, thr, (unsigned)o) ? v[k] * dscale : 0.f;
            t[k] = tts[(long long)b * Tp + j];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (b0 + k >= bend) break;
            if (t[k] != cur) {
                if (acc != 0.f) atomicAdd(dpos + cur * H + c, acc);
                cur = t[k]; acc = 0.f;
            }
            acc += v[k];
        }
    }
    if (acc != 0.f) atomicAdd(dpos + cur * H + c, acc);
}

int posgrad_launch(const void* dx, const int64_t* tts, float* dpos, int M, int H, float drop_p, uint32_t seed,
                   uint32_t site, hipStream_t s, int Tp, int npre, int dx_dtype) {
    const unsigned thr = drop_threshold(drop_p);
    const float dscale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    if (Tp <= 0) Tp = M;   // (callers without a batch structure: one "sample" of M positions)
    NBCI_REQUIRE(M % Tp == 0, NBCI_ESHAPE, "posgrad: rows must be a whole number of samples");
    const int Bn = M / Tp, zs = Bn >= 32 ? 4 : (Bn >= 16 ? 2 : 1);   // batch slices: enough loads in flight, still 16 x fewer atomics at B = 64
    DISPATCH_DTYPE(dx_dtype, T, hipLaunchKernelGGL((posgrad_kernel<T>), dim3((unsigned)((H + 255) / 256), (unsigned)Tp, (unsigned)zs), dim3(256), 0, s,
                                                   (const T*)dx, tts, dpos, Bn, H, thr, dscale, drop_key(seed, site), Tp, npre));
    return check_launch("posgrad");
}

// ------------------------------------------------------------------------------------------
// RoPE (optional; off in configs/ndt1.yaml:60)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rope_kernel(T* __restrict__ qkv, const int64_t* __restrict__ tts,
                                                   const float* __restrict__ cos_t, const float* __restrict__ sin_t, int M,
                                                   int H, int nh, int inverse) {
    const int hd = H / nh, half = hd / 2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)M * 2 * nh * half;  // q and k thirds
    if (i >= total) return;
    const int d = (int)(i % half);
    const int h = (int)((i / half) % nh);
    const int which = (int)((i / ((long long)half * nh)) % 2);
    const int row = (int)(i / ((long long)half * nh * 2));
    const long long pos = tts[row];
    const float c1 = cos_t[pos * hd + d], s1 = sin_t[pos * hd + d];
    const float c2 = cos_t[pos * hd + d + half], s2 = sin_t[pos * hd + d + half];
    const long long o = (long long)row * 3 * H + (long long)which * H + h * hd + d;
    const float x1 = ldf<T>(qkv, o), x2 = ldf<T>(qkv, o + half);
    float o1, o2;
    if (!inverse) { o1 = x1 * c1 - x2 * s1; o2 = x2 * c2 + x1 * s2; }
    else { o1 = x1 * c1 + x2 * s2; o2 = x2 * c2 - x1 * s1; }
    stf<T>(qkv, o, o1);
    stf<T>(qkv, o + half, o2);
}

int rope_launch(void* qkv, int dtype, const int64_t* tts, const float* cos_t, const float* sin_t, int M, int H, int nh,
                int inverse, hipStream_t s) {
    const long long total = (long long)M * nh * (H / nh);
    DISPATCH_DTYPE(dtype, TT,
                   hipLaunchKernelGGL((rope_kernel<TT>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (TT*)qkv,
                                      tts, cos_t, sin_t, M, H, nh, inverse));
    return check_launch("rope");
}

// ------------------------------------------------------------------------------------------
// head: log-softmax + argmax; CTC; greedy decode + edit distance
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void logsoftmax_kernel(const float* __restrict__ logits, int ldl,
                                                         float* __restrict__ preds, int32_t* __restrict__ argmax, int M,
                                                         int V) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* lr = logits + (long long)row * ldl;
    float mx = -INFINITY;
    int am = 0x7fffffff;
    for (int j = lane; j < V; j += 64) {
        const float v = lr[j];
        if (v > mx) { mx = v; am = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(mx, o, 64);
        const int oa = __shfl_xor(am, o, 64);
        if (om > mx || (om == mx && oa < am)) { mx = om; am = oa; }
    }
    float sum = 0.f;
    for (int j = lane; j < V; j += 64) sum += expf(lr[j] - mx);
    const float lse = logf(wave_sum(sum));
    for (int j = lane; j < V; j += 64) preds[(long long)row * V + j] = (lr[j] - mx) - lse;
    if (argmax && lane == 0) argmax[row] = am;
}

int logsoftmax_launch(const float* logits, int ldl, float* preds, int32_t* argmax, int M, int V, hipStream_t s) {
    hipLaunchKernelGGL(logsoftmax_kernel, dim3((M + 3) / 4), dim3(256), 0, s, logits, ldl, preds, argmax, M, V);
    return check_launch("logsoftmax");
}

__device__ __forceinline__ float log_add3(float a, float b, float c) {
    const float m = fmaxf(fmaxf(a, b), c);
    if (m == -INFINITY) return -INFINITY;
    // hardware exp2 / log2 (1 ulp): this sits on the serial chain of the frame loop; the loss stays within 2e-6 relative
    // of the libm version over 143 frames (golden CTC cases: rtol 2e-5)
    return m + __logf(__expf(a - m) + __expf(b - m) + __expf(c - m));
}

size_t ctc_alpha_floats(int B, int Tp, int S) { return 2 * (size_t)B * Tp * (2 * (size_t)S + 1); }  // alpha + beta

// One workgroup per sample. The alpha recursion (threads 0..127) and the beta recursion (threads
// 128..255) advance together, one frame per barrier, each keeping its running row in LDS and
// writing the full lattice to the HBM workspace; the occupancy sums and the gradient are then a
// fully parallel pass over (frame, state).
// The frame loop is a chain of Tb dependent steps, so nothing slow may sit inside it: the sample's
// log-probabilities are staged in LDS once (STAGED; a global read per step costs an L2 round trip), and the step
// barrier is a raw s_barrier behind lgkmcnt(0) only — __syncthreads() would also wait for the lattice stores
// (vmcnt), which nothing reads before the pass after the loop.
template <typename TD, bool STAGED>
__global__ __launch_bounds__(256) void ctc_kernel(const float* __restrict__ preds, const int64_t* __restrict__ targets,
                                                  const int32_t* __restrict__ in_lens, const int64_t* __restrict__ tgt_lens,
                                                  int Tp, int V, int S, int blank, int zero_inf, float* __restrict__ loss,
                                                  float* __restrict__ ws, TD* __restrict__ dlogits, int ldd,
                                                  float grad_scale, int Bn, int chunk) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Lmax = 2 * S + 1;
    float* rowA = sm;                       // [2][Lmax] alpha double buffer
    float* rowB = sm + 2 * Lmax;            // [2][Lmax] beta double buffer
    int* ext = (int*)(sm + 4 * Lmax);       // [Lmax]
    float* occ = sm + 5 * Lmax;             // [chunk][V] (gradient pass; chunk = Tp when the whole sample fits)
    float* lpl = occ + chunk * V;           // [Tp][V] staged log-probabilities (STAGED)
    __shared__ float s_nll;
    const int b = blockIdx.x, tid = threadIdx.x;
    int Tb = in_lens[b];
    if (Tb > Tp) Tb = Tp;
    if (Tb < 0) Tb = 0;
    int Sb = (int)tgt_lens[b];
    if (Sb > S) Sb = S;
    const int L = 2 * Sb + 1;
    const float* lpg = preds + (long long)b * Tp * V;
    if constexpr (STAGED) {   // eight loads in flight per thread: the plain copy loop waited for each load before its LDS store (~23 round trips)
        const int n = Tb * V;
        for (int base = 0; base < n; base += 256 * 8) {
            float r[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int i = base + k * 256 + tid; r[k] = i < n ? lpg[i] : 0.f; }
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int i = base + k * 256 + tid; if (i < n) lpl[i] = r[k]; }
        }
    }
    const float* lp = STAGED ? lpl : lpg;
    float* aw = ws + (long long)b * Tp * Lmax;
    float* bw = ws + ((long long)Bn + b) * Tp * Lmax;
    for (int s = tid; s < L; s += 256) ext[s] = (s & 1) ? (int)targets[(long long)b * S + (s >> 1)] : blank;
    __syncthreads();
    const bool is_alpha = tid < 128;
    const int ht = tid & 127;
    // Usual case (2S + 1 <= 128): the whole lattice row lives in ONE wave's registers, two states per lane (lane i: blank state 2i and
    // label state 2i + 1), alpha in wave 0 and beta in wave 1. A frame step is then register arithmetic plus one (alpha) or two (beta)
    // DPP wave shifts for the neighbour lane's states — no LDS row, no barrier, nothing to wait for but the log-add itself (the previous
    // version: one state per thread over two waves each, an LDS round trip and a four-wave barrier per frame, ~0.4 us per frame).
    // The frame's two log-probabilities are read one step ahead; lattice rows go to the workspace with fire-and-forget stores.
    if (L <= 128) {
        const int wv = tid >> 6, i = tid & 63;
        if (wv < 2) {
            const bool alpha = wv == 0;
            const bool live0 = 2 * i < L, live1 = 2 * i + 1 < L;
            const int e = live1 ? ext[2 * i + 1] : blank;
            // skip transitions: alpha into state 2i+1 from 2i-1 ; beta into state 2i+1 from 2i+3
            const bool skipa = live1 && i >= 1 && e != blank && e != ext[2 * i - 1];
            const bool skipb = live1 && 2 * i + 3 < L && ext[2 * i + 3] != blank && ext[2 * i + 3] != e;
            // log-add on the raw hardware exp2 / log2 (1 ulp; the sum lies in [1, 3], so no denormal fix-ups are needed), branch-free:
            // with every term -inf the floor keeps (x - m) = -inf, the sum is 0 and log2 returns -inf
            constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
            auto lse2 = [&](float x, float y) {
                const float m = fmaxf(fmaxf(x, y), -3.0e38f);
                return m + LN2 * __builtin_amdgcn_logf(__builtin_amdgcn_exp2f((x - m) * LOG2E) + __builtin_amdgcn_exp2f((y - m) * LOG2E));
            };
            auto lse3 = [&](float x, float y, float z) {
                const float m = fmaxf(fmaxf(fmaxf(x, y), z), -3.0e38f);
                return m + LN2 * __builtin_amdgcn_logf(__builtin_amdgcn_exp2f((x - m) * LOG2E) + __builtin_amdgcn_exp2f((y - m) * LOG2E) +
                                                       __builtin_amdgcn_exp2f((z - m) * LOG2E));
            };
            // States past the sample's lattice (s >= L) need no masking: beta's only ever see -inf (nothing above them is initialised),
            // alpha's turn finite but feed nothing below them; they are simply not stored.
            const int t0 = alpha ? 0 : Tb - 1, dt = alpha ? 1 : -1;
            const float* pbp = lp + blank + (long long)t0 * V;   // this frame's log-probabilities (blank / the lane's label), stepped by dt * V
            const float* pep = lp + e + (long long)t0 * V;
            float* latp = (alpha ? aw : bw) + (long long)t0 * Lmax + 2 * i;
            const int dV = dt * V, dL = dt * Lmax;
            const int ninf = __builtin_bit_cast(int, -INFINITY);
            float v0 = -INFINITY, v1 = -INFINITY;
            float pb = Tb > 0 ? *pbp : 0.f, pe = Tb > 0 ? *pep : 0.f;
            for (int step = 0; step < Tb; ++step) {
                const int adv = (step + 1 < Tb) ? 1 : 0;
                const float nb = pbp[adv * dV], ne = pep[adv * dV];   // next frame's log-probabilities, requested a step ahead
                if (step == 0) {
                    if (alpha) { v0 = (i == 0) ? pb : -INFINITY; v1 = (i == 0 && live1) ? pe : -INFINITY; }
                    else { v0 = (2 * i == L - 1) ? pb : -INFINITY; v1 = (live1 && 2 * i + 1 == L - 2) ? pe : -INFINITY; }
                } else if (alpha) {
                    const float pl = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(ninf, __builtin_bit_cast(int, v1), 0x138, 0xF, 0xF, false));   // wave_shr:1: state 2i-1
                    const float n0 = lse2(v0, pl) + pb;
                    v1 = lse3(v1, v0, skipa ? pl : -INFINITY) + pe;
                    v0 = n0;
                } else {
                    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(ninf, __builtin_bit_cast(int, v0), 0x130, 0xF, 0xF, false));   // wave_shl:1: state 2i+2
                    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(ninf, __builtin_bit_cast(int, v1), 0x130, 0xF, 0xF, false));   // state 2i+3
                    v0 = lse2(v0, v1) + pb;
                    v1 = lse3(v1, r0, skipb ? r1 : -INFINITY) + pe;
                }
                if (live0) latp[0] = v0;
                if (live1) latp[1] = v1;
                pb = nb; pe = ne;
                pbp += adv * dV; pep += adv * dV; latp += dL;
            }
            if (alpha && Tb > 0) {   // the last row, where the tail below expects it
                float* last = rowA + ((Tb - 1) & 1) * Lmax;
                if (live0) last[2 * i] = v0;
                if (live1) last[2 * i + 1] = v1;
            }
        }
    } else
    for (int step = 0; step < Tb; ++step) {
        if (is_alpha) {
            const int t = step;
            const float* prev = rowA + ((t + 1) & 1) * Lmax;
            float* cur = rowA + (t & 1) * Lmax;
            for (int s = ht; s < L; s += 128) {
                float v;
                if (t == 0) {
                    v = (s < 2) ? lp[ext[s]] : -INFINITY;
                } else {
                    const float a0 = prev[s];
                    const float a1 = s >= 1 ? prev[s - 1] : -INFINITY;
                    const float a2 = (s >= 2 && ext[s] != blank && ext[s] != ext[s - 2]) ? prev[s - 2] : -INFINITY;
                    v = log_add3(a0, a1, a2) + lp[(long long)t * V + ext[s]];
                }
                cur[s] = v; aw[(long long)t * Lmax + s] = v;
            }
        } else {
            const int t = Tb - 1 - step;
            const float* prev = rowB + ((step + 1) & 1) * Lmax;
            float* cur = rowB + (step & 1) * Lmax;
            for (int s = ht; s < L; s += 128) {
                float v;
                if (step == 0) {
                    v = (s >= L - 2) ? lp[(long long)t * V + ext[s]] : -INFINITY;
                } else {
                    const float b0 = prev[s];
                    const float b1 = s + 1 < L ? prev[s + 1] : -INFINITY;
                    const float b2 = (s + 2 < L && ext[s + 2] != blank && ext[s + 2] != ext[s]) ? prev[s + 2] : -INFINITY;
                    v = log_add3(b0, b1, b2) + lp[(long long)t * V + ext[s]];
                }
                cur[s] = v; bw[(long long)t * Lmax + s] = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this thread's row writes have landed in LDS
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    __syncthreads();
    if (tid == 0) {
        float ll;
        if (Tb > 0) {
            const float* last = rowA + ((Tb - 1) & 1) * Lmax;
            const float a = last[L - 1], c = L > 1 ? last[L - 2] : -INFINITY;
            const float m = fmaxf(a, c);
            ll = (m == -INFINITY) ? -INFINITY : m + logf(expf(a - m) + expf(c - m));
        } else {
            ll = (Sb == 0) ? 0.f : -INFINITY;
        }
        s_nll = -ll;
    }
    __syncthreads();
    const float nll = s_nll;
    const bool finite = nll < INFINITY;
    if (tid == 0) loss[b] = finite ? nll : (zero_inf ? 0.f : INFINITY);
    if (!dlogits) return;
    // rows beyond the input length (and whole infeasible samples) get zero gradient
    const int tz = finite ? Tb : 0;
    for (long long i = tid + (long long)tz * ldd; i < (long long)Tp * ldd; i += 256) stf<TD>(dlogits, (long long)b * Tp * ldd + i, 0.f);
    if (!finite || Tb == 0) return;
    // ---- occupancy: occ[t][c] = sum_{s: ext[s] = c} exp(alpha + beta - lp + nll), all (t, s) in parallel; `chunk` frames per pass
    // (one pass unless the sample has more frames than fit the LDS: the reference allows up to max_F = 1024 tokens)
    for (int c0 = 0; c0 < Tb; c0 += chunk) {
        const int c1 = min(Tb, c0 + chunk);
        __syncthreads();   // the previous chunk's occ has been read; (first pass) this block's alpha / beta global writes are visible to its threads
        for (int i = tid; i < (c1 - c0) * V; i += 256) occ[i] = 0.f;
        __syncthreads();
        // wave w takes frames c0 + w, + 4, ...; lanes run along the states. Four frames' lattice rows are loaded together:
        // they were just written by this block and come back from L2 (~1 us each if taken one by one; eight together measured no faster).
        constexpr int NF = 4;
        for (int t0 = c0 + (tid >> 6); t0 < c1; t0 += 4 * NF) {
            for (int s = tid & 63; s < L; s += 64) {
                float ab[NF];
#pragma unroll
                for (int k = 0; k < NF; ++k) {
                    const int t = t0 + 4 * k;
                    ab[k] = t < c1 ? aw[(long long)t * Lmax + s] + bw[(long long)t * Lmax + s] : -INFINITY;
                }
                const int e = ext[s];
#pragma unroll
                for (int k = 0; k < NF; ++k) {
                    const int t = t0 + 4 * k;
                    if (ab[k] > -INFINITY) atomicAdd(&occ[(t - c0) * V + e], expf(ab[k] - lp[(long long)t * V + e] + nll));
                }
            }
        }
        __syncthreads();
        for (int i = tid + c0 * ldd; i < c1 * ldd; i += 256) {
            const int t = i / ldd, c = i % ldd;
            const float v = (c < V) ? (expf(lp[(long long)t * V + c]) - occ[(t - c0) * V + c]) * grad_scale : 0.f;
            stf<TD>(dlogits, ((long long)b * Tp + t) * ldd + c, v);
        }
    }
}

// ---- "meet in the middle" variant (2S + 1 <= 128 and everything fits the 160 KB LDS: the train step's shape) -----------------------
// The plain kernel above runs the alpha and beta recursions over ALL frames (30 us at 143 frames), writes both lattices to the
// workspace and then spends another 38 us reading them back for the occupancy sums. Here wave 0 runs alpha over the first half of
// the frames and wave 1 runs beta over the second half, both keeping their rows in LDS; after one barrier each wave continues into
// the half where the OTHER lattice is already known, so the posterior exp(alpha + beta - lp + nll) of a frame is formed in the same
// step that produces the frame's row — no lattice in HBM, no separate occupancy pass. The likelihood comes from the identity
// P = sum_s alpha_t(s) beta_t(s) / y_t(s) (any t) at the first frame each wave has both rows for.
template <typename TD>
__global__ __launch_bounds__(256) void ctc_mid_kernel(const float* __restrict__ preds, const int64_t* __restrict__ targets,
                                                      const int32_t* __restrict__ in_lens, const int64_t* __restrict__ tgt_lens,
                                                      int Tp, int V, int S, int blank, int zero_inf, float* __restrict__ loss,
                                                      TD* __restrict__ dlogits, int ldd, float grad_scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Lmax = 2 * S + 1, Hm = (Tp + 1) / 2;
    float* latA = sm;                      // [Hm][Lmax] alpha rows of frames [0, m)
    float* latB = latA + Hm * Lmax;        // [Hm][Lmax] beta rows of frames [m, Tb): row t - m
    float* occ = latB + Hm * Lmax;         // [Tp][V] posterior mass per (frame, class)
    float* lpl = occ + Tp * V;             // [Tp][V] staged log-probabilities
    int* ext = (int*)(lpl + Tp * V);       // [Lmax]
    __shared__ float s_nll;
    const int b = blockIdx.x, tid = threadIdx.x;
    int Tb = in_lens[b];
    if (Tb > Tp) Tb = Tp;
    if (Tb < 0) Tb = 0;
    int Sb = (int)tgt_lens[b];
    if (Sb > S) Sb = S;
    const int L = 2 * Sb + 1;
    const float* lpg = preds + (long long)b * Tp * V;
    {
        const int n = Tb * V;
        for (int base = 0; base < n; base += 256 * 8) {
            float r[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int i = base + k * 256 + tid; r[k] = i < n ? lpg[i] : 0.f; }
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int i = base + k * 256 + tid; if (i < n) { lpl[i] = r[k]; occ[i] = 0.f; } }
        }
    }
    for (int s = tid; s < L; s += 256) ext[s] = (s & 1) ? (int)targets[(long long)b * S + (s >> 1)] : blank;
    if (tid == 0) s_nll = (Tb == 0 && Sb == 0) ? 0.f : INFINITY;   // (Tb = 0: the empty path has probability 1 for an empty target only)
    __syncthreads();
    const int wv = tid >> 6, i = tid & 63;
    const int m = Tb / 2;
    constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
    const bool alpha = wv == 0;
    const bool live0 = 2 * i < L, live1 = 2 * i + 1 < L;
    const int e = (wv < 2 && live1) ? ext[2 * i + 1] : blank;
    const bool skipa = wv < 2 && live1 && i >= 1 && e != blank && e != ext[2 * i - 1];
    const bool skipb = wv < 2 && live1 && 2 * i + 3 < L && ext[2 * i + 3] != blank && ext[2 * i + 3] != e;
    auto lse2 = [&](float x, float y) {
        const float mx = fmaxf(fmaxf(x, y), -3.0e38f);
        return mx + LN2 * __builtin_amdgcn_logf(__builtin_amdgcn_exp2f((x - mx) * LOG2E) + __builtin_amdgcn_exp2f((y - mx) * LOG2E));
    };
    auto lse3 = [&](float x, float y, float z) {
        const float mx = fmaxf(fmaxf(fmaxf(x, y), z), -3.0e38f);
        return mx + LN2 * __builtin_amdgcn_logf(__builtin_amdgcn_exp2f((x - mx) * LOG2E) + __builtin_amdgcn_exp2f((y - mx) * LOG2E) +
                                                __builtin_amdgcn_exp2f((z - mx) * LOG2E));
    };
    const int ninf = __builtin_bit_cast(int, -INFINITY);
    float v0 = -INFINITY, v1 = -INFINITY;
    const float* lpb = lpl + blank;
    const float* lpe = lpl + e;
    // one frame of the recursion for this wave's lattice (states past L need no masking: see ctc_kernel). ALPHA is a compile-time constant:
    // the loop body is straight-line code for one direction
    auto step = [&](auto alpha_c, bool init, float pb, float pe) {
        constexpr bool ALPHA = decltype(alpha_c)::value;
        if constexpr (ALPHA) {
            if (init) { v0 = (i == 0) ? pb : -INFINITY; v1 = (i == 0 && live1) ? pe : -INFINITY; }
            else {
                const float pl = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(ninf, __builtin_bit_cast(int, v1), 0x138, 0xF, 0xF, false));   // wave_shr:1
                const float n0 = lse2(v0, pl) + pb;
                v1 = lse3(v1, v0, skipa ? pl : -INFINITY) + pe;
                v0 = n0;
            }
        } else {
            if (init) { v0 = (2 * i == L - 1) ? pb : -INFINITY; v1 = (live1 && 2 * i + 1 == L - 2) ? pe : -INFINITY; }
            else {
                const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(ninf, __builtin_bit_cast(int, v0), 0x130, 0xF, 0xF, false));   // wave_shl:1
                const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(ninf, __builtin_bit_cast(int, v1), 0x130, 0xF, 0xF, false));
                v0 = lse2(v0, v1) + pb;
                v1 = lse3(v1, r0, skipb ? r1 : -INFINITY) + pe;
            }
        }
    };
    float nll = INFINITY;
    auto run = [&](auto alpha_c, auto second_c) {
        constexpr bool ALPHA = decltype(alpha_c)::value, SECOND = decltype(second_c)::value;
        constexpr int dt = ALPHA ? 1 : -1;
        const int n = SECOND ? (ALPHA ? Tb - m : m) : (ALPHA ? m : Tb - m);
        float* lat = ALPHA ? latA : latB - (long long)m * Lmax;           // row t of the wave's own lattice
        const float* oth = ALPHA ? latB - (long long)m * Lmax : latA;     // row t of the OTHER lattice
        int t = SECOND ? (ALPHA ? m : m - 1) : (ALPHA ? 0 : Tb - 1);
        const int t_init = ALPHA ? 0 : Tb - 1;
        float pb = n > 0 ? lpb[t * V] : 0.f, pe = n > 0 ? lpe[t * V] : 0.f;
        for (int st = 0; st < n; ++st) {
            const int tn = t + dt, tc = min(max(tn, 0), Tb - 1);
            const float nb = lpb[tc * V], ne = lpe[tc * V];
            if constexpr (!SECOND) {
                step(alpha_c, t == t_init, pb, pe);
                if (live0) lat[t * Lmax + 2 * i] = v0;
                if (live1) lat[t * Lmax + 2 * i + 1] = v1;
            } else {
                const float o1 = live1 ? oth[t * Lmax + 2 * i + 1] : -INFINITY;
                step(alpha_c, t == t_init, pb, pe);
                const float x1 = live1 ? v1 + o1 - pe : -INFINITY;   // log alpha beta / y of the label state
                if (st == 0) {   // log-likelihood = logsumexp over ALL states of this frame (the only place the blank states' products are needed)
                    const float x0 = live0 ? v0 + oth[t * Lmax + 2 * i] - pb : -INFINITY;
                    const float mx = wave_max(fmaxf(x0, x1));
                    if (mx > -INFINITY) {
                        const float sm2 = wave_sum(expf(x0 - mx) + expf(x1 - mx));
                        nll = -(mx + logf(sm2));
                    }
                    if (ALPHA && i == 0) s_nll = nll;
                }
                // posterior mass of the LABEL classes only: a frame's posteriors sum to one, so the blank column is filled in afterwards as
                // 1 - (the labels' sum) instead of 61 same-address adds (or a wave reduction) per frame
                if (nll < INFINITY && x1 > -INFINITY) atomicAdd(&occ[t * V + e], __builtin_amdgcn_exp2f((x1 + nll) * LOG2E));
            }
            pb = nb; pe = ne; t = tn;
        }
    };
    if (Tb > 0) {
        if (wv == 0) run(std::true_type{}, std::false_type{});
        else if (wv == 1) run(std::false_type{}, std::false_type{});
    }
    __syncthreads();
    if (Tb > 0) {
        if (wv == 0) run(std::true_type{}, std::true_type{});
        else if (wv == 1) run(std::false_type{}, std::true_type{});
    }
    __syncthreads();
    nll = s_nll;
    const bool finite = nll < INFINITY;
    if (tid == 0) loss[b] = finite ? nll : (zero_inf ? 0.f : INFINITY);
    if (!dlogits) return;
    // rows beyond the input length (and whole infeasible samples) get zero gradient
    const int tz = finite ? Tb : 0;
    for (long long k = tid + (long long)tz * ldd; k < (long long)Tp * ldd; k += 256) stf<TD>(dlogits, (long long)b * Tp * ldd + k, 0.f);
    if (!finite || Tb == 0) return;
    for (int t = tid; t < Tb; t += 256) {   // the blank column: what the labels leave of the frame's unit mass
        float sl = 0.f;
        for (int c = 0; c < V; ++c) sl += (c == blank) ? 0.f : occ[t * V + c];
        occ[t * V + blank] = 1.0f - sl;
    }
    __syncthreads();
    for (int k = tid; k < Tb * ldd; k += 256) {
        const int t = k / ldd, c = k % ldd;
        const float v = (c < V) ? (expf(lpl[t * V + c]) - occ[t * V + c]) * grad_scale : 0.f;
        stf<TD>(dlogits, ((long long)b * Tp + t) * ldd + c, v);
    }
}

int ctc_launch(const float* preds, const int64_t* targets, const int32_t* in_lens, const int64_t* tgt_lens, int B, int Tp,
               int V, int S, int blank, int zero_infinity, float* loss, float* alpha_ws, void* dlogits, int d_dtype,
               int ldd, float grad_scale, hipStream_t s) {
    const size_t fixed = (size_t)5 * (2 * S + 1) * sizeof(float), per_frame = (size_t)V * sizeof(float);
    NBCI_REQUIRE(fixed + 16 * per_frame <= 64000, NBCI_ESHAPE, "ctc: targets x vocab too large for the LDS-resident lattice rows");
    NBCI_REQUIRE(blank >= 0 && blank < V, NBCI_EINVAL, "ctc: blank id out of range");
    const size_t lds1 = fixed + (size_t)Tp * per_frame, lds2 = lds1 + (size_t)Tp * per_frame;
    const size_t lds_mid = ((size_t)2 * ((Tp + 1) / 2) * (2 * S + 1) + (size_t)2 * Tp * V + (2 * S + 1)) * sizeof(float);
    static const bool mid_on = measure_env("NBCI_CTC_MID", 1) != 0;
    if (mid_on && 2 * S + 1 <= 128 && lds_mid <= 160000) {   // both half lattices, the log-probabilities and the posteriors in LDS: no workspace traffic
        if (lds_mid > 65536) {
            const int r = ensure_dyn_lds(d_dtype == NBCI_BF16 ? (const void*)ctc_mid_kernel<bf16_t> : (const void*)ctc_mid_kernel<float>, 160000, "ctc");
            if (r != NBCI_OK) return r;
        }
        DISPATCH_DTYPE(d_dtype, TD,
                       hipLaunchKernelGGL((ctc_mid_kernel<TD>), dim3(B), dim3(256), lds_mid, s, preds, targets, in_lens, tgt_lens, Tp, V, S, blank,
                                          zero_infinity, loss, (TD*)dlogits, ldd, grad_scale));
        return check_launch("ctc");
    }
    if (lds2 <= 64000) {   // room to stage the log-probabilities as well
        DISPATCH_DTYPE(d_dtype, TD,
                       hipLaunchKernelGGL((ctc_kernel<TD, true>), dim3(B), dim3(256), lds2, s, preds, targets, in_lens, tgt_lens, Tp, V,
                                          S, blank, zero_infinity, loss, alpha_ws, (TD*)dlogits, ldd, grad_scale, B, Tp));
    } else {   // occupancy pass in chunks of frames (multiples of 16: the waves' frame stride) that fit 64 KB
        int chunk = Tp;
        if (lds1 > 64000) chunk = (int)((64000 - fixed) / per_frame) / 16 * 16;
        DISPATCH_DTYPE(d_dtype, TD,
                       hipLaunchKernelGGL((ctc_kernel<TD, false>), dim3(B), dim3(256), fixed + (size_t)chunk * per_frame, s, preds, targets, in_lens,
                                          tgt_lens, Tp, V, S, blank, zero_infinity, loss, alpha_ws, (TD*)dlogits, ldd, grad_scale, B, chunk));
    }
    return check_launch("ctc");
}

// greedy decode with the reference's collapse rule + token Levenshtein. One wave per sample: the
// path is staged in LDS, lane 0 does the (inherently sequential) collapse, then the edit-distance
// table is swept by anti-diagonals with 64 cells in flight.
__global__ __launch_bounds__(64) void per_kernel(const int32_t* __restrict__ argmax, const int64_t* __restrict__ targets,
                                                 const int64_t* __restrict__ tgt_lens, int Tp, int S, int blank,
                                                 int32_t* __restrict__ decoded, int32_t* __restrict__ dec_lens,
                                                 int32_t* __restrict__ errors) {
    extern __shared__ int smi[];
    int* path = smi;                // [Tp]
    int* dec = path + Tp;           // [Tp]
    int* tg = dec + Tp;             // [S + 1]
    int* d0 = tg + S + 1;           // three diagonals, indexed by i (prediction position), [Tp + 2] each
    int* d1 = d0 + Tp + 2;
    int* d2 = d1 + Tp + 2;
    __shared__ int s_n;
    const int b = blockIdx.x, lane = threadIdx.x;
    for (int t = lane; t < Tp; t += 64) path[t] = argmax[(long long)b * Tp + t];
    int nt = (int)tgt_lens[b];
    if (nt > S) nt = S;
    for (int j = lane; j < nt; j += 64) tg[j] = (int)targets[(long long)b * S + j];
    __syncthreads();
    if (lane == 0) {
        int n = 0, last = -1;
        for (int t = 0; t < Tp; ++t) {
            const int idx = path[t];
            if (idx != last && idx != blank) { dec[n++] = idx; last = idx; }  // `last` only moves on emission
        }
        s_n = n;
    }
    __syncthreads();
    const int n = s_n;
    for (int t = lane; t < Tp; t += 64) decoded[(long long)b * Tp + t] = t < n ? dec[t] : -1;
    // " ".join([]).split(" ") == [""]: an empty side counts as ONE empty token (id -1)
    const int na = n > 0 ? n : 1, nb = nt > 0 ? nt : 1;
    int *p2 = d0, *p1 = d1, *cur = d2;
    if (lane == 0) p1[0] = 0;  // diagonal 0
    __syncthreads();
    for (int d = 1; d <= na + nb; ++d) {
        const int ilo = d - nb > 0 ? d - nb : 0, ihi = d < na ? d : na;
        for (int i = ilo + lane; i <= ihi; i += 64) {
            const int j = d - i;
            int v;
            if (i == 0) v = j;
            else if (j == 0) v = i;
            else {
                const int x = n > 0 ? dec[i - 1] : -1, y = nt > 0 ? tg[j - 1] : -1;
                v = p1[i - 1] + 1;                          // D[i-1][j]
                const int ins = p1[i] + 1;                  // D[i][j-1]
                const int sub = p2[i - 1] + (x != y ? 1 : 0);
                v = ins < v ? ins : v;
                v = sub < v ? sub : v;
            }
            cur[i] = v;
        }
        __syncthreads();
        int* tmp = p2; p2 = p1; p1 = cur; cur = tmp;
    }
    if (lane == 0) {
        dec_lens[b] = n;
        errors[2 * b] = p1[na];
        errors[2 * b + 1] = nb;
    }
}

int per_launch(const int32_t* argmax, const int64_t* targets, const int64_t* tgt_lens, int B, int Tp, int S, int blank,
               int32_t* decoded, int32_t* dec_lens, int32_t* errors, int32_t* scratch, hipStream_t s) {
    (void)scratch;  // kept in the C-ABI for callers that sized it; the table now lives in LDS
    const size_t lds = (size_t)(2 * Tp + (S + 1) + 3 * (Tp + 2)) * sizeof(int);
    NBCI_REQUIRE(lds <= 60000, NBCI_ESHAPE, "per: sequence too long for the LDS-resident edit distance");
    hipLaunchKernelGGL(per_kernel, dim3(B), dim3(64), lds, s, argmax, targets, tgt_lens, Tp, S, blank, decoded, dec_lens, errors);
    return check_launch("per");
}

// ------------------------------------------------------------------------------------------
// embedder.adapt (ndt1.py:124-129,170-171): one embed_spikes Linear per recording day, picked per sample
// ------------------------------------------------------------------------------------------
// forward: copy each sample's day weights (D*N, activation dtype) next to each other so the embed is ONE batched GEMM,
// and fill the per-row table the GEMM's residual gather uses to add that day's bias (row (b,t) -> day[b])
template <typename T>
__global__ __launch_bounds__(256) void adapt_gather_kernel(const T* __restrict__ wsrc, long long day_stride, const int64_t* __restrict__ day,
                                                           T* __restrict__ wsel, int64_t* __restrict__ rows, int wn, int Tt, int ndays) {
    const int b = blockIdx.y;
    long long d = day[b];
    d = d < 0 ? 0 : (d >= ndays ? ndays - 1 : d);   // (validated on the host side of the module; never index out of the table)
    for (int i = blockIdx.x * 256 + threadIdx.x; i < wn; i += gridDim.x * 256) wsel[(long long)b * wn + i] = wsrc[d * day_stride + i];
    if (blockIdx.x == 0)
        for (int t = threadIdx.x; t < Tt; t += 256) rows[(long long)b * Tt + t] = d;
}

int adapt_gather_launch(const void* wsrc, long long day_stride, const int64_t* day, void* wsel, int64_t* rows, int dtype, int B, int wn,
                        int T, int ndays, hipStream_t s) {
    NBCI_REQUIRE(wsrc && day && wsel && rows && B > 0 && wn > 0 && ndays > 0, NBCI_EINVAL, "adapt_gather: bad argument");
    DISPATCH_DTYPE(dtype, TT, hipLaunchKernelGGL((adapt_gather_kernel<TT>), dim3((unsigned)((wn + 1023) / 1024), B), dim3(256), 0, s,
                                                 (const TT*)wsrc, day_stride, day, (TT*)wsel, rows, wn, T, ndays));
    return check_launch("adapt_gather");
}

// backward, step 1: per-sample column sums of d(pre-activation) (B,T,D) -> bsum (B,D) f32 (that sample's bias gradient)
template <typename T>
__global__ __launch_bounds__(256) void adapt_bias_kernel(const T* __restrict__ dpre, float* __restrict__ bsum, int Tt, int D) {
    __shared__ float red[4][64];
    const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    float acc = 0.f;
    if (c < D)
        for (int t = g; t < Tt; t += 4) acc += ldf<T>(dpre, ((long long)b * Tt + t) * D + c);
    red[g][threadIdx.x & 63] = acc;
    __syncthreads();
    if (g == 0 && c < D) bsum[(long long)b * D + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// backward, step 2: every thread owns one element of the day layer (weight then bias) and adds the samples' partial
// gradients into the rows of their days in a FIXED order (deterministic; two samples of one day never race)
__global__ __launch_bounds__(256) void adapt_scatter_kernel(const float* __restrict__ wpart, const float* __restrict__ bpart,
                                                            const int64_t* __restrict__ day, float* __restrict__ gw, float* __restrict__ gb,
                                                            long long day_stride, int B, int wn, int D, int ndays) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= wn + D) return;
    for (int b = 0; b < B; ++b) {
        long long d = day[b];
        d = d < 0 ? 0 : (d >= ndays ? ndays - 1 : d);
        if (e < wn) gw[d * day_stride + e] += wpart[(long long)b * wn + e];
        else gb[d * day_stride + (e - wn)] += bpart[(long long)b * D + (e - wn)];
    }
}

int adapt_grads_launch(const void* dpre, int dtype, const float* wpart, float* bpart, const int64_t* day, float* gw, float* gb,
                       long long day_stride, int B, int T, int D, int wn, int ndays, hipStream_t s) {
    NBCI_REQUIRE(dpre && wpart && bpart && day && gw && gb, NBCI_EINVAL, "adapt_grads: null argument");
    DISPATCH_DTYPE(dtype, TT, hipLaunchKernelGGL((adapt_bias_kernel<TT>), dim3((unsigned)((D + 63) / 64), B), dim3(256), 0, s, (const TT*)dpre,
                                                 bpart, T, D));
    hipLaunchKernelGGL(adapt_scatter_kernel, dim3((unsigned)((wn + D + 255) / 256)), dim3(256), 0, s, wpart, (const float*)bpart, day, gw, gb,
                       day_stride, B, wn, D, ndays);
    return check_launch("adapt_grads");
}

// out = src * gate, f32 -> activation dtype: an external f32 gradient pushed back through a stored activation derivative
template <typename T>
__global__ __launch_bounds__(256) void gate_cast_kernel(const float* __restrict__ src, const T* __restrict__ gate, T* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) stf<T>(out, i, src[i] * ldf<T>(gate, i));
}

int gate_cast_launch(const float* src, const void* gate, void* out, int dtype, long long n, hipStream_t s) {
    NBCI_REQUIRE(src && gate && out && n > 0, NBCI_EINVAL, "gate_cast: null argument");
    DISPATCH_DTYPE(dtype, TT, hipLaunchKernelGGL((gate_cast_kernel<TT>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src,
                                                 (const TT*)gate, (TT*)out, n));
    return check_launch("gate_cast");
}

// Running statistics of the train loop, one launch per step instead of ~9 tiny framework kernels:
// stats[0] += sum(loss), stats[1] += n_examples, and (PER on) stats[2] += sum(err[:,0]) / sum(err[:,1]), stats[3] += 1
// — the reference's per-step bookkeeping (trainer.py:353-362: loss sum, example count, per-batch metric ratio).
__global__ __launch_bounds__(64) void step_stats_kernel(double* __restrict__ stats, const float* __restrict__ loss, int B, double n_examples,
                                                        const int32_t* __restrict__ err) {
    const int lane = threadIdx.x;
    double ls = 0.0;
    long long e0 = 0, e1 = 0;
    for (int i = lane; i < B; i += 64) {
        ls += (double)loss[i];
        if (err) { e0 += err[2 * i]; e1 += err[2 * i + 1]; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ls += __shfl_xor(ls, o, 64);
        e0 += __shfl_xor(e0, o, 64);
        e1 += __shfl_xor(e1, o, 64);
    }
    if (lane == 0) {
        stats[0] += ls;
        stats[1] += n_examples;
        if (err) { stats[2] += (double)e0 / (double)e1; stats[3] += 1.0; }
    }
}

int step_stats_launch(double* stats, const float* loss, int B, double n_examples, const int32_t* err, hipStream_t s) {
    NBCI_REQUIRE(stats && loss && B > 0, NBCI_EINVAL, "step_stats: null argument");
    hipLaunchKernelGGL(step_stats_kernel, dim3(1), dim3(64), 0, s, stats, loss, B, n_examples, err);
    return check_launch("step_stats");
}

// ------------------------------------------------------------------------------------------
// BCI coupler splice (models/bci.py:143-166)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void splice_fwd_kernel(const T* __restrict__ text, const T* __restrict__ spikes, T* __restrict__ out,
                                                         const int64_t* __restrict__ tmask, const int64_t* __restrict__ svalid,
                                                         int64_t* __restrict__ mask_out, const int64_t* __restrict__ targets,
                                                         int64_t* __restrict__ targets_out, const int64_t* __restrict__ split,
                                                         int B, int Lt, int Ts, int H) {
    const int L = Lt + Ts;
    const int row = blockIdx.x;           // (b, pos)
    const int b = row / L, pos = row % L;
    int d = (int)split[b];
    d = d < 0 ? 0 : (d > Lt ? Lt : d);
    const bool is_spike = pos >= d && pos < d + Ts;
    const int src = is_spike ? pos - d : (pos < d ? pos : pos - Ts);
    const T* s = is_spike ? spikes + ((long long)b * Ts + src) * H : (text ? text + ((long long)b * Lt + src) * H : nullptr);
    T* o = out + (long long)row * H;
    constexpr int E = 16 / (int)sizeof(T);   // 16-byte copies when the rows allow it (H_llm = 4096: one row = 8 KB bf16)
    if (H % E == 0 && ((uintptr_t)o % 16 == 0) && (!s || (uintptr_t)s % 16 == 0)) {
        const uint4* s4 = (const uint4*)s;
        uint4* o4 = (uint4*)o;
        for (int c = threadIdx.x; c < H / E; c += 256) o4[c] = s ? s4[c] : make_uint4(0u, 0u, 0u, 0u);
    } else {
        for (int c = threadIdx.x; c < H; c += 256) o[c] = s ? s[c] : (T)0.0f;
    }
    if (threadIdx.x == 0) {
        if (mask_out) mask_out[row] = is_spike ? (svalid ? svalid[b * Ts + src] : 1) : (tmask ? tmask[b * Lt + src] : 1);
        if (targets_out) targets_out[row] = is_spike ? -100 : (targets ? targets[b * Lt + src] : -100);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void splice_bwd_kernel(const T* __restrict__ dout, T* __restrict__ dtext, T* __restrict__ dspikes,
                                                         const int64_t* __restrict__ split, int B, int Lt, int Ts, int H) {
    const int L = Lt + Ts;
    const int row = blockIdx.x;
    const int b = row / L, pos = row % L;
    int d = (int)split[b];
    d = d < 0 ? 0 : (d > Lt ? Lt : d);
    const bool is_spike = pos >= d && pos < d + Ts;
    const int src = is_spike ? pos - d : (pos < d ? pos : pos - Ts);
    T* dst = is_spike ? dspikes + ((long long)b * Ts + src) * H : (dtext ? dtext + ((long long)b * Lt + src) * H : nullptr);
    if (!dst) return;
    const T* g = dout + (long long)row * H;
    constexpr int E = 16 / (int)sizeof(T);
    if (H % E == 0 && ((uintptr_t)g % 16 == 0) && ((uintptr_t)dst % 16 == 0)) {
        const uint4* g4 = (const uint4*)g;
        uint4* d4 = (uint4*)dst;
        for (int c = threadIdx.x; c < H / E; c += 256) d4[c] = g4[c];
    } else {
        for (int c = threadIdx.x; c < H; c += 256) dst[c] = g[c];
    }
}

int splice_fwd_launch(const void* text, const void* spikes, void* out, int dtype, const int64_t* tmask, const int64_t* svalid,
                      int64_t* mask_out, const int64_t* targets, int64_t* targets_out, const int64_t* split, int B, int Lt, int Ts,
                      int H, hipStream_t s) {
    NBCI_REQUIRE(spikes && out && split && B > 0 && Ts > 0 && Lt >= 0 && H > 0, NBCI_EINVAL, "splice: bad arguments");
    DISPATCH_DTYPE(dtype, TT,
                   hipLaunchKernelGGL((splice_fwd_kernel<TT>), dim3(B * (Lt + Ts)), dim3(256), 0, s, (const TT*)text, (const TT*)spikes,
                                      (TT*)out, tmask, svalid, mask_out, targets, targets_out, split, B, Lt, Ts, H));
    return check_launch("splice_fwd");
}

int splice_bwd_launch(const void* dout, void* dtext, void* dspikes, int dtype, const int64_t* split, int B, int Lt, int Ts, int H,
                      hipStream_t s) {
    NBCI_REQUIRE(dout && dspikes && split, NBCI_EINVAL, "splice: bad arguments");
    DISPATCH_DTYPE(dtype, TT,
                   hipLaunchKernelGGL((splice_bwd_kernel<TT>), dim3(B * (Lt + Ts)), dim3(256), 0, s, (const TT*)dout, (TT*)dtext,
                                      (TT*)dspikes, split, B, Lt, Ts, H));
    return check_launch("splice_bwd");
}

// ------------------------------------------------------------------------------------------
// fold the replicated small-vector accumulators into the flat gradient buffer and clear them
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void fold_replicas_kernel(float* __restrict__ rep, long long stride, int nrep,
                                                            const int* __restrict__ flat_of, int cbegin, int cend,
                                                            float* __restrict__ grads) {
    const int ci = cbegin + blockIdx.x * 64 + threadIdx.x;
    if (ci >= cend) return;
    float s = 0.f;
    if (nrep == 32) {  // all 32 loads in flight at once (the dependent-load loop was latency-bound)
        float v[32];
#pragma unroll
        for (int r = 0; r < 32; ++r) v[r] = rep[(long long)r * stride + ci];
#pragma unroll
        for (int r = 0; r < 32; ++r) { s += v[r]; rep[(long long)r * stride + ci] = 0.f; }
    } else {
        for (int r = 0; r < nrep; ++r) { s += rep[(long long)r * stride + ci]; rep[(long long)r * stride + ci] = 0.f; }
    }
    const int fo = flat_of[ci];
    if (fo >= 0) grads[fo] += s;
}

int fold_replicas_launch(float* rep, long long stride, int nrep, const int* flat_of, int cbegin, int cend, float* grads,
                         hipStream_t s) {
    if (cend <= cbegin) return NBCI_OK;
    hipLaunchKernelGGL(fold_replicas_kernel, dim3((cend - cbegin + 63) / 64), dim3(64), 0, s, rep, stride, nrep, flat_of, cbegin,
                       cend, grads);
    return check_launch("fold_replicas");
}

// ------------------------------------------------------------------------------------------
// fused AdamW over the flat parameter buffer
// ------------------------------------------------------------------------------------------
template <bool ZERO, typename TG>   // TG = float, or bf16_t: the gradient as a bf16 all-reduce bucket left it (no widening pass)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, TG* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16_t* __restrict__ plp, long long n, float lr,
                                                    float b1, float b2, float eps, float wd, float inv_bc1,
                                                    float inv_sqrt_bc2, float gscale) {
    long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    const long long stride = (long long)gridDim.x * 256 * 4;
    for (; i + 3 < n; i += stride) {
        float4 pv = *(float4*)(p + i), mv = *(float4*)(m + i), vv = *(float4*)(v + i);
        float gg[4];
        if constexpr (sizeof(TG) == 4) { const float4 gv = *(const float4*)(g + i); gg[0] = gv.x; gg[1] = gv.y; gg[2] = gv.z; gg[3] = gv.w; }
        else { const bf16x4 gv = *(const bf16x4*)(g + i); gg[0] = bf2f(gv[0]); gg[1] = bf2f(gv[1]); gg[2] = bf2f(gv[2]); gg[3] = bf2f(gv[3]); }
        float pp[4] = {pv.x, pv.y, pv.z, pv.w}, mm[4] = {mv.x, mv.y, mv.z, mv.w}, v2[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gr = gg[e] * gscale;
            pp[e] *= (1.0f - lr * wd);
            mm[e] = b1 * mm[e] + (1.0f - b1) * gr;
            v2[e] = b2 * v2[e] + (1.0f - b2) * gr * gr;
            const float denom = sqrtf(v2[e]) * inv_sqrt_bc2 + eps;
            pp[e] -= (lr * inv_bc1) * (mm[e] / denom);
        }
        *(float4*)(p + i) = make_float4(pp[0], pp[1], pp[2], pp[3]);
        *(float4*)(m + i) = make_float4(mm[0], mm[1], mm[2], mm[3]);
        *(float4*)(v + i) = make_float4(v2[0], v2[1], v2[2], v2[3]);
        if (plp) { bf16x4 o = {f2bf(pp[0]), f2bf(pp[1]), f2bf(pp[2]), f2bf(pp[3])}; *(bf16x4*)(plp + i) = o; }
        if constexpr (ZERO && sizeof(TG) == 4) *(float4*)(g + i) = make_float4(0.f, 0.f, 0.f, 0.f);   // zero_grad (trainer.py:342) in the pass that consumed the gradient
    }
}

int adamw_launch(float* p, void* g, float* m, float* v, void* p_lp, int64_t n, float lr, float beta1, float beta2,
                 float eps, float wd, float bc1, float bc2, float grad_scale, hipStream_t s, bool zero_grad, int max_blocks, bool g_bf16) {
    NBCI_REQUIRE(n % 4 == 0, NBCI_ESHAPE, "adamw: flat buffer length must be a multiple of 4");
    NBCI_REQUIRE(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % (g_bf16 ? 8 : 16) == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0),
                 NBCI_EALIGN, "adamw: buffers must be 16-byte aligned");
    // max_blocks: a caller that runs the update beside other kernels (aux stream) keeps it to a few workgroups per CU so that
    // those kernels' workgroups still find wave slots; 2 x 256 threads per CU keep ~32 KB of loads in flight per CU
    const long long blocks = std::min<long long>(max_blocks > 0 ? max_blocks : 2048, (n / 4 + 255) / 256);
    const dim3 grid((unsigned)std::max<long long>(1, blocks));
    if (g_bf16)
        hipLaunchKernelGGL((adamw_kernel<false, bf16_t>), grid, dim3(256), 0, s, p, (bf16_t*)g, m, v, (bf16_t*)p_lp,
                           (long long)n, lr, beta1, beta2, eps, wd, 1.0f / bc1, 1.0f / sqrtf(bc2), grad_scale);
    else if (zero_grad)
        hipLaunchKernelGGL((adamw_kernel<true, float>), grid, dim3(256), 0, s, p, (float*)g, m, v, (bf16_t*)p_lp,
                           (long long)n, lr, beta1, beta2, eps, wd, 1.0f / bc1, 1.0f / sqrtf(bc2), grad_scale);
    else
        hipLaunchKernelGGL((adamw_kernel<false, float>), grid, dim3(256), 0, s, p, (float*)g, m, v, (bf16_t*)p_lp,
                           (long long)n, lr, beta1, beta2, eps, wd, 1.0f / bc1, 1.0f / sqrtf(bc2), grad_scale);
    return check_launch("adamw");
}

}  // namespace nbci
