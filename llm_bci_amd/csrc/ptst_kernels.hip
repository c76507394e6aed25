// ptst_kernels.hip — the non-GEMM kernels only the PatchTST path needs (models/patchtst.py over HF transformers'
// PatchTSTModel: random patch masking, patchify, shared patch embedding + sincos positions, BatchNorm over all
// (batch, channel, patch) rows, channel mean-pooling, per-patch mlm loss). All HBM-bound, one pass over their tensors;
// BatchNorm column statistics are chunk partials combined with Chan's update (no atomics, deterministic, no E[x^2]-mean^2
// cancellation). See kernels.h for the launch API.
#include "kernels.h"

namespace nbci {

template <typename T> __device__ __forceinline__ void stq(T* p, long long i, float v);
template <> __device__ __forceinline__ void stq<float>(float* p, long long i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void stq<bf16_t>(bf16_t* p, long long i, float v) { p[i] = f2bf(v); }
template <typename T> __device__ __forceinline__ float ldq(const T* p, long long i);
template <> __device__ __forceinline__ float ldq<float>(const float* p, long long i) { return p[i]; }
template <> __device__ __forceinline__ float ldq<bf16_t>(const bf16_t* p, long long i) { return bf2f(p[i]); }

static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NBCI_EHIP, std::string(what) + ": " + hipGetErrorString(e));
    return NBCI_OK;
}

// ------------------------------------------------------------------------------------------
// random_masking (transformers PatchTST): per (b,c) row the P - len_keep patches with the largest noise are masked.
// One block per row; rank by counting (ties broken by index, = a stable argsort).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ptst_mask_kernel(unsigned char* __restrict__ mask, int B, int C, int P, int keep,
                                                        int channel_consistent, uint32_t seed, uint32_t site) {
    extern __shared__ unsigned noise[];
    const int row = blockIdx.x, b = row / C;
    for (int p = threadIdx.x; p < P; p += 256)
        noise[p] = rng_u32(seed, site, (uint32_t)((channel_consistent ? b : row) * P + p));
    __syncthreads();
    for (int p = threadIdx.x; p < P; p += 256) {
        const unsigned v = noise[p];
        int rank = 0;
        for (int q = 0; q < P; ++q) rank += (noise[q] < v) || (noise[q] == v && q < p);
        mask[(long long)row * P + p] = rank >= keep;
    }
}

int ptst_mask_launch(uint8_t* mask, int B, int C, int P, double ratio, int channel_consistent, uint32_t seed, uint32_t site,
                     hipStream_t s) {
    NBCI_REQUIRE(ratio >= 0.0 && ratio < 1.0, NBCI_EINVAL, "Mask ratio has to be between 0 and 1.");
    NBCI_REQUIRE(P <= 8192, NBCI_ESHAPE, "ptst mask: at most 8192 patches");
    NBCI_REQUIRE((long long)B * C * P < (1ll << 32), NBCI_ESHAPE, "ptst mask: tensor too large for the RNG counter");
    const int keep = (int)((double)P * (1.0 - ratio));   // int(sequence_length * (1 - mask_ratio))
    hipLaunchKernelGGL(ptst_mask_kernel, dim3(B * C), dim3(256), P * sizeof(unsigned), s, mask, B, C, P, keep, channel_consistent, seed, site);
    return check_launch("ptst_mask");
}

// ------------------------------------------------------------------------------------------
// patchify (PatchTSTPatchify): patch[b,c,p,j] = x[b, start + p*stride + j, c]; xm = masked copy (mask_value where masked).
// 32x32 LDS tile: reads coalesced along channels, writes coalesced along (p,j).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ptst_patchify_kernel(const float* __restrict__ x, float* __restrict__ patch, float* __restrict__ xm,
                                                            const unsigned char* __restrict__ mask, int T, int C, int P, int pl, int stride,
                                                            int start, float mask_value) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, e0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int E = P * pl;
    for (int r = ty; r < 32; r += 8) {
        const int e = e0 + r, c = c0 + tx;
        float v = 0.f;
        if (e < E && c < C) { const int p = e / pl, j = e - p * pl; v = x[((long long)b * T + start + p * stride + j) * C + c]; }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, e = e0 + tx;
        if (c < C && e < E) {
            const float v = tile[tx][r];
            const long long o = ((long long)b * C + c) * E + e;
            if (patch) patch[o] = v;
            const bool m = mask && mask[((long long)b * C + c) * P + e / pl];
            xm[o] = m ? mask_value : v;
        }
    }
}

int ptst_patchify_launch(const float* x, float* patch, float* xm, const uint8_t* mask, int B, int T, int C, int P, int pl, int stride,
                         int start, float mask_value, hipStream_t s) {
    hipLaunchKernelGGL(ptst_patchify_kernel, dim3((C + 31) / 32, (P * pl + 31) / 32, B), dim3(256), 0, s, x, patch, xm, mask, T, C, P, pl,
                       stride, start, mask_value);
    return check_launch("ptst_patchify");
}

// ------------------------------------------------------------------------------------------
// shared patch embedding + positions (+ positional dropout): h[row,:] = xm[row,:] W^T + b + pos[row % P, :]
// K = patch_length (10): a per-thread dot product, the kernel is bound by the (M, D) f32 write.
// ------------------------------------------------------------------------------------------
template <int PLMAX>
__global__ __launch_bounds__(256) void ptst_embed_kernel(const float* __restrict__ xm, const float* __restrict__ W, const float* __restrict__ bias,
                                                         const float* __restrict__ pos, float* __restrict__ h, long long M, int P, int pl, int D,
                                                         unsigned thr, float dscale, uint32_t key) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // one thread = (row, 4 consecutive d)
    const int dq = D / 4;
    if (i >= M * dq) return;
    const long long row = i / dq;
    const int d = (int)(i % dq) * 4;
    float xv[PLMAX];
#pragma unroll
    for (int j = 0; j < PLMAX; ++j) xv[j] = j < pl ? xm[row * pl + j] : 0.f;
    const int p = (int)(row % P);
    float r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float acc = bias[d + e];
        const float* w = W + (long long)(d + e) * pl;
#pragma unroll
        for (int j = 0; j < PLMAX; ++j) if (j < pl) acc += xv[j] * w[j];
        r[e] = acc + pos[(long long)p * D + d + e];
    }
    const long long o = row * D + d;
    if (thr) drop4(key, thr, (unsigned)o, dscale, r);
    *(float4*)(h + o) = make_float4(r[0], r[1], r[2], r[3]);
}

// Same arithmetic with the weights held in registers: a thread owns 4 output columns for ALL the rows of its block's slice (its 4 x pl
// weights and 4 biases are loaded once) and walks the rows; per row it reads the patch (pl values, the same addresses for the 64 threads
// of the row: one cache line) and one float4 of the position table. The per-output version above issues ~64 load instructions per four
// outputs (2.0 ms at 420 k rows x 256: the 430 MB result is written at 0.2 TB/s); this one ~12.
constexpr int EMB_ROWS = 64;   // rows per block
template <int PLMAX, typename TH>   // TH: storage of the residual stream (f32, or bf16)
__global__ __launch_bounds__(256) void ptst_embed_rows_kernel(const float* __restrict__ xm, const float* __restrict__ W, const float* __restrict__ bias,
                                                              const float* __restrict__ pos, TH* __restrict__ h, long long M, int P, int pl, int D,
                                                              unsigned thr, float dscale, uint32_t key) {
    const int dq = D / 4, rpp = 256 / dq;            // threads per row, rows per pass (the launcher checks 256 % dq == 0)
    const int d = (threadIdx.x % dq) * 4, rsub = threadIdx.x / dq;
    float w[4][PLMAX], bs[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        bs[e] = bias[d + e];
#pragma unroll
        for (int j = 0; j < PLMAX; ++j) w[e][j] = j < pl ? W[(long long)(d + e) * pl + j] : 0.f;
    }
    const long long r_end = min(M, ((long long)blockIdx.x + 1) * EMB_ROWS);
    for (long long row = (long long)blockIdx.x * EMB_ROWS + rsub; row < r_end; row += rpp) {
        float xv[PLMAX];
#pragma unroll
        for (int j = 0; j < PLMAX; ++j) xv[j] = j < pl ? xm[row * pl + j] : 0.f;
        const float4 pe = *(const float4*)(pos + (long long)(row % P) * D + d);
        float r[4] = {bs[0], bs[1], bs[2], bs[3]};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < PLMAX; ++j) if (j < pl) r[e] += xv[j] * w[e][j];   // (same order of additions as the kernel above)
        r[0] += pe.x; r[1] += pe.y; r[2] += pe.z; r[3] += pe.w;
        const long long o = row * D + d;
        if (thr) drop4(key, thr, (unsigned)o, dscale, r);
        st4f(h + o, make_float4(r[0], r[1], r[2], r[3]));
    }
}

int ptst_embed_launch(const float* xm, const float* W, const float* bias, const float* pos, void* h, long long M, int P, int pl, int D,
                      float drop_p, uint32_t seed, uint32_t site, hipStream_t s, int h_dtype) {
    NBCI_REQUIRE(pl <= 32 && D % 4 == 0, NBCI_ESHAPE, "ptst embed: patch_length <= 32 and d_model % 4 == 0");
    NBCI_REQUIRE(M * D < (1ll << 32), NBCI_ESHAPE, "ptst embed: tensor too large for the dropout counter");
    const unsigned thr = drop_threshold(drop_p);
    const float dscale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    const long long n = M * (D / 4);
    dim3 g((unsigned)((n + 255) / 256));
    if (pl <= 16 && D / 4 <= 256 && 256 % (D / 4) == 0) {
        const dim3 gr((unsigned)((M + EMB_ROWS - 1) / EMB_ROWS));
        if (h_dtype == NBCI_BF16)
            hipLaunchKernelGGL((ptst_embed_rows_kernel<16, bf16_t>), gr, dim3(256), 0, s, xm, W, bias, pos, (bf16_t*)h, M, P, pl, D, thr, dscale, drop_key(seed, site));
        else
            hipLaunchKernelGGL((ptst_embed_rows_kernel<16, float>), gr, dim3(256), 0, s, xm, W, bias, pos, (float*)h, M, P, pl, D, thr, dscale, drop_key(seed, site));
        return check_launch("ptst_embed");
    }
    NBCI_REQUIRE(h_dtype == NBCI_F32, NBCI_ESHAPE, "ptst embed: a bf16 residual stream needs patch_length <= 16 and d_model / 4 dividing 256");
    if (pl <= 16) hipLaunchKernelGGL((ptst_embed_kernel<16>), g, dim3(256), 0, s, xm, W, bias, pos, (float*)h, M, P, pl, D, thr, dscale, drop_key(seed, site));
    else hipLaunchKernelGGL((ptst_embed_kernel<32>), g, dim3(256), 0, s, xm, W, bias, pos, (float*)h, M, P, pl, D, thr, dscale, drop_key(seed, site));
    return check_launch("ptst_embed");
}

// Weight gradient of the shared patch embedding: dW[d][j] += sum_rows de[row][d] * xm[row][j] (D x pl outputs, K = all M = B*C*P rows).
// As a GEMM this is a 256 x 10 output with K = 420 k on the exact-f32 path: 870 us. It is a streaming reduction: a thread owns column(s)
// d of de and keeps its pl partial sums in registers while its block walks a slice of the rows (the patch row xm[row][0..pl) is the same
// for every thread: staged through LDS in chunks, read back as broadcasts); at the end the block's D x pl partials go through LDS so that
// each atomic wave-instruction covers 64 CONSECUTIVE floats of dW. Reads de once: ~80 us at 420 k x 256.
constexpr int EWG_CHUNK = 128;   // rows of xm staged per pass
// A thread owns FOUR adjacent columns (one 16-byte load per row) of every rpp-th row of the chunk (rpp = 256 / (D / 4) row groups per
// block); the row groups' partial sums meet in LDS at the end. (One column per thread kept 1 KB per wave-load in flight: 1.4 TB/s.)
template <int PLMAX>
__global__ __launch_bounds__(256) void ptst_embed_wgrad_kernel(const float* __restrict__ de, const float* __restrict__ xm, float* __restrict__ dW,
                                                              long long M, int pl, int D, long long rows_per_block) {
    extern __shared__ float sm[];                   // [EWG_CHUNK][PLMAX] patch rows; afterwards [rpp][D][pl] partials
    const int dq = D / 4, rpp = 256 / dq;           // (the launcher checks 256 % dq == 0)
    const int d = (threadIdx.x % dq) * 4, rsub = threadIdx.x / dq;
    float acc[4][PLMAX];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < PLMAX; ++j) acc[e][j] = 0.f;
    for (int i = threadIdx.x; i < EWG_CHUNK * PLMAX; i += 256) sm[i] = 0.f;   // (the stage's columns >= pl stay zero)
    const long long r0 = (long long)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    for (long long rc = r0; rc < r1; rc += EWG_CHUNK) {
        const int nr = (int)min((long long)EWG_CHUNK, r1 - rc);
        __syncthreads();
        for (int i = threadIdx.x; i < nr * pl; i += 256) sm[(i / pl) * PLMAX + i % pl] = xm[rc * pl + i];
        __syncthreads();
#pragma unroll 4
        for (int r = rsub; r < nr; r += rpp) {
            const float4 g = *(const float4*)(de + (rc + r) * D + d);
#pragma unroll
            for (int j = 0; j < PLMAX; ++j) {
                const float xv = sm[r * PLMAX + j];
                acc[0][j] += g.x * xv; acc[1][j] += g.y * xv; acc[2][j] += g.z * xv; acc[3][j] += g.w * xv;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < PLMAX; ++j) if (j < pl) sm[((long long)rsub * D + d + e) * pl + j] = acc[e][j];
    __syncthreads();
    for (int i = threadIdx.x; i < D * pl; i += 256) {
        float t = 0.f;
        for (int g2 = 0; g2 < rpp; ++g2) t += sm[(long long)g2 * D * pl + i];
        atomicAdd(dW + i, t);
    }
}

int ptst_embed_wgrad_launch(const float* de, const float* xm, float* dW, long long M, int pl, int D, hipStream_t s) {
    NBCI_REQUIRE(pl >= 1 && pl <= 16 && D % 4 == 0 && D / 4 <= 256 && 256 % (D / 4) == 0, NBCI_ESHAPE,
                 "ptst embed wgrad: patch_length <= 16 and d_model / 4 dividing 256");
    long long bx = std::min<long long>((M + EWG_CHUNK - 1) / EWG_CHUNK, 1024);
    if (bx < 1) bx = 1;
    long long rpb = (M + bx - 1) / bx;
    rpb = (rpb + EWG_CHUNK - 1) / EWG_CHUNK * EWG_CHUNK;
    bx = (M + rpb - 1) / rpb;
    const int rpp = 256 / (D / 4);
    const size_t lds = sizeof(float) * (size_t)std::max(EWG_CHUNK * 16, rpp * D * pl);
    NBCI_REQUIRE(lds <= 65536, NBCI_ESHAPE, "ptst embed wgrad: d_model x patch_length too large for the LDS reduction");
    hipLaunchKernelGGL((ptst_embed_wgrad_kernel<16>), dim3((unsigned)bx), dim3(256), lds, s, de, xm, dW, M, pl, D, rpb);
    return check_launch("ptst_embed_wgrad");
}

// ------------------------------------------------------------------------------------------
// BatchNorm1d over rows (nn.BatchNorm1d(D) on (B*C, D, P): statistics over all M = B*C*P rows per feature)
// ------------------------------------------------------------------------------------------
constexpr int BN_ROWS = 512;   // rows per chunk

// per (chunk, column): count, mean, M2 — shifted accumulation inside the chunk
template <typename TX>
__global__ __launch_bounds__(256) void bn_stats_kernel(const TX* __restrict__ x, float* __restrict__ part, long long M, int D) {
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= D) return;
    const long long r0 = (long long)blockIdx.x * BN_ROWS;
    const long long r1 = r0 + BN_ROWS < M ? r0 + BN_ROWS : M;
    const float K = (float)x[r0 * D + c];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
    for (long long r = r0; r < r1; ++r) { const float v = (float)x[r * D + c] - K; s1 += v; s2 += v * v; }
    const float n = (float)(r1 - r0);
    const float mean = K + s1 / n;
    const float m2 = fmaxf(s2 - s1 * s1 / n, 0.f);
    float* o = part + ((long long)blockIdx.x * 2) * D;
    o[c] = mean; o[D + c] = m2;
}

// combine the chunk partials (Chan et al.), produce mean / rstd for the normalisation and update the running statistics
// (momentum 0.1, unbiased variance) in train mode; in eval mode just turn the running statistics into mean / rstd.
// One block = 16 channels x 16 groups: group g combines chunks g, g + 16, ... in order, then the 16 group results are combined in
// order by the channel's first thread (a fixed tree: deterministic). One thread per channel walking all chunks — 820 dependent
// double-precision updates at 420 k rows — took 270 us per call, three times the parallel statistics pass it follows.
constexpr int BN_FG = 16;   // groups (and channels) per block
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int nchunks, long long M, int D, float eps,
                                                          float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ run_mean,
                                                          float* __restrict__ run_var, int train) {
    __shared__ double sn[BN_FG][BN_FG], smu[BN_FG][BN_FG], sm2[BN_FG][BN_FG];
    const int cl = threadIdx.x % BN_FG, g = threadIdx.x / BN_FG;
    const int c = blockIdx.x * BN_FG + cl;
    if (!train) {
        if (g == 0 && c < D) { mean[c] = run_mean[c]; rstd[c] = 1.0f / sqrtf(run_var[c] + eps); }
        return;
    }
    double n = 0.0, mu = 0.0, m2 = 0.0;
    if (c < D) {
#pragma unroll 4
        for (int k = g; k < nchunks; k += BN_FG) {
            const long long r0 = (long long)k * BN_ROWS;
            const double nb = (double)((r0 + BN_ROWS < M ? r0 + BN_ROWS : M) - r0);
            const double mb = part[((long long)k * 2) * D + c], m2b = part[((long long)k * 2 + 1) * D + c];
            const double delta = mb - mu, tot = n + nb;
            mu += delta * nb / tot;
            m2 += m2b + delta * delta * n * nb / tot;
            n = tot;
        }
    }
    sn[g][cl] = n; smu[g][cl] = mu; sm2[g][cl] = m2;
    __syncthreads();
    if (g != 0 || c >= D) return;
    for (int j = 1; j < BN_FG; ++j) {
        const double nb = sn[j][cl];
        if (nb == 0.0) continue;
        const double delta = smu[j][cl] - mu, tot = n + nb;
        mu += delta * nb / tot;
        m2 += sm2[j][cl] + delta * delta * n * nb / tot;
        n = tot;
    }
    const double var = m2 / n;
    mean[c] = (float)mu; rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    run_mean[c] = 0.9f * run_mean[c] + 0.1f * (float)mu;
    run_var[c] = 0.9f * run_var[c] + 0.1f * (float)(n > 1.0 ? m2 / (n - 1.0) : var);
}

template <typename TO, typename TX>
__global__ __launch_bounds__(256) void bn_apply_kernel(const TX* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ w, const float* __restrict__ b, TO* __restrict__ y, long long n4,
                                                       int D) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)((i * 4) % D);
    const float4 v = ld4f(x + i * 4);
    const float4 mu = *(const float4*)(mean + c), rs = *(const float4*)(rstd + c), ww = *(const float4*)(w + c), bb = *(const float4*)(b + c);
    stq<TO>(y, i * 4 + 0, (v.x - mu.x) * rs.x * ww.x + bb.x);
    stq<TO>(y, i * 4 + 1, (v.y - mu.y) * rs.y * ww.y + bb.y);
    stq<TO>(y, i * 4 + 2, (v.z - mu.z) * rs.z * ww.z + bb.z);
    stq<TO>(y, i * 4 + 3, (v.w - mu.w) * rs.w * ww.w + bb.w);
}

// bn_apply + MX fp8 copy of the result (fp8.hip): 8 consecutive threads hold one 32-element block of a row (D % 32 == 0)
template <typename TO, typename TX>
__global__ __launch_bounds__(256) void bn_apply_q_kernel(const TX* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ w, const float* __restrict__ b, TO* __restrict__ y,
                                                         uint8_t* __restrict__ q, uint8_t* __restrict__ sc, long long n4, int D) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    if (i < n4) {
        const int c = (int)((i * 4) % D);
        const float4 v = ld4f(x + i * 4);
        const float4 mu = *(const float4*)(mean + c), rs = *(const float4*)(rstd + c), ww = *(const float4*)(w + c), bb = *(const float4*)(b + c);
        o[0] = (v.x - mu.x) * rs.x * ww.x + bb.x; o[1] = (v.y - mu.y) * rs.y * ww.y + bb.y;
        o[2] = (v.z - mu.z) * rs.z * ww.z + bb.z; o[3] = (v.w - mu.w) * rs.w * ww.w + bb.w;
        stq<TO>(y, i * 4 + 0, o[0]); stq<TO>(y, i * 4 + 1, o[1]); stq<TO>(y, i * 4 + 2, o[2]); stq<TO>(y, i * 4 + 3, o[3]);
    }
    float amax = fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3])));
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64)); amax = fmaxf(amax, __shfl_xor(amax, 2, 64)); amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
    if (i >= n4) return;
    const unsigned bits = __float_as_uint(amax);
    const int ex = (int)((bits >> 23) & 0xFF);
    int E = ex - 127 - 8;
    if (E < -127) E = -127;
    if (amax * __uint_as_float((unsigned)(127 - E) << 23) > 448.f && E < 127) ++E;   // (fp8.hip mx_scale_byte: nothing saturates)
    const unsigned sb = ex == 0 ? 0u : (unsigned)(E + 127);
    const float inv = __uint_as_float((unsigned)(254 - (int)sb) << 23);
    float a0 = fminf(fmaxf(o[0] * inv, -448.f), 448.f), a1 = fminf(fmaxf(o[1] * inv, -448.f), 448.f);
    float a2 = fminf(fmaxf(o[2] * inv, -448.f), 448.f), a3 = fminf(fmaxf(o[3] * inv, -448.f), 448.f);
    const unsigned lo = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a0, a1, 0, false) & 0xFFFFu;
    const unsigned hi = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a2, a3, 0, false) & 0xFFFFu;
    *(unsigned*)(q + i * 4) = lo | (hi << 16);
    if ((i & 7) == 0) sc[i >> 3] = (uint8_t)sb;     // element index 4 i -> block (4 i) / 32
}

size_t bn_partial_floats(long long M, int D) { return (size_t)((M + BN_ROWS - 1) / BN_ROWS) * 2 * D; }

template <typename TX>
static int batchnorm_fwd_t(const TX* x, const float* w, const float* b, float* run_mean, float* run_var, int train, float eps, void* y,
                           int y_dtype, float* mean, float* rstd, float* partials, long long M, int D, hipStream_t s, void* q8, void* q8_scales) {
    NBCI_REQUIRE(D % 4 == 0, NBCI_ESHAPE, "batchnorm: features must be a multiple of 4");
    const int nchunks = (int)((M + BN_ROWS - 1) / BN_ROWS);
    if (train) {
        hipLaunchKernelGGL((bn_stats_kernel<TX>), dim3(nchunks, (D + 255) / 256), dim3(256), 0, s, x, partials, M, D);
        int rc = check_launch("bn_stats");
        if (rc != NBCI_OK) return rc;
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((D + BN_FG - 1) / BN_FG), dim3(256), 0, s, partials, nchunks, M, D, eps, mean, rstd, run_mean, run_var, train);
    int rc = check_launch("bn_finalize");
    if (rc != NBCI_OK) return rc;
    const long long n4 = M * D / 4;
    dim3 g((unsigned)((n4 + 255) / 256));
    if (q8) {   // also the MX fp8 copy the block-scaled QKV GEMM reads (blocks of 32 along the feature axis)
        NBCI_REQUIRE(D % 32 == 0 && q8_scales, NBCI_ESHAPE, "batchnorm: the fp8 copy needs features in multiples of 32");
        if (y_dtype == NBCI_BF16) hipLaunchKernelGGL((bn_apply_q_kernel<bf16_t, TX>), g, dim3(256), 0, s, x, mean, rstd, w, b, (bf16_t*)y, (uint8_t*)q8, (uint8_t*)q8_scales, n4, D);
        else hipLaunchKernelGGL((bn_apply_q_kernel<float, TX>), g, dim3(256), 0, s, x, mean, rstd, w, b, (float*)y, (uint8_t*)q8, (uint8_t*)q8_scales, n4, D);
        return check_launch("bn_apply_q");
    }
    if (y_dtype == NBCI_BF16) hipLaunchKernelGGL((bn_apply_kernel<bf16_t, TX>), g, dim3(256), 0, s, x, mean, rstd, w, b, (bf16_t*)y, n4, D);
    else hipLaunchKernelGGL((bn_apply_kernel<float, TX>), g, dim3(256), 0, s, x, mean, rstd, w, b, (float*)y, n4, D);
    return check_launch("bn_apply");
}
int batchnorm_fwd_launch(const void* x, const float* w, const float* b, float* run_mean, float* run_var, int train, float eps, void* y,
                         int y_dtype, float* mean, float* rstd, float* partials, long long M, int D, hipStream_t s, void* q8, void* q8_scales,
                         int x_dtype) {
    if (x_dtype == NBCI_BF16)
        return batchnorm_fwd_t<bf16_t>((const bf16_t*)x, w, b, run_mean, run_var, train, eps, y, y_dtype, mean, rstd, partials, M, D, s, q8, q8_scales);
    return batchnorm_fwd_t<float>((const float*)x, w, b, run_mean, run_var, train, eps, y, y_dtype, mean, rstd, partials, M, D, s, q8, q8_scales);
}

// backward pass 1: per (chunk, column) sums of dy and dy * xhat
template <typename TY, typename TX>   // TY: the incoming gradient (f32, or bf16 as a bf16 data-gradient GEMM writes it); TX: the saved input
__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const TY* __restrict__ dy, const TX* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, float* __restrict__ part, long long M, int D) {
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= D) return;
    const long long r0 = (long long)blockIdx.x * BN_ROWS;
    const long long r1 = r0 + BN_ROWS < M ? r0 + BN_ROWS : M;
    const float mu = mean[c], rs = rstd[c];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
    for (long long r = r0; r < r1; ++r) { const float g = (float)dy[r * D + c]; s1 += g; s2 += g * ((float)x[r * D + c] - mu) * rs; }
    float* o = part + ((long long)blockIdx.x * 2) * D;
    o[c] = s1; o[D + c] = s2;
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ part, int nchunks, int D, float* __restrict__ coef,
                                                              float* __restrict__ dw, float* __restrict__ db, const float* __restrict__ w,
                                                              const float* __restrict__ rstd, float invM, int train) {
    __shared__ double sa[BN_FG][BN_FG], sb[BN_FG][BN_FG];   // (16 channels x 16 groups per block, fixed combination order: see bn_finalize_kernel)
    const int cl = threadIdx.x % BN_FG, g = threadIdx.x / BN_FG;
    const int c = blockIdx.x * BN_FG + cl;
    double a = 0.0, b = 0.0;
    if (c < D) {
#pragma unroll 4
        for (int k = g; k < nchunks; k += BN_FG) { a += part[((long long)k * 2) * D + c]; b += part[((long long)k * 2 + 1) * D + c]; }
    }
    sa[g][cl] = a; sb[g][cl] = b;
    __syncthreads();
    if (g != 0 || c >= D) return;
    for (int j = 1; j < BN_FG; ++j) { a += sa[j][cl]; b += sb[j][cl]; }
    db[c] += (float)a; dw[c] += (float)b;
    // per-column coefficients of pass 2: dx += A dy - B - C (x - mean)
    const float wr = w[c] * rstd[c];
    coef[c] = wr;
    coef[D + c] = train ? wr * (float)a * invM : 0.f;
    coef[2 * D + c] = train ? wr * rstd[c] * (float)b * invM : 0.f;
}

// pass 2: dx += w * rstd * (dy - sum_dy / M - xhat * sum_dyxhat / M)   (eval mode: dx += dy * w * rstd), as dx += A dy - B - C (x - mean)
// with the per-column A, B, C of the finalize kernel: four vector loads of column data per thread instead of twenty scalar ones, two
// float4 elements per thread in flight
template <typename TY, typename TX, typename TD>   // TD: the gradient stream dx (read, added to, written: f32 arithmetic, one rounding)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const TY* __restrict__ dy, const TX* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ coef, TD* __restrict__ dx, long long n4, int D) {
    const long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 2;
    float4 g[2], v[2], o[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const long long i = i0 + u;
        if (i < n4) { g[u] = ld4f(dy + i * 4); v[u] = ld4f(x + i * 4); o[u] = ld4f(dx + i * 4); }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const long long i = i0 + u;
        if (i >= n4) break;
        const int c = (int)((i * 4) % D);
        const float4 A = *(const float4*)(coef + c), B = *(const float4*)(coef + D + c), Cc = *(const float4*)(coef + 2 * D + c), mu = *(const float4*)(mean + c);
        o[u].x += A.x * g[u].x - B.x - Cc.x * (v[u].x - mu.x);
        o[u].y += A.y * g[u].y - B.y - Cc.y * (v[u].y - mu.y);
        o[u].z += A.z * g[u].z - B.z - Cc.z * (v[u].z - mu.z);
        o[u].w += A.w * g[u].w - B.w - Cc.w * (v[u].w - mu.w);
        st4f(dx + i * 4, o[u]);
    }
}

template <typename TY, typename TX, typename TD>
static int batchnorm_bwd_t(const TY* dy, const TX* x, const float* mean, const float* rstd, const float* w, TD* dx, float* dw, float* db,
                           float* partials, float* sums, long long M, int D, int train, hipStream_t s) {
    const int nchunks = (int)((M + BN_ROWS - 1) / BN_ROWS);
    hipLaunchKernelGGL((bn_bwd_stats_kernel<TY, TX>), dim3(nchunks, (D + 255) / 256), dim3(256), 0, s, dy, x, mean, rstd, partials, M, D);
    int rc = check_launch("bn_bwd_stats");
    if (rc != NBCI_OK) return rc;
    NBCI_REQUIRE(D % 4 == 0, NBCI_ESHAPE, "batchnorm backward: features must be a multiple of 4");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((D + BN_FG - 1) / BN_FG), dim3(256), 0, s, partials, nchunks, D, sums, dw, db, w, rstd, 1.0f / (float)M,
                       train);
    rc = check_launch("bn_bwd_finalize");
    if (rc != NBCI_OK) return rc;
    const long long n4 = M * D / 4;
    hipLaunchKernelGGL((bn_bwd_apply_kernel<TY, TX, TD>), dim3((unsigned)((n4 + 511) / 512)), dim3(256), 0, s, dy, x, mean, sums, dx, n4, D);
    return check_launch("bn_bwd_apply");
}
// dy_dtype: the incoming gradient; stream_dtype: the saved input x AND the gradient stream dx (both f32, or both bf16)
int batchnorm_bwd_launch(const void* dy, const void* x, const float* mean, const float* rstd, const float* w, void* dx, float* dw, float* db,
                         float* partials, float* sums, long long M, int D, int train, hipStream_t s, int dy_dtype, int stream_dtype) {
    if (stream_dtype == NBCI_BF16) {
        NBCI_REQUIRE(dy_dtype == NBCI_BF16, NBCI_EINVAL, "batchnorm backward: bf16 streams take a bf16 incoming gradient");
        return batchnorm_bwd_t<bf16_t, bf16_t, bf16_t>((const bf16_t*)dy, (const bf16_t*)x, mean, rstd, w, (bf16_t*)dx, dw, db, partials, sums, M, D, train, s);
    }
    if (dy_dtype == NBCI_BF16)
        return batchnorm_bwd_t<bf16_t, float, float>((const bf16_t*)dy, (const float*)x, mean, rstd, w, (float*)dx, dw, db, partials, sums, M, D, train, s);
    return batchnorm_bwd_t<float, float, float>((const float*)dy, (const float*)x, mean, rstd, w, (float*)dx, dw, db, partials, sums, M, D, train, s);
}

// ------------------------------------------------------------------------------------------
// PredictHead pooling (patchtst.py:89): pooled[(b,p), :] = mean_c h[b,c,p,:]; backward broadcasts d/C to every channel
// ------------------------------------------------------------------------------------------
// Two passes, both deterministic: (1) thread (chunk, b, p, d/4) sums ITS chunk of the channels into partial[chunk][...]; (2) thread
// (b, p, d/4) adds the chunks in order, scales by 1 / C and stores. One thread walking all C = 1024 channels of its (b, p, d/4) - 26 k
// threads chip-wide, a dependent load per channel - took 411 us for 215 MB.
constexpr int POOL_CHUNKS = 32;
template <typename TH>
__global__ __launch_bounds__(256) void ptst_pool_part_kernel(const TH* __restrict__ h, float* __restrict__ partial, int B, int C, int P, int D, int cpc) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // (b, p, d/4)
    const int dq = D / 4;
    const long long n = (long long)B * P * dq;
    if (i >= n) return;
    const int d = (int)(i % dq) * 4;
    const long long bp = i / dq;
    const int p = (int)(bp % P), b = (int)(bp / P);
    const int c0 = blockIdx.y * cpc, c1 = min(C, c0 + cpc);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int c = c0; c < c1; ++c) {
        const float4 v = ld4f(h + (((long long)b * C + c) * P + p) * D + d);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    *(float4*)(partial + ((long long)blockIdx.y * n + i) * 4) = a;
}
template <typename TO>
__global__ __launch_bounds__(256) void ptst_pool_sum_kernel(const float* __restrict__ partial, TO* __restrict__ pooled, long long n, int nchunks, float inv) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < nchunks; ++k) {
        const float4 v = *(const float4*)(partial + ((long long)k * n + i) * 4);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    stq<TO>(pooled, i * 4 + 0, a.x * inv); stq<TO>(pooled, i * 4 + 1, a.y * inv); stq<TO>(pooled, i * 4 + 2, a.z * inv); stq<TO>(pooled, i * 4 + 3, a.w * inv);
}

template <typename TD>
__global__ __launch_bounds__(256) void ptst_pool_bwd_kernel(const float* __restrict__ dpooled, TD* __restrict__ dh, int B, int C, int P, int D) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // (b, c, p, d/4)
    const int dq = D / 4;
    if (i >= (long long)B * C * P * dq) return;
    const int d = (int)(i % dq) * 4;
    const long long row = i / dq;
    const int p = (int)(row % P);
    const int b = (int)(row / ((long long)C * P));
    const float4 v = *(const float4*)(dpooled + ((long long)b * P + p) * D + d);
    const float inv = 1.0f / (float)C;
    st4f(dh + row * D + d, make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv));
}

size_t ptst_pool_partial_floats(int B, int P, int D) { return (size_t)POOL_CHUNKS * B * P * D; }
int ptst_pool_fwd_launch(const void* h, void* pooled, int dtype, int B, int C, int P, int D, hipStream_t s, int h_dtype, float* partial) {
    NBCI_REQUIRE(partial && D % 4 == 0, NBCI_EINVAL, "pool: scratch for the channel-chunk partial sums is required");
    const long long n = (long long)B * P * (D / 4);
    const int cpc = (C + POOL_CHUNKS - 1) / POOL_CHUNKS, nchunks = (C + cpc - 1) / cpc;
    dim3 g((unsigned)((n + 255) / 256), (unsigned)nchunks);
    if (h_dtype == NBCI_BF16) hipLaunchKernelGGL((ptst_pool_part_kernel<bf16_t>), g, dim3(256), 0, s, (const bf16_t*)h, partial, B, C, P, D, cpc);
    else hipLaunchKernelGGL((ptst_pool_part_kernel<float>), g, dim3(256), 0, s, (const float*)h, partial, B, C, P, D, cpc);
    int rc = check_launch("ptst_pool_part");
    if (rc != NBCI_OK) return rc;
    const float inv = 1.0f / (float)C;
    if (dtype == NBCI_BF16) hipLaunchKernelGGL((ptst_pool_sum_kernel<bf16_t>), dim3(g.x), dim3(256), 0, s, partial, (bf16_t*)pooled, n, nchunks, inv);
    else hipLaunchKernelGGL((ptst_pool_sum_kernel<float>), dim3(g.x), dim3(256), 0, s, partial, (float*)pooled, n, nchunks, inv);
    return check_launch("ptst_pool_sum");
}

int ptst_pool_bwd_launch(const float* dpooled, void* dh, int B, int C, int P, int D, hipStream_t s, int dh_dtype) {
    const long long n = (long long)B * C * P * (D / 4);
    if (dh_dtype == NBCI_BF16) hipLaunchKernelGGL((ptst_pool_bwd_kernel<bf16_t>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dpooled, (bf16_t*)dh, B, C, P, D);
    else hipLaunchKernelGGL((ptst_pool_bwd_kernel<float>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dpooled, (float*)dh, B, C, P, D);
    return check_launch("ptst_pool_bwd");
}

// patchtst.py:239: lens = trunc(1 + (len - patch_length) / patch_stride)
__global__ void ptst_lens_kernel(const long long* __restrict__ lens, int* __restrict__ out, int B, int pl, int stride) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i < B) out[i] = (int)truncf(1.0f + ((float)lens[i] - (float)pl) / (float)stride);
}

int ptst_lens_launch(const int64_t* lens, int32_t* out, int B, int pl, int stride, hipStream_t s) {
    hipLaunchKernelGGL(ptst_lens_kernel, dim3((B + 63) / 64), dim3(64), 0, s, (const long long*)lens, out, B, pl, stride);
    return check_launch("ptst_lens");
}

// ------------------------------------------------------------------------------------------
// PretrainHead tail + masked loss (patchtst.py:139-154, 225-231): rows (b,c,p), pl values each.
// row mask = model mask & (spikes_mask windows starting at bin 0 — the reference's unfold, NOT the patchifier's start).
// ------------------------------------------------------------------------------------------
template <typename TO>
__global__ __launch_bounds__(256) void ptst_mlm_loss_kernel(const float* __restrict__ pred, int ldp, const float* __restrict__ target,
                                                            const unsigned char* __restrict__ mask, const long long* __restrict__ smask,
                                                            float* __restrict__ preds_out, unsigned char* __restrict__ mask_out,
                                                            TO* __restrict__ dpred, float* __restrict__ loss, unsigned long long* __restrict__ nex,
                                                            long long M, int T, int C, int P, int pl, int stride, int kind, float gscale) {
    __shared__ float red[4];
    __shared__ unsigned cnt[4];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // (row, j)
    float lsum = 0.f;
    unsigned lcnt = 0;
    if (i < M * pl) {
        const long long row = i / pl;
        const int j = (int)(i - row * pl);
        const int p = (int)(row % P);
        const int b = (int)(row / ((long long)C * P));
        bool valid = true;
        for (int q = 0; q < pl; ++q) valid = valid && smask[(long long)b * T + p * stride + q] != 0;
        const bool m = mask[row] && valid;
        const float raw = pred[row * ldp + j], y = target[i];
        float pr = raw, el, dl;
        if (kind == NBCI_LOSS_POISSON_LOG) { const float e = expf(pr); el = e - y * pr; dl = e - y; }
        else if (kind == NBCI_LOSS_POISSON_RATE) { pr = fmaxf(raw, 0.f); el = pr - y * logf(pr + 1e-8f); dl = raw > 0.f ? 1.f - y / (pr + 1e-8f) : 0.f; }
        else { const float df = pr - y; el = df * df; dl = 2.f * df; }
        preds_out[i] = pr;
        if (j == 0) { mask_out[row] = m; lcnt = m; }
        if (m) lsum = el;
        if (dpred) stq<TO>(dpred, row * ldp + j, m ? dl * gscale : 0.f);
    }
    lsum = wave_sum(lsum);
    const float fc = wave_sum((float)lcnt);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = lsum; cnt[threadIdx.x >> 6] = (unsigned)fc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(loss, red[0] + red[1] + red[2] + red[3]);
        atomicAdd(nex, (unsigned long long)(cnt[0] + cnt[1] + cnt[2] + cnt[3]));
    }
}

int ptst_mlm_loss_launch(const float* pred, int ldp, const float* target, const uint8_t* mask, const int64_t* smask, float* preds_out,
                         uint8_t* mask_out, void* dpred, int d_dtype, float* loss, int64_t* n_examples, int B, int T, int C, int P, int pl,
                         int stride, int kind, float grad_scale, hipStream_t s) {
    NBCI_REQUIRE(kind >= NBCI_LOSS_POISSON_LOG && kind <= NBCI_LOSS_MSE, NBCI_EINVAL, "ptst mlm loss: unknown loss kind");
    const long long M = (long long)B * C * P;
    NBCI_CHECK_HIP(hipMemsetAsync(loss, 0, 4, s));
    NBCI_CHECK_HIP(hipMemsetAsync(n_examples, 0, 8, s));
    if (dpred) NBCI_CHECK_HIP(hipMemsetAsync(dpred, 0, (size_t)M * ldp * (d_dtype == NBCI_BF16 ? 2 : 4), s));
    dim3 g((unsigned)((M * pl + 255) / 256));
    if (d_dtype == NBCI_BF16)
        hipLaunchKernelGGL((ptst_mlm_loss_kernel<bf16_t>), g, dim3(256), 0, s, pred, ldp, target, mask, (const long long*)smask, preds_out, mask_out,
                           (bf16_t*)dpred, loss, (unsigned long long*)n_examples, M, T, C, P, pl, stride, kind, grad_scale);
    else
        hipLaunchKernelGGL((ptst_mlm_loss_kernel<float>), g, dim3(256), 0, s, pred, ldp, target, mask, (const long long*)smask, preds_out, mask_out,
                           (float*)dpred, loss, (unsigned long long*)n_examples, M, T, C, P, pl, stride, kind, grad_scale);
    return check_launch("ptst_mlm_loss");
}

}  // namespace nbci
