// attention.hip — fused masked attention for the NDT1 token counts (T' <= 160, head 128, bf16):
// replaces F.scaled_dot_product_attention(q, k, v, attn_mask, dropout_p) of the reference
// (models/ndt1.py:289) and its autograd, including the mask of ndt1.py:435-437 (eye | ctx & key
// validity, computed on the fly) and the attention-prob dropout.
//
// One workgroup (8 waves) per (batch, head). K and V of that head live in LDS as 256-byte-row
// images that serve both ds_read_b128 row reads and ds_read_b64_tr_b16 transposed reads
// (XOR swizzle (b) of the CDNA4 guide). Each wave owns 16-query blocks; the whole score row
// (<= 160 keys) stays in registers, so softmax needs only two cross-lane steps and P never
// leaves registers: the QK^T accumulator IS the B operand of the PV product once the k-order
// permutation pi(g, j) = {4g+j | 16+4g+(j-4)} is applied to the V operand's row addresses.
//   fwd     : S = q k^T, masked softmax, dropout, O = Pd v   -> ad (merged layout) + row log-sum-exp (f32)
//   bwd_dq  : 32 keys at a time: P = exp(S - lse); dP = (da v^T) * keep; dS = P (dP - delta) / sqrt(d) with
//             delta = rowsum(dP P) = da . O taken from the stored forward output (no full score row in registers:
//             nine waves fit, one per 16-query block); dq = dS k -> dqkv; stores dS and Pd (bf16) for the second kernel
//   bwd_dkv : dk = dS^T q, dv = Pd^T da  (contraction over queries: both operands tr-read)
// Larger T', other head sizes and the f32 parity path use the batched-GEMM + softmax kernels.
#include <cstdlib>
#include <cstring>

#include "kernels.h"

namespace nbci {

constexpr int AT_TPAD = 160;   // padded keys / queries (10 blocks of 16, 5 MFMA k-steps of 32)
constexpr int AT_NB = 10;
constexpr int AT_HD = 128;
constexpr int AT_THREADS = 512;

// byte offset of 16-byte chunk `ch` (0..15) of row `row` in a 256-byte-row image
__device__ __forceinline__ int img_off(int row, int ch) {
    return row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
}

// global (rows x 128 bf16, row stride ld elements) -> LDS image, rows >= nrows zero-filled
__device__ __forceinline__ void load_image(char* img, const bf16_t* src, long long ld, int nrows, int tid) {
    for (int i = tid; i < AT_TPAD * 16; i += (int)blockDim.x) {
        const int row = i >> 4, ch = i & 15;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (row < nrows) v = *(const uint4*)(src + (long long)row * ld + ch * 8);
        *(uint4*)(img + img_off(row, ch)) = v;
    }
}

// per-workgroup key table in LDS: 1 = the key exists and is unmasked. (Read from global per score element it was 40 dependent
// loads per lane and the longest phase of the forward kernel: 6.2 of 17 us per workgroup, in-kernel stamps, tools/attn_stamps.py.)
__device__ __forceinline__ void load_key_valid(int* kv, const int32_t* tmask_row, int Tp, int tid) {
    for (int i = tid; i < AT_TPAD; i += (int)blockDim.x) kv[i] = (i < Tp && tmask_row[i] != 0) ? 1 : 0;
}

// K and V images together, every global load of a pass issued before the first LDS store. (One load -> one store per
// iteration, as load_image does, exposed a memory round trip per 16-byte chunk: 4.2 of the forward kernel's 15 us.)
__device__ __forceinline__ void load_images_kv(char* sK, char* sV, const bf16_t* srcK, const bf16_t* srcV, long long ld, int nrows, int tid, int krows = AT_TPAD) {
    constexpr int U = 5;   // 2560 chunks per image / 576 threads (143 tokens -> 9 waves)
    const int nthr = (int)blockDim.x;
    for (int i0 = tid; i0 < AT_TPAD * 16; i0 += U * nthr) {
        uint4 vk[U], vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * nthr, row = i >> 4, ch = i & 15;
            vk[u] = make_uint4(0u, 0u, 0u, 0u);
            vv[u] = make_uint4(0u, 0u, 0u, 0u);
            if (i < AT_TPAD * 16 && row < nrows) {
                vk[u] = *(const uint4*)(srcK + (long long)row * ld + ch * 8);
                vv[u] = *(const uint4*)(srcV + (long long)row * ld + ch * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * nthr, row = i >> 4, ch = i & 15;
            if (i < AT_TPAD * 16) {
                if (row < krows) *(uint4*)(sK + img_off(row, ch)) = vk[u];   // (the 9-block forward keeps a 144-row K image)
                *(uint4*)(sV + img_off(row, ch)) = vv[u];
            }
        }
    }
}

__device__ __forceinline__ bf16x8 row_frag(const char* img, int row, int ks, int g) {
    return *(const bf16x8*)(img + img_off(row, 4 * ks + g));
}

// transposed fragment: 8 image ROWS (two groups of 4, given by r_lo / r_hi for this lane's group)
// x 16 columns starting at element column c0; lane i of each 16-lane group receives column c0 + i.
__device__ __forceinline__ bf16x8 tr_frag(const char* img, int r_lo, int r_hi, int c0, int i16) {
    const int q = i16 >> 2, p = i16 & 3;
    const int col = c0 + 4 * p;                 // element column of this lane's address
    const int ch = col >> 3, within = (col & 7) * 2;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(img + img_off(r_lo + q, ch) + within));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(img + img_off(r_hi + q, ch) + within));
    union { struct { s16x4 a, b; } p2; bf16x8 v; } u;
    u.p2.a = lo; u.p2.b = hi;
    return u.v;
}

__device__ __forceinline__ bool ctx_ok(int i, int j, int f, int bk) {
    if (f >= -1 && j - i > f) return false;
    if (bk >= -1 && i - j > bk) return false;
    return true;
}

struct AttnArgs {
    const bf16_t* qkv;     // (B*Tp, 3H)
    const int32_t* tmask;  // (B, Tp)
    int B, nh, Tp, H;
    int cf, cb;
    float scale;
    unsigned p_thr; float p_scale; uint32_t p_key;   // attention-prob dropout
    unsigned o_thr; float o_scale; uint32_t o_key;   // attention-output dropout (forward only)
    bf16_t* ad;            // fwd out / bwd in (B*Tp, H): dropout(merge_heads(Pd v))
    float* lse;            // fwd out / bwd in (B, nh, Tp): log-sum-exp of the scaled, masked score row
    const bf16_t* da;      // bwd in  (B*Tp, H): d loss / d (Pd v)
    bf16_t* dS;            // bwd scratch (B, nh, Tp, ldP)
    bf16_t* Pd;            // bwd scratch (B, nh, Tp, ldP)
    int ldP;
    bf16_t* dqkv;          // bwd out (B*Tp, 3H)
    float* bias_grad;      // optional (3H): column sums of dqkv
    RepCfg rc;             // replication of bias_grad
};

// scores of one 16-query block against all keys: acc[kb][r] = S[query i16][key 16kb + 4g + r]
template <int NB>
__device__ __forceinline__ void score_block(const char* sK, const bf16x8 (&qf)[4], f32x4 (&acc)[NB], int i16, int g) {
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
        acc[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            acc[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(sK, 16 * kb + i16, ks, g), qf[ks], acc[kb], 0, 0, 0);
        if (kb & 1) __builtin_amdgcn_sched_barrier(0);  // keep the LDS fragment loads from being hoisted en bloc (VGPR pressure)
    }
}

// masked softmax of the register-resident row; returns normalised probabilities in place.
// kbits: the workgroup's key-valid bits (bit k of word k / 32). Each lane first gathers the bits of ITS keys (block kb, keys
// 16 kb + 4 g + r -> bit 4 kb + r of vlo / vhi) and sets the diagonal's; per element the test is then one bit (FULLCTX: the context
// span is unbounded, configs/ndt1.yaml) - the int table + five compares per element this replaces were a third of the kernel's VALU work.
template <int NB, bool FULLCTX>
__device__ __forceinline__ void softmax_rows(f32x4 (&acc)[NB], const AttnArgs& a, const unsigned* kbits, int query, int qb, int i16, int g, float* lse_out) {
    unsigned vlo = 0u, vhi = 0u;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
        const unsigned nib = (kbits[kb >> 1] >> (16 * (kb & 1) + 4 * g)) & 0xFu;
        if (kb < 8) vlo |= nib << (4 * kb); else vhi |= nib << (4 * (kb - 8));
    }
    if ((i16 >> 2) == g) {   // the diagonal is always attendable (ndt1.py:436)
        if (qb < 8) vlo |= 1u << (4 * qb + (i16 & 3)); else vhi |= 1u << (4 * (qb - 8) + (i16 & 3));
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
        const unsigned nib = kb < 8 ? (vlo >> (4 * kb)) : (vhi >> (4 * (kb - 8)));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            bool ok = ((nib >> r) & 1u) != 0u;
            if (!FULLCTX) { const int key = 16 * kb + 4 * g + r; ok = ok && ((key == query) || ctx_ok(query, key, a.cf, a.cb)); }
            const float s = ok ? acc[kb][r] : -INFINITY;   // (raw scores: the positive scale is applied inside the exponent)
            acc[kb][r] = s;
            mx = fmaxf(mx, s);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float c2 = a.scale * 1.4426950408889634f, nm = -mx * c2;   // exp((s - mx) * scale) = exp2(s * c2 - mx * c2); exp2(-inf) = 0
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(fmaf(acc[kb][r], c2, nm));
            acc[kb][r] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    if (lse_out && g == 0) *lse_out = mx * a.scale + __logf(sum);   // (the diagonal is always attendable: sum > 0)
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[kb][r] *= inv;
}

__device__ __forceinline__ bf16x8 pack8(const f32x4& lo, const f32x4& hi) {
    bf16x8 o = {f2bf(lo[0]), f2bf(lo[1]), f2bf(lo[2]), f2bf(lo[3]), f2bf(hi[0]), f2bf(hi[1]), f2bf(hi[2]), f2bf(hi[3])};
    return o;
}

#ifdef NBCI_STAMPS   // measurement build only (tools/attn_stamps.py): wall-clock stamps (100 MHz) of wave 0's phases per workgroup
static __device__ unsigned long long g_astamps[1024 * 8];
#define ASTAMP(slot) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_astamps[blockIdx.x * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define ASTAMP(slot) do { } while (0)
#endif

// NB = 16-key blocks held per score row: 10 (T' <= 160), or 9 (T' <= 144) with a 144-row K image and at most 102 registers, so that TWO
// workgroups (18 waves) share a CU and all B x heads workgroups run in one round
template <int NB, bool FULLCTX>
__global__ __launch_bounds__(640, NB == 9 ? 5 : 3) void attn_fwd_kernel(AttnArgs a) {   // one wave per 16-query block (<= 10 waves)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    ASTAMP(0);
    char* sK = smem;
    char* sV = smem + NB * 16 * 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / a.nh, h = blockIdx.x % a.nh;
    const long long ld = 3LL * a.H;
    const bf16_t* base = a.qkv + (long long)b * a.Tp * ld + h * AT_HD;
    load_images_kv(sK, sV, base + a.H, base + 2 * a.H, ld, a.Tp, tid, NB * 16);
    unsigned* sbits = (unsigned*)(smem + (NB * 16 + AT_TPAD) * 256);   // [6] key-valid bits
    if (tid < 64) {
        const int32_t* tm = a.tmask + (long long)b * a.Tp;
#pragma unroll
        for (int w = 0; w < 3; ++w) {
            const int k = 64 * w + lane;
            const unsigned long long m = __ballot(k < a.Tp && tm[k < a.Tp ? k : 0] != 0);
            if (lane == 0) { sbits[2 * w] = (unsigned)m; sbits[2 * w + 1] = (unsigned)(m >> 32); }
        }
    }
    __syncthreads();
    ASTAMP(1);
    const int nqb = (a.Tp + 15) / 16;
    for (int qb = wave; qb < nqb; qb += (int)(blockDim.x >> 6)) {
        const int query = 16 * qb + i16;
        const int qrow = query < a.Tp ? query : a.Tp - 1;
        bf16x8 qf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(base + (long long)qrow * ld + 32 * ks + 8 * g);
        f32x4 acc[NB];
        score_block<NB>(sK, qf, acc, i16, g);
        ASTAMP(2);
        softmax_rows<NB, FULLCTX>(acc, a, sbits, qrow, qrow >> 4, qrow & 15, g, a.lse ? a.lse + (long long)blockIdx.x * a.Tp + qrow : nullptr);
        ASTAMP(3);
        if (a.p_thr) {
            const unsigned rbase = (unsigned)(((long long)blockIdx.x * a.Tp + qrow) * a.Tp);
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) {   // (keys >= T' hold p = 0 already; their draws are never looked at)
                float keep[4];
                drop4_any(a.p_key, a.p_thr, rbase + (unsigned)(16 * kb + 4 * g), a.p_scale, keep);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[kb][r] *= keep[r];
            }
        }
        f32x4 o[8];
        ASTAMP(4);
#pragma unroll
        for (int db = 0; db < 8; ++db) o[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < (NB + 1) / 2; ++s) {
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            const bf16x8 pf = pack8(acc[2 * s], 2 * s + 1 < NB ? acc[2 * s + 1] : zero4);   // (NB = 9: the V image keeps 160 zero-padded rows)
#pragma unroll
            for (int db = 0; db < 8; ++db)
                o[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(sV, 32 * s + 4 * g, 32 * s + 16 + 4 * g, 16 * db, i16), pf,
                                                                o[db], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        ASTAMP(5);
        if (query < a.Tp) {
            const long long obase = ((long long)b * a.Tp + query) * a.H + h * AT_HD;
#pragma unroll
            for (int db = 0; db < 8; ++db) {
                float v[4] = {o[db][0], o[db][1], o[db][2], o[db][3]};
                const long long oi = obase + 16 * db + 4 * g;
                if (a.o_thr) drop4(a.o_key, a.o_thr, (unsigned)oi, a.o_scale, v);
                bf16x4 ov = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
                *(bf16x4*)(a.ad + oi) = ov;
            }
        }
    }
    ASTAMP(6);
}

__global__ __launch_bounds__(640) void attn_bwd_dq_kernel(AttnArgs a) {   // one wave per 16-query block (<= 10 waves)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;
    char* sV = smem + AT_TPAD * 256;
    float* sbias = (float*)(smem + 2 * AT_TPAD * 256);  // [128] per-workgroup column sums (bias gradient)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < AT_HD) sbias[tid] = 0.f;
    const int i16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / a.nh, h = blockIdx.x % a.nh;
    const long long ld = 3LL * a.H;
    const bf16_t* base = a.qkv + (long long)b * a.Tp * ld + h * AT_HD;
    load_images_kv(sK, sV, base + a.H, base + 2 * a.H, ld, a.Tp, tid);
    __syncthreads();
    const int nqb = (a.Tp + 15) / 16;
    const float inv_o_scale = 1.0f / a.o_scale;
    for (int qb = wave; qb < nqb; qb += (int)(blockDim.x >> 6)) {
        const int query = 16 * qb + i16;
        const int qrow = query < a.Tp ? query : a.Tp - 1;
        bf16x8 qf[4], df[4];
        float delta = 0.f;
        {
            const long long arow = ((long long)b * a.Tp + qrow) * a.H + h * AT_HD;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                qf[ks] = *(const bf16x8*)(base + (long long)qrow * ld + 32 * ks + 8 * g);
                df[ks] = *(const bf16x8*)(a.da + arow + 32 * ks + 8 * g);
                const bf16x8 of = *(const bf16x8*)(a.ad + arow + 32 * ks + 8 * g);
#pragma unroll
                for (int e = 0; e < 8; ++e) delta += bf2f(df[ks][e]) * bf2f(of[e]);
            }
        }
        // delta = rowsum(dP P) = da . O ; the stored output is dropout(O) and da is zero wherever it was dropped
        delta += __shfl_xor(delta, 16, 64);
        delta += __shfl_xor(delta, 32, 64);
        delta *= inv_o_scale;
        const float lse = a.lse[(long long)blockIdx.x * a.Tp + qrow];
        const unsigned rbase = (unsigned)(((long long)blockIdx.x * a.Tp + qrow) * a.Tp);
        const long long srow = ((long long)blockIdx.x * a.Tp + query) * a.ldP;
        f32x4 o[8];
#pragma unroll
        for (int db = 0; db < 8; ++db) o[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < AT_NB / 2; ++s) {
            f32x4 sc[2], dp[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int kb = 2 * s + j;
                sc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                dp[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    sc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(sK, 16 * kb + i16, ks, g), qf[ks], sc[j], 0, 0, 0);
                    dp[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(sV, 16 * kb + i16, ks, g), df[ks], dp[j], 0, 0, 0);   // dPd = da . v^T
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int kb = 2 * s + j;
                float pdv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = 16 * kb + 4 * g + r;
                    // (the forward kernel's LDS key table does not pay here: the reads are hoisted and this kernel, at its 168-VGPR limit, spills)
                    const bool ok = key < a.Tp && ((key == qrow) || (ctx_ok(qrow, key, a.cf, a.cb) && a.tmask[b * a.Tp + key] != 0));
                    const float p = ok ? __expf(sc[j][r] * a.scale - lse) : 0.f;
                    float keep = (key < a.Tp) ? 1.f : 0.f;
                    if (a.p_thr && key < a.Tp) keep = drop_keep(a.p_key, a.p_thr, rbase + key) ? a.p_scale : 0.f;
                    pdv[r] = p * keep;                                        // Pd
                    dp[j][r] = p * (dp[j][r] * keep - delta) * a.scale;       // dS (scaled), in place
                }
                const int key0 = 16 * kb + 4 * g;
                if (query < a.Tp && key0 < a.ldP) {
                    bf16x4 pv = {f2bf(pdv[0]), f2bf(pdv[1]), f2bf(pdv[2]), f2bf(pdv[3])};
                    *(bf16x4*)(a.Pd + srow + key0) = pv;
                    bf16x4 sv = {f2bf(dp[j][0]), f2bf(dp[j][1]), f2bf(dp[j][2]), f2bf(dp[j][3])};
                    *(bf16x4*)(a.dS + srow + key0) = sv;
                }
            }
            // dq[query][d] += sum over these 32 keys of dS[query][key] k[key][d]   (k image read transposed)
            const bf16x8 sf = pack8(dp[0], dp[1]);
#pragma unroll
            for (int db = 0; db < 8; ++db)
                o[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(sK, 32 * s + 4 * g, 32 * s + 16 + 4 * g, 16 * db, i16), sf,
                                                                o[db], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        const long long obase = ((long long)b * a.Tp + qrow) * ld + h * AT_HD;
#pragma unroll
        for (int db = 0; db < 8; ++db) {
            bf16x4 ov = {f2bf(o[db][0]), f2bf(o[db][1]), f2bf(o[db][2]), f2bf(o[db][3])};
            if (query < a.Tp) *(bf16x4*)(a.dqkv + obase + 16 * db + 4 * g) = ov;
            if (a.bias_grad) {  // query-bias gradient: column sums of the stored dq
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = query < a.Tp ? bf2f(ov[r]) : 0.f;
                    t = row16_sum(t);
                    if (i16 == 0) atomicAdd(&sbias[16 * db + 4 * g + r], t);   // LDS atomic
                }
            }
        }
    }
    if (a.bias_grad) {  // one contiguous global atomic pass per workgroup
        __syncthreads();
        if (tid < AT_HD) atomicAdd(rep_ptr(a.bias_grad, a.rc, blockIdx.x) + h * AT_HD + tid, sbias[tid]);
    }
}

// X image: [160 query rows][160 keys] bf16, 320-byte rows, 16-byte chunk ^= f(row) within each 64-B group
__device__ __forceinline__ int ximg_off(int row, int col) {  // byte offset of element (row, col)
    const int ch = col >> 3;
    return row * 320 + (((ch & ~3) | ((ch & 3) ^ (row & 3))) << 4) + (col & 7) * 2;
}

// grid (B*nh, 2): z = 0 -> dk = dS^T q ; z = 1 -> dv = Pd^T da.  out[key][d] = sum_q X[q][key] Y[q][d]
__global__ __launch_bounds__(640) void attn_bwd_dkv_kernel(AttnArgs a) {   // one wave per 16-key block (<= 10 waves)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sY = smem;                          // [160][128] image
    char* sX = smem + AT_TPAD * 256;          // [160][160]
    float* sbias = (float*)(smem + AT_TPAD * 256 + AT_TPAD * 320);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < AT_HD) sbias[tid] = 0.f;
    const int i16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / a.nh, h = blockIdx.x % a.nh;
    const int which = blockIdx.y;
    const long long ld3 = 3LL * a.H;
    const bf16_t* X = (which == 0 ? a.dS : a.Pd) + (long long)blockIdx.x * a.Tp * a.ldP;
    if (which == 0) load_image(sY, a.qkv + (long long)b * a.Tp * ld3 + h * AT_HD, ld3, a.Tp, tid);
    else load_image(sY, a.da + (long long)b * a.Tp * a.H + h * AT_HD, a.H, a.Tp, tid);
    for (int i = tid; i < AT_TPAD * 20; i += (int)blockDim.x) {
        const int row = i / 20, ch = i % 20;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (row < a.Tp && ch * 8 < a.ldP) v = *(const uint4*)(X + (long long)row * a.ldP + ch * 8);
        *(uint4*)(sX + ximg_off(row, ch * 8)) = v;
    }
    __syncthreads();
    const int q = i16 >> 2, p = i16 & 3;
    for (int kb = wave; kb < (a.Tp + 15) / 16; kb += (int)(blockDim.x >> 6)) {
        f32x4 o[8];
#pragma unroll
        for (int db = 0; db < 8; ++db) o[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < AT_NB / 2; ++s) {
            const int r0 = 32 * s + 8 * g;   // natural k order: query rows r0..r0+3 and r0+4..r0+7
            bf16x8 xf;
            {
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(sX + ximg_off(r0 + q, 16 * kb + 4 * p)));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(sX + ximg_off(r0 + 4 + q, 16 * kb + 4 * p)));
                union { struct { s16x4 a, b; } p2; bf16x8 v; } u;
                u.p2.a = lo; u.p2.b = hi;
                xf = u.v;
            }
#pragma unroll
            for (int db = 0; db < 8; ++db)
                o[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(sY, r0, r0 + 4, 16 * db, i16), xf, o[db], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        const int key = 16 * kb + i16;
        const long long obase = ((long long)b * a.Tp + key) * ld3 + (which == 0 ? a.H : 2 * a.H) + h * AT_HD;
#pragma unroll
        for (int db = 0; db < 8; ++db) {
            bf16x4 ov = {f2bf(o[db][0]), f2bf(o[db][1]), f2bf(o[db][2]), f2bf(o[db][3])};
            if (key < a.Tp) *(bf16x4*)(a.dqkv + obase + 16 * db + 4 * g) = ov;
            if (a.bias_grad) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = key < a.Tp ? bf2f(ov[r]) : 0.f;
                    t = row16_sum(t);
                    if (i16 == 0) atomicAdd(&sbias[16 * db + 4 * g + r], t);
                }
            }
        }
    }
    if (a.bias_grad) {
        __syncthreads();
        if (tid < AT_HD) atomicAdd(rep_ptr(a.bias_grad, a.rc, blockIdx.x) + (which == 0 ? a.H : 2 * a.H) + h * AT_HD + tid, sbias[tid]);
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Backward in ONE launch for T' <= 144 (nine 16-row blocks): dS and Pd never go through HBM. Phase 1 is the dq kernel above with the
// two bf16x4 global stores per 16-key block (16 rows x 32 B per wave instruction: partial cache lines, the slowest thing in that
// kernel) replaced by LDS stores into two [query][key] images; after a barrier the K / V images are no longer needed, the q and da
// rows of this (batch, head) take their place, and phase 2 is the dk / dv kernel above reading its X operands from those images.
//   LDS: K 36 KB | V 36 KB | X(dS) 42.75 KB | X(Pd) 42.75 KB | bias sums 1.5 KB | key bits = 159.0 KB: one workgroup per CU.
//   X image: 8-row blocks of 288-byte rows, 2432 bytes apart: consecutive blocks start 128 bytes apart modulo the 256-byte bank row,
//   so the two 16-lane groups of a transposed read (rows r .. r+3 of blocks 4s and 4s+1) never share a bank.
// Key validity: one 32-bit word per lane pair of blocks would do; here each lane keeps the 36 bits of ITS keys (block kb, keys
// 16 kb + 4 g + r) in two registers, built once per query block from the workgroup's key bit table - the per-element mask rule of the
// dq kernel (a global load of the token mask + five compares per score element) was a third of its VALU work.
constexpr int AF_NB = 9;                       // 16-row blocks (T' <= 144)
constexpr int AF_ROWS = AF_NB * 16;
constexpr int AF_XBLK = 8 * 288 + 128;         // bytes per 8-row block of an X image
constexpr int AF_XIMG = (AF_ROWS / 8) * AF_XBLK;
__device__ __forceinline__ int xf_off(int row, int colbyte) { return (row >> 3) * AF_XBLK + (row & 7) * 288 + colbyte; }

// global (rows x 128 bf16) -> 256-byte-row image of `img_rows` rows, rows >= nrows zero-filled; all loads of a pass before its stores
__device__ __forceinline__ void load_image_rows(char* img, const bf16_t* src, long long ld, int nrows, int img_rows, int tid) {
    constexpr int U = 4;
    const int nthr = (int)blockDim.x, total = img_rows * 16;
    for (int i0 = tid; i0 < total; i0 += U * nthr) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * nthr, row = i >> 4, ch = i & 15;
            v[u] = make_uint4(0u, 0u, 0u, 0u);
            if (i < total && row < nrows) v[u] = *(const uint4*)(src + (long long)row * ld + ch * 8);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * nthr, row = i >> 4, ch = i & 15;
            if (i < total) *(uint4*)(img + img_off(row, ch)) = v[u];
        }
    }
}

template <bool FULLCTX>   // FULLCTX: context.forward / backward = -2 (configs/ndt1.yaml): validity is the key bit or the diagonal
__global__ __launch_bounds__(576) void attn_bwd_fused_kernel(AttnArgs a) {   // one wave per 16-row block (<= 9 waves)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;                                  // phase 2: the q rows
    char* sV = smem + AF_ROWS * 256;                  // phase 2: the da rows   (K rows 144..159 of a transposed read fall in here: finite)
    char* sXs = smem + 2 * AF_ROWS * 256;             // dS  [query][key]
    char* sXp = sXs + AF_XIMG;                        // Pd  [query][key]
    float* sbias = (float*)(sXp + AF_XIMG);           // [3][128]
    unsigned* sbits = (unsigned*)(sbias + 3 * AT_HD); // [5] key-valid bits (bit k of word k/32)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = (int)(blockDim.x >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / a.nh, h = blockIdx.x % a.nh;
    const long long ld = 3LL * a.H;
    const bf16_t* base = a.qkv + (long long)b * a.Tp * ld + h * AT_HD;
    for (int i = tid; i < 3 * AT_HD; i += (int)blockDim.x) sbias[i] = 0.f;
    load_images_kv(sK, sV, base + a.H, base + 2 * a.H, ld, a.Tp, tid, AF_ROWS);   // (V rows 144..159 land in the X image: overwritten below)
    if (wave == 0) {
        const int32_t* tm = a.tmask + (long long)b * a.Tp;
#pragma unroll
        for (int w = 0; w < 3; ++w) {
            const int k = 64 * w + lane;
            const unsigned long long m = __ballot(k < a.Tp && tm[k < a.Tp ? k : 0] != 0);
            if (lane == 0) { sbits[2 * w] = (unsigned)m; if (w < 2) sbits[2 * w + 1] = (unsigned)(m >> 32); }
        }
    }
    __syncthreads();
    const int nqb = (a.Tp + 15) / 16;
    const float inv_o_scale = 1.0f / a.o_scale;
    const float c2 = a.scale * 1.4426950408889634f;   // exp(s * scale - lse) = exp2(s * c2 - lse * log2 e)
    // ---------------- phase 1: per 16-query block, 32 keys at a time: S, dP -> Pd, dS (LDS) ; dq
    for (int qb = wave; qb < nqb; qb += nwaves) {
        const int query = 16 * qb + i16;
        const int qrow = query < a.Tp ? query : a.Tp - 1;
        const bool qok = query < a.Tp;
        bf16x8 qf[4], df[4];
        float delta = 0.f;
        {
            const long long arow = ((long long)b * a.Tp + qrow) * a.H + h * AT_HD;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                qf[ks] = *(const bf16x8*)(base + (long long)qrow * ld + 32 * ks + 8 * g);
                df[ks] = *(const bf16x8*)(a.da + arow + 32 * ks + 8 * g);
                const bf16x8 of = *(const bf16x8*)(a.ad + arow + 32 * ks + 8 * g);
#pragma unroll
                for (int e = 0; e < 8; ++e) delta += bf2f(df[ks][e]) * bf2f(of[e]);
            }
        }
        delta += __shfl_xor(delta, 16, 64);
        delta += __shfl_xor(delta, 32, 64);
        delta *= inv_o_scale;
        const float nl2 = -a.lse[(long long)blockIdx.x * a.Tp + qrow] * 1.4426950408889634f;
        const unsigned rbase = (unsigned)(((long long)blockIdx.x * a.Tp + qrow) * a.Tp);
        // this lane's key bits: bit 4 kb + r of vlo (kb < 8) / bit r of vhi (kb = 8) <=> key 16 kb + 4 g + r may be attended
        unsigned vlo = 0u, vhi = 0u;
#pragma unroll
        for (int kb = 0; kb < AF_NB; ++kb) {
            const unsigned nib = (sbits[kb >> 1] >> (16 * (kb & 1) + 4 * g)) & 0xFu;
            if (kb < 8) vlo |= nib << (4 * kb); else vhi = nib;
        }
        if ((i16 >> 2) == g) {   // the diagonal is always attendable (ndt1.py:436)
            if (qb < 8) vlo |= 1u << (4 * qb + (i16 & 3)); else vhi |= 1u << (i16 & 3);
        }
        f32x4 o[8];
#pragma unroll
        for (int db = 0; db < 8; ++db) o[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < (AF_NB + 1) / 2; ++s) {
            f32x4 sc[2], dp[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int kb = 2 * s + j;
                sc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                dp[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (kb < AF_NB) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        sc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(sK, 16 * kb + i16, ks, g), qf[ks], sc[j], 0, 0, 0);
                        dp[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(sV, 16 * kb + i16, ks, g), df[ks], dp[j], 0, 0, 0);   // dPd = da . v^T
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int kb = 2 * s + j;
                if (kb >= AF_NB) { dp[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; continue; }
                const unsigned nib = kb < 8 ? (vlo >> (4 * kb)) : vhi;
                const int key0 = 16 * kb + 4 * g;
                float keep[4] = {1.f, 1.f, 1.f, 1.f};
                if (a.p_thr) drop4_any(a.p_key, a.p_thr, rbase + (unsigned)key0, a.p_scale, keep);
                float pdv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    bool ok = ((nib >> r) & 1u) != 0u;
                    if (!FULLCTX) ok = ok && ((key0 + r == qrow) || ctx_ok(qrow, key0 + r, a.cf, a.cb));
                    const float p = (ok && qok) ? __builtin_amdgcn_exp2f(fmaf(sc[j][r], c2, nl2)) : 0.f;
                    pdv[r] = p * keep[r];                                        // Pd
                    dp[j][r] = p * (dp[j][r] * keep[r] - delta) * a.scale;       // dS (scaled), in place
                }
                bf16x4 pv = {f2bf(pdv[0]), f2bf(pdv[1]), f2bf(pdv[2]), f2bf(pdv[3])};
                bf16x4 sv = {f2bf(dp[j][0]), f2bf(dp[j][1]), f2bf(dp[j][2]), f2bf(dp[j][3])};
                *(bf16x4*)(sXp + xf_off(query, 2 * key0)) = pv;
                *(bf16x4*)(sXs + xf_off(query, 2 * key0)) = sv;
            }
            // dq[query][d] += sum over these 32 keys of dS[query][key] k[key][d]   (k image read transposed)
            const bf16x8 sf = pack8(dp[0], dp[1]);
#pragma unroll
            for (int db = 0; db < 8; ++db)
                o[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(sK, 32 * s + 4 * g, 32 * s + 16 + 4 * g, 16 * db, i16), sf,
                                                                o[db], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        const long long obase = ((long long)b * a.Tp + qrow) * ld + h * AT_HD;
#pragma unroll
        for (int db = 0; db < 8; ++db) {
            bf16x4 ov = {f2bf(o[db][0]), f2bf(o[db][1]), f2bf(o[db][2]), f2bf(o[db][3])};
            if (qok) *(bf16x4*)(a.dqkv + obase + 16 * db + 4 * g) = ov;
            if (a.bias_grad) {  // query-bias gradient: column sums of the stored dq
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = qok ? bf2f(ov[r]) : 0.f;
                    t = row16_sum(t);
                    if (i16 == 0) atomicAdd(&sbias[16 * db + 4 * g + r], t);   // LDS atomic
                }
            }
        }
    }
    __syncthreads();   // every wave is done with the K / V images; the X images are complete
    // ---------------- phase 2: dk = dS^T q, dv = Pd^T da  (contraction over queries: both operands read transposed)
    load_image_rows(sK, base, ld, a.Tp, AF_ROWS, tid);
    load_image_rows(sV, a.da + (long long)b * a.Tp * a.H + h * AT_HD, a.H, a.Tp, AF_ROWS, tid);
    __syncthreads();
    const int q4 = i16 >> 2, p4 = i16 & 3;
    const int qrows = 16 * nqb;   // X rows written by phase 1
    for (int kb = wave; kb < nqb; kb += nwaves) {
        f32x4 ok_[8], ov_[8];
#pragma unroll
        for (int db = 0; db < 8; ++db) { ok_[db] = (f32x4){0.f, 0.f, 0.f, 0.f}; ov_[db] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int s = 0; s < (AF_NB + 1) / 2; ++s) {
            if (32 * s >= qrows) break;   // (wave-uniform)
            const int r0 = 32 * s + 8 * g;   // natural k order: query rows r0..r0+3 and r0+4..r0+7
            const bool live = r0 < qrows;    // the last step of an odd block count has 16 query rows only
            const int rr = live ? r0 : 0;
            bf16x8 xs, xp;
            {
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sXs + xf_off(rr + q4, 2 * (16 * kb + 4 * p4))));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sXs + xf_off(rr + 4 + q4, 2 * (16 * kb + 4 * p4))));
                union { struct { s16x4 a, b; } p2; bf16x8 v; } u;
                u.p2.a = lo; u.p2.b = hi; xs = u.v;
                lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sXp + xf_off(rr + q4, 2 * (16 * kb + 4 * p4))));
                hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sXp + xf_off(rr + 4 + q4, 2 * (16 * kb + 4 * p4))));
                u.p2.a = lo; u.p2.b = hi; xp = u.v;
                if (!live) {
                    const bf16x8 z = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
                    xs = z; xp = z;
                }
            }
#pragma unroll
            for (int db = 0; db < 8; ++db) {
                ok_[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(sK, r0, r0 + 4, 16 * db, i16), xs, ok_[db], 0, 0, 0);
                ov_[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(sV, r0, r0 + 4, 16 * db, i16), xp, ov_[db], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const int key = 16 * kb + i16;
        const long long obase = ((long long)b * a.Tp + key) * ld + h * AT_HD;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
#pragma unroll
            for (int db = 0; db < 8; ++db) {
                const f32x4 acc = which == 0 ? ok_[db] : ov_[db];
                bf16x4 ov = {f2bf(acc[0]), f2bf(acc[1]), f2bf(acc[2]), f2bf(acc[3])};
                if (key < a.Tp) *(bf16x4*)(a.dqkv + obase + (which + 1) * a.H + 16 * db + 4 * g) = ov;
                if (a.bias_grad) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float t = key < a.Tp ? bf2f(ov[r]) : 0.f;
                        t = row16_sum(t);
                        if (i16 == 0) atomicAdd(&sbias[(which + 1) * AT_HD + 16 * db + 4 * g + r], t);
                    }
                }
            }
        }
    }
    if (a.bias_grad) {  // one contiguous global atomic pass per workgroup: q, k, v column sums
        __syncthreads();
        for (int i = tid; i < 3 * AT_HD; i += (int)blockDim.x)
            atomicAdd(rep_ptr(a.bias_grad, a.rc, blockIdx.x) + (i / AT_HD) * a.H + h * AT_HD + (i % AT_HD), sbias[i]);
    }
}

// ---- host ------------------------------------------------------------------------------------
bool attn_fused_eligible(int dtype, int Tp, int H, int nh) {
    return dtype == NBCI_BF16 && nh > 0 && H / nh == AT_HD && Tp >= 1 && Tp <= AT_TPAD;
}

static int set_lds(const void* fn, int bytes) { return ensure_dyn_lds(fn, bytes, "attention"); }

static AttnArgs base_args(const void* qkv, const int32_t* tmask, int B, int nh, int Tp, int H, int cf, int cb, float drop_p,
                          uint32_t seed, uint32_t site_p) {
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.qkv = (const bf16_t*)qkv; a.tmask = tmask; a.B = B; a.nh = nh; a.Tp = Tp; a.H = H; a.cf = cf; a.cb = cb;
    a.scale = 1.0f / sqrtf((float)AT_HD);
    a.p_thr = drop_threshold(drop_p); a.p_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f; a.p_key = drop_key(seed, site_p);
    return a;
}

int attn_fwd_launch(const void* qkv, const int32_t* tmask, void* ad, float* lse, int B, int nh, int Tp, int H, int cf, int cb, float drop_p,
                    uint32_t seed, uint32_t site_p, uint32_t site_o, hipStream_t s) {
    NBCI_REQUIRE(attn_fused_eligible(NBCI_BF16, Tp, H, nh), NBCI_ESHAPE, "fused attention: needs head 128 and T' <= 160");
    const int lds = 2 * AT_TPAD * 256 + AT_TPAD * 4, lds9 = (144 + AT_TPAD) * 256 + AT_TPAD * 4;
    {
        int r = set_lds((const void*)attn_fwd_kernel<10, true>, lds); if (r) return r;
        r = set_lds((const void*)attn_fwd_kernel<10, false>, lds); if (r) return r;
        r = set_lds((const void*)attn_fwd_kernel<9, true>, lds9); if (r) return r;
    }
    AttnArgs a = base_args(qkv, tmask, B, nh, Tp, H, cf, cb, drop_p, seed, site_p);
    a.o_thr = a.p_thr; a.o_scale = a.p_scale; a.o_key = drop_key(seed, site_o);
    a.ad = (bf16_t*)ad; a.lse = lse;
    const int nblk = (Tp + 15) / 16;
    static const bool nb9 = measure_env("NBCI_ATTN_NB9", 1) != 0;
    const bool full = cf == -2 && cb == -2;
    if (prof_on()) prof_note_symbol((nb9 && full && Tp <= 144 && nblk >= 2) ? "attn_fwd_kernel<9, true>" : (full ? "attn_fwd_kernel<10, true>" : "attn_fwd_kernel<10, false>"));
    if (nb9 && full && Tp <= 144 && nblk >= 2) {   // (with a bounded context span the 96-register variant spills 53 registers: ten-block kernel)
        hipLaunchKernelGGL((attn_fwd_kernel<9, true>), dim3(B * nh), dim3(64 * nblk), lds9, s, a);
    } else {
        if (full) hipLaunchKernelGGL((attn_fwd_kernel<10, true>), dim3(B * nh), dim3(64 * (nblk < 2 ? 2 : nblk)), lds, s, a);
        else hipLaunchKernelGGL((attn_fwd_kernel<10, false>), dim3(B * nh), dim3(64 * (nblk < 2 ? 2 : nblk)), lds, s, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NBCI_EHIP, std::string("attn_fwd: ") + hipGetErrorString(e));
    return NBCI_OK;
}

int attn_bwd_launch(const void* qkv, const int32_t* tmask, const void* ad, const float* lse, const void* da, void* dS, void* Pd, int ldP,
                    void* dqkv, float* bias_grad, int B, int nh, int Tp, int H, int cf, int cb, float drop_p, uint32_t seed, uint32_t site_p,
                    hipStream_t s, RepCfg rc) {
    NBCI_REQUIRE(attn_fused_eligible(NBCI_BF16, Tp, H, nh), NBCI_ESHAPE, "fused attention: needs head 128 and T' <= 160");
    NBCI_REQUIRE(ldP % 8 == 0 && ldP >= Tp && ldP <= AT_TPAD, NBCI_ESHAPE, "fused attention: bad ldP");
    const int lds1 = 2 * AT_TPAD * 256 + AT_HD * 4, lds2 = AT_TPAD * 256 + AT_TPAD * 320 + AT_HD * 4;
    {
        int r = set_lds((const void*)attn_bwd_dq_kernel, lds1); if (r) return r;
        r = set_lds((const void*)attn_bwd_dkv_kernel, lds2); if (r) return r;
    }
    AttnArgs a = base_args(qkv, tmask, B, nh, Tp, H, cf, cb, drop_p, seed, site_p);
    NBCI_REQUIRE(ad && lse, NBCI_EINVAL, "fused attention backward: needs the forward output and its log-sum-exp");
    a.ad = (bf16_t*)ad; a.lse = (float*)lse; a.o_scale = a.p_scale;
    a.da = (const bf16_t*)da; a.dS = (bf16_t*)dS; a.Pd = (bf16_t*)Pd; a.ldP = ldP; a.dqkv = (bf16_t*)dqkv; a.bias_grad = bias_grad; a.rc = rc;
    const int nblk = (Tp + 15) / 16;
    static const bool one_launch = measure_env("NBCI_ATTN_BWD1", 1) != 0;   // A/B: 0 = the dq + dk/dv pair
    if (one_launch && Tp <= AF_ROWS) {
        constexpr int lds3 = 2 * AF_ROWS * 256 + 2 * AF_XIMG + 3 * AT_HD * 4 + 32;
        {
            int r = set_lds((const void*)attn_bwd_fused_kernel<true>, lds3); if (r) return r;
            r = set_lds((const void*)attn_bwd_fused_kernel<false>, lds3); if (r) return r;
        }
        const dim3 blk(64 * (nblk < 2 ? 2 : nblk));
        if (prof_on()) prof_note_symbol((cf == -2 && cb == -2) ? "attn_bwd_fused_kernel<true>" : "attn_bwd_fused_kernel<false>");
        if (cf == -2 && cb == -2) hipLaunchKernelGGL(attn_bwd_fused_kernel<true>, dim3(B * nh), blk, lds3, s, a);
        else hipLaunchKernelGGL(attn_bwd_fused_kernel<false>, dim3(B * nh), blk, lds3, s, a);
        hipError_t e3 = hipGetLastError();
        if (e3 != hipSuccess) return fail(NBCI_EHIP, std::string("attn_bwd (one launch): ") + hipGetErrorString(e3));
        return NBCI_OK;
    }
    if (prof_on()) prof_note_symbol("attn_bwd_dq_kernel + attn_bwd_dkv_kernel");
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(B * nh), dim3(64 * (nblk < 2 ? 2 : nblk)), lds1, s, a);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(B * nh, 2), dim3(64 * (nblk < 2 ? 2 : nblk)), lds2, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NBCI_EHIP, std::string("attn_bwd: ") + hipGetErrorString(e));
    return NBCI_OK;
}

}  // namespace nbci

#ifdef NBCI_STAMPS
extern "C" int nbci_debug_read_attn_stamps(unsigned long long* host, int nblocks) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(nbci::g_astamps), (size_t)nblocks * 8 * sizeof(unsigned long long));
}
#endif
