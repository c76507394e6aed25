// attn_flash.hip — streaming (online-softmax) unmasked attention on MFMA for long sequences and head sizes 32 / 64 / 96 /
// 128 in bf16: the iTransformer shape (1501 channel tokens, 8 heads x 96; torch.nn.TransformerEncoderLayer under
// models/itransformer.py:158-173) and PatchTST's (205 patches, head 32). No score tensor ever reaches HBM; only the row
// log-sum-exp (and dO.O) are kept for the backward.
//
// One WAVE owns 16 queries (fwd, dq) or 16 keys (dk/dv) and walks the other axis in steps of 32; the four waves of a
// workgroup are independent (no barriers): every LDS image is wave-private and LDS executes a wave's operations in order.
// Operand plumbing per 32-step (16x16x32 MFMA, swapped operands: lane (i16, g) of the result owns row i16, columns 4g..4g+3):
//   * "row" fragments (16 rows x 32 k, lane = row i16, 16-byte chunk g) are loaded straight from global memory;
//   * the operand that must be read TRANSPOSED (V for P.V, K for dS.K, Q / dO for dS^T.Q / Pd^T.dO) is copied into a
//     32-row x 256-byte image (XOR-swizzled chunks) and fetched with ds_read_b64_tr_b16;
//   * the probabilities never leave registers: the two 16-column score tiles of a step ARE the second MFMA operand once
//     the transposed operand's rows are taken in the order pi(g, j) = {4g + j | 16 + 4g + (j - 4)}.
// Dropout bits = the counter stream of the unfused softmax kernel (index ((unit*S + query)*S + key)): both paths and the
// oracle draw identical masks.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "kernels.h"

namespace nbci {

static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NBCI_EHIP, std::string(what) + ": " + hipGetErrorString(e));
    return NBCI_OK;
}

struct FAArgs {
    const bf16_t* qkv;   // (NS*S, 3H)
    bf16_t* out;         // (NS*S, H)
    float* L;            // (NS*nh, S)
    float* Dsum;         // (NS*nh, S)
    const bf16_t* dout;  // (NS*S, H)
    bf16_t* dqkv;        // (NS*S, 3H)
    int NS, nh, S, H;
    float scale;
    unsigned thr; float dscale; uint32_t key;
    // 32 x 32-tile backward with dropout: the dq kernel leaves the keep bits of every (32 queries x 32 keys) tile here for the dk/dv kernel:
    // 32 words per tile, tile (unit, key block kb, query block qb) at ((unit * nblk + kb) * nblk + qb) * 32; word k = key k, bit q = query q kept
    uint32_t* keepbits; int nblk;
    // When the FORWARD of this layer ran with dropout it left the same words in a buffer kept per lse pointer (fa_layer_bits): the backward's dq kernel then
    // takes its lane masks from there (kw_in = 1: no hash, no stores) and the dk/dv kernel reads the same buffer.
    int kw_in;
    // MASK kernels (NDT1: models/ndt1.py:30-41,435-437): key j is visible to query i iff j == i, or the context span allows
    // (i, j) AND token j is valid. tmask (NS, S) int32; cf / cb = context.forward / backward (-2 = unbounded).
    const int32_t* tmask; int cf, cb;
    // dropout of the attention OUTPUT (ndt1.py:292) fused into the forward's store: index = element offset in (NS*S, H)
    unsigned thr_out; float oscale; uint32_t key_out;
};

__device__ __forceinline__ bool fa_ctx(int i, int j, int f, int bk) {   // create_context_mask (ndt1.py:30-41), as kernels.hip ctx_allowed
    if (f >= -1 && j - i > f) return false;
    if (bk >= -1 && i - j > bk) return false;
    return true;
}
__device__ __forceinline__ bool fa_ctx_nb(int i, int j, int f, int bk) {   // the same, without short-circuit control flow
    return ((f < -1) | (j - i <= f)) & ((bk < -1) | (i - j <= bk));
}
// validity bits of keys k0 .. k0+31 of sequence `sq` (bit b = key k0 + b valid and inside the sequence)
__device__ __forceinline__ unsigned fa_valid_bits(const FAArgs& a, int sq, int k0, int lane) {
    bool v = false;
    if (lane < 32 && k0 + lane < a.S) v = a.tmask[(long long)sq * a.S + k0 + lane] != 0;
    return (unsigned)__builtin_amdgcn_ballot_w64(v);
}

constexpr int FA_IMG = 32 * 256;   // bytes of one wave-private image (32 rows x 256 B)

__device__ __forceinline__ int fa_off(int row, int ch) {   // 16-byte chunk `ch` of row `row`
    return row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
}

__device__ __forceinline__ bf16x8 fa_tr(const char* img, int r_lo, int r_hi, int c0, int i16) {
    const int q = i16 >> 2, p = i16 & 3;
    const int col = c0 + 4 * p;
    const int ch = col >> 3, within = (col & 7) * 2;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + fa_off(r_lo + q, ch) + within));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + fa_off(r_hi + q, ch) + within));
    union { struct { s16x4 a, b; } p2; bf16x8 v; } u;
    u.p2.a = lo; u.p2.b = hi;
    return u.v;
}

__device__ __forceinline__ bf16x8 fa_pack(const f32x4& lo, const f32x4& hi) {
    bf16x8 o = {f2bf(lo[0]), f2bf(lo[1]), f2bf(lo[2]), f2bf(lo[3]), f2bf(hi[0]), f2bf(hi[1]), f2bf(hi[2]), f2bf(hi[3])};
    return o;
}

// the keep decisions of the 4 consecutive dropout counters idx0 .. idx0+3 (any parity): the counters pair up two to a 32-bit hash (nbci_common.h
// drop_pair), so 3 hashes cover them instead of 4 element-wise drop_keep calls. The four 16-bit draws are the 64-bit window at half-word offset
// (idx0 & 1) of h0 | h1 | h2: two funnel shifts; a high half is compared in place (hi >= thr <=> word >= thr << 16).
__device__ __forceinline__ void fa_keep4_bits(uint32_t key, uint32_t thr, uint32_t idx0, bool (&k)[4]) {
    const uint32_t p0 = idx0 >> 1;
    const uint32_t h0 = mix32(p0 ^ key), h1 = mix32((p0 + 1u) ^ key), h2 = mix32((p0 + 2u) ^ key);
    const uint32_t sh = (idx0 & 1u) << 4;
    const uint32_t w0 = __builtin_amdgcn_alignbit(h1, h0, sh), w1 = __builtin_amdgcn_alignbit(h2, h1, sh);
    const uint32_t thr_hi = thr << 16;
    k[0] = (w0 & 0xFFFFu) >= thr; k[1] = w0 >= thr_hi; k[2] = (w1 & 0xFFFFu) >= thr; k[3] = w1 >= thr_hi;
}
// ... as multipliers (scale or 0)
__device__ __forceinline__ void fa_keep4(uint32_t key, uint32_t thr, uint32_t idx0, float scale, float (&k)[4]) {
    bool b[4];
    fa_keep4_bits(key, thr, idx0, b);
    k[0] = b[0] ? scale : 0.f; k[1] = b[1] ? scale : 0.f; k[2] = b[2] ? scale : 0.f; k[3] = b[3] ? scale : 0.f;
}

// One step's operands: two row-fragment sets (16 rows x 32 k per (t, ks); lane = row i16, 16-byte chunk g) of two strided
// matrices X and Y, rows r0 .. r0+31 (clamped to S-1: the out-of-range rows only meet zero probabilities). ALL global loads
// of a step are issued here, one step ahead of their use (register double buffer), so a wave sees no load latency.
template <int HD> struct FaBuf { bf16x8 x[2][HD / 32], y[2][HD / 32]; };

template <int HD>
__device__ __forceinline__ void fa_load(FaBuf<HD>& b, const bf16_t* xp, long long ldx, const bf16_t* yp, long long ldy, int r0, int S, int i16) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        int row = r0 + 16 * t + i16;
        if (row > S - 1) row = S - 1;
        const bf16_t* xr = xp + (long long)row * ldx;
        const bf16_t* yr = yp + (long long)row * ldy;
#pragma unroll
        for (int ks = 0; ks < HD / 32; ++ks) { b.x[t][ks] = *(const bf16x8*)(xr + 32 * ks); b.y[t][ks] = *(const bf16x8*)(yr + 32 * ks); }
    }
}

// the fragments a lane holds are exactly the 16-byte chunks (row 16t + i16, chunk 4ks + g) of the 32-row image
template <int HD>
__device__ __forceinline__ void fa_to_image(char* img, const bf16x8 (&f)[2][HD / 32], int i16, int g) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ks = 0; ks < HD / 32; ++ks) *(bf16x8*)(img + fa_off(16 * t + i16, 4 * ks + g)) = f[t][ks];
}

#ifndef FA_NQ
#define FA_NQ 2   // 16-row tiles per wave: every streamed K/V (Q/dO) fragment and transposed read is used FA_NQ times
#endif
constexpr int NQ = FA_NQ;

template <int HD, bool TAIL, bool MASK, int NQ, bool SHARED = false>
__device__ __forceinline__ void fa_fwd_step(const FAArgs& a, char* img, const FaBuf<HD>& b, int k0, const bf16x8 (&qf)[NQ][HD / 32],
                                            f32x4 (&o)[NQ][HD / 16], float (&m)[NQ], float (&l)[NQ], const unsigned (&rbase)[NQ],
                                            const int (&qidx)[NQ], int sq, int lane) {
    constexpr int KS = HD / 32, NDB = HD / 16;
    const int i16 = lane & 15, g = lane >> 4;
    unsigned vb = 0u;
    if constexpr (MASK) vb = fa_valid_bits(a, sq, k0, lane);
    f32x4 sc[NQ][2];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            sc[qi][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) sc[qi][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b.x[t][ks], qf[qi][ks], sc[qi][t], 0, 0, 0);
        }
    if constexpr (!SHARED) fa_to_image<HD>(img, b.y, i16, g);   // V rows of this step, for the transposed read below (SHARED: img IS the workgroup's V stage)
    bf16x8 pf[NQ];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        float cm = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = sc[qi][t][r] * a.scale;
                if (TAIL && k0 + 16 * t + 4 * g + r >= a.S) s = -INFINITY;
                if constexpr (MASK) {
                    const int key = k0 + 16 * t + 4 * g + r;
                    const bool ok = key == qidx[qi] || (fa_ctx(qidx[qi], key, a.cf, a.cb) && ((vb >> (16 * t + 4 * g + r)) & 1u));
                    if (!ok) s = -INFINITY;
                }
                sc[qi][t][r] = s;
                cm = fmaxf(cm, s);
            }
        cm = fmaxf(cm, __shfl_xor(cm, 16, 64));
        cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
        // the accumulator is rescaled only when some row's running maximum actually moved (wave-uniform branch): after
        // the first few steps it rarely does, and the rescale would drag all HD/16 accumulator tiles through the VALU
        if (__builtin_amdgcn_ballot_w64(cm > m[qi]) != 0ull) {
            const float mn = fmaxf(m[qi], cm);
            const float corr = (m[qi] == mn) ? 1.0f : __expf(m[qi] - mn);   // (MASK: a row that has met no visible key yet keeps m = mn = -inf)
            m[qi] = mn;
            l[qi] *= corr;
#pragma unroll
            for (int db = 0; db < NDB; ++db) { o[qi][db][0] *= corr; o[qi][db][1] *= corr; o[qi][db][2] *= corr; o[qi][db][3] *= corr; }
        }
        float ps = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float keep[4] = {1.f, 1.f, 1.f, 1.f};
            if (a.thr) fa_keep4(a.key, a.thr, rbase[qi] + (unsigned)(k0 + 16 * t + 4 * g), a.dscale, keep);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // exp(-inf) = 0 for keys past the end; MASK: a row may not have met a visible key yet (m = -inf): (-inf) - (-inf) is NaN
                const float p = (MASK && sc[qi][t][r] == -INFINITY) ? 0.f : __expf(sc[qi][t][r] - m[qi]);
                ps += p;
                sc[qi][t][r] = p * keep[r];
            }
        }
        l[qi] += ps;
        pf[qi] = fa_pack(sc[qi][0], sc[qi][1]);
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
        const bf16x8 vt = fa_tr(img, 4 * g, 16 + 4 * g, 16 * db, i16);
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) o[qi][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt, pf[qi], o[qi][db], 0, 0, 0);
    }
    asm volatile("" ::: "memory");
}

// software pipeline over the streamed axis: operands of step s+1 are requested before step s is computed
#ifndef FA_PREFETCH
#define FA_PREFETCH 0   // measured: the double buffer costs occupancy (346 VGPRs in dk/dv at head 96) and buys nothing — the kernels are bound by L2 -> register traffic, not latency
#endif
#if FA_PREFETCH
#define FA_PIPELINE(LOAD, STEP)                                                          \
    {                                                                                    \
        const int nsteps = (a.S + 31) / 32;                                              \
        const bool ragged = (a.S & 31) != 0;                                             \
        FaBuf<HD> bufA, bufB;                                                            \
        LOAD(bufA, 0);                                                                   \
        for (int st = 0; st < nsteps; st += 2) {                                         \
            if (st + 1 < nsteps) LOAD(bufB, 32 * (st + 1));                              \
            if (ragged && st == nsteps - 1) STEP(true, bufA, 32 * st); else STEP(false, bufA, 32 * st); \
            if (st + 1 < nsteps) {                                                       \
                if (st + 2 < nsteps) LOAD(bufA, 32 * (st + 2));                          \
                if (ragged && st + 1 == nsteps - 1) STEP(true, bufB, 32 * (st + 1)); else STEP(false, bufB, 32 * (st + 1)); \
            }                                                                            \
        }                                                                                \
    }
#else
#define FA_PIPELINE(LOAD, STEP)                                                          \
    {                                                                                    \
        const int nsteps = (a.S + 31) / 32;                                              \
        const bool ragged = (a.S & 31) != 0;                                             \
        for (int st = 0; st < nsteps; ++st) {                                            \
            FaBuf<HD> bufA;                                                              \
            LOAD(bufA, 32 * st);                                                         \
            if (ragged && st == nsteps - 1) STEP(true, bufA, 32 * st); else STEP(false, bufA, 32 * st); \
        }                                                                                \
    }
#endif

// NQ = 16-query tiles per wave (forward: FA_NQF; the backward kernels keep FA_NQ, they run out of registers beyond two)
template <int HD, bool MASK, int NQ>
__global__ __launch_bounds__(256, (HD <= 96 && NQ == 2) ? 2 : 1) void fattn_fwd_kernel(FAArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = HD / 32, NDB = HD / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int qb0 = (blockIdx.y * 4 + wave) * NQ;
    if (16 * qb0 >= a.S) return;                 // whole wave: no workgroup-level synchronisation anywhere
    char* img = smem + wave * FA_IMG;
    const int unit = blockIdx.x, sq = unit / a.nh, h = unit % a.nh;
    const long long ld = 3LL * a.H;
    const bf16_t* base = a.qkv + (long long)sq * a.S * ld + h * HD;
    bf16x8 qf[NQ][KS];
    f32x4 o[NQ][NDB];
    float m[NQ], l[NQ];
    unsigned rbase[NQ];
    int query[NQ], qidx[NQ];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        query[qi] = 16 * (qb0 + qi) + i16;
        const int qrow = query[qi] < a.S ? query[qi] : a.S - 1;
        qidx[qi] = qrow;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[qi][ks] = *(const bf16x8*)(base + (long long)qrow * ld + 32 * ks + 8 * g);
#pragma unroll
        for (int db = 0; db < NDB; ++db) o[qi][db] = (f32x4){0.f, 0.f, 0.f, 0.f};
        m[qi] = -INFINITY; l[qi] = 0.f;
        rbase[qi] = (unsigned)(((long long)unit * a.S + qrow) * a.S);
    }
    const bf16_t* kp = base + a.H + 8 * g;
    const bf16_t* vp = base + 2 * a.H + 8 * g;
#define FWD_LOAD(B, R0) fa_load<HD>(B, kp, ld, vp, ld, R0, a.S, i16)
#define FWD_STEP(T, B, R0) fa_fwd_step<HD, T, MASK, NQ>(a, img, B, R0, qf, o, m, l, rbase, qidx, sq, lane)
    FA_PIPELINE(FWD_LOAD, FWD_STEP)
#undef FWD_LOAD
#undef FWD_STEP
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        float lt = l[qi];
        lt += __shfl_xor(lt, 16, 64);
        lt += __shfl_xor(lt, 32, 64);
        if (query[qi] < a.S) {
            const float inv = 1.0f / lt;
            const long long obase = ((long long)sq * a.S + query[qi]) * a.H + h * HD;
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                float v[4] = {o[qi][db][0] * inv, o[qi][db][1] * inv, o[qi][db][2] * inv, o[qi][db][3] * inv};
                if (MASK && a.thr_out) drop4(a.key_out, a.thr_out, (unsigned)(obase + 16 * db + 4 * g), a.oscale, v);   // ndt1.py:292
                bf16x4 ov = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
                *(bf16x4*)(a.out + obase + 16 * db + 4 * g) = ov;
            }
            if (g == 0) a.L[(long long)unit * a.S + query[qi]] = m[qi] + __logf(lt);
        }
    }
}

template <int HD, bool TAIL, bool MASK, bool SHARED = false>
__device__ __forceinline__ void fa_bwdq_step(const FAArgs& a, char* img, const FaBuf<HD>& b, int k0, const bf16x8 (&qf)[NQ][HD / 32],
                                             const bf16x8 (&df)[NQ][HD / 32], f32x4 (&dq)[NQ][HD / 16], const float (&Li)[NQ], const float (&D)[NQ],
                                             const unsigned (&rbase)[NQ], const int (&qidx)[NQ], int sq, int lane) {
    constexpr int KS = HD / 32, NDB = HD / 16;
    const int i16 = lane & 15, g = lane >> 4;
    unsigned vb = 0u;
    if constexpr (MASK) vb = fa_valid_bits(a, sq, k0, lane);
    f32x4 sc[NQ][2], dp[NQ][2];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            sc[qi][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dp[qi][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                sc[qi][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b.x[t][ks], qf[qi][ks], sc[qi][t], 0, 0, 0);
                dp[qi][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b.y[t][ks], df[qi][ks], dp[qi][t], 0, 0, 0);   // dPd[query][key] = dO . v
            }
        }
    if constexpr (!SHARED) fa_to_image<HD>(img, b.x, i16, g);   // K rows of this step, for the transposed read
    bf16x8 sf[NQ];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float keep[4] = {1.f, 1.f, 1.f, 1.f};
            if (a.thr) fa_keep4(a.key, a.thr, rbase[qi] + (unsigned)(k0 + 16 * t + 4 * g), a.dscale, keep);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float p = __expf(sc[qi][t][r] * a.scale - Li[qi]);
                if (TAIL && k0 + 16 * t + 4 * g + r >= a.S) p = 0.f;
                if constexpr (MASK) {
                    const int key = k0 + 16 * t + 4 * g + r;
                    if (!(key == qidx[qi] || (fa_ctx(qidx[qi], key, a.cf, a.cb) && ((vb >> (16 * t + 4 * g + r)) & 1u)))) p = 0.f;
                }
                sc[qi][t][r] = p * (dp[qi][t][r] * keep[r] - D[qi]) * a.scale;   // dS, scaled
            }
        }
        sf[qi] = fa_pack(sc[qi][0], sc[qi][1]);
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
        const bf16x8 kt = fa_tr(img, 4 * g, 16 + 4 * g, 16 * db, i16);
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) dq[qi][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, sf[qi], dq[qi][db], 0, 0, 0);
    }
    asm volatile("" ::: "memory");
}

template <int HD, bool MASK>
__global__ __launch_bounds__(256, HD <= 96 ? 2 : 1) void fattn_bwd_q_kernel(FAArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = HD / 32, NDB = HD / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int qb0 = (blockIdx.y * 4 + wave) * NQ;
    if (16 * qb0 >= a.S) return;
    char* img = smem + wave * FA_IMG;
    const int unit = blockIdx.x, sq = unit / a.nh, h = unit % a.nh;
    const long long ld = 3LL * a.H;
    const bf16_t* base = a.qkv + (long long)sq * a.S * ld + h * HD;
    bf16x8 qf[NQ][KS], df[NQ][KS];
    f32x4 dq[NQ][NDB];
    float D[NQ], Li[NQ];
    unsigned rbase[NQ];
    int query[NQ], qidx[NQ];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        query[qi] = 16 * (qb0 + qi) + i16;
        const int qrow = query[qi] < a.S ? query[qi] : a.S - 1;
        qidx[qi] = qrow;
        const bf16_t* dop = a.dout + ((long long)sq * a.S + qrow) * a.H + h * HD;
        const bf16_t* op = a.out + ((long long)sq * a.S + qrow) * a.H + h * HD;
        float d = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[qi][ks] = *(const bf16x8*)(base + (long long)qrow * ld + 32 * ks + 8 * g);
            df[qi][ks] = *(const bf16x8*)(dop + 32 * ks + 8 * g);
            const bf16x8 of = *(const bf16x8*)(op + 32 * ks + 8 * g);
#pragma unroll
            for (int e = 0; e < 8; ++e) d += bf2f(df[qi][ks][e]) * bf2f(of[e]);
        }
        d += __shfl_xor(d, 16, 64);
        d += __shfl_xor(d, 32, 64);
        // with the output dropout fused into the forward both factors carry keep * scale: dO . O_pre = (dO_masked . O_stored) / scale
        if (MASK && a.thr_out) d *= 1.0f / a.oscale;
        D[qi] = d;
        Li[qi] = a.L[(long long)unit * a.S + qrow];
#pragma unroll
        for (int db = 0; db < NDB; ++db) dq[qi][db] = (f32x4){0.f, 0.f, 0.f, 0.f};
        rbase[qi] = (unsigned)(((long long)unit * a.S + qrow) * a.S);
    }
    const bf16_t* kp = base + a.H + 8 * g;
    const bf16_t* vp = base + 2 * a.H + 8 * g;
#define BQ_LOAD(B, R0) fa_load<HD>(B, kp, ld, vp, ld, R0, a.S, i16)
#define BQ_STEP(T, B, R0) fa_bwdq_step<HD, T, MASK>(a, img, B, R0, qf, df, dq, Li, D, rbase, qidx, sq, lane)
    FA_PIPELINE(BQ_LOAD, BQ_STEP)
#undef BQ_LOAD
#undef BQ_STEP
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi)
        if (query[qi] < a.S) {
            const long long obase = ((long long)sq * a.S + query[qi]) * ld + h * HD;
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                bf16x4 ov = {f2bf(dq[qi][db][0]), f2bf(dq[qi][db][1]), f2bf(dq[qi][db][2]), f2bf(dq[qi][db][3])};
                *(bf16x4*)(a.dqkv + obase + 16 * db + 4 * g) = ov;
            }
            if (g == 0) a.Dsum[(long long)unit * a.S + query[qi]] = D[qi];
        }
}

template <int HD, bool TAIL, bool MASK, int NQ, bool SHARED = false>
__device__ __forceinline__ void fa_bwdkv_step(const FAArgs& a, char* imgQ, char* imgD, const FaBuf<HD>& b, int q0, const bf16x8 (&kf)[NQ][HD / 32],
                                              const bf16x8 (&vf)[NQ][HD / 32], f32x4 (&dk)[NQ][HD / 16], f32x4 (&dv)[NQ][HD / 16], const float* Lu,
                                              const float* Du, unsigned ubase, const int (&krow)[NQ], const bool (&key_ok)[NQ], const bool (&key_valid)[NQ], int lane,
                                              const float* ld_stage = nullptr) {
    constexpr int KS = HD / 32, NDB = HD / 16;
    const int i16 = lane & 15, g = lane >> 4;
    f32x4 st[NQ][2], dpt[NQ][2];
#pragma unroll
    for (int ki = 0; ki < NQ; ++ki)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            st[ki][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dpt[ki][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                st[ki][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b.x[t][ks], kf[ki][ks], st[ki][t], 0, 0, 0);    // S^T[key][query]
                dpt[ki][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b.y[t][ks], vf[ki][ks], dpt[ki][t], 0, 0, 0);  // dPd^T[key][query]
            }
        }
    if constexpr (!SHARED) { fa_to_image<HD>(imgQ, b.x, i16, g); fa_to_image<HD>(imgD, b.y, i16, g); }
    float Lq[2][4], Dq[2][4];
    if constexpr (SHARED) {   // staged beside the images: [32 L | 32 D] of this step's queries (clamped rows; those only meet p = 0)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float4 lv = *(const float4*)(ld_stage + 16 * t + 4 * g), dvv = *(const float4*)(ld_stage + 32 + 16 * t + 4 * g);
            Lq[t][0] = lv.x; Lq[t][1] = lv.y; Lq[t][2] = lv.z; Lq[t][3] = lv.w;
            Dq[t][0] = dvv.x; Dq[t][1] = dvv.y; Dq[t][2] = dvv.z; Dq[t][3] = dvv.w;
        }
    } else {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = q0 + 16 * t + 4 * g + r;
            const int qc = (TAIL && q >= a.S) ? a.S - 1 : q;
            Lq[t][r] = Lu[qc]; Dq[t][r] = Du[qc];
        }
    }
    bf16x8 pf[NQ], sf[NQ];
#pragma unroll
    for (int ki = 0; ki < NQ; ++ki) {
        f32x4 pd[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = q0 + 16 * t + 4 * g + r;
                float p = __expf(st[ki][t][r] * a.scale - Lq[t][r]);
                if ((TAIL && q >= a.S) || !key_ok[ki]) p = 0.f;
                if constexpr (MASK) {
                    if (!(q == krow[ki] || (fa_ctx(q, krow[ki], a.cf, a.cb) && key_valid[ki]))) p = 0.f;
                }
                float keep = 1.f;
                if (a.thr) {
                    const int qc = (TAIL && q >= a.S) ? a.S - 1 : q;
                    keep = drop_keep(a.key, a.thr, ubase + (unsigned)qc * (unsigned)a.S + (unsigned)krow[ki]) ? a.dscale : 0.f;
                }
                pd[t][r] = p * keep;
                st[ki][t][r] = p * (dpt[ki][t][r] * keep - Dq[t][r]) * a.scale;   // dS^T, scaled
            }
        pf[ki] = fa_pack(pd[0], pd[1]);
        sf[ki] = fa_pack(st[ki][0], st[ki][1]);
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
        const bf16x8 dt = fa_tr(imgD, 4 * g, 16 + 4 * g, 16 * db, i16);
        const bf16x8 qt = fa_tr(imgQ, 4 * g, 16 + 4 * g, 16 * db, i16);
#pragma unroll
        for (int ki = 0; ki < NQ; ++ki) {
            dv[ki][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dt, pf[ki], dv[ki][db], 0, 0, 0);
            dk[ki][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, sf[ki], dk[ki][db], 0, 0, 0);
        }
    }
    asm volatile("" ::: "memory");
}

// NQ = 16-key tiles per wave: two (every streamed Q / dO fragment used twice), or — NBCI_FA_NKV=1, measurement — one tile at two waves per SIMD
template <int HD, bool MASK, int NQ>
__global__ __launch_bounds__(256, (NQ == 1 && HD <= 96) ? 2 : 1) void fattn_bwd_kv_kernel(FAArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = HD / 32, NDB = HD / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int kb0 = (blockIdx.y * 4 + wave) * NQ;
    if (16 * kb0 >= a.S) return;
    char* imgQ = smem + wave * 2 * FA_IMG;
    char* imgD = imgQ + FA_IMG;
    const int unit = blockIdx.x, sq = unit / a.nh, h = unit % a.nh;
    const long long ld = 3LL * a.H;
    const bf16_t* base = a.qkv + (long long)sq * a.S * ld + h * HD;
    const bf16_t* dob = a.dout + (long long)sq * a.S * a.H + h * HD;
    bf16x8 kf[NQ][KS], vf[NQ][KS];
    f32x4 dk[NQ][NDB], dv[NQ][NDB];
    int key[NQ], krow[NQ];
    bool key_ok[NQ], key_valid[NQ];
#pragma unroll
    for (int ki = 0; ki < NQ; ++ki) {
        key[ki] = 16 * (kb0 + ki) + i16;
        key_ok[ki] = key[ki] < a.S;
        krow[ki] = key_ok[ki] ? key[ki] : a.S - 1;
        key_valid[ki] = MASK ? a.tmask[(long long)sq * a.S + krow[ki]] != 0 : true;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kf[ki][ks] = *(const bf16x8*)(base + a.H + (long long)krow[ki] * ld + 32 * ks + 8 * g);
            vf[ki][ks] = *(const bf16x8*)(base + 2 * a.H + (long long)krow[ki] * ld + 32 * ks + 8 * g);
        }
#pragma unroll
        for (int db = 0; db < NDB; ++db) { dk[ki][db] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[ki][db] = dk[ki][db]; }
    }
    const float* Lu = a.L + (long long)unit * a.S;
    const float* Du = a.Dsum + (long long)unit * a.S;
    const unsigned ubase = (unsigned)((long long)unit * a.S * a.S);
    const bf16_t* qp = base + 8 * g;
    const bf16_t* dp_ = dob + 8 * g;
#define BK_LOAD(B, R0) fa_load<HD>(B, qp, ld, dp_, (long long)a.H, R0, a.S, i16)
#define BK_STEP(T, B, R0) fa_bwdkv_step<HD, T, MASK, NQ>(a, imgQ, imgD, B, R0, kf, vf, dk, dv, Lu, Du, ubase, krow, key_ok, key_valid, lane)
    FA_PIPELINE(BK_LOAD, BK_STEP)
#undef BK_LOAD
#undef BK_STEP
#pragma unroll
    for (int ki = 0; ki < NQ; ++ki)
        if (key_ok[ki]) {
            const long long obase = ((long long)sq * a.S + key[ki]) * ld + h * HD;
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                bf16x4 kv = {f2bf(dk[ki][db][0]), f2bf(dk[ki][db][1]), f2bf(dk[ki][db][2]), f2bf(dk[ki][db][3])};
                bf16x4 vv = {f2bf(dv[ki][db][0]), f2bf(dv[ki][db][1]), f2bf(dv[ki][db][2]), f2bf(dv[ki][db][3])};
                *(bf16x4*)(a.dqkv + obase + a.H + 16 * db + 4 * g) = kv;
                *(bf16x4*)(a.dqkv + obase + 2 * a.H + 16 * db + 4 * g) = vv;
            }
        }
}


// =================================================================================================================================
// Shared-stage variants (round 3). Above, every wave streams the WHOLE other axis of its (sequence, head) from L2 into registers: at
// 1501 tokens x head 96 that is 3.5 GB of L2 -> register traffic per backward launch and the kernels sit on that path (8.7 TB/s), not
// on MFMA or VALU. Here the four waves of a workgroup - four (x NQ) 16-row tiles of the stationary axis of ONE (sequence, head) - share
// each 32-row step of the streamed operands: the workgroup brings the two 32 x HD tiles into LDS ONCE (LDS-DMA, 16 bytes per lane,
// lane-linear destination, the image's XOR swizzle applied to the per-lane SOURCE chunk), one step ahead in a second stage; every wave
// takes its row fragments (ds_read_b128) and its transposed fragments (ds_read_b64_tr_b16) from the same image. One barrier per step.
// L2 traffic per workgroup-step is one tile pair instead of four; the wave-private images and their copies disappear.
// Waves whose tiles lie past the sequence end keep loading and meeting the barriers; they skip the arithmetic and the stores.
constexpr int FA2_STAGE = 2 * FA_IMG + 256;   // X image + Y image + (dk/dv kernel) the step's 32 row log-sum-exps and 32 dO.O sums

typedef __attribute__((address_space(1))) void fa_gvoid;
typedef __attribute__((address_space(3))) void fa_lvoid;

// this wave's share of one step's tiles: pieces w and w + 4 (4 rows x 256 B each) of the X image and of the Y image.
// xp / yp: first column of the head in row 0 of the sequence; rows clamped to S - 1 (they only ever meet zero probabilities)
template <int HD>
__device__ __forceinline__ void fa2_stage(char* stage, const bf16_t* xp, long long ldx, const bf16_t* yp, long long ldy, int r0, int S, int w, int lane) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = w + 4 * j;
        const int row = 4 * p + (lane >> 4), slot = lane & 15;
        int ch = slot ^ (((row & 3) << 2) | ((row >> 2) & 3));      // the chunk whose home is this 16-byte slot (fa_off is an involution per row)
        if (ch >= HD / 8) ch = 0;                                   // (slots no fragment read touches: any valid address)
        int gr = r0 + row;
        if (gr > S - 1) gr = S - 1;
        __builtin_amdgcn_global_load_lds((fa_gvoid*)(xp + (long long)gr * ldx + ch * 8), (fa_lvoid*)(stage + p * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((fa_gvoid*)(yp + (long long)gr * ldy + ch * 8), (fa_lvoid*)(stage + FA_IMG + p * 1024), 16, 0, 0);
    }
}

// row fragments of a step out of the shared images (what fa_load fetched from global memory)
template <int HD>
__device__ __forceinline__ void fa2_frags(FaBuf<HD>& b, const char* stage, int i16, int g) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ks = 0; ks < HD / 32; ++ks) {
            b.x[t][ks] = *(const bf16x8*)(stage + fa_off(16 * t + i16, 4 * ks + g));
            b.y[t][ks] = *(const bf16x8*)(stage + FA_IMG + fa_off(16 * t + i16, 4 * ks + g));
        }
}

#define FA2_PIPELINE(XP, LDX, YP, LDY, STEP, EXTRA)                                                \
    {                                                                                              \
        const int nsteps = (a.S + 31) / 32;                                                        \
        const bool ragged = (a.S & 31) != 0;                                                       \
        fa2_stage<HD>(smem, XP, LDX, YP, LDY, 0, a.S, wave, lane);                                 \
        EXTRA(smem, 0);                                                                            \
        __syncthreads();                                                                           \
        for (int st = 0; st < nsteps; ++st) {                                                      \
            char* cur = smem + (st & 1) * FA2_STAGE;                                               \
            if (st + 1 < nsteps) { fa2_stage<HD>(smem + ((st + 1) & 1) * FA2_STAGE, XP, LDX, YP, LDY, 32 * (st + 1), a.S, wave, lane); EXTRA(smem + ((st + 1) & 1) * FA2_STAGE, 32 * (st + 1)); } \
            if (active) {                                                                          \
                FaBuf<HD> buf;                                                                     \
                fa2_frags<HD>(buf, cur, i16, g);                                                   \
                if (ragged && st == nsteps - 1) STEP(true, buf, cur, 32 * st); else STEP(false, buf, cur, 32 * st); \
            }                                                                                      \
            __syncthreads();   /* next stage landed (hipcc drains the LDS-DMA ahead of the barrier); everyone is done with `cur` */ \
        }                                                                                          \
    }

template <int HD, bool MASK, int NQ>
__global__ __launch_bounds__(256, HD <= 96 ? 2 : 1) void fattn2_fwd_kernel(FAArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = HD / 32, NDB = HD / 16;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    const int qb0 = (blockIdx.y * 4 + wave) * NQ;
    const bool active = 16 * qb0 < a.S;
    const int unit = blockIdx.x, sq = unit / a.nh, h = unit % a.nh;
    const long long ld = 3LL * a.H;
    const bf16_t* base = a.qkv + (long long)sq * a.S * ld + h * HD;
    bf16x8 qf[NQ][KS];
    f32x4 o[NQ][NDB];
    float m[NQ], l[NQ];
    unsigned rbase[NQ];
    int query[NQ], qidx[NQ];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        query[qi] = 16 * (qb0 + qi) + i16;
        const int qrow = query[qi] < a.S ? query[qi] : a.S - 1;
        qidx[qi] = qrow;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[qi][ks] = *(const bf16x8*)(base + (long long)qrow * ld + 32 * ks + 8 * g);
#pragma unroll
        for (int db = 0; db < NDB; ++db) o[qi][db] = (f32x4){0.f, 0.f, 0.f, 0.f};
        m[qi] = -INFINITY; l[qi] = 0.f;
        rbase[qi] = (unsigned)(((long long)unit * a.S + qrow) * a.S);
    }
#define FWD2_STEP(T, B, ST, R0) fa_fwd_step<HD, T, MASK, NQ, true>(a, (ST) + FA_IMG, B, R0, qf, o, m, l, rbase, qidx, sq, lane)
#define FA2_NOEXTRA(ST, R0) do { } while (0)
    FA2_PIPELINE(base + a.H, ld, base + 2 * a.H, ld, FWD2_STEP, FA2_NOEXTRA)
#undef FWD2_STEP
    if (!active) return;
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        float lt = l[qi];
        lt += __shfl_xor(lt, 16, 64);
        lt += __shfl_xor(lt, 32, 64);
        if (query[qi] < a.S) {
            const float inv = 1.0f / lt;
            const long long obase = ((long long)sq * a.S + query[qi]) * a.H + h * HD;
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                float v[4] = {o[qi][db][0] * inv, o[qi][db][1] * inv, o[qi][db][2] * inv, o[qi][db][3] * inv};
                if (MASK && a.thr_out) drop4(a.key_out, a.thr_out, (unsigned)(obase + 16 * db + 4 * g), a.oscale, v);   // ndt1.py:292
                bf16x4 ov = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
                *(bf16x4*)(a.out + obase + 16 * db + 4 * g) = ov;
            }
            if (g == 0) a.L[(long long)unit * a.S + query[qi]] = m[qi] + __logf(lt);
        }
    }
}

template <int HD, bool MASK>
__global__ __launch_bounds__(256, HD <= 96 ? 2 : 1) void fattn2_bwd_q_kernel(FAArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = HD / 32, NDB = HD / 16;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    const int qb0 = (blockIdx.y * 4 + wave) * NQ;
    const bool active = 16 * qb0 < a.S;
    const int unit = blockIdx.x, sq = unit / a.nh, h = unit % a.nh;
    const long long ld = 3LL * a.H;
    const bf16_t* base = a.qkv + (long long)sq * a.S * ld + h * HD;
    bf16x8 qf[NQ][KS], df[NQ][KS];
    f32x4 dq[NQ][NDB];
    float D[NQ], Li[NQ];
    unsigned rbase[NQ];
    int query[NQ], qidx[NQ];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        query[qi] = 16 * (qb0 + qi) + i16;
        const int qrow = query[qi] < a.S ? query[qi] : a.S - 1;
        qidx[qi] = qrow;
        const bf16_t* dop = a.dout + ((long long)sq * a.S + qrow) * a.H + h * HD;
        const bf16_t* op = a.out + ((long long)sq * a.S + qrow) * a.H + h * HD;
        float d = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[qi][ks] = *(const bf16x8*)(base + (long long)qrow * ld + 32 * ks + 8 * g);
            df[qi][ks] = *(const bf16x8*)(dop + 32 * ks + 8 * g);
            const bf16x8 of = *(const bf16x8*)(op + 32 * ks + 8 * g);
#pragma unroll
            for (int e = 0; e < 8; ++e) d += bf2f(df[qi][ks][e]) * bf2f(of[e]);
        }
        d += __shfl_xor(d, 16, 64);
        d += __shfl_xor(d, 32, 64);
        if (MASK && a.thr_out) d *= 1.0f / a.oscale;
        D[qi] = d;
        Li[qi] = a.L[(long long)unit * a.S + qrow];
#pragma unroll
        for (int db = 0; db < NDB; ++db) dq[qi][db] = (f32x4){0.f, 0.f, 0.f, 0.f};
        rbase[qi] = (unsigned)(((long long)unit * a.S + qrow) * a.S);
    }
#define BQ2_STEP(T, B, ST, R0) fa_bwdq_step<HD, T, MASK, true>(a, (ST), B, R0, qf, df, dq, Li, D, rbase, qidx, sq, lane)
    FA2_PIPELINE(base + a.H, ld, base + 2 * a.H, ld, BQ2_STEP, FA2_NOEXTRA)
#undef BQ2_STEP
    if (!active) return;
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi)
        if (query[qi] < a.S) {
            const long long obase = ((long long)sq * a.S + query[qi]) * ld + h * HD;
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                bf16x4 ov = {f2bf(dq[qi][db][0]), f2bf(dq[qi][db][1]), f2bf(dq[qi][db][2]), f2bf(dq[qi][db][3])};
                *(bf16x4*)(a.dqkv + obase + 16 * db + 4 * g) = ov;
            }
            if (g == 0) a.Dsum[(long long)unit * a.S + query[qi]] = D[qi];
        }
}

template <int HD, bool MASK, int NQ>
__global__ __launch_bounds__(256, (NQ == 1 && HD <= 96) ? 2 : 1) void fattn2_bwd_kv_kernel(FAArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = HD / 32, NDB = HD / 16;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    const int kb0 = (blockIdx.y * 4 + wave) * NQ;
    const bool active = 16 * kb0 < a.S;
    const int unit = blockIdx.x, sq = unit / a.nh, h = unit % a.nh;
    const long long ld = 3LL * a.H;
    const bf16_t* base = a.qkv + (long long)sq * a.S * ld + h * HD;
    const bf16_t* dob = a.dout + (long long)sq * a.S * a.H + h * HD;
    bf16x8 kf[NQ][KS], vf[NQ][KS];
    f32x4 dk[NQ][NDB], dv[NQ][NDB];
    int key[NQ], krow[NQ];
    bool key_ok[NQ], key_valid[NQ];
#pragma unroll
    for (int ki = 0; ki < NQ; ++ki) {
        key[ki] = 16 * (kb0 + ki) + i16;
        key_ok[ki] = key[ki] < a.S;
        krow[ki] = key_ok[ki] ? key[ki] : a.S - 1;
        key_valid[ki] = MASK ? a.tmask[(long long)sq * a.S + krow[ki]] != 0 : true;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kf[ki][ks] = *(const bf16x8*)(base + a.H + (long long)krow[ki] * ld + 32 * ks + 8 * g);
            vf[ki][ks] = *(const bf16x8*)(base + 2 * a.H + (long long)krow[ki] * ld + 32 * ks + 8 * g);
        }
#pragma unroll
        for (int db = 0; db < NDB; ++db) { dk[ki][db] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[ki][db] = dk[ki][db]; }
    }
    const float* Lu = a.L + (long long)unit * a.S;
    const float* Du = a.Dsum + (long long)unit * a.S;
    const unsigned ubase = (unsigned)((long long)unit * a.S * a.S);
#define BK2_STEP(T, B, ST, R0) fa_bwdkv_step<HD, T, MASK, NQ, true>(a, (ST), (ST) + FA_IMG, B, R0, kf, vf, dk, dv, Lu, Du, ubase, krow, key_ok, key_valid, lane, (const float*)((ST) + 2 * FA_IMG))
    // wave 0, one 4-byte LDS-DMA instruction per step: lanes 0..31 the row log-sum-exps, lanes 32..63 the dO.O sums of queries R0 .. R0 + 31
#define BK2_EXTRA(ST, R0)                                                                                                  \
    do {                                                                                                                   \
        if (wave == 0) {                                                                                                   \
            int qq = (R0) + (lane & 31);                                                                                   \
            if (qq > a.S - 1) qq = a.S - 1;                                                                                \
            __builtin_amdgcn_global_load_lds((fa_gvoid*)((lane < 32 ? Lu : Du) + qq), (fa_lvoid*)((ST) + 2 * FA_IMG), 4, 0, 0); \
        }                                                                                                                  \
    } while (0)
    FA2_PIPELINE(base, ld, dob, (long long)a.H, BK2_STEP, BK2_EXTRA)
#undef BK2_EXTRA
#undef BK2_STEP
    if (!active) return;
#pragma unroll
    for (int ki = 0; ki < NQ; ++ki)
        if (key_ok[ki]) {
            const long long obase = ((long long)sq * a.S + key[ki]) * ld + h * HD;
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                bf16x4 kv = {f2bf(dk[ki][db][0]), f2bf(dk[ki][db][1]), f2bf(dk[ki][db][2]), f2bf(dk[ki][db][3])};
                bf16x4 vv = {f2bf(dv[ki][db][0]), f2bf(dv[ki][db][1]), f2bf(dv[ki][db][2]), f2bf(dv[ki][db][3])};
                *(bf16x4*)(a.dqkv + obase + a.H + 16 * db + 4 * g) = kv;
                *(bf16x4*)(a.dqkv + obase + 2 * a.H + 16 * db + 4 * g) = vv;
            }
        }
}


// =================================================================================================================================
// Round 4: 32 x 32 score tiles. The kernels above give a wave 16 x 32 score tiles from 16x16x32 MFMAs: four lanes share a query (two
// cross-lane exchanges per 16-query tile and step for the running maximum), every MFMA blocks the SIMD's vector issue for 8 of its 16
// cycles, and a wave carries two query tiles (212+ registers: two waves per SIMD). Here a wave owns 32 queries and computes
// S^T = K . Q^T with v_mfma_f32_32x32x16_bf16 (8 of 32 cycles): lane (q = lane % 32, hi = lane / 32) holds the 16 scores of query q
// against keys {4 hi + (r & 3) + 8 (r >> 2)} of the 32-key step - ONE exchange (v_permlane32_swap) per step for the row maximum, and the
// scores, packed to bf16 in register order, ARE the B operand of O^T += V^T . P^T (two k = 16 slabs) once V's transposed fragment takes
// its keys in the same order. The dropout scale 1 / (1 - p) is applied once, with 1 / l, at the end (the kept probabilities enter P.V
// unscaled).
// Backward on the same tiles. dq kernel: a wave owns 32 queries (lane = query), streams K / V: S^T and dPd^T = V . dO^T come out with the keys
// of a step in a lane's registers, so dS packs straight into the B operand of dQ^T += K^T . dS^T. dk/dv kernel: a wave owns 32 keys (lane = key),
// streams Q / dO (+ the row statistics): S = Q . K^T and dPd = dO . V^T with the step's queries in registers, P and dS pack into the B operands of
// dV^T += dO^T . P and dK^T += Q^T . dS.
// Plumbing shared by the three kernels:
//   * every LDS read is inline asm. Through C++ reads / the transposed-read builtin hipcc puts `s_waitcnt vmcnt(0)` in front of the first read of
//     every step - it cannot tell the read from the LDS-DMA of the NEXT stage still in flight - and the prefetch stops overlapping the step
//     (gemm_glds.h has the same note). The compiler neither counts nor waits for an asm read: fa3_lgkm<N>() before the first use;
//   * a lane's LDS addresses are ONE set of stage-relative byte offsets computed before the loop (Fa3Lane); stage (0 / 1) and image (X / Y /
//     statistics) are the instruction's immediate offset, the loop is unrolled over the two stages. (Left to the compiler, every address of both
//     stages became a loop-invariant register: ~90 VGPRs, spills in the dk/dv kernel at head 96.);
//   * the staging pointers advance by 32 rows per step (Fa3Stager); only the one ragged stage at the end of the sequence is clamped.
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ float fa3_swap32(float x) {   // the value the lane 32 away holds
#if __has_builtin(__builtin_amdgcn_permlane32_swap)
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(threadIdx.x & 32 ? r[0] : r[1]);
#else
    return __shfl_xor(x, 32, 64);
#endif
}

template <int IMM> __device__ __forceinline__ bf16x8 fa3_row(unsigned off) {   // 16 bytes at stage-relative offset off + IMM
    bf16x8 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(off), "n"(IMM));
    return r;
}
template <int IMM> __device__ __forceinline__ float4 fa3_f4(unsigned off) {
    float4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(off), "n"(IMM));
    return r;
}
template <int IMM> __device__ __forceinline__ s16x4 fa3_tr(unsigned off) {
    s16x4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(off), "n"(IMM));
    return r;
}
// at most N of this wave's LDS reads still outstanding (LDS returns in order; any other lgkm traffic only makes the wait more conservative)
template <int N> __device__ __forceinline__ void fa3_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N > 15 ? 15 : N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// end of a step: this wave's share of the next stage has landed, then everyone's has (and everyone is done with the current stage)
__device__ __forceinline__ void fa3_stage_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}
__device__ __forceinline__ bf16x8 fa3_join(const s16x4& a, const s16x4& b) {
    union { struct { s16x4 a, b; } p2; bf16x8 v; } u;
    u.p2.a = a; u.p2.b = b;
    return u.v;
}
__device__ __forceinline__ bf16x8 fa3_pack8(const f32x16& s, int o) {
    const bf16x8 r = {f2bf(s[o]), f2bf(s[o + 1]), f2bf(s[o + 2]), f2bf(s[o + 3]), f2bf(s[o + 4]), f2bf(s[o + 5]), f2bf(s[o + 6]), f2bf(s[o + 7])};
    return r;
}

// Workgroup -> (unit = sequence x head, block of 128 stationary rows). Every workgroup of a unit streams that unit's whole other axis (576 KB of K / V at
// 1501 tokens x head 96), so the workgroups of one unit must run on ONE XCD at about the same time for its 4 MB L2 to serve all but the first read:
// the hardware deals consecutive workgroup ids round-robin to the 8 XCDs, so XCD x = id % 8 takes the units x, x + 8, .. and walks each unit's blocks
// consecutively. (With id = block x units + unit the 512 resident workgroups touched every unit at once - 9 MB per XCD - and the forward ran at the
// 5.2 TB/s the fabric behind the L2 delivers: 885 MB per launch at 1501 x 96 x 128 units.) Returns false for the few ids past the last unit.
__device__ __forceinline__ bool fa3_unit_block(int nunits, int nblocks, int& unit, int& blk) {
    const int id = blockIdx.x, x = id & 7, j = id >> 3;
    unit = x + 8 * (j / nblocks);
    blk = j % nblocks;
    return unit < nunits;
}
__host__ inline int fa3_grid(int nunits, int nblocks) { return 8 * ((nunits + 7) / 8) * nblocks; }

#ifndef FA3_SSTORE
#define FA3_SSTORE 1   // the dq kernel writes the keep words with scalar stores (0: v_writelane + one vector store)
#endif
// sixteen scalar 8-byte stores of a step's compare lane masks kb[0..15] to the tile at `kw` (one s_nop: last compare -> first store)
#define FA3_STORE_MASKS(kb, kw)                                                                                                                     \
    asm volatile("s_nop 4\n\t"                                                                                                                       \
                 "s_store_dwordx2 %0, %16, 0x0\n\ts_store_dwordx2 %1, %16, 0x8\n\ts_store_dwordx2 %2, %16, 0x10\n\ts_store_dwordx2 %3, %16, 0x18\n\t"      \
                 "s_store_dwordx2 %4, %16, 0x20\n\ts_store_dwordx2 %5, %16, 0x28\n\ts_store_dwordx2 %6, %16, 0x30\n\ts_store_dwordx2 %7, %16, 0x38\n\t"    \
                 "s_store_dwordx2 %8, %16, 0x40\n\ts_store_dwordx2 %9, %16, 0x48\n\ts_store_dwordx2 %10, %16, 0x50\n\ts_store_dwordx2 %11, %16, 0x58\n\t"  \
                 "s_store_dwordx2 %12, %16, 0x60\n\ts_store_dwordx2 %13, %16, 0x68\n\ts_store_dwordx2 %14, %16, 0x70\n\ts_store_dwordx2 %15, %16, 0x78"    \
                 ::"s"(kb[0]), "s"(kb[1]), "s"(kb[2]), "s"(kb[3]), "s"(kb[4]), "s"(kb[5]), "s"(kb[6]), "s"(kb[7]), "s"(kb[8]), "s"(kb[9]), "s"(kb[10]),      \
                 "s"(kb[11]), "s"(kb[12]), "s"(kb[13]), "s"(kb[14]), "s"(kb[15]), "s"(kw)                                                        \
                 : "memory")
typedef __attribute__((ext_vector_type(8))) uint64_t u64x8;
// dword slot of key k (0..31) inside a tile's 32 keep words: the dq kernel's compare for register 4 r4 + e yields the words of keys 8 r4 + e and 8 r4 + 4 + e
// as one 64-bit lane mask, stored as one unit
__device__ __forceinline__ int fa3_kslot(int k) { return 2 * (4 * (k >> 3) + (k & 3)) + ((k >> 2) & 1); }

template <int HD> struct Fa3Lane {
    unsigned row0;             // row fragment 0 (16 bytes: columns 8 hi ..) of row lane % 32 of an image; fragment kk (columns 16 kk + 8 hi ..) = row0 ^ (kk << 5):
                               // the chunk index 2 kk + hi meets the swizzle by XOR and stays below 16
    __device__ __forceinline__ unsigned row(int kk) const { return row0 ^ (unsigned)(kk << 5); }
    unsigned tr[4];            // transposed fragments of column block 0: rows {8 j + 4 hi + (lane % 16) / 4}, see Fa3T; block db = tr[j] ^ (db << 6)
                               // (the block index only meets the swizzle's upper two bits: chunk = ((db ^ q) << 2) | ..; smem is 256-byte aligned)
    unsigned stat;             // this lane's first staged statistic (hi selects the second group of four)
    __device__ __forceinline__ void init(const char* smem, int lane) {
        const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)smem;
        const int q32 = lane & 31, hi = lane >> 5, i16 = lane & 15, d16 = (lane >> 4) & 1;
        row0 = base + fa_off(q32, hi);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = 16 * d16 + 4 * (i16 & 3);
            tr[j] = base + fa_off(8 * j + 4 * hi + (i16 >> 2), col >> 3) + (col & 7) * 2;
        }
        stat = base + 16 * hi;
    }
};
// X^T fragments of one 32-column block for the two k = 16 slabs of a step: rows {4 hi .. +3, 8 + 4 hi .. +3} (slab 0) and the same + 16 (slab 1)
struct Fa3T { s16x4 h[4]; };
template <int HD, int IMM> __device__ __forceinline__ void fa3_read_t(Fa3T& v, const Fa3Lane<HD>& ln, int db) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v.h[j] = fa3_tr<IMM>(ln.tr[j] ^ (unsigned)(db << 6));
}

// this wave's share of a stage: pieces w and w + 4 (4 rows x 256 B each) of the X image and of the Y image, 16 bytes per lane by LDS-DMA, the
// image's XOR swizzle applied to the per-lane SOURCE chunk. Addresses = a wave-uniform base that advances 32 rows per step (scalar registers) + a
// constant 32-bit byte offset per lane and piece (the launcher checks S x row pitch < 4 GB).
template <int HD> struct Fa3Stager {
    const char* xb;     // row 0 of the next step to stage (uniform)
    const char* yb;
    const char* x0;     // row 0 of the sequence (uniform; the clamped stage)
    const char* y0;
    unsigned px, py;    // row pitches in bytes
    unsigned xo[2], yo[2];
    bool live[2];       // this lane's 16-byte slot of piece j holds a chunk of the head (heads below 128 fill 4 / 8 / 12 of a row's 16 slots): the other lanes
                        // sit the LDS-DMA out (EXEC) instead of fetching filler - at head 32 that was 3/4 of the L2 -> LDS traffic (3.7 GB per launch at PatchTST's shape)
    __device__ __forceinline__ void init(const bf16_t* xp, long long ldx, const bf16_t* yp, long long ldy, int w, int lane) {
        xb = x0 = (const char*)xp; yb = y0 = (const char*)yp;
        px = (unsigned)(2 * ldx); py = (unsigned)(2 * ldy);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = 4 * (w + 4 * j) + (lane >> 4);
            xo[j] = (unsigned)r * px + chunk_bytes(r, lane);
            yo[j] = (unsigned)r * py + chunk_bytes(r, lane);
            live[j] = chunk_of(r, lane) < HD / 8;
        }
    }
    static __device__ __forceinline__ int chunk_of(int r, int lane) {   // the chunk whose home is this lane's 16-byte slot (fa_off is an involution per row)
        return (lane & 15) ^ (((r & 3) << 2) | ((r >> 2) & 3));
    }
    static __device__ __forceinline__ unsigned chunk_bytes(int r, int lane) {
        const int ch = chunk_of(r, lane);
        return (unsigned)(ch < HD / 8 ? ch : 0) * 16u;
    }
    // CLAMP: rows past the end of the sequence read row S - 1 (they only ever meet zero probabilities); r0 = first row of the step being staged
    template <bool CLAMP> __device__ __forceinline__ void issue(char* stage, int w, int lane, int r0, int S) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const char* xs;
            const char* ys;
            if constexpr (CLAMP) {
                const int r = 4 * (w + 4 * j) + (lane >> 4);
                int gr = r0 + r;
                if (gr > S - 1) gr = S - 1;
                xs = x0 + ((unsigned)gr * px + chunk_bytes(r, lane));
                ys = y0 + ((unsigned)gr * py + chunk_bytes(r, lane));
            } else {
                xs = xb + xo[j];
                ys = yb + yo[j];
            }
            if (HD == 128 || live[j]) {
                __builtin_amdgcn_global_load_lds((fa_gvoid*)xs, (fa_lvoid*)(stage + (w + 4 * j) * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((fa_gvoid*)ys, (fa_lvoid*)(stage + FA_IMG + (w + 4 * j) * 1024), 16, 0, 0);
            }
        }
        xb += 32u * px; yb += 32u * py;
    }
};

// The loop the three kernels share. `st` stages ahead; STEP(stage, tail, k0) computes one step out of stage `stage` (compile-time 0 / 1).
#define FA3_LOOP(ACTIVE, STAGE_EXTRA, STEP)                                                                           \
    {                                                                                                                 \
        const int nsteps = (a.S + 31) / 32, nfull = a.S / 32;                                                         \
        if (nsteps > 1 || nfull == 1) stg.template issue<false>(smem, wave, lane, 0, a.S); else stg.template issue<true>(smem, wave, lane, 0, a.S); \
        STAGE_EXTRA(smem, 0);                                                                                         \
        fa3_stage_barrier();                                                                                          \
        for (int st = 0; st < nfull; st += 2) {                                                                       \
            if (st + 1 < nsteps) {                                                                                    \
                if (st + 1 < nfull) stg.template issue<false>(smem + FA2_STAGE, wave, lane, 32 * (st + 1), a.S);            \
                else stg.template issue<true>(smem + FA2_STAGE, wave, lane, 32 * (st + 1), a.S);                            \
                STAGE_EXTRA(smem + FA2_STAGE, 32 * (st + 1));                                                         \
            }                                                                                                         \
            if (ACTIVE) STEP(0, false, 32 * st);                                                                      \
            fa3_stage_barrier();                                                                                      \
            if (st + 1 >= nfull) break;                                                                               \
            if (st + 2 < nsteps) {                                                                                    \
                if (st + 2 < nfull) stg.template issue<false>(smem, wave, lane, 32 * (st + 2), a.S);                        \
                else stg.template issue<true>(smem, wave, lane, 32 * (st + 2), a.S);                                        \
                STAGE_EXTRA(smem, 32 * (st + 2));                                                                     \
            }                                                                                                         \
            if (ACTIVE) STEP(1, false, 32 * (st + 1));                                                                \
            fa3_stage_barrier();                                                                                      \
        }                                                                                                             \
        if ((ACTIVE) && nfull < nsteps) {                                                                             \
            if (nfull & 1) STEP(1, true, 32 * nfull); else STEP(0, true, 32 * nfull);                                 \
        }                                                                                                             \
    }
#define FA3_NOEXTRA(ST, R0) do { } while (0)

template <int HD, int STG, bool TAIL, bool MASK, bool DROP>
__device__ __forceinline__ void fa3_fwd_step(const FAArgs& a, const Fa3Lane<HD>& ln, int k0, const bf16x8 (&qf)[HD / 16], f32x16 (&o)[HD / 32],
                                             float& m, float& l, float c, unsigned rbase, int qidx, int sq, int lane, uint32_t* kw) {
    constexpr int KK = HD / 16, XI = STG * FA2_STAGE, YI = XI + FA_IMG;   // X = K rows, Y = V rows
    const int hi = lane >> 5;
    bf16x8 kr[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) kr[kk] = fa3_row<XI>(ln.row(kk));
    Fa3T vf;
    fa3_read_t<HD, YI>(vf, ln, 0);   // in flight under the softmax
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    fa3_lgkm<4>();
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[kk], qf[kk], s, 0, 0, 0);
    unsigned vb = 0u;
    if constexpr (MASK) vb = fa_valid_bits(a, sq, k0, lane);
    if constexpr (TAIL || MASK) {
        // branch-free on purpose: with `if (!ok) s[r] = -INFINITY` under the short-circuit form hipcc (ROCm 7.2) hoisted the -inf into the new
        // vector's register, then overwrote it with the whole-vector copy of the divergent element insert - the mask never applied
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kofs = 4 * hi + (r & 3) + 8 * (r >> 2), key = k0 + kofs;
            bool ok = !TAIL || key < a.S;
            if constexpr (MASK) ok = ok & ((key == qidx) | (fa_ctx_nb(qidx, key, a.cf, a.cb) & (((vb >> kofs) & 1u) != 0u)));
            const float sv = s[r];
            s[r] = ok ? sv : -INFINITY;
        }
    }
    // (asm: through fmaxf hipcc first canonicalises every MFMA result - v_max_f32 x, x - before the v_max3_f32; the scores are never signalling NaNs.
    // hipcc's hazard recognizer does not look into asm, so the wait states a vector instruction needs behind the matrix pipe's write - 11 for this
    // 8-pass product - are spelled out: in a plain step nothing else stands between the last S product and the first v_max3)
    asm volatile("s_nop 15\n\ts_nop 1" : "+v"(s));
    float cm;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(cm) : "v"(s[0]), "v"(s[1]), "v"(s[2]));
#pragma unroll
    for (int r = 3; r < 15; r += 2) asm("v_max3_f32 %0, %1, %2, %3" : "=v"(cm) : "v"(cm), "v"(s[r]), "v"(s[r + 1]));
    cm = fmaxf(cm, s[15]);
    cm = fmaxf(cm, fa3_swap32(cm));
    if (__builtin_amdgcn_ballot_w64(cm > m) != 0ull) {   // some row's maximum moved (wave-uniform; rare after the first steps)
        const float mn = fmaxf(m, cm);
        const float corr = (m == mn) ? 1.0f : __builtin_amdgcn_exp2f((m - mn) * c);   // (MASK: a row with no visible key yet keeps -inf)
        m = mn;
        l *= corr;
#pragma unroll
        for (int db = 0; db < HD / 32; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] *= corr;
    }
    const float mc = (MASK && m == -INFINITY) ? 0.f : m * c;   // (all of this row's scores are -inf so far: exp2(-inf - 0) = 0)
    f32x2 ps2 = {0.f, 0.f};   // (two running sums: v_pk_add_f32)
    uint64_t kb[16];
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        bool keep[4] = {true, true, true, true};
        if constexpr (DROP) fa_keep4_bits(a.key, a.thr, rbase + (unsigned)(k0 + 8 * r4 + 4 * hi), keep);
        float p[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) p[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[4 * r4 + e], c, -mc));
        ps2 += (f32x2){p[0], p[1]};
        ps2 += (f32x2){p[2], p[3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s[4 * r4 + e] = (DROP && !keep[e]) ? 0.f : p[e];
            if constexpr (DROP) kb[4 * r4 + e] = __builtin_amdgcn_ballot_w64(keep[e]);   // (the compare's lane mask = the words of keys 8 r4 + e, 8 r4 + 4 + e)
        }
    }
    l += ps2[0] + ps2[1];
    if constexpr (DROP) { if (kw) FA3_STORE_MASKS(kb, kw); }   // for both backward kernels (fa_layer_bits); wave-uniform pointer
    const bf16x8 p0 = fa3_pack8(s, 0), p1 = fa3_pack8(s, 8);
#pragma unroll
    for (int db = 0; db < HD / 32; ++db) {
        fa3_lgkm<0>();
        const bf16x8 v0 = fa3_join(vf.h[0], vf.h[1]), v1 = fa3_join(vf.h[2], vf.h[3]);
        if (db + 1 < HD / 32) fa3_read_t<HD, YI>(vf, ln, db + 1);   // the next block's reads fly under this block's MFMAs
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, p0, o[db], 0, 0, 0);
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, p1, o[db], 0, 0, 0);
    }
}

#ifndef FA3_W96
#define FA3_W96 3   // head 96 forward: three waves per SIMD (168 registers): 171 -> 164 us without, 262 -> 239 us with dropout at 16 x 8 x 1501 x 96
#endif
template <int HD, bool MASK, bool DROP>
__global__ __launch_bounds__(256, HD <= 64 ? 3 : HD <= 96 ? FA3_W96 : 2) void fattn3_fwd_kernel(FAArgs a) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q32 = lane & 31, hi = lane >> 5;
    int unit, blk;
    if (!fa3_unit_block(a.NS * a.nh, (a.S + 127) / 128, unit, blk)) return;
    const int q0 = (blk * 4 + wave) * 32;
    const int sq = unit / a.nh, h = unit % a.nh;
    const long long ld = 3LL * a.H;
    const bf16_t* base = a.qkv + (long long)sq * a.S * ld + h * HD;
    const bool active = q0 < a.S;   // a wave past the end of the sequence: its share of the loads and the barriers, nothing else
    Fa3Stager<HD> stg;
    stg.init(base + a.H, ld, base + 2 * a.H, ld, wave, lane);
    Fa3Lane<HD> ln;
    ln.init(smem, lane);
    const int query = q0 + q32, qrow = query < a.S ? query : a.S - 1;
    bf16x8 qf[HD / 16];
#pragma unroll
    for (int kk = 0; kk < HD / 16; ++kk) qf[kk] = *(const bf16x8*)(base + (long long)qrow * ld + 16 * kk + 8 * hi);
    f32x16 o[HD / 32];
#pragma unroll
    for (int db = 0; db < HD / 32; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m = -INFINITY, l = 0.f;
    const float c = a.scale * 1.44269504088896341f;
    const unsigned rbase = (unsigned)(((long long)unit * a.S + qrow) * a.S);
    uint32_t* kwq = nullptr;   // (uniform) the tile (unit, key block 0, this wave's query block) of the layer's keep words, when the launcher provided a buffer
    if constexpr (DROP) { if (a.keepbits) kwq = a.keepbits + ((long long)unit * a.nblk * a.nblk + (q0 >> 5)) * 32; }
#define F3_STEP(STG, T, K0) fa3_fwd_step<HD, STG, T, MASK, DROP>(a, ln, K0, qf, o, m, l, c, rbase, qrow, sq, lane, \
                                                                 kwq ? kwq + (long long)((K0) >> 5) * a.nblk * 32 : nullptr)
    FA3_LOOP(active, FA3_NOEXTRA, F3_STEP)
#undef F3_STEP
    if constexpr (DROP) asm volatile("s_dcache_wb" ::: "memory");
    if (!active) return;
    const float lt = l + fa3_swap32(l);
    if (query < a.S) {
        const float inv = a.dscale / lt;
        const long long obase = ((long long)sq * a.S + query) * a.H + h * HD;
#pragma unroll
        for (int db = 0; db < HD / 32; ++db)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int d = 32 * db + 8 * r4 + 4 * hi;
                float v[4] = {o[db][4 * r4] * inv, o[db][4 * r4 + 1] * inv, o[db][4 * r4 + 2] * inv, o[db][4 * r4 + 3] * inv};
                if (MASK && a.thr_out) drop4(a.key_out, a.thr_out, (unsigned)(obase + d), a.oscale, v);   // ndt1.py:292
                const bf16x4 ov = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
                *(bf16x4*)(a.out + obase + d) = ov;
            }
        if (hi == 0) a.L[(long long)unit * a.S + query] = m * a.scale + __logf(lt);
    }
}

template <int HD, int STG, bool TAIL, bool MASK, bool DROP, bool KWIN>
__device__ __forceinline__ void fa3_bwdq_step(const FAArgs& a, const Fa3Lane<HD>& ln, int k0, const bf16x8 (&qf)[HD / 16], const bf16x8 (&df)[HD / 16],
                                              f32x16 (&dq)[HD / 32], float L2, float D, float c, unsigned rbase, int qidx, int sq, int lane,
                                              uint32_t* kw, char* smem) {
    constexpr int KK = HD / 16, XI = STG * FA2_STAGE, YI = XI + FA_IMG;   // X = K rows, Y = V rows
    if constexpr (KWIN) {
        // the masks of the step after next: one lane pulls their 128-byte line toward the L2 (a 4-byte LDS-DMA into this stage's unused statistics slot: no
        // register to keep alive; the step's closing vmcnt(0) covers it). The words were written a whole forward + half a backward ago and are read by ONE
        // blocking scalar load per step.
        if (lane == 0 && k0 + 64 < a.S)
            __builtin_amdgcn_global_load_lds((fa_gvoid*)(kw + 2LL * a.nblk * 32), (fa_lvoid*)(smem + XI + 2 * FA_IMG), 4, 0, 0);
    }
    const int hi = lane >> 5;
    bf16x8 kr[KK], vr[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) kr[kk] = fa3_row<XI>(ln.row(kk));
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) vr[kk] = fa3_row<YI>(ln.row(kk));
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    fa3_lgkm<KK>();
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[kk], qf[kk], s, 0, 0, 0);
    fa3_lgkm<0>();
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vr[kk], df[kk], dp, 0, 0, 0);   // dPd[key][query] = v . dO
    Fa3T kt;
    fa3_read_t<HD, XI>(kt, ln, 0);
    u64x8 km0, km1;   // KWIN: the step's sixteen lane masks as the forward stored them (km0[r] for r < 8, km1[r - 8]); valid when the asm returns
    if constexpr (KWIN) {
        asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=&s"(km0), "=&s"(km1) : "s"(kw) : "memory");
    }
    unsigned vb = 0u;
    if constexpr (MASK) vb = fa_valid_bits(a, sq, k0, lane);
    unsigned kword = 0u;   // DROP, FA3_SSTORE == 0: lane k < 32 collects the word of key k0 + k (bit q = query q of this wave keeps it)
    uint64_t kb[16];       // DROP, FA3_SSTORE == 1: the compares' lane masks: kb[4 r4 + e] = words of keys 8 r4 + e (low half) and 8 r4 + 4 + e (high half)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        bool keep[4] = {true, true, true, true};
        if constexpr (DROP && !KWIN) fa_keep4_bits(a.key, a.thr, rbase + (unsigned)(k0 + 8 * r4 + 4 * hi), keep);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 4 * r4 + e, kofs = 8 * r4 + 4 * hi + e, key = k0 + kofs;
            float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c, -L2));
            bool ok = !TAIL || key < a.S;
            if constexpr (MASK) ok = ok & ((key == qidx) | (fa_ctx_nb(qidx, key, a.cf, a.cb) & (((vb >> kofs) & 1u) != 0u)));
            if (TAIL || MASK) p = ok ? p : 0.f;
            float t = dp[r];
            if constexpr (KWIN) {
                // (the multiply first, in C++: hipcc's hazard recognizer does not look into asm, and an asm instruction reading an MFMA result directly
                // gets none of the wait states the matrix pipe needs - seen: v_cndmask right behind the last dPd product, dq wrong by whole factors)
                const uint64_t mk = r < 8 ? km0[r & 7] : km1[r & 7];
                const float td = t * a.dscale;
                asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(t) : "v"(td), "s"(mk));
            } else if constexpr (DROP) {
                t = keep[e] ? t * a.dscale : 0.f;
                // the compare's lane mask IS the pair of words of keys 8 r4 + e (lanes 0..31 = the 32 queries) and 8 r4 + 4 + e (lanes 32..63)
                const uint64_t b = __builtin_amdgcn_ballot_w64(keep[e]);
#if FA3_SSTORE
                kb[r] = b;
#else
                // (s_nop: a VALU-written SGPR is not safe to read in the very next v_writelane - hipcc's hazard recognizer does not look into asm;
                // without it a few keys per tile carried the previous compare's bits)
                asm("s_nop 4\n\tv_writelane_b32 %0, %1, %3\n\tv_writelane_b32 %0, %2, %4"
                    : "+v"(kword) : "s"((unsigned)b), "s"((unsigned)(b >> 32)), "n"(8 * r4 + e), "n"(8 * r4 + 4 + e));
#endif
            }
            s[r] = p * (t - D);   // dS / scale (the scale goes into the output)
        }
    }
    if constexpr (DROP) {
#if FA3_SSTORE
        // the 16 lane masks ARE the tile's 32 words: sixteen scalar 8-byte stores, no vector instruction (slot 2 (4 r4 + e) + half; fa3_kslot is the
        // reader's side). The kernel ends with s_dcache_wb.
        if constexpr (!KWIN) FA3_STORE_MASKS(kb, kw);
#else
        if (lane < 32) kw[fa3_kslot(lane)] = kword;
#endif
    }
    const bf16x8 d0 = fa3_pack8(s, 0), d1 = fa3_pack8(s, 8);
#pragma unroll
    for (int db = 0; db < HD / 32; ++db) {
        fa3_lgkm<0>();
        const bf16x8 k0f = fa3_join(kt.h[0], kt.h[1]), k1f = fa3_join(kt.h[2], kt.h[3]);
        if (db + 1 < HD / 32) fa3_read_t<HD, XI>(kt, ln, db + 1);
        dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0f, d0, dq[db], 0, 0, 0);
        dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1f, d1, dq[db], 0, 0, 0);
    }
}

template <int HD, bool MASK, bool DROP, bool KWIN = false>
// (more waves per SIMD at heads 32 / 64 - four / three here, three in the dk/dv kernel - measured: within +- 3 % at PatchTST's and the head-64 shapes)
__global__ __launch_bounds__(256, HD <= 96 ? 2 : 1) void fattn3_bwd_q_kernel(FAArgs a) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q32 = lane & 31, hi = lane >> 5;
    int unit, blk;
    if (!fa3_unit_block(a.NS * a.nh, (a.S + 127) / 128, unit, blk)) return;
    const int q0 = (blk * 4 + wave) * 32;
    const int sq = unit / a.nh, h = unit % a.nh;
    const long long ld = 3LL * a.H;
    const bf16_t* base = a.qkv + (long long)sq * a.S * ld + h * HD;
    const bool active = q0 < a.S;
    Fa3Stager<HD> stg;
    stg.init(base + a.H, ld, base + 2 * a.H, ld, wave, lane);
    Fa3Lane<HD> ln;
    ln.init(smem, lane);
    const int query = q0 + q32, qrow = query < a.S ? query : a.S - 1;
    const bf16_t* dop = a.dout + ((long long)sq * a.S + qrow) * a.H + h * HD;
    const bf16_t* op = a.out + ((long long)sq * a.S + qrow) * a.H + h * HD;
    bf16x8 qf[HD / 16], df[HD / 16];
    float D = 0.f;
#pragma unroll
    for (int kk = 0; kk < HD / 16; ++kk) {
        qf[kk] = *(const bf16x8*)(base + (long long)qrow * ld + 16 * kk + 8 * hi);
        df[kk] = *(const bf16x8*)(dop + 16 * kk + 8 * hi);
        const bf16x8 of = *(const bf16x8*)(op + 16 * kk + 8 * hi);
#pragma unroll
        for (int e = 0; e < 8; ++e) D += bf2f(df[kk][e]) * bf2f(of[e]);
    }
    D += fa3_swap32(D);
    // with the output dropout fused into the forward both factors carry keep * scale: dO . O_pre = (dO_masked . O_stored) / scale
    if (MASK && a.thr_out) D *= 1.0f / a.oscale;
    const float L2 = a.L[(long long)unit * a.S + qrow] * 1.44269504088896341f;
    f32x16 dq[HD / 32];
#pragma unroll
    for (int db = 0; db < HD / 32; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;
    const float c = a.scale * 1.44269504088896341f;
    const unsigned rbase = (unsigned)(((long long)unit * a.S + qrow) * a.S);
    uint32_t* kwq = nullptr;   // the 32 words of tile (unit, key block 0, this wave's query block)
    if constexpr (DROP) kwq = a.keepbits + ((long long)unit * a.nblk * a.nblk + (q0 >> 5)) * 32;
#define BQ3_STEP(STG, T, K0) fa3_bwdq_step<HD, STG, T, MASK, DROP, KWIN>(a, ln, K0, qf, df, dq, L2, D, c, rbase, qrow, sq, lane, kwq + (long long)((K0) >> 5) * a.nblk * 32, smem)
    FA3_LOOP(active, FA3_NOEXTRA, BQ3_STEP)
#undef BQ3_STEP
#if FA3_SSTORE
    if constexpr (DROP && !KWIN) asm volatile("s_dcache_wb" ::: "memory");   // the scalar stores of the keep words leave the scalar data cache
#endif
    if (!active || query >= a.S) return;
    const long long obase = ((long long)sq * a.S + query) * ld + h * HD;
#pragma unroll
    for (int db = 0; db < HD / 32; ++db)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int d = 32 * db + 8 * r4 + 4 * hi;
            const bf16x4 ov = {f2bf(dq[db][4 * r4] * a.scale), f2bf(dq[db][4 * r4 + 1] * a.scale), f2bf(dq[db][4 * r4 + 2] * a.scale),
                               f2bf(dq[db][4 * r4 + 3] * a.scale)};
            *(bf16x4*)(a.dqkv + obase + d) = ov;
        }
    if (hi == 0) a.Dsum[(long long)unit * a.S + query] = D;
}

// dk/dv kernel, head 96: 144 stationary registers + the step's working set sit a few registers above the 256 of two waves per SIMD, and what hipcc spills
// (V fragments) comes back from scratch behind `s_waitcnt vmcnt(0)`, which drains the stage prefetch. The last NL of a wave's KK stationary V fragments
// therefore live in a wave-private LDS slot (1 KB each) and are read with the step's dO rows.
__host__ __device__ constexpr int fa3_kv_lds_frags(int hd) { return hd == 96 ? 2 : 0; }

template <int HD, int STG, bool TAIL, bool MASK, bool DROP>
__device__ __forceinline__ void fa3_bwdkv_step(const FAArgs& a, const Fa3Lane<HD>& ln, int q0, const bf16x8 (&kf)[HD / 16], const bf16x8 (&vf)[HD / 16],
                                               f32x16 (&dk)[HD / 32], f32x16 (&dv)[HD / 32], float c, int krow, bool key_valid, int lane,
                                               unsigned& kwv, const char* kw_next, unsigned vsl) {
    constexpr int KK = HD / 16, XI = STG * FA2_STAGE, YI = XI + FA_IMG, SI = XI + 2 * FA_IMG, NL = fa3_kv_lds_frags(HD);   // X = Q rows, Y = dO rows, [32 L | 32 D]
    const int hi = lane >> 5;
    // DROP: kwv = the keep word of this lane's key for this step's 32 queries (bit q), fetched one step ahead; a lane's queries are {8 r4 + 4 hi + e}
    unsigned kwn = 0u;
    if constexpr (DROP) { if (kw_next) kwn = *(const uint32_t*)(kw_next + (unsigned)(4 * fa3_kslot(lane & 31))); }   // (kw_next is wave-uniform)
    const unsigned kws = kwv >> (4 * hi);
    // register budget (head 96: 144 stationary + 32 scores at two waves per SIMD): the dO rows take the Q rows' registers once S is issued, the row
    // statistics arrive four queries at a time, one group ahead of their use
    f32x16 st, dpt;
#pragma unroll
    for (int r = 0; r < 16; ++r) { st[r] = 0.f; dpt[r] = 0.f; }
    {
        bf16x8 qr[KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) qr[kk] = fa3_row<XI>(ln.row(kk));
        fa3_lgkm<0>();
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qr[kk], kf[kk], st, 0, 0, 0);     // S[query][key]
        asm volatile("" : "+v"(st));   // (pins the products above the next reads: they are pure, the selection DAG would sink them past the asm)
    }
    {
        bf16x8 dr[KK], vl[NL > 0 ? NL : 1];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) dr[kk] = fa3_row<YI>(ln.row(kk));
#pragma unroll
        for (int j = 0; j < NL; ++j) vl[j] = j == 0 ? fa3_row<0>(vsl) : j == 1 ? fa3_row<1024>(vsl) : fa3_row<2048>(vsl);   // (the V fragments parked in LDS)
        fa3_lgkm<0>();
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
            dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dr[kk], kk < KK - NL ? vf[kk] : vl[kk - (KK - NL)], dpt, 0, 0, 0);   // dPd[query][key] = dO . v
        asm volatile("" : "+v"(dpt));
    }
    Fa3T dt, qt;
    fa3_read_t<HD, YI>(dt, ln, 0);
    fa3_read_t<HD, XI>(qt, ln, 0);
    float4 Ln = fa3_f4<SI>(ln.stat), Dn = fa3_f4<SI + 128>(ln.stat);   // log-sum-exps / dO.O sums of queries {4 hi .. + 3}; next: + 8
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const float4 Lc = Ln, Dc = Dn;
        if (r4 == 0) { Ln = fa3_f4<SI + 32>(ln.stat); Dn = fa3_f4<SI + 160>(ln.stat); }
        if (r4 == 1) { Ln = fa3_f4<SI + 64>(ln.stat); Dn = fa3_f4<SI + 192>(ln.stat); }
        if (r4 == 2) { Ln = fa3_f4<SI + 96>(ln.stat); Dn = fa3_f4<SI + 224>(ln.stat); }
        if (r4 < 3) fa3_lgkm<2>(); else fa3_lgkm<0>();
        const float Lv[4] = {Lc.x, Lc.y, Lc.z, Lc.w}, Dv[4] = {Dc.x, Dc.y, Dc.z, Dc.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 4 * r4 + e, q = q0 + 8 * r4 + 4 * hi + e;
            float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], c, -1.44269504088896341f * Lv[e]));
            bool ok = !TAIL || q < a.S;
            if constexpr (MASK) ok = ok & ((q == krow) | (fa_ctx_nb(q, krow, a.cf, a.cb) & key_valid));
            if (TAIL || MASK) p = ok ? p : 0.f;
            float t = dpt[r], pk = p;
            if constexpr (DROP) {   // all-ones / zero from the query's bit; 1 / (1 - p) reaches dV at the end
                const unsigned km = (unsigned)__builtin_amdgcn_sbfe((int)kws, 8 * r4 + e, 1);
                t = __uint_as_float(__float_as_uint(t) & km) * a.dscale;
                pk = __uint_as_float(__float_as_uint(pk) & km);
            }
            dpt[r] = pk;                          // the dropped probability takes dPd's register
            st[r] = p * (t - Dv[e]) * a.scale;   // dS
        }
    }
    const bf16x8 p0 = fa3_pack8(dpt, 0), p1 = fa3_pack8(dpt, 8), s0 = fa3_pack8(st, 0), s1 = fa3_pack8(st, 8);
#pragma unroll
    for (int db = 0; db < HD / 32; ++db) {
        fa3_lgkm<0>();
        const bf16x8 d0f = fa3_join(dt.h[0], dt.h[1]), d1f = fa3_join(dt.h[2], dt.h[3]);
        const bf16x8 q0f = fa3_join(qt.h[0], qt.h[1]), q1f = fa3_join(qt.h[2], qt.h[3]);
        if (db + 1 < HD / 32) { fa3_read_t<HD, YI>(dt, ln, db + 1); fa3_read_t<HD, XI>(qt, ln, db + 1); }
        dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0f, p0, dv[db], 0, 0, 0);
        dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1f, p1, dv[db], 0, 0, 0);
        dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q0f, s0, dk[db], 0, 0, 0);
        dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q1f, s1, dk[db], 0, 0, 0);
    }
    if constexpr (DROP) kwv = kwn;
}

template <int HD, bool MASK, bool DROP>
__global__ __launch_bounds__(256, HD <= 96 ? 2 : 1) void fattn3_bwd_kv_kernel(FAArgs a) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q32 = lane & 31, hi = lane >> 5;
    int unit, blk;
    if (!fa3_unit_block(a.NS * a.nh, (a.S + 127) / 128, unit, blk)) return;
    const int k0 = (blk * 4 + wave) * 32;
    const int sq = unit / a.nh, h = unit % a.nh;
    const long long ld = 3LL * a.H;
    const bf16_t* base = a.qkv + (long long)sq * a.S * ld + h * HD;
    const bf16_t* dob = a.dout + (long long)sq * a.S * a.H + h * HD;
    const float* Lu = a.L + (long long)unit * a.S;
    const float* Du = a.Dsum + (long long)unit * a.S;
    const bool active = k0 < a.S;
    Fa3Stager<HD> stg;
    stg.init(base, ld, dob, (long long)a.H, wave, lane);
    Fa3Lane<HD> ln;
    ln.init(smem, lane);
    const int key = k0 + q32, krow = key < a.S ? key : a.S - 1;
    const bool key_valid = MASK ? a.tmask[(long long)sq * a.S + krow] != 0 : true;
    bf16x8 kf[HD / 16], vf[HD / 16];
#pragma unroll
    for (int kk = 0; kk < HD / 16; ++kk) {
        kf[kk] = *(const bf16x8*)(base + a.H + (long long)krow * ld + 16 * kk + 8 * hi);
        vf[kk] = *(const bf16x8*)(base + 2 * a.H + (long long)krow * ld + 16 * kk + 8 * hi);
    }
    constexpr int NL = fa3_kv_lds_frags(HD), KKR = HD / 16 - NL;
    char* vslot = smem + 2 * FA2_STAGE + wave * (NL * 1024) + lane * 16;
    const unsigned vsl = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)vslot;
#pragma unroll
    for (int j = 0; j < NL; ++j) *(bf16x8*)(vslot + j * 1024) = vf[KKR + j];   // (wave-private: no barrier; LDS executes a wave's operations in order)
    f32x16 dk[HD / 32], dv[HD / 32];
#pragma unroll
    for (int db = 0; db < HD / 32; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }
    const float c = a.scale * 1.44269504088896341f;
    // one 4-byte LDS-DMA instruction per step of wave 0 (the row log-sum-exps of queries R0 .. R0 + 31) and of wave 1 (their dO.O sums), lanes 0..31
    const float* statp = wave == 0 ? Lu : Du;   // (uniform)
#define BK3_EXTRA(ST, R0)                                                                                                  \
    do {                                                                                                                   \
        if (wave < 2 && lane < 32) {                                                                                       \
            int qq = (R0) + lane;                                                                                          \
            if (qq > a.S - 1) qq = a.S - 1;                                                                                \
            __builtin_amdgcn_global_load_lds((fa_gvoid*)((const char*)statp + (unsigned)(4 * qq)), (fa_lvoid*)((ST) + 2 * FA_IMG + 128 * wave), 4, 0, 0); \
        }                                                                                                                  \
    } while (0)
    const char* kwk = nullptr;   // (uniform) the tiles of this wave's key block, query block 0; a lane's key's word at + 4 (lane % 32)
    unsigned kwv = 0u;
    if constexpr (DROP) {
        kwk = (const char*)(a.keepbits + ((long long)unit * a.nblk + (k0 >> 5)) * a.nblk * 32);
        if (active) kwv = *(const uint32_t*)(kwk + (unsigned)(4 * fa3_kslot(q32)));
    }
#define BK3_STEP(STG, T, Q0) fa3_bwdkv_step<HD, STG, T, MASK, DROP>(a, ln, Q0, kf, vf, dk, dv, c, krow, key_valid, lane, kwv, \
                                                                    (DROP && (Q0) + 32 < a.S) ? kwk + (((Q0) >> 5) + 1) * 128 : nullptr, vsl)
    FA3_LOOP(active, BK3_EXTRA, BK3_STEP)
#undef BK3_STEP
#undef BK3_EXTRA
    if (!active || key >= a.S) return;
    const long long obase = ((long long)sq * a.S + key) * ld + h * HD;
#pragma unroll
    for (int db = 0; db < HD / 32; ++db)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int d = 32 * db + 8 * r4 + 4 * hi;
            const bf16x4 kv = {f2bf(dk[db][4 * r4]), f2bf(dk[db][4 * r4 + 1]), f2bf(dk[db][4 * r4 + 2]), f2bf(dk[db][4 * r4 + 3])};
            const float ds = DROP ? a.dscale : 1.0f;
            const bf16x4 vv = {f2bf(dv[db][4 * r4] * ds), f2bf(dv[db][4 * r4 + 1] * ds), f2bf(dv[db][4 * r4 + 2] * ds), f2bf(dv[db][4 * r4 + 3] * ds)};
            *(bf16x4*)(a.dqkv + obase + a.H + d) = kv;
            *(bf16x4*)(a.dqkv + obase + 2 * a.H + d) = vv;
        }
}

bool fattn_eligible(int dtype, int S, int H, int nh) {
    const char* e = measure_env_str("NBCI_FLASH_ATTN");   // (measurement builds; the NDT1 plan has its own switch, read at plan creation)
    const bool off = e && e[0] == '0';
    if (off || dtype != NBCI_BF16 || nh <= 0 || H % nh) return false;
    const int hd = H / nh;
    return (hd == 32 || hd == 64 || hd == 96 || hd == 128) && S >= 1 && H % 8 == 0;
}

static int fa_args(FAArgs& a, int NS, int nh, int S, int H, float drop_p, uint32_t seed, uint32_t site) {
    NBCI_REQUIRE((long long)NS * nh * S * (long long)S < (1ll << 32), NBCI_ESHAPE, "flash attention: too large for the 32-bit dropout counter");
    NBCI_REQUIRE((long long)S * 6 * H < (1ll << 32), NBCI_ESHAPE, "flash attention: one sequence of qkv rows must stay below 4 GB (32-bit staging offsets)");
    a.NS = NS; a.nh = nh; a.S = S; a.H = H;
    a.scale = 1.0f / sqrtf((float)(H / nh));
    a.thr = drop_threshold(drop_p);
    a.dscale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    a.key = drop_key(seed, site);
    return NBCI_OK;
}

// keep-bit scratch of the 32 x 32-tile backward, one buffer per (device, stream), grown on demand, freed by nbci_release_scratch
struct FaScratch { uint32_t* p; size_t bytes; };
static std::mutex g_fa_mu;
static std::map<std::pair<int, hipStream_t>, FaScratch> g_fa_scratch;
static int fa_keepbits(hipStream_t stream, size_t bytes, uint32_t** out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(NBCI_EHIP, "flash attention scratch: hipGetDevice");
    std::lock_guard<std::mutex> lk(g_fa_mu);
    const auto key = std::make_pair(dev, stream);
    auto it = g_fa_scratch.find(key);
    if (it != g_fa_scratch.end() && it->second.bytes >= bytes) { *out = it->second.p; return NBCI_OK; }
    if (it != g_fa_scratch.end()) {   // an earlier launch on this stream may still be using the old buffer
        if (hipStreamSynchronize(stream) != hipSuccess) return fail(NBCI_EHIP, "flash attention scratch: sync");
        (void)hipFree(it->second.p);
        g_fa_scratch.erase(it);
    }
    FaScratch sc{nullptr, bytes};
    if (hipMalloc((void**)&sc.p, bytes) != hipSuccess) return fail(NBCI_EHIP, "flash attention scratch: hipMalloc");
    g_fa_scratch[key] = sc;
    *out = sc.p;
    return NBCI_OK;
}
// keep words of a LAYER, written by its forward and read by both backward kernels: one buffer per (device, lse pointer) - the caller's log-sum-exp buffer
// identifies the layer -, tagged with what the forward drew them for; a backward whose tag does not match draws them itself (fa_keepbits)
struct FaLayerBits { uint32_t* p; size_t bytes; uint32_t key; unsigned thr; int NS, nh, S; };
static std::map<std::pair<int, const void*>, FaLayerBits> g_fa_layer;
static int fa_layer_bits_for_forward(hipStream_t stream, const FAArgs& a, size_t bytes, uint32_t** out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(NBCI_EHIP, "flash attention layer scratch: hipGetDevice");
    std::lock_guard<std::mutex> lk(g_fa_mu);
    // (a process that builds and drops many models leaves entries behind under lse pointers that are gone: past 4 GB everything is dropped once - the
    // layers in use re-create theirs on their next forward)
    size_t total = 0;
    for (auto& kv : g_fa_layer) total += kv.second.bytes;
    if (total + bytes > (4ull << 30) && g_fa_layer.find(std::make_pair(dev, (const void*)a.L)) == g_fa_layer.end()) {
        if (hipDeviceSynchronize() != hipSuccess) return fail(NBCI_EHIP, "flash attention layer scratch: sync");
        for (auto& kv : g_fa_layer) (void)hipFree(kv.second.p);
        g_fa_layer.clear();
    }
    FaLayerBits& e = g_fa_layer[std::make_pair(dev, (const void*)a.L)];
    if (e.bytes < bytes) {
        if (e.p) {   // an earlier launch may still be using the old buffer
            if (hipStreamSynchronize(stream) != hipSuccess) return fail(NBCI_EHIP, "flash attention layer scratch: sync");
            (void)hipFree(e.p);
        }
        e.p = nullptr; e.bytes = 0;
        if (hipMalloc((void**)&e.p, bytes) != hipSuccess) { g_fa_layer.erase(std::make_pair(dev, (const void*)a.L)); return fail(NBCI_EHIP, "flash attention layer scratch: hipMalloc"); }
        e.bytes = bytes;
    }
    e.key = a.key; e.thr = a.thr; e.NS = a.NS; e.nh = a.nh; e.S = a.S;
    *out = e.p;
    return NBCI_OK;
}
static uint32_t* fa_layer_bits_for_backward(const FAArgs& a) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_fa_mu);
    auto it = g_fa_layer.find(std::make_pair(dev, (const void*)a.L));
    if (it == g_fa_layer.end()) return nullptr;
    const FaLayerBits& e = it->second;
    return (e.key == a.key && e.thr == a.thr && e.NS == a.NS && e.nh == a.nh && e.S == a.S) ? e.p : nullptr;
}
int fattn_release() {
    std::lock_guard<std::mutex> lk(g_fa_mu);
    for (auto& kv : g_fa_scratch) (void)hipFree(kv.second.p);
    g_fa_scratch.clear();
    for (auto& kv : g_fa_layer) (void)hipFree(kv.second.p);
    g_fa_layer.clear();
    return NBCI_OK;
}

static int g_fa_force16 = 0;   // (measurement: fa_dispatch's comparison run)

template <int HD, bool MASK>
static int fa_launch(int which, const FAArgs& a0, hipStream_t s) {
    FAArgs a = a0;
    dim3 g(a.NS * a.nh, (a.S + 64 * NQ - 1) / (64 * NQ));
    static const int shared = measure_env("NBCI_FA_SHARED", 1);   // measurement: 0 = the wave-private streaming kernels above
    static const int tiles32 = measure_env("NBCI_FA_TILES32", 1);  // measurement: 0 = the 16 x 32 score-tile kernels
    static const int tiles32b = measure_env("NBCI_FA_TILES32_BWD", 1);   // measurement: 0 = the 16 x 32 backward kernels
    // (head 128: the backward kernels take 316 - 416 registers, one wave per SIMD, and still beat the 16 x 32 pair: 1119 -> 730 us at 64 x 8 x 593 x 128)
    if (tiles32 && !g_fa_force16 && (which == 0 || tiles32b)) {
        const dim3 g3(fa3_grid(a.NS * a.nh, (a.S + 127) / 128));
        static const int fwdbits = measure_env("NBCI_FA_FWD_BITS", 1);   // measurement: 0 = the dq kernel draws the keep bits itself
        if (which == 0) {
            if (prof_on()) prof_note_symbol("fattn3_fwd_kernel");
            if (a.thr && fwdbits && a.L) {
                a.nblk = (a.S + 31) / 32;
                const int rc = fa_layer_bits_for_forward(s, a, (size_t)a.NS * a.nh * a.nblk * a.nblk * 128, &a.keepbits);
                if (rc != NBCI_OK) return rc;
            }
            if (a.thr) hipLaunchKernelGGL((fattn3_fwd_kernel<HD, MASK, true>), g3, dim3(256), 2 * FA2_STAGE, s, a);
            else hipLaunchKernelGGL((fattn3_fwd_kernel<HD, MASK, false>), g3, dim3(256), 2 * FA2_STAGE, s, a);
        } else if (which == 1) {
            if (a.thr) {
                a.nblk = (a.S + 31) / 32;
                a.keepbits = fwdbits ? fa_layer_bits_for_backward(a) : nullptr;
                a.kw_in = a.keepbits != nullptr;
                if (!a.kw_in) {
                    const int rc = fa_keepbits(s, (size_t)a.NS * a.nh * a.nblk * a.nblk * 128, &a.keepbits);
                    if (rc != NBCI_OK) return rc;
                }
            }
            if (prof_on()) prof_note_symbol("fattn3_bwd_q_kernel");
            if (a.thr && a.kw_in) hipLaunchKernelGGL((fattn3_bwd_q_kernel<HD, MASK, true, true>), g3, dim3(256), 2 * FA2_STAGE, s, a);
            else if (a.thr) hipLaunchKernelGGL((fattn3_bwd_q_kernel<HD, MASK, true>), g3, dim3(256), 2 * FA2_STAGE, s, a);
            else hipLaunchKernelGGL((fattn3_bwd_q_kernel<HD, MASK, false>), g3, dim3(256), 2 * FA2_STAGE, s, a);
        } else {
            if (a.thr) {   // (the layer's buffer the forward filled, else the one the dq launch on this stream just filled)
                a.nblk = (a.S + 31) / 32;
                a.keepbits = fwdbits ? fa_layer_bits_for_backward(a) : nullptr;
                if (!a.keepbits) {
                    const int rc = fa_keepbits(s, (size_t)a.NS * a.nh * a.nblk * a.nblk * 128, &a.keepbits);
                    if (rc != NBCI_OK) return rc;
                }
            }
            if (prof_on()) prof_note_symbol("fattn3_bwd_kv_kernel");
            constexpr int lds = 2 * FA2_STAGE + 4 * 1024 * fa3_kv_lds_frags(HD);
            if (a.thr) hipLaunchKernelGGL((fattn3_bwd_kv_kernel<HD, MASK, true>), g3, dim3(256), lds, s, a);
            else hipLaunchKernelGGL((fattn3_bwd_kv_kernel<HD, MASK, false>), g3, dim3(256), lds, s, a);
        }
        return check_launch("flash attention (32 x 32 tiles)");
    }
    if (shared && (HD < 128 || which == 0)) {   // the four waves of a workgroup share every streamed tile through LDS (two stages of two 8 KB images);
                                                  // head 128 backward: the shared dk/dv kernel spills (35 registers) and runs 27 % slower: wave-private kernels
        if (prof_on()) prof_note_symbol(which == 0 ? "fattn2_fwd_kernel" : which == 1 ? "fattn2_bwd_q_kernel" : "fattn2_bwd_kv_kernel");
        // dk/dv: ONE 16-key tile per wave at two waves per SIMD (with the streamed tiles shared, halving a wave's keys no longer doubles the L2
        // traffic that mattered): 1063 -> 853 us at 16 x 8 x 1501 x 96, 1877 -> 1669 us at 2048 x 8 x 205 x 32 (NBCI_FA_NKV=2: two tiles, one wave per SIMD)
        static const int nkv1 = measure_env("NBCI_FA_NKV", 1);
        if (which == 0) hipLaunchKernelGGL((fattn2_fwd_kernel<HD, MASK, 2>), g, dim3(256), 2 * FA2_STAGE, s, a);
        else if (which == 1) hipLaunchKernelGGL((fattn2_bwd_q_kernel<HD, MASK>), g, dim3(256), 2 * FA2_STAGE, s, a);
        else if (nkv1 == 1) hipLaunchKernelGGL((fattn2_bwd_kv_kernel<HD, MASK, 1>), dim3(a.NS * a.nh, (a.S + 63) / 64), dim3(256), 2 * FA2_STAGE, s, a);
        else hipLaunchKernelGGL((fattn2_bwd_kv_kernel<HD, MASK, 2>), g, dim3(256), 2 * FA2_STAGE, s, a);
        return check_launch("flash attention (shared stages)");
    }
    if (which == 0) {
        // forward: two 16-query tiles per wave. Up to head 96 the kernel (and the dq kernel) is told to fit two waves per SIMD
        // (__launch_bounds__(256, 2)): left alone the compiler spread 180 + 92 registers over VGPRs and AGPRs for ONE wave per SIMD;
        // 212 VGPRs, no spill, twice the occupancy: 551 -> 427 us at 16 x 8 heads x 1501 x 96 (dq + dk/dv 1172 -> 1105 us; the dk/dv
        // kernel spills under the same bound and runs 2 x slower: left alone). Head 128 cannot fit two waves; there three tiles per
        // wave (every streamed K / V fragment used three times) are worth 5 % on long sequences (139 -> 132 us at 593 tokens), while
        // at head 96 they lose to the occupancy (495 us) and at head 32 to the softmax's VALU work (758 vs 998 us).
        static const int env_nqf = measure_env("NBCI_FA_NQF", 0);   // measurement: force 2 or 3
        if (env_nqf ? env_nqf == 3 : (HD > 96 && a.S >= 400)) {
            dim3 gf(a.NS * a.nh, (a.S + 64 * 3 - 1) / (64 * 3));
            hipLaunchKernelGGL((fattn_fwd_kernel<HD, MASK, 3>), gf, dim3(256), 4 * FA_IMG, s, a);
        } else {
            hipLaunchKernelGGL((fattn_fwd_kernel<HD, MASK, 2>), g, dim3(256), 4 * FA_IMG, s, a);
        }
    } else if (which == 1) hipLaunchKernelGGL((fattn_bwd_q_kernel<HD, MASK>), g, dim3(256), 4 * FA_IMG, s, a);
    else {
        static const int env_nkv = measure_env("NBCI_FA_NKV", 0);   // measurement: 1 = one key tile per wave
        if (env_nkv == 1) {
            dim3 g1(a.NS * a.nh, (a.S + 64 - 1) / 64);
            hipLaunchKernelGGL((fattn_bwd_kv_kernel<HD, MASK, 1>), g1, dim3(256), 8 * FA_IMG, s, a);
        } else {
            hipLaunchKernelGGL((fattn_bwd_kv_kernel<HD, MASK, 2>), g, dim3(256), 8 * FA_IMG, s, a);
        }
    }
    return check_launch("flash attention");
}

static int fa_dispatch_raw(int which, const FAArgs& a, hipStream_t s) {
    if (a.tmask) {
        switch (a.H / a.nh) {
            case 32: return fa_launch<32, true>(which, a, s);
            case 64: return fa_launch<64, true>(which, a, s);
            case 96: return fa_launch<96, true>(which, a, s);
            default: return fa_launch<128, true>(which, a, s);
        }
    }
    switch (a.H / a.nh) {
        case 32: return fa_launch<32, false>(which, a, s);
        case 64: return fa_launch<64, false>(which, a, s);
        case 96: return fa_launch<96, false>(which, a, s);
        default: return fa_launch<128, false>(which, a, s);
    }
}

#ifdef NBCI_MEASURE
static void fa_checksum(const char* what, const FAArgs& a, hipStream_t s) {
    const size_t no = (size_t)a.NS * a.S * a.H, nl = (size_t)a.NS * a.nh * a.S, nq = 3 * no;
    (void)hipStreamSynchronize(s);
    std::vector<uint16_t> h(no), q(nq);
    std::vector<uint32_t> L(nl);
    (void)hipMemcpy(h.data(), a.out, no * 2, hipMemcpyDeviceToHost);
    (void)hipMemcpy(q.data(), a.qkv, nq * 2, hipMemcpyDeviceToHost);
    (void)hipMemcpy(L.data(), a.L, nl * 4, hipMemcpyDeviceToHost);
    uint64_t co = 0, cl = 0, cq = 0;
    for (size_t i = 0; i < no; ++i) co = co * 1315423911ull + h[i];
    for (size_t i = 0; i < nq; ++i) cq = cq * 1315423911ull + q[i];
    for (size_t i = 0; i < nl; ++i) cl = cl * 1315423911ull + L[i];
    fprintf(stderr, "[fa_sum] %s NS %d S %d H %d out@%p %016llx lse@%p %016llx qkv %016llx\n", what, a.NS, a.S, a.H, (void*)a.out, (unsigned long long)co, (void*)a.L,
            (unsigned long long)cl, (unsigned long long)cq);
}
#endif

static int fa_dispatch(int which, const FAArgs& a, hipStream_t s) {
#ifdef NBCI_MEASURE
    if (measure_env("NBCI_FA_CHECK", 0) && which == 1) fa_checksum("bwd", a, s);
    // NBCI_FA_CHECK=1 (measurement builds): every forward also runs the 16 x 32-tile kernels into scratch and reports the largest differences
    static const int chk = measure_env("NBCI_FA_CHECK", 0);
    if (chk && which == 0) {
        const size_t no = (size_t)a.NS * a.S * a.H, nl = (size_t)a.NS * a.nh * a.S;
        bf16_t* o2 = nullptr; float* l2 = nullptr;
        if (hipMalloc((void**)&o2, no * 2) != hipSuccess || hipMalloc((void**)&l2, nl * 4) != hipSuccess) return fail(NBCI_EHIP, "fa check: hipMalloc");
        FAArgs b = a; b.out = o2; b.L = l2;
        g_fa_force16 = chk == 2 ? 0 : 1;   // (2: the other way round - the 32 x 32 kernel writes the scratch, the 16 x 32 one the real buffers)
        int rc = fa_dispatch_raw(0, b, s);
        g_fa_force16 = chk == 2 ? 1 : 0;
        if (rc == NBCI_OK) rc = fa_dispatch_raw(0, a, s);
        g_fa_force16 = 0;
        (void)hipStreamSynchronize(s);
        std::vector<uint16_t> h1(no), h2(no);
        std::vector<float> L1(nl), L2(nl);
        (void)hipMemcpy(h1.data(), a.out, no * 2, hipMemcpyDeviceToHost); (void)hipMemcpy(h2.data(), o2, no * 2, hipMemcpyDeviceToHost);
        (void)hipMemcpy(L1.data(), a.L, nl * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(L2.data(), l2, nl * 4, hipMemcpyDeviceToHost);
        auto f = [](uint16_t v) { uint32_t u = (uint32_t)v << 16; float x; memcpy(&x, &u, 4); return x; };
        double mo = 0, ml = 0, so = 0, sr = 0; size_t wo = 0, wl = 0, nbad = 0;
        for (size_t i = 0; i < no; ++i) { const double d = fabs((double)f(h1[i]) - f(h2[i])); so += d; sr += fabs((double)f(h2[i])); if (d > mo) { mo = d; wo = i; } if (d > 0.02 * (fabs((double)f(h2[i])) + 0.01)) ++nbad; }
        for (size_t i = 0; i < nl; ++i) { const double d = fabs((double)L1[i] - L2[i]); if (d > ml) { ml = d; wl = i; } }
        fprintf(stderr, "[fa_check] NS %d nh %d S %d H %d mask %d thr %u thr_out %u cf %d cb %d: out l1rel %.3e, %zu elements off by > 2 %%, max diff %.3e at row %zu col %zu (%g vs %g); lse max diff %.3e at unit %zu query %zu (%g vs %g)\n",
                a.NS, a.nh, a.S, a.H, a.tmask != nullptr, a.thr, a.thr_out, a.cf, a.cb, so / (sr + 1e-30), nbad, mo, wo / a.H, wo % a.H, f(h1[wo]), f(h2[wo]), ml, wl / a.S, wl % a.S,
                L1[wl], L2[wl]);
        (void)hipFree(o2); (void)hipFree(l2);
        fa_checksum("fwd", a, s);
        return rc;
    }
#endif
    return fa_dispatch_raw(which, a, s);
}

int fattn_fwd_launch(const void* qkv, void* out, float* L, int NS, int nh, int S, int H, float drop_p, uint32_t seed, uint32_t site, hipStream_t s) {
    NBCI_REQUIRE(fattn_eligible(NBCI_BF16, S, H, nh), NBCI_ESHAPE, "flash attention: bf16, head size 32 / 64 / 96 / 128");
    ProfScope ps("fa_fwd_kernel", 4.0 * S * S * H * NS, 2.0 * NS * S * 4 * H, s);   // q, k, v in; merged output out
    FAArgs a{};
    int rc = fa_args(a, NS, nh, S, H, drop_p, seed, site);
    if (rc != NBCI_OK) return rc;
    a.qkv = (const bf16_t*)qkv; a.out = (bf16_t*)out; a.L = L;
    return fa_dispatch(0, a, s);
}

int fattn_bwd_launch(const void* qkv, const void* out, const void* dout, const float* L, float* Dsum, void* dqkv, int NS, int nh, int S, int H,
                     float drop_p, uint32_t seed, uint32_t site, hipStream_t s) {
    NBCI_REQUIRE(fattn_eligible(NBCI_BF16, S, H, nh), NBCI_ESHAPE, "flash attention: bf16, head size 32 / 64 / 96 / 128");
    FAArgs a{};
    int rc = fa_args(a, NS, nh, S, H, drop_p, seed, site);
    if (rc != NBCI_OK) return rc;
    a.qkv = (const bf16_t*)qkv; a.out = (bf16_t*)out; a.dout = (const bf16_t*)dout; a.L = (float*)L; a.Dsum = Dsum; a.dqkv = (bf16_t*)dqkv;
    // ONE profiling scope per launch (the per-kernel table is keyed by rocprof symbol: a scope over both launches filed their sum under
    // the second kernel's name). Algorithmic work of the backward = 10 S^2 H per sequence (S recomputed once, dP, dQ, dK, dV): the dq launch
    // carries S, dP, dQ (6), the dk / dv launch dK, dV (4; ITS recomputation of S and dP is not algorithmic work).
    {
        ProfScope ps("fa_bwd_q_kernel", 6.0 * S * S * H * NS, 2.0 * NS * S * 6 * H, s);    // q, k, v, out, d out in; dq out
        rc = fa_dispatch(1, a, s);
    }
    if (rc != NBCI_OK) return rc;
    ProfScope ps("fa_bwd_kv_kernel", 4.0 * S * S * H * NS, 2.0 * NS * S * 6 * H, s);       // q, k, v, d out in; dk, dv out
    return fa_dispatch(2, a, s);
}

// ---- masked variants for NDT1 (any length; the fused one-workgroup kernel of attention.hip covers T' <= 160 at head 128) ----
static void fa_mask_args(FAArgs& a, const int32_t* tmask, int cf, int cb, float drop_p, uint32_t seed, uint32_t site_out) {
    a.tmask = tmask; a.cf = cf; a.cb = cb;
    a.thr_out = drop_threshold(drop_p);
    a.oscale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    a.key_out = drop_key(seed, site_out);
}

int fattn_masked_fwd_launch(const void* qkv, const int32_t* tmask, void* out, float* L, int NS, int nh, int S, int H, int cf, int cb,
                            float drop_p, uint32_t seed, uint32_t site_prob, uint32_t site_out, hipStream_t s) {
    NBCI_REQUIRE(fattn_eligible(NBCI_BF16, S, H, nh) && tmask, NBCI_ESHAPE, "masked flash attention: bf16, head size 32 / 64 / 96 / 128, token mask");
    NBCI_REQUIRE((long long)NS * S * H < (1ll << 32), NBCI_ESHAPE, "masked flash attention: output too large for the 32-bit dropout counter");
    FAArgs a{};
    int rc = fa_args(a, NS, nh, S, H, drop_p, seed, site_prob);
    if (rc != NBCI_OK) return rc;
    fa_mask_args(a, tmask, cf, cb, drop_p, seed, site_out);
    a.qkv = (const bf16_t*)qkv; a.out = (bf16_t*)out; a.L = L;
    return fa_dispatch(0, a, s);
}

// dout = the gradient of the DROPPED output already multiplied by its keep mask (the caller's out-proj data-gradient GEMM does it)
int fattn_masked_bwd_launch(const void* qkv, const int32_t* tmask, const void* out, const void* dout, const float* L, float* Dsum, void* dqkv,
                            int NS, int nh, int S, int H, int cf, int cb, float drop_p, uint32_t seed, uint32_t site_prob, hipStream_t s) {
    NBCI_REQUIRE(fattn_eligible(NBCI_BF16, S, H, nh) && tmask, NBCI_ESHAPE, "masked flash attention: bf16, head size 32 / 64 / 96 / 128, token mask");
    FAArgs a{};
    int rc = fa_args(a, NS, nh, S, H, drop_p, seed, site_prob);
    if (rc != NBCI_OK) return rc;
    fa_mask_args(a, tmask, cf, cb, drop_p, seed, 0);
    a.qkv = (const bf16_t*)qkv; a.out = (bf16_t*)out; a.dout = (const bf16_t*)dout; a.L = (float*)L; a.Dsum = Dsum; a.dqkv = (bf16_t*)dqkv;
    {
        ProfScope ps("fa_bwd_q_kernel", 6.0 * S * S * H * NS, 2.0 * NS * S * 6 * H, s);
        rc = fa_dispatch(1, a, s);
    }
    if (rc != NBCI_OK) return rc;
    ProfScope ps("fa_bwd_kv_kernel", 4.0 * S * S * H * NS, 2.0 * NS * S * 6 * H, s);
    return fa_dispatch(2, a, s);
}

}  // namespace nbci
