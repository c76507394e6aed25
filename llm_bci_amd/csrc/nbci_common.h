// nbci_common.h — shared device/host helpers for the NDT1 hot-path kernels (gfx950 only).
//
// Everything in csrc/ is written for MI355X (gfx950, wave64). No portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>
#include <string>

namespace nbci {

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

// ---- error plumbing (thread-local last error string; C-ABI returns int status) ----
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define NBCI_OK 0
#define NBCI_EINVAL (-1)
#define NBCI_ESHAPE (-2)
#define NBCI_EALIGN (-3)
#define NBCI_EWORKSPACE (-4)
#define NBCI_EHIP (-5)

#define NBCI_CHECK_HIP(expr)                                                      \
    do {                                                                          \
        hipError_t _e = (expr);                                                   \
        if (_e != hipSuccess)                                                     \
            return ::nbci::fail(NBCI_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

#define TRY_(x)                       \
    do {                              \
        int _r = (x);                 \
        if (_r != NBCI_OK) return _r; \
    } while (0)

#define NBCI_REQUIRE(cond, code, msg)                    \
    do {                                                 \
        if (!(cond)) return ::nbci::fail((code), (msg)); \
    } while (0)

// Per-launch timing of ANY kernel of the library (bench.py's roofline leg; off by default, no cost beyond one branch): a scope records
// a HIP event on the launch's stream before and after it together with the launch's ALGORITHMIC work (flops, bytes: operands read
// once, results written once). The launcher that picks a template instantiation notes the kernel's symbol (as rocprofv3 prints it).
bool prof_on();
void prof_begin(const char* name, double flops, double bytes, int gemm_kind, hipStream_t s);
void prof_end(hipStream_t s);
void prof_note_symbol(const char* sym);
struct ProfScope {
    hipStream_t s; bool on;
    ProfScope(const char* name, double flops, double bytes, hipStream_t st, int kind = -1) : s(st), on(prof_on()) { if (on) prof_begin(name, flops, bytes, kind, st); }
    ~ProfScope() { if (on) prof_end(s); }
};
int prof_collect_text(char* buf, long long cap);   // one line per symbol: "symbol\tlaunches\tms\tflops\tbytes\n"

// Measurement switches. The environment variables that pick kernel variants for in-box A/B runs exist only in a
// MEASUREMENT build (tools/build_variant.sh measure -DNBCI_MEASURE, loaded with NBCI_LIB=...): the shipped library compiles the
// defaults in and reads no environment on its launch paths. (What a plan reads once at creation - NBCI_FUSED_ATTN, NBCI_FLASH_ATTN,
// NBCI_FLASH_MIN_TOKENS, the parity tests' reference paths - stays, stored in the plan.)
#ifdef NBCI_MEASURE
inline int measure_env(const char* name, int dflt) { const char* e = getenv(name); return (e && e[0]) ? atoi(e) : dflt; }
inline const char* measure_env_str(const char* name) { return getenv(name); }
#else
constexpr int measure_env(const char*, int dflt) { return dflt; }
constexpr const char* measure_env_str(const char*) { return nullptr; }
#endif

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (device, kernel): a process that drives two devices
// must set it on each. Remembers what was granted per (current device, kernel); thread-safe; a cheap map lookup on the launch path.
int ensure_dyn_lds(const void* kernel, int bytes, const char* what);
// CUs of the current device (hipDeviceProp.multiProcessorCount, cached per device) unless nbci_set_available_cus narrowed it
int available_cus();

// ---- bf16 <-> f32 (plain casts: hipcc emits v_cvt_pk_bf16_f32, NaN-preserving) ----
__device__ __forceinline__ float bf2f(bf16_t x) { return (float)x; }
__device__ __forceinline__ bf16_t f2bf(float x) { return (bf16_t)x; }

// four consecutive elements of a row as f32: an f32 row, or a bf16 one (a bf16 residual stream: half the bytes, widened here)
template <typename T>
__device__ __forceinline__ float4 ld4f(const T* p) {
    if constexpr (sizeof(T) == 4) {
        return *(const float4*)p;
    } else {
        const bf16x4 t = *(const bf16x4*)p;
        return make_float4(bf2f(t[0]), bf2f(t[1]), bf2f(t[2]), bf2f(t[3]));
    }
}
template <typename T>
__device__ __forceinline__ void st4f(T* p, const float4& v) {
    if constexpr (sizeof(T) == 4) {
        *(float4*)p = v;
    } else {
        const bf16x4 o = {f2bf(v.x), f2bf(v.y), f2bf(v.z), f2bf(v.w)};
        *(bf16x4*)p = o;
    }
}

// ---- stateless counter RNG -------------------------------------------------------
// One 32-bit draw per (seed, site, element index). "site" separates the dropout /
// noise call sites of one train step; the backward pass regenerates the same bits.
__host__ __device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ uint32_t rng_u32(uint32_t seed, uint32_t site, uint32_t idx) {
    uint32_t h = mix32(idx ^ (seed * 0x9E3779B9U + 0x85EBCA6BU));
    h = mix32(h ^ (site * 0xC2B2AE35U + 0x27D4EB2FU));
    return h;
}
// ---- dropout draws: ONE 32-bit hash per PAIR of elements, 16 bits each (GEMM epilogues own
// 4 consecutive elements per lane, so dropout costs two hashes per lane-quad; a full 32-bit
// draw per element made the epilogue VALU work rival the K=1024 MFMA work).
//   key  = drop_key(seed, site)                     (host, once per launch)
//   keep = drop_keep(key, thr16, idx)  with thr16 = floor(p * 65536): P(drop) = thr16 / 65536
__host__ __device__ __forceinline__ uint32_t drop_key(uint32_t seed, uint32_t site) {
    return mix32(seed * 0x9E3779B9U + site * 0x85EBCA6BU + 0x27D4EB2FU);
}
__host__ __device__ __forceinline__ uint32_t drop_threshold(float p) {
    double t = (double)p * 65536.0;
    if (t <= 0.0) return 0u;
    if (t >= 65535.0) return 65535u;
    return (uint32_t)t;
}
__host__ __device__ __forceinline__ uint32_t drop_pair(uint32_t key, uint32_t idx) { return mix32((idx >> 1) ^ key); }
__host__ __device__ __forceinline__ bool drop_keep(uint32_t key, uint32_t thr16, uint32_t idx) {
    const uint32_t h = drop_pair(key, idx);
    return ((idx & 1u) ? (h >> 16) : (h & 0xFFFFu)) >= thr16;
}
// 4 consecutive elements starting at an EVEN idx: two hashes
__device__ __forceinline__ void drop4(uint32_t key, uint32_t thr16, uint32_t idx, float scale, float (&v)[4]) {
    const uint32_t h0 = drop_pair(key, idx), h1 = drop_pair(key, idx + 2);
    v[0] = ((h0 & 0xFFFFu) >= thr16) ? v[0] * scale : 0.f;
    v[1] = ((h0 >> 16) >= thr16) ? v[1] * scale : 0.f;
    v[2] = ((h1 & 0xFFFFu) >= thr16) ? v[2] * scale : 0.f;
    v[3] = ((h1 >> 16) >= thr16) ? v[3] * scale : 0.f;
}
// 4 consecutive elements starting at ANY idx (the same bits as drop_keep per element): three pair hashes cover both parities
// (attention probabilities: row base = (row index) * T', odd for odd T'); out[r] = keep ? scale : 0
// (the four 16-bit draws are the 64-bit window at half-word offset (idx & 1) of ha | hb | hc: two funnel shifts instead of six selects; a high half
// is compared in place: hi >= thr <=> word >= thr << 16)
__device__ __forceinline__ void drop4_any(uint32_t key, uint32_t thr16, uint32_t idx, float scale, float (&out)[4]) {
    const uint32_t base = idx >> 1;
    const uint32_t ha = mix32(base ^ key), hb = mix32((base + 1u) ^ key), hc = mix32((base + 2u) ^ key);
    const uint32_t sh = (idx & 1u) << 4;
    const uint32_t w0 = __builtin_amdgcn_alignbit(hb, ha, sh), w1 = __builtin_amdgcn_alignbit(hc, hb, sh);
    const uint32_t thr_hi = thr16 << 16;
    out[0] = (w0 & 0xFFFFu) >= thr16 ? scale : 0.f; out[1] = w0 >= thr_hi ? scale : 0.f;
    out[2] = (w1 & 0xFFFFu) >= thr16 ? scale : 0.f; out[3] = w1 >= thr_hi ? scale : 0.f;
}
__device__ __forceinline__ float rng_uniform01(uint32_t u) {  // (0,1]
    return ((float)(u >> 8) + 1.0f) * (1.0f / 16777216.0f);
}
// standard normal via Box-Muller from two draws
__device__ __forceinline__ float rng_normal(uint32_t seed, uint32_t site, uint32_t idx) {
    uint32_t a = rng_u32(seed, site, idx);
    uint32_t b = rng_u32(seed ^ 0x5bd1e995U, site + 0x1000193U, idx);
    float u1 = rng_uniform01(a), u2 = rng_uniform01(b);
    return sqrtf(-2.0f * __logf(u1)) * __cosf(6.283185307179586f * u2);
}

// ---- replicated accumulators ---------------------------------------------------------------
// Float atomics to ONE address serialise at ~100 ns per adder on MI355X, so small vectors that many
// workgroups sum into (bias / LayerNorm gradients) are spread over `n` replicas `stride` floats
// apart; workgroup `blk` adds into replica blk % n and a fold kernel sums the replicas afterwards.
struct RepCfg { long long stride; int n; };
__device__ __forceinline__ float* rep_ptr(float* p, RepCfg rc, unsigned blk) {
    return rc.n > 1 ? p + (long long)(blk % (unsigned)rc.n) * rc.stride : p;
}

// optional fused tail of the LayerNorm backward: out = dropout(dx_new) in the GEMM operand dtype,
// colsum += column sums of out (the bias gradient of the Linear that produced the residual branch)
struct LnCast { void* out; int bf16; unsigned thr; float scale; uint32_t key; float* colsum; int rpg, gpitch, goff, nskip; };   // nskip: first rows of every rpg-row group left out of colsum   // rpg > 0: output row remap (kernels.hip ln_bwd)

// ---- wave64 reductions ------------------------------------------------------------
// sum over the 16 lanes of a DPP row (lanes 16 g .. 16 g + 15) with four VALU adds — quad swaps, then the mirrored half / row — instead
// of four ds_bpermute round trips through the LDS crossbar; every lane of the row ends up with the total
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    return v;
}

__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)));
    return v;
}
// whole-wave reductions (all 64 lanes active): the four steps inside a 16-lane row are DPP adds on the VALU, only the two steps across
// rows go through ds_bpermute (six of those per reduction sat on the per-row critical path of the LayerNorm kernels)
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    v = row16_max(v);
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}

// activation ids shared by GEMM epilogues and elementwise kernels
enum Act { ACT_NONE = 0, ACT_SOFTSIGN = 1, ACT_GELU = 2, ACT_RELU = 3, ACT_TANH = 4 };

// erf by Abramowitz & Stegun 7.1.26 (|abs err| <= 1.5e-7): ~12 VALU ops + one exp + one rcp instead
// of libm erff's ~40; the GELU sits in GEMM epilogues where every op per element is paid 2048 MACs apart.
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __frcp_rn(1.0f + 0.3275911f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float r = 1.0f - poly * __expf(-ax * ax);
    return copysignf(r, x);
}

__device__ __forceinline__ float act_fwd(int act, float x) {
    switch (act) {
        case ACT_SOFTSIGN: return x / (1.0f + fabsf(x));
        case ACT_GELU: return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752f));
        case ACT_RELU: return x > 0.f ? x : 0.f;
        case ACT_TANH: return tanhf(x);
        default: return x;
    }
}
__device__ __forceinline__ float act_bwd(int act, float x);
// activation and its derivative in one go (shares the erf / exp of the GELU): the forward GEMM can then
// store act'(pre) instead of pre, and the backward gate is a plain multiply
__device__ __forceinline__ void act_fwd_bwd(int act, float x, float& y, float& dy) {
    if (act == ACT_GELU) {
        const float z = x * 0.70710678118654752f, az = fabsf(z);
        const float t = __frcp_rn(1.0f + 0.3275911f * az);
        const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
        const float ex = __expf(-az * az);                 // = exp(-x^2 / 2)
        const float cdf = 0.5f * (1.0f + copysignf(1.0f - poly * ex, x));
        y = x * cdf;
        dy = cdf + x * 0.3989422804014327f * ex;
    } else {
        y = act_fwd(act, x);
        dy = act_bwd(act, x);
    }
}

// derivative wrt the pre-activation x
__device__ __forceinline__ float act_bwd(int act, float x) {
    switch (act) {
        case ACT_SOFTSIGN: { float d = 1.0f + fabsf(x); return 1.0f / (d * d); }
        case ACT_GELU: {
            float cdf = 0.5f * (1.0f + erf_fast(x * 0.70710678118654752f));
            float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
            return cdf + x * pdf;
        }
        case ACT_RELU: return x > 0.f ? 1.f : 0.f;
        case ACT_TANH: { float t = tanhf(x); return 1.f - t * t; }
        default: return 1.0f;
    }
}

// derivative wrt the pre-activation, from the activation's OUTPUT y (activations whose derivative is a function of y)
__device__ __forceinline__ float act_bwd_from_output(int act, float y) {
    switch (act) {
        case ACT_SOFTSIGN: { const float d = 1.0f - fabsf(y); return d * d; }  // y = x/(1+|x|) => 1/(1+|x|) = 1-|y|
        case ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case ACT_TANH: return 1.f - y * y;
        default: return 1.f;
    }
}

}  // namespace nbci
