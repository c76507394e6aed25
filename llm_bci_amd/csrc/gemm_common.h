// gemm_common.h — pieces shared by the two GEMM kernels (see gemm.hip for the design notes).
//
// C[z] = alpha * A[z](M x K) . B[z]^T(N x K), f32 accumulate, fused epilogue.
// Layouts: each operand is either k-major (storage rows = m/n, k contiguous) or
// row-major-in-k (storage rows = k, m/n contiguous). The second kind is read out of LDS with
// ds_read_b64_tr_b16 (bf16) so forward (x.W^T), data-grad (dy.W) and weight-grad (dy^T.x)
// products all run on one kernel without transposed copies in HBM.
// Storage rows may be an overlapping-window view (rpb/gstride): nn.Unfold + Linear of the
// reference (models/ndt1.py:138,180) is a plain GEMM over that view.
//
// Tile: 128 x 128 x BK (BK = 64 bf16 / 16 f32), 256 threads = 4 waves, each wave a 64 x 64
// sub-tile as 4 x 4 MFMA 16x16 blocks (v_mfma_f32_16x16x32_bf16 / v_mfma_f32_16x16x4_f32).
// The MFMA is issued with operands swapped (B fragment first) so every lane ends up with
// 4 CONSECUTIVE n for one m: epilogue loads/stores are 16-byte (f32) / 8-byte (bf16) wide.
// Register-staged global->LDS with two LDS stages: one barrier per K tile.
#pragma once
#include "nbci_common.h"
#include "../../include/nbci.h"

namespace nbci {

constexpr int GEMM_BM = 128;
constexpr int GEMM_BN = 128;
constexpr int GEMM_THREADS = 256;

template <typename T> struct GemmTile;
template <> struct GemmTile<bf16_t> {
    static constexpr int BK = 64;        // k per tile
    static constexpr int E = 8;          // elements per 16-byte chunk
    static constexpr int NCH = 4;        // chunks per thread per operand per tile
    static constexpr int REGION = 16384; // LDS bytes per operand per stage
};
template <> struct GemmTile<float> {
    static constexpr int BK = 16;
    static constexpr int E = 4;
    static constexpr int NCH = 2;
    static constexpr int REGION = 9216;  // max(128*17*4, 16*144*4)
};
constexpr int F32_KM_STRIDE = 17;   // dwords per row, k-major f32 tile [128][16+1]
constexpr int F32_RM_STRIDE = 144;  // dwords per row, row-major f32 tile [16][128+16]

#ifdef NBCI_STAMPS   // measurement build only (tools/gemm_stamps.py): per-workgroup wall-clock stamps (100 MHz) of the kernel's phases
static __device__ unsigned long long g_stamps[8192 * 8];
static __device__ unsigned long long g_kstamps[1024 * 64];   // [workgroup][K tile 4..11][slot]
#define STAMP(slot)                                                                                   \
    do {                                                                                              \
        if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 8192) g_stamps[blockIdx.x * 8 + (slot)] = wall_clock64(); \
    } while (0)
#else
#define STAMP(slot) do { } while (0)
#endif

struct OperandK {  // device copy of nbci_operand, batch offset already applied
    const void* ptr;
    long long ld;
    int rpb;
    long long gstride;
    int vec;  // 16-byte loads legal
};

struct GemmK {
    int M, N, K;
    OperandK A, B;
    long long azs1, azs2, bzs1, bzs2;
    void* C; void* C2;
    long long ldc, czs1, czs2;
    int c_bf16;
    int zdiv;
    int splitk;
    int tiles_per_split;
    int tiles_m, tiles_n;
    float alpha, beta;
    const float* bias;
    int act;
    float drop_scale; unsigned drop_thr; unsigned drop_key;
    float* colsum;  // optional: colsum[(coff % ldc) + n] += sum over rows of the stored value
    RepCfg colsum_rc; int tile_row;  // replica config; tile_row = replica selector (set by the kernel)
    const float* residual; long long ldr;   // (bf16 storage when residual_bf16: the pointer is then a bf16_t*)
    int residual_bf16;
    const long long* residual_rows;  // optional gather: residual row for output row m
    int residual_first;              // add residual before act/dropout (embed: proj + pos, then dropout)
    const void* gate; long long ldg; int gate_act;  // v *= act'(gate[m][n]) (gate in the input dtype)
    int gate_bf16;
    int gate_coff;   // the batch offset of C applies to the gate too (gate has C's layout)
    int c2_grad;   // C2 receives act'(pre-activation) instead of the pre-activation
    int cvec;  // vector C/residual accesses legal
    int epi_mode;   // EPI_* feature bits of this problem's epilogue (epi_mode_of), set by the host: picks a specialised row loop
    int dbg;   // NBCI_GEMM_DBG ablation bits (measurement only): 1 = epilogue computes but does not store, 2 = no K loop (both outside the K loop: a flag tested inside it slows the loop itself)
};

__device__ __forceinline__ long long row_offset(const OperandK& o, int r) {
    if (o.rpb > 0) return (long long)(r / o.rpb) * o.gstride + (long long)(r % o.rpb) * o.ld;
    return (long long)r * o.ld;
}

// f(k) used to XOR-swizzle the 32-byte granules of the row-major bf16 tile [64][128]:
// the 8 rows touched by one 32-lane half of a ds_read_b64_tr_b16 get 8 distinct granules.
__device__ __forceinline__ int rm_swz(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

// ---- global -> registers --------------------------------------------------------------
template <typename T, bool KMAJOR>
__device__ __forceinline__ void load_chunks(const OperandK& o, int row0, int R, int k0, int K,
                                            uint4 (&regs)[GemmTile<T>::NCH], int t) {
    constexpr int E = GemmTile<T>::E;
    constexpr int BK = GemmTile<T>::BK;
    constexpr int CPR = KMAJOR ? (BK / E) : (128 / E);  // chunks per storage row of the tile
    constexpr int RPP = GEMM_THREADS / CPR;             // storage rows per pass
    const int lc = (t % CPR) * E;
    const int lr = t / CPR;
    const int rows_lim = KMAJOR ? R : K;
    const int cols_lim = KMAJOR ? K : R;
    const T* base = (const T*)o.ptr;
#pragma unroll
    for (int i = 0; i < GemmTile<T>::NCH; ++i) {
        const int srow = (KMAJOR ? row0 : k0) + lr + i * RPP;
        const int scol = (KMAJOR ? k0 : row0) + lc;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (srow < rows_lim && scol < cols_lim) {
            const T* p = base + row_offset(o, srow) + scol;
            if (o.vec && scol + E <= cols_lim) {
                v = *(const uint4*)p;
            } else {
                alignas(16) T tmp[E];
#pragma unroll
                for (int e = 0; e < E; ++e) tmp[e] = (scol + e < cols_lim) ? p[e] : (T)0.0f;
                v = *(const uint4*)tmp;
            }
        }
        regs[i] = v;
    }
}

// ---- registers -> LDS -----------------------------------------------------------------
template <typename T, bool KMAJOR>
__device__ __forceinline__ void store_chunks(char* s, const uint4 (&regs)[GemmTile<T>::NCH], int t) {
    constexpr int E = GemmTile<T>::E;
    constexpr int BK = GemmTile<T>::BK;
    constexpr int CPR = KMAJOR ? (BK / E) : (128 / E);
    constexpr int RPP = GEMM_THREADS / CPR;
    const int c = t % CPR;
    const int lr = t / CPR;
#pragma unroll
    for (int i = 0; i < GemmTile<T>::NCH; ++i) {
        const int row = lr + i * RPP;
        if constexpr (sizeof(T) == 2) {
            if constexpr (KMAJOR) {  // [128][64] bf16, 128-B rows, chunk ^= (row>>1)&7
                *(uint4*)(s + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = regs[i];
            } else {                 // [64][128] bf16, 256-B rows, 32-B granule ^= rm_swz(k)
                const int phys = ((((c >> 1) ^ rm_swz(row)) << 1) | (c & 1));
                *(uint4*)(s + row * 256 + (phys << 4)) = regs[i];
            }
        } else {
            if constexpr (KMAJOR) {  // [128][17] f32 (padded: scalar stores)
                float* d = (float*)s + row * F32_KM_STRIDE + c * 4;
                d[0] = __uint_as_float(regs[i].x); d[1] = __uint_as_float(regs[i].y);
                d[2] = __uint_as_float(regs[i].z); d[3] = __uint_as_float(regs[i].w);
            } else {                 // [16][144] f32
                *(uint4*)((float*)s + row * F32_RM_STRIDE + c * 4) = regs[i];
            }
        }
    }
}

// ---- LDS -> MFMA fragments (bf16) -------------------------------------------------------
template <bool KMAJOR>
__device__ __forceinline__ bf16x8 read_frag_bf16(const char* s, int r0, int ks, int i16, int g) {
    if constexpr (KMAJOR) {
        const int row = r0 + i16;
        const int chunk = 4 * ks + g;
        return *(const bf16x8*)(s + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
    } else {
        // lane 4q+p of each 16-lane group addresses row q, columns 4p..4p+3 of a 4x16 block;
        // lane i receives column i (= m/n index r0+i), rows in elements 0..3.
        const int q = i16 >> 2, p = i16 & 3;
        const int col = r0 + 4 * p;
        s16x4 lo, hi;
        {
            const int krow = 32 * ks + 8 * g + q;
            const char* a = s + krow * 256 + ((((col >> 4) ^ rm_swz(krow)) << 5) | ((col & 15) << 1));
            lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
        }
        {
            const int krow = 32 * ks + 8 * g + 4 + q;
            const char* a = s + krow * 256 + ((((col >> 4) ^ rm_swz(krow)) << 5) | ((col & 15) << 1));
            hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
        }
        union { struct { s16x4 a, b; } p2; bf16x8 v; } u;
        u.p2.a = lo; u.p2.b = hi;
        return u.v;
    }
}

template <typename T, bool AK, bool BKM>
__device__ __forceinline__ void compute_tile(const char* sA, const char* sB, f32x4 (&acc)[4][4],
                                             int wm, int wn, int lane) {
    const int i16 = lane & 15, g = lane >> 4;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], bf[4];
#pragma unroll
            for (int sb = 0; sb < 4; ++sb) af[sb] = read_frag_bf16<AK>(sA, wm * 64 + sb * 16, ks, i16, g);
#pragma unroll
            for (int sb = 0; sb < 4; ++sb) bf[sb] = read_frag_bf16<BKM>(sB, wn * 64 + sb * 16, ks, i16, g);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[ni], af[mi], acc[mi][ni], 0, 0, 0);
        }
    } else {
        const float* fA = (const float*)sA;
        const float* fB = (const float*)sB;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int sb = 0; sb < 4; ++sb) {
                const int ra = wm * 64 + sb * 16 + i16;
                a[sb] = AK ? fA[ra * F32_KM_STRIDE + 4 * kk + g] : fA[(4 * kk + g) * F32_RM_STRIDE + ra];
                const int rb = wn * 64 + sb * 16 + i16;
                b[sb] = BKM ? fB[rb * F32_KM_STRIDE + 4 * kk + g] : fB[(4 * kk + g) * F32_RM_STRIDE + rb];
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[ni], a[mi], acc[mi][ni], 0, 0, 0);
        }
    }
}

// One lane's share of the fused epilogue: v[0..3] = alpha * accumulator of C[m][n .. n+3] (m < M, n < N checked by
// the caller). Applies bias / activation / gate / dropout / residual in the order the model needs, stores C (and C2),
// and adds the stored values into csum[0..3] when column sums are requested.
// The activation ladders with the switch OUTSIDE the 4-element loop: one uniform branch ladder per output row instead of four
// (called with a constant code, act_fwd & co fold to the one formula: same arithmetic, same bits).
__device__ __forceinline__ void act_fwd4(int act, float (&v)[4]) {
#define NBCI_ACT4(CODE) case CODE: _Pragma("unroll") for (int e = 0; e < 4; ++e) v[e] = act_fwd(CODE, v[e]); break;
    switch (act) { NBCI_ACT4(ACT_SOFTSIGN) NBCI_ACT4(ACT_GELU) NBCI_ACT4(ACT_RELU) NBCI_ACT4(ACT_TANH) default: break; }
#undef NBCI_ACT4
}
__device__ __forceinline__ void act_fwd_bwd4(int act, float (&v)[4], float (&dact)[4]) {
#define NBCI_ACT4(CODE) case CODE: _Pragma("unroll") for (int e = 0; e < 4; ++e) { float y; act_fwd_bwd(CODE, v[e], y, dact[e]); v[e] = y; } break;
    switch (act) {
        NBCI_ACT4(ACT_SOFTSIGN) NBCI_ACT4(ACT_GELU) NBCI_ACT4(ACT_RELU) NBCI_ACT4(ACT_TANH)
        default: _Pragma("unroll") for (int e = 0; e < 4; ++e) { float y; act_fwd_bwd(ACT_NONE, v[e], y, dact[e]); v[e] = y; } break;
    }
#undef NBCI_ACT4
}
__device__ __forceinline__ void act_bwd_from_output_mul4(int act, const float (&g)[4], float (&v)[4]) {
#define NBCI_ACT4(CODE) case CODE: _Pragma("unroll") for (int e = 0; e < 4; ++e) v[e] *= act_bwd_from_output(CODE, g[e]); break;
    switch (act) {
        NBCI_ACT4(ACT_SOFTSIGN) NBCI_ACT4(ACT_GELU) NBCI_ACT4(ACT_RELU) NBCI_ACT4(ACT_TANH)
        default: _Pragma("unroll") for (int e = 0; e < 4; ++e) v[e] *= act_bwd_from_output(ACT_NONE, g[e]); break;
    }
#undef NBCI_ACT4
}

__device__ __forceinline__ void act_bwd_mul4(int act, const float (&g)[4], float (&v)[4]) {
#define NBCI_ACT4(CODE) case CODE: _Pragma("unroll") for (int e = 0; e < 4; ++e) v[e] *= act_bwd(CODE, g[e]); break;
    switch (act) {
        NBCI_ACT4(ACT_SOFTSIGN) NBCI_ACT4(ACT_GELU) NBCI_ACT4(ACT_RELU) NBCI_ACT4(ACT_TANH)
        default: _Pragma("unroll") for (int e = 0; e < 4; ++e) v[e] *= act_bwd(ACT_NONE, g[e]); break;
    }
#undef NBCI_ACT4
}

// FULL: the lane's 4 elements are inside N and 16-byte accesses are legal -- decided ONCE per thread by the caller, so the row
// loop carries one copy of each step instead of a vector / scalar branch at every load and store. (In-kernel stamps,
// tools/gemm_stamps.py: the branch maze of the one-size-fits-all version cost ~900 cycles per output row, 7.2 us of a 24 us
// K = 1024 tile.)
// MODE >= 0: the feature set is a compile-time constant (EPI_* bits) and the row loop is straight-line code for exactly that
// epilogue; MODE < 0: every feature is tested at run time (any combination). The row-contiguous epilogue issues one wave64
// VALU instruction per 4 cycles per SIMD with two waves per SIMD: at ~70 instructions per 4 outputs the run-time-tested loop
// cost 6 us of a 24 us K = 1024 tile after the first clean-up (7.2 before).
enum : int {
    EPI_BIAS = 1, EPI_C2GRAD = 2, EPI_RES_FIRST = 4, EPI_RES_LAST = 8, EPI_RES_ROWS = 16, EPI_ACT = 32, EPI_GATE_MUL = 64,
    EPI_GATE_OUT = 128, EPI_DROP = 256, EPI_COLSUM = 512, EPI_CBF16 = 1024, EPI_BETA = 2048, EPI_GATE_PRE = 4096, EPI_RESBF16 = 8192, EPI_GENERIC = 1 << 30
};
__host__ __device__ inline int epi_mode_of(const GemmK& d) {
    int m = 0;
    if (d.bias) m |= EPI_BIAS;
    if (d.C2) m |= d.c2_grad ? EPI_C2GRAD : EPI_GENERIC;
    if (d.residual) m |= d.residual_first ? EPI_RES_FIRST : EPI_RES_LAST;
    if (d.residual && d.residual_rows) m |= EPI_RES_ROWS;
    if (d.residual && d.residual_bf16) m |= EPI_RESBF16;
    if (d.act != 0 && !(d.C2 && d.c2_grad)) m |= EPI_ACT;
    if (d.gate) {
        const bool fast = d.gate_bf16 && (d.ldg & 3) == 0;   // one 8-byte load per 4 outputs
        m |= !fast ? EPI_GENERIC : (d.gate_act < 0 ? EPI_GATE_MUL : (d.gate_act >= 64 ? EPI_GATE_OUT : EPI_GATE_PRE));
    }
    if (d.drop_thr) m |= EPI_DROP;
    if (d.colsum) m |= EPI_COLSUM;
    if (d.c_bf16) m |= EPI_CBF16;
    if (d.beta != 0.f) m |= d.c_bf16 ? EPI_GENERIC : EPI_BETA;
    return m;
}
#define EPI_HAS(bit, cond) ((MODE < 0) ? bool(cond) : bool(MODE & (bit)))

template <bool FULL, int MODE = -1>
__device__ __forceinline__ void epi_apply_t(const GemmK& d, float (&v)[4], int m, int n, long long coff, float (&csum)[4]) {
    static_assert(FULL || MODE < 0, "specialised epilogues are for whole 4-element groups");
    const long long cidx = coff + (long long)m * d.ldc + n;
    if (EPI_HAS(EPI_BIAS, d.bias)) {
        if constexpr (FULL) {
            const float4 b4 = *(const float4*)(d.bias + n);
            v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (n + e < d.N) v[e] += d.bias[n + e];
        }
    }
    float dact[4] = {1.f, 1.f, 1.f, 1.f};
    bool act_done = false;
    if (EPI_HAS(EPI_C2GRAD, d.C2 && d.c2_grad)) {
        act_fwd_bwd4(d.act, v, dact);
        act_done = true;
    }
    if (EPI_HAS(EPI_C2GRAD, d.C2)) {
        const float* src = EPI_HAS(EPI_C2GRAD, d.c2_grad) ? dact : v;
        if (EPI_HAS(EPI_CBF16, d.c_bf16)) {
            bf16_t* c2 = (bf16_t*)d.C2 + cidx;
            if constexpr (FULL) { bf16x4 o = {f2bf(src[0]), f2bf(src[1]), f2bf(src[2]), f2bf(src[3])}; *(bf16x4*)c2 = o; }
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (n + e < d.N) c2[e] = f2bf(src[e]);
            }
        } else {
            float* c2 = (float*)d.C2 + cidx;
            if constexpr (FULL) *(float4*)c2 = make_float4(src[0], src[1], src[2], src[3]);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (n + e < d.N) c2[e] = src[e];
            }
        }
    }
    auto add_residual = [&]() {
        const long long rr = EPI_HAS(EPI_RES_ROWS, d.residual_rows) ? d.residual_rows[(coff ? coff / d.ldc : 0) + m] : (long long)m;   // gather index: the GLOBAL output row (batched GEMMs: coff = batch offset)
        if (EPI_HAS(EPI_RESBF16, d.residual_bf16)) {   // a bf16 residual stream: widened here, the sum is rounded once at the store
            const bf16_t* r = (const bf16_t*)d.residual + rr * d.ldr + n;
            if constexpr (FULL) { const bf16x4 r4 = *(const bf16x4*)r; v[0] += bf2f(r4[0]); v[1] += bf2f(r4[1]); v[2] += bf2f(r4[2]); v[3] += bf2f(r4[3]); }
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (n + e < d.N) v[e] += bf2f(r[e]);
            }
            return;
        }
        const float* r = d.residual + rr * d.ldr + n;
        if constexpr (FULL) { const float4 r4 = *(const float4*)r; v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w; }
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (n + e < d.N) v[e] += r[e];
        }
    };
    if (EPI_HAS(EPI_RES_FIRST, d.residual && d.residual_first)) add_residual();
    if (EPI_HAS(EPI_ACT, d.act != ACT_NONE && !act_done)) act_fwd4(d.act, v);
    if (EPI_HAS(EPI_GATE_MUL | EPI_GATE_OUT | EPI_GATE_PRE, d.gate)) {
        bool done = false;
        if constexpr (FULL) {
            if (EPI_HAS(EPI_GATE_MUL | EPI_GATE_OUT | EPI_GATE_PRE, d.gate_bf16 && (d.ldg & 3) == 0)) {   // the train step's cases: one 8-byte load
                const bf16x4 g4 = *(const bf16x4*)((const bf16_t*)d.gate + (d.gate_coff ? coff : 0) + (long long)m * d.ldg + n);
                const float g[4] = {bf2f(g4[0]), bf2f(g4[1]), bf2f(g4[2]), bf2f(g4[3])};
                if (EPI_HAS(EPI_GATE_MUL, d.gate_act < 0)) { v[0] *= g[0]; v[1] *= g[1]; v[2] *= g[2]; v[3] *= g[3]; }
                else if (EPI_HAS(EPI_GATE_OUT, d.gate_act >= 64)) act_bwd_from_output_mul4(d.gate_act - 64, g, v);   // gate = the activation's output
                else act_bwd_mul4(d.gate_act, g, v);                                                                    // gate = the pre-activation
                done = true;
            }
        }
        if (!done) {
            const long long gi = (d.gate_coff ? coff : 0) + (long long)m * d.ldg + n;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (n + e < d.N) {
                    const float gv = d.gate_bf16 ? bf2f(((const bf16_t*)d.gate)[gi + e]) : ((const float*)d.gate)[gi + e];
                    // gate_act < 0: gate already holds act'; >= 64: gate holds the activation's output
                    v[e] *= (d.gate_act < 0) ? gv : (d.gate_act >= 64 ? act_bwd_from_output(d.gate_act - 64, gv) : act_bwd(d.gate_act, gv));
                }
            }
        }
    }
    if (EPI_HAS(EPI_DROP, d.drop_thr)) {
        // dropout stream index = element offset inside C (so a head-batched GEMM that writes
        // the merged (B*T', H) layout draws the same bits as a flat pass over that layout)
        const unsigned idx = (unsigned)cidx;
        if ((idx & 1u) == 0u) {
            drop4(d.drop_key, d.drop_thr, idx, d.drop_scale, v);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = drop_keep(d.drop_key, d.drop_thr, idx + e) ? v[e] * d.drop_scale : 0.f;
        }
    }
    if (EPI_HAS(EPI_RES_LAST, d.residual && !d.residual_first)) add_residual();
    if (EPI_HAS(EPI_COLSUM, d.colsum)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) csum[e] += v[e];
    }
    if (EPI_HAS(EPI_CBF16, d.c_bf16)) {
        bf16_t* c = (bf16_t*)d.C + cidx;
        if constexpr (FULL) { bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])}; *(bf16x4*)c = o; }
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (n + e < d.N) c[e] = f2bf(v[e]);
        }
    } else {
        float* c = (float*)d.C + cidx;
        if (EPI_HAS(EPI_BETA, d.beta != 0.f)) {
            if constexpr (FULL) { const float4 o = *(const float4*)c; v[0] += d.beta * o.x; v[1] += d.beta * o.y; v[2] += d.beta * o.z; v[3] += d.beta * o.w; }
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (n + e < d.N) v[e] += d.beta * c[e];
            }
        }
        if constexpr (FULL) *(float4*)c = make_float4(v[0], v[1], v[2], v[3]);
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (n + e < d.N) c[e] = v[e];
        }
    }
}

__device__ __forceinline__ void epi_apply(const GemmK& d, float (&v)[4], int m, int n, long long coff, float (&csum)[4]) {
    if ((n + 3 < d.N) && d.cvec) epi_apply_t<true>(d, v, m, n, coff, csum);
    else epi_apply_t<false>(d, v, m, n, coff, csum);
}

// ---- epilogue shared by both kernels. The wave owns rows [mw, mw + 16*MI) x cols [nw, nw + 16*NI);
// lane owns m = mw + 16*mi + (lane&15), n = nw + 16*ni + 4*(lane>>4) + 0..3.
template <int MI, int NI>
__device__ __forceinline__ void gemm_epilogue(const GemmK& d, f32x4 (&acc)[MI][NI], int mw, int nw, long long coff,
                                              int lane, int w, char* smem) {
    const int i16 = lane & 15, g = lane >> 4;
    if constexpr (MI == 4 && NI == 4) {
    if (d.splitk > 1) {
        // Split-K partials go to C with f32 atomics. A wave-instruction of float atomics runs at full
        // rate only when it covers 256 contiguous bytes, so each wave first transposes its 64x64
        // accumulator through LDS (XOR-swizzled float4 columns: conflict-free both ways) and then
        // issues one atomic per ROW: 64 lanes = 64 consecutive n.
        float* sw = (float*)smem + w * 4096;  // 16 KB per wave; main-loop LDS is dead after the last barrier
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int ml = mi * 16 + i16, c4 = ni * 4 + g;
                *(float4*)(sw + ml * 64 + ((c4 ^ (ml & 15)) << 2)) =
                    make_float4(acc[mi][ni][0] * d.alpha, acc[mi][ni][1] * d.alpha, acc[mi][ni][2] * d.alpha, acc[mi][ni][3] * d.alpha);
            }
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS writes have landed (wave-private region)
        const int nn = nw + lane;
        float* cbase = (float*)d.C + nn;
        for (int ml = 0; ml < 64; ++ml) {
            const int m = mw + ml;
            if (m >= d.M) break;
            const float val = sw[ml * 64 + ((((lane >> 2) ^ (ml & 15)) << 2) | (lane & 3))];
            if (nn < d.N) atomicAdd(cbase + (long long)m * d.ldc, val);
        }
        return;
    }
    }
    if (d.dbg & 1) {   // ablation: keep the accumulators alive, store nothing
        float t = 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) t += acc[mi][ni][0] + acc[mi][ni][1] + acc[mi][ni][2] + acc[mi][ni][3];
        if (t == 1.2345e-30f) ((float*)d.C)[0] = t;
        return;
    }
    float csum[NI][4];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) csum[a][b] = 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = mw + mi * 16 + i16;
        if (m >= d.M) continue;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = nw + ni * 16 + 4 * g;
            if (n >= d.N) continue;
            float v[4] = {acc[mi][ni][0] * d.alpha, acc[mi][ni][1] * d.alpha,
                          acc[mi][ni][2] * d.alpha, acc[mi][ni][3] * d.alpha};
            epi_apply(d, v, m, n, coff, csum[ni]);
        }
    }
    if (d.colsum) {
        // bias gradient fused into the producing GEMM: reduce this wave's 64 rows (16 lanes x 4 mi)
        // with shuffles, then one atomic per column per wave
        const int cbase = (int)(coff % d.ldc);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = csum[ni][e];
                t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 8, 64);
                const int n = nw + ni * 16 + 4 * g + e;
                if (i16 == 0 && n < d.N) atomicAdd(rep_ptr(d.colsum, d.colsum_rc, (unsigned)(mw >> 4) + (unsigned)(coff / d.ldc)) + cbase + n, t);
            }
    }
}

// ---- row-contiguous epilogue for the direct-to-LDS kernels (no split-K). In the accumulator layout a wave store
// covers 16 rows x 64 B (f32) or 32 B (bf16): partial cache lines, assembled in L2 from several instructions of
// several waves while the K loop of the co-resident workgroup streams through the same L2. Here the accumulators
// go through the (dead) staging LDS once, as an f32 tile [BM][128 + 4]; afterwards a wave instruction covers whole
// rows: lane owns n = 4 * (t & 31) .. + 3 of row t / 32 (+ nthreads / 32 per pass): 512 B (f32) / 256 B (bf16)
// contiguous per row for C, C2, the residual and the gate. The per-element work is epi_apply(), unchanged.
constexpr int EPI_LD = 132;   // floats per LDS row: 16 B of padding spreads the 16 rows of a fragment over all banks
// The row loop over an f32 LDS tile [rows][EPI_LD] of 128 columns: thread t owns columns 4 * (t & 31) .. + 3 of rows t / 32
// (+ nthreads / 32 per pass); m_first = global row of the tile's row 0, n0 = global column of its column 0.
__device__ __forceinline__ void epi_tile_rows(const GemmK& d, const float* tile, int rows, int m_first, int n0, long long coff, int t,
                                              int nthreads, float (&csum)[4]) {
    const int c4 = 4 * (t & 31), n = n0 + c4;
    if (n < d.N) {
        const int rstep = nthreads >> 5;
        if ((n + 3 < d.N) && d.cvec) {   // (the thread's columns are fixed: one decision for all its rows)
#define NBCI_EPI_ROWS(MODE_)                                                         \
    for (int r = t >> 5; r < rows; r += rstep) {                                     \
        const int m = m_first + r;                                                   \
        if (m >= d.M) break;                                                         \
        const float4 a = *(const float4*)(tile + r * EPI_LD + c4);                   \
        float v[4] = {a.x, a.y, a.z, a.w};                                           \
        epi_apply_t<true, MODE_>(d, v, m, n, coff, csum);                            \
    }
#define NBCI_EPI_CASE(MODE_) case (MODE_): NBCI_EPI_ROWS(MODE_) break;
                switch (d.epi_mode) {   // the train steps' epilogues, straight-line; anything else: the run-time-tested loop
                    NBCI_EPI_CASE(0)
                    NBCI_EPI_CASE(EPI_CBF16)
                    NBCI_EPI_CASE(EPI_BIAS)
                    NBCI_EPI_CASE(EPI_BIAS | EPI_CBF16)
                    NBCI_EPI_CASE(EPI_BIAS | EPI_ACT | EPI_CBF16)
                    NBCI_EPI_CASE(EPI_BIAS | EPI_C2GRAD | EPI_CBF16)
                    NBCI_EPI_CASE(EPI_BIAS | EPI_DROP | EPI_RES_LAST)
                    NBCI_EPI_CASE(EPI_BIAS | EPI_RES_LAST)
                    NBCI_EPI_CASE(EPI_BIAS | EPI_RES_FIRST | EPI_RES_ROWS | EPI_DROP)
                    NBCI_EPI_CASE(EPI_BIAS | EPI_RES_FIRST | EPI_RES_ROWS)
                    NBCI_EPI_CASE(EPI_BIAS | EPI_RES_FIRST | EPI_RES_ROWS | EPI_DROP | EPI_CBF16)              // bf16 residual stream (NDT1)
                    NBCI_EPI_CASE(EPI_BIAS | EPI_RES_FIRST | EPI_RES_ROWS | EPI_CBF16)
                    NBCI_EPI_CASE(EPI_BIAS | EPI_DROP | EPI_RES_LAST | EPI_RESBF16 | EPI_CBF16)
                    NBCI_EPI_CASE(EPI_BIAS | EPI_RES_LAST | EPI_RESBF16 | EPI_CBF16)
                    NBCI_EPI_CASE(EPI_RES_LAST | EPI_RESBF16 | EPI_CBF16)                                      // bf16 gradient stream + data gradient (iTransformer)
                    NBCI_EPI_CASE(EPI_DROP | EPI_CBF16)
                    NBCI_EPI_CASE(EPI_COLSUM | EPI_CBF16)
                    NBCI_EPI_CASE(EPI_GATE_MUL | EPI_COLSUM | EPI_CBF16)
                    NBCI_EPI_CASE(EPI_GATE_MUL | EPI_CBF16)
                    NBCI_EPI_CASE(EPI_GATE_OUT | EPI_CBF16)
                    NBCI_EPI_CASE(EPI_BETA)
                    NBCI_EPI_CASE(EPI_RES_LAST)                                             // iTransformer / PatchTST
                    NBCI_EPI_CASE(EPI_BIAS | EPI_DROP)
                    NBCI_EPI_CASE(EPI_BIAS | EPI_ACT | EPI_DROP | EPI_CBF16)
                    NBCI_EPI_CASE(EPI_BIAS | EPI_C2GRAD | EPI_DROP | EPI_CBF16)
                    NBCI_EPI_CASE(EPI_GATE_MUL | EPI_DROP | EPI_COLSUM | EPI_CBF16)
                    NBCI_EPI_CASE(EPI_GATE_PRE | EPI_DROP | EPI_COLSUM)
                    NBCI_EPI_CASE(EPI_GATE_PRE | EPI_COLSUM | EPI_CBF16)
                    NBCI_EPI_CASE(EPI_GATE_PRE | EPI_DROP | EPI_COLSUM | EPI_CBF16)
                    default: NBCI_EPI_ROWS(-1) break;
                }
#undef NBCI_EPI_CASE
#undef NBCI_EPI_ROWS
        } else {
            for (int r = t >> 5; r < rows; r += rstep) {
                const int m = m_first + r;
                if (m >= d.M) break;
                const float4 a = *(const float4*)(tile + r * EPI_LD + c4);
                float v[4] = {a.x, a.y, a.z, a.w};
                epi_apply_t<false>(d, v, m, n, coff, csum);
            }
        }
    }
}

// CH > 1: the tile goes through LDS in CH row chunks (tall tiles whose f32 image would not leave room for two
// workgroups per CU); needs every wave to span all rows (mw_l = 0) and MI % CH == 0.
template <int MI, int NI, int CH = 1>
__device__ __forceinline__ void gemm_epilogue_tile(const GemmK& d, f32x4 (&acc)[MI][NI], int mw_l, int nw_l, int m0, int n0, int bm,
                                                   long long coff, int t, int nthreads, char* smem) {
    static_assert(MI % CH == 0, "row chunks must split the wave's row blocks evenly");
    constexpr int MC = MI / CH;
    const int lane = t & 63, i16 = lane & 15, g = lane >> 4;
    float* tile = (float*)smem;
    if (d.dbg & 1) {   // ablation: keep the accumulators alive, store nothing
        float s = 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) s += acc[mi][ni][0] + acc[mi][ni][1] + acc[mi][ni][2] + acc[mi][ni][3];
        if (s == 1.2345e-30f) ((float*)d.C)[0] = s;
        return;
    }
    const int c4 = 4 * (t & 31), n = n0 + c4;
    float csum[4] = {0.f, 0.f, 0.f, 0.f};
    const int rows = bm / CH;
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
        __syncthreads();   // every wave has left the K loop (the previous chunk): the LDS tile is free
#pragma unroll
        for (int mi = 0; mi < MC; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                *(float4*)(tile + (mw_l + mi * 16 + i16) * EPI_LD + nw_l + ni * 16 + 4 * g) =
                    make_float4(acc[ch * MC + mi][ni][0] * d.alpha, acc[ch * MC + mi][ni][1] * d.alpha, acc[ch * MC + mi][ni][2] * d.alpha,
                                acc[ch * MC + mi][ni][3] * d.alpha);
        __syncthreads();
        if (ch == 0) STAMP(6);
        epi_tile_rows(d, tile, rows, m0 + ch * rows, n0, coff, t, nthreads, csum);
    }
    if (d.colsum) {
        // bias gradient: the waves' column sums (the two half-waves of a wave hold the same columns) meet in the - now free - LDS tile and
        // ONE lane-contiguous atomic per column leaves the workgroup: 2 wave-instructions of 64 consecutive floats per tile. (Before, every
        // wave added its own four adjacent columns per lane: 4 x waves instructions per tile, each touching 32 words spread over 512 B -
        // the shape f32 atomics are slowest in; it showed in the short-K projections of PatchTST.)
        const int cbase = (int)(coff % d.ldc);
        const int nwaves = nthreads >> 6;
        __syncthreads();   // the row loops are done with the tile
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float s = csum[e];
            s += __shfl_xor(s, 32, 64);
            if (lane < 32) tile[(t >> 6) * 128 + c4 + e] = s;
        }
        __syncthreads();
        if (t < 128 && n0 + t < d.N) {
            float s = 0.f;
            for (int w2 = 0; w2 < nwaves; ++w2) s += tile[w2 * 128 + t];
            atomicAdd(rep_ptr(d.colsum, d.colsum_rc, (unsigned)(m0 >> 4) + (unsigned)(coff / d.ldc)) + cbase + n0 + t, s);
        }
    }
}


}  // namespace nbci
