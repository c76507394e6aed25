// gemm.hip — the one MFMA GEMM of the NDT1 hot path (gfx950, wave64).
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
    const float* residual; long long ldr;
    const long long* residual_rows;  // optional gather: residual row for output row m
    int residual_first;              // add residual before act/dropout (embed: proj + pos, then dropout)
    const void* gate; long long ldg; int gate_act;  // v *= act'(gate[m][n]) (gate in the input dtype)
    int gate_bf16;
    int cvec;  // vector C/residual accesses legal
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

template <typename T, bool AK, bool BKM>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_kernel(GemmK d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BK = GemmTile<T>::BK;
    constexpr int REGION = GemmTile<T>::REGION;
    constexpr int NCH = GemmTile<T>::NCH;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int w = t >> 6;
    const int wm = w >> 1, wn = w & 1;

    // XCD-aware bijective remap: blocks b, b+8, ... share an XCD (round-robin dispatch), so give
    // every XCD a contiguous run of tiles; consecutive tiles share the A row panel in that L2.
    const int nwg = d.tiles_m * d.tiles_n;
    int wg;
    {
        const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int tm = wg / d.tiles_n, tn = wg % d.tiles_n;
    const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;

    const int z = blockIdx.y;
    int kt_begin = 0, kt_end = (d.K + BK - 1) / BK;
    OperandK A = d.A, B = d.B;
    long long coff = 0;
    if (d.splitk > 1) {
        kt_begin = z * d.tiles_per_split;
        kt_end = min(kt_end, kt_begin + d.tiles_per_split);
        if (kt_begin >= kt_end) return;
    } else {
        const int z1 = z / d.zdiv, z2 = z % d.zdiv;
        A.ptr = (const T*)A.ptr + z1 * d.azs1 + z2 * d.azs2;
        B.ptr = (const T*)B.ptr + z1 * d.bzs1 + z2 * d.bzs2;
        coff = z1 * d.czs1 + z2 * d.czs2;
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    uint4 ra[NCH], rb[NCH];
    load_chunks<T, AK>(A, m0, d.M, kt_begin * BK, d.K, ra, t);
    load_chunks<T, BKM>(B, n0, d.N, kt_begin * BK, d.K, rb, t);
    store_chunks<T, AK>(smem, ra, t);
    store_chunks<T, BKM>(smem + REGION, rb, t);
    __syncthreads();

    int cur = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const bool more = (kt + 1 < kt_end);
        if (more) {
            load_chunks<T, AK>(A, m0, d.M, (kt + 1) * BK, d.K, ra, t);
            load_chunks<T, BKM>(B, n0, d.N, (kt + 1) * BK, d.K, rb, t);
        }
        const char* sA = smem + cur * 2 * REGION;
        compute_tile<T, AK, BKM>(sA, sA + REGION, acc, wm, wn, lane);
        if (more) {
            char* nA = smem + (cur ^ 1) * 2 * REGION;
            store_chunks<T, AK>(nA, ra, t);
            store_chunks<T, BKM>(nA + REGION, rb, t);
        }
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: lane owns m = .. + (lane&15), n = .. + 4*(lane>>4) + 0..3 ---------------
    const int i16 = lane & 15, g = lane >> 4;
    if (d.splitk > 1) {
        // Split-K partials go to C with f32 atomics. A wave-instruction of float atomics runs at full
        // rate only when it covers 256 contiguous bytes, so each wave first transposes its 64x64
        // accumulator through LDS (XOR-swizzled float4 columns: conflict-free both ways) and then
        // issues one atomic per ROW: 64 lanes = 64 consecutive n.
        float* sw = (float*)smem + w * 4096;  // 16 KB per wave; main-loop LDS is dead after the last barrier
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int ml = mi * 16 + i16, c4 = ni * 4 + g;
                *(float4*)(sw + ml * 64 + ((c4 ^ (ml & 15)) << 2)) =
                    make_float4(acc[mi][ni][0] * d.alpha, acc[mi][ni][1] * d.alpha, acc[mi][ni][2] * d.alpha, acc[mi][ni][3] * d.alpha);
            }
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS writes have landed (wave-private region)
        const int nn = n0 + wn * 64 + lane;
        float* cbase = (float*)d.C + nn;
        for (int ml = 0; ml < 64; ++ml) {
            const int m = m0 + wm * 64 + ml;
            if (m >= d.M) break;
            const float val = sw[ml * 64 + ((((lane >> 2) ^ (ml & 15)) << 2) | (lane & 3))];
            if (nn < d.N) atomicAdd(cbase + (long long)m * d.ldc, val);
        }
        return;
    }
    float csum[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) csum[a][b] = 0.f;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wm * 64 + mi * 16 + i16;
        if (m >= d.M) continue;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + wn * 64 + ni * 16 + 4 * g;
            if (n >= d.N) continue;
            float v[4] = {acc[mi][ni][0] * d.alpha, acc[mi][ni][1] * d.alpha,
                          acc[mi][ni][2] * d.alpha, acc[mi][ni][3] * d.alpha};
            const long long cidx = coff + (long long)m * d.ldc + n;
            const bool full = (n + 3 < d.N) && d.cvec;
            if (d.bias) {
                if (full) {
                    const float4 b4 = *(const float4*)(d.bias + n);
                    v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (n + e < d.N) v[e] += d.bias[n + e];
                }
            }
            if (d.C2) {
                if (d.c_bf16) {
                    bf16_t* c2 = (bf16_t*)d.C2 + cidx;
                    if (full) { bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])}; *(bf16x4*)c2 = o; }
                    else { for (int e = 0; e < 4; ++e) if (n + e < d.N) c2[e] = f2bf(v[e]); }
                } else {
                    float* c2 = (float*)d.C2 + cidx;
                    if (full) *(float4*)c2 = make_float4(v[0], v[1], v[2], v[3]);
                    else { for (int e = 0; e < 4; ++e) if (n + e < d.N) c2[e] = v[e]; }
                }
            }
            if (d.residual && d.residual_first) {
                const long long rr = d.residual_rows ? d.residual_rows[m] : (long long)m;
                const float* r = d.residual + rr * d.ldr + n;
                if (full) { const float4 r4 = *(const float4*)r; v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w; }
                else { for (int e = 0; e < 4; ++e) if (n + e < d.N) v[e] += r[e]; }
            }
            if (d.act != ACT_NONE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = act_fwd(d.act, v[e]);
            }
            if (d.gate) {
                const long long gi = (long long)m * d.ldg + n;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (n + e < d.N) {
                        const float gv = d.gate_bf16 ? bf2f(((const bf16_t*)d.gate)[gi + e]) : ((const float*)d.gate)[gi + e];
                        v[e] *= act_bwd(d.gate_act, gv);
                    }
                }
            }
            if (d.drop_thr) {
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
            if (d.residual && !d.residual_first) {
                const long long rr = d.residual_rows ? d.residual_rows[m] : (long long)m;
                const float* r = d.residual + rr * d.ldr + n;
                if (full) { const float4 r4 = *(const float4*)r; v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w; }
                else { for (int e = 0; e < 4; ++e) if (n + e < d.N) v[e] += r[e]; }
            }
            if (d.colsum) {
#pragma unroll
                for (int e = 0; e < 4; ++e) csum[ni][e] += v[e];
            }
            if (d.c_bf16) {
                bf16_t* c = (bf16_t*)d.C + cidx;
                if (full) { bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])}; *(bf16x4*)c = o; }
                else { for (int e = 0; e < 4; ++e) if (n + e < d.N) c[e] = f2bf(v[e]); }
            } else {
                float* c = (float*)d.C + cidx;
                if (d.beta != 0.f) {
                    if (full) { const float4 o = *(const float4*)c; v[0] += d.beta * o.x; v[1] += d.beta * o.y; v[2] += d.beta * o.z; v[3] += d.beta * o.w; }
                    else { for (int e = 0; e < 4; ++e) if (n + e < d.N) v[e] += d.beta * c[e]; }
                }
                if (full) *(float4*)c = make_float4(v[0], v[1], v[2], v[3]);
                else { for (int e = 0; e < 4; ++e) if (n + e < d.N) c[e] = v[e]; }
            }
        }
    }
    if (d.colsum) {
        // bias gradient fused into the producing GEMM: reduce this wave's 64 rows (16 lanes x 4 mi)
        // with shuffles, then one atomic per column per wave
        const int cbase = (int)(coff % d.ldc);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = csum[ni][e];
                t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 8, 64);
                const int n = n0 + wn * 64 + ni * 16 + 4 * g + e;
                if (i16 == 0 && n < d.N) atomicAdd(d.colsum + cbase + n, t);
            }
    }
}

// ---- host side ---------------------------------------------------------------------------
static bool operand_vec_ok(const nbci_operand& o, int E, size_t esz) {
    if (((uintptr_t)o.ptr) % 16) return false;
    if (o.ld % E) return false;
    if (o.rpb > 0 && (o.gstride % E)) return false;
    if ((o.zs1 % E) || (o.zs2 % E)) return false;
    (void)esz;
    return true;
}

template <typename T, bool AK, bool BKM>
static int launch_inst(const GemmK& k, dim3 grid, hipStream_t stream) {
    // split-K transposes the 4 x 16 KB accumulator tiles through LDS in its epilogue
    const int lds = (k.splitk > 1 && 4 * GemmTile<T>::REGION < 65536) ? 65536 : 4 * GemmTile<T>::REGION;
    hipLaunchKernelGGL((gemm_kernel<T, AK, BKM>), grid, dim3(GEMM_THREADS), lds, stream, k);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NBCI_EHIP, std::string("gemm launch: ") + hipGetErrorString(e));
    return NBCI_OK;
}

template <typename T>
static int launch_layout(const GemmK& k, bool ak, bool bk, dim3 grid, hipStream_t s) {
    if (ak && bk) return launch_inst<T, true, true>(k, grid, s);
    if (ak && !bk) return launch_inst<T, true, false>(k, grid, s);
    if (!ak && bk) return launch_inst<T, false, true>(k, grid, s);
    return launch_inst<T, false, false>(k, grid, s);
}

int gemm_launch(const nbci_gemm_desc& d, hipStream_t stream) {
    NBCI_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0, NBCI_ESHAPE, "gemm: M,N,K must be positive");
    NBCI_REQUIRE(d.A.ptr && d.B.ptr && d.C, NBCI_EINVAL, "gemm: null operand");
    NBCI_REQUIRE(d.in_dtype == NBCI_F32 || d.in_dtype == NBCI_BF16, NBCI_EINVAL, "gemm: bad in_dtype");
    NBCI_REQUIRE(d.c_dtype == NBCI_F32 || d.c_dtype == NBCI_BF16, NBCI_EINVAL, "gemm: bad c_dtype");
    const int batch = d.batch > 0 ? d.batch : 1;
    const int splitk = d.splitk > 1 ? d.splitk : 1;
    NBCI_REQUIRE(!(splitk > 1 && batch > 1), NBCI_EINVAL, "gemm: splitk and batch are exclusive");
    NBCI_REQUIRE(!(splitk > 1 && d.c_dtype != NBCI_F32), NBCI_EINVAL, "gemm: splitk needs f32 C");
    NBCI_REQUIRE(!(splitk > 1 && (d.bias || d.act || d.residual || d.C2 || d.gate || d.colsum || d.drop_p > 0.f)), NBCI_EINVAL,
                 "gemm: splitk supports alpha only");
    NBCI_REQUIRE(!(d.beta != 0.f && d.c_dtype != NBCI_F32), NBCI_EINVAL, "gemm: beta needs f32 C");
    NBCI_REQUIRE(d.drop_p >= 0.f && d.drop_p < 1.f, NBCI_EINVAL, "gemm: drop_p out of [0,1)");
    const size_t csz = d.c_dtype == NBCI_BF16 ? 2 : 4;
    NBCI_REQUIRE(((uintptr_t)d.C) % csz == 0, NBCI_EALIGN, "gemm: C misaligned");

    GemmK k;
    k.M = d.M; k.N = d.N; k.K = d.K;
    const int E = d.in_dtype == NBCI_BF16 ? 8 : 4;
    const int BK = d.in_dtype == NBCI_BF16 ? 64 : 16;
    k.A = {d.A.ptr, d.A.ld, d.A.rpb, d.A.gstride, operand_vec_ok(d.A, E, 0) ? 1 : 0};
    k.B = {d.B.ptr, d.B.ld, d.B.rpb, d.B.gstride, operand_vec_ok(d.B, E, 0) ? 1 : 0};
    k.azs1 = d.A.zs1; k.azs2 = d.A.zs2; k.bzs1 = d.B.zs1; k.bzs2 = d.B.zs2;
    k.C = d.C; k.C2 = d.C2; k.ldc = d.ldc; k.czs1 = d.czs1; k.czs2 = d.czs2;
    k.c_bf16 = d.c_dtype == NBCI_BF16;
    k.zdiv = d.zdiv > 0 ? d.zdiv : 1;
    k.splitk = splitk;
    const int ktiles = (d.K + BK - 1) / BK;
    k.tiles_per_split = (ktiles + splitk - 1) / splitk;
    k.tiles_m = (d.M + GEMM_BM - 1) / GEMM_BM;
    k.tiles_n = (d.N + GEMM_BN - 1) / GEMM_BN;
    k.alpha = d.alpha; k.beta = d.beta;
    k.bias = d.bias; k.act = d.act;
    k.drop_thr = drop_threshold(d.drop_p);
    k.drop_scale = d.drop_p > 0.f ? 1.0f / (1.0f - d.drop_p) : 1.0f;
    k.drop_key = drop_key(d.seed, d.site);
    k.colsum = d.colsum;
    k.residual = d.residual; k.ldr = d.ldr;
    k.residual_rows = (const long long*)d.residual_rows; k.residual_first = d.residual_first;
    k.gate = d.gate; k.ldg = d.ldg; k.gate_act = d.gate_act; k.gate_bf16 = d.in_dtype == NBCI_BF16;
    // vector epilogue: 4 consecutive n at 16-byte (f32) / 8-byte (bf16) alignment
    bool cvec = (d.ldc % 4 == 0) && (d.czs1 % 4 == 0) && (d.czs2 % 4 == 0) && (((uintptr_t)d.C) % 16 == 0);
    if (d.C2) cvec = cvec && (((uintptr_t)d.C2) % 16 == 0);
    if (d.bias) cvec = cvec && (((uintptr_t)d.bias) % 16 == 0);
    if (d.residual) cvec = cvec && (d.ldr % 4 == 0) && (((uintptr_t)d.residual) % 16 == 0);
    k.cvec = cvec ? 1 : 0;

    dim3 grid(k.tiles_m * k.tiles_n, splitk > 1 ? splitk : batch);
    if (d.in_dtype == NBCI_BF16) return launch_layout<bf16_t>(k, d.A.kmajor != 0, d.B.kmajor != 0, grid, stream);
    return launch_layout<float>(k, d.A.kmajor != 0, d.B.kmajor != 0, grid, stream);
}

}  // namespace nbci

// ---- optional per-launch timing (bench.py roofline leg): HIP events around every GEMM launch ----
#include <vector>
namespace nbci {
struct ProfRec { hipEvent_t a, b; double flops; int kind; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;

void gemm_profile_enable(bool on) { g_prof_on = on; }
bool gemm_profile_on() { return g_prof_on; }

int gemm_launch_timed(const nbci_gemm_desc& d, hipStream_t stream) {
    if (!g_prof_on) return gemm_launch(d, stream);
    ProfRec r;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return fail(NBCI_EHIP, "event create");
    const int batch = d.batch > 0 ? d.batch : 1;
    r.flops = 2.0 * d.M * (double)d.N * d.K * batch;
    r.kind = (d.in_dtype == NBCI_BF16 ? 4 : 0) + (d.A.kmajor ? 2 : 0) + (d.B.kmajor ? 1 : 0);
    (void)hipEventRecord(r.a, stream);
    const int rc = gemm_launch(d, stream);
    (void)hipEventRecord(r.b, stream);
    g_prof.push_back(r);
    return rc;
}

// out[kind] = {total ms, total flops, launches} for kind = dtype*4 + A.kmajor*2 + B.kmajor (8 kinds)
int gemm_profile_collect(double* out24) {
    for (int i = 0; i < 24; ++i) out24[i] = 0.0;
    for (auto& r : g_prof) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess)
            return fail(NBCI_EHIP, "profile collect");
        out24[r.kind * 3 + 0] += ms; out24[r.kind * 3 + 1] += r.flops; out24[r.kind * 3 + 2] += 1.0;
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    }
    g_prof.clear();
    return NBCI_OK;
}
}  // namespace nbci
