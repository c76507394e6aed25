// gemm_pc.hip — bf16 GEMM with PRODUCER and CONSUMER waves: 144 x 256 output tile per workgroup, one workgroup per CU.
//
// Why: the 2-stage kernel of gemm_glds.hip runs two 144 x 128 workgroups per CU. Per K tile of 64 a
// CU then stages 2 x (18 + 16) KB = 68 KB through LDS-DMA for 4.7 MFLOP (69 FLOP per staged byte, ~72 GB/s per CU in the K
// loop, which is where L2 -> LDS transfers of this kind level off), its 8 waves read 176 KB of fragments (144 x 32 wave tiles)
// and every wave spends ~450 of its ~2100 cycles per K tile ISSUING its 8-9 LDS-DMA pieces, during which it issues no MFMA.
// Here:
//   * ONE 144 x 256 tile per CU: 18 + 32 = 50 KB staged per K tile for the same 4.7 MFLOP (94 FLOP per staged byte);
//   * 4 consumer waves, one per SIMD, each a 144 x 64 wave tile (9 x 4 MFMA blocks, 144 accumulator registers): 13 fragment
//     reads per 36 MFMAs instead of 11 per 18 -> 104 KB of fragment reads per K tile instead of 176 KB;
//   * 4 producer waves (the SIMD partners of the consumers) issue ALL LDS-DMA; consumers never touch global memory in the loop;
//   * 3 LDS stages of 50 KB; ONE s_barrier per K tile, placed in the MIDDLE of the consumers' tile so that the fragments of the
//     next tile's first k-step are prefetched behind the MFMAs of this tile's second k-step (no restart bubble per K tile);
//   * the consumers' fragment reads are a rolling software pipeline: B fragments double-buffered, each A fragment re-read for
//     the next k-step as soon as its row of MFMAs has been issued; every wait is a COUNTED lgkmcnt (12 reads stay in flight).
// Layouts: A k-major (token-major activations), B k-major (forward, x W^T) or row-major-in-k (data gradient, dy W); K % 64 == 0,
// no split-K, no window views (those stay on gemm_glds.hip). LDS images and swizzles are the ones of gemm_glds.h; the 256-column
// B tile is two 128-column images side by side.
//
// Synchronisation protocol (tile k lives in stage k % 3; B_k = the k-th workgroup barrier):
//   producers : issue(0), issue(1); vmcnt -> tile 0 landed; B_0; for k = 1 .. nt-1 { vmcnt(0): tile k landed; B_k; issue(k+1) }
//   consumers : B_0; read step (0,0); for i = 0 .. nt-1 { step (i,0) prefetching (i,1); B_{i+1}; step (i,1) prefetching (i+1,0) }
// issue(k+1) goes to stage (k-2) % 3, whose last fragment reads were consumed by MFMAs issued before the consumers reached B_k.
// Every wave executes exactly nt barriers in the loop (B_0 .. B_{nt-1}).
#include <cstdlib>

#include "gemm_glds.h"

namespace nbci {

constexpr int PC_BM = 144, PC_BN = 256, PC_MI = 9, PC_NI = 4;
constexpr int PC_A_BYTES = PC_BM * 128, PC_B_BYTES = 32768, PC_STAGE = PC_A_BYTES + PC_B_BYTES, PC_NS = 3;
constexpr int PC_LDS = PC_NS * PC_STAGE;   // 153,600 B (the epilogue's two f32 images [144][132] x 2 = 152,064 B fit inside)
constexpr int PC_THREADS = 512;

// rows R .. 8 of one k-step (compile-time recursion: the re-read's row offset is an instruction immediate)
template <bool BKM, int R>
__device__ __forceinline__ void pc_rows(f32x4 (&acc)[PC_MI][PC_NI], bf16x8 (&af)[PC_MI], const bf16x8 (&bc)[PC_NI], unsigned a_next) {
    if constexpr (R < PC_MI) {
        // Outstanding reads when row R waits: A'(R+1..8) of the current set (8 - R), the 4 (k-major) or 8 (transposed) reads of
        // the next B set, and the R re-reads issued in this step: 12 resp. 16 (the counter saturates at 15: one read early).
        if constexpr (BKM) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);   // MFMAs do not touch memory: without the fence hipcc may hoist them above the wait
#pragma unroll
        for (int ni = 0; ni < PC_NI; ++ni)
            acc[R][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bc[ni], af[R], acc[R][ni], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        af[R] = ds_read_b128_asm<2048 * R>(a_next);   // the same row block for the NEXT k-step (the MFMAs above read af[R] at issue)
        __builtin_amdgcn_sched_barrier(0);
        pc_rows<BKM, R + 1>(acc, af, bc, a_next);
    }
}

// one k-step of a consumer wave: 36 MFMAs on (bc, af) while the fragments of the NEXT k-step are read into (bn, af)
template <bool BKM, int KS_NEXT>
__device__ __forceinline__ void pc_step(f32x4 (&acc)[PC_MI][PC_NI], bf16x8 (&af)[PC_MI], const bf16x8 (&bc)[PC_NI], bf16x8 (&bn)[PC_NI],
                                        unsigned a_next, unsigned b_next, const unsigned (&tb_next)[PC_NI]) {
    if constexpr (BKM) km_read_frags<PC_NI>(bn, b_next);
    else tr_read_frags<KS_NEXT, PC_NI>(bn, tb_next);
    __builtin_amdgcn_sched_barrier(0);
    pc_rows<BKM, 0>(acc, af, bc, a_next);
}

template <bool BKM>
__global__ __launch_bounds__(PC_THREADS) void gemm_pc_kernel(GemmK d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);

    // XCD-aware bijective remap + grouped raster (as gemm_glds.hip): blocks b, b + 8, ... share an XCD
    const int nwg = d.tiles_m * d.tiles_n;
    int wg;
    {
        const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    int tm, tn;
    {
        const int per_group = 8 * d.tiles_n;
        const int grp = wg / per_group, in_grp = wg % per_group;
        const int first_m = grp * 8;
        const int gsize = min(8, d.tiles_m - first_m);
        tm = first_m + in_grp % gsize;
        tn = in_grp / gsize;
    }
    const int m0 = tm * PC_BM, n0 = tn * PC_BN;
    const int z = blockIdx.y;
    const int z1 = z / d.zdiv, z2 = z % d.zdiv;
    const long long coff = z1 * d.czs1 + z2 * d.czs2;
    const int nt = d.K / 64;

    if (w >= 4) {
        // ------------------------------------------------------------------------------------------------ producers
        const int pw = w - 4;
        OperandK A = d.A, B = d.B;
        A.ptr = (const bf16_t*)A.ptr + z1 * d.azs1 + z2 * d.azs2;
        B.ptr = (const bf16_t*)B.ptr + z1 * d.bzs1 + z2 * d.bzs2;
        GldsOperand<true, 18, 4> ga;
        GldsOperand<BKM, 16, 4> gb0, gb1;
        glds_setup<true, 18, 4>(ga, A, m0, d.M, pw, lane);
        glds_setup<BKM, 16, 4>(gb0, B, n0, d.N, pw, lane);
        glds_setup<BKM, 16, 4>(gb1, B, n0 + 128, d.N, pw, lane);
        const int lw = (pw < 2 ? 5 : 4) + 8;   // LDS-DMA instructions this wave issues per tile
        auto issue = [&](int k, int stage) {
            char* s = smem + stage * PC_STAGE;
            glds_stage<true, 18, 4>(ga, A, s, k, pw);
            glds_stage<BKM, 16, 4>(gb0, B, s + PC_A_BYTES, k, pw);
            glds_stage<BKM, 16, 4>(gb1, B, s + PC_A_BYTES + 16384, k, pw);
        };
        issue(0, 0);
        if (nt > 1) issue(1, 1);
        wait_vmcnt(nt > 1 ? lw : 0);
        __builtin_amdgcn_s_barrier();   // B_0
        asm volatile("" ::: "memory");
        int stage = 2;
        for (int k = 1; k < nt; ++k) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile k (the only one in flight) has landed
            __builtin_amdgcn_s_barrier();                      // B_k: tile k visible; stage (k+1) % 3 is no longer read
            asm volatile("" ::: "memory");
            if (k + 1 < nt) issue(k + 1, stage);
            stage = stage == 2 ? 0 : stage + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        // ------------------------------------------------------------------------------------------------ consumers
        f32x4 acc[PC_MI][PC_NI];
#pragma unroll
        for (int i = 0; i < PC_MI; ++i)
#pragma unroll
            for (int j = 0; j < PC_NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int i16 = lane & 15, g = lane >> 4;
        const int sub = w >> 1, cb = (w & 1) * 64;   // this wave's 64 columns: 128-column image `sub`, columns cb .. cb + 63 of it
        const char* sB = smem + PC_A_BYTES + sub * 16384;
        const unsigned a_base = km_frag_base(smem, 0, i16, g);   // stage 0, k-step 0 (k-step 1: ^ 64; row block r: + 2048 r)
        unsigned b_base = 0, tb[PC_NI] = {0u, 0u, 0u, 0u};
        if constexpr (BKM) b_base = km_frag_base(sB, cb, i16, g);
        else {
#pragma unroll
            for (int ni = 0; ni < PC_NI; ++ni) tb[ni] = tr_frag_base(sB, cb + ni * 16, i16, g);
        }
        bf16x8 af[PC_MI], bf0[PC_NI], bf1[PC_NI];
        __builtin_amdgcn_s_barrier();   // B_0: tile 0 is in stage 0
        asm volatile("" ::: "memory");
        // fragments of step (0, 0)
        if constexpr (BKM) km_read_frags<PC_NI>(bf0, b_base); else tr_read_frags<0, PC_NI>(bf0, tb);
        km_read_frags<PC_MI>(af, a_base);
        __builtin_amdgcn_sched_barrier(0);
        unsigned so = 0;   // byte offset of the current tile's stage
        for (int i = 0; i < nt; ++i) {
            // step (i, 0): prefetch (i, 1) from the SAME stage
            {
                unsigned tbn[PC_NI];
#pragma unroll
                for (int ni = 0; ni < PC_NI; ++ni) tbn[ni] = tb[ni] + so;
                pc_step<BKM, 1>(acc, af, bf0, bf1, (a_base + so) ^ 64u, (b_base + so) ^ 64u, tbn);
            }
            if (i + 1 < nt) {
                __builtin_amdgcn_s_barrier();   // B_{i+1}: tile i + 1 has landed in the next stage
                asm volatile("" ::: "memory");
            }
            so = so == 2u * PC_STAGE ? 0u : so + PC_STAGE;
            // step (i, 1): prefetch (i + 1, 0) from the NEXT stage (after the last tile: a harmless read of stale LDS bytes)
            {
                unsigned tbn[PC_NI];
#pragma unroll
                for (int ni = 0; ni < PC_NI; ++ni) tbn[ni] = tb[ni] + so;
                pc_step<BKM, 0>(acc, af, bf1, bf0, a_base + so, b_base + so, tbn);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the over-read of the last step has returned before LDS is reused
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();   // E: every consumer has finished reading the stages (producers are waiting here too)
        asm volatile("" ::: "memory");
        float* tile = (float*)smem + sub * (PC_BM * EPI_LD);
#pragma unroll
        for (int mi = 0; mi < PC_MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < PC_NI; ++ni)
                *(float4*)(tile + (mi * 16 + i16) * EPI_LD + cb + ni * 16 + 4 * g) =
                    make_float4(acc[mi][ni][0] * d.alpha, acc[mi][ni][1] * d.alpha, acc[mi][ni][2] * d.alpha, acc[mi][ni][3] * d.alpha);
    }
    if (w >= 4) {
        __builtin_amdgcn_s_barrier();   // E (producers' side)
        asm volatile("" ::: "memory");
    }
    __syncthreads();   // the two f32 images are complete
    // row-contiguous epilogue by all 8 waves: image h = columns n0 + 128 h .. + 127
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        float csum[4] = {0.f, 0.f, 0.f, 0.f};
        const int nh = n0 + 128 * h;
        epi_tile_rows(d, (const float*)smem + h * (PC_BM * EPI_LD), PC_BM, m0, nh, coff, t, PC_THREADS, csum);
        if (d.colsum) {   // bias gradient: the two half-waves hold the same columns; one atomic per column per wave
            const int cbase = (int)(coff % d.ldc), n = nh + 4 * (t & 31);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s = csum[e];
                s += __shfl_xor(s, 32, 64);
                if (lane < 32 && n + e < d.N)
                    atomicAdd(rep_ptr(d.colsum, d.colsum_rc, (unsigned)(m0 >> 4) + (unsigned)(t >> 6) + (unsigned)(coff / d.ldc)) + cbase + n + e, s);
            }
        }
    }
}

// ---- host ------------------------------------------------------------------------------------------------------------
bool glds_view(const nbci_gemm_desc& d);

// 0 = never, 1 = where the tile cost model of gemm_glds_launch prefers it (default), 2 = whenever eligible (tests / measurement).
// Initialised from NBCI_GEMM_PC; nbci_debug_gemm_pc() changes it at run time.
static int g_pc_mode = -1;
int gemm_pc_mode() {
    if (g_pc_mode < 0) { const char* e = getenv("NBCI_GEMM_PC"); g_pc_mode = e ? atoi(e) : 0; }
    return g_pc_mode;
}
void gemm_pc_set_mode(int m) { g_pc_mode = m; }

// the shapes this kernel takes: k-major A, whole K tiles, no split-K, no views; the caller has checked glds eligibility
bool gemm_pc_eligible(const nbci_gemm_desc& d, const GemmK& k) {
    return d.A.kmajor && d.K % 64 == 0 && d.K >= 128 && k.splitk == 1 && !glds_view(d) && d.N >= 256;
}

// workgroup rounds on the chip x work per workgroup, for the tile cost model of gemm_glds_launch (one workgroup per CU)
long gemm_pc_tiles(const nbci_gemm_desc& d) {
    const int batch = d.batch > 0 ? d.batch : 1;
    return (long)((d.M + PC_BM - 1) / PC_BM) * ((d.N + PC_BN - 1) / PC_BN) * batch;
}

int gemm_pc_launch(const nbci_gemm_desc& d, GemmK k, hipStream_t stream) {
    const int batch = d.batch > 0 ? d.batch : 1;
    k.tiles_m = (d.M + PC_BM - 1) / PC_BM;
    k.tiles_n = (d.N + PC_BN - 1) / PC_BN;
    static bool attr = false;
    if (!attr) {
        hipError_t e1 = hipFuncSetAttribute((const void*)gemm_pc_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS);
        hipError_t e2 = hipFuncSetAttribute((const void*)gemm_pc_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS);
        if (e1 != hipSuccess || e2 != hipSuccess) return fail(NBCI_EHIP, "gemm_pc: LDS attribute");
        attr = true;
    }
    dim3 grid(k.tiles_m * k.tiles_n, batch);
    if (d.B.kmajor) hipLaunchKernelGGL((gemm_pc_kernel<true>), grid, dim3(PC_THREADS), PC_LDS, stream, k);
    else hipLaunchKernelGGL((gemm_pc_kernel<false>), grid, dim3(PC_THREADS), PC_LDS, stream, k);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NBCI_EHIP, std::string("gemm_pc launch: ") + hipGetErrorString(e));
    return NBCI_OK;
}

}  // namespace nbci
