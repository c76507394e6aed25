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
// Synchronisation protocol (tile k lives in stage k % 3; B_k = the k-th workgroup barrier; FREE_c = an LDS word per consumer wave):
//   producers : issue(0), issue(1); for k = 0 .. nt-1 { vmcnt(pieces of tile k+1): tile k landed; B_k; poll FREE >= k; issue(k+2) }
//   consumers : B_0; read step (0,0); for i = 0 .. nt-1 { step (i,0) prefetching (i,1); FREE_c = i + 1; B_{i+1};
//                                                          step (i,1) prefetching (i+1,0) }
// TWO tiles are in flight (k+1 landing while k+2 is issued): with one, a K tile cost issue (740 cycles: 50 KB through the CU's
// 64 B/clk L2 -> LDS path) + landing (880) + barrier = 1 770 cycles against 1 152 cycles of MFMA work (in-kernel stamps,
// profiles/r02_gemm_pc_stamps_*.txt). issue(k+2) overwrites stage (k-1) % 3: at B_k every consumer has ISSUED its last read of
// tile k-1 (they are the re-reads of step (k-1,0)) and then written FREE_c = k through the same in-order LDS queue, so a producer
// that has read FREE_c >= k from LDS knows those reads have executed; the DMA it issues afterwards cannot overtake them.
// Every wave executes exactly nt barriers in the loop (B_0 .. B_{nt-1}).
#include <cstdlib>

#include "gemm_glds.h"

namespace nbci {

constexpr int PC_BM = 144, PC_BN = 256, PC_MI = 9, PC_NI = 4;
constexpr int PC_A_BYTES = PC_BM * 128, PC_B_BYTES = 32768, PC_STAGE = PC_A_BYTES + PC_B_BYTES, PC_NS = 3;
constexpr int PC_FLAGS = PC_NS * PC_STAGE;   // FREE words: 4 consumers x 256 B (every lane writes its own word: no write conflict)
constexpr int PC_LDS = PC_FLAGS + 1024;      // 154,624 B (the epilogue's two f32 images [144][132] x 2 = 152,064 B fit below the flags)
constexpr int PC_THREADS = 512;

#ifdef NBCI_STAMPS   // measurement build only (tools/gemm_pc_stamps.py): producer wave 4 of every workgroup, shader cycles per K tile:
                     // [0] past B_k, [1] tile k+1 issued, [2] tile k+1 landed (vmcnt(0)); wall clock (100 MHz): entry, B_0, loop end, stores issued
static __device__ unsigned long long g_pc_tile[1024 * 64 * 4];
static __device__ unsigned long long g_pc_wall[1024 * 8];
#define PC_TSTAMP(k, slot) do { if (pstamp && (k) < 64) g_pc_tile[((size_t)blockIdx.x * 64 + (k)) * 4 + (slot)] = clock64(); } while (0)
#define PC_WSTAMP(slot) do { if (threadIdx.x == 256 && blockIdx.y == 0 && blockIdx.x < 1024) g_pc_wall[blockIdx.x * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define PC_TSTAMP(k, slot) do { } while (0)
#define PC_WSTAMP(slot) do { } while (0)
#endif

// rows R .. 8 of one k-step (compile-time recursion: the re-read's row offset is an instruction immediate)
template <bool BKM, int EXTRA, int R>
__device__ __forceinline__ void pc_rows(f32x4 (&acc)[PC_MI][PC_NI], bf16x8 (&af)[PC_MI], const bf16x8 (&bc)[PC_NI], unsigned a_next) {
    if constexpr (R < PC_MI) {
        // Outstanding LDS operations when row R waits: A'(R+1..8) of the current set (8 - R), EXTRA (the FREE word written
        // between the two steps of a tile), the 4 (k-major) or 8 (transposed) reads of the next B set, and the R re-reads issued
        // in this step: 12 + EXTRA resp. 16 + EXTRA (the counter saturates at 15: the wait is then a read or two early).
        constexpr int N = (BKM ? 12 : 16) + EXTRA;
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N < 15 ? N : 15) : "memory");
        __builtin_amdgcn_sched_barrier(0);   // MFMAs do not touch memory: without the fence hipcc may hoist them above the wait
#pragma unroll
        for (int ni = 0; ni < PC_NI; ++ni)
            acc[R][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bc[ni], af[R], acc[R][ni], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        af[R] = ds_read_b128_asm<2048 * R>(a_next);   // the same row block for the NEXT k-step (the MFMAs above read af[R] at issue)
        __builtin_amdgcn_sched_barrier(0);
        pc_rows<BKM, EXTRA, R + 1>(acc, af, bc, a_next);
    }
}

// one k-step of a consumer wave: 36 MFMAs on (bc, af) while the fragments of the NEXT k-step are read into (bn, af)
template <bool BKM, int KS_NEXT, int EXTRA>
__device__ __forceinline__ void pc_step(f32x4 (&acc)[PC_MI][PC_NI], bf16x8 (&af)[PC_MI], const bf16x8 (&bc)[PC_NI], bf16x8 (&bn)[PC_NI],
                                        unsigned a_next, unsigned b_next, const unsigned (&tb_next)[PC_NI]) {
    if constexpr (BKM) km_read_frags<PC_NI>(bn, b_next);
    else tr_read_frags<KS_NEXT, PC_NI>(bn, tb_next);
    __builtin_amdgcn_sched_barrier(0);
    pc_rows<BKM, EXTRA, 0>(acc, af, bc, a_next);
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

    PC_WSTAMP(0);
    if (w >= 4) {
        // ------------------------------------------------------------------------------------------------ producers
        const int pw = w - 4;
#ifdef NBCI_STAMPS
        const bool pstamp = threadIdx.x == 256 && blockIdx.y == 0 && blockIdx.x < 1024;
#endif
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
        const unsigned flag_addr = lds_addr(smem + PC_FLAGS) + (lane & 3) * 256;   // lane l polls consumer l & 3
        issue(0, 0);
        if (nt > 1) issue(1, 1);
        int stage = 2;
        for (int k = 0; k < nt; ++k) {
            wait_vmcnt(k + 1 < nt ? lw : 0);                   // tile k has landed (tile k + 1 may still be in flight)
            PC_TSTAMP(k, 2);
            __builtin_amdgcn_s_barrier();                      // B_k: tile k visible to the consumers
            asm volatile("" ::: "memory");
            if (k == 0) PC_WSTAMP(1);
            PC_TSTAMP(k, 0);
            if (k + 2 < nt) {
                // stage (k + 2) % 3 held tile k - 1: every consumer must have issued its last read of it (FREE_c >= k)
                for (;;) {
                    unsigned f;
                    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(f) : "v"(flag_addr) : "memory");
                    if (__builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64((int)f >= k) == ~0ull)) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                issue(k + 2, stage);
            }
            PC_TSTAMP(k, 1);
            stage = stage == 2 ? 0 : stage + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PC_WSTAMP(2);
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
        const unsigned flag_addr = lds_addr(smem + PC_FLAGS) + w * 256 + lane * 4;
        // FREE_c = 0 before anybody polls it (LDS keeps the previous workgroup's bytes); completed before B_0 orders it for the producers
        asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(flag_addr), "v"(0) : "memory");
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
                pc_step<BKM, 1, 0>(acc, af, bf0, bf1, (a_base + so) ^ 64u, (b_base + so) ^ 64u, tbn);
            }
            // every read of tile i's stage has been issued: FREE_c = i + 1 through the same in-order LDS queue
            asm volatile("ds_write_b32 %0, %1" ::"v"(flag_addr), "v"(i + 1) : "memory");
            __builtin_amdgcn_sched_barrier(0);
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
                pc_step<BKM, 0, 1>(acc, af, bf1, bf0, a_base + so, b_base + so, tbn);
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
    PC_WSTAMP(3);
    // row-contiguous epilogue by all 8 waves: image h = columns n0 + 128 h .. + 127
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        float csum[4] = {0.f, 0.f, 0.f, 0.f};
        const int nh = n0 + 128 * h;
        epi_tile_rows(d, (const float*)smem + h * (PC_BM * EPI_LD), PC_BM, m0, nh, coff, t, PC_THREADS, csum);
        if (d.colsum) {   // bias gradient: the eight waves' column sums meet in image h (read out by now) and ONE lane-contiguous atomic
                          // per column leaves the workgroup (gemm_common.h, gemm_epilogue_tile)
            const int cbase = (int)(coff % d.ldc);
            float* red = (float*)smem + h * (PC_BM * EPI_LD);
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s = csum[e];
                s += __shfl_xor(s, 32, 64);
                if (lane < 32) red[(t >> 6) * 128 + 4 * (t & 31) + e] = s;
            }
            __syncthreads();
            if (t < 128 && nh + t < d.N) {
                float s = 0.f;
#pragma unroll
                for (int w2 = 0; w2 < PC_THREADS / 64; ++w2) s += red[w2 * 128 + t];
                atomicAdd(rep_ptr(d.colsum, d.colsum_rc, (unsigned)(m0 >> 4) + (unsigned)(coff / d.ldc)) + cbase + nh + t, s);
            }
        }
    }
    PC_WSTAMP(4);
}

// ---- host ------------------------------------------------------------------------------------------------------------
bool glds_view(const nbci_gemm_desc& d);

// 0 = never, 1 = where the tile cost model of gemm_glds_launch prefers it (default), 2 = whenever eligible (tests / measurement).
// Initialised from NBCI_GEMM_PC; nbci_debug_gemm_pc() changes it at run time.
static int g_pc_mode = -1;
int gemm_pc_mode() {
    if (g_pc_mode < 0) g_pc_mode = measure_env("NBCI_GEMM_PC", 1);
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
    TRY_(ensure_dyn_lds(d.B.kmajor ? (const void*)gemm_pc_kernel<true> : (const void*)gemm_pc_kernel<false>, PC_LDS, "gemm_pc"));
    dim3 grid(k.tiles_m * k.tiles_n, batch);
    if (prof_on()) prof_note_symbol(d.B.kmajor ? "gemm_pc_kernel<true>" : "gemm_pc_kernel<false>");
    if (d.B.kmajor) hipLaunchKernelGGL((gemm_pc_kernel<true>), grid, dim3(PC_THREADS), PC_LDS, stream, k);
    else hipLaunchKernelGGL((gemm_pc_kernel<false>), grid, dim3(PC_THREADS), PC_LDS, stream, k);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NBCI_EHIP, std::string("gemm_pc launch: ") + hipGetErrorString(e));
    return NBCI_OK;
}

}  // namespace nbci

#ifdef NBCI_STAMPS
extern "C" int nbci_debug_read_pc_stamps(unsigned long long* tile, unsigned long long* wall, int nblocks) {
    if (hipMemcpyFromSymbol(tile, HIP_SYMBOL(nbci::g_pc_tile), (size_t)nblocks * 64 * 4 * sizeof(unsigned long long)) != hipSuccess) return -1;
    return (int)hipMemcpyFromSymbol(wall, HIP_SYMBOL(nbci::g_pc_wall), (size_t)nblocks * 8 * sizeof(unsigned long long));
}
#endif
