// gemm_glds.hip — bf16 fast-path GEMM: operands go HBM -> LDS directly (global_load_lds, 16 B per
// lane, no VGPR staging), two LDS stages, one barrier per K tile.
//
// LDS image = the one gemm_kernel builds (so the fragment readers are shared): a wave-instruction
// of global_load_lds writes 1 KiB of LDS linearly (base + lane*16), therefore the XOR swizzles of
// the image are applied to each lane's SOURCE address instead:
//   k-major tile  [rows][64 k]   128-B rows : piece = 8 rows ; lane -> row l>>3, physical 16-B
//                                             chunk l&7 holds logical chunk (l&7) ^ ((row>>1)&7)
//   row-major-in-k [64 k][128]   256-B rows : piece = 4 k-rows ; lane -> k-row l>>4, physical chunk
//                                             l&15 holds logical (((l&15)>>1) ^ f(k))<<1 | (l&1)
// Rows past the operand's extent are CLAMPED to a valid row instead of zero-filled: they only feed
// accumulator rows/cols that the epilogue never stores. Only a partial last K tile needs real
// zeros; it goes through the masked register-staged loader of gemm_common.h.
//
// Tile height is a template parameter (BM = 16*MI*WM): for the token-major GEMMs of the train step
// (M = B*T' = 9152) BM = 144 gives 64 x 8 = 512 tiles = exactly two per CU, where 128 x 128 tiles
// need a second, 12 %-full round.
#include <cstdlib>

#include <atomic>
#include <map>
#include <mutex>
#include "gemm_glds.h"

namespace nbci {


template <bool AK, bool BKM, int WM, int WN, int MI, int NI, bool VIEW = false>
__device__ __forceinline__ void gemm_glds_body(const GemmK& d, const int block_x, const int block_y, char* smem) {
    constexpr int BM = WM * MI * 16, BN = WN * NI * 16;
    static_assert(BN == 128, "B tile is always 128 wide");
    static_assert(AK || BM == 128, "row-major-in-k A needs 256-byte tile rows");
    constexpr int A_BYTES = BM * 128;       // k-major: BM rows x 128 B ; row-major-in-k: 64 x 256 B
    constexpr int B_BYTES = 16384;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int NPA = A_BYTES / 1024, NPB = 16;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = w / WN, wn = w % WN;
    STAMP(0);

    // XCD-aware bijective remap: blocks b, b+8, ... share an XCD (round-robin dispatch), so give every
    // XCD a contiguous run of work items. Split-K launches are 1-D over (split, tile) with the split
    // index slowest: one XCD then owns (most of) one K slice, whose A/B slices fit its 4 MB L2 and are
    // fetched from HBM once instead of once per tile row/column.
    const int nwg = d.tiles_m * d.tiles_n;
    const int nitems = nwg * (d.splitk > 1 ? d.splitk : 1);
    int item;
    {
        const int orig = block_x, xcd = orig & 7, q = nitems >> 3, r = nitems & 7;
        item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int wg = item % nwg;
    const int zsplit = item / nwg;
    // grouped raster: 8 tile-rows x successive tile-columns, so the ~64 tiles an XCD runs at once
    // form an 8 x 8 patch (8 A panels + 8 B panels in its 4 MB L2) instead of 1 x 64
    int tm, tn;
    {
        const int per_group = 8 * d.tiles_n;
        const int grp = wg / per_group, in_grp = wg % per_group;
        const int first_m = grp * 8;
        const int gsize = min(8, d.tiles_m - first_m);
        tm = first_m + in_grp % gsize;
        tn = in_grp / gsize;
    }
    const int m0 = tm * BM, n0 = tn * BN;

    const int z = d.splitk > 1 ? zsplit : block_y;
    const int ktiles = (d.dbg & 2) ? 0 : (d.K + 63) / 64;
    int kt_begin = 0, kt_end = ktiles;
    OperandK A = d.A, B = d.B;
    long long coff = 0;
    if (d.splitk > 1) {
        kt_begin = z * d.tiles_per_split;
        kt_end = min(kt_end, kt_begin + d.tiles_per_split);
        if (kt_begin >= kt_end) return;
    } else {
        const int z1 = z / d.zdiv, z2 = z % d.zdiv;
        A.ptr = (const bf16_t*)A.ptr + z1 * d.azs1 + z2 * d.azs2;
        B.ptr = (const bf16_t*)B.ptr + z1 * d.bzs1 + z2 * d.bzs2;
        coff = z1 * d.czs1 + z2 * d.czs2;
    }
    const int kt_full_end = min(kt_end, d.K / 64);  // tiles that are entirely inside K

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    GldsOperand<AK, NPA> ga;
    GldsOperand<BKM, NPB> gb;
    glds_setup<AK, NPA>(ga, A, m0, d.M, w, lane);
    glds_setup<BKM, NPB>(gb, B, n0, d.N, w, lane);

    if constexpr (VIEW) {
        static_assert(!BKM, "view kernels: B is row-major-in-k (A too in the weight-gradient layout)");
        if constexpr (!AK) glds_view_seek<NPA, 4>(ga, A, kt_begin);
        glds_view_seek<NPB, 4>(gb, B, kt_begin);
    }
    int cur = 0;
    if (kt_begin < kt_full_end) {
        glds_stage<AK, NPA, 4, VIEW>(ga, A, smem, kt_begin, w);
        glds_stage<BKM, NPB, 4, VIEW>(gb, B, smem + A_BYTES, kt_begin, w);
    }
    __syncthreads();  // drains the LDS-DMA (hipcc emits vmcnt(0) ahead of the barrier)
    STAMP(1);
    for (int kt = kt_begin; kt < kt_full_end; ++kt) {
#ifdef NBCI_STAMPS   // K-loop stamps (shader cycles) of wave 0, K tiles 4..11: 0 loop top, 1 LDS-DMA issued, 2 first k-step's MFMAs issued +
                     // second k-step's fragments landed, 3 second k-step's MFMAs issued, 4 past the barrier
        unsigned long long* kst = (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 1024 && kt >= 4 && kt < 12)
                                      ? g_kstamps + ((size_t)blockIdx.x * 8 + (kt - 4)) * 8 : nullptr;
        if (kst) kst[0] = clock64();
#endif
#if defined(NBCI_ABLATE) && NBCI_ABLATE == 2
        if (false) {
#else
        if (kt + 1 < kt_full_end) {
#endif
            char* nx = smem + (cur ^ 1) * STAGE;
            glds_stage<AK, NPA, 4, VIEW>(ga, A, nx, kt + 1, w);
            glds_stage<BKM, NPB, 4, VIEW>(gb, B, nx + A_BYTES, kt + 1, w);
        }
        const char* sA = smem + cur * STAGE;
#ifdef NBCI_STAMPS
        if (kst) kst[1] = clock64();
        compute_tile_g<AK, BKM, MI, NI>(sA, sA + A_BYTES, acc, wm * MI * 16, wn * NI * 16, lane, kst);
        if (kst) kst[3] = clock64();
        __syncthreads();
        if (kst) kst[4] = clock64();
#else
        compute_tile_g<AK, BKM, MI, NI>(sA, sA + A_BYTES, acc, wm * MI * 16, wn * NI * 16, lane);
        __syncthreads();
#endif
        cur ^= 1;
    }
    if constexpr (BM == 128) {
        if (kt_full_end < kt_end) {  // partial last K tile: masked register-staged loads (zero fill)
            uint4 ra[4], rb[4];
            load_chunks<bf16_t, AK>(A, m0, d.M, kt_full_end * 64, d.K, ra, t);
            load_chunks<bf16_t, BKM>(B, n0, d.N, kt_full_end * 64, d.K, rb, t);
            char* sA = smem + cur * STAGE;
            store_chunks<bf16_t, AK>(sA, ra, t);
            store_chunks<bf16_t, BKM>(sA + A_BYTES, rb, t);
            __syncthreads();
            compute_tile_g<AK, BKM, MI, NI>(sA, sA + A_BYTES, acc, wm * MI * 16, wn * NI * 16, lane);
            __syncthreads();
        }
    }
    if constexpr (MI == 4 && NI == 4) {
        if (d.splitk > 1) {   // split-K partials: LDS-transposed atomics
            gemm_epilogue<MI, NI>(d, acc, m0 + wm * MI * 16, n0 + wn * NI * 16, coff, lane, w, smem);
            return;
        }
    }
    constexpr int CH = (WM == 1 && MI == 10) ? 2 : 1;   // 160-row tiles: two 80-row chunks (42 KB) keep two workgroups per CU
    STAMP(2);
    gemm_epilogue_tile<MI, NI, CH>(d, acc, wm * MI * 16, wn * NI * 16, m0, n0, BM, coff, t, GEMM_THREADS, smem);
    STAMP(3);
#ifdef NBCI_STAMPS
    __builtin_amdgcn_s_waitcnt(0);   // (vmcnt(0): this wave's stores acknowledged)
    STAMP(4);
    if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 8192) {
        unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID
        g_stamps[blockIdx.x * 8 + 5] = hwid | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);   // XCC_ID
    }
#endif
}

template <bool AK, bool BKM, int WM, int WN, int MI, int NI, bool VIEW = false>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_glds_kernel(GemmK d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm_glds_body<AK, BKM, WM, WN, MI, NI, VIEW>(d, blockIdx.x, blockIdx.y, smem);
}

// Several independent GEMMs of one layout in ONE launch (block ranges by prefix sums). Used for the
// weight gradients of a layer: each is only 64-192 output tiles with K = all tokens, so alone it
// either leaves CUs idle or needs split-K (atomics at ~1.3 TB/s); together they fill the chip with
// full-K tiles, no atomics, deterministic sums.
constexpr int GEMM_GROUP_MAX = 6;
struct GemmGroup {
    int n;
    int start[GEMM_GROUP_MAX + 1];
    GemmK sub[GEMM_GROUP_MAX];
};

template <bool AK, bool BKM>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_glds_group_kernel(GemmGroup grp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int gi = 0;
#pragma unroll
    for (int i = 1; i < GEMM_GROUP_MAX; ++i)
        if (i < grp.n && (int)blockIdx.x >= grp.start[i]) gi = i;
    gemm_glds_body<AK, BKM, 2, 2, 4, 4>(grp.sub[gi], blockIdx.x - grp.start[gi], 0, smem);
}


// ---- multi-stage variant: NSTAGE LDS stages, NSTAGE-1 K tiles in flight, counted vmcnt -----------------
// The 2-stage kernel has one K tile in flight per workgroup: fine when two workgroups share a CU and the
// grid fills the chip, but a workgroup that is ALONE on its CU (small M: few tiles) then pays a full
// L2/HBM round trip per K tile (~3 us measured at 72 workgroups). Here tile t+NSTAGE-1 is staged while
// tile t is computed; the wait before each barrier is a COUNTED s_waitcnt vmcnt (never 0 inside the loop)
// and the barrier is a raw s_barrier, so the LDS-DMA stays in flight across barriers.
//   <.., 2,2,4,4, 4>: 128 x 128 tile, 4 waves, 4 stages (128 KB LDS): small grids (<= 1 workgroup per CU)
//   <.., 2,4,9,2, 3>: 288 x 128 tile, 8 waves, 3 stages (156 KB LDS): opt-in (NBCI_GEMM3=1), measured no
//                     faster than the 2-stage 144-row kernel on the full-chip shapes
template <bool AK, bool BKM, int WM, int WN, int MI, int NI, int NSTAGE, bool VIEW = false>   // VIEW: B (row-major-in-k) may be an overlapping-window view
__global__ __launch_bounds__(WM * WN * 64) void gemm_glds_ms_kernel(GemmK d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = WM * WN;
    constexpr int BM = WM * MI * 16, BN = WN * NI * 16;
    static_assert(BN == 128, "B tile is always 128 wide");
    static_assert(AK || BM == 128, "row-major-in-k A needs 256-byte tile rows");
    constexpr int A_BYTES = BM * 128, STAGE = A_BYTES + 16384;
    constexpr int NPA = A_BYTES / 1024, NPB = 16;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = w / WN, wn = w % WN;
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
    const int m0 = tm * BM, n0 = tn * 128;
    OperandK A = d.A, B = d.B;
    const int z = blockIdx.y;
    const int z1 = z / d.zdiv, z2 = z % d.zdiv;
    A.ptr = (const bf16_t*)A.ptr + z1 * d.azs1 + z2 * d.azs2;
    B.ptr = (const bf16_t*)B.ptr + z1 * d.bzs1 + z2 * d.bzs2;
    const long long coff = z1 * d.czs1 + z2 * d.czs2;
    const int nt = d.K / 64;

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    GldsOperand<AK, NPA, NW> ga;
    GldsOperand<BKM, NPB, NW> gb;
    glds_setup<AK, NPA, NW>(ga, A, m0, d.M, w, lane);
    glds_setup<BKM, NPB, NW>(gb, B, n0, d.N, w, lane);
    if constexpr (VIEW) {   // (tiles are staged strictly in K order below: the view addressing steps from tile to tile)
        static_assert(AK && !BKM, "multi-stage view kernel: k-major A (plain or window view), row-major-in-k B view");
        glds_view_seek<NPB, NW>(gb, B, 0);
    }
    const int lw = (NPA - w + NW - 1) / NW + (NPB - w + NW - 1) / NW;   // LDS-DMA instructions this wave issues per tile

#pragma unroll
    for (int p = 0; p < NSTAGE - 1; ++p)
        if (p < nt) {
            glds_stage<AK, NPA, NW>(ga, A, smem + p * STAGE, p, w);
            glds_stage<BKM, NPB, NW, VIEW>(gb, B, smem + p * STAGE + A_BYTES, p, w);
        }
    int cur = 0, nxt = NSTAGE - 1;   // stage holding tile kt ; stage that tile kt + NSTAGE - 1 goes to
    for (int kt = 0; kt < nt; ++kt) {
        // tile kt has landed once only the younger tiles' loads (already issued: kt+1 .. kt+NSTAGE-2) remain
        const int ahead = min(NSTAGE - 2, nt - 1 - kt);
        wait_vmcnt(lw * ahead);
        __builtin_amdgcn_s_barrier();   // every wave's pieces of tile kt are visible; stage `nxt` is no longer read
        asm volatile("" ::: "memory");
        if (kt + NSTAGE - 1 < nt) {
            char* nx = smem + nxt * STAGE;
            glds_stage<AK, NPA, NW>(ga, A, nx, kt + NSTAGE - 1, w);
            glds_stage<BKM, NPB, NW, VIEW>(gb, B, nx + A_BYTES, kt + NSTAGE - 1, w);
        }
        const char* sA = smem + cur * STAGE;
        compute_tile_g<AK, BKM, MI, NI>(sA, sA + A_BYTES, acc, wm * MI * 16, wn * NI * 16, lane);
        cur = (cur == NSTAGE - 1) ? 0 : cur + 1;
        nxt = (nxt == NSTAGE - 1) ? 0 : nxt + 1;
    }
    gemm_epilogue_tile<MI, NI>(d, acc, wm * MI * 16, wn * NI * 16, m0, n0, BM, coff, t, NW * 64, smem);
}

template <bool AK, bool BKM, int WM, int WN, int MI, int NI, int NSTAGE, bool VIEW = false>
static int launch_ms(const GemmK& k, dim3 grid, hipStream_t s) {
    constexpr int lds = NSTAGE * (WM * MI * 16 * 128 + 16384);
    TRY_(ensure_dyn_lds((const void*)gemm_glds_ms_kernel<AK, BKM, WM, WN, MI, NI, NSTAGE, VIEW>, lds, "gemm_glds_ms"));
    if (prof_on()) {
        static const std::string sym = std::string("gemm_glds_ms_kernel<") + (AK ? "true" : "false") + ", " + (BKM ? "true" : "false") + ", " + std::to_string(WM) + ", " +
                                       std::to_string(WN) + ", " + std::to_string(MI) + ", " + std::to_string(NI) + ", " + std::to_string(NSTAGE) + ", " + (VIEW ? "true" : "false") + ">";
        prof_note_symbol(sym.c_str());
    }
    hipLaunchKernelGGL((gemm_glds_ms_kernel<AK, BKM, WM, WN, MI, NI, NSTAGE, VIEW>), grid, dim3(WM * WN * 64), lds, s, k);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NBCI_EHIP, std::string("gemm_glds_ms launch: ") + hipGetErrorString(e));
    return NBCI_OK;
}

// ---- host --------------------------------------------------------------------------------------
static bool glds_operand_ok(const nbci_operand& o, int R) {
    if (((uintptr_t)o.ptr) % 16) return false;
    if (o.ld % 8 || (o.rpb > 0 && o.gstride % 8) || (o.zs1 % 8) || (o.zs2 % 8)) return false;
    // padded extent must stay inside the storage row (a window view's rows are longer than ld by design)
    if (!o.kmajor && o.rpb == 0 && o.ld < ((R + 7) & ~7)) return false;
    return true;
}

// a row-major-in-k operand that is an overlapping-window view: its k rows need a division per K tile (VIEW kernels)
bool glds_view(const nbci_gemm_desc& d) { return (!d.A.kmajor && d.A.rpb > 0) || (!d.B.kmajor && d.B.rpb > 0); }

bool glds_eligible(const nbci_gemm_desc& d, const GemmK& k) {
    (void)k;
    if (d.in_dtype != NBCI_BF16 || d.K < 64) return false;
    if (glds_view(d) && d.B.kmajor) return false;   // the view loops exist for a row-major-in-k B (A: either layout)
    return glds_operand_ok(d.A, d.M) && glds_operand_ok(d.B, d.N);
}

template <bool AK, bool BKM, int WM, int WN, int MI, int NI, bool VIEW = false>
static int launch_glds(const GemmK& k, dim3 grid, hipStream_t s) {
    constexpr int stage2 = 2 * (WM * MI * 16 * 128 + 16384), epi = WM * MI * 16 * EPI_LD * 4 / ((WM == 1 && MI == 10) ? 2 : 1);   // K-loop stages / epilogue tile (chunk)
    constexpr int lds = stage2 > epi ? stage2 : epi;
    if (lds > 65536) TRY_(ensure_dyn_lds((const void*)gemm_glds_kernel<AK, BKM, WM, WN, MI, NI, VIEW>, lds, "gemm_glds"));
    if (prof_on()) {
        static const std::string sym = std::string("gemm_glds_kernel<") + (AK ? "true" : "false") + ", " + (BKM ? "true" : "false") + ", " + std::to_string(WM) + ", " +
                                       std::to_string(WN) + ", " + std::to_string(MI) + ", " + std::to_string(NI) + ", " + (VIEW ? "true" : "false") + ">";
        prof_note_symbol(sym.c_str());
    }
    hipLaunchKernelGGL((gemm_glds_kernel<AK, BKM, WM, WN, MI, NI, VIEW>), grid, dim3(GEMM_THREADS), lds, s, k);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NBCI_EHIP, std::string("gemm_glds launch: ") + hipGetErrorString(e));
    return NBCI_OK;
}

template <int WM, int WN, int MI, int NI>
static int launch_glds_layout(const GemmK& k, bool ak, bool bk, dim3 grid, hipStream_t s) {
    if constexpr (WM * MI * 16 == 128) {
        if (ak && bk) return launch_glds<true, true, WM, WN, MI, NI>(k, grid, s);
        if (ak && !bk) return launch_glds<true, false, WM, WN, MI, NI>(k, grid, s);
        if (!ak && bk) return launch_glds<false, true, WM, WN, MI, NI>(k, grid, s);
        return launch_glds<false, false, WM, WN, MI, NI>(k, grid, s);
    } else {  // tall tiles exist for k-major A only
        if (bk) return launch_glds<true, true, WM, WN, MI, NI>(k, grid, s);
        return launch_glds<true, false, WM, WN, MI, NI>(k, grid, s);
    }
}

bool gemm_streamk_wanted(const nbci_gemm_desc* descs, const GemmK* ks, int n);   // gemm_streamk.hip
int gemm_streamk_launch(const nbci_gemm_desc* descs, const GemmK* ks, int n, hipStream_t stream);

int gemm_group_launch(const nbci_gemm_desc* descs, const GemmK* ks, int n, hipStream_t stream) {
    NBCI_REQUIRE(n >= 1 && n <= GEMM_GROUP_MAX, NBCI_EINVAL, "gemm group: 1..6 problems");
    GemmGroup grp;
    grp.n = n;
    grp.start[0] = 0;
    const bool ak = descs[0].A.kmajor != 0, bk = descs[0].B.kmajor != 0;
    for (int i = 0; i < n; ++i) {
        NBCI_REQUIRE((descs[i].A.kmajor != 0) == ak && (descs[i].B.kmajor != 0) == bk, NBCI_EINVAL, "gemm group: mixed layouts");
        NBCI_REQUIRE(ks[i].splitk == 1 && (descs[i].batch <= 1), NBCI_EINVAL, "gemm group: no split-K / batch");
        NBCI_REQUIRE(glds_eligible(descs[i], ks[i]) && !glds_view(descs[i]), NBCI_EALIGN, "gemm group: operand not eligible for the direct-to-LDS path");
        grp.sub[i] = ks[i];
        grp.sub[i].tiles_m = (descs[i].M + 127) / 128;
        grp.start[i + 1] = grp.start[i] + grp.sub[i].tiles_m * grp.sub[i].tiles_n;
    }
    for (int i = n; i < GEMM_GROUP_MAX; ++i) grp.start[i + 1] = grp.start[n];
    if (gemm_streamk_wanted(descs, ks, n)) return gemm_streamk_launch(descs, ks, n, stream);   // tile count leaves slots idle: deal out K tiles instead
    dim3 grid(grp.start[n]);
    constexpr int lds = 128 * EPI_LD * 4;   // the epilogue tile (67.6 KB) is a little larger than the two K-loop stages
    TRY_(ensure_dyn_lds(ak ? (bk ? (const void*)gemm_glds_group_kernel<true, true> : (const void*)gemm_glds_group_kernel<true, false>)
                           : (bk ? (const void*)gemm_glds_group_kernel<false, true> : (const void*)gemm_glds_group_kernel<false, false>), lds, "gemm group"));
    if (prof_on()) prof_note_symbol((std::string("gemm_glds_group_kernel<") + (ak ? "true" : "false") + ", " + (bk ? "true" : "false") + ">").c_str());
    if (ak && bk) hipLaunchKernelGGL((gemm_glds_group_kernel<true, true>), grid, dim3(GEMM_THREADS), lds, stream, grp);
    else if (ak && !bk) hipLaunchKernelGGL((gemm_glds_group_kernel<true, false>), grid, dim3(GEMM_THREADS), lds, stream, grp);
    else if (!ak && bk) hipLaunchKernelGGL((gemm_glds_group_kernel<false, true>), grid, dim3(GEMM_THREADS), lds, stream, grp);
    else hipLaunchKernelGGL((gemm_glds_group_kernel<false, false>), grid, dim3(GEMM_THREADS), lds, stream, grp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NBCI_EHIP, std::string("gemm group launch: ") + hipGetErrorString(e));
    return NBCI_OK;
}

bool gemm_pc_eligible(const nbci_gemm_desc& d, const GemmK& k);
int gemm_pc_mode();

// CUs the tile cost model counts on: the current device's (hipDeviceProp.multiProcessorCount, looked up once per device), or
// fewer when nbci_set_available_cus narrowed it (an overlapped all-reduce occupies some) - an explicit process-wide setting of the caller's
static std::atomic<int> g_cu_override{0};
void set_available_cus(int cus) { g_cu_override.store(cus); }
int available_cus() {
    const int o = g_cu_override.load();
    static std::mutex mu;
    static std::map<int, int> per_dev;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return o > 0 ? o : 256;
    int n;
    {
        std::lock_guard<std::mutex> l(mu);
        auto it = per_dev.find(dev);
        if (it == per_dev.end()) {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
            it = per_dev.emplace(dev, v).first;
        }
        n = it->second;
    }
    return (o > 0 && o < n) ? o : n;
}
long gemm_pc_tiles(const nbci_gemm_desc& d);
int gemm_pc_launch(const nbci_gemm_desc& d, GemmK k, hipStream_t stream);

int gemm_glds_launch(const nbci_gemm_desc& d, GemmK k, hipStream_t stream) {
    const int batch = d.batch > 0 ? d.batch : 1;
    const int splitk = k.splitk;
    // producer / consumer kernel (gemm_pc.hip): 144 x 256 tiles, ONE workgroup per CU. In-box A/B on the step's shapes
    // (tools/time_gemm_pc.py, profiles/r02_gemm_pc_ab.txt): 3 - 7 % faster than the two-workgroup kernels from K = 3072 up (fewer
    // staged and fragment bytes per FLOP), equal at K = 1024 with one round of tiles, 4 % slower at K = 1024 with 3 - 4 rounds (a
    // single resident workgroup overlaps nothing with its fill and epilogue). Mode 1 (default) therefore takes it for K >= 2048
    // when the last round of 256 workgroups is at least 3/4 full; mode 2 = whenever eligible (tests, measurement); 0 = never.
    const int pc_mode = gemm_pc_mode();
    if (pc_mode && gemm_pc_eligible(d, k)) {
        const long tiles = gemm_pc_tiles(d);
        const long cus = available_cus(), last = tiles % cus;
        if (pc_mode == 2 || (d.K >= 2048 && tiles >= 3 * cus / 4 && (last == 0 || last >= 3 * cus / 4))) return gemm_pc_launch(d, k, stream);
    }
    // tile height: minimise (rounds of 2 blocks/CU) x (rows per tile). Tall tiles need k-major A,
    // whole K tiles and no split-K.
    int bm = 128;
    const long avail = available_cus();
    if (d.A.kmajor && d.K % 64 == 0 && splitk == 1) {
        // cost model: (rounds) x (rows per tile) / (relative main-loop speed). The 3-stage 288-row kernel
        // runs one workgroup per CU (rounds of 256 tiles) but streams ~1.5x faster per row.
        double best = -1;
        const int cands[5] = {128, 144, 160, 288, 80};  // (160 rows: two-chunk epilogue, 208 VGPRs; 192 rows spill accumulators with hipcc 7.2: not offered)
        static const bool g3_off = measure_env("NBCI_GEMM3", 0) != 1;  // opt-in: measured no faster than the 2-stage kernel
        // 80-row tiles (MI = 5): 52 KB of LDS per workgroup -> THREE workgroups per CU, whose fills / epilogues overlap each other's K loops.
        // NBCI_GEMM_BM80 (measurement builds): 0 never, 1 by the cost model below (default), 2 whenever eligible
        static const int bm80 = measure_env("NBCI_GEMM_BM80", 1);
        static const double bm80_pen = (double)measure_env("NBCI_GEMM_BM80_PEN", 135) / 100.0;   // K-loop cost per row relative to the 144-row tile (49 vs 68 FLOP per staged byte)
        for (int c : cands) {
            if (c == 288 && (g3_off || d.K < 192 || glds_view(d))) continue;
            if (c == 160 && glds_view(d)) continue;   // (view launches pick their own 160-row case below)
            if (c == 80 && (bm80 == 0 || glds_view(d))) continue;
            const long tiles = (long)((d.M + c - 1) / c) * k.tiles_n * batch;
            if (c == 288 && tiles < 192) continue;   // one workgroup per CU: only worth it when the chip fills
            const long s1 = avail, s2 = 2 * avail, s3 = 3 * avail;   // workgroup slots per round (one / two / three workgroups per CU)
            double cost = c == 288 ? (double)((tiles + s1 - 1) / s1) * c / 2.0 / 1.5 : (double)((tiles + s2 - 1) / s2) * c;
            if (c == 80) {   // measured (profiles/r04_gemm_bm80_shapes.txt): a gain only where the 80-row tiles fit ONE round of at most two per CU
                             // (M = 4576 / 2288 x N = 1024: 0.86 of the 128-row launch); equal or slower with more tiles or rounds
                // and not below the grid sizes the multi-stage small-grid kernels take (128-row tiles <= one per CU: B = 16 equal, B = 8 / 4
                // 13 - 18 % slower per step with the 80-row kernel, profiles/r04_gemm_bm80_steps.txt)
                const long tiles128 = (long)((d.M + 127) / 128) * k.tiles_n * batch;
                if (bm80 != 2 && (tiles > s2 || tiles128 <= s1)) continue;
                cost = bm80 == 2 ? 0.0 : (double)c * bm80_pen;
            }
            if (best < 0 || cost < best) { best = cost; bm = c; }
        }
    }
    k.tiles_m = (d.M + bm - 1) / bm;
    dim3 grid(k.tiles_m * k.tiles_n * (splitk > 1 ? splitk : 1), splitk > 1 ? 1 : batch);
    const bool ak = d.A.kmajor != 0, bk = d.B.kmajor != 0;
    if (glds_view(d)) {   // (eligibility: B row-major-in-k)
        if (!ak) return launch_glds<false, false, 2, 2, 4, 4, true>(k, grid, stream);
        // small grids (the embedder's phase GEMM at small batches: B = 8 -> 80 tiles of 128 rows, K = 8192, one workgroup per CU on a
        // third of the chip, a load round trip per K tile in the 2-stage kernel: 123 us): 64-row tiles, five stages in flight
        if (d.K % 64 == 0 && d.K >= 256 && splitk == 1 && d.M > 64 && (long)((d.M + 127) / 128) * k.tiles_n * batch <= 128 && measure_env("NBCI_GEMM_S64", 1) != 0) {
            k.tiles_m = (d.M + 63) / 64;
            return launch_ms<true, false, 1, 4, 4, 2, 5, true>(k, dim3(k.tiles_m * k.tiles_n, batch), stream);
        }
        // k-major A: 160-row tiles as well (the embedder's phase GEMM has M = B * T/stride = 9600 rows: 60 x 8 = 480 tiles, one round,
        // where 128- / 144-row tiles need a second, mostly empty one)
        if (d.K % 64 == 0 && splitk == 1) {
            const long t160 = (long)((d.M + 159) / 160) * k.tiles_n * batch;
            const long s2 = 2 * avail;
            const double c160 = (double)((t160 + s2 - 1) / s2) * 160, ccur = (double)(((long)grid.x * grid.y + s2 - 1) / s2) * bm;
            if (c160 < ccur) {
                k.tiles_m = (d.M + 159) / 160;
                return launch_glds<true, false, 1, 4, 10, 2, true>(k, dim3(k.tiles_m * k.tiles_n, batch), stream);
            }
        }
        return bm == 144 ? launch_glds<true, false, 1, 4, 9, 2, true>(k, grid, stream) : launch_glds<true, false, 2, 2, 4, 4, true>(k, grid, stream);
    }
    // small grids (at most one workgroup per CU): nothing else hides the per-tile load latency -> 4-stage pipeline
    static const bool ms_off = measure_env("NBCI_GEMM_MS", 1) == 0;
    // very small grids (<= half the CUs at 128-row tiles): 64-row tiles double the workgroup count; 5 stages of 24 KB
    static const bool s64_off = measure_env("NBCI_GEMM_S64", 1) == 0;
    if (!ms_off && !s64_off && bm == 128 && ak && splitk == 1 && d.K % 64 == 0 && d.K >= 256 && (long)grid.x * grid.y <= 128 && d.M > 64) {
        k.tiles_m = (d.M + 63) / 64;
        dim3 g64(k.tiles_m * k.tiles_n, batch);
        return bk ? launch_ms<true, true, 1, 4, 4, 2, 5>(k, g64, stream) : launch_ms<true, false, 1, 4, 4, 2, 5>(k, g64, stream);
    }
    if (!ms_off && bm == 128 && splitk == 1 && d.K % 64 == 0 && d.K >= 256 && (long)grid.x * grid.y <= 256) {
        if (ak && bk) return launch_ms<true, true, 2, 2, 4, 4, 4>(k, grid, stream);
        if (ak && !bk) return launch_ms<true, false, 2, 2, 4, 4, 4>(k, grid, stream);
        if (!ak && bk) return launch_ms<false, true, 2, 2, 4, 4, 4>(k, grid, stream);
        return launch_ms<false, false, 2, 2, 4, 4, 4>(k, grid, stream);
    }
    switch (bm) {
        case 80: return launch_glds_layout<1, 4, 5, 2>(k, ak, bk, grid, stream);
        case 144: return launch_glds_layout<1, 4, 9, 2>(k, ak, bk, grid, stream);
        case 160: return launch_glds_layout<1, 4, 10, 2>(k, ak, bk, grid, stream);
        case 288:
            return bk ? launch_ms<true, true, 2, 4, 9, 2, 3>(k, grid, stream) : launch_ms<true, false, 2, 4, 9, 2, 3>(k, grid, stream);
        default: return launch_glds_layout<2, 2, 4, 4>(k, ak, bk, grid, stream);
    }
}

}  // namespace nbci

#ifdef NBCI_STAMPS
extern "C" int nbci_debug_read_kstamps(unsigned long long* host, int nblocks) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(nbci::g_kstamps), (size_t)nblocks * 64 * sizeof(unsigned long long));
}
extern "C" int nbci_debug_read_stamps(unsigned long long* host, int nblocks) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(nbci::g_stamps), (size_t)nblocks * 8 * sizeof(unsigned long long));
}
#endif
