// gemm_streamk.hip — balanced launch of a GEMM group whose tile count does not fill the chip's workgroup slots.
//
// A layer's four weight gradients are 384 full-K tiles of 128 x 128 (K = all tokens) on 512 workgroup slots (256 CUs x 2): with
// one tile per workgroup a quarter of the slots idle for the whole launch, and tiles cannot be cut to fit (1024- and 3072-row
// problems, 6.3 M outputs / 512 = 12 288 per tile has no 16-multiple factorisation that divides both). Splitting K with f32
// atomics was measured slower than the imbalance it removes. Here the launch is S workgroups, one per slot, and
// the K TILES of all output tiles — tile-major, I = sum(tiles_p x ktiles_p) of them — are dealt out in equal contiguous runs
// ("stream-K"): every workgroup does I / S K tiles, whatever the tile count. A run that starts or ends inside an output tile
// yields a partial accumulator:
//   * the workgroup that holds a tile's LAST K tile is the tile's owner: it adds the others' partials to its accumulators and runs
//     the normal epilogue (alpha, beta, ... — anything gemm_epilogue_tile does), so the result is stored once and the summation
//     order is fixed (deterministic, unlike atomics);
//   * every other workgroup touching the tile is a contributor: it writes its accumulators — registers as they are, 1 KB per
//     wave instruction — to its own 64 KB slot of a scratch buffer and publishes a flag (release, agent scope).
// A workgroup has at most one contributor run (the one that does not reach its tile's end) and does it FIRST, so by the time an
// owner has finished its own share the partials it needs have usually been there for a while. The owner spins on the
// contributors' flags. Progress: a contributor run waits for nothing, so an owner only ever waits for work that is running or will
// run as soon as a slot frees - provided every workgroup of the launch gets a slot eventually, which the grid size guarantees (one
// workgroup per slot of the CURRENT device's CUs: available_cus(), from hipDeviceProp unless the caller narrowed it). Contiguous and
// aligned schemes put a tile's contributors at LOWER block indices than its owner (dispatched first in practice; speed, not
// correctness). The opt-in blocked scheme (mode 4) does not: its helpers sit at HIGHER block indices than the owners they serve
// (owner j * 8 + xcd waits for helper j * 8 + nb + xcd / per), so there the argument is residency alone.
// Placement (b % 8 = which blocks share an XCD) is a speed assumption only: partials and flags go past every cache level.
// Every spin is BOUNDED (SK_SPIN_LIMIT sleeps, about a second): on expiry the owner gives up, counts the event in the scratch
// buffer's timeout word and stores its tile as NaN - a LOUD wrong tile (the next loss / AdamW step is NaN) instead of a hung GPU or a
// silently short sum; nbci_streamk_timeouts() reads the count, NativeTrainer.read_stats / save_checkpoint and bench.py raise on it.
// Flags carry the launch's epoch (a per-process counter), so they are never reset.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

#include "gemm_glds.h"
#include "kernels.h"

namespace nbci {

constexpr int SK_MAX = 6;
constexpr int SK_SPIN_LIMIT = 1 << 22;   // x s_sleep(8) ~ 0.25 us each
constexpr int SK_MAXP = 8;   // aligned scheme: most pieces (scratch slots) one helper workgroup may produce
constexpr int SK_LDS = 128 * EPI_LD * 4;   // staging / epilogue image; one more 16-byte slot behind it holds the owner's "gave up" word
struct StreamK {
    int n;
    int tile_start[SK_MAX + 1];    // prefix sums of output tiles per problem
    int iter_start[SK_MAX + 1];    // prefix sums of K tiles (tile-major) per problem
    int kt[SK_MAX];                // K tiles per output tile
    int xcd_tile[9];               // XCD x works on tiles [xcd_tile[x], xcd_tile[x + 1])
    int wpx;                       // workgroups per XCD (grid = 8 wpx)
    float* partial;                // [grid][16][256] float4
    int* flags;                    // [grid]
    int* timeouts;                 // one word: owners that gave up waiting (see SK_SPIN_LIMIT)
    int epoch;
    int dbg;                       // measurement only: 1 = no partial exchange at all (wrong results)
    int aligned, q, lk, ex, tx, r, maxp;   // aligned scheme (see the kernel): owner K tiles, remainder, helpers and tiles per XCD, tiles per helper (0: not integral), slots per helper
    int nb;                        // aligned == 2 (blocked scheme): number of 64-tile blocks = owner XCDs
    GemmK sub[SK_MAX];
};

struct SkPos { int p, tile, k; };   // problem, tile inside the problem, K tile inside the output tile

__device__ __forceinline__ SkPos sk_locate(const StreamK& s, int it) {
    int p = 0;
#pragma unroll
    for (int i = 1; i < SK_MAX; ++i)
        if (i < s.n && it >= s.iter_start[i]) p = i;
    const int r = it - s.iter_start[p];
    SkPos o;
    o.p = p; o.tile = r / s.kt[p]; o.k = r - o.tile * s.kt[p];
    return o;
}
__device__ __forceinline__ int sk_iter_of_tile(const StreamK& s, int gtile) {   // first K tile of global output tile `gtile`
    int p = 0;
#pragma unroll
    for (int i = 1; i < SK_MAX; ++i)
        if (i < s.n && gtile >= s.tile_start[i]) p = i;
    return s.iter_start[p] + (gtile - s.tile_start[p]) * s.kt[p];
}
// run j of an XCD whose K tiles are [i0, i0 + len): [i0 + j len / wpx, i0 + (j + 1) len / wpx)
__device__ __forceinline__ int sk_run_begin(int i0, int len, int wpx, int j) { return i0 + (int)(((long long)j * len) / wpx); }

// a pointer the inline asm below takes as a SCALAR base: made wave-uniform explicitly (the "s" constraint does not insert the readfirstlane
// when the compiler keeps a — provably uniform — value in vector registers)
__device__ __forceinline__ float4* sk_uniform(float4* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (float4*)(((unsigned long long)hi << 32) | lo);
}

// One piece: K tiles [kb, ke) of output tile `tile` of problem p. write_slot >= 0: a contributor piece, the raw accumulators go to that
// scratch slot and its flag is raised; otherwise an owner piece: the partials in slot_of(0 .. nadd-1) are added (in that order) and the
// normal epilogue stores the tile.
template <bool AK, bool BKM, typename SlotOf>
__device__ __forceinline__ void sk_piece(const StreamK& s, char* smem, int p, int tile, int kb, int ke, int write_slot, int nadd, SlotOf slot_of) {
    constexpr int MI = 4, NI = 4, BM = 128;
    constexpr int A_BYTES = BM * 128, STAGE = A_BYTES + 16384, NPA = A_BYTES / 1024, NPB = 16;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = w / 2, wn = w % 2;
    const GemmK& d = s.sub[p];
    int tm, tn;
    {
        const int per_group = 8 * d.tiles_n;
        const int grp = tile / per_group, in_grp = tile % per_group;
        const int first_m = grp * 8;
        const int gsize = min(8, d.tiles_m - first_m);
        tm = first_m + in_grp % gsize;
        tn = in_grp / gsize;
    }
    const int m0 = tm * BM, n0 = tn * 128;
    f32x4 acc[MI][NI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    GldsOperand<AK, NPA> ga;
    GldsOperand<BKM, NPB> gb;
    glds_setup<AK, NPA>(ga, d.A, m0, d.M, w, lane);
    glds_setup<BKM, NPB>(gb, d.B, n0, d.N, w, lane);
    int cur = 0;
    glds_stage<AK, NPA, 4>(ga, d.A, smem, kb, w);
    glds_stage<BKM, NPB, 4>(gb, d.B, smem + A_BYTES, kb, w);
    __syncthreads();
    for (int kt = kb; kt < ke; ++kt) {
        if (kt + 1 < ke) {
            char* nx = smem + (cur ^ 1) * STAGE;
            glds_stage<AK, NPA, 4>(ga, d.A, nx, kt + 1, w);
            glds_stage<BKM, NPB, 4>(gb, d.B, nx + A_BYTES, kt + 1, w);
        }
        const char* sA = smem + cur * STAGE;
        compute_tile_g<AK, BKM, MI, NI>(sA, sA + A_BYTES, acc, wm * MI * 16, wn * NI * 16, lane);
        __syncthreads();
        cur ^= 1;
    }
    if (ke == s.kt[p] && (d.K & 63)) {   // the piece that ends the tile also takes the partial last K tile: masked register-staged loads (zero fill)
        uint4 ra[4], rb[4];
        load_chunks<bf16_t, AK>(d.A, m0, d.M, ke * 64, d.K, ra, t);
        load_chunks<bf16_t, BKM>(d.B, n0, d.N, ke * 64, d.K, rb, t);
        char* sA = smem + cur * STAGE;
        store_chunks<bf16_t, AK>(sA, ra, t);
        store_chunks<bf16_t, BKM>(sA + A_BYTES, rb, t);
        __syncthreads();
        compute_tile_g<AK, BKM, MI, NI>(sA, sA + A_BYTES, acc, wm * MI * 16, wn * NI * 16, lane);
        __syncthreads();
    }
    if (write_slot >= 0) {
        if (!(s.dbg & 1)) {
            float4* slot = sk_uniform((float4*)s.partial + (size_t)write_slot * (MI * NI * GEMM_THREADS));
            // write-through stores (sc0 sc1: past every non-coherent cache level), acknowledged before the flag goes out: no cache-wide
            // writeback / invalidate, which would also throw out the operand panels the other workgroups are reusing.
            // (s_nop 4: the scalar base may have just been written by a VALU instruction — v_readlane of a spilled SGPR — and the hazard
            // recogniser does not look inside inline asm: VALU-writes-SGPR -> VMEM-reads-it needs 5 wait states)
#pragma unroll
            for (int a = 0; a < MI; ++a)
#pragma unroll
                for (int b = 0; b < NI; ++b)
                    asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 sc0 sc1" ::"v"(t * 16), "v"(acc[a][b]), "s"(slot + (a * NI + b) * GEMM_THREADS) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) __hip_atomic_store(s.flags + write_slot, s.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    volatile int* gave_up = (volatile int*)(smem + SK_LDS);   // written by thread 0 only, monotone within a piece
    int timed_out = 0;
    const bool exchange = nadd > 0 && !(s.dbg & 1);
    for (int i = 0; i < nadd && !(s.dbg & 1); ++i) {
        const int sl = slot_of(i);
        const float4* slot = sk_uniform((float4*)s.partial + (size_t)sl * (MI * NI * GEMM_THREADS));
        if (t == 0) {
            int spins = 0;
            while (__hip_atomic_load(s.flags + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != s.epoch) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > SK_SPIN_LIMIT) { atomicAdd(s.timeouts, 1); timed_out = 1; break; }
            }
            *gave_up = timed_out;
        }
        __syncthreads();
        // the partial was never cached here (first touch) but may be on another XCD: read it past the caches as well
#pragma unroll
        for (int a = 0; a < MI; ++a) {
            f32x4 v[NI];
#pragma unroll
            for (int b = 0; b < NI; ++b)
                asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 sc0 sc1" : "=v"(v[b]) : "v"(t * 16), "s"(slot + (a * NI + b) * GEMM_THREADS) : "memory");
            // (the loaded registers are operands of the wait: the adds below must not be scheduled above it)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])::"memory");
            static_assert(NI == 4, "the wait names the four loaded registers");
#pragma unroll
            for (int b = 0; b < NI; ++b) acc[a][b] += v[b];
        }
    }
    if (exchange && *gave_up) {   // (the last iteration's barrier ordered thread 0's write before this read) a contributor never showed up:
        const float qnan = __builtin_nanf("");   // poison the tile so that the failure cannot pass for a gradient
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
            for (int b = 0; b < NI; ++b) acc[a][b] = (f32x4){qnan, qnan, qnan, qnan};
    }
    gemm_epilogue_tile<MI, NI>(d, acc, wm * MI * 16, wn * NI * 16, m0, n0, BM, 0, t, GEMM_THREADS, smem);
    __syncthreads();   // the epilogue's LDS tile is read out before the next piece stages into it
}

template <bool AK, bool BKM>
__global__ __launch_bounds__(GEMM_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_streamk_kernel(StreamK s_by_value) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // The description is read where it lies, in the kernel-argument segment (it is the only argument: offset 0). Indexing the by-value
    // copy with a run-time problem number made the compiler move the whole 2 KB struct to scratch once a third call site appeared.
    const StreamK& s = *(const StreamK*)__builtin_amdgcn_kernarg_segment_ptr();
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    if (s.aligned == 2) {
        // ---- blocked scheme (the NDT1 layer group: six blocks of 8 x 8 tiles, one K, 512 slots). A block's 64 tiles share 8 + 8
        // operand panels; with its 64 owner workgroups on ONE XCD, all in lockstep over K tiles [0, q), every panel slab crosses the
        // fabric once for that XCD instead of once per XCD that holds some of the block's tiles. The remaining XCDs run the helpers:
        // workgroup j of helper XCD hx does K tiles [q, kt) of tile j in each of its `per` blocks, one after the other — again 64
        // workgroups in lockstep on one block. Helper j serves owner j (block indices within 8 of each other: dispatched together).
        const int nb = s.nb, per = nb / (8 - nb);
        if (j >= 64) return;
        if (xcd >= nb) {
            const int hx = xcd - nb;
            for (int k = 0; k < per; ++k) {
                const int gt = (hx * per + k) * 64 + j;
                int p = 0;
#pragma unroll
                for (int i = 1; i < SK_MAX; ++i)
                    if (i < s.n && gt >= s.tile_start[i]) p = i;
                sk_piece<AK, BKM>(s, smem, p, gt - s.tile_start[p], s.q, s.kt[0], (int)blockIdx.x * s.maxp + k, 0, [](int) { return 0; });
            }
            return;
        }
        const int gt = xcd * 64 + j;
        int p = 0;
#pragma unroll
        for (int i = 1; i < SK_MAX; ++i)
            if (i < s.n && gt >= s.tile_start[i]) p = i;
        sk_piece<AK, BKM>(s, smem, p, gt - s.tile_start[p], 0, s.q, -1, 1, [&](int) { return (j * 8 + nb + xcd / per) * s.maxp + xcd % per; });
        return;
    }
    if (s.aligned) {
        // ---- aligned scheme (all problems share K, fewer tiles than slots): the XCD's tx tiles have one OWNER workgroup each (blocks
        // j >= ex) doing K tiles [0, q) in lockstep — same panel reuse as one workgroup per tile — and ex HELPER workgroups (j < ex,
        // i.e. lower block indices: dispatched first, never waiting) sweep the remainders [q, kt) tile after tile in equal runs.
        const int ex = s.ex, tx = s.tx, lk = s.lk, q = s.q, L = tx * lk;
        auto run_begin = [&](int c) { return (int)(((long long)c * L) / ex); };
        if (j < ex) {
            int pos = run_begin(j);
            const int end = run_begin(j + 1), u0 = pos / lk;
            while (pos < end) {
                const int u = pos / lk, koff = pos - u * lk, n = min(lk - koff, end - pos);
                const int gt = xcd * tx + (s.r > 0 ? (u % s.r) * ex + u / s.r : u);   // helpers running side by side work on neighbouring tiles
                int p = 0;
#pragma unroll
                for (int i = 1; i < SK_MAX; ++i)
                    if (i < s.n && gt >= s.tile_start[i]) p = i;
                sk_piece<AK, BKM>(s, smem, p, gt - s.tile_start[p], q + koff, q + koff + n, (int)blockIdx.x * s.maxp + (u - u0), 0, [](int) { return 0; });
                pos += n;
            }
            return;
        }
        const int tau = j - ex;
        if (tau >= tx) return;
        const int gt = xcd * tx + tau, u = s.r > 0 ? (tau % ex) * s.r + tau / ex : tau;
        auto helper_of = [&](int pos) {
            int c = (int)(((long long)pos * ex) / L);
            while (c > 0 && run_begin(c) > pos) --c;
            while (run_begin(c + 1) <= pos) ++c;
            return c;
        };
        const int c_lo = helper_of(u * lk), c_hi = helper_of((u + 1) * lk - 1);
        int p = 0;
#pragma unroll
        for (int i = 1; i < SK_MAX; ++i)
            if (i < s.n && gt >= s.tile_start[i]) p = i;
        sk_piece<AK, BKM>(s, smem, p, gt - s.tile_start[p], 0, q, -1, c_hi - c_lo + 1,
                          [&](int i) { const int c = c_lo + i; return (c * 8 + xcd) * s.maxp + (u - run_begin(c) / lk); });
        return;
    }
    // ---- contiguous scheme (any K per problem, any tile count): the XCD's K tiles, tile-major, in wpx equal runs
    const int i0 = sk_iter_of_tile(s, s.xcd_tile[xcd]);
    const int len = (s.xcd_tile[xcd + 1] == s.tile_start[s.n] ? s.iter_start[s.n] : sk_iter_of_tile(s, s.xcd_tile[xcd + 1])) - i0;
    if (len <= 0) return;
    const int g_begin = sk_run_begin(i0, len, s.wpx, j);
    int g_end = sk_run_begin(i0, len, s.wpx, j + 1);
    if (g_begin >= g_end) return;   // (host guarantees len >= wpx; kept for safety: an empty run neither owns nor contributes)

    // the run's last piece: a contributor piece if it stops short of its tile's last K tile
    int first = -1, first_end = 0;   // [first, first_end): piece to do first (-1: none)
    {
        const SkPos e = sk_locate(s, g_end - 1);
        if (e.k + 1 < s.kt[e.p]) {
            first = max(g_begin, g_end - 1 - e.k);
            first_end = g_end;
            g_end = first;
        }
    }
    if (first >= 0) {
        const SkPos ps = sk_locate(s, first);
        sk_piece<AK, BKM>(s, smem, ps.p, ps.tile, ps.k, ps.k + (first_end - first), (int)blockIdx.x, 0, [](int) { return 0; });
    }
    for (int g = g_begin; g < g_end;) {
        const SkPos ps = sk_locate(s, g);
        const int ktn = s.kt[ps.p];
        int jc = j;
        if (ps.k > 0) {   // contributors: the runs of this XCD that cover [tile begin, g): run indices jc .. j - 1
            const int tile_it = g - ps.k;
            jc = (int)((((long long)(tile_it - i0)) * s.wpx) / len);
            while (jc > 0 && sk_run_begin(i0, len, s.wpx, jc) > tile_it) --jc;
            while (sk_run_begin(i0, len, s.wpx, jc + 1) <= tile_it) ++jc;
        }
        sk_piece<AK, BKM>(s, smem, ps.p, ps.tile, ps.k, ktn, -1, j - jc, [&](int i) { return (jc + i) * 8 + xcd; });   // an owner piece always runs to its tile's end
        g += ktn - ps.k;
    }
}

// ---- host ----------------------------------------------------------------------------------------------------------------
struct SkScratch { float* partial; int* flags; int slots; };   // flags[slots] = the timeout word
static std::mutex g_sk_mu;
static std::map<std::pair<int, hipStream_t>, SkScratch> g_sk_scratch;
static std::atomic<int> g_sk_epoch{1};
static int g_sk_mode = -1;   // -1: read NBCI_STREAMK on first use (default 1)

int gemm_streamk_mode() {
    if (g_sk_mode < 0) {
        g_sk_mode = measure_env("NBCI_STREAMK", 1);
    }
    return g_sk_mode;
}
void gemm_streamk_set_mode(int m) { g_sk_mode = m; }

// owners that gave up waiting for a partial since the scratch buffers were created (synchronises every stream that has one)
int gemm_streamk_timeouts(long long* out) {
    std::lock_guard<std::mutex> lk(g_sk_mu);
    long long n = 0;
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (auto& kv : g_sk_scratch) {
        int v = 0;
        (void)hipSetDevice(kv.first.first);
        if (hipStreamSynchronize(kv.first.second) != hipSuccess ||
            hipMemcpy(&v, kv.second.flags + kv.second.slots, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) {
            (void)hipSetDevice(cur);
            return fail(NBCI_EHIP, "stream-K timeouts: read back");
        }
        n += v;
    }
    (void)hipSetDevice(cur);
    *out = n;
    return NBCI_OK;
}

int gemm_streamk_release() {
    std::lock_guard<std::mutex> lk(g_sk_mu);
    for (auto& kv : g_sk_scratch) {
        (void)hipFree(kv.second.partial);
        (void)hipFree(kv.second.flags);
    }
    g_sk_scratch.clear();
    return NBCI_OK;
}

static int sk_scratch(hipStream_t stream, int slots, SkScratch& out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(NBCI_EHIP, "stream-K scratch: hipGetDevice");
    std::lock_guard<std::mutex> lk(g_sk_mu);
    auto key = std::make_pair(dev, stream);
    auto it = g_sk_scratch.find(key);
    if (it != g_sk_scratch.end() && it->second.slots >= slots) { out = it->second; return NBCI_OK; }
    if (it != g_sk_scratch.end()) {
        // an earlier launch on this stream may still be reading the old buffers
        if (hipStreamSynchronize(stream) != hipSuccess) return fail(NBCI_EHIP, "stream-K scratch: sync");
        (void)hipFree(it->second.partial);
        (void)hipFree(it->second.flags);
        g_sk_scratch.erase(it);
    }
    SkScratch s;
    s.slots = slots;
    if (hipMalloc((void**)&s.partial, (size_t)slots * 16 * GEMM_THREADS * sizeof(float4)) != hipSuccess)
        return fail(NBCI_EHIP, "stream-K scratch: hipMalloc (partials)");
    if (hipMalloc((void**)&s.flags, ((size_t)slots + 1) * sizeof(int)) != hipSuccess) {
        (void)hipFree(s.partial);
        return fail(NBCI_EHIP, "stream-K scratch: hipMalloc (flags)");
    }
    if (hipMemsetAsync(s.flags, 0, ((size_t)slots + 1) * sizeof(int), stream) != hipSuccess) return fail(NBCI_EHIP, "stream-K scratch: memset");
    g_sk_scratch[key] = s;
    out = s;
    return NBCI_OK;
}

// Fills the launch description; false when the group cannot be dealt out (an XCD's share of the tiles with fewer K tiles than workgroups).
static bool sk_plan(const nbci_gemm_desc* descs, const GemmK* ks, int n, StreamK& s) {
    if (n < 1 || n > SK_MAX) return false;
    s.n = n;
    s.tile_start[0] = 0;
    s.iter_start[0] = 0;
    long iters = 0;
    for (int i = 0; i < n; ++i) {
        if (descs[i].K < 64) return false;   // (kt counts the FULL K tiles; a partial last one rides with the piece that ends its tile)
        s.sub[i] = ks[i];
        s.sub[i].tiles_m = (descs[i].M + 127) / 128;
        s.kt[i] = descs[i].K / 64;
        const int tl = s.sub[i].tiles_m * s.sub[i].tiles_n;
        iters += (long)tl * s.kt[i];
        if (iters > (1L << 30)) return false;
        s.tile_start[i + 1] = s.tile_start[i] + tl;
        s.iter_start[i + 1] = s.iter_start[i] + tl * s.kt[i];
    }
    for (int i = n; i < SK_MAX; ++i) { s.tile_start[i + 1] = s.tile_start[n]; s.iter_start[i + 1] = s.iter_start[n]; s.kt[i] = 1; }
    const int tiles = s.tile_start[n];
    const int slots = 2 * (available_cus() & ~3);   // a multiple of 8
    if (slots < 64) return false;
    s.wpx = slots / 8;
    for (int x = 0; x <= 8; ++x) s.xcd_tile[x] = (int)(((long)tiles * x) / 8);
    for (int x = 0; x < 8; ++x) {   // every run needs at least one K tile (an XCD without tiles simply returns)
        long it = 0;
        for (int g = s.xcd_tile[x]; g < s.xcd_tile[x + 1]; ++g) {
            int p = 0;
            while (p + 1 < n && g >= s.tile_start[p + 1]) ++p;
            it += s.kt[p];
        }
        if (it != 0 && it < s.wpx) return false;
    }
    // aligned scheme: one K for all problems, tiles a multiple of 8 and fewer than the slots, a remainder of at least one K tile per tile,
    // at most SK_MAXP pieces per helper — and only where the contiguous runs' K phases do NOT fall into a few aligned classes by
    // themselves: with tiles / slots = a / b in lowest terms there are b phase classes; up to 4 (the NDT1 layer group: 3 / 4) the
    // contiguous scheme measured faster inside the train step (140 vs 148 us per launch), beyond it the aligned one (iTransformer
    // group, 21 / 32: 318 vs 374 us). NBCI_STREAMK_ALIGNED=0 / 1 forces never / whenever it applies (A/B); mode 3 does the latter too.
    s.aligned = 0; s.q = s.lk = s.ex = s.tx = s.r = 0; s.maxp = 1; s.nb = 0;
    {   // blocked scheme: every problem a whole number of 64-tile blocks with 8 tile columns, one K, 4 / 6 / 7 blocks on 512 slots.
        // OFF unless asked for (mode 4 or NBCI_STREAMK_BLOCKED=1): on the NDT1 layer group it halves the fabric traffic (FETCH_SIZE
        // 591 -> 279 MB per launch against 187 MB algorithmic) and is nevertheless 3 % slower inside the step (143.4 vs 138.8 us).
        static const bool env_blocked = measure_env("NBCI_STREAMK_BLOCKED", 0) == 1;
        bool ok = (env_blocked || gemm_streamk_mode() == 4) && slots == 512 && tiles % 64 == 0;
        for (int i = 0; i < n && ok; ++i) ok = s.kt[i] == s.kt[0] && s.sub[i].tiles_n == 8 && s.sub[i].tiles_m % 8 == 0;
        const int nb = tiles / 64;
        if (ok && (nb == 4 || nb == 6 || nb == 7)) {
            const int per = nb / (8 - nb);
            const int q = (s.kt[0] * nb + 7) / 8;
            if (q < s.kt[0] && per <= SK_MAXP) {
                s.aligned = 2; s.nb = nb; s.q = q; s.lk = s.kt[0] - q; s.maxp = per;
                return true;
            }
        }
    }
    static const int env_aligned = measure_env("NBCI_STREAMK_ALIGNED", -1);
    const bool no_aligned = env_aligned == 0;
    bool want = env_aligned == 1 || gemm_streamk_mode() == 3;
    if (!want && tiles > 0) {
        long a = tiles, b = slots;
        while (b) { const long t = a % b; a = b; b = t; }
        want = slots / a > 4;
    }
    bool same_k = want;
    for (int i = 1; i < n; ++i) same_k = same_k && s.kt[i] == s.kt[0];
    if (!no_aligned && same_k && tiles % 8 == 0 && tiles < slots) {
        const int kt = s.kt[0];
        const int q = (int)(((long)kt * tiles + slots - 1) / slots), lk = kt - q;
        const int ex = (slots - tiles) / 8, tx = tiles / 8;
        if (lk >= 1 && (long)tx * lk >= ex) {
            const long L = (long)tx * lk;
            const int run_max = (int)((L + ex - 1) / ex);
            const int maxp = (run_max + lk - 1) / lk + 1;
            if (maxp <= SK_MAXP) {
                s.aligned = 1; s.q = q; s.lk = lk; s.ex = ex; s.tx = tx; s.maxp = maxp;
                s.r = (tx % ex == 0) ? tx / ex : 0;
            }
        }
    }
    return true;
}

// Worth it when one-tile-per-workgroup rounds would leave a good part of the last round's slots empty (`tiles` 128 x 128 tiles on
// `slots` = 2 x available CUs). The caller has checked that every problem is a plain (non-view, unbatched, unsplit) direct-to-LDS one.
bool gemm_streamk_wanted(const nbci_gemm_desc* descs, const GemmK* ks, int n) {
    const int mode = gemm_streamk_mode();
    StreamK s;
    if (mode == 0 || !sk_plan(descs, ks, n, s)) return false;
    if (mode >= 2) return true;        // forced (tests)
    const long tiles = s.tile_start[n], iters = s.iter_start[n], slots = 8L * s.wpx;
    const long rounds = (tiles + slots - 1) / slots;
    if (rounds > 3 || iters < slots * 16) return false;
    return (double)tiles / (double)(rounds * slots) < 0.88;   // the classic launch would idle > 12 % of its slot-rounds
}

// what the grouped launch would do with this group (no launch, no device access): out = {dealt out (0 / 1), scheme (0 contiguous runs,
// 1 aligned, 2 blocked), workgroups, owner K tiles q, remainder K tiles, scratch slots, tiles, K tiles per tile of problem 0}
void gemm_streamk_describe(const nbci_gemm_desc* descs, const GemmK* ks, int n, int32_t* out8) {
    StreamK s;
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    if (!gemm_streamk_wanted(descs, ks, n) || !sk_plan(descs, ks, n, s)) return;
    out8[0] = 1; out8[1] = s.aligned; out8[2] = 8 * s.wpx; out8[3] = s.q; out8[4] = s.lk; out8[5] = 8 * s.wpx * (s.aligned ? s.maxp : 1);
    out8[6] = s.tile_start[n]; out8[7] = s.kt[0];
}

int gemm_streamk_launch(const nbci_gemm_desc* descs, const GemmK* ks, int n, hipStream_t stream) {
    StreamK s;
    NBCI_REQUIRE(sk_plan(descs, ks, n, s), NBCI_EINVAL, "gemm stream-K: group cannot be dealt out");
    const bool ak = descs[0].A.kmajor != 0, bk = descs[0].B.kmajor != 0;
    const int slots = 8 * s.wpx;
    SkScratch sc;
    int rc = sk_scratch(stream, slots * (s.aligned ? s.maxp : 1), sc);   // (slot index = block index x maxp + piece)
    if (rc != NBCI_OK) return rc;
    s.partial = sc.partial;
    s.flags = sc.flags;
    s.timeouts = sc.flags + sc.slots;
    s.epoch = g_sk_epoch.fetch_add(1);
    s.dbg = measure_env("NBCI_STREAMK_DBG", 0);
    constexpr int lds = SK_LDS + 16;
    TRY_(ensure_dyn_lds(ak ? (bk ? (const void*)gemm_streamk_kernel<true, true> : (const void*)gemm_streamk_kernel<true, false>)
                           : (bk ? (const void*)gemm_streamk_kernel<false, true> : (const void*)gemm_streamk_kernel<false, false>), lds, "gemm stream-K"));
    dim3 grid(slots);
    if (prof_on()) prof_note_symbol((std::string("gemm_streamk_kernel<") + (ak ? "true" : "false") + ", " + (bk ? "true" : "false") + ">").c_str());
    if (ak && bk) hipLaunchKernelGGL((gemm_streamk_kernel<true, true>), grid, dim3(GEMM_THREADS), lds, stream, s);
    else if (ak && !bk) hipLaunchKernelGGL((gemm_streamk_kernel<true, false>), grid, dim3(GEMM_THREADS), lds, stream, s);
    else if (!ak && bk) hipLaunchKernelGGL((gemm_streamk_kernel<false, true>), grid, dim3(GEMM_THREADS), lds, stream, s);
    else hipLaunchKernelGGL((gemm_streamk_kernel<false, false>), grid, dim3(GEMM_THREADS), lds, stream, s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NBCI_EHIP, std::string("gemm stream-K launch: ") + hipGetErrorString(e));
    return NBCI_OK;
}

}  // namespace nbci
