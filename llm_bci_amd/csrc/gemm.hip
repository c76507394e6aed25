// gemm.hip — the MFMA GEMM of the NDT1 hot path (gfx950, wave64): generic kernel + host dispatch.
//
// C[z] = alpha * A[z](M x K) . B[z]^T(N x K), f32 accumulate, fused epilogue.
// Layouts: each operand is either k-major (storage rows = m/n, k contiguous) or
// row-major-in-k (storage rows = k, m/n contiguous). The second kind is read out of LDS with
// ds_read_b64_tr_b16 (bf16) so forward (x.W^T), data-grad (dy.W) and weight-grad (dy^T.x)
// products all run on one kernel family without transposed copies in HBM.
// Storage rows may be an overlapping-window view (rpb/gstride): nn.Unfold + Linear of the
// reference (models/ndt1.py:138,180) is a plain GEMM over that view.
//
// Two kernels share the LDS image, the fragment readers and the epilogue (gemm_common.h):
//  * gemm_kernel (this file): register-staged global->LDS with full edge masking; any shape,
//    alignment and dtype (bf16 or exact f32 MFMA). Tile 128 x 128 x BK, 4 waves of 64 x 64.
//  * gemm_glds_kernel (gemm_glds.hip): bf16 fast path, global_load_lds (direct-to-LDS, 16 B per
//    lane) for every full K tile, tile height chosen per problem so the grid fills whole rounds.
// The MFMA is issued with operands swapped (B fragment first) so every lane ends up with
// 4 CONSECUTIVE n for one m: epilogue loads/stores are 16-byte (f32) / 8-byte (bf16) wide.
#include <set>
#include <cstdio>
#include <cstring>
#include "gemm_common.h"

namespace nbci {

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
    // XCD-aware bijective remap: blocks b, b+8, ... share an XCD (round-robin dispatch), so give every
    // XCD a contiguous run of work items. Split-K launches are 1-D over (split, tile) with the split
    // index slowest: one XCD then owns (most of) one K slice, whose A/B slices fit its 4 MB L2 and are
    // fetched from HBM once instead of once per tile row/column.
    const int nwg = d.tiles_m * d.tiles_n;
    const int nitems = nwg * (d.splitk > 1 ? d.splitk : 1);
    int item;
    {
        const int orig = blockIdx.x, xcd = orig & 7, q = nitems >> 3, r = nitems & 7;
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
    const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;

    const int z = d.splitk > 1 ? zsplit : (int)blockIdx.y;
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

    if (d.splitk > 1) {
        gemm_epilogue<4, 4>(d, acc, m0 + wm * 64, n0 + wn * 64, coff, lane, w, smem);
        return;
    }
    gemm_epilogue_tile<4, 4>(d, acc, wm * 64, wn * 64, m0, n0, GEMM_BM, coff, t, GEMM_THREADS, smem);
}

// ---- host side ---------------------------------------------------------------------------
bool glds_eligible(const nbci_gemm_desc& d, const GemmK& k);
bool glds_view(const nbci_gemm_desc& d);
int gemm_glds_launch(const nbci_gemm_desc& d, GemmK k, hipStream_t stream);
int gemm_group_launch(const nbci_gemm_desc* descs, const GemmK* ks, int n, hipStream_t stream);
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
    // split-K transposes the 4 x 16 KB accumulator tiles through LDS in its epilogue; otherwise the row-contiguous
    // epilogue needs a 128 x 132 f32 tile (67.6 KB: above the 64 KB default limit -> attribute)
    constexpr int epi = GEMM_BM * EPI_LD * 4;
    const int lds = k.splitk > 1 ? (4 * GemmTile<T>::REGION < 65536 ? 65536 : 4 * GemmTile<T>::REGION)
                                 : (4 * GemmTile<T>::REGION < epi ? epi : 4 * GemmTile<T>::REGION);
    TRY_(ensure_dyn_lds((const void*)gemm_kernel<T, AK, BKM>, epi, "gemm"));
    if (prof_on()) {
        static const std::string sym = std::string("gemm_kernel<") + (sizeof(T) == 2 ? "__bf16" : "float") + ", " + (AK ? "true" : "false") + ", " + (BKM ? "true" : "false") + ">";
        prof_note_symbol(sym.c_str());
    }
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

int build_gemmk(const nbci_gemm_desc& d, GemmK& k) {   // (also used by mlp_strip.hip)
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
    k.colsum_rc = RepCfg{d.colsum_rep_stride, d.colsum_nrep > 1 ? d.colsum_nrep : 1};
    k.tile_row = 0;
    k.residual = (const float*)d.residual; k.ldr = d.ldr; k.residual_bf16 = (d.residual && d.residual_dtype == NBCI_BF16) ? 1 : 0;
    k.residual_rows = (const long long*)d.residual_rows; k.residual_first = d.residual_first;
    k.gate = d.gate; k.ldg = d.ldg; k.gate_act = d.gate_act; k.gate_bf16 = d.in_dtype == NBCI_BF16; k.gate_coff = d.gate_follows_c ? 1 : 0;
    k.c2_grad = d.c2_grad;
    // vector epilogue: 4 consecutive n at 16-byte (f32) / 8-byte (bf16) alignment
    bool cvec = (d.ldc % 4 == 0) && (d.czs1 % 4 == 0) && (d.czs2 % 4 == 0) && (((uintptr_t)d.C) % 16 == 0);
    if (d.C2) cvec = cvec && (((uintptr_t)d.C2) % 16 == 0);
    if (d.bias) cvec = cvec && (((uintptr_t)d.bias) % 16 == 0);
    if (d.residual) cvec = cvec && (d.ldr % 4 == 0) && (((uintptr_t)d.residual) % 16 == 0);
    k.cvec = cvec ? 1 : 0;
    { static const int dbg = measure_env("NBCI_GEMM_DBG", 0); k.dbg = dbg; }
    k.epi_mode = epi_mode_of(k);
    {   // NBCI_GEMM_LOGMODE=1: print each distinct epilogue feature set once (which ones deserve a specialised row loop)
        static const bool logm = measure_env("NBCI_GEMM_LOGMODE", 0) == 1;
        if (logm) {
            static std::set<int> seen;
            if (seen.insert(k.epi_mode).second) fprintf(stderr, "[nbci] gemm epilogue mode 0x%x (M %d N %d K %d)\n", k.epi_mode, d.M, d.N, d.K);
        }
    }
    { static const bool gen = measure_env_str("NBCI_GEMM_GENERIC_EPI") != nullptr; if (gen) k.epi_mode = EPI_GENERIC; }   // A/B: force the run-time-tested row loop
    return NBCI_OK;
}

// NBCI_GEMM_LOG=<file>: one line per GEMM kernel launch, in launch order ("M N K batch kind nproblems"; kind as in
// nbci_profile_collect; grouped launches list every problem: "G n M N K M N K ..."). Measurement aid: tools/roofline_report.py joins
// it with a rocprofv3 kernel trace of the same run (launch i of the trace = line i), so every launch's algorithmic FLOPs are known.
static FILE* gemm_log_file() {
    static FILE* f = [] { const char* e = measure_env_str("NBCI_GEMM_LOG"); return (e && e[0]) ? fopen(e, "w") : (FILE*)nullptr; }();
    return f;
}
static int kind_of(const nbci_gemm_desc& d) { return (d.in_dtype == NBCI_BF16 ? 4 : 0) + (d.A.kmajor ? 2 : 0) + (d.B.kmajor ? 1 : 0); }

int gemm_launch(const nbci_gemm_desc& d, hipStream_t stream) {
    GemmK k;
    int rc = build_gemmk(d, k);
    if (rc != NBCI_OK) return rc;
    if (FILE* f = gemm_log_file()) { fprintf(f, "S %d %d %d %d %d\n", d.M, d.N, d.K, d.batch > 0 ? d.batch : 1, kind_of(d)); fflush(f); }
    const int batch = d.batch > 0 ? d.batch : 1;
    const int splitk = d.splitk > 1 ? d.splitk : 1;
    if (d.in_dtype == NBCI_BF16 && glds_eligible(d, k)) return gemm_glds_launch(d, k, stream);
    dim3 grid(k.tiles_m * k.tiles_n * (splitk > 1 ? splitk : 1), splitk > 1 ? 1 : batch);
    if (d.in_dtype == NBCI_BF16) return launch_layout<bf16_t>(k, d.A.kmajor != 0, d.B.kmajor != 0, grid, stream);
    return launch_layout<float>(k, d.A.kmajor != 0, d.B.kmajor != 0, grid, stream);
}

// Host-only: how gemm_grouped_launch would launch this group (nbci_debug_gemm_grouped_plan). out8[0] = -1: one launch per problem.
void gemm_streamk_describe(const nbci_gemm_desc* descs, const GemmK* ks, int n, int32_t* out8);
int gemm_grouped_describe(const nbci_gemm_desc* descs, int n, int32_t* out8) {
    NBCI_REQUIRE(descs && n >= 1 && out8, NBCI_EINVAL, "gemm grouped plan: no problems");
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    GemmK ks[6];
    bool ok = n <= 6;
    for (int i = 0; i < n && ok; ++i) {
        int rc = build_gemmk(descs[i], ks[i]);
        if (rc != NBCI_OK) return rc;
        ok = descs[i].in_dtype == NBCI_BF16 && ks[i].splitk == 1 && descs[i].batch <= 1 && glds_eligible(descs[i], ks[i]) && !glds_view(descs[i]) &&
             (descs[i].A.kmajor != 0) == (descs[0].A.kmajor != 0) && (descs[i].B.kmajor != 0) == (descs[0].B.kmajor != 0);
    }
    if (!ok) { out8[0] = -1; return NBCI_OK; }
    gemm_streamk_describe(descs, ks, n, out8);
    return NBCI_OK;
}

// Up to 6 independent bf16 GEMMs of one layout in one launch; falls back to one launch each when a
// problem does not qualify for the direct-to-LDS kernel.
int gemm_grouped_launch(const nbci_gemm_desc* descs, int n, hipStream_t stream) {
    NBCI_REQUIRE(descs && n >= 1, NBCI_EINVAL, "gemm grouped: no problems");
    GemmK ks[6];
    bool ok = n <= 6;
    for (int i = 0; i < n && ok; ++i) {
        int rc = build_gemmk(descs[i], ks[i]);
        if (rc != NBCI_OK) return rc;
        ok = descs[i].in_dtype == NBCI_BF16 && ks[i].splitk == 1 && descs[i].batch <= 1 && glds_eligible(descs[i], ks[i]) && !glds_view(descs[i]) &&
             (descs[i].A.kmajor != 0) == (descs[0].A.kmajor != 0) && (descs[i].B.kmajor != 0) == (descs[0].B.kmajor != 0);
    }
    if (ok) {
        if (FILE* f = gemm_log_file()) {
            fprintf(f, "G %d %d", n, kind_of(descs[0]));
            for (int i = 0; i < n; ++i) fprintf(f, " %d %d %d", descs[i].M, descs[i].N, descs[i].K);
            fprintf(f, "\n"); fflush(f);
        }
        return gemm_group_launch(descs, ks, n, stream);
    }
    for (int i = 0; i < n; ++i) {
        int rc = gemm_launch(descs[i], stream);
        if (rc != NBCI_OK) return rc;
    }
    return NBCI_OK;
}

}  // namespace nbci

// ---- optional per-launch timing (bench.py roofline leg): HIP events around every launch a ProfScope brackets ----
#include <map>
#include <mutex>
#include <vector>
namespace nbci {
struct ProfRec { hipEvent_t a, b; double flops, bytes; int kind; std::string name; };
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static thread_local ProfRec t_open;          // the scope this thread has open (launches do not nest)
static thread_local bool t_open_valid = false;

void gemm_profile_enable(bool on) { std::lock_guard<std::mutex> l(g_prof_mu); g_prof_on = on; }
bool gemm_profile_on() { return g_prof_on; }
bool prof_on() { return g_prof_on; }

void prof_begin(const char* name, double flops, double bytes, int kind, hipStream_t s) {
    if (t_open_valid) return;                // (a nested scope: the outer one times the launch)
    ProfRec r;
    if (hipEventCreate(&r.a) != hipSuccess) return;
    if (hipEventCreate(&r.b) != hipSuccess) { (void)hipEventDestroy(r.a); return; }
    r.flops = flops; r.bytes = bytes; r.kind = kind; r.name = name ? name : "";
    (void)hipEventRecord(r.a, s);
    t_open = r; t_open_valid = true;
}
void prof_note_symbol(const char* sym) { if (t_open_valid && sym) t_open.name = sym; }
void prof_end(hipStream_t s) {
    if (!t_open_valid) return;
    (void)hipEventRecord(t_open.b, s);
    { std::lock_guard<std::mutex> l(g_prof_mu); g_prof.push_back(t_open); }
    t_open_valid = false;
}

static double gemm_bytes(const nbci_gemm_desc& d) {   // operands once in, result once out (+ the residual / gate / second output the epilogue touches)
    const double es = d.in_dtype == NBCI_BF16 ? 2 : 4, cs = d.c_dtype == NBCI_BF16 ? 2 : 4, batch = d.batch > 0 ? d.batch : 1;
    double b = es * ((double)d.M * d.K + (double)d.N * d.K) * batch + cs * (double)d.M * d.N * batch;
    if (d.residual && !d.residual_rows) b += (d.residual_dtype == NBCI_BF16 ? 2.0 : 4.0) * d.M * d.N * batch;
    if (d.gate) b += es * (double)d.M * d.N * batch;
    if (d.C2) b += cs * (double)d.M * d.N * batch;
    if (d.beta != 0.f) b += cs * (double)d.M * d.N * batch;
    return b;
}

int gemm_launch_timed(const nbci_gemm_desc& d, hipStream_t stream) {
    if (!g_prof_on) return gemm_launch(d, stream);
    const int batch = d.batch > 0 ? d.batch : 1;
    ProfScope ps("gemm", 2.0 * d.M * (double)d.N * d.K * batch, gemm_bytes(d), stream, kind_of(d));
    return gemm_launch(d, stream);
}

int gemm_grouped_launch_timed(const nbci_gemm_desc* descs, int n, hipStream_t stream) {
    if (!g_prof_on) return gemm_grouped_launch(descs, n, stream);
    double fl = 0.0, by = 0.0;
    for (int i = 0; i < n; ++i) { fl += 2.0 * descs[i].M * (double)descs[i].N * descs[i].K; by += gemm_bytes(descs[i]); }
    ProfScope ps("gemm_grouped", fl, by, stream, kind_of(descs[0]));
    return gemm_grouped_launch(descs, n, stream);
}

static int prof_drain(std::vector<ProfRec>& out) {
    { std::lock_guard<std::mutex> l(g_prof_mu); out.swap(g_prof); }
    return NBCI_OK;
}

// out[kind] = {total ms, total flops, launches} for the GEMM launches, kind = dtype*4 + A.kmajor*2 + B.kmajor (8 kinds); other
// records are dropped. (Kept for callers of the first interface; prof_collect_text reports every kernel.)
int gemm_profile_collect(double* out24) {
    for (int i = 0; i < 24; ++i) out24[i] = 0.0;
    std::vector<ProfRec> recs;
    prof_drain(recs);
    int rc = NBCI_OK;
    for (auto& r : recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) rc = fail(NBCI_EHIP, "profile collect");
        else if (r.kind >= 0 && r.kind < 8) { out24[r.kind * 3 + 0] += ms; out24[r.kind * 3 + 1] += r.flops; out24[r.kind * 3 + 2] += 1.0; }
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    }
    return rc;
}

int prof_collect_text(char* buf, long long cap) {
    std::vector<ProfRec> recs;
    prof_drain(recs);
    struct Agg { double ms = 0, flops = 0, bytes = 0; long n = 0; };
    std::map<std::string, Agg> agg;
    int rc = NBCI_OK;
    for (auto& r : recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) rc = fail(NBCI_EHIP, "profile collect");
        else { Agg& a = agg[r.name]; a.ms += ms; a.flops += r.flops; a.bytes += r.bytes; a.n += 1; }
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    }
    std::string out;
    char line[640];
    for (auto& kv : agg) {
        snprintf(line, sizeof(line), "%s\t%ld\t%.6f\t%.6e\t%.6e\n", kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.flops, kv.second.bytes);
        out += line;
    }
    if ((long long)out.size() + 1 > cap) return fail(NBCI_EINVAL, "profile_collect_text: buffer too small");
    memcpy(buf, out.c_str(), out.size() + 1);
    return rc;
}
}  // namespace nbci
