// mlp_strip.hip — the MLP half of an encoder layer in ONE launch with a row strip resident on its CU (DESIGN.md §4 "Round 4"):
//     g = act(h W_up^T + b_up)   (also stored, with act', for the backward)          reference: NeuralMLP.forward, models/ndt1.py:224-227
//     y = x + dropout(g W_down^T + b_down)                                          and the residual add of NeuralEncoderLayer.forward, :328
// Today these are two GEMM launches (each one round of 144 x 128 tiles: 36 + 35 us at M = 9152, of which 2 x 15 us are K loop; the rest is launch
// ramp, pipeline fill and epilogue, paid twice, and g makes a round trip through L2 / HBM in between). Here ONE workgroup per CU owns a strip of
// R = ceil(M / CUs) rows (36 at M = 9152; at most 40) for the whole chain:
//   phase 1: for every 128-column tile of W_up: stream its 16 K tiles (and the strip's h tiles) through a three-stage LDS ring by LDS-DMA, 12 MFMAs
//            per wave and K tile; epilogue: bias + GELU -> g strip into LDS (bf16, 16 k-tile images, the A operand of phase 2) and g / act' to global;
//   phase 2: for every 128-column tile of W_down: stream its K tiles, A fragments come from the resident g strip; epilogue = the normal fused one
//            (bias, dropout, residual, store).
// Every workgroup pulls both weight matrices (4 MB at 1024 x 1024) through its CU once: `tools/probe_weight_stream.hip` measures that floor at 34 us for
// 256 workgroups (123 GB/s per CU with the next tile in flight) - 36 FLOP per staged byte, i.e. load-bound by design; the bet is that 34 us of streaming
// + one fill + small epilogues beats 71 us of two launches.
// Prototype scope: bf16, k-major operands, K1 % 64 == 0, N1 % 128 == 0 and N1 <= 1024 (the g strip must fit LDS), N2 % 128 == 0, M <= 40 x CUs.
#include <cstdlib>

#include "gemm_glds.h"

namespace nbci {

int build_gemmk(const nbci_gemm_desc& d, GemmK& k);   // gemm.hip

constexpr int MS_ROWS = 40;                 // rows of a strip image: 5 LDS-DMA pieces of 8 rows x 128 B
constexpr int MS_AIMG = MS_ROWS * 128;      // 5 120 B: one k-tile image [40 rows][64 k] of the strip
constexpr int MS_ASTAGE = 48 * 128;         // a stage reserves 48 rows for the A tile: fragment reads of row block 2 (rows 32 .. 47) stay inside it
constexpr int MS_STAGE = MS_ASTAGE + 16384; // + the 128 x 64 B tile
constexpr int MS_NS = 3;
constexpr int MS_THREADS = 512;      // 8 waves, two per SIMD (each covers the other's fragment-read / DMA-issue latencies): 16 output columns per wave
constexpr int MS_NW = MS_THREADS / 64;

struct MlpStrip {
    GemmK up, down;
    int rows;        // rows per strip
    int lds_g;       // bytes of the g strip region = (N1 / 64) * MS_AIMG
};

__device__ __forceinline__ void ms_wait_vmcnt(int n) {   // n is wave-uniform; s_waitcnt needs an immediate
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

__global__ __launch_bounds__(MS_THREADS) void mlp_strip_kernel(MlpStrip P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const MlpStrip& p = *(const MlpStrip*)__builtin_amdgcn_kernarg_segment_ptr();
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int i16 = lane & 15, g4 = lane >> 4;
    const int m0 = blockIdx.x * p.rows;
    const int M = p.up.M;
    if (m0 >= M) return;
    const int mlim = min(M, m0 + p.rows);
    char* gimg = smem;                    // [N1 / 64][40 rows][64 k] bf16, swizzled as the GEMM kernels' k-major A images
    char* ring = smem + p.lds_g;
    const int K1 = p.up.K, N1 = p.up.N, N2 = p.down.N;
    const int kt1 = K1 / 64, nt1 = N1 / 128, kt2 = N1 / 64, nt2 = N2 / 128;
    const int steps1 = kt1 * nt1, steps = steps1 + kt2 * nt2;

    GldsOperand<true, 5, MS_NW> ga;
    GldsOperand<true, 16, MS_NW> gb1, gb2;
    glds_setup<true, 5, MS_NW>(ga, p.up.A, m0, M, w, lane);
    glds_setup<true, 16, MS_NW>(gb1, p.up.B, 0, N1, w, lane);
    glds_setup<true, 16, MS_NW>(gb2, p.down.B, 0, N2, w, lane);
    const bf16_t* b1_base = gb1.base;
    const bf16_t* b2_base = gb2.base;
    constexpr int NBW = 16 / MS_NW;       // B pieces per wave and tile
    const int na = (w < 5 % MS_NW ? 1 : 0) + 5 / MS_NW;   // this wave's LDS-DMA pieces of an A tile (5 pieces over the waves)

    // DMA of step s into ring stage s % 3; returns the wave's number of LDS-DMA instructions
    auto issue = [&](int s) -> int {
        char* st = ring + (s % MS_NS) * MS_STAGE;
        if (s < steps1) {
            const int nt = s / kt1, kt = s - nt * kt1;
            glds_stage<true, 5, MS_NW>(ga, p.up.A, st, kt, w);
            gb1.base = b1_base + (long long)nt * 128 * p.up.B.ld;
            glds_stage<true, 16, MS_NW>(gb1, p.up.B, st + MS_ASTAGE, kt, w);
            return na + NBW;
        }
        const int s2 = s - steps1;
        const int nt = s2 / kt2, kt = s2 - nt * kt2;
        gb2.base = b2_base + (long long)nt * 128 * p.down.B.ld;
        glds_stage<true, 16, MS_NW>(gb2, p.down.B, st + MS_ASTAGE, kt, w);
        return NBW;
    };

    constexpr int NI = 8 / MS_NW;         // 16-column blocks per wave (128-column tiles)
    f32x4 acc[3][NI];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int ops_next = 0;                     // LDS-DMA instructions of the tile AFTER the one about to be waited for (already in flight)
    bool drain = false;                   // the last step ended with global stores: their count in vmcnt is not known exactly -> wait for everything
    issue(0);
    if (steps > 1) ops_next = issue(1);
    // per-workgroup views of the two problems: rows beyond this strip belong to the next workgroup
    GemmK dn = p.down;
    dn.M = mlim;
    for (int s = 0; s < steps; ++s) {
        ms_wait_vmcnt(drain ? 0 : (s + 1 < steps ? ops_next : 0));   // tile s has landed (tile s + 1 may still be in flight)
        drain = false;
        __builtin_amdgcn_s_barrier();     // every wave's pieces of tile s are visible; the stage of tile s - 1 is free; (s == steps1: the g strip is complete)
        asm volatile("" ::: "memory");
        if (s + 2 < steps) ops_next = issue(s + 2); else ops_next = 0;
        // NOTE ops_next now describes tile s + 2; the wait at step s + 1 needs the count of tile s + 2 = exactly this value
        const char* st = ring + (s % MS_NS) * MS_STAGE;
        if (s < steps1) {
            compute_tile_g<true, true, 3, NI>(st, st + MS_ASTAGE, acc, 0, 16 * NI * w, lane);
            const int nt = s / kt1, kt = s - nt * kt1;
            if (kt == kt1 - 1) {
                // ---- phase-1 epilogue of column tile nt: bias + activation (+ act'), g into the LDS strip and g / act' to global
#pragma unroll
                for (int mi = 0; mi < 3; ++mi) {
                    const int r = 16 * mi + i16, m = m0 + r;
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        const int n = nt * 128 + 16 * NI * w + 16 * ni + 4 * g4;
                        const float4 b4 = *(const float4*)(p.up.bias + n);
                        float v[4] = {acc[mi][ni][0] + b4.x, acc[mi][ni][1] + b4.y, acc[mi][ni][2] + b4.z, acc[mi][ni][3] + b4.w};
                        float da[4];
                        act_fwd_bwd4(p.up.act, v, da);
                        const bf16x4 gv = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
                        if (r < MS_ROWS) {
                            const int c = n & 63;
                            *(bf16x4*)(gimg + (n >> 6) * MS_AIMG + r * 128 + ((((c >> 3) ^ ((r >> 1) & 7))) << 4) + (c & 7) * 2) = gv;
                        }
                        if (m < mlim) {
                            *(bf16x4*)((bf16_t*)p.up.C + (long long)m * p.up.ldc + n) = gv;
                            if (p.up.C2) {
                                const bf16x4 dv = {f2bf(da[0]), f2bf(da[1]), f2bf(da[2]), f2bf(da[3])};
                                *(bf16x4*)((bf16_t*)p.up.C2 + (long long)m * p.up.ldc + n) = dv;
                            }
                        }
                        acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the strip writes have landed before the next barrier publishes them
                drain = true;
            }
        } else {
            const int s2 = s - steps1;
            const int nt = s2 / kt2, kt = s2 - nt * kt2;
            compute_tile_g<true, true, 3, NI>(gimg + kt * MS_AIMG, st + MS_ASTAGE, acc, 0, 16 * NI * w, lane);
            if (kt == kt2 - 1) {
                gemm_epilogue<3, NI>(dn, acc, m0, nt * 128 + 16 * NI * w, 0, lane, w, ring);
#pragma unroll
                for (int mi = 0; mi < 3; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
                drain = true;
            }
        }
    }
}

int mlp_strip_launch(const nbci_gemm_desc& up, const nbci_gemm_desc& down, hipStream_t stream) {
    NBCI_REQUIRE(up.in_dtype == NBCI_BF16 && down.in_dtype == NBCI_BF16 && up.c_dtype == NBCI_BF16, NBCI_EINVAL, "mlp strip: bf16 operands, bf16 intermediate");
    NBCI_REQUIRE(up.A.kmajor && up.B.kmajor && down.B.kmajor && up.A.rpb == 0 && up.B.rpb == 0 && down.B.rpb == 0, NBCI_EINVAL, "mlp strip: k-major plain operands");
    NBCI_REQUIRE(up.M == down.M && up.N == down.K && up.K % 64 == 0 && up.N % 128 == 0 && up.N <= 1024 && down.N % 128 == 0, NBCI_ESHAPE,
                 "mlp strip: K1 % 64, N1 % 128 (<= 1024), N2 % 128, and the down projection consumes the up projection's output");
    NBCI_REQUIRE(up.bias && (up.batch <= 1) && (down.batch <= 1) && up.splitk <= 1 && down.splitk <= 1 && !up.residual && !up.gate && up.drop_p == 0.f &&
                 !down.C2 && !down.gate && !down.colsum && !up.colsum && up.ldc % 4 == 0,
                 NBCI_EINVAL, "mlp strip: up = bias + activation (+ act' copy), down = bias / dropout / residual");
    const int cus = available_cus();
    const int rows = (up.M + cus - 1) / cus;
    NBCI_REQUIRE(rows <= MS_ROWS, NBCI_ESHAPE, "mlp strip: more than 40 rows per CU");
    MlpStrip p;
    TRY_(build_gemmk(up, p.up));
    TRY_(build_gemmk(down, p.down));
    NBCI_REQUIRE(p.up.cvec && p.down.cvec, NBCI_EALIGN, "mlp strip: 16-byte aligned outputs / bias / residual");
    p.rows = rows;
    p.lds_g = (up.N / 64) * MS_AIMG;
    const int lds = p.lds_g + MS_NS * MS_STAGE;
    TRY_(ensure_dyn_lds((const void*)mlp_strip_kernel, lds, "mlp_strip"));
    const int grid = (up.M + rows - 1) / rows;
    if (prof_on()) prof_note_symbol("mlp_strip_kernel");
    hipLaunchKernelGGL(mlp_strip_kernel, dim3(grid), dim3(MS_THREADS), lds, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NBCI_EHIP, std::string("mlp strip launch: ") + hipGetErrorString(e));
    return NBCI_OK;
}

}  // namespace nbci

extern "C" int nbci_debug_mlp_strip(const nbci_gemm_desc* up, const nbci_gemm_desc* down, nbci_stream_t stream) {
    if (!up || !down) return nbci::fail(NBCI_EINVAL, "mlp strip: null descriptor");
    return nbci::mlp_strip_launch(*up, *down, (hipStream_t)stream);
}
