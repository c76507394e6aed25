// gemm_glds.h — direct-to-LDS staging helpers shared by gemm_glds.hip and gemm_wgrad.hip (layout notes: gemm_glds.hip).
#pragma once
#include "gemm_common.h"

namespace nbci {

typedef __attribute__((address_space(1))) void gvoid;
typedef __attribute__((address_space(3))) void lvoid;

template <bool KMAJOR, int NPIECES, int NW = 4>
struct GldsOperand {
    static constexpr int PER_WAVE = (NPIECES + NW - 1) / NW;
    const bf16_t* base;        // operand base (batch offset applied)
    long long off[PER_WAVE];   // per-piece element offset of this lane's chunk at k-tile 0
    long long step;            // element step per K tile (kmajor: 64; else 64 * ld), 0 if recomputed
    int col[PER_WAVE];         // (!KMAJOR, rpb view) column of the chunk
    int krow[PER_WAVE];        // (!KMAJOR) local k row of this lane in the piece
};

// per-lane source offsets for the pieces this wave stages
template <bool KMAJOR, int NPIECES, int NW = 4>
__device__ __forceinline__ void glds_setup(GldsOperand<KMAJOR, NPIECES, NW>& g, const OperandK& o, int row0, int R, int w, int lane) {
    g.base = (const bf16_t*)o.ptr;
#pragma unroll
    for (int i = 0; i < GldsOperand<KMAJOR, NPIECES, NW>::PER_WAVE; ++i) {
        const int p = w + NW * i;
        if constexpr (KMAJOR) {
            const int rl = 8 * p + (lane >> 3);
            int row = row0 + rl;
            if (row > R - 1) row = R - 1;                       // clamp: garbage rows are never stored
            const int c = (lane & 7) ^ ((rl >> 1) & 7);
            g.off[i] = row_offset(o, row) + c * 8;
            g.col[i] = 0; g.krow[i] = 0;
        } else {
            const int kl = 4 * p + (lane >> 4);
            const int c = ((((lane & 15) >> 1) ^ rm_swz(kl)) << 1) | (lane & 1);
            int col = row0 + c * 8;
            if (col + 8 > ((R + 7) & ~7)) col = 0;              // chunk entirely past the padded extent
            g.col[i] = col; g.krow[i] = kl;
            g.off[i] = (long long)kl * o.ld + col;              // plain (non-view) addressing
        }
    }
    g.step = KMAJOR ? 64 : 64 * o.ld;
}

// Issue this wave's LDS-DMA pieces of K tile `kt`: base + offset + kt * step, with NO branch in the K loop — the
// per-piece "is it a view" / "does this wave own the piece" tests used to cost ~60 scalar + vector instructions and
// half a dozen branches per piece per K tile.
// VIEW = a row-major-in-k operand may be an overlapping-window view (k rows are not equidistant across groups of
// rpb rows). Those kernels walk K sequentially: glds_view_seek() positions every piece at the first K tile (one
// division), then each call issues the CURRENT tile and steps 64 k rows ahead with adds only (`kt` is ignored).
template <bool KMAJOR, int NPIECES, int NW = 4, bool VIEW = false>
__device__ __forceinline__ void glds_stage(GldsOperand<KMAJOR, NPIECES, NW>& g, const OperandK& o, char* lds, int kt, int w) {
    constexpr int PW = GldsOperand<KMAJOR, NPIECES, NW>::PER_WAVE;
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int p = w + NW * i;
        const bf16_t* src;
        if constexpr (VIEW && !KMAJOR) {
            src = g.base + g.off[i];
            g.off[i] += 64 * o.ld;
            if (o.rpb > 0) {
                g.krow[i] += 64;
                while (g.krow[i] >= o.rpb) { g.krow[i] -= o.rpb; g.off[i] += o.gstride - (long long)o.rpb * o.ld; }
            }
        } else {
            src = g.base + g.off[i] + (long long)kt * g.step;
        }
        if constexpr (NPIECES % NW == 0) {
            __builtin_amdgcn_global_load_lds((gvoid*)src, (lvoid*)(lds + p * 1024), 16, 0, 0);
        } else {
            if (i < PW - 1 || p < NPIECES) __builtin_amdgcn_global_load_lds((gvoid*)src, (lvoid*)(lds + p * 1024), 16, 0, 0);
        }
    }
}

// (VIEW kernels) position a row-major-in-k operand at K tile kt0: off = element offset of this lane's chunk there,
// krow = its k row's index inside the group of rpb rows
template <int NPIECES, int NW>
__device__ __forceinline__ void glds_view_seek(GldsOperand<false, NPIECES, NW>& g, const OperandK& o, int kt0) {
#pragma unroll
    for (int i = 0; i < GldsOperand<false, NPIECES, NW>::PER_WAVE; ++i) {
        const int r = kt0 * 64 + g.krow[i];
        g.off[i] = row_offset(o, r) + g.col[i];
        g.krow[i] = o.rpb > 0 ? r % o.rpb : 0;
    }
}

__device__ __forceinline__ void wait_vmcnt(int n) {   // n is wave-uniform; s_waitcnt needs an immediate
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// Transposed fragment reads of a row-major-in-k tile, issued as INLINE ASM. Through the builtin
// (__builtin_amdgcn_ds_read_tr16_b64) hipcc 7.2 cannot tell that the read does not alias an LDS-DMA still in
// flight, and puts `s_waitcnt vmcnt(0)` in front of the first transposed read of every K tile: the prefetch of
// the next tile(s) is drained before the current tile is computed, load and compute run back to back, and deeper
// pipelines buy nothing (measured: weight-gradient GEMMs at one load round trip, ~1.25 us, per K tile whatever
// the stage count). The compiler neither waits for nor counts an asm read, so the caller issues
// `s_waitcnt lgkmcnt(0)` + sched_barrier before the first use (ordering against the LDS-DMA is the K loop's own
// vmcnt + barrier, as for the plain reads).
template <int OFF>
__device__ __forceinline__ s16x4 ds_read_tr_asm(unsigned addr) {
    s16x4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
__device__ __forceinline__ unsigned lds_addr(const char* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}
// byte address (k-step 0, low half) of the fragment of 16 columns starting at r0 (see read_frag_bf16<false>):
// k row = 32 ks + 8 g + q (+ 4 for the high half) -> the k-step adds 8192 B, the high half 1024 B; the granule
// swizzle rm_swz(k row) = (q & 3) | ((g & 1) << 2) does not depend on either
__device__ __forceinline__ unsigned tr_frag_base(const char* s, int r0, int i16, int g) {
    const int q = i16 >> 2, p = i16 & 3;
    const int swz = (q & 3) | ((g & 1) << 2);
    return lds_addr(s) + (8 * g + q) * 256 + ((((r0 >> 4) ^ swz) << 5) | (p << 3));
}
template <int KS>
__device__ __forceinline__ bf16x8 read_frag_tr_asm(unsigned base) {
    union { struct { s16x4 a, b; } p2; bf16x8 v; } u;
    u.p2.a = ds_read_tr_asm<8192 * KS>(base);
    u.p2.b = ds_read_tr_asm<8192 * KS + 1024>(base);
    return u.v;
}

// k-major tile [rows][64 k], 128-B rows, 16-B chunk ^= (row >> 1) & 7 (read_frag_bf16<true>): the fragment of rows
// r0 + 16 sb .. + 15 at k-step ks sits at  (base0 ^ 64 ks) + 2048 sb  (the swizzle term does not depend on sb)
__device__ __forceinline__ unsigned km_frag_base(const char* s, int r0, int i16, int g) {
    return lds_addr(s) + (r0 + i16) * 128 + ((g ^ ((i16 >> 1) & 7)) << 4);
}
template <int OFF>
__device__ __forceinline__ bf16x8 ds_read_b128_asm(unsigned addr) {
    bf16x8 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
template <int N, int SB = 0>
__device__ __forceinline__ void km_read_frags(bf16x8 (&f)[N], unsigned base_ks) {
    if constexpr (SB < N) {
        f[SB] = ds_read_b128_asm<2048 * SB>(base_ks);
        km_read_frags<N, SB + 1>(f, base_ks);
    }
}
template <int KS, int N>
__device__ __forceinline__ void tr_read_frags(bf16x8 (&f)[N], const unsigned (&base_sb)[N]) {
#pragma unroll
    for (int sb = 0; sb < N; ++sb) f[sb] = read_frag_tr_asm<KS>(base_sb[sb]);
}

// One K tile (64 k) of MFMAs for a wave tile of MI x NI 16 x 16 blocks. ALL fragment reads are inline asm, in
// k-step order, and the waits are explicit: hipcc waits lgkmcnt(0) before the first MFMA whatever the order
// (scalar loads share the counter), i.e. for both k-steps' fragments. Here the first k-step's MFMAs start once
// ITS fragments have landed — a counted lgkmcnt: LDS reads return in order, and scalar loads that may also be
// outstanding only make a counted wait conservative — while the second k-step's reads are still in flight.
template <bool AK, bool BKM, int MI, int NI>
__device__ __forceinline__ void compute_tile_g(const char* sA, const char* sB, f32x4 (&acc)[MI][NI], int ar0, int bc0, int lane,
                                               unsigned long long* kst = nullptr) {   // kst: measurement build only (K-loop stamps)
    const int i16 = lane & 15, g = lane >> 4;
    bf16x8 af[2][MI], bf[2][NI];
    unsigned ka[2] = {0u, 0u}, kb[2] = {0u, 0u}, ta[MI], tb[NI];
    if constexpr (AK) { ka[0] = km_frag_base(sA, ar0, i16, g); ka[1] = ka[0] ^ 64u; }
    else {
#pragma unroll
        for (int sb = 0; sb < MI; ++sb) ta[sb] = tr_frag_base(sA, ar0 + sb * 16, i16, g);
    }
    if constexpr (BKM) { kb[0] = km_frag_base(sB, bc0, i16, g); kb[1] = kb[0] ^ 64u; }
    else {
#pragma unroll
        for (int sb = 0; sb < NI; ++sb) tb[sb] = tr_frag_base(sB, bc0 + sb * 16, i16, g);
    }
    if constexpr (BKM) km_read_frags<NI>(bf[0], kb[0]); else tr_read_frags<0, NI>(bf[0], tb);
    if constexpr (AK) km_read_frags<MI>(af[0], ka[0]); else tr_read_frags<0, MI>(af[0], ta);
    if constexpr (BKM) km_read_frags<NI>(bf[1], kb[1]); else tr_read_frags<1, NI>(bf[1], tb);
    if constexpr (AK) km_read_frags<MI>(af[1], ka[1]); else tr_read_frags<1, MI>(af[1], ta);
    constexpr int R1 = (BKM ? NI : 2 * NI) + (AK ? MI : 2 * MI);   // read instructions of the second k-step
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(R1 < 15 ? R1 : 15) : "memory");
    __builtin_amdgcn_sched_barrier(0);   // MFMAs do not touch memory: without the fence hipcc may hoist them above the wait
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[0][ni], af[0][mi], acc[mi][ni], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#ifdef NBCI_STAMPS
    if (kst) { kst[2] = clock64(); __builtin_amdgcn_sched_barrier(0); }   // (lgkmcnt is drained here anyway: s_memtime shares it with the LDS reads)
#endif
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[1][ni], af[1][mi], acc[mi][ni], 0, 0, 0);
}

}  // namespace nbci
