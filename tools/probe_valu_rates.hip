// Issue cost (cycles per wave-instruction on one SIMD, one wave per SIMD, independent chains) of the integer / select / transcendental ops the
// softmax + dropout-hash inner loops are made of, on gfx950. The guide's table has the f32 ops only.
//   hipcc --offload-arch=gfx950 -O3 -o build/probe_valu_rates tools/probe_valu_rates.hip && build/probe_valu_rates
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define REP16(X) X X X X X X X X X X X X X X X X
#define OPS 256   // 16 x 16 instructions per loop iteration over 8 independent registers

template <int OP>
__global__ void probe(unsigned* out, unsigned long long* cyc, int iters, unsigned seed) {
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;
    const unsigned c = 0x7feb352du | seed;
    unsigned long long pka0 = a0, pka1 = a1, pka2 = a2, pka3 = a3, pka4 = a4, pka5 = a5, pka6 = a6, pka7 = a7, pkc = c;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#define ONE(R)                                                                                                   \
    if (OP == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(R) : "v"(c));                                       \
    if (OP == 1) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(R) : "v"(c));                                      \
    if (OP == 2) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(R) : "v"(c));                                          \
    if (OP == 3) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(R));                                                \
    if (OP == 4) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(R) : "v"(c));                                       \
    if (OP == 5) asm volatile("v_exp_f32 %0, %0" : "+v"(R));                                                       \
    if (OP == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(R) : "v"(c));                                 \
    if (OP == 7) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(R) : "v"(c));                                  \
    if (OP == 8) asm volatile("v_add_u32 %0, %0, %1" : "+v"(R) : "v"(c));                                          \
    if (OP == 9) asm volatile("v_cmp_le_u32 vcc, %0, %1" ::"v"(R), "v"(c) : "vcc");                                \
    if (OP == 10) asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(R) : "v"(c));                                     \
    if (OP == 11) asm volatile("v_alignbit_b32 %0, %0, %0, 13" : "+v"(R));                                         \
    if (OP == 12) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(R) : "v"(c));                                         \
    if (OP == 13) asm volatile("v_bfe_u32 %0, %0, 16, 16" : "+v"(R));                                              \
    if (OP == 14) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(R) : "v"(c));                                 \
    if (OP == 15) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(R) : "v"(c) : "s20", "s21");        \
    if (OP == 16) asm volatile("v_cmp_le_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(R) : "v"(c) : "vcc");   \
    if (OP == 17) asm volatile("v_sub_u32 %0, %0, %1\n\tv_ashrrev_i32 %0, 31, %0\n\tv_bfi_b32 %0, %0, 0, %1" : "+v"(R) : "v"(c));   \
    if (OP == 18) asm volatile("v_cmp_le_u32 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(R) : "v"(c) : "s20", "s21");   \
    if (OP == 19) asm volatile("v_bfi_b32 %0, %0, %1, %1" : "+v"(R) : "v"(c));                                     \
    if (OP == 20) asm volatile("v_max_f32 %0, %0, %1" : "+v"(R) : "v"(c));                                         \
    if (OP == 21) asm volatile("v_cmp_gt_f32 vcc, %0, %1" ::"v"(R), "v"(c) : "vcc");                               \
    if (OP == 22) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(R) : "v"(c));                                    \
    if (OP == 23) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(R) : "v"(c));                                     \
    if (OP == 24) asm volatile("v_mov_b32 %0, %1" : "+v"(R) : "v"(c));                                             \
    if (OP == 25) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(unsigned long long*)&pk##R) : "v"(pkc));
#define EIGHT ONE(a0) ONE(a1) ONE(a2) ONE(a3) ONE(a4) ONE(a5) ONE(a6) ONE(a7)
        REP16(EIGHT EIGHT)
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (unsigned)(pka0 ^ pka1 ^ pka2 ^ pka3 ^ pka4 ^ pka5 ^ pka6 ^ pka7);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
static void run(const char* name) {
    const int iters = 2000, blocks = 256, threads = 256;   // one wave per SIMD on every CU
    unsigned* out; unsigned long long* cyc;
    hipMalloc(&out, blocks * threads * 4); hipMalloc(&cyc, blocks * 8);
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 10, 1u);
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1u);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    unsigned long long s = 0;
    for (auto v : h) s += v;
    // s_memtime ticks at 100 MHz on this part? the guide: "tick = shader cycle" for s_memtime
    printf("%-18s %.2f cycles per wave-instruction (one wave per SIMD)\n", name, (double)s / blocks / ((double)iters * OPS));
    hipFree(out); hipFree(cyc);
}

int main() {
    run<8>("v_add_u32"); run<2>("v_xor_b32"); run<3>("v_lshrrev_b32"); run<10>("v_xad_u32"); run<11>("v_alignbit_b32"); run<13>("v_bfe_u32");
    run<0>("v_mul_lo_u32"); run<4>("v_mul_hi_u32"); run<1>("v_mul_u32_u24"); run<7>("v_mad_u32_u24");
    run<6>("v_cndmask_b32 vcc"); run<15>("v_cndmask e64 sgpr"); run<9>("v_cmp_le_u32 vcc"); run<21>("v_cmp_gt_f32 vcc");
    run<16>("cmp+cndmask vcc (pair)"); run<18>("cmp+cndmask sgpr (pair)"); run<17>("sub+ashr+bfi (triple)"); run<19>("v_bfi_b32"); run<20>("v_max_f32");
    run<22>("v_med3_f32"); run<23>("v_fma_f32"); run<24>("v_mov_b32"); run<25>("v_pk_mul_f32"); run<12>("v_mul_f32"); run<14>("v_cvt_pk_bf16_f32"); run<5>("v_exp_f32");
    return 0;
}
