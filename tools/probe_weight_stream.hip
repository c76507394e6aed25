// Floor of an M-STATIONARY fused MLP half (LayerNorm -> up -> GELU -> down -> + residual with a row strip resident on its CU, DESIGN.md §8.1 /
// VERDICT r3 #7 (iv)): every workgroup has to pull BOTH weight matrices (2 x 1024 x 1024 bf16 = 4 MB) through its CU's LDS once per strip, whatever it
// computes. This probe does only that: G workgroups, each streaming `bytes` of the same weights (L2 / Infinity-Cache resident after the first
// touch) through a two-stage LDS ring by LDS-DMA, 16 B per lane, no fragment reads, no MFMA. Its time is a LOWER bound for any such kernel; the two
// launches it would replace take ~70 us in the step (36 + 35 us, 30 us of them K loop).
//   hipcc --offload-arch=gfx950 -O3 -o build/probe_weight_stream tools/probe_weight_stream.hip && build/probe_weight_stream
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef __attribute__((address_space(1))) void gvoid;
typedef __attribute__((address_space(3))) void lvoid;

// one "tile" = 256 threads x 16 B x 8 pieces = 32 KB (a 128 x 64 k B tile is 16 KB, a stage of the GEMM kernels 34 KB)
__global__ __launch_bounds__(256) void stream_kernel(const char* w, long long bytes, int pieces_in_flight) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const long long tile = 32768;
    const int ntiles = (int)(bytes / tile);
    for (int i = 0; i < ntiles; ++i) {
        char* stage = smem + (i & 1) * tile;
        const char* src = w + (long long)i * tile;
#pragma unroll
        for (int p = 0; p < 8; ++p)
            __builtin_amdgcn_global_load_lds((gvoid*)(src + (p * 4 + wave) * 1024 + lane * 16), (lvoid*)(stage + (p * 4 + wave) * 1024), 16, 0, 0);
        if (pieces_in_flight == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // one tile at a time
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                           // the previous tile has landed, this one is in flight
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int main() {
    const long long wbytes = 4ll << 20;   // up + down projection weights, bf16
    char* w;
    hipMalloc(&w, wbytes);
    hipMemset(w, 1, wbytes);
    hipFuncSetAttribute((const void*)stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int grids[] = {256, 286, 512, 572};
    for (int g : grids)
        for (int fl = 0; fl < 2; ++fl) {
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(stream_kernel, dim3(g), dim3(256), 65536, 0, w, wbytes, fl);
            hipEventRecord(a);
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(stream_kernel, dim3(g), dim3(256), 65536, 0, w, wbytes, fl);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            const double us = ms / 20 * 1e3;
            printf("workgroups %3d (%s tile in flight): %7.1f us per launch = %5.1f GB/s per workgroup, %6.2f TB/s chip-wide L2 -> LDS\n", g,
                   fl ? "next" : "no  ", us, wbytes / us / 1e3, (double)g * wbytes / us / 1e6);
        }
    printf("(M = 9152 rows: 286 strips of 32 rows, or 572 of 16; the two launches this would replace: ~70 us in the step)\n");
    return 0;
}
