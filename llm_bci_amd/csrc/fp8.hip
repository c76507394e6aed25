// fp8.hip — MX-scaled fp8 (OCP e4m3) projections on the block-scaled matrix instruction v_mfma_scale_f32_16x16x128_f8f6f4
// (BASELINE configs[4]: PatchTST "fp8 MFMA QKV"). Replaces, in the forward pass, the q / k / v nn.Linear of HF PatchTSTAttention
// that the reference's encoder runs (models/patchtst.py:176,223-225 -> PatchTSTModel): qkv = y1 . Wqkv^T + b.
//
// Quantisation (OCP Microscaling, MXFP8 e4m3): every 32 consecutive elements along K share a power-of-two scale
//   X = 2^(floor(log2(amax)) - 8)      (8 = e4m3's largest exponent; amax = 0 -> 2^-127), stored as the E8M0 byte E + 127,
//   one step larger when amax / X would exceed 448 (the MX rule alone saturates block maxima in (1.75, 2) x 2^n: a 12 % error on
//   the largest element of the block; with the bump nothing saturates);
// elements are round-to-nearest-even e4m3 of v / X (the clamp to +-448 is then only a guard). The instruction applies the scales itself: each lane
// carries the E8M0 byte of ITS 32 values (row = lane & 15, K block = lane >> 4), products and sums are f32.
//   * activations: quantised by the BatchNorm apply pass that produces them (bn_apply in ptst_kernels.hip calls mx_quant_block),
//     so the GEMM reads 1 byte per element instead of 2 and no extra pass over the activations exists;
//   * weights: quantised per forward by mx_quantize_launch (768 x 256 elements: nothing).
// The backward pass keeps bf16 operands (straight-through: gradients are those of the unquantised projection).
// Fragment layout, measured (tools/probe_mfma_scale_fp8.hip: exact integer data; tools/probe_mfma_scale_lanes.hip: which lane's scale
// multiplies which bytes): lane (i16 = lane & 15, g = lane >> 4) supplies row i16 and, of the 128 k of a step, bytes 0..15 =
// k 16g .. 16g+15 and bytes 16..31 = k 64+16g .. 64+16g+15; the E8M0 byte in lane (i16, t) is applied to k-block t (k 32t .. 32t+31)
// of row i16 - so a lane's own 32 bytes belong to TWO blocks (g >> 1 and 2 + (g >> 1)), scaled by the bytes lanes (i16, g >> 1) and
// (i16, 2 + (g >> 1)) carry. With uniform scales any consistent k assignment gives the right sums (which is why a data-only probe
// cannot see this); with per-block scales only this one does. With the B fragment FIRST the result lane (i16, g) owns row m = i16,
// columns n = 4g .. 4g + 3.
#include "kernels.h"

namespace nbci {

typedef int v8i __attribute__((ext_vector_type(8)));

template <typename T> __device__ __forceinline__ float ldf8(const T* p, long long i);
template <> __device__ __forceinline__ float ldf8<float>(const float* p, long long i) { return p[i]; }
template <> __device__ __forceinline__ float ldf8<bf16_t>(const bf16_t* p, long long i) { return bf2f(p[i]); }
template <typename T> __device__ __forceinline__ void stf8(T* p, long long i, float v);
template <> __device__ __forceinline__ void stf8<float>(float* p, long long i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void stf8<bf16_t>(bf16_t* p, long long i, float v) { p[i] = f2bf(v); }
__device__ __forceinline__ void st4(float* p, const float (&v)[4]) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void st4(bf16_t* p, const float (&v)[4]) { bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])}; *(bf16x4*)p = o; }

static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NBCI_EHIP, std::string(what) + ": " + hipGetErrorString(e));
    return NBCI_OK;
}

// E8M0 byte of a block whose largest magnitude is amax
__device__ __forceinline__ unsigned mx_scale_byte(float amax) {
    const unsigned bits = __float_as_uint(amax);
    const int ex = (int)((bits >> 23) & 0xFF);          // biased exponent of amax; 0 for zero / subnormal
    if (ex == 0) return 0u;                              // 2^-127
    int E = ex - 127 - 8;
    if (E < -127) E = -127;
    if (amax * __uint_as_float((unsigned)(127 - E) << 23) > 448.f && E < 127) ++E;   // amax / 2^E in (448, 512): one step up, nothing saturates
    return (unsigned)(E + 127);
}
__device__ __forceinline__ float mx_inv_scale(unsigned sb) {   // 2^-(sb - 127), exact
    return __uint_as_float((unsigned)(254 - (int)sb) << 23);   // sb in [0, 254] -> exponent field 254 - sb (sb = 0: 2^127)
}
// two values -> two e4m3 bytes (round to nearest even, saturating at 448)
__device__ __forceinline__ unsigned mx_pack2(float a, float b) {
    a = fminf(fmaxf(a, -448.f), 448.f);
    b = fminf(fmaxf(b, -448.f), 448.f);
    return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false) & 0xFFFFu;
}

// rows x K (K % 32 == 0) of f32 / bf16 -> e4m3 codes (rows x K bytes) + E8M0 scales (rows x K/32 bytes). One thread per block of 32.
template <typename T>
__global__ __launch_bounds__(256) void mx_quantize_kernel(const T* __restrict__ x, long long ldx, uint8_t* __restrict__ q, uint8_t* __restrict__ sc,
                                                          long long rows, int K) {
    const int nb = K / 32;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * nb) return;
    const long long r = i / nb;
    const int b = (int)(i % nb);
    const T* p = x + r * ldx + 32 * b;
    float v[32];
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) { v[j] = ldf8<T>(p, j); amax = fmaxf(amax, fabsf(v[j])); }
    const unsigned sb = mx_scale_byte(amax);
    const float inv = mx_inv_scale(sb);
    unsigned* out = (unsigned*)(q + r * K + 32 * b);
#pragma unroll
    for (int j = 0; j < 8; ++j) out[j] = mx_pack2(v[4 * j] * inv, v[4 * j + 1] * inv) | (mx_pack2(v[4 * j + 2] * inv, v[4 * j + 3] * inv) << 16);
    sc[r * nb + b] = (uint8_t)sb;
}

int mx_quantize_launch(const void* x, int dtype, long long ldx, void* q, void* scales, long long rows, int K, hipStream_t s) {
    NBCI_REQUIRE(x && q && scales && rows > 0 && K > 0 && K % 32 == 0 && ldx >= K, NBCI_EINVAL, "mx_quantize: bad arguments (K must be a multiple of 32)");
    const long long n = rows * (K / 32);
    dim3 g((unsigned)((n + 255) / 256));
    if (dtype == NBCI_BF16) hipLaunchKernelGGL((mx_quantize_kernel<bf16_t>), g, dim3(256), 0, s, (const bf16_t*)x, ldx, (uint8_t*)q, (uint8_t*)scales, rows, K);
    else hipLaunchKernelGGL((mx_quantize_kernel<float>), g, dim3(256), 0, s, (const float*)x, ldx, (uint8_t*)q, (uint8_t*)scales, rows, K);
    return check_launch("mx_quantize");
}

// C[M][N] (bf16 or f32) = dequant(A8, sA)[M][K] . dequant(W8, sW)[N][K]^T + bias.  K = 128 NK (NK <= 4), f32 accumulate.
// One workgroup = 128 rows x ALL N columns: the A fragments of its rows (NK x 4 x 32 bytes per lane) are loaded ONCE into registers
// and stay there while the workgroup walks the column tiles (the projection weights, 768 x 256 bytes, come back from L2 every time:
// they are tiny); 4 waves = 2 x 2 wave tiles of 64 x 64 per column tile of 128. Fragments come straight from global memory: a
// wave's two load instructions per operand read the two 64-byte halves of 16 rows' 128-byte lines back to back; the scale byte of
// lane (i16, g) is block 4 ks + g of its row. The launch is bound by the C stores (N = 768 bf16 columns per 256 bytes of A): each
// wave tile goes through a padded LDS image once so that every store instruction writes whole 128-byte (bf16) rows.
constexpr int F8_LDT = 72;   // floats per LDS row of a wave's 64 x 64 f32 image (64 + 8: the 16 rows of a fragment spread over the banks)

template <typename TC, int NK>
__global__ __launch_bounds__(256) void gemm_fp8_kernel(const uint8_t* __restrict__ A, const uint8_t* __restrict__ sA, const uint8_t* __restrict__ W,
                                                       const uint8_t* __restrict__ sW, const float* __restrict__ bias, TC* __restrict__ C,
                                                       long long M, int N, long long ldc) {
    __shared__ __attribute__((aligned(16))) float simg[4 * 64 * F8_LDT];
    constexpr int K = 128 * NK, nb = 4 * NK;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, i16 = lane & 15, g = lane >> 4;
    const int wm = w >> 1, wn = w & 1;
    const long long m0 = (long long)blockIdx.x * 128 + wm * 64;
    float* img = simg + w * 64 * F8_LDT;
    v8i af[NK][4];
    int sa[NK][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        long long r = m0 + 16 * i + i16;
        if (r > M - 1) r = M - 1;                        // clamped rows only feed outputs that are never stored
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const uint4* pa = (const uint4*)(A + r * K + 128 * ks + 16 * g);   // k 16g .. +15 and 64 + 16g .. +15 of this step
            const uint4 a0 = pa[0], a1 = pa[4];
            af[ks][i] = (v8i){(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
            sa[ks][i] = sA[r * nb + 4 * ks + g];
        }
    }
    const int tiles_n = (N + 127) / 128;
    for (int tn = 0; tn < tiles_n; ++tn) {
        const int n0 = tn * 128 + wn * 64;
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            v8i bf[4];
            int sb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int n = n0 + 16 * i + i16;
                if (n > N - 1) n = N - 1;
                const uint4* pb = (const uint4*)(W + (long long)n * K + 128 * ks + 16 * g);
                const uint4 b0 = pb[0], b1 = pb[4];
                bf[i] = (v8i){(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
                sb[i] = sW[(long long)n * nb + 4 * ks + g];
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)   // B fragment first: lane (i16, g) of the result owns row m = i16, columns n = 4g .. 4g + 3
                    acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bf[ni], af[ks][mi], acc[mi][ni], 0, 0, 0, sb[ni], 0, sa[ks][mi]);
        }
        // wave-private LDS image (LDS executes a wave's operations in order: no barrier), then whole rows out
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                *(float4*)(img + (16 * mi + i16) * F8_LDT + 16 * ni + 4 * g) = make_float4(acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]);
        const int c8 = 8 * (lane & 7);                   // this lane's 8 consecutive columns of the wave tile
        const int n = n0 + c8;
        float bv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = (bias && n + e < N) ? bias[n + e] : 0.f;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int rl = 8 * it + (lane >> 3);
            const long long m = m0 + rl;
            const float4 x0 = *(const float4*)(img + rl * F8_LDT + c8), x1 = *(const float4*)(img + rl * F8_LDT + c8 + 4);
            const float v[8] = {x0.x + bv[0], x0.y + bv[1], x0.z + bv[2], x0.w + bv[3], x1.x + bv[4], x1.y + bv[5], x1.z + bv[6], x1.w + bv[7]};
            if (m >= M || n >= N) continue;
            TC* cp = C + m * ldc + n;
            if (n + 7 < N && (ldc & 7) == 0) {
                if constexpr (sizeof(TC) == 2) {
                    bf16x8 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3]), f2bf(v[4]), f2bf(v[5]), f2bf(v[6]), f2bf(v[7])};
                    *(bf16x8*)cp = o;
                } else {
                    *(float4*)cp = make_float4(v[0], v[1], v[2], v[3]);
                    *(float4*)(cp + 4) = make_float4(v[4], v[5], v[6], v[7]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (n + e < N) stf8<TC>(cp, e, v[e]);
            }
        }
    }
}

template <typename TC>
static void gemm_fp8_dispatch(int nk, dim3 grid, hipStream_t s, const uint8_t* A, const uint8_t* sA, const uint8_t* W, const uint8_t* sW, const float* bias,
                              TC* C, long long M, int N, long long ldc) {
    switch (nk) {
        case 1: hipLaunchKernelGGL((gemm_fp8_kernel<TC, 1>), grid, dim3(256), 0, s, A, sA, W, sW, bias, C, M, N, ldc); break;
        case 2: hipLaunchKernelGGL((gemm_fp8_kernel<TC, 2>), grid, dim3(256), 0, s, A, sA, W, sW, bias, C, M, N, ldc); break;
        case 3: hipLaunchKernelGGL((gemm_fp8_kernel<TC, 3>), grid, dim3(256), 0, s, A, sA, W, sW, bias, C, M, N, ldc); break;
        default: hipLaunchKernelGGL((gemm_fp8_kernel<TC, 4>), grid, dim3(256), 0, s, A, sA, W, sW, bias, C, M, N, ldc); break;
    }
}

int gemm_fp8_launch(const void* A8, const void* sA, const void* W8, const void* sW, const float* bias, void* C, int c_dtype, long long M, int N, int K,
                    long long ldc, hipStream_t s) {
    NBCI_REQUIRE(A8 && sA && W8 && sW && C && M > 0 && N > 0 && K > 0, NBCI_EINVAL, "gemm_fp8: null operand / bad shape");
    NBCI_REQUIRE(K % 128 == 0 && K <= 512 && ldc >= N, NBCI_ESHAPE, "gemm_fp8: K must be 128, 256, 384 or 512 (the A fragments of a row block stay in registers)");
    NBCI_REQUIRE(((uintptr_t)A8) % 16 == 0 && ((uintptr_t)W8) % 16 == 0, NBCI_EALIGN, "gemm_fp8: operands must be 16-byte aligned");
    const long long tiles = (M + 127) / 128;
    NBCI_REQUIRE(tiles < (1ll << 31), NBCI_ESHAPE, "gemm_fp8: too many row tiles");
    ProfScope ps("gemm_fp8_kernel", 2.0 * M * N * K, (double)M * K * 33.0 / 32.0 + (double)N * K * 33.0 / 32.0 + (double)M * N * (c_dtype == NBCI_BF16 ? 2 : 4), s);
    dim3 grid((unsigned)tiles);
    if (c_dtype == NBCI_BF16)
        gemm_fp8_dispatch<bf16_t>(K / 128, grid, s, (const uint8_t*)A8, (const uint8_t*)sA, (const uint8_t*)W8, (const uint8_t*)sW, bias, (bf16_t*)C, M, N, ldc);
    else
        gemm_fp8_dispatch<float>(K / 128, grid, s, (const uint8_t*)A8, (const uint8_t*)sA, (const uint8_t*)W8, (const uint8_t*)sW, bias, (float*)C, M, N, ldc);
    return check_launch("gemm_fp8");
}

}  // namespace nbci
