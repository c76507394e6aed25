#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
// e4m3 (OCP) encode of small integers / simple values on the host
static uint8_t enc_e4m3(float f) {
    if (f == 0.f) return 0;
    uint8_t s = f < 0 ? 0x80 : 0; f = fabsf(f);
    int e = (int)floorf(log2f(f)); float m = f / ldexpf(1.f, e) - 1.f;   // [0,1)
    int me = (int)lrintf(m * 8.f); if (me == 8) { me = 0; e++; }
    int be = e + 7; if (be <= 0) { return s; }
    return s | (uint8_t)(be << 3) | (uint8_t)me;
}
__global__ void k(const uint8_t* A, const uint8_t* B, float* C, int hyp, int sa, int sb) {
    const int l = threadIdx.x, i16 = l & 15, g = l >> 4;
    v8i a, b;
    uint8_t ab[32], bb[32];
    for (int j = 0; j < 32; ++j) {
        int kk = hyp == 0 ? 32 * g + j : (j < 16 ? 16 * g + j : 64 + 16 * g + (j - 16));
        ab[j] = A[i16 * 128 + kk];      // A[m = i16][k]
        bb[j] = B[i16 * 128 + kk];      // B[n = i16][k]  (stored n-major)
    }
    for (int r = 0; r < 8; ++r) { a[r] = ab[4*r] | (ab[4*r+1] << 8) | (ab[4*r+2] << 16) | (ab[4*r+3] << 24); b[r] = bb[4*r] | (bb[4*r+1] << 8) | (bb[4*r+2] << 16) | (bb[4*r+3] << 24); }
    v4f c = {0, 0, 0, 0};
    // (a, b, c, cbsz = A format (0 = fp8 e4m3), blgp = B format, opsel_a, scale_a, opsel_b, scale_b)
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    for (int r = 0; r < 4; ++r) C[l * 4 + r] = c[r];
}
int main() {
    std::vector<uint8_t> A(16 * 128), B(16 * 128);
    std::vector<float> Af(16 * 128), Bf(16 * 128);
    for (int m = 0; m < 16; ++m) for (int kk = 0; kk < 128; ++kk) { float v = (float)(((m * 7 + kk * 3) % 9) - 4); Af[m*128+kk] = v; A[m*128+kk] = enc_e4m3(v); }
    for (int n = 0; n < 16; ++n) for (int kk = 0; kk < 128; ++kk) { float v = (float)(((n * 5 + kk * 11 + n * n) % 7) - 3); Bf[n*128+kk] = v; B[n*128+kk] = enc_e4m3(v); }
    uint8_t *dA, *dB; float* dC;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 1024);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
    for (int hyp = 0; hyp < 2; ++hyp) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, hyp, 127, 127);
        std::vector<float> C(256); hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
        // try both C maps: (a) row = 4*(l>>4)+r, col = l&15 with D = A.B^T [m][n]; (b) transposed
        int bad_a = 0, bad_b = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
            int row = 4 * (l >> 4) + r, col = l & 15;
            float ra = 0, rb = 0;
            for (int kk = 0; kk < 128; ++kk) { ra += Af[row*128+kk] * Bf[col*128+kk]; rb += Af[col*128+kk] * Bf[row*128+kk]; }
            if (C[l*4+r] != ra) bad_a++;
            if (C[l*4+r] != rb) bad_b++;
        }
        printf("hyp %d: mismatches D[m=row][n=col] %d, D[m=col][n=row] %d  (C[0]=%g)\n", hyp, bad_a, bad_b, C[0]);
    }
    // scale semantics: scale_a = 128 (2^1) in byte 0 -> results x2 ?
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, 0, 128, 127);
    std::vector<float> C2(256); hipMemcpy(C2.data(), dC, 1024, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, 0, 127, 127);
    std::vector<float> C1(256); hipMemcpy(C1.data(), dC, 1024, hipMemcpyDeviceToHost);
    int ok2 = 0; for (int i = 0; i < 256; ++i) ok2 += (C2[i] == 2 * C1[i]);
    printf("scale_a=128 doubles: %d/256\n", ok2);
    return 0;
}
