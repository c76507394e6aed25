// Which lane's E8M0 scale does v_mfma_scale_f32_16x16x128_f8f6f4 apply to which operand bytes? A = ones everywhere; B = ones only at
// positions (lane group g0, bytes J) of every row; the A scale is 2x in lane group gs only. D[m][n] = |J| * (scale applied to (g0, J)).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(float* C, int g0, int jlo, int jhi, int gs, int which) {
    const int l = threadIdx.x, g = l >> 4;
    unsigned char ab[32], bb[32];
    for (int j = 0; j < 32; ++j) { ab[j] = 0x38; bb[j] = (g == g0 && j >= jlo && j < jhi) ? 0x38 : 0; }   // 0x38 = 1.0 in e4m3
    v8i a, b;
    for (int r = 0; r < 8; ++r) { a[r] = ab[4*r] | (ab[4*r+1] << 8) | (ab[4*r+2] << 16) | (ab[4*r+3] << 24); b[r] = bb[4*r] | (bb[4*r+1] << 8) | (bb[4*r+2] << 16) | (bb[4*r+3] << 24); }
    const int s2 = (g == gs) ? 128 : 127;
    v4f c = {0, 0, 0, 0};
    if (which == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, s2, 0, 127);   // scale on the FIRST operand (all ones)
    else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b, a, c, 0, 0, 0, 127, 0, s2);              // scale on the SECOND operand (all ones)
    for (int r = 0; r < 4; ++r) C[l * 4 + r] = c[r];
}
int main() {
    float* dC; hipMalloc(&dC, 1024);
    for (int which = 0; which < 2; ++which) {
        printf("scaled operand = %s; table: rows (g0, J half), columns gs = 0..3 -> D[0][0] / |J|\n", which ? "second" : "first");
        for (int g0 = 0; g0 < 4; ++g0) for (int h = 0; h < 2; ++h) {
            printf("  positions g=%d bytes %2d..%2d :", g0, 16 * h, 16 * h + 15);
            for (int gs = 0; gs < 4; ++gs) {
                hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dC, g0, 16 * h, 16 * h + 16, gs, which);
                std::vector<float> C(256); hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
                printf(" %4.1f", C[0] / 16.f);
            }
            printf("\n");
        }
    }
    return 0;
}
