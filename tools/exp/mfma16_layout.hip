// Check (GPU box) of the operand / accumulator register layout assumed for v_mfma_f32_16x16x32_f16:
//   A[16 x 32]: lane l holds A[l & 15][8 (l >> 4) .. 8 (l >> 4) + 7]
//   B[32 x 16]: lane l holds B[8 (l >> 4) .. + 7][l & 15]
//   D[16 x 16]: lane l, register r holds D[4 (l >> 4) + r][l & 15]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void k(const float* A, const float* B, float* D) {
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)A[i * 32 + 8 * g + e]; b[e] = (_Float16)B[(8 * g + e) * 16 + i]; }
    f32x4 d = {0, 0, 0, 0};
    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + i] = d[r];
}
int main() {
    float hA[16 * 32], hB[32 * 16], hD[256], ref[256];
    for (int i = 0; i < 512; ++i) { hA[i] = (float)((i * 7) % 13 - 6) / 8.0f; hB[i] = (float)((i * 5) % 11 - 5) / 4.0f; }
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { float s = 0; for (int kk = 0; kk < 32; ++kk) s += hA[m * 32 + kk] * hB[kk * 16 + n]; ref[m * 16 + n] = s; }
    float *dA, *dB, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    double e = 0; for (int i = 0; i < 256; ++i) e = fmax(e, fabs(hD[i] - ref[i]));
    printf("max |D - A.B| = %g  -> layout %s\n", e, e < 1e-3 ? "CONFIRMED" : "WRONG");
    return 0;
}
