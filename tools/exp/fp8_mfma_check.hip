// Facts the fp8 cross-product form of conv_proj.hip needs (round 4), measured on the GPU:
//  1. v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands and unit scales: D[i][j] = sum_k A[i][k] B[k][j] under the ASSUMED
//     operand map "lane l holds row / column l & 15, k = 32 (l >> 4) .. + 31 in its 8 registers, byte order = k order", and the
//     usual 16x16 C/D map (register r of lane l = row 4 (l >> 4) + r, column l & 15); exact small-integer operands.
//  2. What `scale` does in v_cvt_scalef32_pk_fp8_f16 (multiply or divide) and in v_cvt_scalef32_pk_f16_fp8.
//  3. Rate: cycles per instruction of the K = 128 fp8 form against v_mfma_f32_16x16x32_f16 (one wave per SIMD, back to back).
//   hipcc --offload-arch=gfx950 -O2 tools/exp/fp8_mfma_check.hip -o tools/exp/_bin/fp8_mfma_check
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));

static float e4m3_to_float(unsigned v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -x : x;
}

__global__ void mfma_kernel(const unsigned char* A /*[16][128]*/, const unsigned char* B /*[128][16]*/, float* D /*[16][16]*/) {
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    i32x8 a, b;
    for (int q = 0; q < 8; ++q) {
        unsigned wa = 0, wb = 0;
        for (int e = 0; e < 4; ++e) {
            const int k = 32 * g + 4 * q + e;
            wa |= (unsigned)A[r * 128 + k] << (8 * e);
            wb |= (unsigned)B[k * 16 + r] << (8 * e);
        }
        a[q] = (int)wa; b[q] = (int)wb;
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int q = 0; q < 4; ++q) D[(4 * g + q) * 16 + r] = c[q];
}

__global__ void cvt_kernel(float* out) {
    h2 a = {(_Float16)3.0f, (_Float16)0.75f};
    s2 r = {0, 0};
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, a, 16.0f, false);
    const unsigned u = __builtin_bit_cast(unsigned, r);
    out[0] = (float)(u & 0xff); out[1] = (float)((u >> 8) & 0xff);
    h2 b = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(0x3840u /* e4m3 1.0 (0x38), 2.0 (0x40) */, 16.0f, false);
    out[2] = (float)b[0]; out[3] = (float)b[1];
    s2 r2 = {0, 0};
    r2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r2, a, 1.0f, true);  // word_sel = true: which half of the dword?
    out[4] = (float)__builtin_bit_cast(unsigned, r2);
}

template <int MODE>
__global__ void rate_kernel(float* out, int iters) {
    i32x8 a8, b8;
    for (int q = 0; q < 8; ++q) { a8[q] = 0x38383838 + threadIdx.x; b8[q] = 0x30303030 + q; }
    f16x8 ah, bh;
    for (int q = 0; q < 8; ++q) { ah[q] = (_Float16)(0.5f + q); bh[q] = (_Float16)(0.25f * threadIdx.x); }
    f32x4 c[8];
    for (int q = 0; q < 8; ++q) c[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (MODE == 0) c[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c[q], 0, 0, 0);
            else c[q] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, c[q], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int q = 0; q < 8; ++q) s += c[q][0];
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (float)(t1 - t0) / (8.0f * iters); out[1] = s; }
}

int main() {
    unsigned char hA[16 * 128], hB[128 * 16];
    srand(3);
    // exact small values: e4m3 codes of {0, +-0.5, +-1, +-1.5, +-2, +-3}
    const unsigned char codes[] = {0x00, 0x30, 0xb0, 0x38, 0xb8, 0x3c, 0xbc, 0x40, 0xc0, 0x44, 0xc4};
    for (auto& v : hA) v = codes[rand() % 11];
    for (auto& v : hB) v = codes[rand() % 11];
    unsigned char *dA, *dB; float *dD, hD[256];
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, 1024);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double ref = 0;
            for (int k = 0; k < 128; ++k) ref += (double)e4m3_to_float(hA[i * 128 + k]) * e4m3_to_float(hB[k * 16 + j]);
            if (fabs(ref - hD[i * 16 + j]) > 1e-4) { if (bad < 5) printf("D[%d][%d] = %g, reference %g\n", i, j, hD[i * 16 + j], ref); ++bad; }
        }
    printf("fp8 K=128 MFMA under the assumed operand map: %d of 256 outputs wrong\n", bad);
    float hc[8];
    hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(1), 0, 0, dD);
    hipMemcpy(hc, dD, 32, hipMemcpyDeviceToHost);
    printf("cvt f16 (3.0, 0.75) -> fp8 with scale 16: codes %02x %02x = %g %g   (x16 would be 48, 12; /16: 0.1875, 0.046875)\n",
           (unsigned)hc[0], (unsigned)hc[1], e4m3_to_float((unsigned)hc[0]), e4m3_to_float((unsigned)hc[1]));
    printf("cvt fp8 (1.0, 2.0) -> f16 with scale 16: %g %g\n", hc[2], hc[3]);
    printf("cvt word_sel=true result dword: %08x\n", (unsigned)hc[4]);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(1024), dim3(256), 0, 0, dD, 2000);
            else hipLaunchKernelGGL(rate_kernel<1>, dim3(1024), dim3(256), 0, 0, dD, 2000);
        }
        hipMemcpy(hc, dD, 8, hipMemcpyDeviceToHost);
        printf("%s: %.1f shader clocks per instruction (one wave per SIMD, 8 independent accumulators, whole chip busy)\n",
               mode == 0 ? "v_mfma_f32_16x16x32_f16       " : "v_mfma_scale_f32_16x16x128 fp8", hc[0]);
    }
    return 0;
}
