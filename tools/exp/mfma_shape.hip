// Microbenchmark (GPU box): sustained wall-clock rate of v_mfma_f32_32x32x16_f16 vs v_mfma_f32_16x16x32_f16 streams with
// their A operands re-read from LDS (ds_read_b128), two waves per SIMD on every CU -- the regime of the projection loop.
//     hipcc --offload-arch=gfx950 -O3 -o mfma_shape tools/exp/mfma_shape.hip && ./mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void k(int iters, float* sink, long long* ticks) {
    __shared__ f16x8 a_s[32 * 64];  // 32 KiB of fragments
    for (int i = threadIdx.x; i < 32 * 64; i += 512) {
        f16x8 v;
        for (int j = 0; j < 8; ++j) v[j] = (_Float16)(0.001f * ((i * 8 + j) % 97) - 0.05f);
        a_s[i] = v;
    }
    f16x8 b[8];
    for (int q = 0; q < 8; ++q)
        for (int j = 0; j < 8; ++j) b[q][j] = (_Float16)(0.01f * ((threadIdx.x + 3 * q + j) % 53) - 0.2f);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const long long w0 = wall_clock64(), c0 = clock64();
    float s = 0.f;
    if (SHAPE == 32) {
        f32x16 acc0 = {}, acc1 = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int f = 0; f < 32; f += 2) {  // 32 fragments -> 48 MFMAs (3 per pair), as in the projection loop
                const f16x8 a1 = a_s[f * 64 + lane], a2 = a_s[(f + 1) * 64 + lane];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b[f & 7], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b[(f + 1) & 7], acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b[f & 7], acc1, 0, 0, 0);
            }
        }
        for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    } else {
        f32x4 m0 = {}, m1 = {}, x0 = {}, x1 = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int f = 0; f < 32; f += 2) {  // same bytes from LDS, 96 MFMAs of half the size (two column blocks share A)
                const f16x8 a1 = a_s[f * 64 + lane], a2 = a_s[(f + 1) * 64 + lane];
                m0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b[f & 7], m0, 0, 0, 0);
                m1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b[(f + 2) & 7], m1, 0, 0, 0);
                x0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b[(f + 1) & 7], x0, 0, 0, 0);
                x1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b[(f + 3) & 7], x1, 0, 0, 0);
                x0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b[f & 7], x0, 0, 0, 0);
                x1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b[(f + 2) & 7], x1, 0, 0, 0);
            }
        }
        for (int i = 0; i < 4; ++i) s += m0[i] + m1[i] + x0[i] + x1[i];
    }
    const long long w1 = wall_clock64(), c1 = clock64();
    if (s == 123.456f) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { ticks[0] = w1 - w0; ticks[1] = c1 - c0; }
}
template <int SHAPE>
void run(const char* name) {
    float* sink; long long* t;
    (void)hipMalloc(&sink, 4); (void)hipMalloc(&t, 16);
    const int iters = 2000;
    k<SHAPE><<<256, 512>>>(iters, sink, t);
    k<SHAPE><<<256, 512>>>(iters, sink, t);
    long long h[2];
    (void)hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    const double sec = h[0] / 100e6;
    const double flop = 256.0 * 8 * iters * 16 * 3 * 2.0 * 32 * 32 * 16;  // CUs x waves x iters x pairs x 3 products (32x32x16 equivalents)
    printf("%-10s wall %.3f ms, shader clock %.0f MHz, %.0f TFLOP/s (fp16 products), %.1f cycles per 32x32x16-equivalent MFMA per SIMD\n", name,
           sec * 1e3, h[1] / (double)h[0] * 100.0, flop / sec / 1e12, h[1] / (double)(iters * 48 * 2));
}
int main() { run<32>("32x32x16"); run<16>("16x16x32"); return 0; }
