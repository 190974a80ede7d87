// Microbenchmark (GPU box): what do the two waves of a SIMD share?  One 512-thread workgroup per CU, waves 0-3 run
// role A and waves 4-7 (their SIMD partners) role B.  Roles: idle, a stream of v_mfma_f32_32x32x16_f16 (24 per
// repetition, two accumulators), or a stream of one kind of vector instruction (208 per repetition, 8 independent
// chains).  Printed: cycles per repetition of a role-A wave and of a role-B wave.
//     hipcc --offload-arch=gfx950 -O3 -o coexec tools/exp/coexec.hip && ./coexec
// Result on MI355X (profiles/r01e_coexec_microbench.txt): fp32 add / mul / fma (packed or not) do NOT co-execute with
// the partner's MFMA stream (their times add); conversions, med3, logic and transcendental ops do; a single wave
// issues a vector instruction about every 7.5 cycles, two waves together twice as many.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum Kind { PK_FMA, FMA, MUL, ADD, PK_MUL, XOR, MED3, CVT_PK_F16, CVT_F32_F16, RCP, EXP, NKIND };
static const char* kind_name[] = {"v_pk_fma_f32", "v_fma_f32", "v_mul_f32", "v_add_f32", "v_pk_mul_f32", "v_xor_b32",
                                  "v_med3_f32", "v_cvt_pk_f16_f32", "v_cvt_f32_f16", "v_rcp_f32", "v_exp_f32"};

template <int K>
__device__ __forceinline__ void valu_burst(f32x2 (&v)[8], int n) {
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if constexpr (K == PK_FMA) v[c] = __builtin_elementwise_fma(v[c], v[c], f32x2{0.5f, 0.25f});
            else if constexpr (K == FMA) v[c].x = fmaf(v[c].x, v[c].x, 0.5f);
            else if constexpr (K == MUL) v[c].x = v[c].x * 1.0001f;
            else if constexpr (K == ADD) v[c].x = v[c].x + 1.0001f;
            else if constexpr (K == PK_MUL) v[c] = v[c] * f32x2{1.0001f, 0.9999f};
            else if constexpr (K == XOR) v[c].x = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v[c].x) ^ 0x12345u);
            else if constexpr (K == MED3) v[c].x = __builtin_amdgcn_fmed3f(v[c].x, -6.0f, v[c].y);
            else if constexpr (K == CVT_PK_F16) v[c].x = __builtin_bit_cast(float, __builtin_convertvector(v[c], f16x2));
            else if constexpr (K == CVT_F32_F16)
                v[c].x = (float)__builtin_bit_cast(_Float16, (unsigned short)__builtin_bit_cast(unsigned, v[c].x));
            else if constexpr (K == RCP) v[c].x = __builtin_amdgcn_rcpf(v[c].x);
            else if constexpr (K == EXP) v[c].x = __builtin_amdgcn_exp2f(v[c].x);
        }
    }
}
// role: 0 idle, 1 MFMA stream, 2 vector stream
template <int K>
__global__ __launch_bounds__(512, 2) void bench(int roleA, int roleB, int reps, long long* out, float* sink) {
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? roleA : roleB;
    f32x16 a = {}, b = {};
    f16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(threadIdx.x * 0.001f + i); y[i] = (_Float16)(0.5f - i); }
    f32x2 v[8];
    for (int c = 0; c < 8; ++c) v[c] = f32x2{0.001f * threadIdx.x, 0.002f * c};
    __syncthreads();
    const long long t0 = clock64();
    if (role == 1) {
        for (int i = 0; i < 12 * reps; ++i) {
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a, 0, 0, 0);
            b = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, b, 0, 0, 0);
        }
    } else if (role == 2) {
        valu_burst<K>(v, 26 * reps);
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += a[i] + b[i];
    for (int c = 0; c < 8; ++c) s += v[c].x + v[c].y;
    if (s == 123.456f) sink[0] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
}
template <int K>
void run(long long* d, float* sink) {
    const int reps = 200;
    const int cases[][2] = {{2, 0}, {2, 2}, {1, 2}};
    double r[3][2];
    for (int c = 0; c < 3; ++c) {
        bench<K><<<256, 512>>>(cases[c][0], cases[c][1], reps, d, sink);
        long long h[8];
        (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
        r[c][0] = (double)h[0] / reps;
        r[c][1] = (double)h[4] / reps;
    }
    printf("%-18s alone %6.0f | two vector waves %6.0f %6.0f | beside an MFMA wave (%4.0f) %6.0f  -> %s\n", kind_name[K], r[0][0],
           r[1][0], r[1][1], r[2][0], r[2][1], r[2][1] > r[0][0] + 0.5 * r[2][0] ? "serialised with MFMA" : "co-executes");
    if constexpr (K + 1 < NKIND) run<K + 1>(d, sink);
}
int main() {
    long long* d;
    float* sink;
    (void)hipMalloc(&d, 64);
    (void)hipMalloc(&sink, 4);
    bench<0><<<256, 512>>>(1, 0, 200, d, sink);
    long long h[8];
    (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("MFMA stream alone: %.0f cycles per 24 MFMAs\n", (double)h[0] / 200);
    bench<0><<<256, 512>>>(1, 1, 200, d, sink);
    (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("two MFMA waves   : %.0f / %.0f cycles per 24 MFMAs\n", (double)h[0] / 200, (double)h[4] / 200);
    printf("cycles per 208 vector instructions:\n");
    run<0>(d, sink);
    return 0;
}
