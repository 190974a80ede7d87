// Microbenchmark (GPU box), second series after coexec.hip: (1) more instruction kinds beside a partner wave's MFMA
// stream (integer add / shift, v_ldexp_f32, v_max_f32, packed fp16 arithmetic, fp64 fma, v_mov), to see what the
// epilogues of the split-precision kernels could be moved to; (2) does v_mfma_f32_16x16x32_f16 honour fp16 SUBNORMAL
// inputs?  (If it does, the residual plane of the fp16x3 split could be kept unscaled and all three products could
// share one accumulator.)
//     hipcc --offload-arch=gfx950 -O3 -o coexec2 tools/exp/coexec2.hip && ./coexec2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum Kind { ADD_U32, LSHL, LDEXP, MAX_F32, AND_B32, PK_FMA_F16, PK_MUL_F16, FMA_F64, MOV, ADD3_U32, PERM, PK_ADD_F32, NKIND };
static const char* kind_name[] = {"v_add_u32", "v_lshlrev_b32", "v_ldexp_f32", "v_max_f32", "v_and_b32", "v_pk_fma_f16",
                                  "v_pk_mul_f16", "v_fma_f64", "v_mov_b32", "v_add3_u32", "v_perm_b32", "v_pk_add_f32"};

template <int K>
__device__ __forceinline__ void valu_burst(f32x2 (&v)[8], int n) {
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            unsigned u = __builtin_bit_cast(unsigned, v[c].x), w = __builtin_bit_cast(unsigned, v[c].y);
            if constexpr (K == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u) : "v"(w));
            else if constexpr (K == LSHL) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u));
            else if constexpr (K == LDEXP) asm volatile("v_ldexp_f32 %0, %0, 1" : "+v"(u));
            else if constexpr (K == MAX_F32) asm volatile("v_max_f32 %0, %0, %1" : "+v"(u) : "v"(w));
            else if constexpr (K == AND_B32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u) : "v"(w));
            else if constexpr (K == PK_FMA_F16) asm volatile("v_pk_fma_f16 %0, %0, %1, %0" : "+v"(u) : "v"(w));
            else if constexpr (K == PK_MUL_F16) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(u) : "v"(w));
            else if constexpr (K == MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(u) : "v"(w));
            else if constexpr (K == ADD3_U32) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(u) : "v"(w));
            else if constexpr (K == PERM) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(u) : "v"(w));
            if constexpr (K == FMA_F64) {
                double d = __builtin_bit_cast(double, v[c]);
                asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d));
                v[c] = __builtin_bit_cast(f32x2, d);
            } else if constexpr (K == PK_ADD_F32) {
                asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(v[c]));
            } else {
                v[c].x = __builtin_bit_cast(float, u);
            }
        }
    }
}
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

// ---- fp16 subnormal inputs of the matrix instruction ----------------------------------------------------------------
// A[i][k] = a0 for k == 0 else 0, B[k][j] = b0 for k == 0 else 0  ->  every D[i][j] = a0 * b0 (fp32, exact).
__global__ void denorm_kernel(const unsigned short* abits, const unsigned short* bbits, int n, float* out) {
    for (int t = 0; t < n; ++t) {
        f16x8 a = {}, b = {};
        if ((threadIdx.x >> 4) == 0) {  // lane group 0 holds k = 0..7
            a[0] = __builtin_bit_cast(_Float16, abits[t]);
            b[0] = __builtin_bit_cast(_Float16, bbits[t]);
        }
        f32x4v c = {0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
        if (threadIdx.x == 0) out[t] = c[0];
    }
}
// does v_cvt_pk_f16_f32 produce subnormals under the default float mode?
__global__ void cvt_kernel(const float* in, int n, unsigned short* out) {
    for (int t = threadIdx.x; t < n; t += 64) {
        f32x2 v = {in[t], in[t]};
        f16x2 h = __builtin_convertvector(v, f16x2);
        out[t] = __builtin_bit_cast(unsigned short, h[0]);
    }
}
static float half_to_float(unsigned short h) {
    const int e = (h >> 10) & 31, m = h & 1023;
    const float s = (h >> 15) ? -1.f : 1.f;
    if (e == 0) return s * ldexpf((float)m, -24);
    return s * ldexpf((float)(m | 1024), e - 25);
}

int main() {
    long long* d;
    float* sink;
    (void)hipMalloc(&d, 64);
    (void)hipMalloc(&sink, 4);
    printf("cycles per 208 vector instructions:\n");
    run<0>(d, sink);

    const unsigned short A[] = {0x0001, 0x0200, 0x03ff, 0x0001, 0x3c00, 0x0155, 0x8001, 0x0400};
    const unsigned short B[] = {0x3c00, 0x3c00, 0x4000, 0x7bff, 0x0001, 0x5640, 0x4500, 0x0001};
    const int n = sizeof(A) / sizeof(A[0]);
    unsigned short *da, *db;
    float* dout;
    (void)hipMalloc(&da, sizeof(A));
    (void)hipMalloc(&db, sizeof(B));
    (void)hipMalloc(&dout, n * sizeof(float));
    (void)hipMemcpy(da, A, sizeof(A), hipMemcpyHostToDevice);
    (void)hipMemcpy(db, B, sizeof(B), hipMemcpyHostToDevice);
    denorm_kernel<<<1, 64>>>(da, db, n, dout);
    float out[16];
    (void)hipMemcpy(out, dout, n * sizeof(float), hipMemcpyDeviceToHost);
    printf("v_mfma_f32_16x16x32_f16 with fp16 subnormal inputs (a * b expected / got):\n");
    int ok = 1;
    for (int t = 0; t < n; ++t) {
        const float e = half_to_float(A[t]) * half_to_float(B[t]);
        printf("  a=0x%04x b=0x%04x  expected % .9e  got % .9e  %s\n", A[t], B[t], e, out[t], e == out[t] ? "ok" : "DIFFERENT");
        ok &= e == out[t];
    }
    printf("fp16 subnormal inputs are %s by the matrix instruction\n", ok ? "HONOURED" : "NOT honoured (flushed)");

    const float F[] = {3.0e-5f, 1.0e-6f, 5.9604645e-8f, 2.0e-8f, -4.5e-6f, 6.1e-5f};
    const int nf = sizeof(F) / sizeof(F[0]);
    float* df;
    unsigned short* dh;
    (void)hipMalloc(&df, sizeof(F));
    (void)hipMalloc(&dh, nf * 2);
    (void)hipMemcpy(df, F, sizeof(F), hipMemcpyHostToDevice);
    cvt_kernel<<<1, 64>>>(df, nf, dh);
    unsigned short hh[16];
    (void)hipMemcpy(hh, dh, nf * 2, hipMemcpyDeviceToHost);
    printf("v_cvt_pk_f16_f32 of values below the fp16 normal range:\n");
    for (int t = 0; t < nf; ++t) printf("  % .7e -> 0x%04x = % .7e\n", F[t], hh[t], half_to_float(hh[t]));
    return 0;
}
