// Minimal stand-alone check of what DESIGN.md section 8 describes: does a wave's integer compare / select chain compute a
// different result when its kernel shares CUs with another stream's matrix-heavy kernel?
//   victim     : every wave holds 12 keys per lane and runs the top-8 selection of neighbor_kernel (64-bit integer keys
//                or the double keys that replaced them) REPS times; any round whose result differs from the wave's first
//                round is counted (the inputs never change, so every difference is a mis-executed instruction)
//   aggressor  : 8-wave workgroups, 48 KiB of LDS, ~200 VGPRs: ds_read_b128 + v_mfma_f32_16x16x32_f16 in a loop
// Runs the victim alone, then together with the aggressor on a second stream, and prints the mismatch counts.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp/coresidency_repro.hip -o tools/exp/_bin/coresidency_repro
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int USE_F64>
__global__ __launch_bounds__(256) void victim_kernel(int reps, unsigned long long* __restrict__ mismatches,
                                                     unsigned long long* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const unsigned wave_id = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    constexpr int NK = 12, K = 8;
    unsigned long long ukeys[NK];
    double dkeys[NK];
#pragma unroll
    for (int q = 0; q < NK; ++q) {
        const unsigned h = hash32(wave_id * 7919u + lane * 131u + q);
        const unsigned d2 = 0x3d000000u + (h & 0x00ffffffu);  // a float bit pattern, like the d^2 keys
        const unsigned c = lane + 64 * q;
        const bool present = (h >> 28) != 0;                   // some candidates are out of range
        ukeys[q] = present ? (((unsigned long long)d2 << 32) | c) : ~0ull;
        dkeys[q] = present ? ((double)d2 * 2097152.0 + (double)c) : 1.0e300;
    }
    unsigned long long first_sig = 0, bad = 0;
    for (int rep = 0; rep < reps; ++rep) {
        unsigned long long sig = 0;
        if (USE_F64) {
            double last = -1.0;
            for (int s = 0; s < K; ++s) {
                double best = 1.0e300;
#pragma unroll
                for (int q = 0; q < NK; ++q) best = fmin(best, dkeys[q] > last ? dkeys[q] : 1.0e300);
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) best = fmin(best, __shfl_xor(best, off, 64));
                last = best;
                sig = sig * 1000003ull + (unsigned long long)best;
            }
        } else {
            unsigned long long last = 0ull;
            for (int s = 0; s < K; ++s) {
                unsigned long long best = ~0ull;
#pragma unroll
                for (int q = 0; q < NK; ++q)
                    if (ukeys[q] > last && ukeys[q] < best) best = ukeys[q];
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    const unsigned long long other = __shfl_xor(best, off, 64);
                    best = other < best ? other : best;
                }
                last = best;
                sig = sig * 1000003ull + best;
            }
        }
        if (rep == 0) first_sig = sig;
        else if (sig != first_sig) ++bad;
        // keep the loop from being collapsed: the keys depend (vacuously) on a value the compiler cannot see through
        asm volatile("" : "+v"(ukeys[0]), "+v"(dkeys[0]));
    }
    if (bad) atomicAdd(mismatches, bad);
    if (first_sig == 0x123456789abcdefull) sink[0] = first_sig;
}

__global__ __launch_bounds__(512) void aggressor_kernel(int iters, float* __restrict__ out) {
    __shared__ u32x4 lds[3072];  // 48 KiB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 3072; i += 512) lds[i] = u32x4{0x3c003c00u + (unsigned)i, 0x3c003c00u, 0x38003800u, 0x3c003c00u};
    __syncthreads();
    f32x4 acc[24];
#pragma unroll
    for (int a = 0; a < 24; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 b[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) b[j] = lds[(wave * 64 + lane + 64 * j) % 3072];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < 24; ++a) {
            const u32x4 w = lds[(it * 24 + a) * 64 % 3008 + lane];
            acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, b[a & 15]), acc[a], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 24; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    if (s == 12345.678f) out[0] = s;
}

template <int USE_F64>
static unsigned long long run(bool with_aggressor, int launches) {
    unsigned long long *d_bad, *d_sink, h_bad = 0;
    float* d_out;
    CK(hipMalloc(&d_bad, 8)); CK(hipMalloc(&d_sink, 8)); CK(hipMalloc(&d_out, 4));
    CK(hipMemset(d_bad, 0, 8));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    for (int i = 0; i < launches; ++i) {
        if (with_aggressor) hipLaunchKernelGGL(aggressor_kernel, dim3(256), dim3(512), 0, sb, 400, d_out);
        hipLaunchKernelGGL(victim_kernel<USE_F64>, dim3(512), dim3(256), 0, sa, 200, d_bad, d_sink);
        if ((i & 7) == 7) { CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb)); }
    }
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(&h_bad, d_bad, 8, hipMemcpyDeviceToHost));
    CK(hipStreamDestroy(sa)); CK(hipStreamDestroy(sb));
    CK(hipFree(d_bad)); CK(hipFree(d_sink)); CK(hipFree(d_out));
    return h_bad;
}

int main() {
    const int launches = 400;  // x 2048 waves x 200 repetitions of the 8-round selection each
    printf("u64 keys, victim alone           : %llu mismatching rounds\n", run<0>(false, launches));
    printf("u64 keys, with the aggressor     : %llu mismatching rounds\n", run<0>(true, launches));
    printf("double keys, victim alone        : %llu mismatching rounds\n", run<1>(false, launches));
    printf("double keys, with the aggressor  : %llu mismatching rounds\n", run<1>(true, launches));
    return 0;
}
