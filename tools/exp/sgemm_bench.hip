// Micro-benchmark + check of arreau_sgemm (csrc/sgemm.h) at the training step's shapes (64 crystals, 532 atoms):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iarreau_amd/csrc tools/exp/sgemm_bench.hip -o tools/exp/_bin/sgemm_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "sgemm.h"

static std::string g_err;
void arreau_set_error(const std::string& m) { g_err = m; }

__global__ void ref_kernel(int M, int N, int K, const float* A, long as0, long as1, const float* B, long bs0, long bs1, float* C) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)M * N) return;
    const int m = (int)(i / N), n = (int)(i % N);
    double acc = 0.0;
    for (int k = 0; k < K; ++k) acc += (double)A[m * as0 + k * as1] * (double)B[k * bs0 + n * bs1];
    C[i] = (float)acc;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Shape { const char* name; int M, N, K; int a_kmajor, b_nmajor; };

int main(int argc, char** argv) {
    const char* filter = argc > 1 ? argv[1] : nullptr;   // substring of the shape name
    const int mode_mask = argc > 2 ? atoi(argv[2]) : 7;  // bit m: run mode m
    const Shape shapes[] = {
        {"edge W2 fwd  [R,128]x[256,128]^T", 68096, 256, 128, 1, 0},
        {"kern fwd     [R,256]x[640,256]^T", 68096, 640, 256, 1, 0},
        {"edge W1 fwd  [R,96]x[128,96]^T  ", 68096, 128, 96, 1, 0},
        {"lin1 fwd     [M,128]x[512,128]^T", 8512, 512, 128, 1, 0},
        {"lin2 fwd     [M,512]x[128,512]^T", 8512, 128, 512, 1, 0},
        {"kern dx      [R,640]x[640,256]  ", 68096, 256, 640, 1, 1},
        {"lin1 dx      [M,512]x[512,128]  ", 8512, 128, 512, 1, 1},
        {"kern dW      [R,640]^Tx[R,256]  ", 640, 256, 68096, 0, 1},
        {"lin1 dW      [M,512]^Tx[M,128]  ", 512, 128, 8512, 0, 1},
        {"readout dW   [M,94]^Tx[M,128]   ", 94, 128, 8512, 0, 1},
        {"fiber        [256,256]x[128,256]^T", 256, 128, 256, 1, 0},
        {"edge W2 dx   [R,256]x[256,128]  ", 68096, 128, 256, 1, 1},
        {"edge W2 dW   [R,256]^Tx[R,128]  ", 256, 128, 68096, 0, 1},
        {"lin2 dx      [M,128]x[128,512]  ", 8512, 512, 128, 1, 1},
        {"lin2 dW      [M,128]^Tx[M,512]  ", 128, 512, 8512, 0, 1},
        {"ragged dW    [8501,500]^Tx[8501,132]", 500, 132, 8501, 0, 1},
        {"ragged dx    [8501,260]x[260,132]", 8501, 132, 260, 1, 1},
        {"sweep 128x128 tiles K=32", 8512, 512, 32, 1, 0},
        {"sweep 128x128 tiles K=128", 8512, 512, 128, 1, 0},
        {"sweep 128x128 tiles K=512", 8512, 512, 512, 1, 0},
        {"sweep 128x128 tiles K=2048", 8512, 512, 2048, 1, 0},
        {"sweep 128x128 tiles K=2048 nmajor", 8512, 512, 2048, 1, 1},
        {"sweep 128x128 tiles K=2048 mmajor", 8512, 512, 2048, 0, 1},
        {"sweep 128x128 4x rows K=128", 34048, 512, 128, 1, 0},
        {"sweep 128x128 4x rows K=2048", 34048, 512, 2048, 1, 0},
    };
    float* partial;
    CK(hipMalloc(&partial, ARREAU_SGEMM_PARTIAL_FLOATS * 4));
    hipStream_t s = nullptr;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const Shape& sh : shapes) {
        if (filter && !strstr(sh.name, filter)) continue;
        const size_t na = (size_t)sh.M * sh.K, nb = (size_t)sh.K * sh.N, nc = (size_t)sh.M * sh.N;
        std::vector<float> ha(na), hb(nb);
        srand(1);
        for (auto& v : ha) v = (float)rand() / RAND_MAX - 0.5f;
        for (auto& v : hb) v = (float)rand() / RAND_MAX - 0.5f;
        float *A, *B, *C, *R;
        CK(hipMalloc(&A, na * 4)); CK(hipMalloc(&B, nb * 4)); CK(hipMalloc(&C, nc * 4)); CK(hipMalloc(&R, nc * 4));
        CK(hipMemcpy(A, ha.data(), na * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(B, hb.data(), nb * 4, hipMemcpyHostToDevice));
        const long as0 = sh.a_kmajor ? sh.K : 1, as1 = sh.a_kmajor ? 1 : sh.M;
        const long bs0 = sh.b_nmajor ? sh.N : 1, bs1 = sh.b_nmajor ? 1 : sh.K;
        hipLaunchKernelGGL(ref_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, s, sh.M, sh.N, sh.K, A, as0, as1, B, bs0, bs1, R);
        CK(hipDeviceSynchronize());
        std::vector<float> hc(nc), hr(nc);
        CK(hipMemcpy(hr.data(), R, nc * 4, hipMemcpyDeviceToHost));
        double maxref = 0;
        for (size_t i = 0; i < nc; ++i) maxref = fmax(maxref, fabs((double)hr[i]));
        printf("%-36s M=%6d N=%4d K=%6d ", sh.name, sh.M, sh.N, sh.K);
        for (int mode = 0; mode < 3; ++mode) {  // 0 exact fp32 MFMA, 1 fp16x3, 2 bf16x6 (round 4)
            if (!((mode_mask >> mode) & 1)) continue;
            auto run = [&]() { return arreau_sgemm(s, partial, sh.M, sh.N, sh.K, A, as0, as1, B, bs0, bs1, C, sh.N, 1.f, 0.f, 1, 0, 0, 0, mode); };
            CK(hipMemset(C, 0, nc * 4));
            if (run()) { printf("launch failed: %s\n", g_err.c_str()); return 1; }
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(hc.data(), C, nc * 4, hipMemcpyDeviceToHost));
            double maxerr = 0;
            for (size_t i = 0; i < nc; ++i) maxerr = fmax(maxerr, fabs((double)hc[i] - hr[i]));
            const int reps = 20;
            for (int i = 0; i < 3; ++i) run();
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < reps; ++i) run();
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = 1e3 * ms / reps, tf = 2.0 * sh.M * sh.N * sh.K / (us * 1e-6) / 1e12;
            printf(" | %s %7.1f us %6.1f TF/s err %.1e", mode == 0 ? "fp32  " : mode == 1 ? "fp16x3" : "bf16x6", us, tf, maxerr / maxref);
        }
        printf("\n");
        CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(R));
    }
    return 0;
}
