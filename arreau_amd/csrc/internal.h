// Internal declarations shared by the translation units of libarreau_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/arreau_hip.h"

#define ARREAU_NUM_ATTR 6      // inv1, inv2, dist, cos a, cos b, cos c   (transforms/invariants.py:85-88)
#define ARREAU_NUM_MONO 83     // distinct monomials of degree 1..3 in 6 variables (6 + 21 + 56)
#define ARREAU_MONO_PAD 96     // padded to 3 MFMA input tiles of 32
#define ARREAU_POLY_COLS 258   // 6 + 36 + 216 columns of PolynomialFeatures(3)  (nn/embedding.py:10-14)
#define ARREAU_ORI 16

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

void arreau_set_error(const std::string& msg);

#define ARREAU_CHECK_HIP(expr)                                                              \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            arreau_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));           \
            return ARREAU_EHIP;                                                             \
        }                                                                                   \
    } while (0)

#define ARREAU_REQUIRE(cond, msg)                                                           \
    do {                                                                                    \
        if (!(cond)) {                                                                      \
            arreau_set_error(std::string(msg));                                            \
            return ARREAU_EINVAL;                                                           \
        }                                                                                   \
    } while (0)

// Packed weights resident in HBM.  All pointers are device pointers into `blob`.
struct arreau_model {
    arreau_config cfg;
    int S, C, D, L, O, W, H, k, T;
    float* blob;
    size_t blob_floats;
    const float* ori;        // [O][3]
    const float* w1p;        // basis layer 1, folded onto 83 monomials, MFMA-packed  out C, in 96
    const float* b1;         // [C]
    const float* w2p;        // basis layer 2, MFMA-packed                            out D, in C
    const float* b2;         // [D]
    const float* wkp;        // [L] conv.kernel.weight, MFMA-packed                   out C, in D
    float* fk;               // [L][O(o)][O(p)][C] fiber kernels / O (written once by the precompute kernel)
    const float* conv_bias;  // [L][C]
    const float* ln_w;       // [L][C]
    const float* ln_b;       // [L][C]
    const float* m1p;        // [L] linear_1, MFMA-packed                             out H, in C
    const float* mb1;        // [L][H]
    const float* m2p;        // [L] linear_2, MFMA-packed                             out C, in H
    const float* mb2;        // [L][C]
    const float* ls;         // [L][C] layer_scale (ones when absent)
    const float* embT;       // [S+78][C] x_embedder.weight transposed
    const float* ro_wT;      // [L][C][S+4] read_out weight transposed
    const float* ro_b;       // [L][S+4]
    const float* t_emb_w;    // [32]
    const float* ve_sigmas;  // [T+1]
    const float* vp_alpha_bars;  // [T+1]
    const float* vp_betas;   // [T+1]
    const float* q1t;        // [T][S][S]
    const float* qmats;      // [T][S][S]
    // raw fiber-basis MLP (only used by the one-time precompute kernel)
    const float* fiber_w1;   // [C][3]
    const float* fiber_b1;   // [C]
    const float* fiber_w2;   // [D][C]
    const float* fiber_b2;   // [D]
    const float* fiber_wk;   // [L][C][D] conv.fiber_kernel.weight
};

// ---------------------------------------------------------------------------------------------
// MFMA building block.  All dense layers are evaluated "transposed":
//     Y^T[out, row] = W[out, in] . X^T[in, row]
// with v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).  A 32x32 output tile lives in 16
// accumulator registers per lane: lane (h = lane>>5, j = lane&31) holds, in register r,
//     Y^T[32u + rho(r,h)][row j],   rho(r,h) = (r&3) + 8*(r>>2) + 4*h.
// Because the reduction index of the NEXT layer is this tile's row index, the accumulator is
// directly the B operand of the next MFMA chain (register r of tile t feeds the k-step whose two
// k values are rho(r,0) and rho(r,1)): activations never leave registers between layers.
// The A operand (weights) is pre-packed on the host so one 16-byte load per lane serves 4 k-steps:
//     P[u][t][q][lane][m] = W[out = 32u + (lane&31)][in = 32t + 8q + 4*(lane>>5) + m]
// (u: out tile, t: in tile, q: 0..3, m: 0..3) -- 1 KiB contiguous per wave-load.
// ---------------------------------------------------------------------------------------------
#define ARREAU_PACK_TILE_FLOATS 1024  // floats per (u,t) tile: 4 q * 64 lanes * 4

__device__ __forceinline__ f32x16 arreau_mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// acc[u] += W(out tile u0+u, in tile t) * b   for u < NT;  wp points at P[u0][t]; u_stride in floats.
template <int NT>
__device__ __forceinline__ void arreau_gemm_intile(f32x16 (&acc)[NT], const float* __restrict__ wp,
                                                   int u_stride, const f32x16& b, int lane) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 a[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u)
            a[u] = *reinterpret_cast<const f32x4*>(wp + (size_t)u * u_stride + q * 256 + lane * 4);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int u = 0; u < NT; ++u) acc[u] = arreau_mfma(a[u][m], b[4 * q + m], acc[u]);
        }
    }
}

// Software-pipelined chain over NIN input tiles: the fragments of k-group g+1 are requested before the
// MFMAs of group g issue, so the L2 latency of the weight stream hides behind 4*NT MFMAs (64 cycles each).
// wp points at P[u0][0]; groups are contiguous (256 floats apart) because t*1024 + q*256 = g*256.
template <int NT, int NIN>
__device__ __forceinline__ void arreau_gemm_chain(f32x16 (&acc)[NT], const float* __restrict__ wp, int u_stride,
                                                  const f32x16 (&b)[NIN], int lane) {
    constexpr int G = NIN * 4;
    const float* base = wp + lane * 4;
    f32x4 cur[NT], nxt[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) cur[u] = *reinterpret_cast<const f32x4*>(base + (size_t)u * u_stride);
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (g + 1 < G) {
#pragma unroll
            for (int u = 0; u < NT; ++u)
                nxt[u] = *reinterpret_cast<const f32x4*>(base + (size_t)u * u_stride + (g + 1) * 256);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int u = 0; u < NT; ++u) acc[u] = arreau_mfma(cur[u][m], b[g >> 2][4 * (g & 3) + m], acc[u]);
        }
#pragma unroll
        for (int u = 0; u < NT; ++u) cur[u] = nxt[u];
    }
}

// Branch-free erf (two polynomial pieces, both evaluated, one selected): the library erff branches on
// |x|, which splits the fully unrolled MFMA chains into hundreds of basic blocks and wrecks register
// allocation.  Coefficients: N. Juffa's single-precision erf (max error < 1 ulp with an accurate exp);
// v_exp_f32 adds about 2e-7 relative on the exp term, far inside the 1e-5 parity budget.
__device__ __forceinline__ float arreau_erf(float a) {
    const float t = fabsf(a);
    const float s = a * a;
    float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = fmaf(r, s, u);
    r = fmaf(r, t, -1.06777877e-1f);
    r = fmaf(r, t, -6.34846687e-1f);
    r = fmaf(r, t, -1.28717512e-1f);
    r = fmaf(r, t, -t);
    r = 1.0f - __builtin_amdgcn_exp2f(r * 1.44269504088896341f);
    r = copysignf(r, a);
    float p = -5.96761703e-4f;
    p = fmaf(p, s, 4.99119423e-3f);
    p = fmaf(p, s, -2.67681349e-2f);
    p = fmaf(p, s, 1.12819925e-1f);
    p = fmaf(p, s, -3.76125336e-1f);
    p = fmaf(p, s, 1.28379166e-1f);
    p = fmaf(p, a, a);
    return t > 0.927734375f ? r : p;
}

// exact (erf) GELU, torch.nn.GELU() default
__device__ __forceinline__ float arreau_gelu(float x) {
    return 0.5f * x * (1.0f + arreau_erf(x * 0.70710678118654752440f));
}

template <int NT>
__device__ __forceinline__ void arreau_gelu_tiles(f32x16 (&acc)[NT]) {
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = arreau_gelu(acc[u][r]);
}

// initialise accumulator tiles with a bias vector: register r of tile u gets bias[32u + rho(r,h)]
template <int NT>
__device__ __forceinline__ void arreau_bias_tiles(f32x16 (&acc)[NT], const float* __restrict__ bias, int h) {
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v = *reinterpret_cast<const f32x4*>(bias + 32 * u + 8 * q + 4 * h);
            acc[u][4 * q + 0] = v[0];
            acc[u][4 * q + 1] = v[1];
            acc[u][4 * q + 2] = v[2];
            acc[u][4 * q + 3] = v[3];
        }
}

// launchers implemented in the other translation units ------------------------------------------
int arreau_launch_fiber_precompute(arreau_model* m, hipStream_t s);
int arreau_launch_neighbor(const float* cart, const float* lattice, const int32_t* offsets, int B, int N,
                           float radius, int k, int32_t* deg, int32_t* src, int32_t* cell, float* dir, float* dist,
                           hipStream_t s);
int arreau_launch_prep(const arreau_model* m, const float* frac, const float* lengths, const float* angles,
                       const int32_t* t, const int32_t* offsets, int B, int N, float* lattice, float* cart,
                       int32_t* batch, float* cvec, hipStream_t s);
int arreau_launch_edge(const arreau_model* m, const float* dir, const float* dist, const int32_t* deg,
                       const int32_t* batch, const float* lattice, int N, float* kbuf, hipStream_t s);
int arreau_launch_embed(const arreau_model* m, const float* frac, const int32_t* types, const float* lattice,
                        const int32_t* batch, const float* cvec, int N, float* x0, hipStream_t s);
int arreau_launch_node_layer(const arreau_model* m, int layer, const float* kbuf, const int32_t* deg,
                             const int32_t* src, const float* x_in, float* x_out, float* xbar, float* vsum,
                             int N, hipStream_t s);
int arreau_launch_readout(const arreau_model* m, const float* xbar, const float* vsum, const int32_t* offsets,
                          int B, int N, float* gs, float* eps, float* logits, float* len0, hipStream_t s);
