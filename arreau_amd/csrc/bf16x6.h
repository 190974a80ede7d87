// Split-precision (bf16x6) building blocks shared by edge_bf16.hip and node_bf16.hip.
//
// Every fp32 product a*b is evaluated as six bf16 products on v_mfma_f32_32x32x16_bf16 with fp32
// accumulation:  a = a1 + a2 + a3,  b = b1 + b2 + b3  (exact 8+8+8-bit truncation splits),
//     a*b ~= a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1        (dropped terms <= 3 * 2^-24 |ab|).
// Weights are split on the host (model.hip, pack_linear_bf16x3) and chunked per 32-row output tile;
// activations are split in registers: registers 8s..8s+7 of a 32x32 fp32 accumulator tile, truncated
// pairwise to bf16, are the B fragment of k-step s of the next layer.
#pragma once
#include "internal.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// GELU with the Abramowitz-Stegun 7.1.26 complementary error function (|error| <= 1.5e-7 on erf, i.e.
// <= 0.75e-7 |x| on GELU): one rcp + one exp2 + 5 fma, branch-free, about half the VALU work of the 1-ulp erf in
// internal.h.  This kernel runs one wave per SIMD, so VALU work is not hidden by a partner wave.
__device__ __forceinline__ float gelu_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f((x * x) * -0.72134752044448170368f);  // exp(-z^2), z^2 = x^2 / 2
    const float half_erfc = 0.5f * (p * t) * e;                                     // 0.5 * erfc(|z|)
    const float phi = x < 0.0f ? half_erfc : 1.0f - half_erfc;
    return x * phi;
}

// ---- bf16x3 planes of a 32x32 fp32 tile (B-operand form) ---------------------------------------------
struct Planes { u32x4 p[3][2]; };  // [plane][k-step s]: 8 bf16 per lane each

__device__ __forceinline__ unsigned pack_hi16(float hi, float lo) {
    // {hi[31:16], lo[31:16]}: two truncated bf16 in one dword (element 2i in the low half)
    return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
__device__ __forceinline__ float bf16_residual(float x) {
    return x - __uint_as_float(__float_as_uint(x) & 0xffff0000u);  // exact
}
__device__ __forceinline__ Planes split_tile(const f32x16& x) {
    Planes r;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            const float lo = x[8 * s + 2 * pp], hi = x[8 * s + 2 * pp + 1];
            r.p[0][s][pp] = pack_hi16(hi, lo);
            const float rlo = bf16_residual(lo), rhi = bf16_residual(hi);
            r.p[1][s][pp] = pack_hi16(rhi, rlo);
            const float slo = bf16_residual(rlo), shi = bf16_residual(rhi);
            r.p[2][s][pp] = pack_hi16(shi, slo);
        }
    return r;
}

__device__ __forceinline__ f32x16 mfma_bf16(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// one output tile: acc += W(chunk) . B   over NIN input tiles.  The three weight planes of k-step ks+1 are
// read from LDS while the six MFMAs of k-step ks (192 cycles) issue, and the order is pinned so that the
// LDS latency never sits between dependent MFMAs.
template <int NIN, int KS0, int KS1>
__device__ __forceinline__ void mma_range(f32x16& acc, const u32x4* __restrict__ buf, const Planes (&b)[NIN], int lane) {
    const u32x4* f = buf + lane;
    u32x4 c1 = f[(size_t)KS0 * 192], c2 = f[(size_t)KS0 * 192 + 64], c3 = f[(size_t)KS0 * 192 + 128];
#pragma unroll
    for (int ks = KS0; ks < KS1; ++ks) {
        u32x4 n1, n2, n3;
        if (ks + 1 < KS1) {
            n1 = f[(size_t)(ks + 1) * 192];
            n2 = f[(size_t)(ks + 1) * 192 + 64];
            n3 = f[(size_t)(ks + 1) * 192 + 128];
        }
        const int t = ks >> 1, s = ks & 1;
        acc = mfma_bf16(c3, b[t].p[0][s], acc);  // small terms first
        acc = mfma_bf16(c2, b[t].p[1][s], acc);
        acc = mfma_bf16(c1, b[t].p[2][s], acc);
        acc = mfma_bf16(c2, b[t].p[0][s], acc);
        acc = mfma_bf16(c1, b[t].p[1][s], acc);
        acc = mfma_bf16(c1, b[t].p[0][s], acc);
        if (ks + 1 < KS1) { c1 = n1; c2 = n2; c3 = n3; }
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);  // the first k-step's fragments
#pragma unroll
    for (int ks = KS0; ks < KS1; ++ks) {
        if (ks + 1 < KS1) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);  // DS reads of k-step ks+1 ...
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);                     // ... ahead of the MFMAs of k-step ks
    }
}

#define EB_MAX_FRAGS 48                    // fragments (1 KiB) in the largest chunk (in = 256: 8 t * 2 s * 3 planes)
#define EB_STAGE ((EB_MAX_FRAGS + 3) / 4)  // fragments a wave stages per chunk

// each wave fetches fragments f = 4 i + wave of the next chunk into registers ...
template <int NF>
__device__ __forceinline__ void stage_load(u32x4 (&st)[EB_STAGE], const u32x4* __restrict__ chunk, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < (NF + 3) / 4; ++i) {
        const int f = 4 * i + wave;
        if (NF % 4 == 0 || f < NF) st[i] = chunk[(size_t)f * 64 + lane];
    }
}
// ... and writes them to the idle LDS buffer once the current chunk's MFMAs are issued
template <int NF>
__device__ __forceinline__ void stage_store(const u32x4 (&st)[EB_STAGE], u32x4* __restrict__ buf, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < (NF + 3) / 4; ++i) {
        const int f = 4 * i + wave;
        if (NF % 4 == 0 || f < NF) buf[(size_t)f * 64 + lane] = st[i];
    }
}


// LDS-DMA staging: fragment f = 4 i + wave of the next chunk goes straight from L2 into the idle LDS buffer
// (global_load_lds_dwordx4: 1 KiB per wave-instruction, destination = wave-uniform base + lane * 16), no VGPR
// staging and no ds_write burst.  The data is ordered for the readers by the issuing wave's vmcnt(0) followed by
// the workgroup barrier (__syncthreads() emits both while a DMA is in flight).
template <int NF>
__device__ __forceinline__ void stage_dma(const u32x4* __restrict__ chunk, u32x4* __restrict__ buf, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < (NF + 3) / 4; ++i) {
        const int f = 4 * i + wave;
        if (NF % 4 == 0 || f < NF)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(chunk + (size_t)f * 64 + lane),
                (__attribute__((address_space(3))) void*)(buf + (size_t)f * 64), 16, 0, 0);
    }
}
