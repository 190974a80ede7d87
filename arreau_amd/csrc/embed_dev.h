// The embedding of the sampler's node features, shared by embed_kernel (node.hip) and the fused neighbour-list + embedding
// launch (graph.hip).  Test infrastructure does not include this file.
#pragma once
#include "internal.h"

// x0[n][o][c] = embT[type_n][c] + cvec[b][c] + sum_v embT[S+74+v][c] * (vec[n][v] . ori[o])
// with vec[n] = (frac_n, lattice rows a, b, c)  (diffusion_loss.py:158; to_from_sphere.py:4-5).
__device__ __forceinline__ void arreau_embed_body(
    unsigned idx /* (atom - n0) * C/4 + float4 column */, const float* __restrict__ frac, const int32_t* __restrict__ types,
    const float* __restrict__ lattice, const int32_t* __restrict__ batch, const float* __restrict__ cvec,
    const float* __restrict__ ori, const float* __restrict__ embT, int S, int C, int n0, int N /* atoms n0 .. N-1 */,
    float* __restrict__ x0, int32_t* __restrict__ status) {
    // thread = (atom n, float4 column c4): the atom's scalar part and the four vector-channel weight rows are
    // loaded once and serve all 16 orientations (16 coalesced 512-byte row stores per 32 lanes)
    // (32-bit index arithmetic: the launcher checks (N - n0) * C / 4 < 2^31.  Per-lane 64-bit compares are avoided in the
    // kernels that may share a CU with another stream's kernels: DESIGN.md section 8)
    const int C4 = C / 4;
    if (idx >= (unsigned)(N - n0) * (unsigned)C4) return;
    const int c4 = (int)(idx % (unsigned)C4);
    const int n = n0 + (int)(idx / (unsigned)C4);
    const int b = batch[n];
    int ty = types[n];
    if ((ty < 0 || ty >= S) && c4 == 0) atomicOr(status, ARREAU_STATUS_BAD_TYPE);  // clamped, but flagged
    ty = ty < 0 ? 0 : (ty >= S ? S - 1 : ty);
    const float* Lm = lattice + 9 * (size_t)b;
    float vec[4][3];
#pragma unroll
    for (int d = 0; d < 3; ++d) vec[0][d] = frac[3 * (size_t)n + d];
#pragma unroll
    for (int v = 0; v < 3; ++v)
#pragma unroll
        for (int d = 0; d < 3; ++d) vec[1 + v][d] = Lm[3 * v + d];
    const f32x4* e4 = reinterpret_cast<const f32x4*>(embT);
    const f32x4 base = e4[(size_t)ty * C4 + c4] + reinterpret_cast<const f32x4*>(cvec)[(size_t)b * C4 + c4];
    f32x4 ev[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) ev[v] = e4[(size_t)(S + 74 + v) * C4 + c4];
    f32x4* out = reinterpret_cast<f32x4*>(x0) + (size_t)n * ARREAU_ORI * C4 + c4;
#pragma unroll
    for (int o = 0; o < ARREAU_ORI; ++o) {
        const float ox = ori[3 * o], oy = ori[3 * o + 1], oz = ori[3 * o + 2];
        f32x4 acc = base;
#pragma unroll
        for (int v = 0; v < 4; ++v) acc += ev[v] * ((vec[v][0] * ox + vec[v][1] * oy) + vec[v][2] * oz);
        out[(size_t)o * C4] = acc;
    }
}

