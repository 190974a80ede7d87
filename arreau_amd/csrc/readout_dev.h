// Device code of the read-out on the fp32 matrix pipe, shared by readout_mfma_kernel (node.hip) and by the per-crystal tail of the
// sampling step (tail.hip).  Test infrastructure does not include this file.
#pragma once
#include "internal.h"

// Read-out on the fp32 matrix pipe (exact fp32 products, v_mfma_f32_32x32x2_f32): a tile of 32 atoms, wave l = layer l.
// Y_l^T[out, atom] = W_l[out, :] . xbar_l[atom, :] for the (S+4 padded to 96) outputs as three 32x32 tiles
// (weights streamed from their packed form, xbar rows loaded straight into the B-operand layout); the L partial
// tiles meet in LDS and are added in layer order (deterministic), biases included, then scaled by 1/L.
// The vector channel (column S) is not read from here (its per-orientation form comes from the MLP kernel).
// USPLIT = 1 (small launches): the workgroup computes only output tile `utile` -- a third of the serial matrix work per wave;
// every tile is computed and summed over the layers exactly as in the unsplit form (bit-identical).
// Called by every thread of the workgroup (one barrier inside); waves beyond the L-th only take part in the sums.  `part` =
// L * ROT * 64 * 16 floats of LDS; atoms n0 .. min(n0 + 32, N) - 1 are written.
template <int C, int ROT /* output tiles */, int USPLIT = 0>
__device__ __forceinline__ void arreau_readout_tile(
    float* part, const float* __restrict__ xbar /*[L][Ntot][C]*/, const float* __restrict__ vsum /*[N][16]*/,
    const float* __restrict__ ro_pack /*[L][ROT][C/32][1024]*/, const float* __restrict__ ro_b /*[L][S+4]*/,
    const float* __restrict__ ori, int S, int L, int Ntot /* batch size: strides xbar */, int n0, int N /* end of the atom range */,
    int utile /* USPLIT: the output tile */, float* __restrict__ eps, float* __restrict__ logits, float* __restrict__ gs /*[N][3]*/,
    int32_t* __restrict__ status) {
    constexpr int TC = C / 32;
    const int lane = threadIdx.x & 63, l = threadIdx.x >> 6;
    const int h = lane >> 5, j = lane & 31;
    const int RO = S + 4;
    const float invL = 1.0f / (float)L;
    if (l < L) {
        const int n = min(n0 + j, N - 1);  // padding atoms read a valid row and write nothing
        const float* rowp = xbar + ((size_t)l * Ntot + n) * C + 4 * h;
        f32x16 bx[TC][1];
#pragma unroll
        for (int t = 0; t < TC; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(rowp + 32 * t + 8 * q);
                bx[t][0][4 * q] = v[0]; bx[t][0][4 * q + 1] = v[1]; bx[t][0][4 * q + 2] = v[2]; bx[t][0][4 * q + 3] = v[3];
            }
        constexpr int G = 4 * TC;
        const float* region = ro_pack + (size_t)l * ROT * TC * ARREAU_PACK_TILE_FLOATS + lane * 4 +
                              (USPLIT ? (size_t)utile * G * 256 : 0);
        f32x4 ring[ARREAU_PF];
#pragma unroll
        for (int i = 0; i < ARREAU_PF; ++i) ring[i] = *reinterpret_cast<const f32x4*>(region + (size_t)i * 256);
#pragma unroll
        for (int ui = 0; ui < (USPLIT ? 1 : ROT); ++ui) {
            const int u = USPLIT ? utile : ui;
            f32x16 acc[1];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = 32 * u + (r & 3) + 8 * (r >> 2) + 4 * h;
                acc[0][r] = col < RO ? ro_b[l * RO + col] : 0.0f;
            }
            arreau_stream_tile<G, TC, 1>(acc, ring, region, ui * G, bx);
            float* dst = part + (((size_t)l * ROT + u) * 64 + lane) * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<f32x4*>(dst + 4 * q) = f32x4{acc[0][4 * q], acc[0][4 * q + 1], acc[0][4 * q + 2], acc[0][4 * q + 3]};
        }
    }
    __syncthreads();
    // ordered sum over the layers; thread -> (tile u, lane ln, register group q)
    bool bad = false;  // a non-finite output (an overflowed fp16 plane upstream, or non-finite inputs) sets the sticky flag
    const int i_beg = USPLIT ? utile * 256 : 0, i_end = USPLIT ? i_beg + 256 : ROT * 64 * 4;
    for (int i = i_beg + threadIdx.x; i < i_end; i += blockDim.x) {
        const int q = i & 3, ln = (i >> 2) & 63, u = i >> 8;
        f32x4 tot = {0.f, 0.f, 0.f, 0.f};
        for (int ll = 0; ll < L; ++ll) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(part + (((size_t)ll * ROT + u) * 64 + ln) * 16 + 4 * q);
            tot[0] += v[0]; tot[1] += v[1]; tot[2] += v[2]; tot[3] += v[3];
        }
        const int n = n0 + (ln & 31);
        if (n < N) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int col = 32 * u + m + 8 * q + 4 * (ln >> 5);
                const float v = tot[m] * invL;
                if (col < RO && col != S && !(fabsf(v) < INFINITY)) bad = true;
                if (col < S) logits[(size_t)n * S + col] = v;
                else if (col > S && col < RO) gs[(size_t)n * 3 + (col - S - 1)] = v;
            }
        }
    }
    // vector channel: eps component d of atom a (sphere_to_vec of the per-orientation dot products)
    if (threadIdx.x < 96 && (!USPLIT || utile == 0)) {
        const int a = threadIdx.x / 3, dd = threadIdx.x - 3 * a;
        const int ni = n0 + a;  // (32-bit bounds check: no per-lane 64-bit integer compares on this path)
        if (ni < N) {
            const size_t n = (size_t)ni;
            float acc = 0.f;
            for (int o = 0; o < 16; ++o) acc += (vsum[n * 16 + o] * invL) * ori[3 * o + dd];
            eps[n * 3 + dd] = acc * (1.0f / 16.0f);
            if (!(fabsf(acc) < INFINITY)) bad = true;
        }
    }
    if (bad) atomicOr(status, ARREAU_STATUS_NONFINITE);
}
