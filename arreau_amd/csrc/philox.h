// Counter-based random numbers for the in-kernel noise of the sampling loop: Philox4x32-10 (Salmon et al., SC'11;
// checked against the Random123 known-answer vectors in tests/test_cabi_exports.py via arreau_philox_fill).
// One call per drawn element, keyed by what the number is FOR, never by which thread draws it:
//     counter = (element index, timestep, draw kind, 0),  key = the 64-bit seed
// so the noise of (seed, timestep, kind, element) is the same whatever the batch composition, launch geometry or
// replay mechanism (eager loop or hipGraph).  The three draws of a step (diffusion_helpers.py:193-197, :79; d3pm.py:206):
#pragma once
#include <stdint.h>

#define ARREAU_DRAW_Z_LATTICE 0u  // randn [B,3]
#define ARREAU_DRAW_Z_FRAC 1u     // randn [N,3]
#define ARREAU_DRAW_U_TYPES 2u    // rand  [N,S]

struct Philox4 { uint32_t x[4]; };

__host__ __device__ inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return Philox4{{c0, c1, c2, c3}};
}

// uniform in [0, 1) with 24 random bits (every value exactly representable; 1.0 cannot occur)
__host__ __device__ inline float philox_uniform(uint64_t seed, uint32_t timestep, uint32_t kind, uint32_t element) {
    const Philox4 r = philox4x32_10(element, timestep, kind, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
    return (float)(r.x[0] >> 8) * (1.0f / 16777216.0f);
}

#ifdef __HIPCC__
// standard normal by Box-Muller from two words of one Philox call: u1 in (0, 1], u2 in [0, 1)
__device__ inline float philox_normal(uint64_t seed, uint32_t timestep, uint32_t kind, uint32_t element) {
    const Philox4 r = philox4x32_10(element, timestep, kind, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float u1 = ((float)(r.x[0] >> 8) + 1.0f) * (1.0f / 16777216.0f);
    const float u2 = (float)(r.x[1] >> 8) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}
#endif
