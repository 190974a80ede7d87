// K6: the reverse updates of one denoising step (diffusion/diffusion_loss.py:338-347).
#include "update_dev.h"


// VP_lattice.reverse_given_x0 (diffusion_helpers.py:185-199) on lengths, then lattice_from_params.
// Note the reference adds `variance * z` (not sqrt(variance)) and zeroes z when t <= 1.
__device__ __forceinline__ void reverse_lattice_body(int gt /* global thread of the lattice part */, float* __restrict__ lengths, const float* __restrict__ angles,
                                       const int32_t* __restrict__ tstep, const int32_t* __restrict__ offsets,
                                       const float* __restrict__ len0, StepNoiseSrc noise,
                                       const float* __restrict__ alpha_bars, const float* __restrict__ betas, int B,
                                       int T, float* __restrict__ lattice, const float* __restrict__ fixed_lengths,
                                       int32_t* __restrict__ status, int b0,
                                       // sampling loop: pool the per-atom read-out here (same ordered sum as
                                       // readout_crystals_kernel) instead of a launch of its own; len0_out receives it
                                       const float* __restrict__ gs_atoms, float* __restrict__ len0_out) {
    // four lanes per crystal: lane i < 3 owns length component i (pooling, update), lane 0 then writes the cell
    const int b = b0 + (gt >> 2), i = gt & 3;
    const bool live = b < B;  // (whole groups of four are live or not; the shuffles below need every lane)
    const int bc = live ? b : B - 1;
    int t = tstep[bc];
    if (live && i == 0 && (t < 1 || t > T)) atomicOr(status, ARREAU_STATUS_BAD_TIMESTEP);  // clamped, but flagged
    t = t < 1 ? 1 : (t > T ? T : t);
    const int first = offsets[bc], last = offsets[bc + 1];
    float mylen = 0.f;
    if (live && i < 3) mylen = reverse_length_component(b, i, t, first, last, lengths, len0, noise, alpha_bars, betas, fixed_lengths, gs_atoms, len0_out);
    const int base = (threadIdx.x & 63) & ~3;
    float newlen[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) newlen[q] = __shfl(mylen, base + q, 64);
    if (!live || i != 0) return;
    // lattice_from_params (lattice_helpers.py:55-105)
    const float* ang = angles + 3 * b;
    const float ca = cosf(ang[0]), cb = cosf(ang[1]), cg = cosf(ang[2]);
    const float sa = sinf(ang[0]), sb = sinf(ang[1]);
    float val = (ca * cb - cg) / (sa * sb);
    val = fminf(fmaxf(val, -1.0f), 1.0f);
    const float gs = acosf(val);
    float* Lm = lattice + 9 * b;
    Lm[0] = newlen[0] * sb;             Lm[1] = 0.0f;                      Lm[2] = newlen[0] * cb;
    Lm[3] = -newlen[1] * sa * cosf(gs); Lm[4] = newlen[1] * sa * sinf(gs); Lm[5] = newlen[1] * ca;
    Lm[6] = 0.0f;                       Lm[7] = 0.0f;                      Lm[8] = newlen[2];
}

// ONE launch for the four updates of a step (round 3; they were two): the first `lat_blocks` workgroups (256 threads = 64
// crystals, four lanes each) run the lattice update, the others the atom update, one wave per atom.  The two parts touch
// disjoint data (lengths / lattice / pooled read-out against coordinates / types), so nothing orders them.
__global__ __launch_bounds__(256) void reverse_kernel(
    int lat_blocks, float* __restrict__ lengths, const float* __restrict__ angles, const int32_t* __restrict__ tstep,
    const int32_t* __restrict__ offsets, const float* __restrict__ len0, StepNoiseSrc noise, const float* __restrict__ alpha_bars,
    const float* __restrict__ betas, int B_lat, int T, float* __restrict__ lattice, const float* __restrict__ fixed_lengths,
    int32_t* __restrict__ status, int b0, const float* __restrict__ gs_atoms, float* __restrict__ len0_out,
    float* __restrict__ frac, int32_t* __restrict__ types, int B, int N, const float* __restrict__ eps,
    const float* __restrict__ logits, const float* __restrict__ ve_sigmas, const float* __restrict__ q1t,
    const float* __restrict__ qmats, int S, const int32_t* __restrict__ const_types, int absorbing, int n0,
    const int32_t* __restrict__ batch,
    // sampling loop: the lattice part is one workgroup per crystal, which also prepares the next step (reverse_crystal_block)
    float* __restrict__ lattice_ws, float* __restrict__ cvec_next, const float* __restrict__ t_emb_w, const float* __restrict__ embT, int C) {
    if ((int)blockIdx.x < lat_blocks && cvec_next != nullptr) {  // (kernel argument: uniform)
        reverse_crystal_block(b0 + (int)blockIdx.x, lengths, angles, tstep, offsets, len0, noise, alpha_bars, betas, T, lattice, fixed_lengths,
                              status, gs_atoms, len0_out, lattice_ws, cvec_next, t_emb_w, embT, S, C);
        return;
    }
    if ((int)blockIdx.x < lat_blocks) {
        reverse_lattice_body(blockIdx.x * blockDim.x + threadIdx.x, lengths, angles, tstep, offsets, len0, noise, alpha_bars, betas, B_lat, T,
                             lattice, fixed_lengths, status, b0, gs_atoms, len0_out);
        return;
    }
    reverse_atoms_body((int)blockIdx.x - lat_blocks, frac, types, tstep, offsets, B, N, eps, logits, noise, ve_sigmas, q1t, qmats, S, T,
                       const_types, absorbing, status, n0, batch);
}

int arreau_launch_reverse(const arreau_model* m, float* d_frac, int32_t* d_types, float* d_lengths, const float* d_angles,
                          const int32_t* d_t, const int32_t* d_off, int B, int N, const float* d_eps,
                          const float* d_logits, const float* d_len0, StepNoiseSrc noise, const int32_t* d_const_types,
                          float* d_lattice, hipStream_t s, const float* d_fixed_lengths, NodeRange r, const float* d_gs_atoms,
                          const int32_t* d_batch, float* d_lattice_ws, float* d_cvec_next) {
    const int n0 = r.n0, n1 = r.n1 < 0 ? N : r.n1, b0 = r.b0, b1 = r.b1 < 0 ? B : r.b1;
    const bool prep_next = d_cvec_next != nullptr;  // one workgroup per crystal, which also prepares the next step
    ARREAU_REQUIRE(!prep_next || d_lattice_ws != nullptr, "reverse update: the next step's set-up needs the workspace lattice");
    const int lat_blocks = b1 > b0 ? (prep_next ? b1 - b0 : (4 * (b1 - b0) + 255) / 256) : 0;
    const int atom_blocks = n1 > n0 ? (n1 - n0 + 3) / 4 : 0;
    if (lat_blocks + atom_blocks > 0) {
        ARREAU_LAUNCH(reverse_kernel, dim3(lat_blocks + atom_blocks), dim3(256), 0, s, lat_blocks, d_lengths, d_angles, d_t, d_off, d_len0,
                      noise, m->vp_alpha_bars, m->vp_betas, b1, m->T, d_lattice, d_fixed_lengths, m->status, b0, d_gs_atoms,
                      d_gs_atoms ? const_cast<float*>(d_len0) : nullptr, d_frac, d_types, B, n1, d_eps, d_logits, m->ve_sigmas, m->q1t,
                      m->qmats, m->S, d_const_types, m->qmats_absorbing, n0, d_batch, d_lattice_ws, d_cvec_next, m->t_emb_w, m->embT, m->C);
        ARREAU_CHECK_HIP(hipGetLastError());
    }
    return ARREAU_OK;
}

extern "C" int arreau_reverse_step(const arreau_model* m, float* d_frac, int32_t* d_types, float* d_lengths,
                                   const float* d_angles, const int32_t* d_t, const int32_t* d_off, int32_t B,
                                   int32_t N, const float* d_eps, const float* d_logits, const float* d_len0,
                                   const float* d_z_lattice, const float* d_z_frac, const float* d_u_types,
                                   float* d_lattice, void* stream) {
    ARREAU_REQUIRE(m && d_frac && d_types && d_lengths && d_angles && d_t && d_off && d_eps && d_logits && d_len0 &&
                       d_z_lattice && d_z_frac && d_u_types && d_lattice, "arreau_reverse_step: null pointer");
    ARREAU_REQUIRE(B >= 1 && N >= 0, "arreau_reverse_step: bad size");
    return arreau_launch_reverse(m, d_frac, d_types, d_lengths, d_angles, d_t, d_off, B, N, d_eps, d_logits, d_len0,
                                 StepNoiseSrc{d_z_lattice, d_z_frac, d_u_types, 0}, nullptr, d_lattice, (hipStream_t)stream);
}

// The sampler's in-kernel noise, written out: out[i] = the draw (seed, timestep, kind, element i) -- standard normal for
// kinds 0/1, uniform [0,1) for kind 2.  For tests (known-answer / statistics) and for reproducing a Philox trajectory
// through arreau_reverse_step.
__global__ void philox_fill_kernel(uint64_t seed, uint32_t timestep, uint32_t kind, int64_t n, float* __restrict__ out,
                                   uint32_t* __restrict__ raw) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (raw) {
        const Philox4 r = philox4x32_10((uint32_t)i, timestep, kind, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
        for (int j = 0; j < 4; ++j) raw[4 * i + j] = r.x[j];
    }
    if (out) out[i] = kind == ARREAU_DRAW_U_TYPES ? philox_uniform(seed, timestep, kind, (uint32_t)i)
                                                  : philox_normal(seed, timestep, kind, (uint32_t)i);
}

extern "C" int arreau_philox_fill(uint64_t seed, int32_t timestep, int32_t kind, int64_t n, float* d_out, uint32_t* d_raw,
                                  void* stream) {
    ARREAU_REQUIRE((d_out || d_raw) && n >= 0 && kind >= 0 && kind <= 2, "arreau_philox_fill: bad argument");
    if (n == 0) return ARREAU_OK;
    ARREAU_LAUNCH(philox_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seed,
                       (uint32_t)timestep, (uint32_t)kind, n, d_out, d_raw);
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}
