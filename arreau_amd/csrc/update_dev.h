// Device code of the reverse updates of one denoising step (K6), shared by the stand-alone kernel of update.hip and by the
// per-crystal tail of the sampling step (tail.hip).  Test infrastructure does not include this file.
#pragma once
#include "internal.h"
#include "philox.h"
#include "prep_dev.h"

#define D3PM_EPS 1e-6f  // d3pm.py:23

// One component of the length update of crystal b at timestep t (VP_lattice.reverse_given_x0; the per-atom read-out is pooled
// here when gs_atoms is given: the ordered sum of readout_crystals_kernel).  Writes lengths[3 b + i] and returns it.
__device__ __forceinline__ float reverse_length_component(int b, int i, int t, int first, int last, float* __restrict__ lengths,
                                                          const float* __restrict__ len0, StepNoiseSrc noise,
                                                          const float* __restrict__ alpha_bars, const float* __restrict__ betas,
                                                          const float* __restrict__ fixed_lengths, const float* __restrict__ gs_atoms,
                                                          float* __restrict__ len0_out) {
    const float* __restrict__ z = noise.z_lattice;
    const float n = (float)(last - first);
    const float ab_t = alpha_bars[t], ab_p = alpha_bars[t - 1], beta = betas[t];
    const float denom = 1.0f - ab_t;
    const float alpha_t = 1.0f - beta;
    const float c0 = sqrtf(ab_p) * beta;
    const float c1 = sqrtf(alpha_t) * (1.0f - ab_p);
    const float variance = (1.0f - ab_p) * beta / denom;
    float pooled;
    if (gs_atoms != nullptr) {
        pooled = 0.f;  // the ordered sum of readout_crystals_kernel, four loads in flight at a time
        int a = first;
        for (; a + 3 < last; a += 4) {
            const float v0 = gs_atoms[(size_t)a * 3 + i], v1 = gs_atoms[(size_t)(a + 1) * 3 + i];
            const float v2 = gs_atoms[(size_t)(a + 2) * 3 + i], v3 = gs_atoms[(size_t)(a + 3) * 3 + i];
            pooled = (((pooled + v0) + v1) + v2) + v3;
        }
        for (; a < last; ++a) pooled += gs_atoms[(size_t)a * 3 + i];
        len0_out[3 * b + i] = pooled;
    } else {
        pooled = len0[3 * b + i];
    }
    const float x0 = pooled * n;  // pred_lengths_0 * num_atoms (diffusion_loss.py:338)
    const float xt = lengths[3 * b + i];
    const float mean = (c0 * x0 + c1 * xt) / denom;
    const float zdraw = z ? z[3 * b + i] : philox_normal(noise.seed, (uint32_t)t, ARREAU_DRAW_Z_LATTICE, 3u * b + i);
    const float zz = t > 1 ? zdraw : 0.0f;
    // fixed-cell sampling (arreau_sample_loop, d_fixed_lengths): the given lengths are re-imposed after the update
    const float mylen = fixed_lengths ? fixed_lengths[3 * b + i] : mean + variance * zz;
    lengths[3 * b + i] = mylen;
    return mylen;
}

// torch.remainder(x, 1) for floats: fmod, then shift negatives up by one (can return exactly 1.0f
// for tiny negative x, like the reference's `% 1`).
__device__ __forceinline__ float remainder_one(float x) {
    float m = fmodf(x, 1.0f);
    if (m != 0.0f && m < 0.0f) m += 1.0f;
    return m;
}

// One wave per atom: VE_pbc.reverse on the fractional coordinates (diffusion_helpers.py:65-81) and
// D3PM.reverse on the atom type (d3pm.py:74-110, 198-215).  Lanes span the S classes (2 per lane).
__device__ __forceinline__ void reverse_one_atom(
    int i /* atom (wave-uniform) */, int lane, float* __restrict__ frac, int32_t* __restrict__ types, const int32_t* __restrict__ tstep,
    const int32_t* __restrict__ offsets, int B, const float* __restrict__ eps,
    const float* __restrict__ logits, StepNoiseSrc noise,
    const float* __restrict__ ve_sigmas, const float* __restrict__ q1t, const float* __restrict__ qmats, int S,
    int T, const int32_t* __restrict__ const_types, int absorbing, int32_t* __restrict__ status,
    const int32_t* __restrict__ batch /* crystal of each atom, or null: searched in `offsets` */) {
    const float* __restrict__ z_frac = noise.z_frac;
    const float* __restrict__ u_types = noise.u_types;
    // crystal of this atom = largest b with offsets[b] <= i: a 64-ary search by the whole wave (each level one
    // load per lane + a ballot) instead of log2(B) dependent loads
    int lo = 0, hi = B;
    if (batch != nullptr) {
        lo = batch[i];  // (the sampling loop has the index from prep_kernel: two dependent loads fewer)
    } else
    while (hi - lo > 1) {
        const int span = hi - lo, step = (span + 63) >> 6;
        const int probe = lo + lane * step;
        const bool le = probe < hi && offsets[probe] <= i;       // monotone in lane: true for lanes 0..c-1
        const int c = __builtin_popcountll(__ballot(le));         // c >= 1 because offsets[lo] <= i
        lo = lo + (c - 1) * step;
        hi = min(lo + step, hi);
    }
    int t = tstep[lo];
    t = t < 1 ? 1 : (t > T ? T : t);

    if (lane < 3) {
        const float s = ve_sigmas[t];
        const float sp = ve_sigmas[t - 1];  // t >= 1 here; the reference's t == 0 branch is unreachable in sampling
        const float s2 = s * s, sp2 = sp * sp;
        const size_t g = 3 * (size_t)i + lane;
        const float mean = frac[g] - eps[g] * (s2 - sp2);
        const float stdv = sqrtf((sp2 * (s2 - sp2)) / s2);
        const float zf = z_frac ? z_frac[g] : philox_normal(noise.seed, (uint32_t)t, ARREAU_DRAW_Z_FRAC, (uint32_t)g);
        frac[g] = remainder_one(mean + stdv * zf);
    }

    // ---- D3PM posterior logits ------------------------------------------------------------------
    const int s0 = lane, s1 = lane + 64;
    const bool v0 = s0 < S, v1 = s1 < S;
    const float* lg = logits + (size_t)i * S;
    const float l0 = v0 ? lg[s0] : -INFINITY, l1 = v1 ? lg[s1] : -INFINITY;
    float post0, post1;
    if (t == 1) {
        post0 = l0; post1 = l1;  // raw x0 logits at the last step (d3pm.py:106-108)
    } else {
        float mx = fmaxf(l0, l1);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        const float e0 = v0 ? expf(l0 - mx) : 0.f, e1 = v1 ? expf(l1 - mx) : 0.f;
        float sum = e0 + e1;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
        const float p0 = e0 / sum, p1 = e1 / sum;  // softmax of the x0 logits; lane holds classes s0, s1
        int xt = types[i];
        if ((xt < 0 || xt >= S) && lane == 0) atomicOr(status, ARREAU_STATUS_BAD_TYPE);  // clamped, but flagged
        xt = xt < 0 ? 0 : (xt >= S ? S - 1 : xt);
        const float* q1row = q1t + ((size_t)(t - 1) * S + xt) * S;  // fact1 = Q_t^T[x_t, :]
        const float* qm = qmats + (size_t)(t - 2) * S * S;          // Qbar_{t-1} (reference index t-2)
        float f2a = 0.f, f2b = 0.f;
        if (absorbing) {
            // Absorbing ("mask") chain: Qbar_t is diagonal plus the mask column (checked on the host for every t at model
            // creation).  The dense loop below adds exact zeros everywhere else, so these are bit for bit its sums: for an
            // ordinary class s only the term c = s, for the mask class the whole column in class order -- without the
            // S x S read per atom.
            const int mask = S - 1;
            const float d0 = v0 ? qm[(size_t)s0 * S + s0] : 0.f, d1 = v1 ? qm[(size_t)s1 * S + s1] : 0.f;
            const float c0 = v0 ? qm[(size_t)s0 * S + mask] : 0.f, c1 = v1 ? qm[(size_t)s1 * S + mask] : 0.f;
            f2a = fmaf(p0, d0, 0.f);
            f2b = fmaf(p1, d1, 0.f);
            float fm = 0.f;
            // (c is wave-uniform: the broadcasts are v_readlane, not LDS-crossbar permutes -- round 3: the 2 S permutes per atom
            // were most of this kernel's time)
            auto lane_value = [](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
            for (int c = 0; c < S; ++c) {
                const float pc = c < 64 ? lane_value(p0, c) : lane_value(p1, c - 64);
                const float qc = c < 64 ? lane_value(c0, c) : lane_value(c1, c - 64);
                fm = fmaf(pc, qc, fm);
            }
            if (s0 == mask) f2a = fm;
            if (s1 == mask) f2b = fm;
        } else
        // fact2 = softmax . Qbar: rows of Qbar are fetched 16 at a time (independent loads in flight), then the
        // softmax entries are broadcast from the lanes that hold them
        for (int c0 = 0; c0 < S; c0 += 16) {
            float qa[16], qb[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c = min(c0 + i, S - 1);
                qa[i] = v0 ? qm[(size_t)c * S + s0] : 0.f;
                qb[i] = v1 ? qm[(size_t)c * S + s1] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c = c0 + i;  // wave-uniform
                float sc = c < 64 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p0), c))
                                  : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p1), c - 64));
                sc = c < S ? sc : 0.f;
                f2a = fmaf(sc, qa[i], f2a);
                f2b = fmaf(sc, qb[i], f2b);
            }
        }
        post0 = v0 ? logf(q1row[s0] + D3PM_EPS) + logf(f2a + D3PM_EPS) : -INFINITY;
        post1 = v1 ? logf(q1row[s1] + D3PM_EPS) + logf(f2b + D3PM_EPS) : -INFINITY;
    }
    // ---- Gumbel arg-max (d3pm.py:206-214) ----------------------------------------------------------
    const float scale = (t != 1) ? 1.0f : 0.2f;
    // (the array-or-generator choice is made on the kernel argument itself -- a scalar compare -- not on a per-lane pointer:
    // no per-lane 64-bit integer compares on this path)
    const bool have_u = u_types != nullptr;
    const float* un = u_types + (have_u ? (size_t)i * S : 0);
    auto draw_u = [&](int s_) {
        return have_u ? un[s_] : philox_uniform(noise.seed, (uint32_t)t, ARREAU_DRAW_U_TYPES, (uint32_t)((size_t)i * S + s_));
    };
    float best = -INFINITY;
    int besti = 0x7fffffff;
    if (v0) {
        const float u = fminf(fmaxf(draw_u(s0), D3PM_EPS), 1.0f);
        best = post0 + (-logf(-logf(u))) * scale;
        besti = s0;
    }
    if (v1) {
        const float u = fminf(fmaxf(draw_u(s1), D3PM_EPS), 1.0f);
        const float val = post1 + (-logf(-logf(u))) * scale;
        if (val > best) { best = val; besti = s1; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ob = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(besti, off, 64);
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }  // first index wins ties
    }
    // use_constant_atomic_symbols (lightning_wrappers/diffusion.py:231-236): the fixed species are re-imposed each step
    if (lane == 0) types[i] = const_types ? const_types[i] : besti;
}

// the form the stand-alone kernel uses: workgroup `blk` of four waves, one atom each
__device__ __forceinline__ void reverse_atoms_body(
    int blk /* block of the atom part */, float* __restrict__ frac, int32_t* __restrict__ types, const int32_t* __restrict__ tstep,
    const int32_t* __restrict__ offsets, int B, int N, const float* __restrict__ eps,
    const float* __restrict__ logits, StepNoiseSrc noise,
    const float* __restrict__ ve_sigmas, const float* __restrict__ q1t, const float* __restrict__ qmats, int S,
    int T, const int32_t* __restrict__ const_types, int absorbing, int32_t* __restrict__ status, int n0,
    const int32_t* __restrict__ batch) {
    const int i = n0 + blk * 4 + (int)(threadIdx.x >> 6);  // atoms n0 .. N-1
    if (i >= N) return;  // wave-uniform; no block-level barrier below
    reverse_one_atom(i, threadIdx.x & 63, frac, types, tstep, offsets, B, eps, logits, noise, ve_sigmas, q1t, qmats, S, T, const_types, absorbing,
                     status, batch);
}

// Sampling loop (round 3): the lattice update of a crystal by ONE workgroup that then also prepares the crystal's NEXT step --
// what prep_kernel (node.hip) would compute at the top of that step from the lengths just written: the cell (into the
// caller's lattice AND the workspace copy the network reads) and the per-crystal part of the embedding for timestep t - 1.
// The step then needs no prep launch (the Cartesian positions, prep's other product, are formed by the neighbour-list waves
// from the fractional coordinates).  Same arithmetic as reverse_lattice_body + prep_kernel, thread for thread.
__device__ __forceinline__ void reverse_crystal_block(int b, float* __restrict__ lengths, const float* __restrict__ angles,
                                                      const int32_t* __restrict__ tstep, const int32_t* __restrict__ offsets,
                                                      const float* __restrict__ len0, StepNoiseSrc noise,
                                                      const float* __restrict__ alpha_bars, const float* __restrict__ betas, int T,
                                                      float* __restrict__ lattice, const float* __restrict__ fixed_lengths,
                                                      int32_t* __restrict__ status, const float* __restrict__ gs_atoms,
                                                      float* __restrict__ len0_out, float* __restrict__ lattice_ws,
                                                      float* __restrict__ cvec_next, const float* __restrict__ t_emb_w,
                                                      const float* __restrict__ embT, int S, int C) {
    __shared__ float newlen[3];
    __shared__ float feat[ARREAU_T_EMB_DIM + ARREAU_N_CRYSTAL_FEATS];
    const int t_raw = tstep[b];
    if (threadIdx.x == 0 && (t_raw < 1 || t_raw > T)) atomicOr(status, ARREAU_STATUS_BAD_TIMESTEP);  // clamped, but flagged
    const int t = t_raw < 1 ? 1 : (t_raw > T ? T : t_raw);
    const int first = offsets[b], last = offsets[b + 1];
    if (threadIdx.x < 3)
        newlen[threadIdx.x] = reverse_length_component(b, threadIdx.x, t, first, last, lengths, len0, noise, alpha_bars, betas, fixed_lengths,
                                                       gs_atoms, len0_out);
    __syncthreads();
    const float* ang = angles + 3 * b;
    if (threadIdx.x == 0) {
        float Lm[9];
        arreau_prep_cell(newlen, ang, Lm);  // lattice_from_params (lattice_helpers.py:55-105)
#pragma unroll
        for (int q = 0; q < 9; ++q) { lattice[9 * b + q] = Lm[q]; lattice_ws[9 * b + q] = Lm[q]; }
    }
    arreau_prep_cvec(t_raw - 1, last - first, newlen, ang, betas, t_emb_w, embT, S, C, T, feat, cvec_next + (size_t)b * C, status);
}

