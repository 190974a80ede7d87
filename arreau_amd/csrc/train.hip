// Score-matching training loss, forward part (BASELINE config 5; DiffusionLoss.__call__,
// diffusion/diffusion_loss.py:204-274): forward noising of (coordinates, atom types, cell lengths), then -- after
// arreau_predict_scores on the noised state -- the three losses and their gradients with respect to the network
// outputs (the seeds of the backward pass).  Latency-bound per-atom / per-crystal work; no MFMA.
#include "internal.h"

#define D3PM_EPS 1e-6f        // d3pm.py:23
#define D3PM_HYBRID 0.001f    // d3pm.py:15 hybrid_loss_coeff

namespace {
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// torch.remainder(x, 1)
__device__ __forceinline__ float remainder_one_t(float x) {
    float m = fmodf(x, 1.0f);
    if (m != 0.0f && m < 0.0f) m += 1.0f;
    return m;
}

// ---------------------------------------------------------------------------------------------
// Per crystal: matrix_to_params (lattice_helpers.py:16-35), VP_lattice.forward on the lengths
// (diffusion_helpers.py:156-163), and the inverse cell for cart_to_frac_coords (:233-251; the reference
// uses pinv, which equals the inverse for the full-rank cells of a dataset).
// ---------------------------------------------------------------------------------------------
__global__ void noise_crystals_kernel(const float* __restrict__ lattice0, const int32_t* __restrict__ tstep,
                                      const float* __restrict__ z_len, const float* __restrict__ alpha_bars, int B,
                                      int T, float* __restrict__ lengths, float* __restrict__ angles,
                                      float* __restrict__ noisy_lengths, float* __restrict__ inv_lattice,
                                      int32_t* __restrict__ status) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* m = lattice0 + 9 * (size_t)b;
    float len[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) len[i] = sqrtf((m[3 * i] * m[3 * i] + m[3 * i + 1] * m[3 * i + 1]) + m[3 * i + 2] * m[3 * i + 2]);
    int t = tstep[b];
    if (t < 1 || t > T) atomicOr(status, ARREAU_STATUS_BAD_TIMESTEP);
    t = t < 1 ? 1 : (t > T ? T : t);
    const float ab = alpha_bars[t];
    const float sa = sqrtf(ab), sb = sqrtf(1.0f - ab);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int j = (i + 1) % 3, k = (i + 2) % 3;
        const float dot = (m[3 * j] * m[3 * k] + m[3 * j + 1] * m[3 * k + 1]) + m[3 * j + 2] * m[3 * k + 2];
        const float c = fminf(fmaxf(dot / (len[j] * len[k]), -1.0f), 1.0f);
        angles[3 * b + i] = acosf(c);
        lengths[3 * b + i] = len[i];
        noisy_lengths[3 * b + i] = sa * len[i] + sb * z_len[3 * b + i];
    }
    // inverse by the adjugate (double: the cell is data, conditioning is not ours to assume)
    const double a = m[0], bb = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i9 = m[8];
    const double det = a * (e * i9 - f * h) - bb * (d * i9 - f * g) + c * (d * h - e * g);
    const double r = 1.0 / det;
    float* o = inv_lattice + 9 * (size_t)b;
    o[0] = (float)((e * i9 - f * h) * r); o[1] = (float)((c * h - bb * i9) * r); o[2] = (float)((bb * f - c * e) * r);
    o[3] = (float)((f * g - d * i9) * r); o[4] = (float)((a * i9 - c * g) * r); o[5] = (float)((c * d - a * f) * r);
    o[6] = (float)((d * h - e * g) * r); o[7] = (float)((bb * g - a * h) * r); o[8] = (float)((a * e - bb * d) * r);
}

// ---------------------------------------------------------------------------------------------
// Per atom (one wave): VE_pbc.forward (diffusion_helpers.py:43-63) incl. min_distance_sqr_pbc over the 27
// images (:254-325, first minimum wins) and cart_to_frac_coords % 1; D3PM.q_sample (d3pm.py:119-127).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void noise_atoms_kernel(
    const float* __restrict__ frac0, const int32_t* __restrict__ types0, const float* __restrict__ lattice0,
    const float* __restrict__ inv_lattice, const int32_t* __restrict__ tstep, const int32_t* __restrict__ offsets, int B,
    int N, const float* __restrict__ z_frac, const float* __restrict__ u_types, const float* __restrict__ ve_sigmas,
    const float* __restrict__ qmats, int S, int T, float* __restrict__ noisy_frac, float* __restrict__ target_eps,
    int32_t* __restrict__ noisy_types, int32_t* __restrict__ status) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave;
    if (i >= N) return;
    int lo = 0, hi = B;  // crystal of this atom (64-ary search, see reverse_atoms_body in update.hip)
    while (hi - lo > 1) {
        const int span = hi - lo, step = (span + 63) >> 6;
        const int probe = lo + lane * step;
        const bool le = probe < hi && offsets[probe] <= i;
        const int c = __builtin_popcountll(__ballot(le));
        lo = lo + (c - 1) * step;
        hi = min(lo + step, hi);
    }
    int t = tstep[lo];
    t = t < 1 ? 1 : (t > T ? T : t);
    const float* Lm = lattice0 + 9 * (size_t)lo;
    // ---- coordinates: lanes 0..26 take one image each -----------------------------------------------------
    {
        const float sigma = ve_sigmas[t];
        float fn[3], f0[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            f0[d] = frac0[3 * (size_t)i + d];
            fn[d] = remainder_one_t(f0[d] + z_frac[3 * (size_t)i + d] * sigma);
        }
        float cn[3], cp[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            cn[j] = (fn[0] * Lm[j] + fn[1] * Lm[3 + j]) + fn[2] * Lm[6 + j];
            cp[j] = (f0[0] * Lm[j] + f0[1] * Lm[3 + j]) + f0[2] * Lm[6 + j];
        }
        const int cell = lane < 27 ? lane : 26;
        const float ca = (float)(cell / 9 - 1), cb = (float)((cell / 3) % 3 - 1), cc = (float)(cell % 3 - 1);
        float v[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float off = (Lm[j] * ca + Lm[3 + j] * cb) + Lm[6 + j] * cc;  // (lattice^T . cell)_j, bmm order
            v[j] = cn[j] - (cp[j] + off);
        }
        const float d2 = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
        // arg-min over the 27 images, smallest index on ties: key = (d2 bits, image) -- d2 >= 0, so the bit pattern orders
        unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)cell;
        if (lane >= 27) key = ~0ull;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const unsigned long long o = __shfl_xor(key, off, 64);
            key = o < key ? o : key;
        }
        const int best = (int)(key & 0xffffffffu);
        float bv[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) bv[j] = __shfl(v[j], best, 64);
        if (lane < 3) {
            const float* iv = inv_lattice + 9 * (size_t)lo;
            const float fr = (bv[0] * iv[lane] + bv[1] * iv[3 + lane]) + bv[2] * iv[6 + lane];
            target_eps[3 * (size_t)i + lane] = remainder_one_t(fr);
            noisy_frac[3 * (size_t)i + lane] = fn[lane];
        }
    }
    // ---- atom type: argmax_s log(Qbar_t[x0, s] + eps) + gumbel(u_s) ---------------------------------------------
    int x0 = types0[i];
    if ((x0 < 0 || x0 >= S) && lane == 0) atomicOr(status, ARREAU_STATUS_BAD_TYPE);
    x0 = x0 < 0 ? 0 : (x0 >= S ? S - 1 : x0);
    const float* qrow = qmats + ((size_t)(t - 1) * S + x0) * S;
    const float* un = u_types + (size_t)i * S;
    float best = -INFINITY;
    int besti = 0x7fffffff;
    for (int s = lane; s < S; s += 64) {
        const float u = fminf(fmaxf(un[s], D3PM_EPS), 1.0f);
        const float val = logf(qrow[s] + D3PM_EPS) + (-logf(-logf(u)));
        if (val > best) { best = val; besti = s; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ob = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(besti, off, 64);
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    if (lane == 0) noisy_types[i] = besti;
}

// ---------------------------------------------------------------------------------------------
// Per atom (one wave, 2 classes per lane, S <= 128): the wrapped coordinate error (diffusion_loss.py:95-110), the D3PM
// hybrid loss terms (d3pm.py:74-117, 146-163) and their gradients with respect to pred_eps and the logits.
//   terms[i] = {frac error, vb, ce}
// ---------------------------------------------------------------------------------------------
struct Post { float a, b; };
// q_posterior_logits for float x0 logits held as (l0, l1) per lane; returns the posterior logits (d3pm.py:74-110)
__device__ __forceinline__ Post posterior(float l0, float l1, bool v0, bool v1, int s0, int s1, const float* q1row,
                                          const float* qm, int S, int t, float& p0, float& p1, float& f0, float& f1,
                                          int absorbing) {
    const float mx = wave_max(fmaxf(v0 ? l0 : -INFINITY, v1 ? l1 : -INFINITY));
    const float e0 = v0 ? expf(l0 - mx) : 0.f, e1 = v1 ? expf(l1 - mx) : 0.f;
    const float sum = wave_sum(e0 + e1);
    p0 = e0 / sum;
    p1 = e1 / sum;
    f0 = 0.f;
    f1 = 0.f;
    if (t == 1) return {l0, l1};
    if (absorbing) {
        // Absorbing ("mask") chain: Qbar is diagonal plus the mask column (checked on the host for every t at model
        // creation).  The dense loop below adds exact zeros everywhere else, so these are bit for bit its sums: for an
        // ordinary class s only the term c = s, for the mask class the whole column in class order (update.hip does the same).
        const int mask = S - 1;
        const float d0 = v0 ? qm[(size_t)s0 * S + s0] : 0.f, d1 = v1 ? qm[(size_t)s1 * S + s1] : 0.f;
        const float c0 = v0 ? qm[(size_t)s0 * S + mask] : 0.f, c1 = v1 ? qm[(size_t)s1 * S + mask] : 0.f;
        if (v0) f0 += p0 * d0;
        if (v1) f1 += p1 * d1;
        float fm = 0.f;
        // (c is wave-uniform: the broadcasts are v_readlane, not LDS-crossbar permutes -- round 5, as update_dev.h since round 3: the
        // 4 S permutes per atom of the two posteriors were most of loss_atoms_kernel's 26 us)
        auto lane_value = [](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
        for (int c = 0; c < S; ++c) {  // wave-uniform
            const float pc = c < 64 ? lane_value(p0, c) : lane_value(p1, c - 64);
            const float qc = c < 64 ? lane_value(c0, c) : lane_value(c1, c - 64);
            fm += pc * qc;
        }
        if (s0 == mask) f0 = fm;
        if (s1 == mask) f1 = fm;
    } else
    for (int c = 0; c < S; ++c) {  // fact2 = softmax . Qbar_{t-1}
        const float sc = c < 64 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p0), c))
                                : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p1), c - 64));
        if (v0) f0 += sc * qm[(size_t)c * S + s0];
        if (v1) f1 += sc * qm[(size_t)c * S + s1];
    }
    Post r;
    r.a = v0 ? logf(q1row[s0] + D3PM_EPS) + logf(f0 + D3PM_EPS) : -INFINITY;
    r.b = v1 ? logf(q1row[s1] + D3PM_EPS) + logf(f1 + D3PM_EPS) : -INFINITY;
    return r;
}

__global__ __launch_bounds__(256) void loss_atoms_kernel(
    const float* __restrict__ pred_eps, const float* __restrict__ target_eps, const float* __restrict__ logits,
    const int32_t* __restrict__ types0, const int32_t* __restrict__ noisy_types, const int32_t* __restrict__ tstep,
    const int32_t* __restrict__ offsets, int B, int N, const float* __restrict__ q1t, const float* __restrict__ qmats,
    int S, int T, float* __restrict__ terms /*[N][3]*/, float* __restrict__ g_eps /*[N][3] or null*/,
    float* __restrict__ g_logits /*[N][S] or null*/, int absorbing) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave;
    if (i >= N) return;
    int lo = 0, hi = B;
    while (hi - lo > 1) {
        const int span = hi - lo, step = (span + 63) >> 6;
        const int probe = lo + lane * step;
        const bool le = probe < hi && offsets[probe] <= i;
        const int c = __builtin_popcountll(__ballot(le));
        lo = lo + (c - 1) * step;
        hi = min(lo + step, hi);
    }
    int t = tstep[lo];
    t = t < 1 ? 1 : (t > T ? T : t);
    const float invN = 1.0f / (float)N;
    // ---- coordinates ----------------------------------------------------------------------------------------
    float ef = 0.f;
    if (lane < 3) {
        const size_t g = 3 * (size_t)i + lane;
        const float d = pred_eps[g] - target_eps[g];
        const float w = fminf(fmaxf(remainder_one_t(fabsf(d)), 0.0f), 1.0f);
        const bool lower = w <= 1.0f - w;
        const float m = lower ? w : 1.0f - w;
        ef = m * m;
        if (g_eps) g_eps[g] = 2.0f * invN * m * (d < 0.f ? -1.0f : 1.0f) * (lower ? 1.0f : -1.0f);
    }
    ef = wave_sum(ef);
    // ---- atom types -------------------------------------------------------------------------------------------
    const int s0 = lane, s1 = lane + 64;
    const bool v0 = s0 < S, v1 = s1 < S;
    int x0 = types0[i], xt = noisy_types[i];
    x0 = x0 < 0 ? 0 : (x0 >= S ? S - 1 : x0);
    xt = xt < 0 ? 0 : (xt >= S ? S - 1 : xt);
    const float* lg = logits + (size_t)i * S;
    const float l0 = v0 ? lg[s0] : 0.f, l1 = v1 ? lg[s1] : 0.f;
    const float* q1row = q1t + ((size_t)(t - 1) * S + xt) * S;
    const float* qm = qmats + (size_t)(t >= 2 ? t - 2 : T - 1) * S * S;  // q_mats[t-2]; index -1 wraps at t = 1 (unused there)
    // x0 given as class index: logits log(onehot + eps) (d3pm.py:80-84)
    const float on = logf(1.0f + D3PM_EPS), offv = logf(D3PM_EPS);
    float tp0, tp1, tf0, tf1, pp0, pp1, pf0, pf1;
    const Post tpost = posterior(s0 == x0 ? on : offv, s1 == x0 ? on : offv, v0, v1, s0, s1, q1row, qm, S, t, tp0, tp1, tf0, tf1, absorbing);
    const Post ppost = posterior(l0, l1, v0, v1, s0, s1, q1row, qm, S, t, pp0, pp1, pf0, pf1, absorbing);
    // vb = sum_s softmax(true + eps) (log_softmax(true + eps) - log_softmax(pred + eps))   (d3pm.py:112-117)
    auto log_softmax2 = [&](float a, float b, float& la, float& lb) {
        const float mx = wave_max(fmaxf(v0 ? a : -INFINITY, v1 ? b : -INFINITY));
        const float sum = wave_sum((v0 ? expf(a - mx) : 0.f) + (v1 ? expf(b - mx) : 0.f));
        const float lse = mx + logf(sum);
        la = a - lse;
        lb = b - lse;
    };
    float lt0, lt1, lp0, lp1;
    log_softmax2(tpost.a + D3PM_EPS, tpost.b + D3PM_EPS, lt0, lt1);
    log_softmax2(ppost.a + D3PM_EPS, ppost.b + D3PM_EPS, lp0, lp1);
    const float P0 = v0 ? expf(lt0) : 0.f, P1 = v1 ? expf(lt1) : 0.f;
    const float Q0 = v0 ? expf(lp0) : 0.f, Q1 = v1 ? expf(lp1) : 0.f;
    const float vb = wave_sum((v0 ? P0 * (lt0 - lp0) : 0.f) + (v1 ? P1 * (lt1 - lp1) : 0.f));
    // cross entropy of the raw logits against x0 (d3pm.py:158-161): pp = softmax(logits) from `posterior`
    float lx0, lx1;
    log_softmax2(l0, l1, lx0, lx1);
    const float ce = -wave_sum((v0 && s0 == x0 ? lx0 : 0.f) + (v1 && s1 == x0 ? lx1 : 0.f));
    if (lane == 0) {
        terms[3 * (size_t)i + 0] = ef;
        terms[3 * (size_t)i + 1] = vb;
        terms[3 * (size_t)i + 2] = ce;
    }
    if (g_logits) {
        // d(ce)/dlogits = softmax - onehot;  d(vb)/d(pred_post) = Q - P, chained through pred_post(logits)
        float gv0 = Q0 - P0, gv1 = Q1 - P1;
        if (t != 1) {
            const float g0 = v0 ? gv0 / (pf0 + D3PM_EPS) : 0.f, g1 = v1 ? gv1 / (pf1 + D3PM_EPS) : 0.f;
            // h_c = sum_s Qbar[c, s] g_s for this lane's classes c = s0, s1 (row c of Qbar against the wave's g)
            float h0 = 0.f, h1 = 0.f;
            if (absorbing) {
                // row c of Qbar holds its diagonal entry and the mask column: the dense chain below in its order (s = c, then
                // s = mask; for the mask row only the diagonal)
                const int mask = S - 1;
                const float gm = mask < 64 ? __shfl(g0, mask, 64) : __shfl(g1, mask - 64, 64);
                if (v0) { h0 += qm[(size_t)s0 * S + s0] * g0; if (s0 != mask) h0 += qm[(size_t)s0 * S + mask] * gm; }
                if (v1) { h1 += qm[(size_t)s1 * S + s1] * g1; if (s1 != mask) h1 += qm[(size_t)s1 * S + mask] * gm; }
            } else
            for (int s = 0; s < S; ++s) {
                const float gs = s < 64 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g0), s))
                                        : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g1), s - 64));
                if (v0) h0 += qm[(size_t)s0 * S + s] * gs;
                if (v1) h1 += qm[(size_t)s1 * S + s] * gs;
            }
            const float dotph = wave_sum((v0 ? pp0 * h0 : 0.f) + (v1 ? pp1 * h1 : 0.f));
            gv0 = pp0 * (h0 - dotph);
            gv1 = pp1 * (h1 - dotph);
        }
        if (v0) g_logits[(size_t)i * S + s0] = invN * ((pp0 - (s0 == x0 ? 1.0f : 0.0f)) + D3PM_HYBRID * gv0);
        if (v1) g_logits[(size_t)i * S + s1] = invN * ((pp1 - (s1 == x0 ? 1.0f : 0.0f)) + D3PM_HYBRID * gv1);
    }
}

// One workgroup: fixed-order sums of the per-atom terms and of the length errors (deterministic).
//   losses = {loss, error_frac_x, error_atomic_type, error_lattice, vb, ce}
__global__ __launch_bounds__(256) void loss_reduce_kernel(const float* __restrict__ terms, int N,
                                                          const float* __restrict__ pred_len,
                                                          const float* __restrict__ lengths,
                                                          const int32_t* __restrict__ offsets, int B,
                                                          float* __restrict__ losses, float* __restrict__ g_len) {
    __shared__ float part[4][256];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < N; i += 256) {
        acc[0] += terms[3 * (size_t)i];
        acc[1] += terms[3 * (size_t)i + 1];
        acc[2] += terms[3 * (size_t)i + 2];
    }
    for (int j = threadIdx.x; j < 3 * B; j += 256) {
        const int b = j / 3;
        const float n = (float)(offsets[b + 1] - offsets[b]);
        const float d = pred_len[j] - lengths[j] / n;  // target = lengths / num_atoms (diffusion_loss.py:264-267)
        acc[3] += d * d;
        if (g_len) g_len[j] = 2.0f * d / (float)(3 * B);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) part[q][threadIdx.x] = acc[q];
    __syncthreads();
    for (int stride = 128; stride >= 1; stride >>= 1) {
        if ((int)threadIdx.x < stride)
#pragma unroll
            for (int q = 0; q < 4; ++q) part[q][threadIdx.x] += part[q][threadIdx.x + stride];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float ef = part[0][0] / (float)N, vb = part[1][0] / (float)N, ce = part[2][0] / (float)N;
        const float el = part[3][0] / (float)(3 * B);
        const float et = vb * D3PM_HYBRID + ce;
        losses[0] = (ef + et) + el;  // coord/atom-type/lattice loss weights are all 1 (diffusion_loss.py:91-93)
        losses[1] = ef;
        losses[2] = et;
        losses[3] = el;
        losses[4] = vb;
        losses[5] = ce;
    }
}
}  // namespace

extern "C" int arreau_diffusion_noise(const arreau_model* m, const float* d_frac0, const int32_t* d_types0,
                                      const float* d_lattice0, const int32_t* d_t, const int32_t* d_off, int32_t B,
                                      int32_t N, const float* d_z_frac, const float* d_u_types, const float* d_z_lengths,
                                      float* d_noisy_frac, float* d_target_eps, int32_t* d_noisy_types,
                                      float* d_noisy_lengths, float* d_lengths, float* d_angles, float* d_inv_lattice,
                                      void* stream) {
    ARREAU_REQUIRE(m && d_frac0 && d_types0 && d_lattice0 && d_t && d_off && d_z_frac && d_u_types && d_z_lengths &&
                       d_noisy_frac && d_target_eps && d_noisy_types && d_noisy_lengths && d_lengths && d_angles &&
                       d_inv_lattice, "arreau_diffusion_noise: null pointer");
    ARREAU_REQUIRE(B >= 1 && N >= 0, "arreau_diffusion_noise: bad size");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(noise_crystals_kernel, dim3((B + 127) / 128), dim3(128), 0, s, d_lattice0, d_t, d_z_lengths,
                       m->vp_alpha_bars, B, m->T, d_lengths, d_angles, d_noisy_lengths, d_inv_lattice, m->status);
    ARREAU_CHECK_HIP(hipGetLastError());
    if (N > 0) {
        hipLaunchKernelGGL(noise_atoms_kernel, dim3((N + 3) / 4), dim3(256), 0, s, d_frac0, d_types0, d_lattice0,
                           d_inv_lattice, d_t, d_off, B, N, d_z_frac, d_u_types, m->ve_sigmas, m->qmats, m->S, m->T,
                           d_noisy_frac, d_target_eps, d_noisy_types, m->status);
        ARREAU_CHECK_HIP(hipGetLastError());
    }
    return ARREAU_OK;
}

extern "C" int arreau_diffusion_losses(const arreau_model* m, const float* d_pred_eps, const float* d_target_eps,
                                       const float* d_logits, const int32_t* d_types0, const int32_t* d_noisy_types,
                                       const int32_t* d_t, const float* d_pred_lengths, const float* d_lengths,
                                       const int32_t* d_off, int32_t B, int32_t N, float* d_terms, float* d_losses,
                                       float* d_grad_eps, float* d_grad_logits, float* d_grad_lengths, void* stream) {
    ARREAU_REQUIRE(m && d_pred_eps && d_target_eps && d_logits && d_types0 && d_noisy_types && d_t && d_pred_lengths &&
                       d_lengths && d_off && d_terms && d_losses, "arreau_diffusion_losses: null pointer");
    ARREAU_REQUIRE(B >= 1 && N >= 1, "arreau_diffusion_losses: bad size");
    ARREAU_REQUIRE(m->S <= 128, "arreau_diffusion_losses: num_atomic_states must be <= 128");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(loss_atoms_kernel, dim3((N + 3) / 4), dim3(256), 0, s, d_pred_eps, d_target_eps, d_logits, d_types0,
                       d_noisy_types, d_t, d_off, B, N, m->q1t, m->qmats, m->S, m->T, d_terms, d_grad_eps, d_grad_logits,
                       m->qmats_absorbing);
    ARREAU_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, s, d_terms, N, d_pred_lengths, d_lengths, d_off, B,
                       d_losses, d_grad_lengths);
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}
