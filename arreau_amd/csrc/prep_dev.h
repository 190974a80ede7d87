// Per-crystal part of the sampler's step set-up, shared by prep_kernel (node.hip) and by the crystal blocks of the update
// launch (update.hip), which prepare the NEXT step inside the sampling loop: cell from (lengths, angles), and the part of the
// embedding that is constant inside a crystal.  Test infrastructure does not include this file.
#pragma once
#include "internal.h"

// lattice_from_params (diffusion/lattice_helpers.py:55-105): rows a, b, c of the cell
__device__ __forceinline__ void arreau_prep_cell(const float* len, const float* ang, float* Lm) {
    const float a = len[0], b = len[1], c = len[2];
    const float ca = cosf(ang[0]), cb = cosf(ang[1]), cg = cosf(ang[2]);
    const float sa = sinf(ang[0]), sb = sinf(ang[1]);
    float val = (ca * cb - cg) / (sa * sb);
    val = fminf(fmaxf(val, -1.0f), 1.0f);
    const float gs = acosf(val);
    Lm[0] = a * sb;             Lm[1] = 0.0f;               Lm[2] = a * cb;
    Lm[3] = -b * sa * cosf(gs); Lm[4] = b * sa * sinf(gs);  Lm[5] = b * ca;
    Lm[6] = 0.0f;               Lm[7] = 0.0f;               Lm[8] = c;
}

// cvec[c] = sum_i feat[i] * embT[S + i][c] over the 64 time features (GaussianFourierProjection of betas[t],
// diffusion_helpers.py:23-25; diffusion_loss.py:126-127) and the 10 crystal features (n, lengths, angles, |lengths / n|:
// diffusion_loss.py:139-149).  Called by every thread of the workgroup (at least 64 + 10 threads; two barriers inside);
// `feat` is workgroup-shared scratch of ARREAU_T_EMB_DIM + ARREAU_N_CRYSTAL_FEATS floats; len / ang may be shared or global.
__device__ __forceinline__ void arreau_prep_cvec(int t, int n_atoms, const float* len, const float* ang,
                                                 const float* __restrict__ betas, const float* __restrict__ t_emb_w,
                                                 const float* __restrict__ embT, int S, int C, int T, float* feat,
                                                 float* __restrict__ cvec_b, int32_t* __restrict__ status) {
    if (threadIdx.x < 32) {
        if ((t < 0 || t > T) && threadIdx.x == 0) atomicOr(status, ARREAU_STATUS_BAD_TIMESTEP);  // clamped, but flagged
        t = t < 0 ? 0 : (t > T ? T : t);
        const float proj = ((betas[t] * t_emb_w[threadIdx.x]) * 2.0f) * 3.14159265358979323846f;
        feat[threadIdx.x] = sinf(proj);
        feat[32 + threadIdx.x] = cosf(proj);
    } else if (threadIdx.x < 32 + ARREAU_N_CRYSTAL_FEATS) {
        const int i = threadIdx.x - 32;
        float v;
        if (i == 0) v = (float)n_atoms;
        else if (i < 4) v = len[i - 1];
        else if (i < 7) v = ang[i - 4];
        else v = fabsf(len[i - 7] / (float)n_atoms);
        feat[ARREAU_T_EMB_DIM + i] = v;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.0f;
#pragma unroll 37  // 74 rows: two batches of independent loads in flight instead of one L2 round trip per row
        for (int i = 0; i < ARREAU_T_EMB_DIM + ARREAU_N_CRYSTAL_FEATS; ++i) acc += feat[i] * embT[(size_t)(S + i) * C + c];
        cvec_b[c] = acc;
    }
}
