// K2+K3: pair invariants -> polynomial features -> basis MLP -> window -> per-layer kernel
// projection, fused in registers, one wave per 32 (edge, orientation) rows.
//
// Replaces, per denoising step (sizes for B=256, n=20: 655 360 rows):
//   transforms/invariants.py:69-88 + geometry/invariants.py:10-31  (attr [E,O,6])
//   nn/embedding.py:10-14                                          (258-column polynomial tensor, 676 MB)
//   models/ponita.py:65,94 + utils/windowing.py:21-29              (kernel_basis [E,O,D], 671 MB)
//   nn/conv.py:110 for all L layers                                (kernel [E,O,C] x L)
// Nothing but the final per-layer kernels ([L][N*k*O][C] fp32) is written to HBM.
#include <utility>

#include "internal.h"

// ---- compile-time monomial table: index f -> (degree, i, j, k), canonical order ----------------
struct MonoIdx { int n, i, j, k; };
__host__ __device__ constexpr MonoIdx mono_idx(int f) {
    int p = 0;
    for (int i = 0; i < 6; ++i, ++p)
        if (p == f) return {1, i, 0, 0};
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j, ++p)
            if (p == f) return {2, i, j, 0};
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j)
            for (int k = j; k < 6; ++k, ++p)
                if (p == f) return {3, i, j, k};
    return {0, 0, 0, 0};  // padding columns 83..95
}

template <int F>
__device__ __forceinline__ float mono_at(const float (&a)[6]) {
    constexpr MonoIdx m = mono_idx(F);
    if constexpr (m.n == 1) return a[m.i];
    else if constexpr (m.n == 2) return a[m.i] * a[m.j];
    else if constexpr (m.n == 3) return (a[m.i] * a[m.j]) * a[m.k];
    else return 0.0f;
}

// register r of input tile T feeds feature 32T + 8(r>>2) + (r&3) on lane half 0 and that + 4 on half 1
template <int T, int... R>
__device__ __forceinline__ f32x16 mono_tile(const float (&a)[6], int h, std::integer_sequence<int, R...>) {
    f32x16 v;
    ((v[R] = h ? mono_at<32 * T + 8 * (R >> 2) + (R & 3) + 4>(a) : mono_at<32 * T + 8 * (R >> 2) + (R & 3)>(a)), ...);
    return v;
}

// Row r of a node's tile: slot = r >> 4, orientation = r & 15.  Workgroup = one receiver node
// (k*16 rows, 4 waves at k = 8); wave w owns slots 2w, 2w+1 -> 32 rows on lanes j = lane & 31
// (both lane halves h = lane >> 5 carry the same row; they differ in which k-index they feed).
template <int C, int D>
__global__ __launch_bounds__(256, 1) void edge_kernel(
    const float* __restrict__ nbr_dir,   // [N][k][3]
    const float* __restrict__ nbr_dist,  // [N][k]
    const int32_t* __restrict__ deg,     // [N]
    const int32_t* __restrict__ batch,   // [N] crystal of node
    const float* __restrict__ lattice,   // [B][9]
    const float* __restrict__ ori,       // [16][3]
    const float* __restrict__ w1p, const float* __restrict__ b1, const float* __restrict__ w2p,
    const float* __restrict__ b2, const float* __restrict__ wkp, float r_max, int N, int k, int L,
    float* __restrict__ kbuf)            // [L][N*k*16][C]
{
    constexpr int TC = C / 32, TD = D / 32, TM = ARREAU_MONO_PAD / 32;
    const int node = blockIdx.x;
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int h = lane >> 5, j = lane & 31;
    const int nd = min(deg[node], k);
    if (2 * wave >= nd) return;  // wave-uniform: both slots of this wave are empty
    const int slot = 2 * wave + (j >> 4);
    const int o = j & 15;
    const int slot_c = min(slot, k - 1);  // k odd: the upper half-tile of the last wave is padding

    // ---- per-row attributes (transforms/invariants.py:82-88) ------------------------------------
    const size_t e = (size_t)node * k + slot_c;
    const float dx = nbr_dir[3 * e + 0], dy = nbr_dir[3 * e + 1], dz = nbr_dir[3 * e + 2];
    const float dist = nbr_dist[e];
    const float ox = ori[3 * o + 0], oy = ori[3 * o + 1], oz = ori[3 * o + 2];
    float a[6];
    a[0] = (dx * ox + dy * oy) + dz * oz;                       // inv1 = dir . o
    {
        const float rx = dx - a[0] * ox, ry = dy - a[0] * oy, rz = dz - a[0] * oz;
        a[1] = sqrtf((rx * rx + ry * ry) + rz * rz);            // inv2 = |dir - inv1 o|
    }
    a[2] = dist;
    {
        // torch CosineSimilarity(dim=-1, eps=1e-8): normalise each vector by max(|v|, eps), then dot
        const float* Lm = lattice + 9 * (size_t)batch[node];
        const float dn = fmaxf(sqrtf((dx * dx + dy * dy) + dz * dz), 1e-8f);
        const float ux = dx / dn, uy = dy / dn, uz = dz / dn;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float lx = Lm[3 * i], ly = Lm[3 * i + 1], lz = Lm[3 * i + 2];
            const float ln = fmaxf(sqrtf((lx * lx + ly * ly) + lz * lz), 1e-8f);
            a[3 + i] = (ux * (lx / ln) + uy * (ly / ln)) + uz * (lz / ln);
        }
    }
    // smooth cutoff (utils/windowing.py:21-29, p = 6), times (d < r_max)
    float window;
    {
        const float u = dist / r_max;
        const float u2 = u * u, u6 = u2 * u2 * u2;
        window = 1.0f - 28.0f * u6 + 48.0f * u6 * u - 21.0f * u6 * u2;
        window = (dist < r_max) ? window : 0.0f;
        if (slot >= nd) window = 0.0f;  // padding slot inside a live wave: contributes nothing
    }

    // ---- layer 1: h = GELU(W1f . mono + b1)            [C x 32 rows] ----------------------------
    // B operand = the 83 distinct monomials (canonical order of fold_poly_weight in model.hip),
    // generated straight into the registers the MFMA reads (no array, no scratch).
    f32x16 acc1[TC];
    arreau_bias_tiles<TC>(acc1, b1, h);
    static_assert(TM == 3, "three monomial tiles");
    {
        const f32x16 bm[TM] = {mono_tile<0>(a, h, std::make_integer_sequence<int, 16>{}),
                               mono_tile<1>(a, h, std::make_integer_sequence<int, 16>{}),
                               mono_tile<2>(a, h, std::make_integer_sequence<int, 16>{})};
        arreau_gemm_chain<TC, TM>(acc1, w1p, TM * ARREAU_PACK_TILE_FLOATS, bm, lane);
    }
    arreau_gelu_tiles<TC>(acc1);

    // ---- layer 2: basis = GELU(W2 . h + b2) * window   [D x 32 rows] ----------------------------
    f32x16 acc2[TD];
    arreau_bias_tiles<TD>(acc2, b2, h);
    arreau_gemm_chain<TD, TC>(acc2, w2p, TC * ARREAU_PACK_TILE_FLOATS, acc1, lane);
#pragma unroll
    for (int u = 0; u < TD; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[u][r] = arreau_gelu(acc2[u][r]) * window;

    // ---- per layer: kernel_l = Wk_l . basis            [C x 32 rows]  (conv.py:110) -------------
    const size_t row = ((size_t)node * k + slot_c) * 16 + o;
    const size_t layer_stride = (size_t)N * k * 16 * C;
    const bool live = slot < k;
    for (int l = 0; l < L; ++l) {
        const float* wl = wkp + (size_t)l * TC * TD * ARREAU_PACK_TILE_FLOATS;
        f32x16 acc3[TC];
#pragma unroll
        for (int u = 0; u < TC; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc3[u][r] = 0.0f;
        arreau_gemm_chain<TC, TD>(acc3, wl, TD * ARREAU_PACK_TILE_FLOATS, acc2, lane);
        if (live) {
            float* dst = kbuf + (size_t)l * layer_stride + row * C;
#pragma unroll
            for (int u = 0; u < TC; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v = {acc3[u][4 * q], acc3[u][4 * q + 1], acc3[u][4 * q + 2], acc3[u][4 * q + 3]};
                    *reinterpret_cast<f32x4*>(dst + 32 * u + 8 * q + 4 * h) = v;
                }
        }
    }
}

int arreau_launch_edge(const arreau_model* m, const float* dir, const float* dist, const int32_t* deg,
                       const int32_t* batch, const float* lattice, int N, float* kbuf, hipStream_t s) {
    if (N == 0) return ARREAU_OK;
    if (m->C == 128 && m->D == 256) {
        hipLaunchKernelGGL((edge_kernel<128, 256>), dim3(N), dim3(64 * ((m->k + 1) / 2)), 0, s, dir, dist, deg, batch,
                           lattice, m->ori, m->w1p, m->b1, m->w2p, m->b2, m->wkp, m->cfg.radius, N, m->k, m->L, kbuf);
    } else {
        arreau_set_error("edge kernel: unsupported (hidden_dim, basis_dim)");
        return ARREAU_EINVAL;
    }
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}
