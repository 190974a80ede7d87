// K2+K3: pair invariants -> polynomial features -> basis MLP -> window -> per-layer kernel
// projection, fused in registers, one wave per 64 (edge, orientation) rows.
//
// Replaces, per denoising step (sizes for B=256, n=20: 655 360 rows):
//   transforms/invariants.py:69-88 + geometry/invariants.py:10-31  (attr [E,O,6])
//   nn/embedding.py:10-14                                          (258-column polynomial tensor, 676 MB)
//   models/ponita.py:65,94 + utils/windowing.py:21-29              (kernel_basis [E,O,D], 671 MB)
//   nn/conv.py:110 for all L layers                                (kernel [E,O,C] x L)
// Nothing but the final per-layer kernels ([L][N*k*O][C] fp32) is written to HBM.
#include <stdlib.h>
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

// One wave owns 64 rows = 4 edge slots x 16 orientations of one receiver, as two 32-row column
// blocks cb = 0, 1 (row j of block cb: slot = slot0 + 2 cb + (j >> 4), orientation = j & 15; both lane
// halves carry the same row and differ in the k index they feed).  One wave per SIMD (the whole
// 512-register file): activations of both blocks stay in registers through the chain
//     monomials [96] -> h [C] -> basis [D] -> kernel_l [C] (l = 0..L-1).
// Output tiles are produced one at a time (out-tile-major), so only 2 accumulator tiles are live per
// GEMM and the three weight matrices become ONE linear stream of 16-byte fragments (w1 | w2 | wk_0..L-1,
// 1 KiB per wave-load, each fragment feeding 8 MFMAs).  The stream is prefetched PF groups ahead through
// a register ring, so L2 latency hides behind PF * 8 MFMAs (= 4096 cycles at PF = 8); no LDS, no barrier.
struct EdgeRow { float a[6]; float window; };

// attributes of one (edge slot, orientation) row  (transforms/invariants.py:82-88)
__device__ __forceinline__ EdgeRow edge_row(const float* __restrict__ nbr_dir, const float* __restrict__ nbr_dist,
                                            const float* __restrict__ ori, const float* __restrict__ Lm, size_t e,
                                            int o, float r_max, bool valid) {
    EdgeRow r;
    const float dx = nbr_dir[3 * e + 0], dy = nbr_dir[3 * e + 1], dz = nbr_dir[3 * e + 2];
    const float dist = nbr_dist[e];
    const float ox = ori[3 * o + 0], oy = ori[3 * o + 1], oz = ori[3 * o + 2];
    r.a[0] = (dx * ox + dy * oy) + dz * oz;  // inv1 = dir . o
    const float rx = dx - r.a[0] * ox, ry = dy - r.a[0] * oy, rz = dz - r.a[0] * oz;
    r.a[1] = sqrtf((rx * rx + ry * ry) + rz * rz);  // inv2 = |dir - inv1 o|
    r.a[2] = dist;
    // torch CosineSimilarity(dim=-1, eps=1e-8): normalise each vector by max(|v|, eps), then dot
    const float dn = fmaxf(sqrtf((dx * dx + dy * dy) + dz * dz), 1e-8f);
    const float ux = dx / dn, uy = dy / dn, uz = dz / dn;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float lx = Lm[3 * i], ly = Lm[3 * i + 1], lz = Lm[3 * i + 2];
        const float ln = fmaxf(sqrtf((lx * lx + ly * ly) + lz * lz), 1e-8f);
        r.a[3 + i] = (ux * (lx / ln) + uy * (ly / ln)) + uz * (lz / ln);
    }
    // smooth cutoff (utils/windowing.py:21-29, p = 6), times (d < r_max)
    const float u = dist / r_max;
    const float u2 = u * u, u6 = u2 * u2 * u2;
    const float w = 1.0f - 28.0f * u6 + 48.0f * u6 * u - 21.0f * u6 * u2;
    r.window = (valid && dist < r_max) ? w : 0.0f;
    return r;
}

template <int C, int D, int NCB, int OCC>
__global__ __launch_bounds__(256, OCC) void edge_kernel(
    const float* __restrict__ nbr_dir,   // [N][k][3]
    const float* __restrict__ nbr_dist,  // [N][k]
    const int32_t* __restrict__ deg,     // [N]
    const int32_t* __restrict__ batch,   // [N] crystal of node
    const float* __restrict__ lattice,   // [B][9]
    const float* __restrict__ ori,       // [16][3]
    const float* __restrict__ stream,    // w1p | w2p | wkp[0..L-1], contiguous fragment stream
    const float* __restrict__ b1, const float* __restrict__ b2, float r_max, int N, int k, int L,
    float* __restrict__ kbuf)            // [L][N*k*16][C]
{
    constexpr int TC = C / 32, TD = D / 32, TM = ARREAU_MONO_PAD / 32;
    constexpr int G1 = TM * 4, G2 = TC * 4, G3 = TD * 4;          // k-groups per output tile
    constexpr int F1 = TC * G1, F2 = TD * G2, F3 = TC * G3;       // groups per matrix
    static_assert(F1 % ARREAU_PF == 0 && F2 % ARREAU_PF == 0 && F3 % ARREAU_PF == 0, "ring phase must repeat");
    const int lane = threadIdx.x & 63;
    const int h = lane >> 5, j = lane & 31;
    constexpr int SLOTS = 2 * NCB;  // edge slots per wave
    const int tiles_per_node = (k + SLOTS - 1) / SLOTS;
    const long long wt = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int node = (int)(wt / tiles_per_node);
    if (node >= N) return;
    const int slot0 = (int)(wt - (long long)node * tiles_per_node) * SLOTS;
    const int nd = min(deg[node], k);
    if (slot0 >= nd) return;  // wave-uniform: all four slots of this wave are empty
    const int o = j & 15;

    // start the weight stream before anything else
    const float* sp = stream + lane * 4;
    f32x4 ring[ARREAU_PF];
#pragma unroll
    for (int i = 0; i < ARREAU_PF; ++i) ring[i] = *reinterpret_cast<const f32x4*>(sp + (size_t)i * 256);

    const float* Lm = lattice + 9 * (size_t)batch[node];
    int slot[NCB];
    size_t row[NCB];
    EdgeRow er[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        slot[cb] = slot0 + 2 * cb + (j >> 4);
        const int sc = min(slot[cb], k - 1);  // padding rows (k not a multiple of 4) read a valid slot
        er[cb] = edge_row(nbr_dir, nbr_dist, ori, Lm, (size_t)node * k + sc, o, r_max, slot[cb] < nd);
        row[cb] = ((size_t)node * k + sc) * 16 + o;
    }

    // ---- layer 1: h = GELU(W1f . mono + b1)   (83 distinct monomials, canonical order of model.hip) ----
    f32x16 acc1[TC][NCB];
    {
        static_assert(TM == 3, "three monomial tiles");
        f32x16 bm[TM][NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            bm[0][cb] = mono_tile<0>(er[cb].a, h, std::make_integer_sequence<int, 16>{});
            bm[1][cb] = mono_tile<1>(er[cb].a, h, std::make_integer_sequence<int, 16>{});
            bm[2][cb] = mono_tile<2>(er[cb].a, h, std::make_integer_sequence<int, 16>{});
        }
#pragma unroll
        for (int u = 0; u < TC; ++u) {
            f32x16 acc[NCB];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[cb] = arreau_bias_tile(b1, u, h);
            __builtin_amdgcn_sched_barrier(0);
            arreau_stream_tile<G1, TM, NCB>(acc, ring, sp, u * G1, bm);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc1[u][cb][r] = arreau_gelu(acc[cb][r]);
        }
    }

    // ---- layer 2: basis = GELU(W2 . h + b2) * window ---------------------------------------------------
    f32x16 acc2[TD][NCB];
    {
        const float* region = sp + (size_t)F1 * 256;
#pragma unroll
        for (int u = 0; u < TD; ++u) {
            f32x16 acc[NCB];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[cb] = arreau_bias_tile(b2, u, h);
            __builtin_amdgcn_sched_barrier(0);
            arreau_stream_tile<G2, TC, NCB>(acc, ring, region, u * G2, acc1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[u][cb][r] = arreau_gelu(acc[cb][r]) * er[cb].window;
        }
    }

    // ---- per layer: kernel_l = Wk_l . basis  (conv.py:110), written tile by tile -------------------------
    const size_t layer_stride = (size_t)N * k * 16 * C;
    for (int l = 0; l < L; ++l) {
        const float* region = sp + (size_t)(F1 + F2 + (size_t)l * F3) * 256;
#pragma unroll
        for (int u = 0; u < TC; ++u) {
            f32x16 acc[NCB];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[cb][r] = 0.0f;
            __builtin_amdgcn_sched_barrier(0);
            arreau_stream_tile<G3, TD, NCB>(acc, ring, region, u * G3, acc2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                if (slot[cb] < k) {
                    float* dst = kbuf + (size_t)l * layer_stride + row[cb] * C + 4 * h + 32 * u;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = {acc[cb][4 * q], acc[cb][4 * q + 1], acc[cb][4 * q + 2], acc[cb][4 * q + 3]};
                        *reinterpret_cast<f32x4*>(dst + 8 * q) = v;
                    }
                }
            }
        }
    }
}

template <int NCB, int OCC>
static void launch_variant(const arreau_model* m, const float* dir, const float* dist, const int32_t* deg,
                           const int32_t* batch, const float* lattice, int N, float* kbuf, hipStream_t s) {
    constexpr int SLOTS = 2 * NCB;
    const long long wave_tiles = (long long)N * ((m->k + SLOTS - 1) / SLOTS);
    ARREAU_LAUNCH((edge_kernel<128, 256, NCB, OCC>), dim3((unsigned)((wave_tiles + 3) / 4)), dim3(256), 0, s, dir,
                       dist, deg, batch, lattice, m->ori, m->w1p, m->b1, m->b2, m->cfg.radius, N, m->k, m->L, kbuf);
}

int arreau_launch_edge(const arreau_model* m, const float* dir, const float* dist, const int32_t* deg,
                       const int32_t* batch, const float* lattice, int N, float* kbuf, hipStream_t s, NodeRange r) {
    if (N == 0) return ARREAU_OK;
    if (!(m->C == 128 && m->D == 256)) {
        arreau_set_error("edge kernel: unsupported (hidden_dim, basis_dim)");
        return ARREAU_EINVAL;
    }
    // variant switch (default 4; the fp32-MFMA kernels stay for A/B runs and as the numerical cross-check):
    // 0 (also 1, 2) = fp32 MFMA, 32 rows/wave, 2 waves/SIMD; 3 = bf16x6 split-precision kernel (edge_bf16.hip);
    // 4 = fp16x3 split-precision kernel (edge_f16.hip; falls back to 3 when a weight does not fit fp16)
    const int variant = m->edge_variant;
    if (variant == 4 && m->f16_ok) {
        m->ran_edge = 4;
        return arreau_launch_edge_f16x3(m, dir, dist, deg, batch, lattice, N, kbuf, s, r);
    }
    if (r.n0 != 0 || (r.n1 >= 0 && r.n1 != N)) {
        arreau_set_error("edge kernel: range launches are implemented for the fp16x3 kernel only");
        return ARREAU_EINVAL;
    }
    if (variant >= 3) {
        m->ran_edge = 3;
        return arreau_launch_edge_bf16x6(m, dir, dist, deg, batch, lattice, N, kbuf, s);
    }
    m->ran_edge = 0;  // (variants 1 and 2, other wave geometries of the same fp32-MFMA kernel, were removed in round 5)
    launch_variant<1, 2>(m, dir, dist, deg, batch, lattice, N, kbuf, s);
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}
