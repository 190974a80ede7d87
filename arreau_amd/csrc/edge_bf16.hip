// K2+K3, split-precision variant: the same fused edge pipeline as edge.hip (pair invariants ->
// monomials -> basis MLP -> window -> L kernel projections, everything in registers), but every fp32
// product is evaluated as six bf16 products on v_mfma_f32_32x32x16_bf16:
//     a = a1 + a2 + a3,  b = b1 + b2 + b3   (exact 8+8+8-bit truncation splits)
//     a*b ~= a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1          (dropped terms <= 3 * 2^-24 |ab|)
// with fp32 accumulation, i.e. fp32-grade results (measured: same 1e-6-level distance to the fp64
// reference as the fp32-MFMA kernel) at 6 x 32 cycles per 32x32x16 block instead of 8 x 64 cycles:
// 2.67x fewer matrix-pipe cycles than v_mfma_f32_32x32x2_f32.
//
// Weights: host-split into three bf16 planes and chunked per output tile (model.hip,
// pack_linear_bf16x3).  A workgroup (4 waves = the 8 edge slots of one receiver, one wave per SIMD,
// 512 registers) shares each chunk through LDS: while the waves run the MFMAs of chunk c from one LDS
// buffer, each wave fetches its quarter of chunk c+1 from L2 into registers and writes it to the other
// buffer afterwards ("issue early / write late"), one barrier per chunk.  Activations never leave
// registers: an accumulator tile is GELU'd, split into its three bf16 planes in place and becomes the B
// operand of the next layer (registers 8s..8s+7 of a 32x32 accumulator are the fragment of k-step s).
#include <stdlib.h>
#include <utility>

#include "internal.h"
#include "bf16x6.h"

// ---- compile-time monomial table (same canonical order as fold_poly_weight in model.hip) -------------
struct MonoIdxB { int n, i, j, k; };
__host__ __device__ constexpr MonoIdxB mono_idx_b(int f) {
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
    return {0, 0, 0, 0};
}
template <int F>
__device__ __forceinline__ float mono_at_b(const float (&a)[6]) {
    constexpr MonoIdxB m = mono_idx_b(F);
    if constexpr (m.n == 1) return a[m.i];
    else if constexpr (m.n == 2) return a[m.i] * a[m.j];
    else if constexpr (m.n == 3) return (a[m.i] * a[m.j]) * a[m.k];
    else return 0.0f;
}
// "accumulator-layout" tile of monomials: register r of tile T holds feature 32T + (r&3) + 8(r>>2) + 4h
template <int T, int... R>
__device__ __forceinline__ f32x16 mono_tile_b(const float (&a)[6], int h, std::integer_sequence<int, R...>) {
    f32x16 v;
    ((v[R] = h ? mono_at_b<32 * T + 8 * (R >> 2) + (R & 3) + 4>(a) : mono_at_b<32 * T + 8 * (R >> 2) + (R & 3)>(a)), ...);
    return v;
}

template <int C, int D>
__global__ __launch_bounds__(256, 1) void edge_kernel_bf16x6(
    const float* __restrict__ nbr_dir,   // [N][k][3]
    const float* __restrict__ nbr_dist,  // [N][k]
    const int32_t* __restrict__ deg,     // [N]
    const int32_t* __restrict__ batch,   // [N] crystal of node
    const float* __restrict__ lattice,   // [B][9]
    const float* __restrict__ ori,       // [16][3]
    const u32x4* __restrict__ stream,    // bf16x3 chunks: w1 (C/32 chunks) | w2 (D/32) | wk_l (L * C/32)
    const float* __restrict__ b1, const float* __restrict__ b2, float r_max, int N, int k, int L,
    float* __restrict__ kbuf, int dbg)   // [L][N*k*16][C]
{
    constexpr int TC = C / 32, TD = D / 32, TM = ARREAU_MONO_PAD / 32;
    constexpr int NF1 = TM * 6, NF2 = TC * 6, NF3 = TD * 6;  // fragments per chunk
    static_assert(NF3 <= EB_MAX_FRAGS && TM == 3, "chunk size");
    static_assert(((TC + TD) & 1) == 0, "buffer parity at the start of the projection loop");
    __shared__ u32x4 lds[2][EB_MAX_FRAGS * 64];  // 2 x 48 KiB
    __shared__ __attribute__((aligned(16))) float otile[4][32 * 36];  // per-wave transpose pad for the stores

    const int node = blockIdx.x;
    const int nd = min(deg[node], k);
    if (nd == 0) return;  // workgroup-uniform: isolated atom, nothing to write
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int h = lane >> 5, j = lane & 31;
    const bool active = 2 * wave < nd;  // wave-uniform; inactive waves still stage weights and meet the barriers
    const int slot = 2 * wave + (j >> 4);
    const int o = j & 15;
    const int slot_c = min(slot, k - 1);

    // first chunk on its way while the row attributes are computed
    u32x4 st[EB_STAGE];
    const u32x4* chunk = stream;
    stage_load<NF1>(st, chunk, wave, lane);

    // ---- per-row attributes (transforms/invariants.py:82-88) ------------------------------------------
    float a[6], window;
    {
        const size_t e = (size_t)node * k + slot_c;
        const float dx = nbr_dir[3 * e + 0], dy = nbr_dir[3 * e + 1], dz = nbr_dir[3 * e + 2];
        const float dist = nbr_dist[e];
        const float ox = ori[3 * o + 0], oy = ori[3 * o + 1], oz = ori[3 * o + 2];
        a[0] = (dx * ox + dy * oy) + dz * oz;
        const float rx = dx - a[0] * ox, ry = dy - a[0] * oy, rz = dz - a[0] * oz;
        a[1] = sqrtf((rx * rx + ry * ry) + rz * rz);
        a[2] = dist;
        const float* Lm = lattice + 9 * (size_t)batch[node];
        const float dn = fmaxf(sqrtf((dx * dx + dy * dy) + dz * dz), 1e-8f);
        const float ux = dx / dn, uy = dy / dn, uz = dz / dn;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float lx = Lm[3 * i], ly = Lm[3 * i + 1], lz = Lm[3 * i + 2];
            const float ln = fmaxf(sqrtf((lx * lx + ly * ly) + lz * lz), 1e-8f);
            a[3 + i] = (ux * (lx / ln) + uy * (ly / ln)) + uz * (lz / ln);
        }
        const float u = dist / r_max;
        const float u2 = u * u, u6 = u2 * u2 * u2;
        const float w = 1.0f - 28.0f * u6 + 48.0f * u6 * u - 21.0f * u6 * u2;
        window = (slot < nd && dist < r_max) ? w : 0.0f;
    }
    Planes bm[TM];
    bm[0] = split_tile(mono_tile_b<0>(a, h, std::make_integer_sequence<int, 16>{}));
    bm[1] = split_tile(mono_tile_b<1>(a, h, std::make_integer_sequence<int, 16>{}));
    bm[2] = split_tile(mono_tile_b<2>(a, h, std::make_integer_sequence<int, 16>{}));

    stage_store<NF1>(st, lds[0], wave, lane);
    __syncthreads();

    // The GELU + bf16 split of output tile u-1 (VALU) is issued together with the MFMAs of tile u: the MFMA
    // chain leaves ~3/4 of the issue slots free, so with one wave per SIMD this is where the VALU work hides.
    // ---- layer 1: h = GELU(W1f . mono + b1) -------------------------------------------------------------
    Planes h1[TC];
    f32x16 pend;
#pragma unroll
    for (int u = 0; u < TC; ++u) {
        const int cur = u & 1;
        chunk += (size_t)NF1 * 64;
        if (u + 1 < TC) stage_load<NF1>(st, chunk, wave, lane); else stage_load<NF2>(st, chunk, wave, lane);
        if (active) {
            f32x16 acc = arreau_bias_tile(b1, u, h);
            mma_range<TM, 0, 2 * TM>(acc, lds[cur], bm, lane);
            if (u > 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) pend[r] = gelu_fast(pend[r]);
                h1[u - 1] = split_tile(pend);
            }
            pend = acc;
        }
        if (u + 1 < TC) stage_store<NF1>(st, lds[cur ^ 1], wave, lane); else stage_store<NF2>(st, lds[cur ^ 1], wave, lane);
        __syncthreads();
    }
    // ---- layer 2: basis = GELU(W2 . h + b2) * window ------------------------------------------------------
    Planes basis[TD];
    if (active) {
#pragma unroll
        for (int r = 0; r < 16; ++r) pend[r] = gelu_fast(pend[r]);
        h1[TC - 1] = split_tile(pend);
    }
#pragma unroll
    for (int u = 0; u < TD; ++u) {
        const int cur = (TC + u) & 1;
        chunk += (size_t)NF2 * 64;
        if (u + 1 < TD) stage_load<NF2>(st, chunk, wave, lane); else stage_load<NF3>(st, chunk, wave, lane);
        if (active) {
            f32x16 acc = arreau_bias_tile(b2, u, h);
            mma_range<TC, 0, 2 * TC>(acc, lds[cur], h1, lane);
            if (u > 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) pend[r] = gelu_fast(pend[r]) * window;
                basis[u - 1] = split_tile(pend);
            }
            pend = acc;
        }
        if (u + 1 < TD) stage_store<NF2>(st, lds[cur ^ 1], wave, lane); else stage_store<NF3>(st, lds[cur ^ 1], wave, lane);
        __syncthreads();
    }
    if (active) {
#pragma unroll
        for (int r = 0; r < 16; ++r) pend[r] = gelu_fast(pend[r]) * window;
        basis[TD - 1] = split_tile(pend);
    }
    // ---- per layer: kernel_l = Wk_l . basis  (conv.py:110), one output tile per chunk ---------------------
    // The tile's stores are issued AFTER the hand-over barrier: vmcnt is in-order, so stores issued before the
    // wait for the staged fragments would put their HBM round trip on every chunk's critical path.
    const size_t layer_stride = (size_t)N * k * 16 * C;
    const size_t row0 = ((size_t)node * k + 2 * wave) * 16;  // first K row of this wave's 32-row tile
    const int nchunks = L * TC;
    int cur = 0;  // (TC + TD) is even
    f32x16 done;  // finished tile of the previous chunk, stored one chunk late (see below)
    float* pad = otile[wave];
    // Transpose a finished 32x32 tile through a wave-private LDS pad so that every store instruction writes
    // whole 128-byte lines (8 lanes per row) instead of 64 scattered 16-byte pieces.
    auto store_tile = [&](const f32x16& t, int cidx_done) {
        const int l = cidx_done / TC, u = cidx_done - l * TC;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = {t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]};
            *reinterpret_cast<f32x4*>(&pad[j * 36 + 8 * q + 4 * h]) = v;
        }
        float* dst = kbuf + (size_t)l * layer_stride + row0 * C + 32 * u + 4 * (lane & 7);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 8 * i + (lane >> 3);  // row of the wave tile: slot 2*wave + (r >> 4), orientation r & 15
            const f32x4 v = *reinterpret_cast<const f32x4*>(&pad[r * 36 + 4 * (lane & 7)]);
            if (2 * wave + (r >> 4) < nd) *reinterpret_cast<f32x4*>(dst + (size_t)r * C) = v;
        }
    };
#pragma unroll 1
    for (int cidx = 0; cidx < nchunks; ++cidx) {
        chunk += (size_t)NF3 * 64;
        const bool more = cidx + 1 < nchunks;  // workgroup-uniform
        // vmcnt retires in order: the next chunk's fragment loads are issued BEFORE the previous tile's stores,
        // so waiting for the fragments (mid-chunk) never waits for the stores' HBM round trip.
        if (more && !(dbg & 1)) stage_load<NF3>(st, chunk, wave, lane);
        if (active && cidx > 0 && !(dbg & 4)) store_tile(done, cidx - 1);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        // the next chunk's fragments go to the idle buffer in the middle of this chunk's MFMA stream, so the
        // LDS write burst (48 KiB per workgroup) overlaps matrix work instead of sitting in front of the barrier
        if (active) mma_range<TD, 0, TD>(acc, lds[cur], basis, lane);
        if (more && !(dbg & 1)) stage_store<NF3>(st, lds[cur ^ 1], wave, lane);
        if (active) mma_range<TD, TD, 2 * TD>(acc, lds[cur], basis, lane);
        if (!(dbg & 2)) __syncthreads();
        done = acc;
        cur ^= 1;
    }
    if (active && !(dbg & 4)) store_tile(done, nchunks - 1);
}

int arreau_launch_edge_bf16x6(const arreau_model* m, const float* dir, const float* dist, const int32_t* deg,
                              const int32_t* batch, const float* lattice, int N, float* kbuf, hipStream_t s) {
    if (N == 0) return ARREAU_OK;
    if (!(m->C == 128 && m->D == 256 && m->k <= 8)) {
        arreau_set_error("edge kernel (bf16x6): unsupported (hidden_dim, basis_dim, max_neighbors)");
        return ARREAU_EINVAL;
    }
    static const int dbg = [] { const char* e = getenv("ARREAU_EDGE_DBG"); return e ? atoi(e) : 0; }();
    ARREAU_LAUNCH((edge_kernel_bf16x6<128, 256>), dim3(N), dim3(256), 0, s, dir, dist, deg, batch, lattice, m->ori,
                       reinterpret_cast<const u32x4*>(m->edge_bf16), m->b1, m->b2, m->cfg.radius, N, m->k, m->L, kbuf, dbg);
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}
