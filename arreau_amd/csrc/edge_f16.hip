// K2+K3, fp16x3 split-precision variant (default): the fused edge pipeline (pair invariants -> monomials ->
// basis MLP -> window -> L kernel projections, activations in registers) with every fp32 product evaluated
// as three fp16 MFMA products (f16x3.h): half the matrix-pipe work and two thirds of the operand bytes of
// the bf16x6 kernel (edge_bf16.hip), 16 registers per activation tile instead of 24 -- which is what lets
// two waves share a SIMD (<= 256 registers each): their vector phases (GELU, splits, attribute set-up) overlap
// each other, and conversions / transcendentals / LDS and store traffic overlap the partner's MFMA stream (fp32
// add / mul / fma do not: tools/exp/coexec.hip).
//
// Workgroup = 8 waves = the 16 edge slots of two receivers (2 waves per SIMD).  Weight chunks (one 32-row
// output tile: 12 / 16 / 32 KiB) are shared through a three-slot LDS ring filled by LDS-DMA
// (global_load_lds_dwordx4): no staging registers, no ds_write, one barrier per chunk in the MIDDLE of the chunk's
// MFMA stream ("ring protocol" below), the copy's latency covered by a whole chunk of matrix work.
#include <stdlib.h>
#include <utility>

#include "f16x3.h"
#include "internal.h"

// ---- compile-time monomial table (same canonical order as fold_poly_weight in model.hip) -------------
struct MonoIdxH { int n, i, j, k; };
__host__ __device__ constexpr MonoIdxH mono_idx_h(int f) {
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
__device__ __forceinline__ float mono_at_h(const float (&a)[6]) {
    constexpr MonoIdxH m = mono_idx_h(F);
    if constexpr (m.n == 1) return a[m.i];
    else if constexpr (m.n == 2) return a[m.i] * a[m.j];
    else if constexpr (m.n == 3) return (a[m.i] * a[m.j]) * a[m.k];
    else return 0.0f;
}
// "accumulator-layout" tile of monomials: register r of tile T holds feature 32T + (r&3) + 8(r>>2) + 4h
template <int T, int... R>
__device__ __forceinline__ f32x16 mono_tile_h(const float (&a)[6], int h, std::integer_sequence<int, R...>) {
    f32x16 v;
    ((v[R] = h ? mono_at_h<32 * T + 8 * (R >> 2) + (R & 3) + 4>(a) : mono_at_h<32 * T + 8 * (R >> 2) + (R & 3)>(a)), ...);
    return v;
}


// Phase timing (tools/edge_timing.py; compiled in only with -DARREAU_EDGE_TIMING): wave 0 of every workgroup adds
// the shader-clock ticks it spent in [set-up, layer 1, layer 2, projections, tail] to these counters.
#ifdef ARREAU_EDGE_TIMING
__device__ unsigned long long arreau_edge_ticks[8];
#define EDGE_TICK(i)                            \
    do {                                        \
        const long long now_ = clock64();       \
        tacc_[i] += now_ - tick_;               \
        tick_ = now_;                           \
    } while (0)
extern "C" int arreau_debug_edge_ticks(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(arreau_edge_ticks), 64) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(arreau_edge_ticks), z, 64) != hipSuccess) return 1;
    }
    return 0;
}
#else
#define EDGE_TICK(i)
#endif

template <int C, int D, int EH_WAVES>
__global__ __launch_bounds__(64 * EH_WAVES, 2) void edge_kernel_f16x3(
    const float* __restrict__ nbr_dir,   // [N][k][3]
    const float* __restrict__ nbr_dist,  // [N][k]
    const int32_t* __restrict__ deg,     // [N]
    const int32_t* __restrict__ batch,   // [N] crystal of node
    const float* __restrict__ lattice,   // [B][9]
    const float* __restrict__ ori,       // [16][3]
    const u32x4* __restrict__ stream,    // fp16x3 chunks: w1 (C/32 chunks) | w2 (D/32) | wk_l (L * C/32)
    const float* __restrict__ b1, const float* __restrict__ b2, float r_max, int N, int k, int L,
    float* __restrict__ kbuf)            // [L][N*k*16][C]
{
    constexpr int TC = C / 32, TD = D / 32, TM = ARREAU_MONO_PAD / 32;
    constexpr int NF1 = TM * 4, NF2 = TC * 4, NF3 = TD * 4;  // 1 KiB fragments per chunk: 12, 16, 32
    static_assert(TM == 3, "chunk geometry");
    __shared__ u32x4 lds[3][NF3 * 64];                                        // 3 x 32 KiB ring of weight chunks
    __shared__ __attribute__((aligned(16))) float otile[EH_WAVES][32 * 36];   // per-wave transpose pad for the stores
    __shared__ __attribute__((aligned(16))) float bias_s[C + D];              // b1 | b2 (no global loads beside the DMA)

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int h = lane >> 5, j = lane & 31;
    const int node_raw = (EH_WAVES / 4) * blockIdx.x + (wave >> 2);
    const int node = min(node_raw, N - 1);
    const int wn = wave & 3;                       // wave within its node: slots 2wn, 2wn+1
    const int nd = node_raw < N ? min(deg[node], k) : 0;
    const bool active = 2 * wn < nd;               // wave-uniform; idle waves still stage weights and meet the barriers
    const int slot = 2 * wn + (j >> 4);
    const int o = j & 15;
    const int slot_c = min(slot, k - 1);

#ifdef ARREAU_EDGE_TIMING
    long long tick_ = clock64();
    long long tacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    const u32x4* dma_src = stream;  // next chunk to copy
    dma_chunk<NF1, EH_WAVES>(dma_src, lds[0], wave, lane);
    dma_src += (size_t)NF1 * 64;
    dma_chunk<NF1, EH_WAVES>(dma_src, lds[1], wave, lane);
    dma_src += (size_t)NF1 * 64;
    if (threadIdx.x < C + D) bias_s[threadIdx.x] = threadIdx.x < C ? b1[threadIdx.x] : b2[threadIdx.x - C];

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
    Planes2 bm[TM];
    bm[0] = split_tile2(mono_tile_h<0>(a, h, std::make_integer_sequence<int, 16>{}));
    bm[1] = split_tile2(mono_tile_h<1>(a, h, std::make_integer_sequence<int, 16>{}));
    bm[2] = split_tile2(mono_tile_h<2>(a, h, std::make_integer_sequence<int, 16>{}));

    dma_wait();
    __syncthreads();

    EDGE_TICK(0);
    // Ring protocol.  Chunk q lives in slot q % 3.  The one barrier per chunk sits in the MIDDLE of the chunk's MFMA
    // stream (SYNC_q), not at its end: by then every wave has left chunk q-1 (so slot (q+2) % 3 is free) and has
    // drained its share of the copy of chunk q+1 (issued at SYNC_{q-1}, a whole chunk of matrix work earlier), so
    // after the barrier chunk q+1 is complete for everybody and the copy of chunk q+2 can start.  A wave therefore
    // runs from the end of one tile (accumulator drain, GELU / plane split or tile store) straight into the next
    // tile's MFMAs without meeting anybody, instead of all waves draining the matrix pipe at a common end-of-chunk
    // barrier; the fragment prefetch (MmaStream2) runs through the barrier.
    unsigned dma_off[NF3 / EH_WAVES];
    dma_offsets<NF3, EH_WAVES>(dma_off, wave, lane);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)&lds[0][0]) + 1024u * wave;
    constexpr unsigned SLOT_BYTES = NF3 * 1024u;
    unsigned (&dma_off2)[NF2 / EH_WAVES] = reinterpret_cast<unsigned(&)[NF2 / EH_WAVES]>(dma_off);  // same offsets, first two
    // ---- layer 1: h = GELU(W1f . mono + b1) -------------------------------------------------------------
    Planes2 h1[TC];
#pragma unroll
    for (int u = 0; u < TC; ++u) {
        f32x16 acc, cross;
        MmaStream2<TM, 2 * TM> ms;
        if (active) {
            acc = arreau_bias_tile(bias_s, u, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) cross[r] = 0.f;
            ms.start(lds[u % 3], lane);
            ms.template run<0, TM>(acc, cross, bm);
        }
        dma_wait();
        __syncthreads();
        if (u + 2 < TC) { dma_chunk<NF1, EH_WAVES>(dma_src, lds[(u + 2) % 3], wave, lane); dma_src += (size_t)NF1 * 64; }
        else { dma_chunk_lean<NF2, EH_WAVES>(dma_src, dma_off2, lds0 + ((u + 2) % 3) * SLOT_BYTES); dma_src += (size_t)NF2 * 64; }
        if (active) {
            ms.template run<TM, 2 * TM>(acc, cross, bm);
            h1[u] = gelu_split_tile2(acc, cross, 1.0f);
        }
    }
    EDGE_TICK(1);
    // ---- layer 2: basis = GELU(W2 . h + b2) * window ------------------------------------------------------
    Planes2 basis[TD];
#pragma unroll
    for (int u = 0; u < TD; ++u) {
        constexpr int Q0 = TC;
        f32x16 acc, cross;
        MmaStream2<TC, 2 * TC> ms;
        if (active) {
            acc = arreau_bias_tile(bias_s + C, u, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) cross[r] = 0.f;
            ms.start(lds[(Q0 + u) % 3], lane);
            ms.template run<0, TC>(acc, cross, h1);
        }
        dma_wait();
        __syncthreads();
        if (u + 2 < TD) { dma_chunk_lean<NF2, EH_WAVES>(dma_src, dma_off2, lds0 + ((Q0 + u + 2) % 3) * SLOT_BYTES); dma_src += (size_t)NF2 * 64; }
        else if (u + 2 - TD < L * TC) { dma_chunk_lean<NF3, EH_WAVES>(dma_src, dma_off, lds0 + ((Q0 + u + 2) % 3) * SLOT_BYTES); dma_src += (size_t)NF3 * 64; }
        if (active) {
            ms.template run<TC, 2 * TC>(acc, cross, h1);
            basis[u] = gelu_split_tile2(acc, cross, window);
        }
    }
    EDGE_TICK(2);
    // ---- per layer: kernel_l = Wk_l . basis  (conv.py:110), one output tile per chunk ---------------------
    const size_t layer_stride = (size_t)N * k * 16 * C;
    const size_t row0 = ((size_t)node * k + 2 * wn) * 16;  // first K row of this wave's 32-row tile
    const int nchunks = L * TC;
    int sl = (TC + TD) % 3;  // ring slot of the current chunk
    float* pad = otile[wave];
    const bool full = 2 * wn + 1 < k;  // wave-uniform: the wave's second slot exists
    // A finished 32x32 tile goes to HBM through a wave-private LDS pad (transposed, so that every store instruction
    // writes whole 128-byte lines: 8 lanes per row), in two steps that sit half a chunk apart in the instruction
    // stream: pad_write right behind the tile's last MFMA, pad_store two k-steps into the next tile's MFMAs.
    auto pad_write = [&](const f32x16& t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = {t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]};
            *reinterpret_cast<f32x4*>(&pad[j * 36 + 8 * q + 4 * h]) = v;
        }
    };
    // Not predicated on the degree: a slot beyond it gets the zeros its window produced (those rows of the K buffer
    // are never read), so a wave issues a FIXED number of stores per tile (4, or 2 for the wave whose second slot does
    // not exist when k is odd) -- which the counted wait at SYNC relies on.
    // Store addressing: wave-uniform byte base of tile (layer l, column tile u) advanced tile by tile + four per-lane
    // 32-bit offsets computed once (global_store ... saddr form: no 64-bit vector address arithmetic in the loop).
    unsigned st_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) st_off[i] = 4u * ((8 * i + (lane >> 3)) * C + 4 * (lane & 7));  // row 8i + lane/8 of the wave tile
    auto pad_store = [&](const char* tile_base /* wave-uniform */) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 8 * i + (lane >> 3);  // slot 2*wn + (r >> 4), orientation r & 15
            const f32x4 v = *reinterpret_cast<const f32x4*>(&pad[r * 36 + 4 * (lane & 7)]);
            if (i < 2 || full) *reinterpret_cast<f32x4*>(const_cast<char*>(tile_base) + st_off[i]) = v;
        }
    };
    // X = k-step at which the chunk's barrier is taken (mid-chunk).  Taking it at different points in the two halves
    // of the workgroup (SIMD partners a half chunk apart) was measured to make no difference: fp32 vector work and
    // MFMAs of SIMD partners serialise anyway, so both halves use the same point.
    // At SYNC the wave's queue holds, oldest first, its 4 DMA copies of the next chunk and -- from the second chunk
    // on -- the stores of the previous tile: the counted wait retires the copies and leaves the stores in flight.
    const char* tile0 = reinterpret_cast<const char*>(kbuf + row0 * C);  // (layer 0, column tile 0) of this wave's rows
    auto proj_loop = [&](auto xtag) {
        constexpr int X = decltype(xtag)::value;
        static_assert(X >= 2 && X < 2 * TD, "barrier position");
        const char* tile_base = tile0;  // tile of the PREVIOUS chunk (the one pad_store writes)
        int u_prev = 0;
#pragma unroll 1
        for (int cidx = 0; cidx < nchunks; ++cidx) {
            const u32x4* buf = lds[sl];
            const unsigned free_slot = lds0 + (sl == 0 ? 2u : (unsigned)sl - 1u) * SLOT_BYTES;  // (sl + 2) % 3
            f32x16 acc, cross;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[r] = 0.0f; cross[r] = 0.0f; }
            MmaStream2<TD, 2 * TD> ms;
            ms.start(buf, lane);
            ms.template run<0, 2>(acc, cross, basis);
            if (cidx > 0) {
                pad_store(tile_base);
                // next tile: 32 columns on, or the first column tile of the next layer
                tile_base += (++u_prev == TC) ? (u_prev = 0, (ptrdiff_t)layer_stride * 4 - (TC - 1) * 128) : 128;
            }
            ms.template run<2, X>(acc, cross, basis);
            if (cidx == 0) dma_wait();
            else if (full) dma_wait_but<4>();
            else dma_wait_but<2>();
            __syncthreads();
            if (cidx + 2 < nchunks) dma_chunk_lean<NF3, EH_WAVES>(dma_src, dma_off, free_slot);
            dma_src += (size_t)NF3 * 64;
            ms.template run<X, 2 * TD>(acc, cross, basis);
            pad_write(fold_cross(acc, cross));
            sl = sl == 2 ? 0 : sl + 1;
        }
        pad_store(tile_base);
    };
    if (!active) {  // no slots: keep the ring turning
#pragma unroll 1
        for (int cidx = 0; cidx < nchunks; ++cidx) {
            const unsigned free_slot = lds0 + (sl == 0 ? 2u : (unsigned)sl - 1u) * SLOT_BYTES;
            dma_wait();
            __syncthreads();
            if (cidx + 2 < nchunks) dma_chunk_lean<NF3, EH_WAVES>(dma_src, dma_off, free_slot);
            dma_src += (size_t)NF3 * 64;
            sl = sl == 2 ? 0 : sl + 1;
        }
    } else {
        proj_loop(std::integral_constant<int, TD>{});
    }
    EDGE_TICK(3);
    EDGE_TICK(4);
#ifdef ARREAU_EDGE_TIMING
    if (threadIdx.x == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&arreau_edge_ticks[i], (unsigned long long)tacc_[i]);
#endif
}

int arreau_launch_edge_f16x3(const arreau_model* m, const float* dir, const float* dist, const int32_t* deg,
                             const int32_t* batch, const float* lattice, int N, float* kbuf, hipStream_t s) {
    if (N == 0) return ARREAU_OK;
    if (!(m->C == 128 && m->D == 256 && m->k <= 8)) {
        arreau_set_error("edge kernel (fp16x3): unsupported (hidden_dim, basis_dim, max_neighbors)");
        return ARREAU_EINVAL;
    }
    // 8 waves = two receivers per workgroup, one workgroup per CU (2 waves per SIMD from the same workgroup).
    // A 4-wave / two-workgroups-per-CU geometry was tried and dropped: it was not run-to-run reproducible.
    hipLaunchKernelGGL((edge_kernel_f16x3<128, 256, 8>), dim3((N + 1) / 2), dim3(512), 0, s, dir, dist, deg, batch,
                       lattice, m->ori, reinterpret_cast<const u32x4*>(m->edge_f16), m->b1, m->b2, m->cfg.radius, N,
                       m->k, m->L, kbuf);
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}
