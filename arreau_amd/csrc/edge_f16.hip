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
// (cache policy of the K-tile stores: default.  Measured at 256 x 20: " nt" +60 us per launch, " sc1" +25, " sc0 sc1" +10.)
#define ARREAU_K_STORE_POLICY ""
// Hazard of the hand-written K-tile stores: a VMEM store of more than 64 bits reads its data registers AFTER it has issued,
// and a VALU write to those registers too soon afterwards corrupts the stored data.  hipcc inserts the wait states for its
// own stores; it does not look inside inline asm.  Found with the 3-byte K format, whose packed data registers the
// compiler re-uses for the next store at once (non-reproducible NaNs; a plain C++ store or four wait states cure it);
// the 16-byte form carries the same wait states although its data registers were never re-used that fast.
#define ARREAU_K_STORE_TAIL "\n\ts_nop 3"
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

// PROJ = false ("basis form", round 3): the kernel stops after layer 2 and stores the windowed basis -- its two fp16 planes,
// already in the B-operand fragment order of v_mfma_f32_16x16x32_f16 -- instead of running the L kernel projections and
// storing their results: one 16 KiB block per edge slot ([k-block 0..7][plane][lane] x 16 bytes: whole 1 KiB, fully
// coalesced store instructions), 1 KiB per (edge, orientation) row written ONCE, where the K stash writes L*C*3 = 1,920
// bytes.  The projections move into the per-layer message kernel (conv_proj.hip), which streams those blocks through LDS.
// BFP8 (basis form only): the residual plane leaves as OCP fp8 e4m3 instead of fp16 -- 3 bytes per basis value, a 12 KiB
// block per slot ([8 k-blocks] x 1 KiB of hi fragments, then [8] x 512 B of lo fragments).  The basis then carries 11 + 4
// significand bits; in the fp32 oracle that changes the network outputs by no more than the 4-byte form does (both at the
// fp32 rounding floor: tools/exp/basis_precision_study.py, profiles/r03_basis_precision_study.txt), and it takes a quarter
// off the only large stream of the step.
// The residual plane of a basis value pair rounded to fp8 e4m3 and widened back (exact): what the basis form stores.  The
// kernels that project IN PLACE (PROJ = true: small launches, the round-2 pair) apply the same rounding to their B operand
// when BFP8 is set, so that every path evaluates the same numbers -- a crystal alone (small-launch kernels) equals, bit for
// bit, the same crystal inside a large batch (basis form).
__device__ __forceinline__ unsigned round_lo_fp8(unsigned lo_pair) {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    typedef short s2_t __attribute__((ext_vector_type(2)));
    s2_t r = {0, 0};
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, __builtin_bit_cast(h2_t, lo_pair), 1.0f, false);
    return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(__builtin_bit_cast(unsigned, r), 1.0f, false));
}

// Re-layout pad (32x32 accumulator layout -> two 16-column blocks): 16-byte cell (row, slot) of a plane, row = 0 .. 31,
// slot = 0 .. 3.  Writers walk the rows lane by lane at a fixed slot, readers too: with cells in plain row-major order
// (64 bytes per row) eight consecutive lanes hit two of the eight 16-byte bank groups -- 9.2 M bank-conflict cycles per
// launch at 256 x 20 (PMC, round 3).  The slot index is therefore XOR-swizzled with bits 1-2 of the row: eight
// consecutive rows cover all eight groups for writers and readers alike.
__device__ __forceinline__ int relayout_cell(int plane, int row, int slot) { return 128 * plane + 4 * row + (slot ^ ((row >> 1) & 3)); }

template <int C, int D, int EH_WAVES, bool K3 /* K tiles as 3-byte floats (internal.h) */, bool PROJ = true, bool BFP8 = false>
__global__ __launch_bounds__(64 * EH_WAVES, 2) void edge_kernel_f16x3(
    const float* __restrict__ nbr_dir,   // [N][k][3]
    const float* __restrict__ nbr_dist,  // [N][k]
    const int32_t* __restrict__ deg,     // [N]
    const int32_t* __restrict__ batch,   // [N] crystal of node
    const float* __restrict__ lattice,   // [B][9]
    const float* __restrict__ ori,       // [16][3]
    const u32x4* __restrict__ stream,    // fp16x3 chunks: w1 (C/32 chunks) | w2 (D/32) | wk_l (L * C/32)
    const float* __restrict__ b1, const float* __restrict__ b2, float r_max, int N, int k, int L,
    float* __restrict__ kbuf,            // PROJ: [L][N*k*16][C];  basis form: [N*k][16 fragments][64 lanes] x 16 bytes
    int n0, int n1)                      // receivers of this launch: n0 .. n1-1 (N stays the batch size: it strides kbuf)
{
    constexpr int TC = C / 32, TD = D / 32, TM = ARREAU_MONO_PAD / 32;
    constexpr int NF1 = TM * 4, NF2 = TC * 4, NF3 = TD * 4;  // 1 KiB fragments per chunk: 12, 16, 32
    static_assert(TM == 3 && (TC & 1) == 0, "chunk geometry (projection tiles are walked in pairs)");
    __shared__ u32x4 lds[3][NF3 * 64];                                        // 3 x 32 KiB ring of weight chunks
    __shared__ __attribute__((aligned(16))) float otile[EH_WAVES][32 * 36];   // per-wave transpose pad for the stores
    __shared__ __attribute__((aligned(16))) float bias_s[C + D];              // b1 | b2 (no global loads beside the DMA)

    const int wave0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane0 = threadIdx.x & 63;
    const int npairs = (n1 - n0 + EH_WAVES / 4 - 1) / (EH_WAVES / 4);

#ifdef ARREAU_EDGE_TIMING
    long long tick_ = clock64();
    long long tacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long c_start_ = tick_, w_start_ = wall_clock64();  // shader clock vs the constant 100 MHz counter
#endif
    // Persistent workgroup: receiver pairs blockIdx.x, blockIdx.x + gridDim.x, ...  The weight ring keeps turning
    // across pairs (the chunk sequence simply repeats), so only the first pair waits for its first two chunks.
    dma_chunk<NF1, EH_WAVES>(stream, lds[0], wave0, lane0);
    dma_chunk<NF1, EH_WAVES>(stream + (size_t)NF1 * 64, lds[1], wave0, lane0);
    if (threadIdx.x < C + D) bias_s[threadIdx.x] = threadIdx.x < C ? b1[threadIdx.x] : b2[threadIdx.x - C];
    int sl = 0;  // ring slot of the current chunk
    constexpr unsigned SLOT_BYTES = NF3 * 1024u;
    auto slot_after = [](int s_, int n) { const int r = s_ + n; return r >= 3 ? r - 3 : r; };  // (s_ + n) % 3 for n <= 2
    const int nchunks = PROJ ? L * TC : 0;
    dma_wait();
    __syncthreads();
    // basis form: stores this wave issued after its last weight copy (they are younger than the copy in the in-order
    // queue, so the counted wait in front of a barrier leaves exactly that many operations in flight)
    int pend = 0;
    auto wait_copies = [&]() {
        if constexpr (PROJ) dma_wait();
        else {
            if (pend == 0) dma_wait();
            else if (pend == 2) dma_wait_but<2>();
            else dma_wait_but<4>();
        }
    };

    for (int pair = blockIdx.x; pair < npairs; pair += gridDim.x) {
        // Everything derived from the lane / wave index is re-derived per pair from values the optimiser cannot see
        // through: hoisted out of the pair loop those addresses and constants would stay in registers through the
        // matrix phases, and this kernel must not spill (scratch traffic would count in vmcnt).
        int lane = lane0, wave = wave0;
        asm volatile("" : "+v"(lane), "+s"(wave));
        const int h = lane >> 5, j = lane & 31;
        const int wn = wave & 3;                       // wave within its node: slots 2wn, 2wn+1
        const unsigned lane16 = 16u * lane;            // per-lane byte offset inside a 1 KiB fragment
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)&lds[0][0]) + 1024u * wave;
        const bool has_next = pair + (int)gridDim.x < npairs;  // workgroup-uniform
        const int node_raw = n0 + (EH_WAVES / 4) * pair + (wave >> 2);
        const int node = min(node_raw, n1 - 1);
        const int nd = node_raw < n1 ? min(deg[node], k) : 0;
        const bool active = 2 * wn < nd;               // wave-uniform; idle waves still copy weights and meet the barriers
        const u32x4* dma_src = stream + (size_t)2 * NF1 * 64;  // next chunk to copy: chunk 2 of this pair

        // Ring protocol.  Chunk q lives in the slot after chunk q-1's (three slots).  The one barrier per chunk sits in
        // the MIDDLE of the chunk's MFMA stream (SYNC_q), not at its end: by then every wave has left chunk q-1 (so its
        // slot is free) and has drained its share of the copy of chunk q+1 (issued at SYNC_{q-1}, a whole chunk of
        // matrix work earlier), so after the barrier chunk q+1 is complete for everybody and the copy of chunk q+2 can
        // start.  A wave therefore runs from the end of one tile (accumulator drain, GELU / plane split or tile store)
        // straight into the next tile's MFMAs without meeting anybody, instead of all waves draining the matrix pipe
        // at a common end-of-chunk barrier; the fragment prefetch (MmaStream2) runs through the barrier.
        auto copy12 = [&](auto qtag) {  // after SYNC_q of a layer-1/2 chunk (q compile-time): start the copy of chunk q+2
            constexpr int q2 = decltype(qtag)::value + 2;
            const int dst = slot_after(sl, 2);
            if constexpr (q2 < TC) {
                dma_chunk<NF1, EH_WAVES>(dma_src, lds[dst], wave, lane);
                dma_src += (size_t)NF1 * 64;
            } else if constexpr (q2 < TC + TD) {
                dma_chunk_lean<NF2, EH_WAVES>(dma_src, lane16, wave, lds0 + dst * SLOT_BYTES);
                dma_src += (size_t)NF2 * 64;
            } else if constexpr (PROJ) {  // first projection chunks (L * TC >= 2)
                dma_chunk_lean<NF3, EH_WAVES>(dma_src, lane16, wave, lds0 + dst * SLOT_BYTES);
                dma_src += (size_t)NF3 * 64;
            } else {  // basis form: the pair ends with layer 2 -- the next pair's first two chunks
                if (has_next) dma_chunk<NF1, EH_WAVES>(stream + (size_t)(q2 - (TC + TD)) * NF1 * 64, lds[dst], wave, lane);
            }
            pend = 0;
        };
        auto copy3 = [&](int cidx) {  // after SYNC of projection chunk cidx
            const int dst = slot_after(sl, 2);
            if (cidx + 2 < nchunks) dma_chunk_lean<NF3, EH_WAVES>(dma_src, lane16, wave, lds0 + dst * SLOT_BYTES);
            else if (has_next)  // the next pair's first two chunks
                dma_chunk<NF1, EH_WAVES>(stream + (size_t)(cidx + 2 - nchunks) * NF1 * 64, lds[dst], wave, lane);
            dma_src += (size_t)NF3 * 64;
        };

        if (!active) {
            // no slots: keep the ring turning.  Basis form (round 5): the stash blocks of this wave's two slots are ZEROED -- the message
            // kernel streams and projects all eight slots of a receiver and no longer selects per value: a slot beyond the degree must
            // project to an exactly zero K tile (conv_proj.hip), as the computed slots beyond the degree do through their window 0.  (Rare: receivers with fewer
            // than seven neighbours.  The stores are drained by the full waits below.)
            if constexpr (!PROJ) {
                if (node_raw < n1) {
                    constexpr unsigned SLOT = BFP8 ? 12288u : 16384u;
                    const char* blk = reinterpret_cast<const char*>(kbuf) + ((size_t)node * k + 2 * wn) * SLOT;
                    const u32x4 zero4 = {0u, 0u, 0u, 0u};
                    const int nblk = 2 * wn + 1 < k ? 2 : 1;
                    for (unsigned o = 0; o < (unsigned)nblk * SLOT; o += 1024u)
                        asm volatile("global_store_dwordx4 %0, %1, %2" ARREAU_K_STORE_TAIL : : "v"(lane16 + o), "v"(zero4), "s"(blk) : "memory");
                }
            }
            [&]<int... Q>(std::integer_sequence<int, Q...>) {
                ((dma_wait(), __syncthreads(), copy12(std::integral_constant<int, Q>{}), sl = slot_after(sl, 1)), ...);
            }(std::make_integer_sequence<int, TC + TD>{});
#pragma unroll 1
            for (int cidx = 0; cidx < nchunks; ++cidx) {
                dma_wait();
                __syncthreads();
                copy3(cidx);
                sl = slot_after(sl, 1);
            }
            EDGE_TICK(3);
            continue;
        }

        const int slot = 2 * wn + (j >> 4);
        const int o = j & 15;
        // A slot past the degree (the odd slot of this wave's pair when nd is odd; nd > 2 wn here) computes with the attributes
        // of the receiver's last edge and a zero window: whatever the caller left in the unused slots, NaN included, its stash
        // block is zeros, which conv_proj.hip's unconditional sum counts on.  (Not `slot < nd ? load : 0`: hipcc turns that into
        // four exec-masked loads in a row, +8 us per launch at 256 x 20; a mask on the loaded words measured +3 us.)
        const int slot_c = min(slot, nd - 1);
        // ---- per-row attributes (transforms/invariants.py:82-88) ------------------------------------------
        float a[6], window;
        {
            const size_t e = (size_t)node * k + slot_c;
            // slots past the degree read as zeros whatever the caller left there: their window is zero below, and a
            // non-finite attribute would turn the stash block into NaN instead of the zeros conv_proj.hip counts on
            const float dx = nbr_dir[3 * e + 0], dy = nbr_dir[3 * e + 1], dz = nbr_dir[3 * e + 2];
            const float dist = nbr_dist[e];
            const float ox = ori[3 * o + 0], oy = ori[3 * o + 1], oz = ori[3 * o + 2];
            a[0] = (dx * ox + dy * oy) + dz * oz;
            const float rx = dx - a[0] * ox, ry = dy - a[0] * oy, rz = dz - a[0] * oz;
            a[1] = sqrtf((rx * rx + ry * ry) + rz * rz);
            a[2] = dist;
            const float* Lm = lattice + 9 * (size_t)batch[node];
            // reciprocals by v_rcp_f32 (1 ulp) instead of IEEE division sequences: far inside the 1e-5 parity budget
            const float inv_dn = __builtin_amdgcn_rcpf(fmaxf(sqrtf((dx * dx + dy * dy) + dz * dz), 1e-8f));
            const float ux = dx * inv_dn, uy = dy * inv_dn, uz = dz * inv_dn;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float lx = Lm[3 * i], ly = Lm[3 * i + 1], lz = Lm[3 * i + 2];
                const float inv_ln = __builtin_amdgcn_rcpf(fmaxf(sqrtf((lx * lx + ly * ly) + lz * lz), 1e-8f));
                a[3 + i] = (ux * (lx * inv_ln) + uy * (ly * inv_ln)) + uz * (lz * inv_ln);
            }
            const float u = dist * __builtin_amdgcn_rcpf(r_max);
            const float u2 = u * u, u6 = u2 * u2 * u2;
            const float w = 1.0f - 28.0f * u6 + 48.0f * u6 * u - 21.0f * u6 * u2;
            window = (slot < nd && dist < r_max) ? w : 0.0f;
        }
        Planes2 bm[TM];
        bm[0] = split_tile2(mono_tile_h<0>(a, h, std::make_integer_sequence<int, 16>{}));
        bm[1] = split_tile2(mono_tile_h<1>(a, h, std::make_integer_sequence<int, 16>{}));
        bm[2] = split_tile2(mono_tile_h<2>(a, h, std::make_integer_sequence<int, 16>{}));
        EDGE_TICK(0);

        // `go` is always 1 but opaque to the compiler: guarding the two MFMA halves of a tile with it keeps every half
        // tile its own basic block.  Without these block boundaries the scheduler overlaps the next tile's MFMAs with
        // this tile's epilogue and the accumulators spill (measured: 140 spilled registers).  Everything a guarded
        // block produces is given a value outside it as well (bias tile / zero planes): a result left undefined on
        // the other path is turned by the optimiser into a value carried around the pair loop (250 spilled registers).
        int go = 1;
        asm volatile("" : "+s"(go));
        // Layers 2 and the projections run on 16x16x32 MFMAs (f16x3.h: higher held clock under the power limit); layer 1
        // stays on 32x32x16 (its B operand, the monomials, is computed per lane in that layout).  Each finished layer-1
        // tile is re-laid once, through the wave's LDS pad, from the (h, row j) lane layout of the 32x32 accumulators
        // into the (c = orientation, g) layout of the two 16-column blocks nb (= the wave's two edge slots): lane
        // (c, g) takes the 16 bytes that lane (h = g >> 1, j = 16 nb + c) holds for k-step s = g & 1.  The pad holds one
        // tile (2 planes x 2 KiB): a tile is written right after its epilogue and read back in front of the next
        // tile's write, a whole chunk later, so neither LDS latency is waited for.  From layer 2 on the chain stays in
        // the 16x16 layout (accumulators of the two 16-row tiles of a chunk = B operand of the next k-block).
        u32x4* pad16 = reinterpret_cast<u32x4*>(otile[wave]);
        const int c16 = lane & 15, g16 = lane >> 4;
        u32x4 h16[2][TC][2];  // layer-1 output as B operand: [column block][k-block = layer-1 tile][plane]
#pragma unroll
        for (int u = 0; u < TC; ++u)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) { h16[nb][u][0] = u32x4{0, 0, 0, 0}; h16[nb][u][1] = u32x4{0, 0, 0, 0}; }
        auto relayout_write = [&](const Planes2& pl) {
#pragma unroll
            for (int plane = 0; plane < 2; ++plane) {
                pad16[relayout_cell(plane, j, 2 * h + 0)] = pl.p[plane][0];
                pad16[relayout_cell(plane, j, 2 * h + 1)] = pl.p[plane][1];
            }
        };
        auto relayout_read = [&](int u) {
#pragma unroll
            for (int plane = 0; plane < 2; ++plane)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    h16[nb][u][plane] = pad16[relayout_cell(plane, 16 * nb + c16, g16)];
        };
        // ---- layer 1: h = GELU(W1f . mono + b1) -------------------------------------------------------------
#pragma unroll
        for (int u = 0; u < TC; ++u) {
            f32x16 acc = arreau_bias_tile(bias_s, u, h), cross;
#pragma unroll
            for (int r = 0; r < 16; ++r) cross[r] = 0.f;
            MmaStream2<TM, 2 * TM> ms;
            ms.start(lds[sl], lane);
            if (go) ms.template run<0, TM>(acc, cross, bm);
            wait_copies();
            __syncthreads();
            if (u + 2 < TC) copy12(std::integral_constant<int, 0>{});          // an NF1 chunk
            else copy12(std::integral_constant<int, TC - 2>{});               // an NF2 chunk
            if (go) {
                ms.template run<TM, 2 * TM>(acc, cross, bm);
                const Planes2 pl = gelu_split_tile2(acc, cross, 1.0f);
                if (u > 0) relayout_read(u - 1);
                relayout_write(pl);
            }
            sl = slot_after(sl, 1);
        }
        relayout_read(TC - 1);
        // the cut-off windows of this lane's two rows (slot 2 wn + nb, orientation c) sit in lanes 16 nb + c
        float win16[2];
        win16[0] = __shfl(window, c16, 64);
        win16[1] = __shfl(window, 16 + c16, 64);
        EDGE_TICK(1);
        // ---- layer 2: basis = GELU(W2 . h + b2) * window ------------------------------------------------------
        u32x4 b16[2][TD][2];  // [column block][k-block = 32 basis functions][plane]
#pragma unroll
        for (int u = 0; u < TD; ++u) {
            Acc16 acc;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const f32x4 bv4 = *reinterpret_cast<const f32x4*>(bias_s + C + 32 * u + 16 * mt + 4 * g16);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    acc.m[mt][nb] = f32x4v{bv4[0], bv4[1], bv4[2], bv4[3]};
                    acc.x[mt][nb] = f32x4v{0.f, 0.f, 0.f, 0.f};
                }
            }
            MmaStream16<TC, true> ms;  // (B = the layer-1 output: activation planes)
            ms.start(lds[sl], lane);
            if (go) ms.template run<0, TC>(acc, h16);
            wait_copies();
            __syncthreads();
            if (u + 2 < TD) copy12(std::integral_constant<int, TC>{});         // an NF2 chunk
            else if constexpr (PROJ) copy12(std::integral_constant<int, TC + TD - 2>{});  // an NF3 chunk
            else if (u + 2 == TD) copy12(std::integral_constant<int, TC + TD - 2>{});     // the next pair's chunk 0 ...
            else copy12(std::integral_constant<int, TC + TD - 1>{});                      // ... and chunk 1
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) { b16[nb][u][0] = u32x4{0, 0, 0, 0}; b16[nb][u][1] = u32x4{0, 0, 0, 0}; }
            if (go) {
                ms.template run<TC, 2 * TC>(acc, h16);
                unsigned lo8x[2][2] = {{0u, 0u}, {0u, 0u}};  // BFP8: the e4m3 residuals of this k-block, [column block][tile mt] x 4 bytes
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int pr = 0; pr < 2; ++pr) {
                            const f32x2 pre = fma2(f32x2{acc.x[mt][nb][2 * pr], acc.x[mt][nb][2 * pr + 1]}, splat2(F16X3_INV_SCALE),
                                                   f32x2{acc.m[mt][nb][2 * pr], acc.m[mt][nb][2 * pr + 1]});
                            const f32x2 gw = gelu_fast2(pre) * splat2(win16[nb]);  // |window| <= 1
                            unsigned hi, lo;
                            if constexpr (BFP8) {
                                // residual straight to e4m3 (f16x3.h: split_pair_fp8); basis form: the bytes are what is stored
                                // (lo8x[nb][mt] = the four bytes of tile mt), in place: widened back for the fp16 products
                                if (pr == 0) split_pair_fp8<false>(gw, hi, lo8x[nb][mt]);
                                else split_pair_fp8<true>(gw, hi, lo8x[nb][mt]);
                                lo = pr == 0 ? widen_lo8<false>(lo8x[nb][mt]) : widen_lo8<true>(lo8x[nb][mt]);
                            } else {
                                split_pair2<false>(gw, hi, lo);
                            }
                            b16[nb][u][0][2 * mt + pr] = hi;
                            if constexpr (PROJ || !BFP8) b16[nb][u][1][2 * mt + pr] = lo;
                        }
                }
                if constexpr (!PROJ) {
                    // the finished k-block of the basis leaves: fragment (u, plane) of slot 2 wn + nb, 1 KiB per instruction
                    // (asm stores: a FIXED count per chunk, which wait_copies() relies on; wait states behind each: the
                    // data registers of a VMEM store of more than 64 bits must not be rewritten at once)
                    const bool full_b = 2 * wn + 1 < k;  // wave-uniform: the wave's second slot exists
                    constexpr unsigned SLOT = BFP8 ? 12288u : 16384u;
                    const char* blk = reinterpret_cast<const char*>(kbuf) + ((size_t)node * k + 2 * wn) * SLOT;
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        if (nb == 1 && !full_b) continue;
                        if constexpr (BFP8) {
                            asm volatile("global_store_dwordx4 %0, %1, %2" ARREAU_K_STORE_TAIL
                                         :
                                         : "v"(lane16), "v"(b16[nb][u][0]), "s"(blk + SLOT * nb + 1024 * u)
                                         : "memory");
                            typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                            const u32x2_t lo = {lo8x[nb][0], lo8x[nb][1]};  // (tile mt = halves 4 mt .. 4 mt + 3 of the lane's eight)
                            asm volatile("global_store_dwordx2 %0, %1, %2"
                                         :
                                         : "v"(lane16 >> 1), "v"(lo), "s"(blk + SLOT * nb + 8192 + 512 * u)
                                         : "memory");
                        } else {
#pragma unroll
                            for (int plane = 0; plane < 2; ++plane)
                                asm volatile("global_store_dwordx4 %0, %1, %2" ARREAU_K_STORE_TAIL
                                             :
                                             : "v"(lane16), "v"(b16[nb][u][plane]), "s"(blk + SLOT * nb + 2048 * u + 1024 * plane)
                                             : "memory");
                        }
                    }
                    pend += full_b ? 4 : 2;
                }
            }
            sl = slot_after(sl, 1);
        }
        EDGE_TICK(2);
        if constexpr (!PROJ) continue;  // basis form: the projections run in the message kernel of each layer (conv_proj.hip)
        // ---- per layer: kernel_l = Wk_l . basis  (conv.py:110), one output tile per chunk ---------------------
        constexpr int KB = K3 ? 3 : 4;  // bytes per K value
        const size_t layer_stride = (size_t)N * k * 16 * C;
        const size_t row0 = ((size_t)node * k + 2 * wn) * 16;  // first K row of this wave's 32-row tile
        const bool full = 2 * wn + 1 < k;  // wave-uniform: the wave's second slot (column block 1) exists
        // Stores: accumulator register r of tile (mt, nb) is output column 16 mt + 4 g + r of row 16 nb + c: one
        // 16-byte store per lane and tile, 64 contiguous bytes per row and instruction (the two mt tiles complete the
        // row's 128-byte line).  Not predicated on the degree: a slot beyond it gets the zeros its window produced
        // (those rows of the K buffer are never read), so a wave issues a FIXED number of stores per tile (4, or 2 for
        // the wave whose second slot does not exist when k is odd) -- which the counted wait at SYNC relies on.
        const unsigned st_off = (unsigned)KB * (c16 * C + 4 * g16);
        auto store_tile16 = [&](const Acc16& a, const char* tile_base /* wave-uniform */) {
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    f32x4v v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaf(a.x[mt][nb][r], F16X3_INV_SCALE, a.m[mt][nb][r]);
                    // asm store: the counted wait at SYNC assumes exactly 4 (or 2) store instructions per tile
                    // (scalar base + one per-lane 32-bit offset + immediate: no 64-bit vector address arithmetic)
                    if (nb == 0 || full) {
                        if constexpr (K3) {
                            const u32x3_k d = arreau_pack_k3(v[0], v[1], v[2], v[3]);
                            asm volatile("global_store_dwordx3 %0, %1, %2 offset:%3" ARREAU_K_STORE_POLICY ARREAU_K_STORE_TAIL
                                         :
                                         : "v"(st_off), "v"(d), "s"(tile_base + 16 * KB * nb * C), "n"(16 * 3 * mt)
                                         : "memory");
                        } else {
                            asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3" ARREAU_K_STORE_POLICY ARREAU_K_STORE_TAIL
                                         :
                                         : "v"(st_off), "v"(v), "s"(tile_base + 64 * nb * C), "n"(64 * mt)
                                         : "memory");
                        }
                    }
                }
        };
        // ST = step (of 2 TD) at which the chunk's barrier is taken (mid-chunk).  At SYNC the wave's queue holds, oldest
        // first, its DMA copies of the next chunk and -- from the second chunk on -- the stores of the previous tile
        // (issued two steps into this chunk): the counted wait retires the copies and leaves the stores in flight.
        constexpr int ST = TD;
        const char* tile_base = reinterpret_cast<const char*>(kbuf) + row0 * C * KB;  // tile the next store_tile16 writes
        int u_cur = 0;
        Acc16 prev;  // accumulators of the previous tile (folded and stored two steps into the next one)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) { prev.m[mt][nb] = f32x4v{0.f, 0.f, 0.f, 0.f}; prev.x[mt][nb] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
        // two accumulator sets alternate between consecutive tiles (the loop is unrolled by two; nchunks = L * TC is even)
        auto chunk = [&](int cidx, Acc16& acc, const Acc16& prv) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) { acc.m[mt][nb] = f32x4v{0.f, 0.f, 0.f, 0.f}; acc.x[mt][nb] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
            MmaStream16<TD> ms;
            ms.start(lds[sl], lane);
            ms.template run<0, 2>(acc, b16);
            if (cidx > 0) {  // the previous tile leaves while this tile's first MFMAs run
                store_tile16(prv, tile_base);
                // next tile: 32 columns on, or the first column tile of the next layer
                tile_base += (++u_cur == TC) ? (u_cur = 0, (ptrdiff_t)layer_stride * KB - (TC - 1) * 32 * KB) : 32 * KB;
            }
            ms.template run<2, ST>(acc, b16);
            if (cidx == 0) dma_wait();
            else if (full) dma_wait_but<4>();
            else dma_wait_but<2>();
            __syncthreads();
            copy3(cidx);
            ms.template run<ST, 2 * TD>(acc, b16);
            sl = slot_after(sl, 1);
        };
        Acc16 accA;
#pragma unroll 1
        for (int cidx = 0; cidx < nchunks; cidx += 2) {
            chunk(cidx, accA, prev);
            chunk(cidx + 1, prev, accA);
        }
        store_tile16(prev, tile_base);
        EDGE_TICK(3);
    }  // pair loop
    // (round 4, lint rule ldsdma-unwaited-exit: no LDS-DMA copy is left in flight when a wave ends -- the last copies of a ring
    // target a chunk nobody will read; the hardware's implicit wait at s_endpgm is not relied upon)
    dma_wait();
    EDGE_TICK(4);
#ifdef ARREAU_EDGE_TIMING
    tacc_[5] = clock64() - c_start_;        // whole kernel, shader-clock ticks (slot 5)
    tacc_[6] = wall_clock64() - w_start_;   // whole kernel, 100 MHz ticks (slot 6)
    if (threadIdx.x == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&arreau_edge_ticks[i], (unsigned long long)tacc_[i]);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Small-batch form of the same kernel: ONE 32-ROW TILE PER WORKGROUP, its output chunks split over eight waves.
//
// Above, a wave carries its 32 rows (two edge slots x 16 orientations) through all 32 weight chunks: 39 k matrix-pipe
// cycles per tile, fine when 20 k tiles keep every SIMD busy, 62 us of one wave's latency when a single crystal of 8
// atoms brings 32 tiles to a 256-CU chip.  Here a workgroup owns one tile and the OUTPUT chunks of each layer are dealt
// to its waves: layer 1 (4 chunks) to waves 0-3, layer 2 (8 chunks) one per wave, the 20 projection chunks round-robin.
// A wave's weight fragments come straight from the packed stream in L2 into registers (no LDS ring, no counted waits);
// the layer outputs meet in LDS (the re-layout pad of the kernel above, now shared by the workgroup).  Every output
// chunk is computed by exactly the instruction sequence of the kernel above -- same operand planes, same k order, same
// epilogues -- so the K tiles are BIT-IDENTICAL: the launcher picks the form by batch size.
// ---------------------------------------------------------------------------------------------------------------------
// measured on MI355X (tools/gpu_edge_split_sweep.sh): 60 / 120 / 200 / 260 / 380 receivers: 19.8 / 31.8 / 56.9 / 71.0 / 88.1 us
// against 65.9 / 67.0 / 69.0 / 71.8 / 74.0 us of the persistent form
#define ARREAU_EDGE_SPLIT_MAX_NODES 240
template <int C, int D, bool K3, bool BFP8>
__global__ __launch_bounds__(512) void edge_kernel_f16x3_split(
    const float* __restrict__ nbr_dir, const float* __restrict__ nbr_dist, const int32_t* __restrict__ deg,
    const int32_t* __restrict__ batch, const float* __restrict__ lattice, const float* __restrict__ ori,
    const u32x4* __restrict__ stream, const float* __restrict__ b1, const float* __restrict__ b2, float r_max, int N, int k,
    int L, float* __restrict__ kbuf, int n0) {
    constexpr int TC = C / 32, TD = D / 32, TM = ARREAU_MONO_PAD / 32;
    constexpr int NF1 = TM * 4, NF2 = TC * 4, NF3 = TD * 4;  // 1 KiB fragments per chunk: 12, 16, 32
    static_assert(TM == 3 && TC == 4 && TD == 8, "wave roles below assume C = 128, D = 256");
    __shared__ u32x4 hpad[TC][256];        // layer-1 tiles in the re-layout format of the kernel above (4 KiB each)
    __shared__ u32x4 bpad[TD][2][2][64];   // layer-2 output as B operand: [k-block u][column block nb][plane][lane]

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int node = n0 + (int)(blockIdx.x >> 2), wn = blockIdx.x & 3;  // receiver, its tile: slots 2 wn, 2 wn + 1
    const int nd = min(deg[node], k);
    if (2 * wn >= nd) return;  // no edge in this tile (the kernel above stores nothing for it either)
    const int h = lane >> 5, j = lane & 31;
    const int c16 = lane & 15, g16 = lane >> 4;

    const u32x4* s1 = stream;                                   // layer-1 chunks
    const u32x4* s2 = stream + (size_t)TC * NF1 * 64;           // layer-2 chunks
    const u32x4* s3 = s2 + (size_t)TD * NF2 * 64;               // projection chunks, L * TC of them
    const int nchunks = L * TC;

    // ---- weight requests first: layer 1 (waves 0-3), layer 2 (chunk = wave), first half of the first projection chunk ----
    u32x4 w1[NF1], w2[NF2], wx[16], wy[16];
    if (wave < TC) {
#pragma unroll
        for (int i = 0; i < NF1; ++i) w1[i] = s1[(size_t)wave * NF1 * 64 + i * 64 + lane];
    }
#pragma unroll
    for (int i = 0; i < NF2; ++i) w2[i] = s2[(size_t)wave * NF2 * 64 + i * 64 + lane];
    auto load_half = [&](u32x4 (&w)[16], int cidx, int half) {
        const u32x4* src = s3 + (size_t)cidx * NF3 * 64 + (size_t)half * 16 * 64 + lane;
#pragma unroll
        for (int i = 0; i < 16; ++i) w[i] = src[i * 64];
    };

    // ---- per-row attributes (transforms/invariants.py:82-88), as in the kernel above ------------------------------
    const int slot = 2 * wn + (j >> 4);
    const int o = j & 15;
    const int slot_c = min(slot, nd - 1);  // (as above: an unused slot computes on its receiver's last edge, window zero)
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
        const float inv_dn = __builtin_amdgcn_rcpf(fmaxf(sqrtf((dx * dx + dy * dy) + dz * dz), 1e-8f));
        const float ux = dx * inv_dn, uy = dy * inv_dn, uz = dz * inv_dn;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float lx = Lm[3 * i], ly = Lm[3 * i + 1], lz = Lm[3 * i + 2];
            const float inv_ln = __builtin_amdgcn_rcpf(fmaxf(sqrtf((lx * lx + ly * ly) + lz * lz), 1e-8f));
            a[3 + i] = (ux * (lx * inv_ln) + uy * (ly * inv_ln)) + uz * (lz * inv_ln);
        }
        const float u = dist * __builtin_amdgcn_rcpf(r_max);
        const float u2 = u * u, u6 = u2 * u2 * u2;
        const float w = 1.0f - 28.0f * u6 + 48.0f * u6 * u - 21.0f * u6 * u2;
        window = (slot < nd && dist < r_max) ? w : 0.0f;
    }
    float win16[2];
    win16[0] = __shfl(window, c16, 64);
    win16[1] = __shfl(window, 16 + c16, 64);

    // ---- layer 1, chunk u = wave (waves 0-3): h = GELU(W1f . mono + b1), into the shared re-layout pad -------------
    if (wave < TC) {
        Planes2 bm[TM];
        bm[0] = split_tile2(mono_tile_h<0>(a, h, std::make_integer_sequence<int, 16>{}));
        bm[1] = split_tile2(mono_tile_h<1>(a, h, std::make_integer_sequence<int, 16>{}));
        bm[2] = split_tile2(mono_tile_h<2>(a, h, std::make_integer_sequence<int, 16>{}));
        f32x16 acc = arreau_bias_tile(b1, wave, h), cross;
#pragma unroll
        for (int r = 0; r < 16; ++r) cross[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2 * TM; ++ks) {
            const int t = ks >> 1, s = ks & 1;
            acc = mfma_f16(w1[2 * ks], bm[t].p[0][s], acc);
            cross = mfma_f16(w1[2 * ks + 1], bm[t].p[0][s], cross);
            acc = mfma_f16(w1[2 * ks], bm[t].p[1][s], acc);  // (activation planes: the residual is unscaled, f16x3.h)
        }
        const Planes2 pl = gelu_split_tile2(acc, cross, 1.0f);
#pragma unroll
        for (int plane = 0; plane < 2; ++plane) {
            hpad[wave][relayout_cell(plane, j, 2 * h + 0)] = pl.p[plane][0];
            hpad[wave][relayout_cell(plane, j, 2 * h + 1)] = pl.p[plane][1];
        }
    }
    __syncthreads();
    asm volatile("" ::: "memory");  // (keeps the later phases' weight requests from being hoisted over this phase)
    load_half(wx, wave, 0);  // first half of this wave's first projection chunk (wave < nchunks: checked by the launcher)

    // ---- layer 2, chunk u = wave: basis = GELU(W2 . h + b2) * window, into the shared B-operand pad ----------------
    {
        Acc16 acc;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const f32x4 bv4 = *reinterpret_cast<const f32x4*>(b2 + 32 * wave + 16 * mt + 4 * g16);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                acc.m[mt][nb] = f32x4v{bv4[0], bv4[1], bv4[2], bv4[3]};
                acc.x[mt][nb] = f32x4v{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int kb = 0; kb < TC; ++kb) {
            // the layer-1 tile kb in the (c = orientation, g) layout of column block nb (re-layout read of the kernel above)
            u32x4 hq[2][2];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int plane = 0; plane < 2; ++plane)
                    hq[nb][plane] = hpad[kb][relayout_cell(plane, 16 * nb + c16, g16)];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const int st = 2 * kb + mt;
                    acc.m[mt][nb] = mfma16_f16(w2[2 * st], hq[nb][0], acc.m[mt][nb]);
                    acc.x[mt][nb] = mfma16_f16(w2[2 * st + 1], hq[nb][0], acc.x[mt][nb]);
                    acc.m[mt][nb] = mfma16_f16(w2[2 * st], hq[nb][1], acc.m[mt][nb]);  // (activation planes)
                }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            u32x4 hi4, lo4;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const f32x2 pre = fma2(f32x2{acc.x[mt][nb][2 * pr], acc.x[mt][nb][2 * pr + 1]}, splat2(F16X3_INV_SCALE),
                                           f32x2{acc.m[mt][nb][2 * pr], acc.m[mt][nb][2 * pr + 1]});
                    const f32x2 gw = gelu_fast2(pre) * splat2(win16[nb]);  // |window| <= 1
                    unsigned hi, lo;
                    if constexpr (BFP8) {  // (the residual as the basis form stores it: f16x3.h, split_pair_fp8)
                        unsigned l8 = 0u;
                        split_pair_fp8<false>(gw, hi, l8);
                        lo = widen_lo8<false>(l8);
                    } else {
                        split_pair2<false>(gw, hi, lo);
                    }
                    hi4[2 * mt + pr] = hi;
                    lo4[2 * mt + pr] = lo;
                }
            bpad[wave][nb][0][lane] = hi4;
            bpad[wave][nb][1][lane] = lo4;
        }
    }
    asm volatile("" ::: "memory");
    load_half(wy, wave, 1);
    __syncthreads();
    asm volatile("" ::: "memory");

    // ---- projections: chunks wave, wave + 8, wave + 16: kernel_l = Wk_l . basis (conv.py:110), one K tile each --------
    const size_t layer_stride = (size_t)N * k * 16 * C;
    const size_t row0 = ((size_t)node * k + 2 * wn) * 16;  // first K row of this tile
    const bool full = 2 * wn + 1 < k;                      // the tile's second slot (column block 1) exists
    Acc16 acc;
    auto zero_acc = [&]() {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) { acc.m[mt][nb] = f32x4v{0.f, 0.f, 0.f, 0.f}; acc.x[mt][nb] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
    };
    auto half_steps = [&](const u32x4 (&w)[16], int half) {  // steps 8 half .. 8 half + 7 of the chunk, from registers
        // B operands (k-block kb of the basis, both column blocks, both planes) from the shared pad; the scheduling fences
        // keep the compiler from hoisting all of a chunk's LDS reads (128 registers beside the 192 of the weight buffers)
        u32x4 bq[2][4];  // one k-block ahead of its MFMAs
#pragma unroll
        for (int i = 0; i < 4; ++i) bq[0][i] = bpad[4 * half][i >> 1][i & 1][lane];
#pragma unroll
        for (int kbi = 0; kbi < 4; ++kbi) {
            if (kbi + 1 < 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) bq[(kbi + 1) & 1][i] = bpad[4 * half + kbi + 1][i >> 1][i & 1][lane];
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const int i = 2 * kbi + mt;
                    acc.m[mt][nb] = mfma16_f16(w[2 * i], bq[kbi & 1][2 * nb], acc.m[mt][nb]);
                    acc.x[mt][nb] = mfma16_f16(w[2 * i], bq[kbi & 1][2 * nb + 1], acc.x[mt][nb]);
                    acc.x[mt][nb] = mfma16_f16(w[2 * i + 1], bq[kbi & 1][2 * nb], acc.x[mt][nb]);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto store_tile = [&](int cidx) {
        const int l = cidx / TC, u = cidx % TC;
        float* tile = kbuf + (size_t)l * layer_stride + row0 * C + 32 * u;
        char* tile3 = reinterpret_cast<char*>(kbuf) + ((size_t)l * layer_stride + row0 * C + 32 * u) * 3;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaf(acc.x[mt][nb][r], F16X3_INV_SCALE, acc.m[mt][nb][r]);
                if (nb == 0 || full) {
                    if constexpr (K3)
                        *reinterpret_cast<u32x3_k*>(tile3 + ((size_t)(16 * nb + c16) * C + 16 * mt + 4 * g16) * 3) =
                            arreau_pack_k3(v[0], v[1], v[2], v[3]);
                    else
                        *reinterpret_cast<f32x4*>(tile + (size_t)(16 * nb + c16) * C + 16 * mt + 4 * g16) = v;
                }
            }
    };
    // two register buffers of half a chunk each: while one is consumed the other is in flight (a third would hide more of
    // the L2 latency but does not fit beside them: 3 x 64 + accumulators + operands spills)
    zero_acc();
    half_steps(wx, 0);
    if (wave + 8 < nchunks) load_half(wx, wave + 8, 0);
    half_steps(wy, 1);
    store_tile(wave);
    if (wave + 8 < nchunks) {
        load_half(wy, wave + 8, 1);
        zero_acc();
        half_steps(wx, 0);
        if (wave + 16 < nchunks) load_half(wx, wave + 16, 0);
        half_steps(wy, 1);
        store_tile(wave + 8);
        if (wave + 16 < nchunks) {
            load_half(wy, wave + 16, 1);
            zero_acc();
            half_steps(wx, 0);
            half_steps(wy, 1);
            store_tile(wave + 16);
        }
    }
}

int arreau_launch_edge_f16x3(const arreau_model* m, const float* dir, const float* dist, const int32_t* deg,
                             const int32_t* batch, const float* lattice, int N, float* kbuf, hipStream_t s, NodeRange r) {
    const int n0 = r.n0, n1 = r.n1 < 0 ? N : r.n1;
    if (n1 <= n0) return ARREAU_OK;
    if (!(m->C == 128 && m->D == 256 && m->k <= 8)) {
        arreau_set_error("edge kernel (fp16x3): unsupported (hidden_dim, basis_dim, max_neighbors)");
        return ARREAU_EINVAL;
    }
    // 8 waves = two receivers per workgroup, one workgroup per CU (2 waves per SIMD from the same workgroup).
    // A 4-wave / two-workgroups-per-CU geometry was tried and dropped: it was not run-to-run reproducible.
    // Persistent: one workgroup per CU walks the receiver pairs with stride gridDim (ARREAU_EDGE_WGS overrides the
    // workgroup count, e.g. to (N+1)/2 for one pair per workgroup).
    static const int wgs_env = [] { const char* e = getenv("ARREAU_EDGE_WGS"); return e ? atoi(e) : 0; }();
    static const int n_cu = [] {  // CUs of the current device (one process drives one GPU)
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            return (int)prop.multiProcessorCount;
        return 256;
    }();
    // Small launches: one tile per workgroup, output chunks split over its waves (bit-identical K tiles): while the tiles
    // are fewer than the chip's wave slots the persistent form above is one wave's latency (62 us at 1 x 8 atoms).
    static const int split_env = [] { const char* e = getenv("ARREAU_EDGE_SPLIT"); return e ? atoi(e) : -1; }();
    const bool split_ok = m->L * 4 >= 8 && m->L * 4 <= 24;
    // Whole-batch launches only: in every single-stream test the K tiles are bit-identical to the persistent form's, but when
    // two slices of a batch run on separate streams (arreau_model_set_batch_layout) and one slice's neighbour-list kernel
    // overlaps the other's tile-per-workgroup edge kernel, about one evaluation in ten came out different at the 1e-5
    // level (tools/exp/debug_sliced4.py / debug_sliced5.py; never with the persistent form, never when the neighbour lists
    // were built before the fork): one receiver of the later slice gets a neighbour list without its nearest candidate.
    // Inputs are identical from evaluation to evaluation; the cause is not found.  Slices exist for batches of thousands of
    // atoms, far above the switch-over, so nothing is lost by keeping the small-launch form to unsliced launches.
    const bool whole_batch = n0 == 0 && n1 == N && r.wg_cap == 0;
    const bool use_split = split_ok && !arreau_basis_form(m, N) &&
                           (split_env >= 0 ? split_env != 0 : (whole_batch && (n1 - n0) <= ARREAU_EDGE_SPLIT_MAX_NODES));
    if (use_split) {
        auto launch = [&](auto kernel) {
            ARREAU_LAUNCH(kernel, dim3((unsigned)(n1 - n0) * 4), dim3(512), 0, s, dir, dist, deg, batch, lattice, m->ori,
                               reinterpret_cast<const u32x4*>(m->edge_f16), m->b1, m->b2, m->cfg.radius, N, m->k, m->L, kbuf, n0);
        };
        if (arreau_basis_fp8(m)) launch(edge_kernel_f16x3_split<128, 256, false, true>);
        else launch(edge_kernel_f16x3_split<128, 256, false, false>);
        ARREAU_CHECK_HIP(hipGetLastError());
        return ARREAU_OK;
    }
    const int npairs = (n1 - n0 + 1) / 2;
    int wgs = wgs_env > 0 ? (wgs_env < npairs ? wgs_env : npairs) : (npairs < n_cu ? npairs : n_cu);
    if (r.wg_cap > 0 && wgs > r.wg_cap) wgs = r.wg_cap;
    // (decided from the WHOLE batch, not from this launch's range: the two forms lay the shared kbuf region out differently,
    // so slices of one batch on either side of the threshold must not mix them -- ADVICE round 3)
    if (arreau_basis_form(m, N)) {  // stop after layer 2, store the basis planes (the node-layer launcher projects them)
        if (arreau_basis_fp8(m))
            ARREAU_LAUNCH((edge_kernel_f16x3<128, 256, 8, false, false, true>), dim3(wgs), dim3(512), 0, s, dir, dist, deg, batch, lattice, m->ori,
                          reinterpret_cast<const u32x4*>(m->edge_f16), m->b1, m->b2, m->cfg.radius, N, m->k, m->L, kbuf, n0, n1);
        else
            ARREAU_LAUNCH((edge_kernel_f16x3<128, 256, 8, false, false, false>), dim3(wgs), dim3(512), 0, s, dir, dist, deg, batch, lattice, m->ori,
                          reinterpret_cast<const u32x4*>(m->edge_f16), m->b1, m->b2, m->cfg.radius, N, m->k, m->L, kbuf, n0, n1);
        ARREAU_CHECK_HIP(hipGetLastError());
        return ARREAU_OK;
    }
    auto launch = [&](auto kernel) {
        ARREAU_LAUNCH(kernel, dim3(wgs), dim3(512), 0, s, dir, dist, deg, batch, lattice, m->ori,
                           reinterpret_cast<const u32x4*>(m->edge_f16), m->b1, m->b2, m->cfg.radius, N, m->k, m->L, kbuf, n0, n1);
    };
    if (arreau_basis_fp8(m)) launch(edge_kernel_f16x3<128, 256, 8, false, true, true>);
    else launch(edge_kernel_f16x3<128, 256, 8, false, true, false>);
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}
