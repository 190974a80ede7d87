// K3+K4a fused per layer (round 3): kernel projection + message passing + spherical convolution of ONE layer, without the
// K round trip through HBM.
//
//   K_l[(n,s,o), c] = sum_d basis[(n,s,o), d] * Wk_l[c, d]                    (ponita/nn/conv.py:110)
//   x1[n,o,c]       = sum_{s < deg[n]} K_l[(n,s,o), c] * x[src(n,s), o, c]    (conv.py:111,131-133 + sum aggregation)
//   x2[n,p,c]       = sum_o x1[n,o,c] * FK_l[o,p,c] + bias[c]                 (conv.py:113-127; FK holds the 1 / O)
//
// Until round 2 the edge kernel evaluated all L projections and stored their results (the "K stash": L*C values per
// (edge, orientation) row, 1.26 GB per step at 256 x 20 as 3-byte floats), and a message kernel per layer read them back
// (conv_kernel_streamed in node.hip) -- 71 % of the edge kernel's matrix work existed to produce bytes whose only purpose
// was to be re-read once.  Now the edge kernel stops after the basis (edge_f16.hip, PROJ = false: two fp16 planes in
// B-operand fragment order, 1 KiB per row, written once) and THIS kernel streams a receiver's basis block through LDS,
// projects it with the layer's weights and consumes the K values in registers:
//
//   workgroup = 8 waves on one CU, persistent over receivers (XCD-aware order as in node.hip), two roles:
//   * waves 0-3, "projection": wave w owns output channels 32 w .. 32 w + 31.  Its share of Wk_l (both fp16 planes of the
//     split-precision scheme, 32 fragments = 128 registers) stays in REGISTERS for the whole launch; per slot s the 16 rows
//     (orientations) x 256 basis functions arrive as MFMA B fragments from the LDS ring (48 MFMAs per slot and wave),
//     the folded K tile is multiplied by the sender's feature rows (gathered through L2), and added to the running sum
//     in slot order -- product rounded, then added, like messages -> index_add_ (and like the kernels this replaces:
//     bit-identical to the edge / conv pair with an fp32 K stash).  The sum goes to an LDS tile.
//   * waves 4-7, "mix": issue the LDS-DMA copies of the basis blocks EIGHT slots ahead (16 KiB per slot, ring of 9 slots =
//     144 KiB: 96-112 KiB of HBM requests in flight per CU without a single staging register), and meanwhile run the
//     depth-wise 16 x 16 orientation mix of the PREVIOUS receiver (fiber kernel slice in 128 registers per thread), two
//     orientations per slot step, and store its result.
//   One workgroup barrier per slot step, in the MIDDLE of the projection waves' MFMA stream (the ring protocol of
//   edge_f16.hip): at SYNC_q everybody has left slot q-1 (its buffer is free: the copy of slot q+7 starts) and the mix
//   waves have drained their share of slot q+1 (hand-counted vmcnt: copies and stores retire in issue order).
//
// Per layer at 256 x 20: 671 MB of basis streamed under 129 GFLOP of fp16 MFMA work.
#include <stdlib.h>

#include "f16x3.h"

#define CP_STR 132  // floats per row of the LDS tile (16 orientations x 128 channels, padded)
// the basis is read once per layer: streaming hint (as for the K stream it replaces, node.hip)
#if !defined(ARREAU_BASIS_LOAD_POL) || ARREAU_BASIS_LOAD_POL == 1
#define ARREAU_BASIS_LOAD_POLICY " nt"
#else
#define ARREAU_BASIS_LOAD_POLICY ""
#endif

namespace {
// one 1 KiB fragment: each lane's 16 bytes at wave-uniform base + per-lane offset -> LDS at M0 (+ 16 lane)
__device__ __forceinline__ void cp_glds16(unsigned lane16, const void* base_uniform, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ARREAU_BASIS_LOAD_POLICY
                 :
                 : "v"(lane16), "s"(base_uniform), "s"(lds_dst)
                 : "memory", "m0");
}
template <int IMM>
__device__ __forceinline__ void cp_store4(const float* base_uniform, unsigned lane_off, float v) {
    asm volatile("global_store_dword %0, %1, %2 offset:%3" : : "v"(lane_off), "v"(v), "s"(base_uniform), "n"(IMM) : "memory");
}
template <int N>
__device__ __forceinline__ void cp_wait_but() {
#ifdef ARREAU_DEBUG_WAIT_ALL
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
}  // namespace

// PW = projection waves (4: two 16-channel tiles per wave, 8 waves per workgroup; 8: one tile per wave, 12 waves per
// workgroup = two projection waves per SIMD, which cover each other's LDS latency -- a lone projection wave per SIMD
// waits for its B fragments in the open: measured 149 us per layer at 256 x 20 with PW = 4)
// BFP8: the residual plane of the stashed basis is OCP fp8 e4m3 (edge_f16.hip): a slot block is 12 KiB -- eight 1 KiB hi
// fragments, then eight 512 B lo fragments (8 bytes per lane), widened to fp16 in registers (exact) in front of their MFMAs.
// X8 (round 4, needs BFP8): the two CROSS products of the split scheme on the fp8 matrix instruction.  They sit 2^-11 below the
// main product, so four significand bits per operand are enough (profiles/r04_cross_precision_study.txt); both cross products
// of a PAIR of k-blocks are one v_mfma_scale_f32_16x16x128_f8f6f4 with unit scales --
//     A = [a1_8(kb) | a1_8(kb+1) | a2_8(kb) | a2_8(kb+1)]   (packed on the host: model.hip, pack_conv_cross_fp8; 64 registers)
//     B = [b2_8(kb) | b2_8(kb+1) | b1_8(kb) | b1_8(kb+1)]   (b2_8 = the stash's residual planes as stored -- the two 8-byte LDS reads
//                                                            land in place --, b1_8 = e4m3(b1), converted here)
// -- 36 cycles for what took four fp16 instructions of 18 (tools/exp/fp8_mfma_check.hip); with the two main products a pair of
// k-blocks costs 72 cycles of the matrix pipe instead of 108.  a1_8 = e4m3(64 a1), a2_8 = e4m3(64 a2): K = main + X / (64 * 2^11).
// A timing-only build that dropped a third of the matrix work measured 140.6 -> 122.3 us per launch at 256 x 20
// (profiles/r04a_conv_proj_sweep.txt); the real kernel, all products kept: 133.8 -> 125.1 us, step 1.414 -> 1.360 ms (same box).
typedef int i32x8 __attribute__((ext_vector_type(8)));
namespace {
// four fp16 -> four e4m3 (round to nearest even; NOT saturating: beyond 464 the instruction returns NaN -- measured in round 5,
// tools/exp/fp8_cvt_check.hip; such a basis value surfaces as NONFINITE and the host repeats the evaluation on fp16 planes) INTO `into`: each conversion writes one 16-bit word of its
// destination and keeps the other, so the destination is an input too -- converting into the register that already is the MFMA
// operand's slot (its old content is dead) costs no move
__device__ __forceinline__ int cp_cvt4_fp8(int into, unsigned h01, unsigned h23) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef short s2 __attribute__((ext_vector_type(2)));
    s2 r = __builtin_bit_cast(s2, into);
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, __builtin_bit_cast(h2, h01), 1.0f, false);
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, __builtin_bit_cast(h2, h23), 1.0f, true);
    return __builtin_bit_cast(int, r);
}
}  // namespace
// X8: the instruction's block scales fold the cross products in -- operand A carries 2^-17 = 1 / (64 * 2^11) (E8M0 byte 127 - 17 in
// every lane), so each fp8 product adds X / (64 * 2^11) straight into the MAIN accumulator: no second accumulator, no fold
#define CP_X8_SCALE_A 0x6e6e6e6e
#define CP_X8_SCALE_B 0x7f7f7f7f

template <int C, int D, int PW, bool BFP8, bool X8 = false>
__global__ __launch_bounds__(64 * (PW + 4), (PW + 4) / 4) void conv_proj_kernel(
    const u32x4* __restrict__ basis,     // [N*8 slots][16 fragments = (k-block, plane)][64 lanes] x 16 bytes  (BFP8: see above)
    const u32x4* __restrict__ wchunks,   // this layer's projection chunks of the packed fp16x3 stream: [C/32][32][64]
    const u32x4* __restrict__ x8w,       // X8: this layer's fp8 cross operands [C/16 tiles][D/64 pairs][2][64 lanes] x 16 bytes
    const int32_t* __restrict__ deg, const int32_t* __restrict__ src,
    const float* __restrict__ x_in,      // [N][16][C]
    const float* __restrict__ fk,        // [16(o)][16(p)][C]
    const float* __restrict__ conv_bias, int n0, int N /* receivers n0 .. n0 + N - 1 (absolute indices into whole-batch arrays) */,
    float* __restrict__ x_conv,          // [N][16][C]
    int reverse)  // walk the receivers from the last to the first (see the launcher: the Infinity Cache holds the END of the last pass)
{
    static_assert(C == 128 && D == 256, "roles and register budgets assume C = 128, D = 256");
    static_assert(!X8 || BFP8, "the fp8 cross products take the stash's e4m3 residual plane as it is stored");
    constexpr int K = 8, NKB = D / 32;
    constexpr unsigned SLOT_BYTES = BFP8 ? 12288 : 16384;   // a slot block of the stash in HBM
    // X8: in LDS a slot takes 16 KiB -- behind the two stashed planes a third one, b1_8 = e4m3(b1), 8 bytes per lane and k-block like
    // the residual plane.  It is written by the MIX wave that copied the fp16 fragments it converts (its own copies, behind its own
    // counted wait, before the barrier that publishes the slot): the projection wave, alone on its SIMD, pays ~7.5 cycles for every
    // vector instruction it issues (a timing-only build without its 32 conversions per slot ran 120 -> 109 us per launch), the mix
    // wave has the time.
    constexpr unsigned SLOT_LDS = X8 ? 16384 : SLOT_BYTES;
    constexpr int NC = SLOT_BYTES / 4096;  // 1 KiB copies per mix wave and slot
    // NINE slot buffers (slot q of the workgroup's sequence lives in buffer q mod 9) + one tile = 152 KiB of the CU's 160:
    // the stream's rate is bytes in flight over the loaded HBM latency (about 4.3 us: 80-96 KiB in flight per CU gave
    // 4.7 TB/s with eight buffers), and an LDS-DMA byte in flight needs its landing place for the whole flight.
    constexpr int RING = 9;
    __shared__ u32x4 ring[RING][SLOT_LDS / 16];
    __shared__ __attribute__((aligned(16))) float tile[16 * CP_STR];      // x1 of the receiver just finished
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;

    // receiver sequence of this workgroup (XCD-aware order, see conv_kernel in node.hip)
    constexpr int GROUP = 32;
    const bool xcd_order = (gridDim.x & 7) == 0 && N >= 32 * GROUP;
    const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
    const int n_iter = xcd_order ? ((N + 8 * GROUP - 1) / (8 * GROUP)) * GROUP : N;
    const int m_step = xcd_order ? wgs_per_xcd : (int)gridDim.x;
    auto fwd_of = [&](int m) { return xcd_order ? ((m / GROUP) * 8 + xcd) * GROUP + (m % GROUP) : m; };
    auto local_of = [&](int m) { return reverse ? N - 1 - fwd_of(m) : fwd_of(m); };  // (only called for valid m)
    auto next_valid = [&](int m) {
        while (m < n_iter && fwd_of(m) >= N) m += m_step;
        return m;
    };
    int m = next_valid(xcd_order ? wg_in_xcd : (int)blockIdx.x);
    if (m >= n_iter) return;  // (workgroup-uniform: no barrier has been reached)
    int mn = next_valid(m + m_step);

    constexpr int MT = 8 / PW;  // 16-channel tiles per projection wave
    if (wave < PW) {
        // =========================================== projection role ===========================================
        const int c16 = lane & 15, g16 = lane >> 4;
        u32x4 A1[MT][NKB], A2[MT][X8 ? 1 : NKB];  // [16-channel tile][k-block]: the two fp16 planes of Wk_l rows 16 * (MT wave + mt) ..
        i32x8 A8[MT][X8 ? NKB / 2 : 1];          // X8: the fp8 cross operands instead of the residual plane
        if constexpr (X8) {
            const u32x4* wc = wchunks + (size_t)((MT * wave) >> 1) * 32 * 64 + lane;
            const int t0 = (MT * wave) & 1;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) A1[mt][kb] = wc[((kb * 2 + t0 + mt) * 2 + 0) * 64];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int kp = 0; kp < NKB / 2; ++kp) {
                    const u32x4* a8 = x8w + ((size_t)((MT * wave + mt) * (NKB / 2) + kp) * 2) * 64 + lane;
                    const u32x4 lo = a8[0], hi = a8[64];
                    A8[mt][kp] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
                }
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(A1[mt][kb]));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int kp = 0; kp < NKB / 2; ++kp) asm volatile("" : "+v"(A8[mt][kp]));
        } else {
            // chunk u = 32 output channels; fragment (kb, tile-in-chunk, plane) at ((kb * 2 + tile) * 2 + plane) * 64
            const u32x4* wc = wchunks + (size_t)((MT * wave) >> 1) * 32 * 64 + lane;
            const int t0 = (MT * wave) & 1;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    A1[mt][kb] = wc[((kb * 2 + t0 + mt) * 2 + 0) * 64];
                    A2[mt][kb] = wc[((kb * 2 + t0 + mt) * 2 + 1) * 64];
                }
            // wait for the weights HERE: left to their first use, the waits would sit inside the receiver loop, where every
            // pass would also wait for the sender rows requested two slots ahead
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(A1[mt][kb]), "+v"(A2[mt][kb]));
        }
        const unsigned cb = 16 * MT * wave + 4 * g16;  // first channel of this lane's four (tile mt adds 16)
        // sender rows: requested two slots ahead into four rotating register sets (hipcc counts these loads itself: this
        // role issues no asm memory operation), consumed slot by slot
        f32x4v xv[4][MT];
        const unsigned x_off = (unsigned)c16 * C + cb;
        // (round 5: a slot beyond the receiver's degree needs no select per value any more -- its stash block is all zeros (the edge
        // kernel writes zero blocks for the slots it does not compute, and a computed slot beyond the degree has window 0), so its K
        // tile is exactly zero and K * x adds nothing, whichever valid row x comes from: eight vector instructions fewer per slot on
        // a wave that has its SIMD to itself)
        auto load_x = [&](const int32_t* srow, int s_, int buf) {
            const int sn = max(srow[s_], 0);  // unused slots: any valid row
            const float* xr = x_in + (size_t)sn * 16 * C;  // wave-uniform base + one 32-bit lane offset
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xv[buf][mt] = *reinterpret_cast<const f32x4v*>(xr + (x_off + 16 * mt));
        };
        {
            const int32_t* srow0 = src + (size_t)(n0 + local_of(m)) * K;
            load_x(srow0, 0, 0);
            load_x(srow0, 1, 1);
        }
        __syncthreads();  // SYNC_-1: slot 0 of the first receiver has landed
        // B fragments travel LDS -> registers DIST k-steps ahead of their MFMAs, through NBUF rotating register sets, across
        // barriers and slot boundaries (a slot's first fragments are requested during the previous slot's last k-steps: that
        // slot has been published by the barrier in the middle of the previous slot).  hipcc, left alone, sinks every
        // ds_read to within one or two MFMAs of its use -- measured: the LDS latency exposed at each of the 64 k-steps of a
        // receiver, a matrix pipe busy 45 % of the time -- so the issue order is pinned with sched_group_barrier below.
        constexpr int DIST = (PW == 4 || X8) ? 2 : 1, NBUF = (PW == 4 || X8) ? 4 : 2;
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
        u32x4 b1[NBUF], b2[NBUF];
        u32x2_t b8[NBUF];  // BFP8: the lo fragment as stored (8 fp8 per lane)
        // X8: the fp8 operand of a pair of k-blocks, [b2_8(kb) | b2_8(kb + 1) | b1_8(kb) | b1_8(kb + 1)]: two blocks alternate (the
        // residual fragments of the next pair arrive from LDS while this pair's block is the MFMA's operand)
        i32x8 B8[2] = {i32x8{0, 0, 0, 0, 0, 0, 0, 0}, i32x8{0, 0, 0, 0, 0, 0, 0, 0}};
        const char* ring_b = reinterpret_cast<const char*>(&ring[0][0]);
        const unsigned hi_off = 16u * lane, lo_off = 8192u + 8u * lane;
        auto frag = [&](unsigned base /* byte offset of the slot buffer: wave-uniform */, int kb, int sl_) {
            if constexpr (X8) {
                b1[sl_] = *reinterpret_cast<const u32x4*>(ring_b + base + hi_off + 1024 * kb);
                const u32x2_t t = *reinterpret_cast<const u32x2_t*>(ring_b + base + lo_off + 512 * kb);
                const u32x2_t t8 = *reinterpret_cast<const u32x2_t*>(ring_b + base + lo_off + 4096 + 512 * kb);  // b1_8 (mix waves)
                B8[(kb >> 1) & 1][2 * (kb & 1)] = (int)t[0];
                B8[(kb >> 1) & 1][2 * (kb & 1) + 1] = (int)t[1];
                B8[(kb >> 1) & 1][4 + 2 * (kb & 1)] = (int)t8[0];
                B8[(kb >> 1) & 1][4 + 2 * (kb & 1) + 1] = (int)t8[1];
            } else if constexpr (BFP8) {
                b1[sl_] = *reinterpret_cast<const u32x4*>(ring_b + base + hi_off + 1024 * kb);
                b8[sl_] = *reinterpret_cast<const u32x2_t*>(ring_b + base + lo_off + 512 * kb);
            } else {
                b1[sl_] = *reinterpret_cast<const u32x4*>(ring_b + base + hi_off + 2048 * kb);
                b2[sl_] = *reinterpret_cast<const u32x4*>(ring_b + base + hi_off + 2048 * kb + 1024);
            }
        };
        // X8: two accumulator sets alternate between consecutive slots -- slot q's epilogue (fold, multiply by the sender rows, ordered
        // add) runs INSIDE slot q + 1's MFMA stream, behind its first pair of k-blocks, instead of holding the matrix pipe up between
        // two slots (round 4: a timing-only build without the epilogue arithmetic ran 131 -> 99 us per launch)
        constexpr int NSET = X8 ? 2 : 1;
        f32x4v am[NSET][MT], ax[X8 ? 1 : NSET][MT];  // (X8: the cross products accumulate into am, scaled by the instruction)
        f32x4v sum[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) sum[mt] = f32x4v{0.f, 0.f, 0.f, 0.f};
        auto epilogue = [&](int set, int xbuf, bool on) {  // product rounded, then added in edge order (messages -> index_add_)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float kv = am[set][mt][r];
                    if constexpr (!X8) kv = fmaf(ax[set][mt][r], F16X3_INV_SCALE, kv);
                    sum[mt][r] = __fadd_rn(sum[mt][r], __fmul_rn(kv, xv[xbuf][mt][r]));  // (a slot beyond the degree: 0 * x)
                    asm volatile("" : "+v"(sum[mt][r]));  // (keeps each add where the select used to pin it: without it hipcc spills)
                }
        };
        auto epilogue_piece = [&](int set, int xbuf, bool on, int v) {  // value v = 4 mt + r of the same arithmetic
            const int mt = v >> 2, r = v & 3;
            if (mt < MT) {
                float kv = am[set][mt][r];
                if constexpr (!X8) kv = fmaf(ax[set][mt][r], F16X3_INV_SCALE, kv);
                sum[mt][r] = __fadd_rn(sum[mt][r], __fmul_rn(kv, xv[xbuf][mt][r]));
                asm volatile("" : "+v"(sum[mt][r]));
            }
        };
        auto write_tile = [&]() {
            // (address re-derived from an opaque copy of the lane index right here: kept across the whole slot sequence it
            // would cost registers the loop does not have)
            int ln = lane;
            asm volatile("" : "+v"(ln));
            float* trow = &tile[(ln & 15) * CP_STR + 16 * MT * wave + 4 * (ln >> 4)];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                *reinterpret_cast<f32x4v*>(trow + 16 * mt) = sum[mt];
                sum[mt] = f32x4v{0.f, 0.f, 0.f, 0.f};
            }
        };
        int rb = 0;  // ring buffer of the current slot
#pragma unroll
        for (int kb = 0; kb < DIST; ++kb) frag(0u, kb, kb);
        while (true) {
            const int n = n0 + local_of(m);
            const bool has_next = mn < n_iter;
            const int nd = min(deg[n], K);
            const int32_t* srow = src + (size_t)n * K;
            const int32_t* srow_next = src + (size_t)(has_next ? n0 + local_of(mn) : n) * K;
#pragma unroll
            for (int s = 0; s < K; ++s) {
                const int set = X8 ? (s & 1) : 0;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    am[set][mt] = f32x4v{0.f, 0.f, 0.f, 0.f};
                    if constexpr (!X8) ax[set][mt] = f32x4v{0.f, 0.f, 0.f, 0.f};
                }
                if (s + 2 < K) load_x(srow, s + 2, (s + 2) & 3);
                else load_x(srow_next, s + 2 - K, (s + 2) & 3);  // (the last receiver re-reads its own rows: never used)
                const unsigned f = (unsigned)rb * SLOT_LDS;
                rb = rb == RING - 1 ? 0 : rb + 1;
                const unsigned f_next = (unsigned)rb * SLOT_LDS;
                auto kstep = [&](int kb) {  // same product order per accumulator as MmaStream16 (edge_f16.hip)
                    if constexpr (X8) return;
                    const int slot = kb % NBUF, kk = kb + DIST;
                    if (kk < NKB) frag(f, kk, kk % NBUF);
                    else frag(f_next, kk - NKB, kk % NBUF);  // first k-blocks of the next slot (published by this slot's barrier;
                                                             // at the very end of the sequence: a buffer nobody writes any
                                                             // more, never used)
                    const int s2 = BFP8 ? 0 : slot;  // (the widened fp8 fragment is made right in front of its MFMAs: one copy)
                    if constexpr (BFP8) {
#pragma unroll
                        for (int w2 = 0; w2 < 2; ++w2) {
                            b2[0][2 * w2] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(b8[slot][w2], 1.0f, false));
                            b2[0][2 * w2 + 1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(b8[slot][w2], 1.0f, true));
                        }
                    }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        am[set][mt] = mfma16_f16(A1[mt][kb], b1[slot], am[set][mt]);
                        ax[set][mt] = mfma16_f16(A1[mt][kb], b2[s2], ax[set][mt]);
                        ax[set][mt] = mfma16_f16(A2[mt][X8 ? 0 : kb], b1[slot], ax[set][mt]);
                    }
                };
                auto kpair = [&](int kp) {  // X8: two k-blocks -- two main products per tile on fp16, both cross products as one fp8 product
                    const int kb0 = 2 * kp, kb1 = kb0 + 1, s0 = kb0 % NBUF, s1 = kb1 % NBUF;
#pragma unroll
                    for (int q2 = 0; q2 < 2; ++q2) {
                        const int kk = kb0 + q2 + DIST;
                        if (kk < NKB) frag(f, kk, kk % NBUF);
                        else frag(f_next, kk - NKB, kk % NBUF);
                    }
                    if constexpr (X8) {
                        i32x8& Bp = B8[kp & 1];  // (residual and fp8 halves of both k-blocks were loaded two k-blocks ago)
                        // (one accumulator per tile now carries all three products of a pair: the tiles alternate, so that no MFMA waits for
                        // the one issued just before it)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) am[set][mt] = mfma16_f16(A1[mt][kb0], b1[s0], am[set][mt]);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) am[set][mt] = mfma16_f16(A1[mt][kb1], b1[s1], am[set][mt]);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            am[set][mt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A8[mt][kp], Bp, am[set][mt], 0, 0, 0, CP_X8_SCALE_A, 0,
                                                                                          CP_X8_SCALE_B);
                    }
                };
                // pair kp with the PREVIOUS slot's epilogue dealt between its six MFMAs (sched_barrier keeps every piece behind its MFMA:
                // the vector instructions issue while the matrix pipe works; left to itself hipcc emits the epilogue as one block)
                auto kpair_with_epilogue = [&](int kp, int pset, int pxbuf, bool pon) {
                    const int kb0 = 2 * kp, kb1 = kb0 + 1, s0 = kb0 % NBUF, s1 = kb1 % NBUF;
#pragma unroll
                    for (int q2 = 0; q2 < 2; ++q2) {
                        const int kk = kb0 + q2 + DIST;
                        if (kk < NKB) frag(f, kk, kk % NBUF);
                        else frag(f_next, kk - NKB, kk % NBUF);
                    }
                    if constexpr (X8) {
                        i32x8& Bp = B8[kp & 1];
                        __builtin_amdgcn_sched_barrier(0);
                        int v = 0;
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            am[set][mt] = mfma16_f16(A1[mt][kb0], b1[s0], am[set][mt]);
                            epilogue_piece(pset, pxbuf, pon, v++);
                            if (mt == 0) epilogue_piece(pset, pxbuf, pon, v++);
                            __builtin_amdgcn_sched_barrier(0);
                        }
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            am[set][mt] = mfma16_f16(A1[mt][kb1], b1[s1], am[set][mt]);
                            epilogue_piece(pset, pxbuf, pon, v++);
                            __builtin_amdgcn_sched_barrier(0);
                        }
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            am[set][mt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A8[mt][kp], Bp, am[set][mt], 0, 0, 0, CP_X8_SCALE_A, 0,
                                                                                          CP_X8_SCALE_B);
                            epilogue_piece(pset, pxbuf, pon, v++);
                            if (mt == 0) epilogue_piece(pset, pxbuf, pon, v++);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                };
                auto pin8 = [&]() {  // one half slot of the pair form: per pair its four fragment requests, then its MFMAs
#pragma unroll
                    for (int i = 0; i < NKB / 4; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 3 * MT, 0);
                    }  // (the two 8-byte planes of a k-block pair arrive as one ds_read2st64_b64 each)
                };
                auto pin = [&]() {  // issue order of one half slot: per k-step its fragment requests, then its MFMAs (vector work floats)
#pragma unroll
                    for (int i = 0; i < NKB / 2; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                 // DS reads of k-step + DIST
                        __builtin_amdgcn_sched_group_barrier(0x008, 3 * MT, 0);           // MFMAs of this k-step
                    }
                };
                if constexpr (X8) {
                    static_assert(!X8 || (DIST == 2 && NBUF == 4), "pair form: fragments two k-blocks ahead through four register sets");
                    kpair(0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 3 * MT, 0);
                    // the previous slot's epilogue (the other accumulator set: its MFMAs retired a pair of k-blocks ago).  The LAST slot of
                    // a receiver keeps its epilogue at its own end: carried into the next receiver it would make both accumulator sets
                    // and a set of sender rows live across the receiver loop, and hipcc then spills the weight registers (round 3 met the
                    // same wall with a deferred epilogue behind the loop)
                    if (s > 0) {
                        kpair_with_epilogue(1, set ^ 1, (s - 1) & 3, s - 1 < nd);
                    } else {
                        kpair(1);
                        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 3 * MT, 0);
                    }
                    __syncthreads();  // SYNC_q
#pragma unroll
                    for (int kp = NKB / 4; kp < NKB / 2; ++kp) kpair(kp);
                    pin8();
                } else {
#pragma unroll
                    for (int kb = 0; kb < NKB / 2; ++kb) kstep(kb);
                    pin();
                    __syncthreads();  // SYNC_q
#pragma unroll
                    for (int kb = NKB / 2; kb < NKB; ++kb) kstep(kb);
                    pin();
                }
                if (!X8 || s == K - 1) epilogue(set, s & 3, s < nd);
            }
            write_tile();  // (the mix waves took the previous receiver's tile into registers right behind SYNC of slot 0)
            if (!has_next) break;
            m = mn;
            mn = next_valid(mn + m_step);
        }
        __syncthreads();  // the last receiver's tile is complete
        return;
    }

    // ================================================= mix role =================================================
    const int t2 = tid - 64 * PW;
    const int c = t2 & 127, ph = t2 >> 7;  // channel, half of the output orientations (p = 8 ph .. 8 ph + 7)
    const int w4 = wave - PW;              // this wave copies fragments 4 w4 .. 4 w4 + 3 of every slot block
    float fkr[16][8];
    float bias = conv_bias[c];
    {
        // scalar base + ONE 32-bit per-lane index for all 128 loads (per-load 64-bit vector addresses would not fit the
        // register file next to their results); consumed row by row: hipcc then waits for these loads here and not -- with
        // the vmcnt(0) it would have to use, since it cannot see the asm copies -- somewhere inside the loop
        const unsigned li = (unsigned)((8 * ph) * C + c);
#pragma unroll
        for (int o = 0; o < 16; ++o) {
#pragma unroll
            for (int p = 0; p < 8; ++p) fkr[o][p] = (fk + (o * 16 + p) * C)[li];
#pragma unroll
            for (int p = 0; p < 8; ++p) asm volatile("" : "+v"(fkr[o][p]));
        }
    }
    asm volatile("" : "+v"(bias));
    const unsigned lane16 = 16u * lane;
    const unsigned ring_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)&ring[0][0]);
    const unsigned ring0 = ring_base + 1024u * NC * w4;
    int wb = 0;  // ring buffer the next copy goes to (copies are issued in slot order)
    auto copy_slot = [&](int n, int s) {  // this wave's quarter of slot s of receiver n -> the next ring buffer
        if constexpr (X8) {
            // the fp16 fragments of k-blocks 2 w4, 2 w4 + 1 and the KiB that holds their two residual fragments: the wave converts
            // exactly what it copied
            const char* g = reinterpret_cast<const char*>(basis) + ((size_t)n * K + s) * SLOT_BYTES;
            const unsigned l = ring_base + wb * SLOT_LDS;
            cp_glds16(lane16, g + 2048 * w4, l + 2048u * w4);
            cp_glds16(lane16, g + 2048 * w4 + 1024, l + 2048u * w4 + 1024u);
            cp_glds16(lane16, g + 8192 + 1024 * w4, l + 8192u + 1024u * w4);
        } else {
            const char* g = reinterpret_cast<const char*>(basis) + ((size_t)n * K + s) * SLOT_BYTES + 1024u * NC * w4;
#pragma unroll
            for (int j = 0; j < NC; ++j) cp_glds16(lane16, g + 1024 * j, ring0 + wb * SLOT_BYTES + 1024u * j);
        }
        wb = wb == RING - 1 ? 0 : wb + 1;
    };
    // X8: b1_8 of the two k-blocks this wave copied, for the slot in ring buffer `cb` (its copies have landed: called behind the
    // counted wait that covers them, in front of the barrier that publishes the slot)
    int cb = 0;
    auto convert_slot = [&]() {
        if constexpr (X8) {
            char* slot = reinterpret_cast<char*>(&ring[0][0]) + (unsigned)cb * SLOT_LDS;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const u32x4 h = *reinterpret_cast<const u32x4*>(slot + 1024 * (2 * w4 + j) + 16 * lane);
                typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                u32x2_t o;
                o[0] = (unsigned)cp_cvt4_fp8(0, h[0], h[1]);
                o[1] = (unsigned)cp_cvt4_fp8(0, h[2], h[3]);
                *reinterpret_cast<u32x2_t*>(slot + 12288 + 512 * (2 * w4 + j) + 8 * lane) = o;
            }
            cb = cb == RING - 1 ? 0 : cb + 1;
        }
    };
    const unsigned st_off = 4u * ((8 * ph) * C + c);
    auto store_out = [&](int n, float (&out)[8]) {
        const float* xb = x_conv + (size_t)n * 16 * C;  // wave-uniform
        cp_store4<0>(xb, st_off, out[0] + bias);
        cp_store4<4 * C>(xb, st_off, out[1] + bias);
        cp_store4<8 * C>(xb, st_off, out[2] + bias);
        cp_store4<12 * C>(xb, st_off, out[3] + bias);
        cp_store4<16 * C>(xb, st_off, out[4] + bias);
        cp_store4<20 * C>(xb, st_off, out[5] + bias);
        cp_store4<24 * C>(xb, st_off, out[6] + bias);
        cp_store4<28 * C>(xb, st_off, out[7] + bias);
    };
    // prologue: all eight slots of the first receiver are requested, slot 0 is waited for (the 7 NC younger copies stay in flight)
    {
        const int n = n0 + local_of(m);
#pragma unroll
        for (int s = 0; s < K; ++s) copy_slot(n, s);
        cp_wait_but<7 * NC>();
        convert_slot();   // slot 0
        __syncthreads();  // SYNC_-1
    }
    int i = 0, n_prev = 0;
    float out[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float tv[16];  // x1[o][c] of the previous receiver
    while (true) {
        const int n = n0 + local_of(m);
        const bool has_next = mn < n_iter;
        const int n_next = has_next ? n0 + local_of(mn) : n;
#pragma unroll
        for (int s = 0; s < K; ++s) {
            // Before SYNC_q (q = 8 i + s) this wave's share of slot q + 1 must have landed.  Younger than that copy, in
            // the in-order queue: the copies of slots q + 2 .. q + 7 (NC each) and -- when steps q - 7 .. q - 1 contain a
            // "phase 7" step that stored a receiver's result (s <= 6, from the third receiver on) -- those 8 stores.  The
            // last receiver of the sequence has no full set of younger copies: it waits for everything.
            if (!has_next) cp_wait_but<0>();
            else if (s <= 6 && i >= 2) cp_wait_but<6 * NC + 8>();
            else cp_wait_but<6 * NC>();
            convert_slot();   // slot q + 1 (at the very end of the sequence: a buffer nobody reads any more)
            __syncthreads();  // SYNC_q: slot q - 1's buffer is free
            if (has_next) copy_slot(n_next, s);  // slot q + 8 = slot s of the next receiver
            if (i > 0) {  // orientation mix of the previous receiver, two input orientations per step (o ascending)
                if (s == 0) {
#pragma unroll
                    for (int o = 0; o < 16; ++o) tv[o] = tile[o * CP_STR + c];
#pragma unroll
                    for (int o = 0; o < 16; ++o) asm volatile("" : "+v"(tv[o]));  // read before this wave meets SYNC_q+1
                }
#pragma unroll
                for (int oo = 0; oo < 2; ++oo) {
                    const int o = 2 * s + oo;
                    const float xo = tv[o];
#pragma unroll
                    for (int p = 0; p < 8; ++p) out[p] = fmaf(xo, fkr[o][p], out[p]);
                }
                if (s == K - 1) {
                    store_out(n_prev, out);
#pragma unroll
                    for (int p = 0; p < 8; ++p) out[p] = 0.f;
                }
            }
        }
        n_prev = n;
        ++i;
        if (!has_next) break;
        m = mn;
        mn = next_valid(mn + m_step);
    }
    // (round 4, lint rule ldsdma-unwaited-exit: no LDS-DMA copy is left in flight when a wave ends -- the last copies of a ring
    // target a chunk nobody will read; the hardware's implicit wait at s_endpgm is not relied upon)
    cp_wait_but<0>();
    __syncthreads();  // the last receiver's tile is complete
#pragma unroll
    for (int o = 0; o < 16; ++o) {
        const float xo = tile[o * CP_STR + c];
#pragma unroll
        for (int p = 0; p < 8; ++p) out[p] = fmaf(xo, fkr[o][p], out[p]);
    }
    store_out(n_prev, out);
}

// the launcher: one persistent workgroup per CU (a multiple of 8 workgroups keeps the XCD-aware order)
int arreau_launch_conv_proj(const arreau_model* m, int layer, const float* basis, const int32_t* deg, const int32_t* src,
                            const float* x_in, float* x_conv, int N, hipStream_t s, NodeRange r) {
    const int n0 = r.n0, n1 = r.n1 < 0 ? N : r.n1;
    const int Ng = n1 - n0;
    if (Ng <= 0) return ARREAU_OK;
    if (!(m->C == 128 && m->D == 256 && m->k == 8 && m->f16_ok)) {
        arreau_set_error("conv_proj kernel: unsupported (hidden_dim, basis_dim, max_neighbors) or weights beyond the fp16 range");
        return ARREAU_EINVAL;
    }
    static const int n_cu = [] {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            return (int)prop.multiProcessorCount;
        return 256;
    }();
    int blocks = Ng < n_cu ? Ng : n_cu;
    if (r.wg_cap > 0 && blocks > r.wg_cap) blocks = r.wg_cap;
    constexpr int TC = 4, TD = 8, NF1 = 12, NF2 = 16, NF3 = 32;  // chunk geometry of the packed edge stream (edge_f16.hip)
    const u32x4* stream = reinterpret_cast<const u32x4*>(m->edge_f16);
    const u32x4* wchunks = stream + ((size_t)TC * NF1 + (size_t)TD * NF2 + (size_t)layer * TC * NF3) * 64;
    // Boustrophedon over the layers (round 4).  The stash (503 MB at 256 x 20) is larger than the 256 MiB Infinity Cache, so a pass
    // that starts where the previous one started finds nothing of it on the die; a pass that starts where the previous one ENDED
    // finds the last ~100-250 MB (MI355X_MICROARCH.md: a line stays while the bytes touched since its last use fit the cache).
    // The edge kernel wrote the stash front to back, so layer 0 reads it back to front, layer 1 front to back, ...  Each receiver
    // is computed exactly as before: bit-identical.  ARREAU_CONV_PROJ_BOUSTROPHEDON=0: every pass front to back (A/B).
    static const int bous_env = [] { const char* e = getenv("ARREAU_CONV_PROJ_BOUSTROPHEDON"); return e ? atoi(e) : 1; }();
    const int reverse = bous_env != 0 && (layer & 1) == 0 ? 1 : 0;
    const u32x4* x8w = reinterpret_cast<const u32x4*>(m->conv_x8) + (size_t)layer * (m->C / 16) * (m->D / 64) * 2 * 64;
    auto launch = [&](auto kernel, int threads) {
        ARREAU_LAUNCH(kernel, dim3(blocks), dim3(threads), 0, s, reinterpret_cast<const u32x4*>(basis), wchunks, x8w, deg, src, x_in,
                      m->fk + (size_t)layer * 16 * 16 * m->C, m->conv_bias + (size_t)layer * m->C, n0, Ng, x_conv, reverse);
    };
    const bool fp8 = arreau_basis_fp8(m);
    arreau_prof_conv(0, s);
    // (round 5: the twelve-wave geometry -- eight projection waves of one tile each, ARREAU_CONV_PROJ_WAVES=8 -- is gone: it never
    // ran faster than this one)
    m->ran_x8 = fp8 && arreau_cross_fp8(m) ? 1 : 0;
    if (m->ran_x8) launch(conv_proj_kernel<128, 256, 4, true, true>, 512);
    else if (fp8) launch(conv_proj_kernel<128, 256, 4, true>, 512);
    else launch(conv_proj_kernel<128, 256, 4, false>, 512);
    arreau_prof_conv(1, s);
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}
