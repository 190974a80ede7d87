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
//   * waves 4-7, "mix": issue the LDS-DMA copies of the basis blocks SEVEN slots ahead (16 KiB per slot, ring of 8 slots =
//     128 KiB: about 100 KiB of HBM requests in flight per CU without a single staging register), and meanwhile run the
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

template <int C, int D>
__global__ __launch_bounds__(512, 2) void conv_proj_kernel(
    const u32x4* __restrict__ basis,     // [N*8 slots][16 fragments = (k-block, plane)][64 lanes] x 16 bytes
    const u32x4* __restrict__ wchunks,   // this layer's projection chunks of the packed fp16x3 stream: [C/32][32][64]
    const int32_t* __restrict__ deg, const int32_t* __restrict__ src,
    const float* __restrict__ x_in,      // [N][16][C]
    const float* __restrict__ fk,        // [16(o)][16(p)][C]
    const float* __restrict__ conv_bias, int n0, int N /* receivers n0 .. n0 + N - 1 (absolute indices into whole-batch arrays) */,
    float* __restrict__ x_conv)          // [N][16][C]
{
    static_assert(C == 128 && D == 256, "roles and register budgets assume C = 128, D = 256");
    constexpr int K = 8, NKB = D / 32;
    constexpr unsigned SLOT_BYTES = 16384;
    __shared__ u32x4 ring[K][SLOT_BYTES / 16];                             // slot s of the current / next receiver
    __shared__ __attribute__((aligned(16))) float tile[2][16 * CP_STR];   // x1 of the current / previous receiver
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;

    // receiver sequence of this workgroup (XCD-aware order, see conv_kernel in node.hip)
    constexpr int GROUP = 32;
    const bool xcd_order = (gridDim.x & 7) == 0 && N >= 32 * GROUP;
    const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
    const int n_iter = xcd_order ? ((N + 8 * GROUP - 1) / (8 * GROUP)) * GROUP : N;
    const int m_step = xcd_order ? wgs_per_xcd : (int)gridDim.x;
    auto local_of = [&](int m) { return xcd_order ? ((m / GROUP) * 8 + xcd) * GROUP + (m % GROUP) : m; };
    auto next_valid = [&](int m) {
        while (m < n_iter && local_of(m) >= N) m += m_step;
        return m;
    };
    int m = next_valid(xcd_order ? wg_in_xcd : (int)blockIdx.x);
    if (m >= n_iter) return;  // (workgroup-uniform: no barrier has been reached)
    int mn = next_valid(m + m_step);

    if (wave < 4) {
        // =========================================== projection role ===========================================
        const int c16 = lane & 15, g16 = lane >> 4;
        u32x4 A1[2][NKB], A2[2][NKB];  // [16-channel tile mt][k-block]: the two fp16 planes of Wk_l rows 32 wave + 16 mt ..
        {
            const u32x4* wc = wchunks + (size_t)wave * 32 * 64 + lane;  // fragment (kb, mt, plane) at ((kb * 2 + mt) * 2 + plane) * 64
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    A1[mt][kb] = wc[((kb * 2 + mt) * 2 + 0) * 64];
                    A2[mt][kb] = wc[((kb * 2 + mt) * 2 + 1) * 64];
                }
            // wait for the weights HERE: left to their first use, the waits would sit inside the receiver loop, where every
            // pass would also wait for the sender rows requested two slots ahead
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) asm volatile("" : "+v"(A1[mt][kb]), "+v"(A2[mt][kb]));
        }
        const unsigned cb = 32 * wave + 4 * g16;  // first channel of this lane's four (tile mt adds 16)
        const u32x4* frag0 = &ring[0][0] + lane;
        // sender rows: requested two slots ahead into four rotating register sets (hipcc counts these loads itself: this
        // role issues no asm memory operation), consumed slot by slot
        f32x4v xv[4][2];
        auto load_x = [&](const int32_t* srow, int s_, int buf) {
            const int sn = max(srow[s_], 0);  // unused slots: any valid row, dropped by the select below
            const float* xr = x_in + ((size_t)sn * 16 + c16) * C + cb;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) xv[buf][mt] = *reinterpret_cast<const f32x4v*>(xr + 16 * mt);
        };
        {
            const int32_t* srow0 = src + (size_t)(n0 + local_of(m)) * K;
            load_x(srow0, 0, 0);
            load_x(srow0, 1, 1);
        }
        __syncthreads();  // SYNC_-1: slot 0 of the first receiver has landed
        int par = 0;
        while (true) {
            const int n = n0 + local_of(m);
            const bool has_next = mn < n_iter;
            const int nd = min(deg[n], K);
            const int32_t* srow = src + (size_t)n * K;
            const int32_t* srow_next = src + (size_t)(has_next ? n0 + local_of(mn) : n) * K;
            f32x4v sum[2] = {f32x4v{0.f, 0.f, 0.f, 0.f}, f32x4v{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int s = 0; s < K; ++s) {
                f32x4v am[2] = {f32x4v{0.f, 0.f, 0.f, 0.f}, f32x4v{0.f, 0.f, 0.f, 0.f}};
                f32x4v ax[2] = {f32x4v{0.f, 0.f, 0.f, 0.f}, f32x4v{0.f, 0.f, 0.f, 0.f}};
                if (s + 2 < K) load_x(srow, s + 2, (s + 2) & 3);
                else load_x(srow_next, s + 2 - K, (s + 2) & 3);  // (the last receiver re-reads its own rows: never used)
                const u32x4* f = frag0 + (size_t)s * (SLOT_BYTES / 16);
                u32x4 b1[2], b2[2];
                b1[0] = f[0];
                b2[0] = f[64];
                auto kstep = [&](int kb) {  // same product order per accumulator as MmaStream16 (edge_f16.hip)
                    const int slot = kb & 1, nslot = slot ^ 1;
                    if (kb + 1 < NKB) {
                        b1[nslot] = f[(size_t)(kb + 1) * 128];
                        b2[nslot] = f[(size_t)(kb + 1) * 128 + 64];
                    }
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        am[mt] = mfma16_f16(A1[mt][kb], b1[slot], am[mt]);
                        ax[mt] = mfma16_f16(A1[mt][kb], b2[slot], ax[mt]);
                        ax[mt] = mfma16_f16(A2[mt][kb], b1[slot], ax[mt]);
                    }
                };
#pragma unroll
                for (int kb = 0; kb < NKB / 2; ++kb) kstep(kb);
                __syncthreads();  // SYNC_q
#pragma unroll
                for (int kb = NKB / 2; kb < NKB; ++kb) kstep(kb);
                const bool on = s < nd;  // product rounded, then added in edge order (messages -> index_add_)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float kv = fmaf(ax[mt][r], F16X3_INV_SCALE, am[mt][r]);
                        sum[mt][r] = on ? __fadd_rn(sum[mt][r], __fmul_rn(kv, xv[s & 3][mt][r])) : sum[mt][r];
                    }
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) *reinterpret_cast<f32x4v*>(&tile[par][c16 * CP_STR + cb + 16 * mt]) = sum[mt];
            par ^= 1;
            if (!has_next) break;
            m = mn;
            mn = next_valid(mn + m_step);
        }
        __syncthreads();  // the last receiver's tile is complete
        return;
    }

    // ================================================= mix role =================================================
    const int t2 = tid - 256;
    const int c = t2 & 127, ph = t2 >> 7;  // channel, half of the output orientations (p = 8 ph .. 8 ph + 7)
    const int w4 = wave - 4;               // this wave copies fragments 4 w4 .. 4 w4 + 3 of every slot block
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
    const unsigned ring0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)&ring[0][0]) + 4096u * w4;
    auto copy_slot = [&](int n, int s) {  // this wave's quarter of slot s of receiver n -> ring[s]
        const char* g = reinterpret_cast<const char*>(basis) + ((size_t)n * K + s) * SLOT_BYTES + 4096u * w4;
#pragma unroll
        for (int j = 0; j < 4; ++j) cp_glds16(lane16, g + 1024 * j, ring0 + s * SLOT_BYTES + 1024u * j);
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
    // prologue: slots 0 .. 6 of the first receiver are requested, slot 0 is waited for (24 younger copies stay in flight)
    {
        const int n = n0 + local_of(m);
#pragma unroll
        for (int s = 0; s < K - 1; ++s) copy_slot(n, s);
        cp_wait_but<24>();
        __syncthreads();  // SYNC_-1
    }
    int i = 0, par = 0, n_prev = 0;
    float out[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    while (true) {
        const int n = n0 + local_of(m);
        const bool has_next = mn < n_iter;
        const int n_next = has_next ? n0 + local_of(mn) : n;
#pragma unroll
        for (int s = 0; s < K; ++s) {
            // Before SYNC_q (q = 8 i + s) this wave's share of slot q + 1 must have landed.  Younger than that copy, in
            // the in-order queue: the copies of slots q + 2 .. q + 6 (4 each) and -- when step q - 6 .. q - 1 contains a
            // "phase 7" step that stored a receiver's result (s <= 5, from the third receiver on) -- those 8 stores.  The
            // last receiver of the sequence has no full set of younger copies: it waits for everything.
            if (!has_next) cp_wait_but<0>();
            else if (s <= 5 && i >= 2) cp_wait_but<28>();
            else cp_wait_but<20>();
            __syncthreads();  // SYNC_q
            if (s == 0) copy_slot(n, K - 1);              // slot q + 7: this receiver's last slot ...
            else if (has_next) copy_slot(n_next, s - 1);  // ... or slot s - 1 of the next receiver
            if (i > 0) {  // orientation mix of the previous receiver, two input orientations per step (o ascending)
#pragma unroll
                for (int oo = 0; oo < 2; ++oo) {
                    const int o = 2 * s + oo;
                    const float xo = tile[par ^ 1][o * CP_STR + c];
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
        par ^= 1;
        ++i;
        if (!has_next) break;
        m = mn;
        mn = next_valid(mn + m_step);
    }
    __syncthreads();  // the last receiver's tile is complete
#pragma unroll
    for (int o = 0; o < 16; ++o) {
        const float xo = tile[par ^ 1][o * CP_STR + c];
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
    ARREAU_LAUNCH((conv_proj_kernel<128, 256>), dim3(blocks), dim3(512), 0, s, reinterpret_cast<const u32x4*>(basis), wchunks, deg, src,
                  x_in, m->fk + (size_t)layer * 16 * 16 * m->C, m->conv_bias + (size_t)layer * m->C, n0, Ng, x_conv);
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}
