// K4b, fp16x3 split precision on v_mfma_f32_16x16x32_f16 (default): the ConvNext block of one layer
// (convnext.py:25-32) + read-out partials (ponita.py:105-117).  Same algorithm, chunking and ring protocol as
// node_f16.hip; what differs is the matrix instruction and with it the register layouts.
//
// Why: under the chip's power limit the 16x16x32 form holds a higher clock than 32x32x16 at the same cycles per FLOP
// (1875 vs 1572 MHz in bare streams, tools/exp/mfma_shape.hip).
//
// Layout.  One wave = 32 rows = two column blocks nb of 16: nb = node of the tile (n = 2 tile + nb), column
// c = lane & 15 = orientation.  Lane group g = lane >> 4 splits the K range: as B operand a lane holds, per 32-wide
// k-block kb, the 8 values e of its row with input index  32 kb + 16 (e >> 2) + 4 g + (e & 3)  (k order 8 g + e);
// as accumulator of the 16-row output tile mt of a 32-row chunk, register r is output row 16 mt + 4 g + r.  The
// accumulators of the two tiles mt of a chunk are therefore, in place, the B operand of the next layer's k-block
// (e = 4 mt + r): the chain stays in registers with no relayout, as in the 32x32 kernels.  Weights are packed to that
// k order on the host (pack_linear_f16x3_m16, native mode).
#include <stdlib.h>
#include <utility>

#include "f16x3.h"
#include "internal.h"

namespace {
// Sum over the 16 lanes of a DPP row (= the 16 orientations of a node at fixed g), every lane receiving the total.
__device__ __forceinline__ float row16_sum_m(float v) {
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, false));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{});  // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>{});  // row_mirror
    return v;
}
// sum over the four lane groups g of a column (lanes c, c+16, c+32, c+48)
__device__ __forceinline__ float group4_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
// planes of 8 fp32 values (e = 0..7) -> two u32x4 (8 halves each)
template <bool CLAMP, bool OWNED = false>
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
        unsigned a, b;
        split_pair_act<OWNED>(f32x2{v[2 * pp], v[2 * pp + 1]}, a, b);  // (activation planes: unscaled residual, f16x3.h)
        hi[pp] = a;
        lo[pp] = b;
    }
}
}  // namespace

template <int C, int H, int NW, int NB, int NSLOT>
__global__ __launch_bounds__(64 * NW, 2) void mlp_kernel_f16x3_m16(
    const float* __restrict__ x_conv,    // [N][16][C]  conv output (pre-LayerNorm)
    const float* __restrict__ x_in, float* __restrict__ x_out,
    const float* __restrict__ ln_w, const float* __restrict__ ln_b,
    const u32x4* __restrict__ stream,    // this layer: [4 quarters][W1 quarter: 4 chunks | W2 quarter: 4 chunks], 16 frags per chunk
    const float* __restrict__ mb1, const float* __restrict__ mb2, const float* __restrict__ ls,
    const float* __restrict__ wv,        // [C] vector read-out weights of this layer (column S of read_out_layers)
    float bv, int n0, int N /* nodes n0 .. N-1 */, int first_layer,
    float* __restrict__ xbar,            // [N][C] this layer
    float* __restrict__ vsum)            // [N][16]
{
    static_assert(C == 128 && H == 512, "chunking below assumes C = 128, H = 512");
    constexpr int KC = C / 32;            // k-blocks of the C inputs
    constexpr int HQ = H / 4, KH = HQ / 32;  // hidden quarter, its k-blocks
    constexpr int NF = KC * 4;            // 16 fragments (16 KiB) per chunk = 32 output rows
    static_assert(KH == KC, "W1 and W2 quarter chunks have the same size");
    // Ring of NSLOT x 16 KiB weight chunks; the copy of a chunk is issued P = NSLOT - 1 chunks ahead of its use.  With
    // three slots (P = 2) a copy has one chunk of matrix work (about 1.2 us at two waves per SIMD) to cross L2 -> LDS,
    // which is the order of the L2 latency itself; the fourth slot doubles that slack (2 x 66 KiB still fits two
    // workgroups per CU).
    static_assert(NSLOT == 3 || NSLOT == 4, "ring depth");
    constexpr int P = NSLOT - 1;
    constexpr int PER = NF / NW;                                      // DMA instructions per wave and chunk
    __shared__ u32x4 lds[NSLOT][NF * 64];
    __shared__ __attribute__((aligned(16))) float bias_s[H];      // mb1: no global loads while a DMA is in flight
    __shared__ __attribute__((aligned(16))) float ep_s[3][C];     // mb2, layer scale, vector read-out weights: the epilogue
                                                                  // takes them from LDS instead of three L2 round trips

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int cp = lane & 15, gp = lane >> 4;  // prologue / matrix-phase copies
    // (32-bit: the launcher's grid is ceil((N - n0) / NB / NW) workgroups, so n0 + NB * tile + nb < N + NB * NW fits an int)
    const int tile = (int)blockIdx.x * NW + wave;
    const bool active = n0 + NB * tile < N;  // wave-uniform; idle waves still copy weights and meet the barriers
    int nrow[NB];                      // node of column block nb (padding: a valid node, nothing written)
    bool valid[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int n_i = n0 + NB * tile + nb;
        valid[nb] = n_i < N;
        nrow[nb] = valid[nb] ? n_i : N - 1;
    }

    const u32x4* dma_src = stream;  // next chunk to copy
    dma_chunk<NF, NW>(dma_src, lds[0], wave, lane);
    dma_src += (size_t)NF * 64;
    dma_chunk<NF, NW>(dma_src, lds[1], wave, lane);
    dma_src += (size_t)NF * 64;
    if constexpr (P == 3) {
        dma_chunk<NF, NW>(dma_src, lds[2], wave, lane);
        dma_src += (size_t)NF * 64;
    }
    for (int i = threadIdx.x; i < H; i += 64 * NW) bias_s[i] = mb1[i];
    for (int i = threadIdx.x; i < C; i += 64 * NW) {
        ep_s[0][i] = mb2[i];
        ep_s[1][i] = ls[i];
        ep_s[2][i] = wv[i];
    }

    // ---- load the rows in B-operand layout, LayerNorm them (eps 1e-5, biased variance), split ---------------
    u32x4 xn[NB][KC][2];  // [column block][k-block][plane]
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const float* rowp = x_conv + ((size_t)nrow[nb] * 16 + cp) * C + 4 * gp;
        float x[KC][8];
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < KC; ++kb)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(rowp + 32 * kb + 16 * hf);
#pragma unroll
                for (int r = 0; r < 4; ++r) x[kb][4 * hf + r] = v[r];
                sum += (v[0] + v[1]) + (v[2] + v[3]);
            }
        const float mean = group4_sum(sum) * (1.0f / C);
        float sq = 0.f;
#pragma unroll
        for (int kb = 0; kb < KC; ++kb)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float dlt = x[kb][e] - mean;
                x[kb][e] = dlt;
                sq += dlt * dlt;
            }
        const float rstd = 1.0f / sqrtf(group4_sum(sq) * (1.0f / C) + 1e-5f);
#pragma unroll
        for (int kb = 0; kb < KC; ++kb) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const f32x4 gw = *reinterpret_cast<const f32x4*>(ln_w + 32 * kb + 16 * hf + 4 * gp);
                const f32x4 gb = *reinterpret_cast<const f32x4*>(ln_b + 32 * kb + 16 * hf + 4 * gp);
#pragma unroll
                for (int r = 0; r < 4; ++r) x[kb][4 * hf + r] = x[kb][4 * hf + r] * rstd * gw[r] + gb[r];
            }
            split8<true>(x[kb], xn[nb][kb][0], xn[nb][kb][1]);
        }
    }
    dma_wait();
    __syncthreads();

    f32x4v acc_o[KC][2][NB];  // [output chunk u][16-row tile mt][column block nb]
#pragma unroll
    for (int u = 0; u < KC; ++u)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc_o[u][mt][nb] = f32x4v{0.f, 0.f, 0.f, 0.f};

    // 32 chunks per layer: per quarter 4 chunks of W1 rows (32 hidden units each) then 4 chunks of W2 columns (32
    // outputs each).  Chunk q sits in ring slot q % 3; SYNC_q in the middle of its MFMA stream (edge_f16.hip).
    int sl = 0;
    const unsigned lane16 = 16u * lane;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)&lds[0][0]) + 1024u * wave;
    // SYNC of chunk q (mid-chunk): chunk q + 1 must have landed for every wave; this wave's younger copies (chunks
    // q + 2 .. q + P - 1, as far as they exist) stay in flight -- vmcnt retires in issue order.  Then the copy of chunk
    // q + P starts into the slot chunk q - 1 just left.
    int q = 0;  // chunk index within the layer (32 chunks)
    constexpr int NCHUNK = 8 * 4;
    auto sync = [&]() {
        const int younger = min(max(NCHUNK - 2 - q, 0), P - 2);
        if (P == 3 && younger == 1) dma_wait_but<PER>();
        else dma_wait();
        __syncthreads();
        if (q + P < NCHUNK) dma_chunk_lean<NF, NW>(dma_src, lane16, wave, lds0 + (unsigned)(sl == 0 ? NSLOT - 1 : sl - 1) * (NF * 1024u));
        dma_src += (size_t)NF * 64;
        ++q;
    };
    constexpr int ST = KC;  // step (of 2 KC) at which the barrier is taken
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
        u32x4 hid[NB][KH][2];  // [column block][k-block = hidden chunk][plane]
#pragma unroll
        for (int u = 0; u < KH; ++u) {  // ---- hidden chunk u = GELU(W1q[u] . xn + b1q[u]) ----
            struct { f32x4v m[2][NB], x[2][NB]; } acc;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(bias_s + w * HQ + 32 * u + 16 * mt + 4 * gp);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    acc.m[mt][nb] = f32x4v{b[0], b[1], b[2], b[3]};
                    acc.x[mt][nb] = f32x4v{0.f, 0.f, 0.f, 0.f};
                }
            }
            if (active) mma16_range<KC, 0, ST, NB>(acc.m, acc.x, lds[sl], xn, lane);
            sync();
            if (active) {
                mma16_range<KC, ST, 2 * KC, NB>(acc.m, acc.x, lds[sl], xn, lane);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    float v[8];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int pr = 0; pr < 2; ++pr) {
                            const f32x2 pre = fma2(f32x2{acc.x[mt][nb][2 * pr], acc.x[mt][nb][2 * pr + 1]}, splat2(F16X3_INV_SCALE),
                                                   f32x2{acc.m[mt][nb][2 * pr], acc.m[mt][nb][2 * pr + 1]});
                            const f32x2 act = gelu_fast2(pre);
                            v[4 * mt + 2 * pr] = act.x;
                            v[4 * mt + 2 * pr + 1] = act.y;
                        }
                    split8<false>(v, hid[nb][u][0], hid[nb][u][1]);
                }
            }
            sl = sl == NSLOT - 1 ? 0 : sl + 1;
        }
#pragma unroll
        for (int u = 0; u < KC; ++u) {  // ---- output chunk u += W2[:, quarter][u] . hid ----
            f32x4v cross[2][NB];  // acc_o[u] itself is the main accumulator
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) cross[mt][nb] = f32x4v{0.f, 0.f, 0.f, 0.f};
            if (active) mma16_range<KH, 0, ST, NB>(acc_o[u], cross, lds[sl], hid, lane);
            sync();
            if (active) {
                mma16_range<KH, ST, 2 * KH, NB>(acc_o[u], cross, lds[sl], hid, lane);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc_o[u][mt][nb][r] = fmaf(cross[mt][nb][r], F16X3_INV_SCALE, acc_o[u][mt][nb][r]);
            }
            sl = sl == NSLOT - 1 ? 0 : sl + 1;
        }
    }
    // (round 4, lint rule ldsdma-unwaited-exit: no LDS-DMA copy is left in flight when a wave ends -- the last copies of a ring
    // target a chunk nobody will read; the hardware's implicit wait at s_endpgm is not relied upon)
    dma_wait();
    if (!active) return;

    // ---- bias, layer scale, residual; write x_out; read-out partials (all in registers) ------------------
    // (lane-derived indices are re-derived here from a value the optimiser cannot see through, so that they do not
    // occupy registers through the matrix phases)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int c = lane_e & 15, g = lane_e >> 4;
    const float inv16 = 1.0f / 16.0f;
    float vdot[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) vdot[nb] = 0.f;
#pragma unroll
    for (int u = 0; u < KC; ++u)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int c0 = 32 * u + 16 * mt + 4 * g;  // this lane's four channels
            const f32x4 b2v = *reinterpret_cast<const f32x4*>(&ep_s[0][c0]);
            const f32x4 lsv = *reinterpret_cast<const f32x4*>(&ep_s[1][c0]);
            const f32x4 wvv = *reinterpret_cast<const f32x4*>(&ep_s[2][c0]);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const size_t off = ((size_t)nrow[nb] * 16 + c) * C + c0;
                const f32x4 xi = *reinterpret_cast<const f32x4*>(x_in + off);
                f32x4 xo;
#pragma unroll
                for (int r = 0; r < 4; ++r) xo[r] = (acc_o[u][mt][nb][r] + b2v[r]) * lsv[r] + xi[r];
                if (valid[nb]) *reinterpret_cast<f32x4*>(x_out + off) = xo;
#pragma unroll
                for (int r = 0; r < 4; ++r) vdot[nb] += xo[r] * wvv[r];
                f32x4 sum;
#pragma unroll
                for (int r = 0; r < 4; ++r) sum[r] = row16_sum_m(xo[r]);
                if (valid[nb] && c == 0) {
                    const f32x4 mean = {sum[0] * inv16, sum[1] * inv16, sum[2] * inv16, sum[3] * inv16};
                    *reinterpret_cast<f32x4*>(xbar + (size_t)nrow[nb] * C + c0) = mean;
                }
            }
        }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const float tot = group4_sum(vdot[nb]);
        if (valid[nb] && g == 0) {
            const size_t o = (size_t)nrow[nb] * 16 + c;
            vsum[o] = (first_layer ? 0.0f : vsum[o]) + (tot + bv);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Small-launch form: ONE NODE PER WORKGROUP, the layer's work dealt to eight waves -- bit-identical to the kernel above.
//
// With few nodes (1 x 8 atoms: 8 tiles of 16 rows) the kernel above is one wave walking through all 32 weight chunks of
// a layer: 22 us of pure latency while 250 CUs idle.  Here
//   phase 1: wave w computes hidden chunks (quarter w >> 1, chunks 2 (w & 1), 2 (w & 1) + 1): 64 of the 512 hidden units
//            (the same instruction sequence per chunk: bias, 8 steps over the LayerNorm'ed input, GELU, plane split); the
//            sixteen chunks meet in LDS as the B operand of linear_2;
//   phase 2: wave w owns ONE 16-channel output tile (chunk u = w >> 1, tile mt = w & 1) and contracts it over all 512
//            hidden units in the order of the kernel above -- quarter by quarter, the cross terms folded into the
//            accumulator after each quarter -- then runs that tile's epilogue (operands requested at the top);
//   the vector read-out needs the tiles' outputs in ONE chain of fused multiply-adds, so they meet in LDS and wave 0 walks
//   the chain in the original order.
// A wave's 64 + 32 weight fragments come straight from the packed stream in L2 into registers (no LDS ring, no counted
// waits).  Every number is produced by the instruction sequence of the kernel above, so the launcher picks the form by
// size (tests: test_small_launch_kernels_are_bit_identical...).
// ---------------------------------------------------------------------------------------------------------------------
namespace {
template <int NST>
__device__ __forceinline__ void load_steps(u32x4 (&w)[2 * NST], const u32x4* __restrict__ chunk, int st0, int lane) {
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        w[2 * i] = chunk[(size_t)(st0 + i) * 128 + lane];
        w[2 * i + 1] = chunk[(size_t)(st0 + i) * 128 + 64 + lane];
    }
}
}  // namespace

// FUSE (round 3): the workgroup first evaluates its node's message passing + spherical convolution itself -- the arithmetic
// of conv_kernel / conv_kernel_streamed (node.hip) on an fp32 K block, same thread roles (512 threads), same order of the
// rounded products and of the mix -- and keeps the 16 convolved rows in LDS instead of reading them back from x_conv: one
// launch per layer instead of two where launches, not bytes, are what a step costs (1 x 8: 15 -> 10 launches per step).
// TRAIN (round 5): the training forward of the block (arreau_general_network, train_net.hip) through this kernel instead of a
// LayerNorm launch and two products with epilogues: the same numbers as the sampling step, plus what the backward pass reads --
// xhat, rstd, the LayerNorm output (wave 0: every wave normalises the 16 rows for its own B operand), every hidden unit before and
// after the GELU (the wave that owns the chunk, 16-byte pieces of four consecutive units per row) and linear_2's output with its
// bias (the wave that owns the tile).  No read-out partials (the training read-outs are one batched product over the kept x_l).
// FUSE + TRAIN (round 5, second session): the training forward's spatial conv + spherical mix inside the same launch too -- one launch
// per layer instead of two: the layer's kernels are a column block of the all-layer matrix [R][L C] (row pitch kl_pitch), the fiber
// kernel carries no 1 / 16 (the training branch divides after the mix, as conv.py does), and x_1 -- which the backward pass reads --
// leaves beside the other saved activations.
struct MlpTrainSave {
    float *xhat, *rstd, *xn, *hpre, *h, *out;  // [M][C], [M], [M][C], [M][H], [M][H], [M][C] of this layer (M = 16 N rows)
    float* x1 = nullptr;                        // FUSE: [M][C] the spatial conv's output
    int kl_pitch = 0;                           // FUSE: floats between consecutive rows of `kl`
};
template <int C, int H, bool FUSE = false, bool TRAIN = false>
__global__ __launch_bounds__(512) void mlp_kernel_f16x3_m16_split(
    const float* __restrict__ x_conv, const float* __restrict__ x_in, float* __restrict__ x_out,
    const float* __restrict__ ln_w, const float* __restrict__ ln_b, const u32x4* __restrict__ stream,
    const float* __restrict__ mb1, const float* __restrict__ mb2, const float* __restrict__ ls,
    const float* __restrict__ wv, float bv, int n0, int first_layer, float* __restrict__ xbar, float* __restrict__ vsum,
    const float* __restrict__ kl /* FUSE: this layer's kernels [N*8*16][C] fp32 */, const int32_t* __restrict__ deg,
    const int32_t* __restrict__ src, const float* __restrict__ fk /* [16][16][C] */, const float* __restrict__ conv_bias,
    MlpTrainSave save) {
    static_assert(C == 128 && H == 512, "chunking below assumes C = 128, H = 512");
    constexpr int KC = C / 32, HQ = H / 4;
    constexpr int TS = 132;             // row stride of the two conv tiles (node.hip: CONV_LDS_STRIDE)
    __shared__ u32x4 hidx[16][2][64];   // hidden chunk (quarter * 4 + u) as B operand: [chunk][plane][lane], 32 KiB
    __shared__ f32x4 xo_s[8][64];       // the eight output tiles' x_out values, for the vector read-out chain
    __shared__ __attribute__((aligned(16))) float ctile[FUSE ? 16 * TS : 4];  // FUSE: messages summed per orientation row
    __shared__ __attribute__((aligned(16))) float xc_s[FUSE ? 16 * TS : 4];   // FUSE: the node's 16 convolved rows

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int q = wave >> 1, e = wave & 1;     // phase 1: hidden quarter, half of the quarter
    const int cp = lane & 15, gp = lane >> 4;
    const int n = n0 + blockIdx.x;
    const u32x4* qs = stream + (size_t)q * 8 * 1024;  // this quarter: 4 linear_1 chunks, then 4 linear_2 chunks

    // batch 1 of the weights: this wave's two linear_1 chunks (u = 2 e, 2 e + 1), 16 fragments each
    u32x4 wa[16], wb[16];
    load_steps<8>(wa, qs + (size_t)(2 * e) * 1024, 0, lane);
    load_steps<8>(wb, qs + (size_t)(2 * e + 1) * 1024, 0, lane);
    // operands of this wave's output tile (chunk u = wave >> 1, tile mt = wave & 1), requested now
    const int ou = wave >> 1, omt = wave & 1;
    const int ep_c0 = 32 * ou + 16 * omt + 4 * gp;  // this lane's four channels
    const size_t ep_off = ((size_t)n * 16 + cp) * C + ep_c0;
    const f32x4 ep_b2 = *reinterpret_cast<const f32x4*>(mb2 + ep_c0);
    const f32x4 ep_ls = *reinterpret_cast<const f32x4*>(ls + ep_c0);
    const f32x4 ep_xi = *reinterpret_cast<const f32x4*>(x_in + ep_off);

    if constexpr (FUSE) {
        // conv.py:111,131-133 + index_add_ over the in-edges, then the depth-wise orientation mix (conv.py:113-127): node.hip's
        // conv_kernel, for this workgroup's node
        const int tid = threadIdx.x;
        const int c = tid & 127, pq = tid >> 7;     // mix role: channel, quarter of the output orientations
        const int c4 = tid & 31, o_row = tid >> 5;  // gather role: float4 column, orientation row
        constexpr int K = 8;
        const int nd = min(deg[n], K);
        const int32_t* srow = src + (size_t)n * K;
        const size_t kp = TRAIN ? (size_t)save.kl_pitch : (size_t)C;
        const size_t kbase = ((size_t)n * K * 16 + o_row) * kp + 4 * c4;
        f32x4 kv[K], xv[K];
#pragma unroll
        for (int s_ = 0; s_ < K; ++s_) {  // unused slots: any valid row, dropped by the select below (their K rows may be uninitialised)
            const int sn = max(srow[s_], 0);
            kv[s_] = *reinterpret_cast<const f32x4*>(kl + kbase + (size_t)s_ * 16 * kp);
            xv[s_] = *reinterpret_cast<const f32x4*>(x_in + ((size_t)sn * 16 + o_row) * C + 4 * c4);
        }
        float fkr[16][4];
#pragma unroll
        for (int o = 0; o < 16; ++o)
#pragma unroll
            for (int p = 0; p < 4; ++p) fkr[o][p] = fk[((size_t)o * 16 + (4 * pq + p)) * C + c];
        const float cbias = conv_bias[c];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s_ = 0; s_ < K; ++s_) {
            const bool on = s_ < nd;  // product rounded, then added in edge order (messages -> index_add_)
            acc[0] = on ? __fadd_rn(acc[0], __fmul_rn(kv[s_][0], xv[s_][0])) : acc[0];
            acc[1] = on ? __fadd_rn(acc[1], __fmul_rn(kv[s_][1], xv[s_][1])) : acc[1];
            acc[2] = on ? __fadd_rn(acc[2], __fmul_rn(kv[s_][2], xv[s_][2])) : acc[2];
            acc[3] = on ? __fadd_rn(acc[3], __fmul_rn(kv[s_][3], xv[s_][3])) : acc[3];
        }
        *reinterpret_cast<f32x4*>(&ctile[o_row * TS + 4 * c4]) = acc;
        if constexpr (TRAIN) *reinterpret_cast<f32x4*>(save.x1 + ((size_t)n * 16 + o_row) * C + 4 * c4) = acc;
        __syncthreads();
        float out[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < 16; ++o) {
            const float xo = ctile[o * TS + c];
#pragma unroll
            for (int p = 0; p < 4; ++p) out[p] += xo * fkr[o][p];
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) xc_s[(4 * pq + p) * TS + c] = TRAIN ? out[p] * (1.0f / 16.0f) + cbias : out[p] + cbias;
        __syncthreads();
    }

    // ---- the node's 16 rows in B-operand layout, LayerNorm (eps 1e-5, biased variance), split (as in the kernel above) ----
    u32x4 xn[KC][2];
    {
        const float* rowp = FUSE ? xc_s + cp * TS + 4 * gp : x_conv + ((size_t)n * 16 + cp) * C + 4 * gp;
        float x[KC][8];
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < KC; ++kb)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(rowp + 32 * kb + 16 * hf);
#pragma unroll
                for (int r = 0; r < 4; ++r) x[kb][4 * hf + r] = v[r];
                sum += (v[0] + v[1]) + (v[2] + v[3]);
            }
        const float mean = group4_sum(sum) * (1.0f / C);
        float sq = 0.f;
#pragma unroll
        for (int kb = 0; kb < KC; ++kb)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float dlt = x[kb][i] - mean;
                x[kb][i] = dlt;
                sq += dlt * dlt;
            }
        const float rstd = 1.0f / sqrtf(group4_sum(sq) * (1.0f / C) + 1e-5f);
        if constexpr (TRAIN) {
            if (wave == 0 && gp == 0) save.rstd[(size_t)n * 16 + cp] = rstd;
        }
#pragma unroll
        for (int kb = 0; kb < KC; ++kb) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const f32x4 gw = *reinterpret_cast<const f32x4*>(ln_w + 32 * kb + 16 * hf + 4 * gp);
                const f32x4 gb = *reinterpret_cast<const f32x4*>(ln_b + 32 * kb + 16 * hf + 4 * gp);
                if constexpr (TRAIN) {
                    if (wave == 0) {
                        const size_t o = ((size_t)n * 16 + cp) * C + 32 * kb + 16 * hf + 4 * gp;
                        f32x4 xh, xo;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            xh[r] = x[kb][4 * hf + r] * rstd;
                            xo[r] = x[kb][4 * hf + r] * rstd * gw[r] + gb[r];
                        }
                        *reinterpret_cast<f32x4*>(save.xhat + o) = xh;
                        *reinterpret_cast<f32x4*>(save.xn + o) = xo;
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) x[kb][4 * hf + r] = x[kb][4 * hf + r] * rstd * gw[r] + gb[r];
            }
            split8<true, FUSE && TRAIN>(x[kb], xn[kb][0], xn[kb][1]);
        }
    }

    // ---- phase 1: two hidden chunks of 32, GELU, split -> the shared B operand of linear_2 ----------------------------
    auto hidden_chunk = [&](const u32x4 (&w)[16], int j) {
        const int u = 2 * e + j;
        f32x4v am[2], ax[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(mb1 + q * HQ + 32 * u + 16 * mt + 4 * gp);
            am[mt] = f32x4v{b[0], b[1], b[2], b[3]};
            ax[mt] = f32x4v{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int st = 0; st < 2 * KC; ++st) {
            const int kb = st >> 1, mt = st & 1;
            am[mt] = mfma16_f16(w[2 * st], xn[kb][0], am[mt]);
            ax[mt] = mfma16_f16(w[2 * st + 1], xn[kb][0], ax[mt]);
            am[mt] = mfma16_f16(w[2 * st], xn[kb][1], am[mt]);  // (activation planes: the residual is unscaled)
        }
        float v[8];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const f32x2 pre = fma2(f32x2{ax[mt][2 * pr], ax[mt][2 * pr + 1]}, splat2(F16X3_INV_SCALE),
                                       f32x2{am[mt][2 * pr], am[mt][2 * pr + 1]});
                const f32x2 act = gelu_fast2(pre);
                v[4 * mt + 2 * pr] = act.x;
                v[4 * mt + 2 * pr + 1] = act.y;
                if constexpr (TRAIN) {  // row (n, cp), hidden units q HQ + 32 u + 16 mt + 4 gp + 2 pr, + 1
                    const size_t o = ((size_t)n * 16 + cp) * H + q * HQ + 32 * u + 16 * mt + 4 * gp + 2 * pr;
                    typedef float f2v __attribute__((ext_vector_type(2)));
                    *reinterpret_cast<f2v*>(save.hpre + o) = f2v{pre.x, pre.y};
                    *reinterpret_cast<f2v*>(save.h + o) = f2v{act.x, act.y};
                }
            }
        u32x4 hi, lo;
        split8<false, FUSE && TRAIN>(v, hi, lo);
        hidx[4 * q + u][0][lane] = hi;
        hidx[4 * q + u][1][lane] = lo;
    };
    // batch 2 of the weights: of linear_2 chunk (quarter w2, u = ou) the steps of this wave's tile: st = 2 kb + omt
    auto load_out = [&](u32x4 (&w)[16], int w0, int w2) {
        const u32x4* chunk = stream + ((size_t)w2 * 8 + 4 + ou) * 1024 + lane;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            w[w0 + 2 * kb] = chunk[(size_t)(2 * kb + omt) * 128];
            w[w0 + 2 * kb + 1] = chunk[(size_t)(2 * kb + omt) * 128 + 64];
        }
    };
    hidden_chunk(wa, 0);
    load_out(wa, 0, 0);
    load_out(wa, 8, 1);
    hidden_chunk(wb, 1);
    load_out(wb, 0, 2);
    load_out(wb, 8, 3);
    __syncthreads();

    // ---- phase 2: this wave's output tile over all 512 hidden units, quarter by quarter (order of the kernel above) -----
    f32x4v acc = f32x4v{0.f, 0.f, 0.f, 0.f};
    auto quarter = [&](const u32x4 (&w)[16], auto w0c, int w2) {
        constexpr int w0 = decltype(w0c)::value;
        f32x4v cross = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const u32x4 bh = hidx[4 * w2 + kb][0][lane], bl = hidx[4 * w2 + kb][1][lane];
            acc = mfma16_f16(w[w0 + 2 * kb], bh, acc);
            cross = mfma16_f16(w[w0 + 2 * kb + 1], bh, cross);
            acc = mfma16_f16(w[w0 + 2 * kb], bl, acc);  // (activation planes: the residual is unscaled)
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = fmaf(cross[r], F16X3_INV_SCALE, acc[r]);
    };
    quarter(wa, std::integral_constant<int, 0>{}, 0);
    quarter(wa, std::integral_constant<int, 8>{}, 1);
    quarter(wb, std::integral_constant<int, 0>{}, 2);
    quarter(wb, std::integral_constant<int, 8>{}, 3);

    // ---- epilogue of the tile: bias, layer scale, residual, x_out, per-channel orientation means ------------------------
    {
        f32x4 xo;
#pragma unroll
        for (int r = 0; r < 4; ++r) xo[r] = (acc[r] + ep_b2[r]) * ep_ls[r] + ep_xi[r];
        *reinterpret_cast<f32x4*>(x_out + ep_off) = xo;
        if constexpr (TRAIN) {
            *reinterpret_cast<f32x4*>(save.out + ep_off) = f32x4{acc[0] + ep_b2[0], acc[1] + ep_b2[1], acc[2] + ep_b2[2], acc[3] + ep_b2[3]};
            return;  // (no barrier follows on this path)
        }
        f32x4 sum;
#pragma unroll
        for (int r = 0; r < 4; ++r) sum[r] = row16_sum_m(xo[r]);
        if (cp == 0) {
            const float inv16 = 1.0f / 16.0f;
            const f32x4 mean = {sum[0] * inv16, sum[1] * inv16, sum[2] * inv16, sum[3] * inv16};
            *reinterpret_cast<f32x4*>(xbar + (size_t)n * C + ep_c0) = mean;
        }
        xo_s[wave][lane] = xo;
    }
    __syncthreads();
    if (wave != 0) return;
    // vector read-out partial: the chain of the kernel above over (u, mt, r), then the four lane groups
    float vdot = 0.f;
#pragma unroll
    for (int u = 0; u < KC; ++u)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const f32x4 xo = xo_s[2 * u + mt][lane];
            const f32x4 wvv = *reinterpret_cast<const f32x4*>(wv + 32 * u + 16 * mt + 4 * gp);
#pragma unroll
            for (int r = 0; r < 4; ++r) vdot += xo[r] * wvv[r];
        }
    const float tot = group4_sum(vdot);
    if (gp == 0) {
        const size_t o = (size_t)n * 16 + cp;
        vsum[o] = (first_layer ? 0.0f : vsum[o]) + (tot + bv);
    }
}

#define ARREAU_MLP_SPLIT_MAX_NODES 512
int arreau_launch_mlp_f16x3_m16_split(const arreau_model* m, int layer, const float* x_conv, const float* x_in, float* x_out,
                                      float* xbar, float* vsum, int Ntot, hipStream_t s, NodeRange r) {
    const int n0 = r.n0, n1 = r.n1 < 0 ? Ntot : r.n1;
    if (n1 <= n0) return ARREAU_OK;
    const int C = m->C, H = m->H;
    if (!(C == 128 && H == 512)) {
        arreau_set_error("mlp kernel (fp16x3, hidden split): unsupported (hidden_dim, widening_factor)");
        return ARREAU_EINVAL;
    }
    const size_t layer_u32x4 = (size_t)2 * H * C * 2 * 2 / 16;
    const u32x4* stream = reinterpret_cast<const u32x4*>(m->mlp_f16m) + (size_t)layer * layer_u32x4;
    ARREAU_LAUNCH((mlp_kernel_f16x3_m16_split<128, 512>), dim3((unsigned)(n1 - n0)), dim3(512), 0, s, x_conv, x_in, x_out,
                       m->ln_w + (size_t)layer * C, m->ln_b + (size_t)layer * C, stream, m->mb1 + (size_t)layer * H,
                       m->mb2 + (size_t)layer * C, m->ls + (size_t)layer * C, m->ro_wv + (size_t)layer * C, m->ro_bv_host[layer],
                       n0, layer == 0 ? 1 : 0, xbar + (size_t)layer * Ntot * C, vsum, (const float*)nullptr, (const int32_t*)nullptr,
                       (const int32_t*)nullptr, (const float*)nullptr, (const float*)nullptr, MlpTrainSave{});
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}

// Message passing + spherical convolution + ConvNext block of one layer in ONE launch (small unsliced launches on an fp32 K
// buffer with k = 8; see the kernel): what conv_kernel_streamed<128, false> followed by the launch above computes, bit for bit.
bool arreau_small_layer_fusable(const arreau_model* m, int N, NodeRange r) {
    const char* e = getenv("ARREAU_FUSE_SMALL");  // 0: two launches per layer (A/B, tests); read per call
    static const int split_env = [] { const char* v = getenv("ARREAU_MLP_SPLIT"); return v ? atoi(v) : -1; }();
    const int n0 = r.n0, n1 = r.n1 < 0 ? N : r.n1;
    const bool whole_batch = n0 == 0 && n1 == N && r.wg_cap == 0;
    return (e == nullptr || atoi(e) != 0) && split_env < 0 && whole_batch && N <= ARREAU_MLP_SPLIT_MAX_NODES && m->mlp_variant == 3 &&
           m->f16_ok && m->k == 8 && m->C == 128 && m->H == 512 && (m->conv_variant == 1 || m->conv_variant == 2) && !arreau_k3(m) &&
           !arreau_basis_form(m, N);
}
int arreau_launch_small_layer(const arreau_model* m, int layer, const float* kbuf, const int32_t* deg, const int32_t* src,
                              const float* x_in, float* x_out, float* xbar, float* vsum, int Ntot, hipStream_t s) {
    if (Ntot <= 0) return ARREAU_OK;
    const int C = m->C, H = m->H;
    const size_t layer_u32x4 = (size_t)2 * H * C * 2 * 2 / 16;
    const u32x4* stream = reinterpret_cast<const u32x4*>(m->mlp_f16m) + (size_t)layer * layer_u32x4;
    const size_t layer_stride = (size_t)Ntot * m->k * 16 * C;
    ARREAU_LAUNCH((mlp_kernel_f16x3_m16_split<128, 512, true>), dim3((unsigned)Ntot), dim3(512), 0, s, (const float*)nullptr, x_in, x_out,
                       m->ln_w + (size_t)layer * C, m->ln_b + (size_t)layer * C, stream, m->mb1 + (size_t)layer * H,
                       m->mb2 + (size_t)layer * C, m->ls + (size_t)layer * C, m->ro_wv + (size_t)layer * C, m->ro_bv_host[layer],
                       0, layer == 0 ? 1 : 0, xbar + (size_t)layer * Ntot * C, vsum, kbuf + layer_stride * layer, deg, src,
                       m->fk + (size_t)layer * 16 * 16 * C, m->conv_bias + (size_t)layer * C, MlpTrainSave{});
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}

// The ConvNext block of the TRAINING forward (train_net.hip): x_conv = the layer's spherical-conv output with bias, x_in / x_out the
// residual stream, everything the backward pass reads written beside (see MlpTrainSave).  The weights come from the packed plane
// stream like in sampling: arreau_repack_mlp_f16x3_m16 below rebuilds it from the fp32 training weights after every optimizer step.
bool arreau_mlp_train_forward_available(const arreau_model* m) {
    const char* e = getenv("ARREAU_TRAIN_FUSED_MLP");  // 0: LayerNorm launch + two products (A/B, tests); read per call
    return (!e || atoi(e) != 0) && m->C == 128 && m->H == 512 && m->mlp_f16m != nullptr;
}
int arreau_launch_mlp_train_forward(const arreau_model* m, int layer, const float* x_conv, const float* x_in, float* x_out, float* xhat,
                                    float* rstd, float* xn, float* hpre, float* h, float* out, int N, hipStream_t s,
                                    const float* kl, int kl_pitch, const int32_t* deg, const int32_t* src, const float* fk, float* x1) {
    if (N <= 0) return ARREAU_OK;
    const int C = m->C, H = m->H;
    const size_t layer_u32x4 = (size_t)2 * H * C * 2 * 2 / 16;
    const u32x4* stream = reinterpret_cast<const u32x4*>(m->mlp_f16m) + (size_t)layer * layer_u32x4;
    if (kl != nullptr) {  // conv + mix + ConvNext block in one launch (k = 8; x_conv unused)
        ARREAU_REQUIRE(m->k == 8 && kl_pitch % 4 == 0 && deg && src && fk && x1, "fused training conv: k = 8 and 16-byte kernel rows");
        MlpTrainSave sv{xhat, rstd, xn, hpre, h, out};
        sv.x1 = x1; sv.kl_pitch = kl_pitch;
        ARREAU_LAUNCH((mlp_kernel_f16x3_m16_split<128, 512, true, true>), dim3((unsigned)N), dim3(512), 0, s, (const float*)nullptr, x_in, x_out,
                      m->ln_w + (size_t)layer * C, m->ln_b + (size_t)layer * C, stream, m->mb1 + (size_t)layer * H, m->mb2 + (size_t)layer * C,
                      m->ls + (size_t)layer * C, (const float*)nullptr, 0.0f, 0, 0, (float*)nullptr, (float*)nullptr, kl, deg, src, fk,
                      m->conv_bias + (size_t)layer * C, sv);
        ARREAU_CHECK_HIP(hipGetLastError());
        return ARREAU_OK;
    }
    ARREAU_LAUNCH((mlp_kernel_f16x3_m16_split<128, 512, false, true>), dim3((unsigned)N), dim3(512), 0, s, x_conv, x_in, x_out,
                  m->ln_w + (size_t)layer * C, m->ln_b + (size_t)layer * C, stream, m->mb1 + (size_t)layer * H, m->mb2 + (size_t)layer * C,
                  m->ls + (size_t)layer * C, (const float*)nullptr, 0.0f, 0, 0, (float*)nullptr, (float*)nullptr, (const float*)nullptr,
                  (const int32_t*)nullptr, (const int32_t*)nullptr, (const float*)nullptr, (const float*)nullptr,
                  MlpTrainSave{xhat, rstd, xn, hpre, h, out});
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}

// Device form of model.hip's pack_linear_f16x3_m16(native) for the block's two Linears: per layer and hidden quarter w, the quarter's
// rows of linear_1 [H][C] (32,768 halves: [u][kb][mt][plane][lane][e]) then its columns of linear_2 [C][H] (the same), both fp16
// planes (hi, residual * 2^11; round to nearest even like the host packer: the streams agree bit for bit).
namespace {
__global__ __launch_bounds__(256) void repack_mlp_f16x3_m16_kernel(const float* __restrict__ lin1 /*[L][H][C]*/, const float* __restrict__ lin2 /*[L][C][H]*/,
                                                                    int L, unsigned short* __restrict__ q) {
    constexpr int C = 128, H = 512, HQ = 128;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;   // one (layer, quarter, matrix, u, kb, mt, lane, e)
    const long per_mat = 4L * 4 * 2 * 64 * 8;                // 16,384 elements of a 128 x 128 quarter
    if (i >= (long)L * 4 * 2 * per_mat) return;
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63), mt = (int)((i >> 9) & 1), kb = (int)((i >> 10) & 3), u = (int)((i >> 12) & 3);
    const int mat = (int)((i >> 14) & 1), w = (int)((i >> 15) & 3), l = (int)(i >> 17);
    const int g = lane >> 4;
    const int o = 32 * u + 16 * mt + (lane & 15);                     // output row of the quarter's matrix
    const int k = 32 * kb + 16 * (e >> 2) + 4 * g + (e & 3);          // input column (native k order)
    const float v = mat == 0 ? lin1[((size_t)l * H + (size_t)w * HQ + o) * C + k] : lin2[((size_t)l * C + o) * H + (size_t)w * HQ + k];
    const _Float16 h1 = (_Float16)v;
    const _Float16 h2 = (_Float16)((v - (float)h1) * 2048.0f);
    const size_t base = (((size_t)l * 4 + w) * 2 + mat) * (size_t)(2 * per_mat) + (((size_t)u * 4 + kb) * 2 + mt) * 2 * 512 + (size_t)lane * 8 + e;
    q[base] = __builtin_bit_cast(unsigned short, h1);
    q[base + 512] = __builtin_bit_cast(unsigned short, h2);
}
}  // namespace
int arreau_repack_mlp_f16x3_m16(arreau_model* m, hipStream_t s) {
    if (!(m->C == 128 && m->H == 512 && m->mlp_f16m)) return ARREAU_OK;
    const long n = (long)m->L * 4 * 2 * 16384;
    ARREAU_LAUNCH(repack_mlp_f16x3_m16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, m->t_lin1, m->t_lin2, m->L,
                  reinterpret_cast<unsigned short*>(const_cast<float*>(m->mlp_f16m)));
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}

int arreau_launch_mlp_f16x3_m16(const arreau_model* m, int layer, const float* x_conv, const float* x_in, float* x_out,
                                float* xbar, float* vsum, int Ntot, hipStream_t s, NodeRange r) {
    const int n0 = r.n0, n1 = r.n1 < 0 ? Ntot : r.n1;
    const int N = n1 - n0;  // nodes of this launch (tile geometry is chosen for them)
    if (N <= 0) return ARREAU_OK;
    const int C = m->C, H = m->H;
    if (!(C == 128 && H == 512)) {
        arreau_set_error("mlp kernel (fp16x3, 16x16x32): unsupported (hidden_dim, widening_factor)");
        return ARREAU_EINVAL;
    }
    constexpr int NW = 4;
    const size_t layer_u32x4 = (size_t)2 * H * C * 2 * 2 / 16;  // bytes of W1 + W2 as 2 fp16 planes, in 16-byte units
    // Tile geometry: one wave = 2 nodes (32 rows, each weight fragment read from LDS serves six MFMAs) or 1 node (16 rows:
    // twice the tiles, a little more than half the time each -- 0.6 measured) -- bit-identical rows either way.  The launch
    // takes whole rounds of tiles over the chip's wave slots (2 workgroups x 4 waves per CU), so pick the geometry whose
    // rounds cost less: 16-row tiles while a batch leaves slots idle (1 x 8: -20 % step time) or has an expensive tail
    // (256 x 20: 2.5 rounds of 0.6 against 1.25 rounds -> 2 of 1.0: 1.70 vs 1.73 ms per step), 32-row tiles for large
    // batches (1024 x 20: 5 full rounds).  ARREAU_MLP_NB = 1 / 2 forces a geometry (tests).
    // Small launches: one node per workgroup, the layer's work dealt to eight waves (bit-identical; see below).  Measured on
    // MI355X: 10.7 us per layer at 8 nodes against 22.5 us; the forms cross where the node-per-workgroup form needs more
    // than about two rounds of the chip.  ARREAU_MLP_SPLIT = 0 / 1 forces a form (tests).
    static const int split_env = [] { const char* e = getenv("ARREAU_MLP_SPLIT"); return e ? atoi(e) : -1; }();
    // (unsliced launches only, like the edge kernel's small-launch form: edge_f16.hip)
    const bool whole_batch = n0 == 0 && n1 == Ntot && r.wg_cap == 0;
    if (split_env >= 0 ? split_env != 0 : (whole_batch && N <= ARREAU_MLP_SPLIT_MAX_NODES))
        return arreau_launch_mlp_f16x3_m16_split(m, layer, x_conv, x_in, x_out, xbar, vsum, Ntot, s, r);
    static const int nb_env = [] { const char* e = getenv("ARREAU_MLP_NB"); return e ? atoi(e) : 0; }();
    static const int wave_slots = [] {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            return 8 * (int)prop.multiProcessorCount;
        return 2048;
    }();
    const long long rounds1 = ((long long)N + wave_slots - 1) / wave_slots;
    const long long rounds2 = (((long long)N + 1) / 2 + wave_slots - 1) / wave_slots;
    const int nb = nb_env == 1 || nb_env == 2 ? nb_env : (6 * rounds1 < 10 * rounds2 ? 1 : 2);
    const long long tiles = ((long long)N + nb - 1) / nb;
    const dim3 grid((unsigned)((tiles + NW - 1) / NW)), block(64 * NW);
    const float* lnw = m->ln_w + (size_t)layer * C;
    const float* lnb = m->ln_b + (size_t)layer * C;
    const u32x4* stream = reinterpret_cast<const u32x4*>(m->mlp_f16m) + (size_t)layer * layer_u32x4;
    // ring depth: 3 slots; ARREAU_MLP_SLOTS=4 selects the four-slot ring (measured on MI355X at 256 x 20: 1.77 vs 1.77 ms
    // per step, no gain -- the L2 -> LDS latency is already covered by one chunk of matrix work -- so the smaller one stays)
    static const int slots = [] { const char* e = getenv("ARREAU_MLP_SLOTS"); return e && atoi(e) == 4 ? 4 : 3; }();
    auto launch = [&](auto kernel) {
        ARREAU_LAUNCH(kernel, grid, block, 0, s, x_conv, x_in, x_out, lnw, lnb, stream, m->mb1 + (size_t)layer * H,
                           m->mb2 + (size_t)layer * C, m->ls + (size_t)layer * C, m->ro_wv + (size_t)layer * C,
                           m->ro_bv_host[layer], n0, n1, layer == 0 ? 1 : 0, xbar + (size_t)layer * Ntot * C, vsum);
    };
    if (nb == 1 && slots == 4) launch(mlp_kernel_f16x3_m16<128, 512, NW, 1, 4>);
    else if (nb == 1) launch(mlp_kernel_f16x3_m16<128, 512, NW, 1, 3>);
    else if (slots == 4) launch(mlp_kernel_f16x3_m16<128, 512, NW, 2, 4>);
    else launch(mlp_kernel_f16x3_m16<128, 512, NW, 2, 3>);
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}
