// K4b, fp16x3 split-precision variant (default): the ConvNext block of one layer (convnext.py:25-32) +
// read-out partials (ponita.py:105-117) with every fp32 product evaluated as three fp16 MFMA products
// (f16x3.h).
//
// One wave = one 32-row tile (2 nodes x 16 orientations).  Workgroup = NW waves sharing each 16 KiB weight
// chunk (one output tile) through a three-slot LDS ring filled by LDS-DMA (f16x3.h, edge_f16.hip "ring
// protocol": the one barrier per chunk sits in the middle of the chunk's MFMA stream, the copy of chunk q+2 starts
// there and is drained at the next one); two workgroups per CU, so one wave's VALU phases (LayerNorm, GELU,
// splits, epilogue) overlap the MFMA stream of the other workgroup's wave on the same SIMD.  Every wave walks the whole
// hidden dimension quarter by quarter,
//     hid  = GELU(W1[quarter] . xn + b1[quarter])      (xn = LayerNorm of the conv output, split in registers)
//     out += W2[:, quarter] . hid
// so there is no cross-wave reduction; the epilogue (bias, layer scale, residual, orientation mean, vector
// read-out) stays in registers.
#include <stdlib.h>
#include <utility>

#include "f16x3.h"
#include "internal.h"

// Sum over the 16 lanes of a DPP row (the 16 orientations of a node), every lane receiving the total: a butterfly
// of v_add_f32 with DPP operands (quad_perm xor 1, quad_perm xor 2, row_half_mirror, row_mirror) -- the same pairs
// in the same order as an xor-shuffle butterfly (bit-identical sums), without its 4 LDS permutes per value.
__device__ __forceinline__ float row16_sum(float v) {
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, false));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{});  // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>{});  // row_mirror
    return v;
}

template <int C, int H, int NW>
__global__ __launch_bounds__(64 * NW, 2) void mlp_kernel_f16x3(
    const float* __restrict__ x_conv,    // [N][16][C]  conv output (pre-LayerNorm)
    const float* __restrict__ x_in, float* __restrict__ x_out,
    const float* __restrict__ ln_w, const float* __restrict__ ln_b,
    const u32x4* __restrict__ stream,    // this layer: [4 quarters][W1 quarter: 4 tiles | W2 quarter: 4 tiles], 16 frags per tile
    const float* __restrict__ mb1, const float* __restrict__ mb2, const float* __restrict__ ls,
    const float* __restrict__ wv,        // [C] vector read-out weights of this layer (column S of read_out_layers)
    float bv, int N, int first_layer,
    float* __restrict__ xbar,            // [N][C] this layer
    float* __restrict__ vsum)            // [N][16]
{
    static_assert(C == 128 && H == 512, "chunking below assumes C = 128, H = 512");
    constexpr int TC = C / 32;
    constexpr int HQ = H / 4, THQ = HQ / 32;
    constexpr int NF = TC * 4;  // 16 fragments (16 KiB) per chunk = one output tile
    static_assert(THQ == TC, "W1 and W2 quarter chunks have the same size");
    __shared__ u32x4 lds[3][NF * 64];                              // 3 x 16 KiB ring of weight chunks
    __shared__ __attribute__((aligned(16))) float bias_s[H];      // mb1: no global loads while a DMA is in flight

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int h = lane >> 5, j = lane & 31;
    const long long tile = (long long)blockIdx.x * NW + wave;
    const bool active = 2 * tile < N;  // wave-uniform; idle waves still stage weights and meet the barriers
    const long long n_ll = 2 * tile + (j >> 4);
    const bool valid = n_ll < N;
    const int n = valid ? (int)n_ll : N - 1;  // padding rows read a valid row and write nothing
    const int o = j & 15;

    const u32x4* dma_src = stream;  // next chunk to copy
    dma_chunk<NF, NW>(dma_src, lds[0], wave, lane);
    dma_src += (size_t)NF * 64;
    dma_chunk<NF, NW>(dma_src, lds[1], wave, lane);
    dma_src += (size_t)NF * 64;
    for (int i = threadIdx.x; i < H; i += 64 * NW) bias_s[i] = mb1[i];

    // ---- load the row in B-operand layout, LayerNorm it (eps 1e-5, biased variance), split ----------------
    const size_t rowoff = ((size_t)n * 16 + o) * C + 4 * h;
    Planes2 xn[TC];
    {
        f32x16 bx[TC];
        const float* rowp = x_conv + rowoff;
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < TC; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(rowp + 32 * t + 8 * q);
                bx[t][4 * q] = v[0]; bx[t][4 * q + 1] = v[1]; bx[t][4 * q + 2] = v[2]; bx[t][4 * q + 3] = v[3];
                sum += (v[0] + v[1]) + (v[2] + v[3]);
            }
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * (1.0f / C);
        float sq = 0.f;
#pragma unroll
        for (int t = 0; t < TC; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d = bx[t][r] - mean;
                bx[t][r] = d;
                sq += d * d;
            }
        sq += __shfl_xor(sq, 32, 64);
        const float rstd = 1.0f / sqrtf(sq * (1.0f / C) + 1e-5f);
#pragma unroll
        for (int t = 0; t < TC; ++t) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(ln_w + 32 * t + 8 * q + 4 * h);
                const f32x4 be = *reinterpret_cast<const f32x4*>(ln_b + 32 * t + 8 * q + 4 * h);
#pragma unroll
                for (int m = 0; m < 4; ++m) bx[t][4 * q + m] = bx[t][4 * q + m] * rstd * g[m] + be[m];
            }
            xn[t] = split_tile2(bx[t]);
        }
    }
    dma_wait();
    __syncthreads();

    f32x16 acc_o[TC];
#pragma unroll
    for (int u = 0; u < TC; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[u][r] = 0.f;

    // 32 chunks per layer: per quarter 4 chunks of W1 rows (hidden tiles 0..3) then 4 chunks of W2 columns
    // (output tiles 0..3).  Chunk q sits in ring slot q % 3.
    int sl = 0;
    const unsigned lane16 = 16u * lane;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)&lds[0][0]) + 1024u * wave;
    auto sync = [&](bool more2) {  // SYNC_q: chunk q+1 complete for everybody, slot of chunk q-1 free -> copy chunk q+2
        dma_wait();
        __syncthreads();
        if (more2) dma_chunk_lean<NF, NW>(dma_src, lane16, wave, lds0 + (sl == 0 ? 2u : (unsigned)sl - 1u) * (NF * 1024u));
        dma_src += (size_t)NF * 64;
    };
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
        Planes2 hid[THQ];
#pragma unroll
        for (int u = 0; u < THQ; ++u) {  // ---- hid tile u = GELU(W1q[u] . xn + b1q[u]) ----
            f32x16 acc = arreau_bias_tile(bias_s + w * HQ, u, h), cross;
#pragma unroll
            for (int r = 0; r < 16; ++r) cross[r] = 0.f;
            if (active) mma_range2<TC, 0, TC>(acc, cross, lds[sl], xn, lane);
            sync(true);  // a W1 chunk is always followed by at least four more
            if (active) {
                mma_range2<TC, TC, 2 * TC>(acc, cross, lds[sl], xn, lane);
                hid[u] = gelu_split_tile2(acc, cross, 1.0f);
            }
            sl = sl == 2 ? 0 : sl + 1;
        }
#pragma unroll
        for (int u = 0; u < TC; ++u) {  // ---- out tile u += W2[:, quarter][u] . hid ----
            f32x16 cross;
#pragma unroll
            for (int r = 0; r < 16; ++r) cross[r] = 0.f;
            if (active) mma_range2<THQ, 0, THQ>(acc_o[u], cross, lds[sl], hid, lane);
            sync(!(w == 3 && u >= TC - 2));
            if (active) {
                mma_range2<THQ, THQ, 2 * THQ>(acc_o[u], cross, lds[sl], hid, lane);
                acc_o[u] = fold_cross(acc_o[u], cross);
            }
            sl = sl == 2 ? 0 : sl + 1;
        }
    }
    // (round 4, lint rule ldsdma-unwaited-exit: no LDS-DMA copy is left in flight when a wave ends -- the last copies of a ring
    // target a chunk nobody will read; the hardware's implicit wait at s_endpgm is not relied upon)
    dma_wait();
    if (!active) return;

    // ---- bias, layer scale, residual; write x_out; read-out partials (all in registers) ------------------
    float vdot = 0.f;
    const float inv16 = 1.0f / 16.0f;
#pragma unroll
    for (int u = 0; u < TC; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c0 = 32 * u + 8 * q + 4 * h;
            const f32x4 b2v = *reinterpret_cast<const f32x4*>(mb2 + c0);
            const f32x4 lsv = *reinterpret_cast<const f32x4*>(ls + c0);
            const f32x4 wvv = *reinterpret_cast<const f32x4*>(wv + c0);
            const f32x4 xi = *reinterpret_cast<const f32x4*>(x_in + rowoff + 32 * u + 8 * q);
            f32x4 xo;
#pragma unroll
            for (int m = 0; m < 4; ++m) xo[m] = (acc_o[u][4 * q + m] + b2v[m]) * lsv[m] + xi[m];
            if (valid) *reinterpret_cast<f32x4*>(x_out + rowoff + 32 * u + 8 * q) = xo;
#pragma unroll
            for (int m = 0; m < 4; ++m) vdot += xo[m] * wvv[m];
            f32x4 sum = xo;
#pragma unroll
            for (int m = 0; m < 4; ++m) sum[m] = row16_sum(sum[m]);
            if (valid && o == 0) {
                const f32x4 mean = {sum[0] * inv16, sum[1] * inv16, sum[2] * inv16, sum[3] * inv16};
                *reinterpret_cast<f32x4*>(xbar + (size_t)n * C + c0) = mean;
            }
        }
    vdot += __shfl_xor(vdot, 32, 64);
    if (valid && h == 0) {
        const size_t g = (size_t)n * 16 + o;
        vsum[g] = (first_layer ? 0.0f : vsum[g]) + (vdot + bv);
    }
}

int arreau_launch_mlp_f16x3(const arreau_model* m, int layer, const float* x_conv, const float* x_in, float* x_out,
                            float* xbar, float* vsum, int N, hipStream_t s) {
    if (N == 0) return ARREAU_OK;
    const int C = m->C, H = m->H;
    if (!(C == 128 && H == 512)) {
        arreau_set_error("mlp kernel (fp16x3): unsupported (hidden_dim, widening_factor)");
        return ARREAU_EINVAL;
    }
    constexpr int NW = 4;
    const size_t layer_u32x4 = (size_t)2 * H * C * 2 * 2 / 16;  // bytes of W1 + W2 as 2 fp16 planes, in 16-byte units
    const long long tiles = ((long long)N + 1) / 2;
    ARREAU_LAUNCH((mlp_kernel_f16x3<128, 512, NW>), dim3((unsigned)((tiles + NW - 1) / NW)), dim3(64 * NW), 0, s,
                       x_conv, x_in, x_out, m->ln_w + (size_t)layer * C, m->ln_b + (size_t)layer * C,
                       reinterpret_cast<const u32x4*>(m->mlp_f16) + (size_t)layer * layer_u32x4,
                       m->mb1 + (size_t)layer * H, m->mb2 + (size_t)layer * C, m->ls + (size_t)layer * C,
                       m->ro_wv + (size_t)layer * C, m->ro_bv_host[layer], N, layer == 0 ? 1 : 0,
                       xbar + (size_t)layer * N * C, vsum);
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}
