// K4b, split-precision variant: the ConvNext block of one layer (convnext.py:25-32) + read-out partials
// (ponita.py:105-117) with every fp32 product evaluated as six bf16 MFMA products (bf16x6.h).
//
// Workgroup = 4 waves = four 32-row tiles (8 nodes x 16 orientations), one wave per SIMD.  Every wave walks
// the whole hidden dimension quarter by quarter,
//     hid  = GELU(W1[quarter] . xn + b1[quarter])      (xn = LayerNorm of the conv output, split in registers)
//     out += W2[:, quarter] . hid
// and the four waves share each 48 KiB weight chunk (two output tiles) through LDS: the next chunk is
// fetched into registers at the top of a chunk and written to the idle LDS buffer in the middle of its
// MFMA stream, one barrier per chunk.  No cross-wave reduction; the epilogue (bias, layer scale, residual,
// orientation mean, vector read-out) stays in registers.
#include <stdlib.h>

#include "bf16x6.h"
#include "internal.h"

#define MLP_FRAGS 48  // fragments per chunk: 2 output tiles x (4 in-tiles x 2 k-steps x 3 planes)

template <int C, int H>
__global__ __launch_bounds__(256, 1) void mlp_kernel_bf16x6(
    const float* __restrict__ x_conv,    // [N][16][C]  conv output (pre-LayerNorm)
    const float* __restrict__ x_in, float* __restrict__ x_out,
    const float* __restrict__ ln_w, const float* __restrict__ ln_b,
    const u32x4* __restrict__ stream,    // this layer: [4 quarters][W1 quarter: 4 tiles | W2 quarter: 4 tiles], 24 frags per tile
    const float* __restrict__ mb1, const float* __restrict__ mb2, const float* __restrict__ ls,
    const float* __restrict__ ro_wT,     // [C][S+4] this layer
    const float* __restrict__ ro_b,      // [S+4]
    int S, int N, int first_layer,
    float* __restrict__ xbar,            // [N][C] this layer
    float* __restrict__ vsum, int dbg)   // [N][16]
{
    static_assert(C == 128 && H == 512, "chunking below assumes C = 128, H = 512");
    constexpr int TC = C / 32;
    constexpr int HQ = H / 4, THQ = HQ / 32;
    constexpr int TILE_FRAGS = TC * 6;  // 24
    static_assert(2 * TILE_FRAGS == MLP_FRAGS && MLP_FRAGS <= EB_MAX_FRAGS && THQ == 4, "chunk size");
    __shared__ u32x4 lds[2][MLP_FRAGS * 64];  // 2 x 48 KiB

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int h = lane >> 5, j = lane & 31;
    const long long tile = (long long)blockIdx.x * 4 + wave;
    const bool active = 2 * tile < N;  // wave-uniform; idle waves still stage weights and meet the barriers
    const long long n_ll = 2 * tile + (j >> 4);
    const bool valid = n_ll < N;
    const int n = valid ? (int)n_ll : N - 1;  // padding rows read a valid row and write nothing
    const int o = j & 15;

    u32x4 st[EB_STAGE];
    const u32x4* chunk = stream;
    stage_load<MLP_FRAGS>(st, chunk, wave, lane);

    // ---- load the row in B-operand layout, LayerNorm it (eps 1e-5, biased variance), split ----------------
    const size_t rowoff = ((size_t)n * 16 + o) * C + 4 * h;
    Planes xn[TC];
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
            xn[t] = split_tile(bx[t]);
        }
    }
    stage_store<MLP_FRAGS>(st, lds[0], wave, lane);
    __syncthreads();

    f32x16 acc_o[TC];
#pragma unroll
    for (int u = 0; u < TC; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[u][r] = 0.f;

    // 16 chunks per layer: per quarter 2 chunks of W1 rows (hidden tiles 0-1, 2-3) then 2 chunks of W2 columns
    // (output tiles 0-1, 2-3).  Buffer parity is static inside the quarter (4 chunks) and repeats.
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
        Planes hid[THQ];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {  // ---- hid tiles 2ch, 2ch+1 ----
            const int cur = ch & 1;
            chunk += (size_t)MLP_FRAGS * 64;
            if (!(dbg & 4)) stage_load<MLP_FRAGS>(st, chunk, wave, lane);  // always another chunk after a W1 chunk
            f32x16 a0 = arreau_bias_tile(mb1 + w * HQ, 2 * ch, h);
            f32x16 a1 = arreau_bias_tile(mb1 + w * HQ, 2 * ch + 1, h);
            if (active && !(dbg & 2)) mma_range<TC, 0, 2 * TC>(a0, lds[cur], xn, lane);
            if (!(dbg & 4)) stage_store<MLP_FRAGS>(st, lds[cur ^ 1], wave, lane);
            if (active) {
                if (!(dbg & 2)) mma_range<TC, 0, 2 * TC>(a1, lds[cur] + (size_t)TILE_FRAGS * 64, xn, lane);
                if (!(dbg & 1)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { a0[r] = gelu_fast(a0[r]); a1[r] = gelu_fast(a1[r]); }
                }
                hid[2 * ch] = split_tile(a0);
                hid[2 * ch + 1] = split_tile(a1);
            }
            __syncthreads();
        }
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {  // ---- out tiles 2ch, 2ch+1 += W2[:, quarter] . hid ----
            const int cur = ch & 1;
            chunk += (size_t)MLP_FRAGS * 64;
            const bool more = !(w == 3 && ch == 1);  // workgroup-uniform
            if (more && !(dbg & 4)) stage_load<MLP_FRAGS>(st, chunk, wave, lane);
            if (active && !(dbg & 2)) mma_range<THQ, 0, 2 * THQ>(acc_o[2 * ch], lds[cur], hid, lane);
            if (more && !(dbg & 4)) stage_store<MLP_FRAGS>(st, lds[cur ^ 1], wave, lane);
            if (active && !(dbg & 2)) mma_range<THQ, 0, 2 * THQ>(acc_o[2 * ch + 1], lds[cur] + (size_t)TILE_FRAGS * 64, hid, lane);
            __syncthreads();
        }
    }
    if (!active || (dbg & 8)) return;

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
            const f32x4 xi = *reinterpret_cast<const f32x4*>(x_in + rowoff + 32 * u + 8 * q);
            f32x4 xo;
#pragma unroll
            for (int m = 0; m < 4; ++m) xo[m] = (acc_o[u][4 * q + m] + b2v[m]) * lsv[m] + xi[m];
            if (valid) *reinterpret_cast<f32x4*>(x_out + rowoff + 32 * u + 8 * q) = xo;
#pragma unroll
            for (int m = 0; m < 4; ++m) vdot += xo[m] * ro_wT[(size_t)(c0 + m) * (S + 4) + S];
            f32x4 sum = xo;
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
#pragma unroll
                for (int m = 0; m < 4; ++m) sum[m] += __shfl_xor(sum[m], off, 64);
            }
            if (valid && o == 0) {
                const f32x4 mean = {sum[0] * inv16, sum[1] * inv16, sum[2] * inv16, sum[3] * inv16};
                *reinterpret_cast<f32x4*>(xbar + (size_t)n * C + c0) = mean;
            }
        }
    vdot += __shfl_xor(vdot, 32, 64);
    if (valid && h == 0) {
        const size_t g = (size_t)n * 16 + o;
        vsum[g] = (first_layer ? 0.0f : vsum[g]) + (vdot + ro_b[S]);
    }
}

int arreau_launch_mlp_bf16x6(const arreau_model* m, int layer, const float* x_conv, const float* x_in, float* x_out,
                             float* xbar, float* vsum, int N, hipStream_t s) {
    if (N == 0) return ARREAU_OK;
    const int C = m->C, H = m->H, S = m->S;
    if (!(C == 128 && H == 512)) {
        arreau_set_error("mlp kernel (bf16x6): unsupported (hidden_dim, widening_factor)");
        return ARREAU_EINVAL;
    }
    const size_t layer_u32x4 = (size_t)2 * H * C * 3 * 2 / 16;  // bytes of W1 + W2 as 3 bf16 planes, in 16-byte units
    static const int dbg = [] { const char* e = getenv("ARREAU_MLP_DBG"); return e ? atoi(e) : 0; }();
    ARREAU_LAUNCH((mlp_kernel_bf16x6<128, 512>), dim3((N + 7) / 8), dim3(256), 0, s, x_conv, x_in, x_out,
                       m->ln_w + (size_t)layer * C, m->ln_b + (size_t)layer * C,
                       reinterpret_cast<const u32x4*>(m->mlp_bf16) + (size_t)layer * layer_u32x4,
                       m->mb1 + (size_t)layer * H, m->mb2 + (size_t)layer * C, m->ls + (size_t)layer * C,
                       m->ro_wT + (size_t)layer * C * (S + 4), m->ro_b + (size_t)layer * (S + 4), S, N,
                       layer == 0 ? 1 : 0, xbar + (size_t)layer * N * C, vsum, dbg);
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}
