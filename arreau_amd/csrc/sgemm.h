// GEMMs of the shape-general / training network (train_net.hip) on the matrix pipe: exact fp32 (v_mfma_f32_32x32x2_f32,
// 157 TFLOP/s peak) and -- round 4 -- split precision on the 16-bit matrix instructions (sgemm_split_kernel below: fp16x3
// for products of O(1) operands, bf16x6 where an operand is a gradient of arbitrary magnitude).
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "f16x3.h"
#include "internal.h"

constexpr size_t ARREAU_SGEMM_PARTIAL_FLOATS = (size_t)64 * 512 * 512;  // split-K partial sums (64 slices of the largest weight)

namespace arreau_sgemm_detail {
// bf16x6 pieces (the scheme of bf16x6.h, which cannot be included next to f16x3.h: both define the kernels' GELU)
typedef __bf16 sg_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned sg_pack_hi16(float hi, float lo) {  // {hi[31:16], lo[31:16]}: two truncated bf16
    return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
__device__ __forceinline__ float sg_bf16_residual(float x) { return x - __uint_as_float(__float_as_uint(x) & 0xffff0000u); }  // exact
__device__ __forceinline__ f32x16 sg_mfma_bf16(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sg_bf16x8, a), __builtin_bit_cast(sg_bf16x8, b), c, 0, 0, 0);
}

// C[m,n] = alpha * sum_k A(m,k) B(k,n) + beta * C[m,n];  A(m,k) = A[m*as0 + k*as1], B(k,n) = B[k*bs0 + n*bs1].
// Exact fp32 on the matrix pipe (v_mfma_f32_32x32x2_f32): a (64 WM) x (64 WN) output tile per workgroup, four waves in a
// 2 x 2 arrangement, each holding WM x WN MFMA tiles of 32 x 32 (16 accumulator registers per tile); operands staged
// k-major in LDS (a lane reads A(m0 + lane % 32, k + lane / 32) and B(k + lane / 32, n0 + lane % 32): conflict-free
// rows).  Accumulator register r of lane (h, j) is C[m0 + (r & 3) + 8 (r >> 2) + 4 h][n0 + j] (internal.h), so stores
// are coalesced along n.
//   WM = WN = 2: 128 x 128 tiles, 32 MFMAs per wave and k-step of 16 -- the large products (edge rows x basis).
//   WM = WN = 1: 64 x 64 tiles, k-step of 32 -- products whose 128 x 128 tiling would leave most of the
//                chip idle (the node-level linears at 64 crystals: 8512 x 128 outputs = 67 tiles of 128 x 128, 266 of
//                64 x 64).
// gridDim.z > 1: split-K, partial sums to `partial[z][M][N]` (reduced in z order by splitk_reduce_kernel: deterministic).
// Operand staging: each thread fetches its 16-byte pieces of the next A and B tiles into registers -- along whichever
// dimension is contiguous in memory -- BEFORE the matrix work of the current tile, and writes them to LDS after it
// (register double buffering: the global latency hides behind the tile's MFMAs).  A second LDS buffer with one barrier
// per k-step was measured and is no faster (tools/exp/sgemm_bench.hip: 330 vs 304 us at 8512 x 512 x 2048): the kernel
// is bound by the matrix pipe of the busiest CU, i.e. by how evenly the tiles divide over the 256 CUs.
// VEC = 0: element-wise path for shapes that are not multiples of four (K = 3, M = 94 ...).
template <int VEC, int WM, int WN, int BK>
__global__ __launch_bounds__(256) void sgemm_kernel(int M, int N, int K, const float* __restrict__ A, long as0, long as1,
                                                    const float* __restrict__ B, long bs0, long bs1, float* __restrict__ C,
                                                    int ldc, float alpha, float beta, int kchunk,
                                                    float* __restrict__ partial, int splits /* k-slices per product */,
                                                    long a_bs, long b_bs, long c_bs /* batch strides: blockIdx.z = batch * splits + slice */) {
    constexpr int TM = 64 * WM, TN = 64 * WN;
    const int bi = (int)blockIdx.z / splits, zi = (int)blockIdx.z - bi * splits;
    A += (long)bi * a_bs;
    B += (long)bi * b_bs;
    C += (long)bi * c_bs;
    constexpr int KP = BK / 4;                              // 16-byte pieces per k-row of a tile
    constexpr int PA = WM * BK / 16, PB = WN * BK / 16;     // pieces per thread (VEC); elements per thread = 4 x that
    __shared__ __attribute__((aligned(16))) float As[BK][TM + 4], Bs[BK][TN + 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, j = lane & 31, wm = (wave >> 1) * (32 * WM), wn = (wave & 1) * (32 * WN);
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
    const int kbeg = zi * kchunk, kend = min(K, kbeg + kchunk);
    f32x16 acc[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const bool a_kmajor = as1 == 1, b_nmajor = bs1 == 1;  // which dimension is contiguous
    f32x4 ra[PA], rb[PB];
    float sa[4 * PA], sb[4 * PB];
    auto fetch = [&](int k0) {
        if (VEC) {
#pragma unroll
            for (int i = 0; i < PA; ++i) {  // TM x BK / 4 pieces of the A tile
                const int idx = tid + 256 * i;
                const int mm = a_kmajor ? idx / KP : (idx % (TM / 4)) * 4, kk = a_kmajor ? (idx % KP) * 4 : idx / (TM / 4);
                const int m = m0 + mm, k = k0 + kk;
                ra[i] = (m < M && k < kend) ? *reinterpret_cast<const f32x4*>(A + (long)m * as0 + (long)k * as1) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const int idx = tid + 256 * i;
                const int nn = b_nmajor ? (idx % (TN / 4)) * 4 : idx / KP, kk = b_nmajor ? idx / (TN / 4) : (idx % KP) * 4;
                const int n = n0 + nn, k = k0 + kk;
                rb[i] = (n < N && k < kend) ? *reinterpret_cast<const f32x4*>(B + (long)k * bs0 + (long)n * bs1) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4 * PA; ++i) {  // TM x BK elements of the A tile
                const int idx = tid + 256 * i;
                const int mm = a_kmajor ? idx / BK : idx % TM, kk = a_kmajor ? idx % BK : idx / TM;
                const int m = m0 + mm, k = k0 + kk;
                sa[i] = (m < M && k < kend) ? A[(long)m * as0 + (long)k * as1] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 4 * PB; ++i) {
                const int idx = tid + 256 * i;
                const int nn = b_nmajor ? idx % TN : idx / BK, kk = b_nmajor ? idx / TN : idx % BK;
                const int n = n0 + nn, k = k0 + kk;
                sb[i] = (n < N && k < kend) ? B[(long)k * bs0 + (long)n * bs1] : 0.f;
            }
        }
    };
    auto stage = [&]() {
        if (VEC) {
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                const int idx = tid + 256 * i;
                if (a_kmajor) {
                    const int mm = idx / KP, kk = (idx % KP) * 4;
#pragma unroll
                    for (int q = 0; q < 4; ++q) As[kk + q][mm] = ra[i][q];
                } else {
                    *reinterpret_cast<f32x4*>(&As[idx / (TM / 4)][(idx % (TM / 4)) * 4]) = ra[i];
                }
            }
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const int idx = tid + 256 * i;
                if (b_nmajor) {
                    *reinterpret_cast<f32x4*>(&Bs[idx / (TN / 4)][(idx % (TN / 4)) * 4]) = rb[i];
                } else {
                    const int nn = idx / KP, kk = (idx % KP) * 4;
#pragma unroll
                    for (int q = 0; q < 4; ++q) Bs[kk + q][nn] = rb[i][q];
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4 * PA; ++i) {
                const int idx = tid + 256 * i;
                if (a_kmajor) As[idx % BK][idx / BK] = sa[i]; else As[idx / TM][idx % TM] = sa[i];
            }
#pragma unroll
            for (int i = 0; i < 4 * PB; ++i) {
                const int idx = tid + 256 * i;
                if (b_nmajor) Bs[idx / TN][idx % TN] = sb[i]; else Bs[idx % BK][idx / BK] = sb[i];
            }
        }
    };
    fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        stage();
        __syncthreads();
        if (k0 + BK < kend) fetch(k0 + BK);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float av[WM], bv[WN];
#pragma unroll
            for (int a = 0; a < WM; ++a) av[a] = As[kk + h][wm + 32 * a + j];
#pragma unroll
            for (int b = 0; b < WN; ++b) bv[b] = Bs[kk + h][wn + 32 * b + j];
#pragma unroll
            for (int a = 0; a < WM; ++a)
#pragma unroll
                for (int b = 0; b < WN; ++b) acc[a][b] = arreau_mfma(av[a], bv[b], acc[a][b]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h, n = n0 + wn + 32 * b + j;
                if (m < M && n < N) {
                    if (splits > 1) partial[((size_t)blockIdx.z * M + m) * N + n] = acc[a][b][r];  // [batch][slice][M][N]
                    else C[(size_t)m * ldc + n] = alpha * acc[a][b][r] + (beta != 0.f ? beta * C[(size_t)m * ldc + n] : 0.f);
                }
            }
}
// Element-wise work fused into the split kernel's epilogue (round 4; never with split-K, whose partial sums pass through
// splitk_reduce_kernel).  All operands share C's shape and pitch; the arithmetic is the element-wise kernels' of train_net.hip
// (same expressions, so a fused and an unfused product agree to the bit):
//   1  bias + GELU:            v += vec[n];  C = v;  out = gelu(v) * (row ? row[m] : 1)        (ponita.py:65: basis MLPs; convnext.py: linear_1)
//   2  GELU backward:          C = v * gelu'(mat[m][n]) * (row ? row[m] : 1)                     (mat = the pre-activation)
//   3  bias, scale, residual:  v += vec[n];  C = v;  out = v * vec2[n] + mat[m][n]               (convnext.py: x + layer_scale * linear_2(...))
struct SgemmEpilogue {
    int kind = 0;
    // round 5: a short reduction (K <= 256) whose output traffic dominates may ask for the 64 x 64 tiles just to get its epilogue inside the
    // product (the 128 x 128 tiles take none: below) -- [64768, 128] x [256, 128]^T + bias + GELU: 30 + 30 us as product + pass
    bool prefer_small = false;
    const float* vec = nullptr;
    const float* vec2 = nullptr;
    const float* row = nullptr;
    const float* mat = nullptr;
    float* out = nullptr;
};
// GELU and its derivative, branch-free (round 5; the element-wise kernels of train_net.hip use the same two functions, so a fused
// and an unfused product still agree to the bit).  Forward: the sampling kernels' own form (f16x3.h gelu_fast2: 2.8e-7 absolute).
// Derivative Phi(x) + x phi(x): Phi from the branch-free erf of internal.h (< 1 ulp + the 2e-7 of v_exp_f32), phi from one
// v_exp_f32.  libm's erff / expf -- branches on |x|, about fifty instructions per element -- made a GELU epilogue cost more than
// the launch it replaced on the 128 x 128 tiles.
__device__ __forceinline__ float sg_gelu_exact(float x) { return gelu_fast(x); }
__device__ __forceinline__ float sg_gelu_grad(float x) {
    return 0.5f * (1.0f + arreau_erf(x * 0.70710678118654752440f)) +
           x * 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);
}
// Split-precision form of the kernel above (round 4; BASELINE configs[4] asks for the training step on matrix-rate
// arithmetic).  Same tiling, same operand addressing, same split-K protocol; what changes is the staging -- every fp32
// operand element is split into 16-bit planes ON ITS WAY INTO LDS (once per element and tile, the cost amortised over the
// 64 / 128 outputs it feeds) -- and the inner loop, which runs v_mfma_f32_32x32x16_{f16,bf16} on 16-byte fragments:
//   MODE 1, fp16x3 (f16x3.h):  a = a1 + a2 / 2^11, two fp16 planes, a b ~= a1 b1 + (a1 b2 + a2 b1) / 2^11: three products,
//           `main` and `cross` accumulators folded at the end.  Operand range: |v| < 65504, full 22-bit precision down to
//           6e-5 -- forward products (activations x weights), the arithmetic the sampling kernels use.
//   MODE 2, bf16x6 (bf16x6.h): a = a1 + a2 + a3 by exact 8-bit truncation, six products into one accumulator, the full fp32
//           exponent range -- every product with a GRADIENT operand (magnitudes from 1e-9, e.g. behind layer_scale = 1e-6, to
//           O(1): an fp16 plane would go subnormal there).
// LDS image: plane-major [P][rows][BK + 8] halves, k contiguous per row (a lane's fragment = 8 consecutive k of its row:
// one ds_read_b128; the 80-byte row stride makes those reads conflict-free).
// Staging by operand layout (AK / BKC: the operand is contiguous along k in memory):
//   k-contiguous (X W^T: activations, weights as stored): 16-byte pieces along k, written as 8-byte pieces per plane;
//   row-contiguous (dY W: the weight read across its rows; dY^T X: both operands of a weight gradient): a LANE owns a
//           ROW of the tile and the wave a run of k -- KPT = 8 (64-row tiles) or 16 (128-row tiles) dword loads per thread,
//           each wave-instruction 256 contiguous bytes of one k-row -- so the planes of its KPT consecutive k leave as one or
//           two ds_write_b128 per plane (rows 80 bytes apart: conflict-free) and no transposing store exists.  The first
//           form of this kernel wrote such operands as 2-byte stores and was slower than the exact kernel (261 -> 325 us).
// Two workgroups per CU (launch bound: 256 registers): the kernel keeps ONE LDS buffer and two barriers per k-step, so a
// workgroup alone on its CU never overlaps staging with matrix work -- the fp16x3 128 x 128 form took 268 registers, ran
// one wave per SIMD and reached 110 TFLOP/s (effective) on [64768, 256] x [640, 256]^T, 15 % of its matrix time.
template <int MODE, int WM, int WN, bool AK, bool BKC, int PD, int BK>
__global__ __launch_bounds__(256, 2) void sgemm_split_kernel(int M, int N, int K, const float* __restrict__ A, long as0, long as1,
                                                             const float* __restrict__ B, long bs0, long bs1, float* __restrict__ C,
                                                             int ldc, float alpha, float beta, int kchunk, float* __restrict__ partial,
                                                             int splits, long a_bs, long b_bs, long c_bs, SgemmEpilogue epi, int swz, int gm, int gn, int zb) {
    constexpr int TM = 64 * WM, TN = 64 * WN, P = MODE == 1 ? 2 : 3, LDK = BK + 8;  // (row pitch 80 / 144 bytes: conflict-free b128 reads)
    // Tile of this workgroup.  swz != 0: a 1-D grid dealt so that the workgroups which read the same operand panel run on ONE XCD, one
    // after the other (consecutive workgroups go to the 8 XCDs in turn, each with its own L2): swz = 1 -- all column tiles of a row
    // tile (the panel of A: activations / gradients, 66-166 MB at 64 crystals, which five column tiles otherwise pull into five L2s);
    // swz = 2 -- all tiles of a k-slice of a split-K product (both slice panels).  Round 4, PMC: the training step moved 6.3 GB per
    // step through the L2s, 4 GB of it in these products -- [64768, 256] x [640, 256]^T 500 MB against 232 MB of operands + output.
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (swz != 0) {
        const int xcd = (int)blockIdx.x & 7, sl = (int)blockIdx.x >> 3;
        if (swz == 1) {
            by = (sl / gn) * 8 + xcd; bx = sl % gn; bz = 0;
            if (by >= gm) return;  // (workgroup-uniform, before any barrier)
        } else {
            const int per = gm * gn, tl = sl % per;
            bz = (sl / per) * 8 + xcd; by = tl / gn; bx = tl % gn;
            if (bz >= zb) return;
        }
    }
    const int bi = bz / splits, zi = bz - bi * splits;
    A += (long)bi * a_bs;
    B += (long)bi * b_bs;
    C += (long)bi * c_bs;
    constexpr int KP = BK / 4;
    constexpr int PA = WM * BK / 16, PB = WN * BK / 16;   // k-contiguous: 16-byte pieces per thread
    constexpr int EA = 4 * PA, EB = 4 * PB;               // row-contiguous: elements (consecutive k of one row) per thread
    __shared__ __attribute__((aligned(16))) unsigned short As[P][TM][LDK], Bs[P][TN][LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31, wm = (wave >> 1) * (32 * WM), wn = (wave & 1) * (32 * WN);
    const int m0 = by * TM, n0 = bx * TN;
    const int kbeg = zi * kchunk, kend = min(K, kbeg + kchunk);
    constexpr int NACC = MODE == 1 ? 2 : 1;  // fp16x3: main + cross
    f32x16 acc[NACC][WM][WN];
#pragma unroll
    for (int q = 0; q < NACC; ++q)
#pragma unroll
        for (int a = 0; a < WM; ++a)
#pragma unroll
            for (int b = 0; b < WN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][a][b][r] = 0.f;
    // PD register sets: the tile PD k-steps ahead is in flight while this one is multiplied (PD = 1 left the global latency
    // exposed -- a timing-only build without the loads ran 204 -> 128 us on [68096, 640] x [640, 256] and 244 -> 123 us on
    // the matching weight gradient, while one without the matrix instructions still took 132 us)
    f32x4 ra[PD][AK ? PA : 1], rb[PD][BKC ? PB : 1];
    float ea[PD][AK ? 1 : EA], eb[PD][BKC ? 1 : EB];
    // row-contiguous operands: this thread's row of the tile and its first k inside the tile
    const int a_row = 64 * (wave % WM) + lane, a_k = (wave / WM) * EA;
    const int b_row = 64 * (wave % WN) + lane, b_k = (wave / WN) * EB;
    // Every load is UNCONDITIONAL (round 4: as predicated loads each sat in its own basic block behind two branches, and hipcc,
    // unable to count them, waited with vmcnt(0) at the top of the loop -- for the tile just requested as well).  Rows beyond M / N
    // are clamped to the last row: their products land in accumulator rows that are never stored.  Beyond K (the last k-step of a
    // ragged K; k-slices are multiples of the k-step) the A operand is zeroed by a select -- in stage(), behind the matrix work:
    // written next to the load, hipcc waits for the load right there -- and B re-reads its last k, an element that takes part
    // in the same outputs anyway (so a non-finite value there reaches nothing it would not reach already).
    // Addresses: a uniform tile base (scalar registers, advanced by the k-step) + per-thread 32-bit element offsets computed ONCE
    // (row clamp folded in).  Computed per load they were ~130 vector instructions per k-step -- 64-bit multiplies and clamps -- and a
    // timing-only build without the loads ran 195 -> 134 us on [68096, 640] x [640, 256] although the loads themselves were prefetched.
    const float* a_tile = A + (AK ? (long)m0 * as0 : (long)m0);
    const float* b_tile = B + (BKC ? (long)n0 * bs1 : (long)n0);
    unsigned a_off[AK ? PA : 1], b_off[BKC ? PB : 1];
    if constexpr (AK) {
#pragma unroll
        for (int i = 0; i < PA; ++i) a_off[i] = (unsigned)(min(tid / KP + (256 / KP) * i, M - 1 - m0) * (int)as0 + (tid % KP) * 4);
    } else {
        a_off[0] = (unsigned)min(a_row, M - 1 - m0);
    }
    if constexpr (BKC) {
#pragma unroll
        for (int i = 0; i < PB; ++i) b_off[i] = (unsigned)(min(tid / KP + (256 / KP) * i, N - 1 - n0) * (int)bs1 + (tid % KP) * 4);
    } else {
        b_off[0] = (unsigned)min(b_row, N - 1 - n0);
    }
    auto fetch = [&](int k0, int u) {
        const bool inside = k0 + BK <= K;  // (uniform) the whole k-step lies inside the matrix: no clamp
        if constexpr (AK) {
            if (inside) {
                const float* src = a_tile + k0;
#pragma unroll
                for (int i = 0; i < PA; ++i) ra[u][i] = *reinterpret_cast<const f32x4*>(src + a_off[i]);
            } else {
                const int shift = min(k0 + (tid % KP) * 4, K - 4) - (tid % KP) * 4;
#pragma unroll
                for (int i = 0; i < PA; ++i) ra[u][i] = *reinterpret_cast<const f32x4*>(a_tile + shift + a_off[i]);
            }
        } else {
            if (inside) {
                const float* src = a_tile + (long)(k0 + a_k) * as1;  // (uniform per wave: a scalar pointer stepped by the pitch)
#pragma unroll
                for (int i = 0; i < EA; ++i, src += as1) ea[u][i] = src[a_off[0]];
            } else {
#pragma unroll
                for (int i = 0; i < EA; ++i) ea[u][i] = (a_tile + (long)min(k0 + a_k + i, K - 1) * as1)[a_off[0]];
            }
        }
        if constexpr (BKC) {
            if (inside) {
                const float* src = b_tile + k0;
#pragma unroll
                for (int i = 0; i < PB; ++i) rb[u][i] = *reinterpret_cast<const f32x4*>(src + b_off[i]);
            } else {
                const int shift = min(k0 + (tid % KP) * 4, K - 4) - (tid % KP) * 4;
#pragma unroll
                for (int i = 0; i < PB; ++i) rb[u][i] = *reinterpret_cast<const f32x4*>(b_tile + shift + b_off[i]);
            }
        } else {
            if (inside) {
                const float* src = b_tile + (long)(k0 + b_k) * bs0;
#pragma unroll
                for (int i = 0; i < EB; ++i, src += bs0) eb[u][i] = src[b_off[0]];
            } else {
#pragma unroll
                for (int i = 0; i < EB; ++i) eb[u][i] = (b_tile + (long)min(k0 + b_k + i, K - 1) * bs0)[b_off[0]];
            }
        }
    };
    // planes of two consecutive values as one dword per plane (element 0 in the low half)
    auto split2 = [&](float v0, float v1, unsigned (&w)[P]) {
        if constexpr (MODE == 1) {
            split_pair2<false>(f32x2{v0, v1}, w[0], w[1]);
        } else {
            w[0] = sg_pack_hi16(v1, v0);
            const float r0 = sg_bf16_residual(v0), r1 = sg_bf16_residual(v1);
            w[1] = sg_pack_hi16(r1, r0);
            w[2] = sg_pack_hi16(sg_bf16_residual(r1), sg_bf16_residual(r0));
        }
    };
    typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
    auto stage = [&](int u, int k0) {
        if constexpr (AK) {
            const bool live = k0 + (tid % KP) * 4 < kend;
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                const int idx = tid + 256 * i;
                const int row = idx / KP, kk = (idx % KP) * 4;
                asm volatile("" : "+v"(ra[u][i]));  // (the select stays here)
                if (!live) ra[u][i] = f32x4{0.f, 0.f, 0.f, 0.f};
                unsigned w0[P], w1[P];
                split2(ra[u][i][0], ra[u][i][1], w0);
                split2(ra[u][i][2], ra[u][i][3], w1);
#pragma unroll
                for (int p = 0; p < P; ++p) *reinterpret_cast<u32x2v*>(&As[p][row][kk]) = u32x2v{w0[p], w1[p]};
            }
        } else {
#pragma unroll
            for (int q = 0; q < EA / 8; ++q) {  // eight consecutive k = 16 bytes per plane
                unsigned w[4][P];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    asm volatile("" : "+v"(ea[u][8 * q + e]));
                    if (k0 + a_k + 8 * q + e >= kend) ea[u][8 * q + e] = 0.f;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) split2(ea[u][8 * q + 2 * e], ea[u][8 * q + 2 * e + 1], w[e]);
#pragma unroll
                for (int p = 0; p < P; ++p) *reinterpret_cast<u32x4*>(&As[p][a_row][a_k + 8 * q]) = u32x4{w[0][p], w[1][p], w[2][p], w[3][p]};
            }
        }
        if constexpr (BKC) {
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const int idx = tid + 256 * i;
                const int row = idx / KP, kk = (idx % KP) * 4;
                unsigned w0[P], w1[P];
                split2(rb[u][i][0], rb[u][i][1], w0);
                split2(rb[u][i][2], rb[u][i][3], w1);
#pragma unroll
                for (int p = 0; p < P; ++p) *reinterpret_cast<u32x2v*>(&Bs[p][row][kk]) = u32x2v{w0[p], w1[p]};
            }
        } else {
#pragma unroll
            for (int q = 0; q < EB / 8; ++q) {
                unsigned w[4][P];
#pragma unroll
                for (int e = 0; e < 4; ++e) split2(eb[u][8 * q + 2 * e], eb[u][8 * q + 2 * e + 1], w[e]);
#pragma unroll
                for (int p = 0; p < P; ++p) *reinterpret_cast<u32x4*>(&Bs[p][b_row][b_k + 8 * q]) = u32x4{w[0][p], w[1][p], w[2][p], w[3][p]};
            }
        }
    };
#pragma unroll
    for (int u = 0; u < PD; ++u) fetch(kbeg + u * BK, u);  // (beyond kend: zeros, no access)
    for (int kb = kbeg; kb < kend; kb += PD * BK) {
#pragma unroll
        for (int u = 0; u < PD; ++u) {
            const int k0 = kb + u * BK;
            if (u > 0 && k0 >= kend) break;  // (workgroup-uniform)
            stage(u, k0);
            __syncthreads();
            if (k0 + PD * BK < kend) fetch(k0 + PD * BK, u);
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                u32x4 fa[WM][P], fb[WN][P];
#pragma unroll
                for (int a = 0; a < WM; ++a)
#pragma unroll
                    for (int p = 0; p < P; ++p) fa[a][p] = *reinterpret_cast<const u32x4*>(&As[p][wm + 32 * a + j][16 * ks + 8 * h]);
#pragma unroll
                for (int b = 0; b < WN; ++b)
#pragma unroll
                    for (int p = 0; p < P; ++p) fb[b][p] = *reinterpret_cast<const u32x4*>(&Bs[p][wn + 32 * b + j][16 * ks + 8 * h]);
#pragma unroll
                for (int a = 0; a < WM; ++a)
#pragma unroll
                    for (int b = 0; b < WN; ++b) {
                        if constexpr (MODE == 1) {
                            acc[0][a][b] = mfma_f16(fa[a][0], fb[b][0], acc[0][a][b]);
                            acc[1][a][b] = mfma_f16(fa[a][0], fb[b][1], acc[1][a][b]);
                            acc[1][a][b] = mfma_f16(fa[a][1], fb[b][0], acc[1][a][b]);
                        } else {  // small terms first (bf16x6.h)
                            acc[0][a][b] = sg_mfma_bf16(fa[a][2], fb[b][0], acc[0][a][b]);
                            acc[0][a][b] = sg_mfma_bf16(fa[a][1], fb[b][1], acc[0][a][b]);
                            acc[0][a][b] = sg_mfma_bf16(fa[a][0], fb[b][2], acc[0][a][b]);
                            acc[0][a][b] = sg_mfma_bf16(fa[a][1], fb[b][0], acc[0][a][b]);
                            acc[0][a][b] = sg_mfma_bf16(fa[a][0], fb[b][1], acc[0][a][b]);
                            acc[0][a][b] = sg_mfma_bf16(fa[a][0], fb[b][0], acc[0][a][b]);
                        }
                    }
            }
            __syncthreads();
        }
    }
    // epilogue: the split-K / beta == 0 forms never read C (a per-element `beta != 0 ? C : 0` made every store a branch + load block)
    float* outp = splits > 1 ? partial + (size_t)bz * M * N : C;
    const int ldo = splits > 1 ? N : ldc;
    const float oa = splits > 1 ? 1.f : alpha;
    auto store_tiles = [&](auto read_c, auto fused) {
#pragma unroll
        for (int a = 0; a < WM; ++a)
#pragma unroll
            for (int b = 0; b < WN; ++b) {
                const int n = n0 + wn + 32 * b + j;
                if (n >= N) continue;
                const int mb = m0 + wm + 32 * a + 4 * h;  // register r holds row mb + (r & 3) + 8 (r >> 2)
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    v[r] = acc[0][a][b][r];
                    if constexpr (MODE == 1) v[r] = fmaf(acc[1][a][b][r], F16X3_INV_SCALE, v[r]);
                    v[r] *= oa;
                }
                if constexpr (decltype(fused)::value) {
                    // the tile's operands first (16 independent loads), then the arithmetic, then the stores: written load - compute -
                    // store per element, every load waited behind the previous store (the pointers may alias for all hipcc knows)
                    float mv[16], rv[16];
                    const float cv = epi.kind == 2 ? 0.f : epi.vec[n], c2 = epi.kind == 3 ? epi.vec2[n] : 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = min(mb + (r & 3) + 8 * (r >> 2), M - 1);
                        mv[r] = epi.kind == 1 ? 0.f : epi.mat[(size_t)m * ldo + n];
                        rv[r] = epi.row ? epi.row[m] : 1.0f;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (epi.kind == 1) { v[r] += cv; mv[r] = sg_gelu_exact(v[r]) * rv[r]; }
                        else if (epi.kind == 2) v[r] = v[r] * sg_gelu_grad(mv[r]) * rv[r];
                        else { v[r] += cv; mv[r] = v[r] * c2 + mv[r]; }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = mb + (r & 3) + 8 * (r >> 2);
                        if (m < M) {
                            outp[(size_t)m * ldo + n] = v[r];
                            if (epi.kind != 2) epi.out[(size_t)m * ldo + n] = mv[r];
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = mb + (r & 3) + 8 * (r >> 2);
                        if (m < M) {
                            float* o = outp + (size_t)m * ldo + n;
                            if constexpr (decltype(read_c)::value) *o = fmaf(beta, *o, v[r]);
                            else *o = v[r];
                        }
                    }
                }
            }
    };
    if (splits == 1 && beta != 0.f) store_tiles(std::true_type{}, std::false_type{});
    else if (splits == 1 && epi.kind != 0) store_tiles(std::false_type{}, std::true_type{});
    else store_tiles(std::false_type{}, std::false_type{});
}
// Sum of the k-slices' partial tiles.  A block covers 256 / ZP output elements; ZP threads share an element, each adding every ZP-th
// slice in ascending order, and the ZP sums are added in ascending order by the first of them: a fixed association, so the result is
// reproducible bit for bit (the order differs from a single running sum; nothing depends on that).  With one thread per element a product
// of 256 slices over 12 K outputs was 48 workgroups of 256 dependent loads each (66 us for 12 MB).
template <int ZP>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, int Z, int M, int N, float* __restrict__ C, int ldc,
                                                            float alpha, float beta, long c_bs /* blockIdx.y = product of the batch */) {
    constexpr int EPB = 256 / ZP;
    __shared__ float sh[ZP][EPB];
    const int el = threadIdx.x % EPB, zl = threadIdx.x / EPB;
    const long i = (long)blockIdx.x * EPB + el;
    const long MN = (long)M * N;
    partial += (size_t)blockIdx.y * Z * MN;
    C += (long)blockIdx.y * c_bs;
    float s = 0.f;
    if (i < MN)
        for (int z = zl; z < Z; z += ZP) s += partial[(size_t)z * MN + i];
    if constexpr (ZP > 1) {
        sh[zl][el] = s;
        __syncthreads();
        if (zl != 0) return;
#pragma unroll
        for (int q = 1; q < ZP; ++q) s += sh[q][el];
    }
    if (i >= MN) return;
    const int m = (int)(i / N), n = (int)(i % N);
    C[(size_t)m * ldc + n] = alpha * s + (beta != 0.f ? beta * C[(size_t)m * ldc + n] : 0.f);
}
// Round 5: the k-slice sums of SEVERAL products in one launch -- the weight gradients of a backward pass, which nothing waits for: each
// product keeps its partial tiles in its own piece of a scratch region and leaves a descriptor; one launch at the end adds them all
// (same threads-per-element choice and the same association per product as splitk_reduce_kernel<ZP>: bit-identical results).
struct SgemmReduceDesc {
    const float* partial;
    float* C;
    int Z, M, N, ldc;
    float alpha, beta;
    int zp;               // threads per output element (1, 4 or 8)
};
constexpr int SGEMM_DEFER_MAX = 40;
struct SgemmReduceList {
    SgemmReduceDesc d[SGEMM_DEFER_MAX];
    unsigned block0[SGEMM_DEFER_MAX];  // first workgroup of each product in the launch (ascending)
    int n;
};
__global__ __launch_bounds__(256) void splitk_reduce_multi_kernel(SgemmReduceList list) {
    __shared__ float sh[256];
    // this workgroup's product = the number of products that start at or before it: the starts arrive as three wide scalar loads and
    // the count is forty scalar compares (a search loop over the descriptors, one dependent scalar load per step, cost more than the
    // sums themselves: 55 us for 125 MB in the first version)
    int q = 0;
#pragma unroll
    for (int i = 1; i < SGEMM_DEFER_MAX; ++i) q += (i < list.n && blockIdx.x >= list.block0[i]) ? 1 : 0;
    const SgemmReduceDesc d = list.d[q];
    const unsigned first = list.block0[q];
    const int ZP = d.zp, EPB = 256 / ZP;
    const int el = threadIdx.x % EPB, zl = threadIdx.x / EPB;
    const long i = (long)(blockIdx.x - first) * EPB + el;
    const long MN = (long)d.M * d.N;
    const float* __restrict__ partial = d.partial;
    float s = 0.f;
    if (i < MN) {
        // (four independent loads in flight, added in slice order: with a run-time stride hipcc leaves a one-load loop, every step the
        // full memory latency -- 32 us for 65 MB)
        int z = zl;
        for (; z + 3 * ZP < d.Z; z += 4 * ZP) {
            const float v0 = partial[(size_t)z * MN + i], v1 = partial[(size_t)(z + ZP) * MN + i];
            const float v2 = partial[(size_t)(z + 2 * ZP) * MN + i], v3 = partial[(size_t)(z + 3 * ZP) * MN + i];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; z < d.Z; z += ZP) s += partial[(size_t)z * MN + i];
    }
    if (ZP > 1) {
        sh[zl * EPB + el] = s;
        __syncthreads();
        if (zl != 0) return;
        for (int w = 1; w < ZP; ++w) s += sh[w * EPB + el];
    }
    if (i >= MN) return;
    const int m = (int)(i / d.N), n = (int)(i % d.N);
    d.C[(size_t)m * d.ldc + n] = d.alpha * s + (d.beta != 0.f ? d.beta * d.C[(size_t)m * d.ldc + n] : 0.f);
}
}  // namespace arreau_sgemm_detail

// Deferred k-slice sums (see splitk_reduce_multi_kernel): scratch region + descriptor list, owned by the caller.
struct arreau_sgemm_defer {
    float* scratch = nullptr;
    size_t cap = 0, used = 0;   // floats
    arreau_sgemm_detail::SgemmReduceList list{};
    unsigned blocks = 0;
};
inline void arreau_sgemm_defer_reset(arreau_sgemm_defer& d) { d.used = 0; d.list.n = 0; d.blocks = 0; }
inline int arreau_sgemm_flush(hipStream_t s, arreau_sgemm_defer& d) {
    if (d.list.n > 0) {
        hipLaunchKernelGGL(arreau_sgemm_detail::splitk_reduce_multi_kernel, dim3(d.blocks), dim3(256), 0, s, d.list);
        ARREAU_CHECK_HIP(hipGetLastError());
    }
    arreau_sgemm_defer_reset(d);
    return ARREAU_OK;
}

// batch > 1: `batch` products of the same shape in one launch (operands a_bs / b_bs / c_bs floats apart; a stride of 0 shares an
// operand): the weight gradients of the L layers, which nothing in the backward pass waits for, leave the chip idle one by one
// (16 tiles each) and fill it together.
inline int arreau_sgemm(hipStream_t s, float* partial, int M, int N, int K, const float* A, long as0, long as1, const float* B,
                        long bs0, long bs1, float* C, int ldc, float alpha = 1.f, float beta = 0.f, int batch = 1, long a_bs = 0,
                        long b_bs = 0, long c_bs = 0, int mode = 0 /* 0: exact fp32; 1: fp16x3; 2: bf16x6 (sgemm_split_kernel) */,
                        const arreau_sgemm_detail::SgemmEpilogue* epi = nullptr, bool* fused = nullptr /* out: the epilogue ran inside the product */,
                        arreau_sgemm_defer* defer = nullptr /* a split-K product leaves its slice sum to arreau_sgemm_flush (C is NOT complete before that) */) {
    using namespace arreau_sgemm_detail;
    if (M == 0 || N == 0 || batch <= 0) return ARREAU_OK;
    // Tile size.  The kernel is bound by the matrix pipe of the busiest CU, so what matters is how evenly the tiles divide
    // over the 256 CUs: 268 tiles of 128 x 128 (8512 x 512 outputs) leave 12 CUs with two workgroups and everybody waits for
    // them (measured 304 us at K = 2048: half the rate of four times the rows); as 64 x 64 tiles the same product is 1064
    // workgroups, 4.2 per CU.  So 128 x 128 only when there are at least two of them per CU.  If even those are too few (weight gradients: a handful of
    // tiles under a reduction over all rows), split K as well: enough slices for about one workgroup per CU, at least two k-steps
    // each, more slices for very long reductions.
    const int tiles128 = ((M + 127) / 128) * ((N + 127) / 128) * batch;
    // A long reduction over a handful of output tiles (the weight gradients: K = all rows) keeps the 128 x 128 tile when K slices
    // of at least eight k-steps can fill the chip with them: a 64 x 64 tile reads each operand twice as often, and those launches
    // are bound by their loads (round 4: [640, 68096] x [68096, 256] without its loads 244 -> 123 us).
    const int z128 = min(64, K / 256);
    const bool long_k = mode != 0 && tiles128 < 512 && K >= 2048 && (long)tiles128 * z128 >= 256;
    const bool small = (tiles128 < 512 && !long_k) || (epi && epi->kind != 0 && epi->prefer_small && mode != 0 && K <= 256 && K >= 64 && batch == 1 && beta == 0.f);
    const int T = small ? 64 : 128;
    const int gm = (M + T - 1) / T, gn = (N + T - 1) / T;
    int Z = 1;
    const int tiles = gm * gn * batch;
    if (long_k) {
        Z = min(z128, (384 + tiles128 - 1) / tiles128);
    } else if (mode != 0 && tiles < 512 && K >= 1024) {
        // split kernel, few 64 x 64 tiles under a long reduction (the weight gradients of the small Linears): slices of at least four
        // k-steps until about a thousand workgroups are in flight -- at five workgroups per CU the chip holds 1,280, and [94, 128] from
        // 8,096 rows x 5 layers ran as 140 workgroups of 36 k-steps each (24.8 us; the partial sums stay within a few MB)
        Z = min(min(256, K / 128), (1024 + tiles - 1) / tiles);
    } else if (tiles < 128 && K >= 256) {
        const int fill = (256 + tiles - 1) / tiles;
        Z = min(64, min(max(fill, K / 1024), K / 128));
    } else if (batch > 1 && tiles < 512 && K >= 2048) {
        Z = min(8, K / 1024);  // a batch of long reductions over few tiles: still about four workgroups per CU
    }
    while (Z > 1 && (size_t)batch * Z * M * N > ARREAU_SGEMM_PARTIAL_FLOATS) --Z;
    // 16-byte operand fetches need the contiguous dimension and the leading dimension to be multiples of four floats
    const bool a_ok = (as1 == 1 && as0 % 4 == 0 && K % 4 == 0) || (as0 == 1 && as1 % 4 == 0 && M % 4 == 0);
    const bool b_ok = (bs1 == 1 && bs0 % 4 == 0 && N % 4 == 0) || (bs0 == 1 && bs1 % 4 == 0 && K % 4 == 0);
    const bool vec = a_ok && b_ok && ((size_t)A % 16 == 0) && ((size_t)B % 16 == 0) && a_bs % 4 == 0 && b_bs % 4 == 0;
    // The split kernel takes every 16-byte-fetchable layout with a reduction long enough to amortise the tile prologue (measured,
    // tools/exp/sgemm_bench.hip, profiles/r04_sgemm_bench.txt).  An operand that is not contiguous along k is fetched as dwords (no
    // alignment needed), so only its k-contiguous operands must pass the 16-byte test.
    const bool ak = as1 == 1, bk = bs0 == 1;
    const bool a_split_ok = ak ? (as0 % 4 == 0 && K % 4 == 0 && (size_t)A % 16 == 0 && a_bs % 4 == 0) : as0 == 1;
    const bool b_split_ok = bk ? (bs1 % 4 == 0 && K % 4 == 0 && (size_t)B % 16 == 0 && b_bs % 4 == 0) : bs1 == 1;
    const bool split = mode != 0 && a_split_ok && b_split_ok && K >= 64;
    const int BK = (small || split) ? 32 : 16;
    const int kchunk = ((K + Z - 1) / Z + BK - 1) / BK * BK;
    Z = (K + kchunk - 1) / kchunk;
    bool deferred = false;
    if (defer && Z > 1 && defer->scratch && defer->used + (size_t)batch * Z * M * N <= defer->cap && defer->list.n + batch <= SGEMM_DEFER_MAX) {
        partial = defer->scratch + defer->used;
        deferred = true;
    }
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(gn, gm, Z * batch), dim3(256), 0, s, M, N, K, A, as0, as1, B, bs0, bs1, C, ldc, alpha, beta, kchunk,
                           partial, Z, a_bs, b_bs, c_bs);
    };
    // the element-wise epilogue runs inside the product where the split kernel takes it whole (no k-slices, no accumulation into C)
    // ... and only on the 64 x 64 tiles: at 128 x 128 (two workgroups per CU, 64 outputs per thread) the erf / exp arithmetic of the GELU
    // epilogues is not hidden behind anything and the fused product ran slower than product + element-wise launch (round 4: 133 + 31 us
    // -> 256 us for the kernel-projection gradient with the GELU derivative)
    // (round 5: again with the branch-free GELU of this file, a quarter of libm's instructions -- 2.143 -> 2.180 ms per step with the
    // three 128 x 128 epilogues fused, alternating runs: 64 outputs per thread behind a single tile leave the epilogue nothing to hide in)
    const bool fuse = epi && epi->kind != 0 && split && small && Z == 1 && beta == 0.f && batch == 1;
    if (fused) *fused = fuse;
    SgemmEpilogue ep = fuse ? *epi : SgemmEpilogue{};
    // XCD-aware tile order (see the kernel): slices of a split-K product together; else, for one product, the column tiles of a row tile
    static const bool swz_on = [] { const char* e = getenv("ARREAU_SGEMM_SWIZZLE"); return !e || atoi(e) != 0; }();
    const int swz = !swz_on ? 0 : (Z > 1 ? 2 : (batch == 1 && gn > 1 ? 1 : 0));
    const int zb = Z * batch;
    const dim3 grid = swz == 1 ? dim3(8u * gn * ((gm + 7) / 8)) : swz == 2 ? dim3(8u * gm * gn * ((zb + 7) / 8)) : dim3(gn, gm, zb);
    auto launch_split = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, grid, dim3(256), 0, s, M, N, K, A, as0, as1, B, bs0, bs1, C, ldc, alpha, beta, kchunk,
                           partial, Z, a_bs, b_bs, c_bs, ep, swz, gm, gn, zb);
    };
    if (split) {
        auto pick = [&](auto mode_c, auto w_c) {
            constexpr int MD = decltype(mode_c)::value, W = decltype(w_c)::value;
            // two tiles in flight wherever the registers allow it at two workgroups per CU (fp16x3 at 128 x 128 holds 128 accumulators)
            constexpr int PD = (MD == 1 && W == 2) ? 1 : 2;
            // (k-steps of 64 for the 64 x 64 tiles -- half the barriers of the node-level products -- measured the same to 6 % slower)
            constexpr int BKS = 32;
            if (ak && bk) launch_split(sgemm_split_kernel<MD, W, W, true, true, PD, BKS>);
            else if (ak) launch_split(sgemm_split_kernel<MD, W, W, true, false, PD, BKS>);
            else if (bk) launch_split(sgemm_split_kernel<MD, W, W, false, true, PD, BKS>);
            else launch_split(sgemm_split_kernel<MD, W, W, false, false, PD, BKS>);
        };
        using std::integral_constant;
        if (small) { if (mode == 1) pick(integral_constant<int, 1>{}, integral_constant<int, 1>{}); else pick(integral_constant<int, 2>{}, integral_constant<int, 1>{}); }
        else { if (mode == 1) pick(integral_constant<int, 1>{}, integral_constant<int, 2>{}); else pick(integral_constant<int, 2>{}, integral_constant<int, 2>{}); }
    } else if (small) {
        if (vec) launch(sgemm_kernel<1, 1, 1, 32>); else launch(sgemm_kernel<0, 1, 1, 32>);
    } else {
        if (vec) launch(sgemm_kernel<1, 2, 2, 16>); else launch(sgemm_kernel<0, 2, 2, 16>);
    }
    ARREAU_CHECK_HIP(hipGetLastError());
    if (Z > 1 && deferred) {
        const long mn = (long)M * N;
        const int zp = Z <= 8 ? 1 : Z <= 32 ? 4 : 8;
        const unsigned nb = (unsigned)((mn + 256 / zp - 1) / (256 / zp));
        for (int b = 0; b < batch; ++b) {
            SgemmReduceDesc& d = defer->list.d[defer->list.n++];
            d.partial = partial + (size_t)b * Z * mn;
            d.C = C + (long)b * c_bs;
            d.Z = Z; d.M = M; d.N = N; d.ldc = ldc; d.alpha = alpha; d.beta = beta; d.zp = zp;
            defer->list.block0[defer->list.n - 1] = defer->blocks;
            defer->blocks += nb;
        }
        defer->used += ((size_t)batch * Z * mn + 63) & ~(size_t)63;
    } else if (Z > 1) {
        const long mn = (long)M * N;
        if (Z <= 8)
            hipLaunchKernelGGL(splitk_reduce_kernel<1>, dim3((unsigned)((mn + 255) / 256), batch), dim3(256), 0, s, partial, Z, M, N, C, ldc, alpha, beta, c_bs);
        else if (Z <= 32)
            hipLaunchKernelGGL(splitk_reduce_kernel<4>, dim3((unsigned)((mn + 63) / 64), batch), dim3(256), 0, s, partial, Z, M, N, C, ldc, alpha, beta, c_bs);
        else
            hipLaunchKernelGGL(splitk_reduce_kernel<8>, dim3((unsigned)((mn + 31) / 32), batch), dim3(256), 0, s, partial, Z, M, N, C, ldc, alpha, beta, c_bs);
        ARREAU_CHECK_HIP(hipGetLastError());
    }
    return ARREAU_OK;
}
