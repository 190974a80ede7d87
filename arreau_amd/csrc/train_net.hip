// Training-mode score network (BASELINE config 5): forward with saved activations and the backward pass, fp32.
//
// The sampling kernels (edge_f16.hip, node*.hip) keep nothing: activations live in registers and only K_l reaches HBM.
// A training step needs every layer's inputs again, so this file evaluates the same network in the textbook form --
// one dense product per Linear (sgemm_kernel below, exact fp32 on the matrix pipe, split-K for the weight gradients), small elementwise /
// gather / reduce kernels between them, every intermediate in HBM -- and walks it backwards.  At config-5 sizes (64
// crystals, about 540 atoms, 69 k (edge, orientation) rows per GPU) the whole step moves a few hundred MB; clarity
// and exact fp32 arithmetic matter more here than the last factor of two.  Measured on MI355X (profiles/r02i_*): 9.6 ms
// forward + backward for 64 crystals / 532 atoms (first version with FMA GEMMs and single-workgroup column sums: 38 ms).
//
// Reference: PonitaFiberBundle.forward (ponita/models/ponita.py:88-123), FiberBundleConv.forward / message
// (ponita/nn/conv.py:105-138; PyG sum aggregation onto edge_index[1]), ConvNext.forward (ponita/nn/convnext.py:20-33),
// the read-outs (ponita.py:126-155); the backward is what autograd derives from those (training_step,
// lightning_wrappers/diffusion.py:108-118).  Gradients are returned in the state_dict layout (arreau_state_dict with
// DEVICE pointers; non-trainable entries are ignored).
#include <string.h>
#include <algorithm>
#include <vector>

#include "internal.h"
#include "sgemm.h"

namespace {
// out[c] = scale * sum_r a[r][c] * (b ? b[r][c] : 1)  (+ out[c] if accumulate), deterministic, in ONE launch: each
// workgroup sums one chunk of rows for 64 columns (four row phases added in a fixed order) into part[chunk][c]; the
// workgroup that finishes last for its column group (device-scope counter) adds the chunks in a fixed order (four
// interleaved phases, combined in order) and resets the counter, so the result does not depend on which workgroup that was.
constexpr int COLSUM_MAX_CHUNKS = 256;
constexpr int COLCOUNT_INTS = 64;
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ a, const float* __restrict__ b, long rows,
                                                     int cols, long rows_per_chunk, float* __restrict__ part,
                                                     int* __restrict__ counters, float scale, int accumulate,
                                                     float* __restrict__ out) {
    __shared__ float sh[4][64];
    __shared__ int is_last;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
    const long r0 = (long)blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    float s = 0.f;
    if (c < cols) {
        long r = r0 + ph;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // eight loads in flight per thread; combined in a fixed order
        const size_t st = (size_t)4 * cols;
        for (; r + 28 < r1; r += 32) {
            const size_t i0 = (size_t)r * cols + c;
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += a[i0 + u * st] * (b ? b[i0 + u * st] : 1.0f);
        }
        for (; r < r1; r += 4) acc[0] += a[(size_t)r * cols + c] * (b ? b[(size_t)r * cols + c] : 1.0f);
        s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    }
    sh[ph][threadIdx.x & 63] = s;
    __syncthreads();
    if (ph == 0 && c < cols)
        part[(size_t)blockIdx.y * cols + c] = ((sh[0][threadIdx.x] + sh[1][threadIdx.x]) + sh[2][threadIdx.x]) + sh[3][threadIdx.x];
    __threadfence();  // this workgroup's partial sums are visible device-wide before it is counted
    __syncthreads();
    if (threadIdx.x == 0) is_last = atomicAdd(&counters[blockIdx.x], 1) == (int)gridDim.y - 1;
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    // (relaxed device-scope atomic loads: written by other workgroups during this launch, so they must not be served from
    // this CU's cache -- and unlike volatile accesses the compiler may keep many of them in flight)
    float tot = 0.f;
    if (c < cols) {
        int i = ph;
        const int n = (int)gridDim.y;
        for (; i + 28 < n; i += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = __hip_atomic_load(part + (size_t)(i + 4 * u) * cols + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int u = 0; u < 8; ++u) tot += v[u];
        }
        for (; i < n; i += 4)
            tot += __hip_atomic_load(part + (size_t)i * cols + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();  // (sh was last read before the counter)
    sh[ph][threadIdx.x & 63] = tot;
    __syncthreads();
    if (ph == 0 && c < cols) {
        tot = ((sh[0][threadIdx.x] + sh[1][threadIdx.x]) + sh[2][threadIdx.x]) + sh[3][threadIdx.x];
        out[c] = scale * tot + (accumulate ? out[c] : 0.f);
    }
    if (threadIdx.x == 0) counters[blockIdx.x] = 0;  // ready for the next launch (stream-ordered)
}

// The same sum for cols % 4 == 0, cols <= 1024, on 16-byte columns: a workgroup covers whole rows (thread = float4 column
// c4 and row phase ph of P = 256 / (cols / 4)), so every wave reads contiguous kilobytes instead of 256-byte pieces of
// rows 2 KB apart.  Two launches (round 2, third version), no atomics and no fences: the pass is latency-bound, not
// bandwidth-bound (the one-launch form took 27 us for 17 MB: 133 workgroups of 64 rows, then ONE workgroup adding the 133
// partial rows behind a device-scope fence; a one-launch two-level tree with 532 workgroups was slower still -- every
// workgroup's release fence writes the L2 back), so a chunk is 16 rows (532 workgroups on the [8512, 512] hidden-layer
// gradients: enough bytes in flight to cover the HBM latency) and a second small launch adds the chunk rows: 32 row
// phases per column in a fixed order, then the phases in order.  Deterministic.
constexpr int COLSUM4_MAX_CHUNKS = 1024;
constexpr int COLSUM4_MAX_BATCH = 8;  // matrices per batched call (the layers of the network)
// GELU_BWD (round 5): `a` is a gradient that still has to pass through a GELU -- a *= gelu'(gpre) * rowscale[row], gelu_backward_kernel4's
// expression, written back in place -- and the column sums are those of the result: the bias gradient of a layer whose d(pre-activation)
// comes out of a 128 x 128 product costs no pass of its own.
// (device body: one matrix, workgroup blockIdx.x = row chunk; `part` / `part2` = this matrix's chunk rows)
template <bool GELU_BWD>
__device__ __forceinline__ void colsum4_partial_body(std::conditional_t<GELU_BWD, f32x4*, const f32x4*> __restrict__ a,
                                                     const f32x4* __restrict__ b, long rows, int cols4, long rows_per_chunk,
                                                     f32x4* __restrict__ part, f32x4* __restrict__ part2, f32x4* __restrict__ scaled_out,
                                                     const f32x4* __restrict__ colscale, const f32x4* __restrict__ gpre,
                                                     const float* __restrict__ rowscale) {
    __shared__ f32x4 sh[256];
    auto fetch = [&](size_t i, long row) {
        f32x4 av = a[i];
        if constexpr (GELU_BWD) {
            const float rs = rowscale ? rowscale[row] : 1.0f;
            const f32x4 pv = gpre[i];
            using arreau_sgemm_detail::sg_gelu_grad;
            av = f32x4{av[0] * sg_gelu_grad(pv[0]) * rs, av[1] * sg_gelu_grad(pv[1]) * rs, av[2] * sg_gelu_grad(pv[2]) * rs,
                       av[3] * sg_gelu_grad(pv[3]) * rs};
            a[i] = av;
        }
        return av;
    };
    const int P = 256 / cols4;
    const int c4 = threadIdx.x % cols4, ph = threadIdx.x / cols4;
    const bool act = ph < P;
    const long r0 = (long)blockIdx.x * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[4] = {zero, zero, zero, zero};  // four 16-byte loads (per operand) in flight; combined in a fixed order
    f32x4 plain[4] = {zero, zero, zero, zero};
    const bool dual = part2 != nullptr;
    const f32x4 cs = scaled_out && act ? colscale[c4] : zero;
    if (act) {
        long r = r0 + ph;
        for (; r + 3 * P < r1; r += 4 * P) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t i = (size_t)(r + u * P) * cols4 + c4;
                const f32x4 av = fetch(i, r + u * P);
                acc[u] += b ? av * b[i] : av;
                if (dual) plain[u] += av;
                if (scaled_out) scaled_out[i] = av * cs;
            }
        }
        for (; r < r1; r += P) {
            const size_t i = (size_t)r * cols4 + c4;
            const f32x4 av = fetch(i, r);
            acc[0] += b ? av * b[i] : av;
            if (dual) plain[0] += av;
            if (scaled_out) scaled_out[i] = av * cs;
        }
    }
    auto block_sum = [&](f32x4 v, f32x4* dst) {  // phases added in order by the phase-0 thread of each column
        __syncthreads();
        sh[threadIdx.x] = v;
        __syncthreads();
        if (ph == 0) {
            f32x4 tot = sh[c4];
            for (int p = 1; p < P; ++p) tot += sh[p * cols4 + c4];
            dst[(size_t)blockIdx.x * cols4 + c4] = tot;
        }
    };
    block_sum((acc[0] + acc[1]) + (acc[2] + acc[3]), part);
    if (dual) block_sum((plain[0] + plain[1]) + (plain[2] + plain[3]), part2);
}
template <bool GELU_BWD>
__global__ __launch_bounds__(256) void colsum4_partial_kernel(std::conditional_t<GELU_BWD, f32x4*, const f32x4*> __restrict__ a,
                                                              const f32x4* __restrict__ b, long rows, int cols4, long rows_per_chunk,
                                                              f32x4* __restrict__ part,   // [batch][chunks][cols4]
                                                              f32x4* __restrict__ part2,  // plain sums of a (or null)
                                                              long a_bs4, long b_bs4,     // blockIdx.y = matrix of the batch (strides in float4)
                                                              f32x4* __restrict__ scaled_out = nullptr,      // also a * colscale (or null):
                                                              const f32x4* __restrict__ colscale = nullptr,  // d(out) = d(x) * layer_scale in the same pass
                                                              const f32x4* __restrict__ gpre = nullptr, const float* __restrict__ rowscale = nullptr)
{
    colsum4_partial_body<GELU_BWD>(a + (long)blockIdx.y * a_bs4, b ? b + (long)blockIdx.y * b_bs4 : nullptr, rows, cols4, rows_per_chunk,
                                   part + (size_t)blockIdx.y * gridDim.x * cols4, part2 ? part2 + (size_t)blockIdx.y * gridDim.x * cols4 : nullptr,
                                   scaled_out, colscale, gpre, rowscale);
}
// Round 5: the chunk sums of SEVERAL matrices of different widths in one launch (blockIdx.y = descriptor; all share the row count and the
// chunking): the three batched passes of a backward pass (L x linear_1.bias, L x norm weight + bias, L x conv bias) as one.
struct ColsumPartialDesc {
    const f32x4* a;
    const f32x4* b;
    f32x4* part;
    f32x4* part2;
    int cols4;
};
constexpr int COLSUM_PARTIAL_MULTI_MAX = 24;
struct ColsumPartialList {
    ColsumPartialDesc d[COLSUM_PARTIAL_MULTI_MAX];
};
struct ColsumGather {   // host side: passes collected for one launch
    ColsumPartialList list{};
    int n = 0, chunks = 0;
    long rows = 0, rpc = 0;
};
__global__ __launch_bounds__(256) void colsum4_partial_multi_kernel(ColsumPartialList list, long rows, long rows_per_chunk) {
    const ColsumPartialDesc d = list.d[blockIdx.y];
    colsum4_partial_body<false>(d.a, d.b, rows, d.cols4, rows_per_chunk, d.part, d.part2, nullptr, nullptr, nullptr, nullptr);
}
// out[c] = scale * sum_chunks part[chunk][c] (+ out[c]); out2[c] = colscale2[c] * sum_chunks part2[chunk][c].
// Workgroup = 8 float4 columns x 32 row phases.
__global__ __launch_bounds__(256) void colsum4_final_kernel(const f32x4* __restrict__ part, const f32x4* __restrict__ part2,
                                                            int chunks, int cols4, float scale, int accumulate,
                                                            float* __restrict__ out, float* __restrict__ out2,
                                                            const float* __restrict__ colscale2,
                                                            long out_bs, long out2_bs) {  // blockIdx.y = matrix of the batch
    __shared__ f32x4 sh[256];
    part += (size_t)blockIdx.y * chunks * cols4;
    if (part2) part2 += (size_t)blockIdx.y * chunks * cols4;
    out += (long)blockIdx.y * out_bs;
    if (out2) out2 += (long)blockIdx.y * out2_bs;
    const int cl = threadIdx.x & 7, ph = threadIdx.x >> 3;
    const int c4 = blockIdx.x * 8 + cl;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    auto column_sum = [&](const f32x4* src) {
        f32x4 tot = zero;
        if (c4 < cols4)
            for (int i = ph; i < chunks; i += 32 * 8) {  // eight loads in flight, added in row order
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = i + 32 * u < chunks ? src[(size_t)(i + 32 * u) * cols4 + c4] : zero;
#pragma unroll
                for (int u = 0; u < 8; ++u) tot += v[u];
            }
        __syncthreads();
        sh[threadIdx.x] = tot;
        __syncthreads();
        f32x4 r = zero;
        if (ph == 0) {
            r = sh[cl];
            for (int p = 1; p < 32; ++p) r += sh[p * 8 + cl];
        }
        return r;
    };
    const f32x4 tot = column_sum(part);
    if (ph == 0 && c4 < cols4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) out[4 * c4 + q] = scale * tot[q] + (accumulate ? out[4 * c4 + q] : 0.f);
    }
    if (part2 != nullptr) {
        const f32x4 tot2 = column_sum(part2);
        if (ph == 0 && c4 < cols4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) out2[4 * c4 + q] = (colscale2 ? colscale2[4 * c4 + q] : 1.0f) * tot2[q];
        }
    }
}

// Round 5: the chunk sums of SEVERAL column-sum passes in one launch.  The bias / norm / layer-scale gradients are results nothing in
// the backward pass waits for, so each pass leaves its chunk rows in its own piece of a scratch region plus a descriptor, and one
// launch at the end of the pass adds them all: colsum4_final_kernel's sums per descriptor (blockIdx.y), bit-identical.
struct ColsumFinalDesc {
    const f32x4* part;
    const f32x4* part2;
    float* out;
    float* out2;
    const float* colscale2;
    int chunks, cols4;
    float scale;
    int accumulate;
};
constexpr int COLSUM_DEFER_MAX = 48;
struct ColsumFinalList {
    ColsumFinalDesc d[COLSUM_DEFER_MAX];
};
__global__ __launch_bounds__(256) void colsum4_final_multi_kernel(ColsumFinalList list) {
    __shared__ f32x4 sh[256];
    const ColsumFinalDesc& d = list.d[blockIdx.y];
    const int cols4 = d.cols4, chunks = d.chunks;
    if ((int)blockIdx.x * 8 >= cols4) return;  // (uniform)
    const f32x4* __restrict__ part = d.part;
    const f32x4* __restrict__ part2 = d.part2;
    const int cl = threadIdx.x & 7, ph = threadIdx.x >> 3;
    const int c4 = blockIdx.x * 8 + cl;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    auto column_sum = [&](const f32x4* src) {
        f32x4 tot = zero;
        if (c4 < cols4)
            for (int i = ph; i < chunks; i += 32 * 8) {  // eight loads in flight, added in row order
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = i + 32 * u < chunks ? src[(size_t)(i + 32 * u) * cols4 + c4] : zero;
#pragma unroll
                for (int u = 0; u < 8; ++u) tot += v[u];
            }
        __syncthreads();
        sh[threadIdx.x] = tot;
        __syncthreads();
        f32x4 r = zero;
        if (ph == 0) {
            r = sh[cl];
            for (int p = 1; p < 32; ++p) r += sh[p * 8 + cl];
        }
        return r;
    };
    const f32x4 tot = column_sum(part);
    if (ph == 0 && c4 < cols4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) d.out[4 * c4 + q] = d.scale * tot[q] + (d.accumulate ? d.out[4 * c4 + q] : 0.f);
    }
    if (part2 != nullptr) {
        const f32x4 tot2 = column_sum(part2);
        if (ph == 0 && c4 < cols4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) d.out2[4 * c4 + q] = (d.colscale2 ? d.colscale2[4 * c4 + q] : 1.0f) * tot2[q];
        }
    }
}

// (the products' epilogues use the same two functions: sgemm.h)
__device__ __forceinline__ float gelu_exact(float x) { return arreau_sgemm_detail::sg_gelu_exact(x); }
__device__ __forceinline__ float gelu_grad(float x) { return arreau_sgemm_detail::sg_gelu_grad(x); }

// pre[r][c] += bias[c]; act[r][c] = gelu(pre) * (rowscale ? rowscale[r] : 1)
__global__ void bias_gelu_kernel(float* __restrict__ pre, const float* __restrict__ bias, const float* __restrict__ rowscale,
                                 long rows, int cols, float* __restrict__ act) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const long r = i / cols;
    const int c = (int)(i % cols);
    const float v = pre[i] + bias[c];
    pre[i] = v;
    act[i] = gelu_exact(v) * (rowscale ? rowscale[r] : 1.0f);
}
// g[r][c] = g[r][c] * gelu'(pre[r][c]) * (rowscale ? rowscale[r] : 1)
__global__ void gelu_backward_kernel(float* __restrict__ g, const float* __restrict__ pre, const float* __restrict__ rowscale,
                                     long rows, int cols) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    g[i] = g[i] * gelu_grad(pre[i]) * (rowscale ? rowscale[i / cols] : 1.0f);
}
// the same two maps on 16-byte columns (cols % 4 == 0): four elements per thread, the same arithmetic per element
__global__ void bias_gelu_kernel4(f32x4* __restrict__ pre, const f32x4* __restrict__ bias, const float* __restrict__ rowscale,
                                  long rows, int cols4, f32x4* __restrict__ act) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols4) return;
    const float rs = rowscale ? rowscale[i / cols4] : 1.0f;
    const f32x4 v = pre[i] + bias[i % cols4];
    pre[i] = v;
    act[i] = f32x4{gelu_exact(v[0]) * rs, gelu_exact(v[1]) * rs, gelu_exact(v[2]) * rs, gelu_exact(v[3]) * rs};
}
__global__ void gelu_backward_kernel4(f32x4* __restrict__ g, const f32x4* __restrict__ pre, const float* __restrict__ rowscale,
                                      long rows, int cols4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols4) return;
    const float rs = rowscale ? rowscale[i / cols4] : 1.0f;
    const f32x4 gv = g[i], pv = pre[i];
    g[i] = f32x4{gv[0] * gelu_grad(pv[0]) * rs, gv[1] * gelu_grad(pv[1]) * rs, gv[2] * gelu_grad(pv[2]) * rs,
                 gv[3] * gelu_grad(pv[3]) * rs};
}
__global__ void add_bias_kernel(float* __restrict__ x, const float* __restrict__ bias, long rows, int cols) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows * cols) x[i] += bias[i % cols];
}

// (edge slot, orientation) rows: pair invariants + edge scalars (transforms/invariants.py:82-88, geometry/invariants.py:
// 10-31), cut-off window (windowing.py:21-29; 0 for unused slots) and the 83 distinct monomials (columns 83..95 zero).
__global__ void edge_rows_kernel(const float* __restrict__ dir, const float* __restrict__ dist, const int32_t* __restrict__ deg,
                                 const int32_t* __restrict__ batch, const float* __restrict__ lattice,
                                 const float* __restrict__ ori, float r_max, int N, int k, float* __restrict__ mono,
                                 float* __restrict__ window) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= (long)N * k * 16) return;
    const int o = (int)(r & 15);
    const long e = r >> 4;
    const int n = (int)(e / k), s = (int)(e % k);
    const float dx = dir[3 * e], dy = dir[3 * e + 1], dz = dir[3 * e + 2], d = dist[e];
    const float ox = ori[3 * o], oy = ori[3 * o + 1], oz = ori[3 * o + 2];
    float a[6];
    a[0] = (dx * ox + dy * oy) + dz * oz;
    const float rx = dx - a[0] * ox, ry = dy - a[0] * oy, rz = dz - a[0] * oz;
    a[1] = sqrtf((rx * rx + ry * ry) + rz * rz);
    a[2] = d;
    const float* Lm = lattice + 9 * (size_t)batch[n];
    const float dn = fmaxf(sqrtf((dx * dx + dy * dy) + dz * dz), 1e-8f);
    for (int i = 0; i < 3; ++i) {
        const float lx = Lm[3 * i], ly = Lm[3 * i + 1], lz = Lm[3 * i + 2];
        const float ln = fmaxf(sqrtf((lx * lx + ly * ly) + lz * lz), 1e-8f);
        a[3 + i] = ((dx / dn) * (lx / ln) + (dy / dn) * (ly / ln)) + (dz / dn) * (lz / ln);
    }
    const float u = d / r_max, u2 = u * u, u6 = u2 * u2 * u2;
    window[r] = (s < deg[n] && d < r_max) ? 1.0f - 28.0f * u6 + 48.0f * u6 * u - 21.0f * u6 * u2 : 0.0f;
    float* mrow = mono + r * ARREAU_MONO_PAD;
    int p = 0;
    for (int i = 0; i < 6; ++i) mrow[p++] = a[i];
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j) mrow[p++] = a[i] * a[j];
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j)
            for (int q = j; q < 6; ++q) mrow[p++] = (a[i] * a[j]) * a[q];
    for (; p < ARREAU_MONO_PAD; ++p) mrow[p] = 0.f;
}

// fiber attributes: poly3(ori_o . ori_p)  [256][3]  (geometry/invariants.py:24; embedding.py:10-14 with one input)
__global__ void fiber_poly_kernel(const float* __restrict__ ori, float* __restrict__ fpoly) {
    const int i = threadIdx.x;  // 256 threads
    const int o = i >> 4, p = i & 15;
    const float a = (ori[3 * o] * ori[3 * p] + ori[3 * o + 1] * ori[3 * p + 1]) + ori[3 * o + 2] * ori[3 * p + 2];
    fpoly[3 * i] = a;
    fpoly[3 * i + 1] = a * a;
    fpoly[3 * i + 2] = (a * a) * a;
}

// dense input features of x_embedder, one row per (atom, orientation):  [one_hot(type) S | t_emb 64 | n | lengths 3 |
// angles 3 | |lengths/n| 3 | vec . ori 4]   (diffusion_loss.py:124-158, position_orientation_graph.py:82-86)
__global__ void features_kernel(const float* __restrict__ frac, const int32_t* __restrict__ types,
                                const float* __restrict__ lengths, const float* __restrict__ angles,
                                const int32_t* __restrict__ tstep, const int32_t* __restrict__ offsets,
                                const int32_t* __restrict__ batch, const float* __restrict__ lattice,
                                const float* __restrict__ betas, const float* __restrict__ t_emb_w,
                                const float* __restrict__ ori, int S, int T, int N, float* __restrict__ F) {
    const int row = blockIdx.x;  // (n, o)
    const int n = row >> 4, o = row & 15;
    const int b = batch[n];
    const int FW = S + 78;
    float* f = F + (size_t)row * FW;
    const float nat = (float)(offsets[b + 1] - offsets[b]);
    int t = tstep[b];
    t = t < 0 ? 0 : (t > T ? T : t);
    const int ty = min(max(types[n], 0), S - 1);
    for (int i = threadIdx.x; i < FW; i += blockDim.x) {
        float v;
        if (i < S) v = i == ty ? 1.0f : 0.0f;
        else if (i < S + 64) {
            const int j = i - S;
            const float proj = ((betas[t] * t_emb_w[j & 31]) * 2.0f) * 3.14159265358979323846f;
            v = j < 32 ? sinf(proj) : cosf(proj);
        } else if (i == S + 64) v = nat;
        else if (i < S + 68) v = lengths[3 * b + (i - S - 65)];
        else if (i < S + 71) v = angles[3 * b + (i - S - 68)];
        else if (i < S + 74) v = fabsf(lengths[3 * b + (i - S - 71)] / nat);
        else {
            const int q = i - S - 74;
            const float* vv = q == 0 ? frac + 3 * (size_t)n : lattice + 9 * (size_t)b + 3 * (q - 1);
            v = (vv[0] * ori[3 * o] + vv[1] * ori[3 * o + 1]) + vv[2] * ori[3 * o + 2];
        }
        f[i] = v;
    }
}

// x1[n,o,c] = sum_{s < deg[n]} kern[(n,s,o),c] * x[src(n,s),o,c]   (conv.py:111,131-133 + sum aggregation)
// (kern is one layer's column block of the all-layer kernel matrix [R][L*C]: row stride ldk)
__global__ void conv_forward_kernel(const float* __restrict__ kern, int ldk, const float* __restrict__ x,
                                    const int32_t* __restrict__ deg, const int32_t* __restrict__ src, int N, int k, int C,
                                    float* __restrict__ x1) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * 16 * C) return;
    const int c = (int)(i % C);
    const long row = i / C;
    const int o = (int)(row & 15), n = (int)(row >> 4);
    const int nd = min(deg[n], k);
    float acc = 0.f;
    for (int s = 0; s < nd; ++s) {
        const int j = src[(size_t)n * k + s];
        // (product rounded, then added in edge order: messages = kernel * x, then index_add_ -- conv.py:131-133; the fused kernels' form)
        acc = __fadd_rn(acc, __fmul_rn(kern[(((size_t)n * k + s) * 16 + o) * ldk + c], x[((size_t)j * 16 + o) * C + c]));
    }
    x1[i] = acc;
}
// Sender-side adjacency of the batch (round 4; it replaces an fp32 atomicAdd scatter, so the weight gradients are now bit
// for bit reproducible).  Edges never leave a crystal, so crystal b's reversed lists live in the k * n_b entries behind
// rev_idx[k * off[b]]: one workgroup per crystal.  The crystal's slot table (sender of every (receiver, slot), -1 for unused
// slots) is copied into LDS once; then ONE WAVE PER SENDER walks it 64 slots at a time -- a ballot of "this slot is mine", its
// population count is the sender's out-degree, the count of set bits below a lane is that slot's place in the list -- first to
// count, then, after an ordered scan of the counts, to fill (sixteen waves per workgroup).  The lists come out in (receiver, slot) order, the order the forward
// pass adds messages in.  k n_b^2 / 64 wave steps per crystal (2.7 us at 64 atoms; the first version, one THREAD per sender with
// two serial scans of the table, took 38 us per training step).  Crystals above RADJ_SLOTS / k atoms read the table from global memory.
constexpr int RADJ_SLOTS = 4096;
__global__ __launch_bounds__(1024) void reverse_adjacency_kernel(const int32_t* __restrict__ off, const int32_t* __restrict__ deg,
                                                                const int32_t* __restrict__ src, int k,
                                                                int32_t* __restrict__ rev_start /*[N]*/, int32_t* __restrict__ rev_cnt /*[N]*/,
                                                                int32_t* __restrict__ rev_idx /*[N*k]: slot index (n * k + s)*/) {
    const int b = blockIdx.x, a0 = off[b], a1 = off[b + 1], nb = a1 - a0, nslots = nb * k;
    __shared__ int32_t s_src[RADJ_SLOTS];
    const bool in_lds = nslots <= RADJ_SLOTS;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    auto sender_of = [&](int e) {  // e = slot index inside the crystal; -1: unused slot
        const int n = a0 + e / k, sl = e - (e / k) * k;
        return sl < min(deg[n], k) ? src[(size_t)n * k + sl] : -1;
    };
    if (in_lds)
        for (int e = threadIdx.x; e < nslots; e += blockDim.x) s_src[e] = sender_of(e);
    __syncthreads();
    auto slot_sender = [&](int e) { return in_lds ? s_src[e] : sender_of(e); };
    // pass 1: out-degree of every sender
    for (int j = a0 + wave; j < a1; j += nwaves) {
        int cnt = 0;
        for (int e0 = 0; e0 < nslots; e0 += 64) {
            const int e = e0 + lane;
            cnt += __builtin_popcountll(__ballot(e < nslots && slot_sender(e) == j));
        }
        if (lane == 0) rev_cnt[j] = cnt;
    }
    __syncthreads();  // (the counts are read back by this workgroup only: workgroup-scope visibility)
    // ordered exclusive scan of the counts: wave 0, 64 senders at a time with a carried base
    if (wave == 0) {
        int base = a0 * k;
        for (int j0 = a0; j0 < a1; j0 += 64) {
            const int j = j0 + lane;
            const int c = j < a1 ? rev_cnt[j] : 0;
            int incl = c;
            for (int d = 1; d < 64; d <<= 1) {
                const int v = __shfl_up(incl, d, 64);
                if (lane >= d) incl += v;
            }
            if (j < a1) rev_start[j] = base + incl - c;
            base += __shfl(incl, 63, 64);
        }
    }
    __syncthreads();
    // pass 2: the lists, in (receiver, slot) order
    for (int j = a0 + wave; j < a1; j += nwaves) {
        int w = rev_start[j];
        for (int e0 = 0; e0 < nslots; e0 += 64) {
            const int e = e0 + lane;
            const bool mine = e < nslots && slot_sender(e) == j;
            const unsigned long long mask = __ballot(mine);
            if (mine) rev_idx[w + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = a0 * k + e;
            w += __builtin_popcountll(mask);
        }
    }
}
// dkern[(n,s,o),c] = dx1[n,o,c] * x[src,o,c]   (receiver side, one thread per element of the kernel matrix)
__global__ void conv_backward_kern_kernel(const float* __restrict__ x, const float* __restrict__ dx1, const int32_t* __restrict__ deg,
                                          const int32_t* __restrict__ src, int N, int k, int C, int ldk, float* __restrict__ dkern) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * k * 16 * C) return;
    const int c = (int)(i % C);
    const long row = i / C;
    const int o = (int)(row & 15);
    const long e = row >> 4;
    const int n = (int)(e / k), s = (int)(e % k);
    const size_t ik = (size_t)row * ldk + c;  // same position in the [R][L*C] matrices
    if (s >= min(deg[n], k)) { dkern[ik] = 0.f; return; }
    const int j = src[e];
    dkern[ik] = dx1[((size_t)n * 16 + o) * C + c] * x[((size_t)j * 16 + o) * C + c];
}
// dx[j,o,c] += sum over the edges (n, s) that j sends, in (receiver, slot) order, of kern[(n,s,o),c] * dx1[n,o,c]
// (sender side: one thread owns one element of dx -- a segmented sum in a fixed order, no atomics)
__global__ void conv_backward_dx_kernel(const float* __restrict__ kern, int ldk, const float* __restrict__ dx1,
                                        const int32_t* __restrict__ rev_start, const int32_t* __restrict__ rev_cnt,
                                        const int32_t* __restrict__ rev_idx, int N, int k, int C, const float* dx_in,
                                        const float* __restrict__ add2, float* dx) {
    // dx = dx_in + (sender-side sum) + add2: dx_in / dx may be the same array (element-wise); add2 = the read-out's contribution to the
    // NEXT layer down (null at layer 0): d x_l is complete when this launch ends
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * 16 * C) return;
    const int c = (int)(i % C);
    const long row = i / C;
    const int o = (int)(row & 15), j = (int)(row >> 4);
    const int st = rev_start[j], cnt = rev_cnt[j];
    float acc = 0.f;
    for (int q = 0; q < cnt; ++q) {
        const int e = rev_idx[st + q], n = e / k;
        acc += kern[((size_t)e * 16 + o) * ldk + c] * dx1[((size_t)n * 16 + o) * C + c];
    }
    float v = dx_in[i] + acc;
    if (add2) v += add2[i];
    dx[i] = v;
}
// x2[n,p,c] = sum_o x1[n,o,c] fk[o,p,c] / 16 + bias[c]   (conv.py:113-127)
// float4 forms of the five kernels around the spatial / spherical convolution (C and the kernel pitch multiples of four; the
// same sums in the same order per element, so the results are bit-identical to the scalar forms): a thread owns four channels,
// a quarter of the address arithmetic and of the load instructions, 16 bytes per lane and request.
__global__ void conv_forward_kernel4(const f32x4* __restrict__ kern, int ldk4, const f32x4* __restrict__ x,
                                     const int32_t* __restrict__ deg, const int32_t* __restrict__ src, int N, int k, int C4,
                                     f32x4* __restrict__ x1) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * 16 * C4) return;
    const int c = (int)(i % C4);
    const long row = i / C4;
    const int o = (int)(row & 15), n = (int)(row >> 4);
    const int nd = min(deg[n], k);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int s = 0; s < nd; ++s) {
        const int j = src[(size_t)n * k + s];
        const f32x4 kv = kern[(((size_t)n * k + s) * 16 + o) * ldk4 + c], xv = x[((size_t)j * 16 + o) * C4 + c];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = __fadd_rn(acc[q], __fmul_rn(kv[q], xv[q]));
    }
    x1[i] = acc;
}
// (device bodies: shared by the two kernels and by the one-launch form below)
__device__ __forceinline__ void conv_backward_kern_body4(long i, const f32x4* __restrict__ x, const f32x4* __restrict__ dx1,
                                                         const int32_t* __restrict__ deg, const int32_t* __restrict__ src, int N, int k,
                                                         int C4, int ldk4, f32x4* __restrict__ dkern) {
    if (i >= (long)N * k * 16 * C4) return;
    const int c = (int)(i % C4);
    const long row = i / C4;
    const int o = (int)(row & 15);
    const long e = row >> 4;
    const int n = (int)(e / k), s = (int)(e % k);
    const size_t ik = (size_t)row * ldk4 + c;
    if (s >= min(deg[n], k)) { dkern[ik] = f32x4{0.f, 0.f, 0.f, 0.f}; return; }
    const int j = src[e];
    const f32x4 a = dx1[((size_t)n * 16 + o) * C4 + c], b = x[((size_t)j * 16 + o) * C4 + c];
    dkern[ik] = f32x4{a[0] * b[0], a[1] * b[1], a[2] * b[2], a[3] * b[3]};
}
__device__ __forceinline__ void conv_backward_dx_body4(long i, const f32x4* __restrict__ kern, int ldk4, const f32x4* __restrict__ dx1,
                                                       const int32_t* __restrict__ rev_start, const int32_t* __restrict__ rev_cnt,
                                                       const int32_t* __restrict__ rev_idx, int N, int k, int C4, const f32x4* dx_in,
                                                       const f32x4* __restrict__ add2, f32x4* dx,
                                                       // round 5: also d(out) of the NEXT layer down = d(x_l) * its layer scale (what the
                                                       // column-sum pass at the top of that iteration wrote: the same multiply), or null
                                                       f32x4* __restrict__ dout_next = nullptr, const f32x4* __restrict__ ls_next = nullptr) {
    if (i >= (long)N * 16 * C4) return;
    const int c = (int)(i % C4);
    const long row = i / C4;
    const int o = (int)(row & 15), j = (int)(row >> 4);
    const int st = rev_start[j], cnt = rev_cnt[j];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int q = 0; q < cnt; ++q) {
        const int e = rev_idx[st + q], n = e / k;
        const f32x4 kv = kern[((size_t)e * 16 + o) * ldk4 + c], dv = dx1[((size_t)n * 16 + o) * C4 + c];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] += kv[r] * dv[r];
    }
    f32x4 d = dx_in[i];
#pragma unroll
    for (int r = 0; r < 4; ++r) d[r] += acc[r];
    if (add2) {
        const f32x4 a2 = add2[i];
#pragma unroll
        for (int r = 0; r < 4; ++r) d[r] += a2[r];
    }
    dx[i] = d;
    if (dout_next) dout_next[i] = d * ls_next[c];
}
__global__ void conv_backward_kern_kernel4(const f32x4* __restrict__ x, const f32x4* __restrict__ dx1, const int32_t* __restrict__ deg,
                                           const int32_t* __restrict__ src, int N, int k, int C4, int ldk4, f32x4* __restrict__ dkern) {
    conv_backward_kern_body4((long)blockIdx.x * blockDim.x + threadIdx.x, x, dx1, deg, src, N, k, C4, ldk4, dkern);
}
__global__ void conv_backward_dx_kernel4(const f32x4* __restrict__ kern, int ldk4, const f32x4* __restrict__ dx1,
                                         const int32_t* __restrict__ rev_start, const int32_t* __restrict__ rev_cnt,
                                         const int32_t* __restrict__ rev_idx, int N, int k, int C4, const f32x4* dx_in,
                                         const f32x4* __restrict__ add2, f32x4* dx) {
    conv_backward_dx_body4((long)blockIdx.x * blockDim.x + threadIdx.x, kern, ldk4, dx1, rev_start, rev_cnt, rev_idx, N, k, C4, dx_in, add2, dx);
}
// Round 5: both gradients of the spatial conv in ONE launch -- they read the same d(x1) and write disjoint arrays, so nothing orders
// them: the first `dx_blocks` workgroups run the sender-side sum (the launch the layer loop waits for: gather chains, started
// first), the rest the receiver-side products (a 35 MB store stream per layer at 64 crystals).  Same arithmetic per element.
__global__ void conv_backward_both_kernel4(int dx_blocks, const f32x4* __restrict__ x, const f32x4* __restrict__ dx1,
                                           const int32_t* __restrict__ deg, const int32_t* __restrict__ src, const f32x4* __restrict__ kern,
                                           int ldk4, const int32_t* __restrict__ rev_start, const int32_t* __restrict__ rev_cnt,
                                           const int32_t* __restrict__ rev_idx, int N, int k, int C4, const f32x4* dx_in,
                                           const f32x4* __restrict__ add2, f32x4* dx, f32x4* __restrict__ dkern,
                                           f32x4* __restrict__ dout_next, const f32x4* __restrict__ ls_next) {
    if ((int)blockIdx.x < dx_blocks)
        conv_backward_dx_body4((long)blockIdx.x * blockDim.x + threadIdx.x, kern, ldk4, dx1, rev_start, rev_cnt, rev_idx, N, k, C4, dx_in, add2, dx,
                               dout_next, ls_next);
    else
        conv_backward_kern_body4((long)(blockIdx.x - dx_blocks) * blockDim.x + threadIdx.x, x, dx1, deg, src, N, k, C4, ldk4, dkern);
}
// Round 5: the spatial conv and the spherical mix behind it as ONE launch, a workgroup per node: x1 (kept: the backward pass reads it)
// goes to memory and to LDS, the mix reads it from there.  The sums and their order per element are conv_forward_kernel4's and
// mix_forward_kernel4's.  Dynamic LDS: 16 * C4 float4.
__global__ __launch_bounds__(512) void conv_mix_forward_kernel4(const f32x4* __restrict__ kern, int ldk4, const f32x4* __restrict__ x,
                                                                const int32_t* __restrict__ deg, const int32_t* __restrict__ src, int N,
                                                                int k, int C4, const f32x4* __restrict__ fk, const f32x4* __restrict__ bias,
                                                                f32x4* __restrict__ x1, f32x4* __restrict__ x2) {
    extern __shared__ f32x4 s_node[];  // [16][C4]
    const int n = blockIdx.x, E = 16 * C4;
    const int nd = min(deg[n], k);
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        const int c = e % C4, o = e / C4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int s = 0; s < nd; ++s) {
            const int j = src[(size_t)n * k + s];
            const f32x4 kv = kern[(((size_t)n * k + s) * 16 + o) * ldk4 + c], xv = x[((size_t)j * 16 + o) * C4 + c];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __fadd_rn(acc[q], __fmul_rn(kv[q], xv[q]));
        }
        x1[(size_t)n * E + e] = acc;
        s_node[e] = acc;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        const int c = e % C4, p = e / C4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < 16; ++o) {
            const f32x4 a = s_node[o * C4 + c], b = fk[((size_t)o * 16 + p) * C4 + c];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += a[r] * b[r];
        }
        const f32x4 bv = bias[c];
        x2[(size_t)n * E + e] = f32x4{acc[0] * (1.0f / 16.0f) + bv[0], acc[1] * (1.0f / 16.0f) + bv[1], acc[2] * (1.0f / 16.0f) + bv[2],
                                      acc[3] * (1.0f / 16.0f) + bv[3]};
    }
}
__global__ void mix_forward_kernel4(const f32x4* __restrict__ x1, const f32x4* __restrict__ fk, const f32x4* __restrict__ bias,
                                    int N, int C4, f32x4* __restrict__ x2) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * 16 * C4) return;
    const int c = (int)(i % C4);
    const long row = i / C4;
    const int p = (int)(row & 15), n = (int)(row >> 4);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int o = 0; o < 16; ++o) {
        const f32x4 a = x1[((size_t)n * 16 + o) * C4 + c], b = fk[((size_t)o * 16 + p) * C4 + c];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] += a[r] * b[r];
    }
    const f32x4 bv = bias[c];
    x2[i] = f32x4{acc[0] * (1.0f / 16.0f) + bv[0], acc[1] * (1.0f / 16.0f) + bv[1], acc[2] * (1.0f / 16.0f) + bv[2],
                  acc[3] * (1.0f / 16.0f) + bv[3]};
}
__global__ void mix_backward_x_kernel4(const f32x4* __restrict__ dx2, const f32x4* __restrict__ fk, int N, int C4,
                                       f32x4* __restrict__ dx1) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * 16 * C4) return;
    const int c = (int)(i % C4);
    const long row = i / C4;
    const int o = (int)(row & 15), n = (int)(row >> 4);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const f32x4 a = dx2[((size_t)n * 16 + p) * C4 + c], b = fk[((size_t)o * 16 + p) * C4 + c];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] += a[r] * b[r];
    }
    dx1[i] = f32x4{acc[0] * (1.0f / 16.0f), acc[1] * (1.0f / 16.0f), acc[2] * (1.0f / 16.0f), acc[3] * (1.0f / 16.0f)};
}
__global__ void mix_forward_kernel(const float* __restrict__ x1, const float* __restrict__ fk, const float* __restrict__ bias,
                                   int N, int C, float* __restrict__ x2) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * 16 * C) return;
    const int c = (int)(i % C);
    const long row = i / C;
    const int p = (int)(row & 15), n = (int)(row >> 4);
    float acc = 0.f;
    for (int o = 0; o < 16; ++o) acc += x1[((size_t)n * 16 + o) * C + c] * fk[((size_t)o * 16 + p) * C + c];
    x2[i] = acc * (1.0f / 16.0f) + bias[c];
}
// dx1[n,o,c] = sum_p dx2[n,p,c] fk[o,p,c] / 16
__global__ void mix_backward_x_kernel(const float* __restrict__ dx2, const float* __restrict__ fk, int N, int C,
                                      float* __restrict__ dx1) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * 16 * C) return;
    const int c = (int)(i % C);
    const long row = i / C;
    const int o = (int)(row & 15), n = (int)(row >> 4);
    float acc = 0.f;
    for (int p = 0; p < 16; ++p) acc += dx2[((size_t)n * 16 + p) * C + c] * fk[((size_t)o * 16 + p) * C + c];
    dx1[i] = acc * (1.0f / 16.0f);
}
// dfk[o,p,c] = sum_n x1[n,o,c] dx2[n,p,c] / 16: chunks of 32 atoms summed by separate workgroups, then the chunks in
// order (deterministic)
constexpr int MIX_CHUNK = 16;
// d(fiber kernel)[o][p][c] = sum_n x1[n][o][c] dx2[n][p][c]: ONE block per (chunk of atoms, layer), a thread per channel holding all
// 16 x 16 sums (round 4; the first form ran a block per (o, p) pair and read both operands 16 times: 81 us at 64 crystals, most of
// it L2 traffic).  Atoms in ascending order inside a chunk, chunks added in order by the final kernel: the same sums as before.
__global__ __launch_bounds__(128) void mix_backward_fk_partial_kernel(const float* __restrict__ x1, const float* __restrict__ dx2, int N, int C,
                                                                      float* __restrict__ part /*[chunks][256][C]*/, int chunk = MIX_CHUNK) {
    const int n0 = blockIdx.x * chunk, n1 = min(N, n0 + chunk);
    // blockIdx.y = layer (the layers' x1 / dx2 / partial sums lie N * 16 * C, resp. chunks * 256 * C floats apart)
    x1 += (size_t)blockIdx.y * N * 16 * C;
    dx2 += (size_t)blockIdx.y * N * 16 * C;
    part += ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc[16][16];
#pragma unroll
        for (int o = 0; o < 16; ++o)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[o][q] = 0.f;
        float xn[16], dn[16];  // the next atom's rows travel while this atom's 256 products run
        auto load = [&](int n) {
#pragma unroll
            for (int o = 0; o < 16; ++o) {
                xn[o] = x1[((size_t)n * 16 + o) * C + c];
                dn[o] = dx2[((size_t)n * 16 + o) * C + c];
            }
        };
        if (n0 < n1) load(n0);
        for (int n = n0; n < n1; ++n) {
            float xv[16], dv[16];
#pragma unroll
            for (int o = 0; o < 16; ++o) { xv[o] = xn[o]; dv[o] = dn[o]; }
            if (n + 1 < n1) load(n + 1);
#pragma unroll
            for (int o = 0; o < 16; ++o)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[o][q] += xv[o] * dv[q];
        }
#pragma unroll
        for (int o = 0; o < 16; ++o)
#pragma unroll
            for (int q = 0; q < 16; ++q) part[(size_t)(o * 16 + q) * C + c] = acc[o][q];
    }
}
__global__ void mix_backward_fk_final_kernel(const float* __restrict__ part, int chunks, int C, float* __restrict__ dfk) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 256 * C) return;
    part += (size_t)blockIdx.y * chunks * 256 * C;  // blockIdx.y = layer
    dfk += (size_t)blockIdx.y * 256 * C;
    float acc = 0.f;
    for (int q = 0; q < chunks; ++q) acc += part[(size_t)q * 256 * C + i];
    dfk[i] = acc * (1.0f / 16.0f);
}
// LayerNorm over C (eps 1e-5, biased variance; convnext.py:25): xhat, rstd saved; y = xhat g + b.  One wave per row.
__global__ __launch_bounds__(256) void ln_forward_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                         const float* __restrict__ b, long rows, int C,
                                                         float* __restrict__ xhat, float* __restrict__ rstd_out,
                                                         float* __restrict__ y) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (size_t)row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    const float mean = s / (float)C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; q += d * d; }
    for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off, 64);
    const float rstd = 1.0f / sqrtf(q / (float)C + 1e-5f);
    if (lane == 0) rstd_out[row] = rstd;
    for (int c = lane; c < C; c += 64) {
        const float h = (xr[c] - mean) * rstd;
        xhat[(size_t)row * C + c] = h;
        y[(size_t)row * C + c] = h * g[c] + b[c];
    }
}
// dx = rstd * (dy - mean(dy) - xhat mean(dy xhat)),  dy = dyn * g
__global__ __launch_bounds__(256) void ln_backward_kernel(const float* __restrict__ dyn, const float* __restrict__ xhat,
                                                          const float* __restrict__ rstd, const float* __restrict__ g,
                                                          long rows, int C, float* __restrict__ dx) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float dy = dyn[(size_t)row * C + c] * g[c];
        s1 += dy;
        s2 += dy * xhat[(size_t)row * C + c];
    }
    for (int off = 32; off >= 1; off >>= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
    const float m1 = s1 / (float)C, m2 = s2 / (float)C, r = rstd[row];
    for (int c = lane; c < C; c += 64) {
        const float dy = dyn[(size_t)row * C + c] * g[c];
        dx[(size_t)row * C + c] = r * (dy - m1 - xhat[(size_t)row * C + c] * m2);
    }
}
// Round 5: LayerNorm backward and the spherical mix's backward in ONE launch, a workgroup of sixteen waves per node: wave p runs
// ln_backward_kernel's row (n, p) -- same lane sums, same butterflies -- and leaves d(x2) in memory (kept per layer for the batched bias
// and fiber-kernel gradients) and in LDS; after the barrier the workgroup forms d(x1) = mix^T d(x2) with mix_backward_x_kernel4's sums.
// Dynamic LDS: 16 * C floats.
__global__ __launch_bounds__(1024) void ln_mix_backward_kernel4(const float* __restrict__ dyn, const float* __restrict__ xhat,
                                                                const float* __restrict__ rstd, const float* __restrict__ g,
                                                                const f32x4* __restrict__ fk, int N, int C, float* __restrict__ dx2,
                                                                f32x4* __restrict__ dx1) {
    extern __shared__ f32x4 s_node[];  // [16][C / 4]
    float* s_dx2 = reinterpret_cast<float*>(s_node);
    const int n = blockIdx.x, p = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long row = (long)n * 16 + p;
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float dy = dyn[(size_t)row * C + c] * g[c];
        s1 += dy;
        s2 += dy * xhat[(size_t)row * C + c];
    }
    for (int off = 32; off >= 1; off >>= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
    const float m1 = s1 / (float)C, m2 = s2 / (float)C, r = rstd[row];
    for (int c = lane; c < C; c += 64) {
        const float dy = dyn[(size_t)row * C + c] * g[c];
        const float v = r * (dy - m1 - xhat[(size_t)row * C + c] * m2);
        dx2[(size_t)row * C + c] = v;
        s_dx2[p * C + c] = v;
    }
    __syncthreads();
    const int C4 = C / 4, E = 16 * C4;
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        const int c = e % C4, o = e / C4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const f32x4 a = s_node[q * C4 + c], b = fk[((size_t)o * 16 + q) * C4 + c];
#pragma unroll
            for (int w = 0; w < 4; ++w) acc[w] += a[w] * b[w];
        }
        dx1[(size_t)n * E + e] = f32x4{acc[0] * (1.0f / 16.0f), acc[1] * (1.0f / 16.0f), acc[2] * (1.0f / 16.0f), acc[3] * (1.0f / 16.0f)};
    }
}
// x_next = out * ls + x   (convnext.py:30-32)
__global__ void scale_residual_kernel(const float* __restrict__ out, const float* __restrict__ ls, const float* __restrict__ x,
                                      long rows, int C, float* __restrict__ xn) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows * C) xn[i] = out[i] * ls[i % C] + x[i];
}
// out[r][c] += bias[c];  xn[r][c] = out[r][c] * ls[c] + x[r][c]   (ConvNext tail, convnext.py:30-32)
__global__ void bias_scale_residual_kernel(float* __restrict__ out, const float* __restrict__ bias, const float* __restrict__ ls,
                                           const float* __restrict__ x, long rows, int C, float* __restrict__ xn) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * C) return;
    const float v = out[i] + bias[i % C];
    out[i] = v;
    xn[i] = v * ls[i % C] + x[i];
}
__global__ void scale_cols_kernel(const float* __restrict__ a, const float* __restrict__ colscale, long rows, int C,
                                  float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows * C) out[i] = a[i] * colscale[i % C];
}
__global__ void affine_cols_kernel(const float* __restrict__ a, const float* __restrict__ colscale, const float* __restrict__ colbias,
                                   long rows, int C, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows * C) out[i] = a[i] * colscale[i % C] + colbias[i % C];
}
// read-outs from rbar[(n,o)][S+4] = mean over layers (bias added here): logits = mean_o, eps = sum_o vec * ori / 16,
// gs = mean_o (ponita.py:108-117,126-155); len0 by readout_crystals-style ordered sums
// Round 5: `layers` > 0 -- rbar holds the L per-layer products [L][M][RO] and the sum over the layers is formed here, a running sum from
// zero in layer order per element: what the ordered-sum launch in front of this kernel wrote, bit for bit, without that launch.
__global__ void train_outputs_kernel(const float* __restrict__ rbar, const float* __restrict__ ro_b, const float* __restrict__ ori,
                                     int S, int L, int N, float* __restrict__ eps, float* __restrict__ logits,
                                     float* __restrict__ gs, int layers = 0) {
    const int n = blockIdx.x, RO = S + 4;
    const size_t layer_stride = (size_t)N * 16 * RO;
    auto rb = [&](int o, int j) {
        const size_t i = ((size_t)n * 16 + o) * RO + j;
        if (layers == 0) return rbar[i];
        float s = 0.f;
        for (int l = 0; l < layers; ++l) s += rbar[(size_t)l * layer_stride + i];
        return 1.0f * s + 0.0f;
    };
    for (int j = threadIdx.x; j < RO + 2; j += blockDim.x) {
        if (j < S || (j > S && j < RO)) {
            float bsum = 0.f;
            for (int l = 0; l < L; ++l) bsum += ro_b[l * RO + j];
            float acc = 0.f;
            for (int o = 0; o < 16; ++o) acc += rb(o, j);
            const float v = acc * (1.0f / 16.0f) + bsum / (float)L;
            if (j < S) logits[(size_t)n * S + j] = v;
            else gs[(size_t)n * 3 + (j - S - 1)] = v;
        } else {  // j == S, RO, RO+1 -> the three components of the vector read-out
            const int d = j == S ? 0 : (j - RO + 1);
            float bsum = 0.f;
            for (int l = 0; l < L; ++l) bsum += ro_b[l * RO + S];
            bsum /= (float)L;
            float acc = 0.f;
            for (int o = 0; o < 16; ++o) acc += (rb(o, S) + bsum) * ori[3 * o + d];
            eps[(size_t)n * 3 + d] = acc * (1.0f / 16.0f);
        }
    }
}
__global__ void pool_crystals_kernel(const float* __restrict__ gs, const int32_t* __restrict__ offsets, int B,
                                     float* __restrict__ len0) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 3 * B) return;
    const int b = idx / 3, g = idx - 3 * b;
    float acc = 0.f;
    for (int n = offsets[b]; n < offsets[b + 1]; ++n) acc += gs[(size_t)n * 3 + g];
    len0[idx] = acc;
}
// d rbar[(n,o)][j] from the gradient seeds
__global__ void train_outputs_backward_kernel(const float* __restrict__ g_eps, const float* __restrict__ g_logits,
                                              const float* __restrict__ g_len0, const int32_t* __restrict__ batch,
                                              const float* __restrict__ ori, int S, int N, float* __restrict__ drbar,
                                              int ld /* row pitch of drbar: S + 4 rounded up to a multiple of four, pad columns zero */) {
    const int row = blockIdx.x, n = row >> 4, o = row & 15, RO = S + 4;
    const int b = batch[n];
    for (int j = threadIdx.x; j < ld; j += blockDim.x) {
        float v;
        if (j >= RO) v = 0.f;
        else if (j < S) v = g_logits[(size_t)n * S + j] * (1.0f / 16.0f);
        else if (j == S)
            v = ((g_eps[3 * (size_t)n] * ori[3 * o] + g_eps[3 * (size_t)n + 1] * ori[3 * o + 1]) +
                 g_eps[3 * (size_t)n + 2] * ori[3 * o + 2]) * (1.0f / 16.0f);
        else v = g_len0[3 * (size_t)b + (j - S - 1)] * (1.0f / 16.0f);
        drbar[(size_t)row * ld + j] = v;
    }
}
// index, in the canonical monomial order of edge_rows_kernel, of the monomial that column `col` of
// PolynomialFeatures(3) over 6 attributes holds (embedding.py:10-14: x_i at i, x_i x_j at 6 + 6 i + j, x_i x_j x_k at
// 42 + 36 i + 6 j + k)
__device__ int mono_of_poly_column(int col) {
    int a[3], n;
    if (col < 6) { n = 1; a[0] = col; }
    else if (col < 42) { n = 2; a[0] = (col - 6) / 6; a[1] = (col - 6) % 6; }
    else { n = 3; a[0] = (col - 42) / 36; a[1] = ((col - 42) / 6) % 6; a[2] = (col - 42) % 6; }
    for (int x = 0; x < n; ++x)  // sort the multiset
        for (int y = x + 1; y < n; ++y)
            if (a[y] < a[x]) { const int t = a[x]; a[x] = a[y]; a[y] = t; }
    int p = 0, found = -1;  // index in the canonical monomial order (same enumeration as edge_rows_kernel)
    for (int i1 = 0; i1 < 6 && found < 0; ++i1, ++p)
        if (n == 1 && a[0] == i1) found = p;
    for (int i1 = 0; i1 < 6; ++i1)
        for (int j1 = i1; j1 < 6; ++j1, ++p)
            if (found < 0 && n == 2 && a[0] == i1 && a[1] == j1) found = p;
    for (int i1 = 0; i1 < 6; ++i1)
        for (int j1 = i1; j1 < 6; ++j1)
            for (int k1 = j1; k1 < 6; ++k1, ++p)
                if (found < 0 && n == 3 && a[0] == i1 && a[1] == j1 && a[2] == k1) found = p;
    return found;
}
// basis_fn.1.weight gradient [C][258] from the gradient of the folded weight [C][96]: every permutation column of a
// monomial receives that monomial's gradient
__global__ void unfold_poly_grad_kernel(const float* __restrict__ dw1f, int C, float* __restrict__ dw1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * ARREAU_POLY_COLS) return;
    const int c = i / ARREAU_POLY_COLS, col = i % ARREAU_POLY_COLS;
    dw1[i] = dw1f[c * ARREAU_MONO_PAD + mono_of_poly_column(col)];
}
// the forward direction: fold basis_fn.1.weight [C][258] onto the 83 monomials (columns of one monomial summed in
// column order), padding columns zero -- the device twin of fold_poly_weight in model.hip
// The columns of every monomial, ascending, -1 terminated (at most 3! = 6 permutations): built once per training context.
__global__ void mono_columns_kernel(int32_t* __restrict__ tab /*[ARREAU_MONO_PAD][8]*/) {
    const int mi = blockIdx.x * blockDim.x + threadIdx.x;
    if (mi >= ARREAU_MONO_PAD) return;
    int n = 0;
    if (mi < ARREAU_NUM_MONO)
        for (int col = 0; col < ARREAU_POLY_COLS; ++col)
            if (mono_of_poly_column(col) == mi && n < 8) tab[mi * 8 + n++] = col;
    for (; n < 8; ++n) tab[mi * 8 + n] = -1;
}
// a thread per (channel, monomial): its columns from the table, added in ascending order as before (round 4: scanning all 258 columns
// per output element was 28-34 us of the optimizer tail)
__global__ void fold_poly_weight_kernel(const float* __restrict__ w1, int C, const int32_t* __restrict__ tab, float* __restrict__ w1f) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * ARREAU_MONO_PAD) return;
    const int c = i / ARREAU_MONO_PAD, mi = i % ARREAU_MONO_PAD;
    float acc = 0.f;
    for (int q = 0; q < 8; ++q) {
        const int col = tab[mi * 8 + q];
        if (col < 0) break;
        acc += w1[c * ARREAU_POLY_COLS + col];
    }
    w1f[i] = acc;
}
// Weight refresh after an optimizer step: up to 24 device-to-device copies as ONE launch (they were 19 hipMemcpyAsync = 19
// blit-kernel launches, 85 us of the step's 330 us optimizer tail; blockIdx.y = segment).
struct CopySegments {
    float* dst[24];
    const float* src[24];
    unsigned n[24];
};
__global__ void copy_segments_kernel(CopySegments seg) {
    float* __restrict__ d = seg.dst[blockIdx.y];
    const float* __restrict__ s = seg.src[blockIdx.y];
    const unsigned n = seg.n[blockIdx.y];
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) d[i] = s[i];
}
__global__ void transpose_kernel(const float* __restrict__ in, int rows, int cols, float* __restrict__ out) {  // out[c][r]
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows * cols) out[(size_t)(i % cols) * rows + i / cols] = in[i];
}

inline unsigned blocks(long n, int per = 256) { return (unsigned)((n + per - 1) / per); }
}  // namespace

// ---------------------------------------------------------------------------------------------
// context: buffers of one training step, owned by the model, grown on demand
// ---------------------------------------------------------------------------------------------
struct arreau_train_ctx {
    int N = 0, B = 0, capN = 0, capB = 0;
    // arithmetic of the dense products (sgemm.h): 0 exact fp32 MFMA, 1 fp16x3, 2 bf16x6.  The training step runs its forward
    // products on fp16x3 (activations x weights: the sampling kernels' arithmetic) and every product with a gradient operand
    // on bf16x6 (full exponent range); the shape-general SAMPLING path stays on the exact kernel (it is the arithmetic
    // cross-check of the fused kernels).  ARREAU_TRAIN_GEMM=exact|split|fp16 (default split).
    int fwd_mode = 0, bwd_mode = 0;
    float* buf = nullptr;
    size_t buf_floats = 0;
    // plain row-major weights (in the model blob): [C][96], [D][C], [L][C][D], [L][H][C], [L][C][H], [L][S+4][C]
    const float *w1f, *w2, *wk, *lin1, *lin2, *ro_w;
    // forward state
    int32_t *batch, *deg, *src, *cell, *rev_start, *rev_cnt, *rev_idx, *mono_cols;
    float *lattice, *cart, *cvec, *dir, *dist;
    float *mono, *window, *h1pre, *h1, *h2pre, *kb, *fpoly, *fh1pre, *fh1, *fh2pre, *fkb, *F;
    float *x, *x1, *xhat, *rstd, *xn, *hpre, *h, *out, *fk, *rbar, *gs, *kern;
    // backward temporaries
    float *dx, *dxro, *rbar_all, *dfkb_all, *dtmp, *dh, *drbar, *dx1, *dkern, *dkb, *dh1, *dfk, *dfkb, *dfh1, *dw1f, *partial, *scratch_cols, *colpart;
    float* robias;  // [ROP] the one column sum of d(rbar) (every layer's read-out bias gradient)
    float *dxn_all, *dx2_all;  // [L][M][C]: d(LayerNorm output) and d(spherical conv output), for the batched bias / norm gradients
    float *xn_all, *dout_all, *dfk_all;  // [L][...]: LayerNorm outputs (forward), d(out) and d(fiber kernel) (backward), for the batched weight gradients
    int32_t* colcount;  // colsum_kernel's arrival counters (one per 64-column group; zero between launches)
    // The fiber branch (fiber basis MLP -> fiber kernels forward; their gradients backward) depends on the weights alone: a couple of
    // dozen launches over 256 rows, 4-6 us each on a handful of CUs.  They run on a second stream beside the edge-level products
    // (fork / join by events; ARREAU_TRAIN_SIDE_STREAM=0 keeps them in line) with their own split-K and column-sum scratch.
    // Measured (64 crystals, alternating on one box, 3 x 60 steps): 2.130 -> 2.108 ms per step -- 1 %, not the 0.19 ms the branch takes
    // in line: launches that share the chip slow each other (a 32-workgroup side launch beside a product of 1,024 workgroups, two per
    // CU, cost that product a third of a round: 30 -> 39 us in the kernel trace).
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_rev = nullptr;   // the reversed adjacency of the forward pass is complete (side stream; the backward pass waits for it)
    bool rev_pending = false;
    float *partial2 = nullptr, *colpart2 = nullptr;
    int32_t* colcount2 = nullptr;
    // Round 5: deferred reductions of the backward pass (main stream only): the k-slice sums of the weight-gradient products and the
    // chunk sums of the column-sum passes, each ONE launch at the end of arreau_train_backward (sgemm.h: arreau_sgemm_defer;
    // colsum4_final_multi_kernel).  Shared by the side stream's copy of this struct (pointers), never used through it.
    struct Deferred {
        arreau_sgemm_defer gemm;
        ColsumFinalList cols{};
        int ncols = 0, max_colblocks = 0;
        float* colscratch = nullptr;
        size_t colcap = 0, colused = 0;  // floats
    };
    Deferred* defer = nullptr;
    const int32_t *tstep, *offsets, *types;
    const float *frac, *lengths, *angles;
};

namespace {
struct Carve {
    float* base;
    size_t off = 0;
    template <typename T>
    T* take(size_t n) {
        off = (off + 63) & ~(size_t)63;
        T* r = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += (n * sizeof(T) + 3) / 4;
        return r;
    }
};
constexpr size_t PARTIAL_FLOATS = ARREAU_SGEMM_PARTIAL_FLOATS;

size_t layout(arreau_train_ctx& t, const arreau_model* m, int N, int B, float* base) {
    Carve c{base};
    const size_t C = m->C, D = m->D, L = m->L, H = m->H, k = m->k, S = m->S, RO = S + 4;
    const size_t R = (size_t)N * k * 16, M = (size_t)N * 16;
    t.w1f = m->t_w1f; t.w2 = m->t_w2; t.wk = m->t_wk; t.lin1 = m->t_lin1; t.lin2 = m->t_lin2; t.ro_w = m->t_ro_w;
    t.batch = c.take<int32_t>(N); t.deg = c.take<int32_t>(N); t.src = c.take<int32_t>(N * k); t.cell = c.take<int32_t>(N * k);
    t.rev_start = c.take<int32_t>(N); t.rev_cnt = c.take<int32_t>(N); t.rev_idx = c.take<int32_t>(N * k);
    t.mono_cols = c.take<int32_t>(ARREAU_MONO_PAD * 8);
    t.lattice = c.take<float>(B * 9); t.cart = c.take<float>(N * 3); t.cvec = c.take<float>(B * C);
    t.dir = c.take<float>(N * k * 3); t.dist = c.take<float>(N * k);
    t.mono = c.take<float>(R * ARREAU_MONO_PAD); t.window = c.take<float>(R);
    t.h1pre = c.take<float>(R * C); t.h1 = c.take<float>(R * C); t.h2pre = c.take<float>(R * D); t.kb = c.take<float>(R * D);
    t.fpoly = c.take<float>(256 * 3); t.fh1pre = c.take<float>(256 * C); t.fh1 = c.take<float>(256 * C);
    t.fh2pre = c.take<float>(256 * D); t.fkb = c.take<float>(256 * D); t.F = c.take<float>(M * (S + 78));
    t.x = c.take<float>((L + 1) * M * C); t.x1 = c.take<float>(L * M * C); t.xhat = c.take<float>(L * M * C);
    t.rstd = c.take<float>(L * M); t.xn = c.take<float>(M * C); t.hpre = c.take<float>(L * M * H); t.h = c.take<float>(L * M * H);
    t.out = c.take<float>(L * M * C); t.fk = c.take<float>(L * 256 * C); t.rbar = c.take<float>(M * RO); t.rbar_all = c.take<float>(L * M * RO); t.gs = c.take<float>(N * 3);
    t.kern = c.take<float>(R * L * C);   // all layers' spatial kernels, [R][L*C] (one GEMM: the basis is layer-independent)
    t.dx = c.take<float>(L * M * C);   // d x_l of every layer (round 5: kept per layer for the batched layer-scale / linear_2.bias column sums)
    t.dxro = c.take<float>(L * M * C); t.dtmp = c.take<float>(M * C); t.dh = c.take<float>(L * M * H); t.drbar = c.take<float>(M * ((RO + 3) & ~(size_t)3));
    t.xn_all = c.take<float>(L * M * C); t.dout_all = c.take<float>(L * M * C); t.dfk_all = c.take<float>(L * 256 * C);
    t.dxn_all = c.take<float>(L * M * C); t.dx2_all = c.take<float>(L * M * C);  // kept per layer for the batched weight gradients
    t.dx1 = c.take<float>(M * C); t.dkern = c.take<float>(R * L * C); t.dkb = c.take<float>(R * D); t.dh1 = c.take<float>(R * C);
    t.dfk = c.take<float>(256 * C); t.dfkb = c.take<float>(256 * D); t.dfkb_all = c.take<float>(L * 256 * D); t.dfh1 = c.take<float>(256 * C);
    t.dw1f = c.take<float>(C * ARREAU_MONO_PAD); t.partial = c.take<float>(PARTIAL_FLOATS); t.scratch_cols = c.take<float>(1024); t.robias = c.take<float>(1024);
    t.colpart = c.take<float>((size_t)2 * COLSUM4_MAX_BATCH * COLSUM4_MAX_CHUNKS * 1024);  // (two results per pass, up to eight matrices per call)
    t.colcount = c.take<int32_t>(COLCOUNT_INTS);
    t.partial2 = c.take<float>(PARTIAL_FLOATS);
    t.colpart2 = c.take<float>((size_t)2 * COLSUM4_MAX_BATCH * COLSUM4_MAX_CHUNKS * 1024);
    t.colcount2 = c.take<int32_t>(COLCOUNT_INTS);
    {   // deferred reductions: room for every weight gradient's k-slices at once (each product is capped at PARTIAL_FLOATS) and for
        // the chunk rows of every column-sum pass of a backward pass; a request that does not fit runs its reduction at once
        float* gs = c.take<float>(3 * PARTIAL_FLOATS);
        const size_t colcap = (size_t)2 * COLSUM4_MAX_BATCH * COLSUM4_MAX_CHUNKS * 1024;
        float* cs = c.take<float>(colcap);
        if (t.defer) {
            t.defer->gemm.scratch = gs; t.defer->gemm.cap = 3 * PARTIAL_FLOATS;
            t.defer->colscratch = cs; t.defer->colcap = colcap;
        }
    }
    return c.off;
}

// Round 5: launch merges of the training step (conv + mix forward, LayerNorm + mix backward, both conv gradients in one launch,
// column-sum and split-K reductions deferred to one launch each at the end of the backward pass).  ARREAU_TRAIN_FUSE=0 restores the
// one-kernel-per-operation sequence (same arithmetic per element: tests compare the two bit for bit).
inline bool train_fuse_on() {
    const char* e = getenv("ARREAU_TRAIN_FUSE");  // read per call (A/B runs, tests): a few dozen getenv per step
    return !e || atoi(e) != 0;
}
typedef arreau_sgemm_detail::SgemmEpilogue Epi;
int gemm(hipStream_t s, arreau_train_ctx& t, int mode, int M, int N, int K, const float* A, long as0, long as1, const float* B, long bs0,
         long bs1, float* C, int ldc, float alpha = 1.f, float beta = 0.f, const Epi* epi = nullptr, bool* fused = nullptr) {
    return arreau_sgemm(s, t.partial, M, N, K, A, as0, as1, B, bs0, bs1, C, ldc, alpha, beta, 1, 0, 0, 0, mode, epi, fused);
}
// Y[rows][out] = X[rows][in] . W[out][in]^T
int linear(hipStream_t s, arreau_train_ctx& t, long rows, int in, int out, const float* X, const float* W, float* Y,
           float alpha = 1.f, float beta = 0.f, const Epi* epi = nullptr, bool* fused = nullptr) {
    return gemm(s, t, t.fwd_mode, (int)rows, out, in, X, in, 1, W, 1, in, Y, out, alpha, beta, epi, fused);
}
// dX[rows][in] (+)= dY[rows][out] . W[out][in]
int linear_dx(hipStream_t s, arreau_train_ctx& t, long rows, int in, int out, const float* dY, const float* W, float* dX,
              float alpha = 1.f, float beta = 0.f, const Epi* epi = nullptr, bool* fused = nullptr) {
    return gemm(s, t, t.bwd_mode, (int)rows, in, out, dY, out, 1, W, in, 1, dX, in, alpha, beta, epi, fused);
}
// the same for `batch` layers in one launch (dY / X / dW of consecutive layers dy_bs / x_bs / out * in floats apart; 0 = shared)
int linear_dw_batched(hipStream_t s, arreau_train_ctx& t, int batch, long rows, int in, int out, const float* dY, long dy_bs,
                      const float* X, long x_bs, float* dW, float alpha = 1.f, bool defer = false, int dy_ld = 0 /* row pitch of dY (0: out) */) {
    return arreau_sgemm(s, t.partial, out, in, (int)rows, dY, 1, dy_ld ? dy_ld : out, X, in, 1, dW, in, alpha, 0.f, batch, dy_bs, x_bs, (long)out * in, t.bwd_mode,
                        nullptr, nullptr, defer && train_fuse_on() && t.defer ? &t.defer->gemm : nullptr);
}
// dW[out][in] = alpha * dY[rows][out]^T . X[rows][in]
int linear_dw(hipStream_t s, arreau_train_ctx& t, long rows, int in, int out, const float* dY, const float* X, float* dW,
              float alpha = 1.f, bool defer = false) {
    return linear_dw_batched(s, t, 1, rows, in, out, dY, 0, X, 0, dW, alpha, defer);
}
// `batch` > 1: the same sums for `batch` matrices (a / b a_bs / b_bs floats apart, results out_bs / out2_bs apart) in the two
// launches of one -- the bias gradients of the L layers (16-byte columns only).
int colsum(hipStream_t s, arreau_train_ctx& t, const float* a, const float* b, long rows, int cols, float scale, float* out,
           int accumulate = 0, float* out2 = nullptr, const float* colscale2 = nullptr, int batch = 1, long a_bs = 0, long b_bs = 0,
           long out_bs = 0, long out2_bs = 0, float* scaled_out = nullptr /* also a * colscale_out (16-byte path, one matrix) */,
           const float* colscale_out = nullptr, bool* wrote_scaled = nullptr,
           bool defer = false /* the sums may wait for flush_deferred (main stream of the backward pass only) */, long colscale2_bs = 0,
           const float* gelu_pre = nullptr /* 16-byte path, one matrix: a *= gelu'(gelu_pre) * gelu_rowscale[row] in place first */,
           const float* gelu_rowscale = nullptr,
           ColsumGather* gather = nullptr /* deferred passes over the same rows: collected here, launched together by launch_gathered */) {
    if (wrote_scaled) *wrote_scaled = false;
    if (cols > 1024) {
        arreau_set_error("colsum: more than 1024 columns");
        return ARREAU_EINVAL;
    }
    const bool path16 = cols % 4 == 0 && (size_t)a % 16 == 0 && (b == nullptr || (size_t)b % 16 == 0) && a_bs % 4 == 0 && b_bs % 4 == 0;
    if (gelu_pre && !(path16 && batch == 1 && (size_t)gelu_pre % 16 == 0)) {
        arreau_set_error("colsum: the GELU-backward form needs 16-byte columns");
        return ARREAU_EINVAL;
    }
    if (path16) {
        const int chunks = (int)std::min<long>(COLSUM4_MAX_CHUNKS, std::max<long>(1, rows / 16));
        const long rpc = (rows + chunks - 1) / chunks;
        if (batch > COLSUM4_MAX_BATCH) {
            arreau_set_error("colsum: batch beyond the partial-sum scratch");
            return ARREAU_EINVAL;
        }
        f32x4* part = reinterpret_cast<f32x4*>(t.colpart);
        f32x4* part2 = b && out2 ? part + (size_t)COLSUM4_MAX_BATCH * COLSUM4_MAX_CHUNKS * 256 : nullptr;
        const bool dual = b && out2;
        const size_t need = (size_t)batch * chunks * cols * (dual ? 2 : 1);  // floats
        arreau_train_ctx::Deferred* df = defer && train_fuse_on() ? t.defer : nullptr;
        const bool deferred = df && df->colscratch && df->colused + need <= df->colcap && df->ncols + batch <= COLSUM_DEFER_MAX && (!out2 || b);
        if (deferred) {
            part = reinterpret_cast<f32x4*>(df->colscratch + df->colused);
            part2 = dual ? part + (size_t)batch * chunks * (cols / 4) : nullptr;
            df->colused += (need + 63) & ~(size_t)63;
            for (int i = 0; i < batch; ++i) {
                ColsumFinalDesc& d = df->cols.d[df->ncols++];
                d.part = part + (size_t)i * chunks * (cols / 4);
                d.part2 = dual ? part2 + (size_t)i * chunks * (cols / 4) : nullptr;
                d.out = out + (long)i * out_bs;
                d.out2 = dual ? out2 + (long)i * out2_bs : nullptr;
                d.colscale2 = colscale2 ? colscale2 + (long)i * colscale2_bs : nullptr;
                d.chunks = chunks; d.cols4 = cols / 4; d.scale = scale; d.accumulate = accumulate;
            }
            df->max_colblocks = std::max(df->max_colblocks, (cols / 4 + 7) / 8);
        }
        if (deferred && gather && !gelu_pre && !scaled_out && gather->n + batch <= COLSUM_PARTIAL_MULTI_MAX &&
            (gather->n == 0 || (gather->rows == rows && gather->chunks == chunks))) {
            gather->rows = rows; gather->chunks = chunks; gather->rpc = rpc;
            for (int i = 0; i < batch; ++i) {
                ColsumPartialDesc& d = gather->list.d[gather->n++];
                d.a = reinterpret_cast<const f32x4*>(a) + (long)i * (a_bs / 4);
                d.b = b ? reinterpret_cast<const f32x4*>(b) + (long)i * (b_bs / 4) : nullptr;
                d.part = part + (size_t)i * chunks * (cols / 4);
                d.part2 = part2 ? part2 + (size_t)i * chunks * (cols / 4) : nullptr;
                d.cols4 = cols / 4;
            }
            return ARREAU_OK;
        }
        if (gelu_pre)
            hipLaunchKernelGGL(colsum4_partial_kernel<true>, dim3(chunks, batch), dim3(256), 0, s, reinterpret_cast<f32x4*>(const_cast<float*>(a)),
                               reinterpret_cast<const f32x4*>(b), rows, cols / 4, rpc, part, part2, a_bs / 4, b_bs / 4, (f32x4*)nullptr,
                               (const f32x4*)nullptr, reinterpret_cast<const f32x4*>(gelu_pre), gelu_rowscale);
        else
        hipLaunchKernelGGL(colsum4_partial_kernel<false>, dim3(chunks, batch), dim3(256), 0, s, reinterpret_cast<const f32x4*>(a),
                           reinterpret_cast<const f32x4*>(b), rows, cols / 4, rpc, part, part2, a_bs / 4, b_bs / 4,
                           batch == 1 ? reinterpret_cast<f32x4*>(scaled_out) : (f32x4*)nullptr, reinterpret_cast<const f32x4*>(colscale_out),
                           (const f32x4*)nullptr, (const float*)nullptr);
        if (wrote_scaled) *wrote_scaled = batch == 1 && scaled_out != nullptr;
        ARREAU_CHECK_HIP(hipGetLastError());
        if (deferred) return ARREAU_OK;
        hipLaunchKernelGGL(colsum4_final_kernel, dim3((cols / 4 + 7) / 8, batch), dim3(256), 0, s, part, part2, chunks, cols / 4, scale,
                           accumulate, out, part2 ? out2 : nullptr, colscale2, out_bs, out2_bs);
        ARREAU_CHECK_HIP(hipGetLastError());
        if (out2 && !b) {
            arreau_set_error("colsum: a second result needs a second operand");
            return ARREAU_EINVAL;
        }
        return ARREAU_OK;
    }
    if (batch > 1) {  // element-wise path (columns not a multiple of four): one matrix after the other
        for (int i = 0; i < batch; ++i) {
            const int rc = colsum(s, t, a + (long)i * a_bs, b ? b + (long)i * b_bs : nullptr, rows, cols, scale, out + (long)i * out_bs, accumulate,
                                  out2 ? out2 + (long)i * out2_bs : nullptr, colscale2);
            if (rc) return rc;
        }
        return ARREAU_OK;
    }
    const int chunks = (int)std::min<long>(COLSUM_MAX_CHUNKS, std::max<long>(1, rows / 128));
    const long rpc = (rows + chunks - 1) / chunks;
    hipLaunchKernelGGL(colsum_kernel, dim3((cols + 63) / 64, chunks), dim3(256), 0, s, a, b, rows, cols, rpc, t.colpart,
                       t.colcount, scale, accumulate, out);
    ARREAU_CHECK_HIP(hipGetLastError());
    if (out2) {  // element-wise path: the plain sum as a second pass
        hipLaunchKernelGGL(colsum_kernel, dim3((cols + 63) / 64, chunks), dim3(256), 0, s, a, (const float*)nullptr, rows, cols, rpc,
                           t.colpart, t.colcount, 1.0f, 0, out2);
        ARREAU_CHECK_HIP(hipGetLastError());
        if (colscale2) {
            hipLaunchKernelGGL(scale_cols_kernel, dim3(1 + cols / 256), dim3(256), 0, s, out2, colscale2, 1L, cols, out2);
            ARREAU_CHECK_HIP(hipGetLastError());
        }
    }
    return ARREAU_OK;
}
int launch_bias_gelu(hipStream_t s, float* pre, const float* bias, const float* rowscale, long rows, int cols, float* act) {
    if (rows <= 0) return ARREAU_OK;
    if (cols % 4 == 0 && ((size_t)pre | (size_t)bias | (size_t)act) % 16 == 0) {
        const long n4 = rows * (cols / 4);
        hipLaunchKernelGGL(bias_gelu_kernel4, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, reinterpret_cast<f32x4*>(pre),
                           reinterpret_cast<const f32x4*>(bias), rowscale, rows, cols / 4, reinterpret_cast<f32x4*>(act));
    } else {
        hipLaunchKernelGGL(bias_gelu_kernel, dim3((unsigned)((rows * cols + 255) / 256)), dim3(256), 0, s, pre, bias, rowscale, rows,
                           cols, act);
    }
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}
int launch_gelu_backward(hipStream_t s, float* g, const float* pre, const float* rowscale, long rows, int cols) {
    if (rows <= 0) return ARREAU_OK;
    if (cols % 4 == 0 && ((size_t)g | (size_t)pre) % 16 == 0) {
        const long n4 = rows * (cols / 4);
        hipLaunchKernelGGL(gelu_backward_kernel4, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, reinterpret_cast<f32x4*>(g),
                           reinterpret_cast<const f32x4*>(pre), rowscale, rows, cols / 4);
    } else {
        hipLaunchKernelGGL(gelu_backward_kernel, dim3((unsigned)((rows * cols + 255) / 256)), dim3(256), 0, s, g, pre, rowscale, rows,
                           cols);
    }
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}
// pre = X W^T + bias, act = gelu(pre) * rowscale: inside the product where the split kernel takes it, else as the element-wise launch
int linear_bias_gelu(hipStream_t s, arreau_train_ctx& t, long rows, int in, int out, const float* X, const float* W, const float* bias,
                     const float* rowscale, float* pre, float* act) {
    Epi e;
    e.kind = 1; e.vec = bias; e.row = rowscale; e.out = act;
    e.prefer_small = train_fuse_on();
    bool fused = false;
    int rc = linear(s, t, rows, in, out, X, W, pre, 1.f, 0.f, &e, &fused);
    if (rc || fused) return rc;
    return launch_bias_gelu(s, pre, bias, rowscale, rows, out, act);
}
// dX = (dY W) * gelu'(pre) * rowscale
int linear_dx_gelu_backward(hipStream_t s, arreau_train_ctx& t, long rows, int in, int out, const float* dY, const float* W, const float* pre,
                            const float* rowscale, float* dX) {
    Epi e;
    e.kind = 2; e.mat = pre; e.row = rowscale;
    bool fused = false;
    int rc = linear_dx(s, t, rows, in, out, dY, W, dX, 1.f, 0.f, &e, &fused);
    if (rc || fused) return rc;
    return launch_gelu_backward(s, dX, pre, rowscale, rows, in);
}
// out[m][n] = sum over z = 0 .. Z - 1, in that order, of part[z][m][n] (a running sum from zero: the association of Z accumulating launches)
int ordered_sum(hipStream_t s, const float* part, int Z, int M, int N, float* out) {
    hipLaunchKernelGGL(arreau_sgemm_detail::splitk_reduce_kernel<1>, dim3((unsigned)(((long)M * N + 255) / 256), 1), dim3(256), 0, s, part, Z, M, N, out, N,
                       1.0f, 0.0f, 0L);
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}
// the deferred reductions of a backward pass (arreau_train_ctx::Deferred): one launch for the column sums, one for the k-slice sums
void reset_deferred(arreau_train_ctx& t) {
    if (!t.defer) return;
    arreau_sgemm_defer_reset(t.defer->gemm);
    t.defer->ncols = 0; t.defer->max_colblocks = 0; t.defer->colused = 0;
}
int flush_deferred_colsums(hipStream_t s, arreau_train_ctx& t) {
    arreau_train_ctx::Deferred* df = t.defer;
    if (df && df->ncols > 0) {
        hipLaunchKernelGGL(colsum4_final_multi_kernel, dim3(df->max_colblocks, df->ncols), dim3(256), 0, s, df->cols);
        ARREAU_CHECK_HIP(hipGetLastError());
    }
    if (df) { df->ncols = 0; df->max_colblocks = 0; df->colused = 0; }
    return ARREAU_OK;
}
int launch_gathered(hipStream_t s, ColsumGather& g) {
    if (g.n > 0) {
        hipLaunchKernelGGL(colsum4_partial_multi_kernel, dim3(g.chunks, g.n), dim3(256), 0, s, g.list, g.rows, g.rpc);
        ARREAU_CHECK_HIP(hipGetLastError());
    }
    g.n = 0;
    return ARREAU_OK;
}
int flush_deferred_gemms(hipStream_t s, arreau_train_ctx& t) {
    return t.defer ? arreau_sgemm_flush(s, t.defer->gemm) : ARREAU_OK;
}
#define V4(p) reinterpret_cast<const f32x4*>(p)
#define V4W(p) reinterpret_cast<f32x4*>(p)
#define TRY(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)
#define LAUNCH(kernel, grid, block, ...)                                  \
    do {                                                                  \
        hipLaunchKernelGGL(kernel, grid, block, 0, s, __VA_ARGS__);       \
        ARREAU_CHECK_HIP(hipGetLastError());                              \
    } while (0)
}  // namespace

void arreau_train_ctx_destroy(arreau_train_ctx* t) {
    if (!t) return;
    if (t->buf) (void)hipFree(t->buf);
    if (t->side) (void)hipStreamDestroy(t->side);
    if (t->ev_fork) (void)hipEventDestroy(t->ev_fork);
    if (t->ev_join) (void)hipEventDestroy(t->ev_join);
    if (t->ev_rev) (void)hipEventDestroy(t->ev_rev);
    delete t->defer;
    delete t;
}

static int ensure_ctx(arreau_model* m, int N, int B, hipStream_t s) {
    arreau_train_ctx* t = m->train;
    if (t && N <= t->capN && B <= t->capB) {
        layout(*t, m, t->capN, t->capB, t->buf);
        t->N = N; t->B = B;
        return ARREAU_OK;
    }
    // Growing means a device synchronisation, a free and an allocation of gigabytes (milliseconds): a training loop whose
    // batches creep upwards in size must not pay that every few steps, so a regrown context gets 25 % headroom.
    int capN = N, capB = B;
    if (t) {
        capN = std::max(N, t->capN) + std::max(N, t->capN) / 4;
        capB = std::max(B, t->capB) + std::max(B, t->capB) / 4;
        ARREAU_CHECK_HIP(hipStreamSynchronize(s));
        arreau_train_ctx_destroy(t);
        m->train = nullptr;
        // a cached step graph of arreau_sample_loop (general path) points into the block just freed: never replay it
        memset(m->graph_key, 0, sizeof(m->graph_key));
    }
    t = new arreau_train_ctx();
    t->defer = new arreau_train_ctx::Deferred();
    t->capN = capN; t->capB = capB;
    arreau_train_ctx probe;
    t->buf_floats = layout(probe, m, capN, capB, nullptr);
    hipError_t e = hipMalloc((void**)&t->buf, t->buf_floats * sizeof(float));
    if (e != hipSuccess) {
        delete t->defer;
        delete t;
        arreau_set_error(std::string("hipMalloc(training buffers): ") + hipGetErrorString(e));
        return ARREAU_EHIP;
    }
    layout(*t, m, capN, capB, t->buf);
    t->N = N; t->B = B;
    m->train = t;
    ARREAU_CHECK_HIP(hipMemsetAsync(t->scratch_cols, 0, 1024 * sizeof(float), s));
    ARREAU_CHECK_HIP(hipMemsetAsync(t->colcount, 0, COLCOUNT_INTS * sizeof(int32_t), s));
    ARREAU_CHECK_HIP(hipMemsetAsync(t->colcount2, 0, COLCOUNT_INTS * sizeof(int32_t), s));
    {
        // (read per context, i.e. per model: a test can build the one-stream form beside the default one in one process)
        const bool side_on = [] { const char* e = getenv("ARREAU_TRAIN_SIDE_STREAM"); return !e || atoi(e) != 0; }();
        if (side_on) {
            // (default priority: a lowest-priority side stream measured the same to slightly worse, tools/exp/ab_side_stream.sh)
            ARREAU_CHECK_HIP(hipStreamCreateWithFlags(&t->side, hipStreamNonBlocking));
            ARREAU_CHECK_HIP(hipEventCreateWithFlags(&t->ev_fork, hipEventDisableTiming));
            ARREAU_CHECK_HIP(hipEventCreateWithFlags(&t->ev_join, hipEventDisableTiming));
            ARREAU_CHECK_HIP(hipEventCreateWithFlags(&t->ev_rev, hipEventDisableTiming));
        }
    }
    hipLaunchKernelGGL(mono_columns_kernel, dim3(1), dim3(128), 0, s, t->mono_cols);
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}

// Side stream of the fiber branch.  fork: everything queued on `main_stream` so far happens before the side work; `ts` becomes a view
// of the context with the side stream's own scratch, `s` the stream to launch on (the main stream itself when the side stream is
// off: then this is a no-op and the work stays in line).  join: the main stream waits for the side work recorded in ev_join.
static int fork_side(arreau_train_ctx& t, hipStream_t main_stream, arreau_train_ctx& ts, hipStream_t& s) {
    ts = t;
    s = main_stream;
    if (!t.side) return ARREAU_OK;
    ARREAU_CHECK_HIP(hipEventRecord(t.ev_fork, main_stream));
    ARREAU_CHECK_HIP(hipStreamWaitEvent(t.side, t.ev_fork, 0));
    ts.partial = t.partial2; ts.colpart = t.colpart2; ts.colcount = t.colcount2;
    s = t.side;
    return ARREAU_OK;
}
static int join_side(arreau_train_ctx& t, hipStream_t main_stream) {
    if (!t.side) return ARREAU_OK;
    ARREAU_CHECK_HIP(hipStreamWaitEvent(main_stream, t.ev_join, 0));
    return ARREAU_OK;
}

// The network from the edge basis to the read-outs, on graph arrays and layer-0 features supplied by the caller (t.x).
int arreau_general_network(arreau_model* m, const arreau_graph_view& g, const int32_t* d_off, int B, int N, float* d_eps,
                           float* d_logits, float* d_len0, hipStream_t s) {
    ARREAU_REQUIRE(m->train && m->train->capN >= N && m->train->capB >= B, "arreau_general_network: context not prepared");
    arreau_train_ctx& t = *m->train;
    const int C = m->C, D = m->D, L = m->L, H = m->H, k = m->k, S = m->S, RO = S + 4;
    const long R = (long)N * k * 16, M = (long)N * 16;
    if (N == 0) return ARREAU_OK;
    // edge basis: kb = gelu(W2 gelu(W1 poly + b1) + b2) * window   (ponita.py:65,94)
    auto edge_basis = [&]() -> int {
        LAUNCH(edge_rows_kernel, dim3(blocks(R)), dim3(256), g.dir, g.dist, g.deg, g.batch, g.lattice, m->ori, m->cfg.radius, N, k,
               t.mono, t.window);
        if (train_fuse_on()) {   // (round 5: bias + GELU inside the product on its 64 x 64 tiles, as layer 2 and the ConvNext block had them)
            TRY(linear_bias_gelu(s, t, R, ARREAU_MONO_PAD, C, t.mono, t.w1f, m->b1, (const float*)nullptr, t.h1pre, t.h1));
        } else {
            TRY(linear(s, t, R, ARREAU_MONO_PAD, C, t.mono, t.w1f, t.h1pre));
            TRY(launch_bias_gelu(s, t.h1pre, m->b1, (const float*)nullptr, R, C, t.h1));
        }
        TRY(linear_bias_gelu(s, t, R, C, D, t.h1, t.w2, m->b2, (const float*)t.window, t.h2pre, t.kb));
        // kernel_l = kb . Wk_l^T for all layers in one product (conv.py:110; conv.kernel.weight stacked [L*C][D])
        TRY(linear(s, t, R, D, L * C, t.kb, t.wk, t.kern));
        return ARREAU_OK;
    };
    {   // fiber basis (ponita.py:66,95) and the fiber kernels of all layers, fk_l = fkb . Wfk_l^T (conv.py:113-116; one batched product):
        // functions of the weights alone -- on the side stream, beside the edge basis
        hipStream_t main_stream = s;
        arreau_train_ctx ts = t;
        hipStream_t s = main_stream;   // (LAUNCH and the helpers below take the stream by this name)
        TRY(fork_side(t, main_stream, ts, s));
        // (the main stream's products are handed to the driver before the side branch's small launches: no difference in a free-running
        // loop -- the host is far ahead -- but under a tracer, whose launches cost more, the main stream no longer sits idle here)
        TRY(edge_basis());
        LAUNCH(fiber_poly_kernel, dim3(1), dim3(256), m->ori, ts.fpoly);
        TRY(linear(s, ts, 256, 3, C, ts.fpoly, m->fiber_w1, ts.fh1pre));
        TRY(launch_bias_gelu(s, ts.fh1pre, m->fiber_b1, (const float*)nullptr, 256L, C, ts.fh1));
        TRY(linear_bias_gelu(s, ts, 256, C, D, ts.fh1, m->fiber_w2, m->fiber_b2, (const float*)nullptr, ts.fh2pre, ts.fkb));
        TRY(arreau_sgemm(s, ts.partial, 256, C, D, ts.fkb, D, 1, m->fiber_wk, 1, D, ts.fk, C, 1.f, 0.f, L, 0, (long)C * D, 256L * C, ts.fwd_mode));
        if (t.side) ARREAU_CHECK_HIP(hipEventRecord(t.ev_join, t.side));
    }
    TRY(join_side(t, s));   // the fiber kernels: first used by the layer loop below
    for (int l = 0; l < L; ++l) {
        const float* xl = t.x + (size_t)l * M * C;
        float* xnext = t.x + (size_t)(l + 1) * M * C;
        float* x1 = t.x1 + (size_t)l * M * C;
        float* fk = t.fk + (size_t)l * 256 * C;
        if (t.fwd_mode == 1 && arreau_mlp_train_forward_available(m) && train_fuse_on() && k == 8) {
            // round 5: spatial conv + spherical mix + LayerNorm + linear_1 + GELU + linear_2 + layer scale + residual as ONE launch of the
            // sampling step's one-node-per-workgroup kernel (node_f16m.hip, FUSE + TRAIN), which writes everything the backward pass reads
            TRY(arreau_launch_mlp_train_forward(m, l, nullptr, xl, xnext, t.xhat + (size_t)l * M * C, t.rstd + (size_t)l * M,
                                                t.xn_all + (size_t)l * M * C, t.hpre + (size_t)l * M * H, t.h + (size_t)l * M * H,
                                                t.out + (size_t)l * M * C, N, s, t.kern + (size_t)l * C, L * C, g.deg, g.src, fk, x1));
            continue;
        }
        if (C % 4 == 0 && train_fuse_on() && C <= 2048) {
            hipLaunchKernelGGL(conv_mix_forward_kernel4, dim3((unsigned)N), dim3(512), (size_t)16 * C * sizeof(float), s, V4(t.kern + (size_t)l * C),
                               L * C / 4, V4(xl), g.deg, g.src, N, k, C / 4, V4(fk), V4(m->conv_bias + (size_t)l * C), V4W(x1), V4W(t.dtmp));
            ARREAU_CHECK_HIP(hipGetLastError());
        } else if (C % 4 == 0) {
            LAUNCH(conv_forward_kernel4, dim3(blocks(M * C / 4)), dim3(256), V4(t.kern + (size_t)l * C), L * C / 4, V4(xl), g.deg, g.src, N, k,
                   C / 4, V4W(x1));
            LAUNCH(mix_forward_kernel4, dim3(blocks(M * C / 4)), dim3(256), V4(x1), V4(fk), V4(m->conv_bias + (size_t)l * C), N, C / 4, V4W(t.dtmp));
        } else {
            LAUNCH(conv_forward_kernel, dim3(blocks(M * C)), dim3(256), t.kern + (size_t)l * C, L * C, xl, g.deg, g.src, N, k, C, x1);
            LAUNCH(mix_forward_kernel, dim3(blocks(M * C)), dim3(256), x1, fk, m->conv_bias + (size_t)l * C, N, C, t.dtmp);
        }
        float* hpre = t.hpre + (size_t)l * M * H;
        float* h = t.h + (size_t)l * M * H;
        float* out = t.out + (size_t)l * M * C;
        if (t.fwd_mode == 1 && arreau_mlp_train_forward_available(m)) {
            // LayerNorm + linear_1 + GELU + linear_2 + layer scale + residual as ONE launch of the sampling step's kernel, which also
            // writes what the backward pass reads (node_f16m.hip, TRAIN): 3 launches per layer instead of 5
            TRY(arreau_launch_mlp_train_forward(m, l, t.dtmp, xl, xnext, t.xhat + (size_t)l * M * C, t.rstd + (size_t)l * M,
                                                t.xn_all + (size_t)l * M * C, hpre, h, out, N, s));
            continue;
        }
        LAUNCH(ln_forward_kernel, dim3(blocks(M, 4)), dim3(256), t.dtmp, m->ln_w + (size_t)l * C, m->ln_b + (size_t)l * C, M, C,
               t.xhat + (size_t)l * M * C, t.rstd + (size_t)l * M, t.xn_all + (size_t)l * M * C);
        TRY(linear_bias_gelu(s, t, M, C, H, t.xn_all + (size_t)l * M * C, t.lin1 + (size_t)l * H * C, m->mb1 + (size_t)l * H, (const float*)nullptr,
                             hpre, h));
        {   // out = h W2^T + b2;  x_{l+1} = out * layer_scale + x_l  (in the product's epilogue where the split kernel runs it)
            Epi e;
            e.kind = 3; e.vec = m->mb2 + (size_t)l * C; e.vec2 = m->ls + (size_t)l * C; e.mat = xl; e.out = xnext;
            bool fused = false;
            TRY(linear(s, t, M, H, C, h, t.lin2 + (size_t)l * C * H, out, 1.f, 0.f, &e, &fused));
            if (!fused)
                LAUNCH(bias_scale_residual_kernel, dim3(blocks(M * C)), dim3(256), out, m->mb2 + (size_t)l * C, m->ls + (size_t)l * C, xl, M, C, xnext);
        }
    }
    // read-outs of all layers, averaged (ponita.py:105,108; biases are added in train_outputs_kernel): ONE batched product over the kept
    // x_1 .. x_L and a sum in layer order -- the same products and the same association as L accumulating launches inside the loop
    TRY(arreau_sgemm(s, t.partial, (int)M, RO, C, t.x + (size_t)M * C, C, 1, t.ro_w, 1, C, t.rbar_all, RO, 1.0f / (float)L, 0.f, L, (long)M * C,
                     (long)RO * C, (long)M * RO, t.fwd_mode));
    if (train_fuse_on()) {
        LAUNCH(train_outputs_kernel, dim3(N), dim3(128), t.rbar_all, m->ro_b, m->ori, S, L, N, d_eps, d_logits, t.gs, L);
    } else {
        TRY(ordered_sum(s, t.rbar_all, L, (int)M, RO, t.rbar));
        LAUNCH(train_outputs_kernel, dim3(N), dim3(128), t.rbar, m->ro_b, m->ori, S, L, N, d_eps, d_logits, t.gs, 0);
    }
    LAUNCH(pool_crystals_kernel, dim3(blocks(3 * B, 128)), dim3(128), t.gs, d_off, B, d_len0);
    m->ran_edge = m->ran_mlp = ARREAU_VARIANT_GENERAL;
    m->ran_conv = ARREAU_VARIANT_GENERAL;
    return ARREAU_OK;
}

float* arreau_general_x0(arreau_model* m, int N, int B, hipStream_t s) {
    if (ensure_ctx(m, N, B, s) != ARREAU_OK) return nullptr;
    m->train->fwd_mode = 0;  // sampling through the shape-general path: exact fp32 products
    return m->train->x;
}

extern "C" int arreau_train_forward(arreau_model* m, const float* d_frac, const int32_t* d_types, const float* d_lengths,
                                    const float* d_angles, const int32_t* d_t, const int32_t* d_off, int32_t B, int32_t N,
                                    float* d_eps, float* d_logits, float* d_len0, void* stream) {
    ARREAU_REQUIRE(m && d_frac && d_types && d_lengths && d_angles && d_t && d_off && d_eps && d_logits && d_len0,
                   "arreau_train_forward: null pointer");
    ARREAU_REQUIRE(B >= 1 && N >= 1, "arreau_train_forward: bad size");
    hipStream_t s = (hipStream_t)stream;
    TRY(ensure_ctx(m, N, B, s));
    arreau_train_ctx& t = *m->train;
    const int C = m->C, k = m->k, S = m->S;
    const long M = (long)N * 16;
    t.tstep = d_t; t.offsets = d_off; t.types = d_types; t.frac = d_frac; t.lengths = d_lengths; t.angles = d_angles;
    {
        static const int env = [] {
            const char* e = getenv("ARREAU_TRAIN_GEMM");
            return !e || strcmp(e, "split") == 0 ? 1 : strcmp(e, "fp16") == 0 ? 2 : 0;
        }();
        // (fp16x3 while the weights fit fp16 and the caller has not asked for the full-range kernels -- arreau_model_set_variant(-1 or 3, 1),
        // what PONITA_DIFFUSION.training_step does after a non-finite step: the operand bounds of arreau_model_create do not follow
        // the optimizer)
        t.fwd_mode = env == 0 ? 0 : (m->f16_ok && !m->train_full_range ? 1 : 2);
        t.bwd_mode = env == 0 ? 0 : (env == 2 && m->f16_ok ? 1 : 2);
    }
    const bool side_setup = t.side != nullptr && train_fuse_on();
    // (a previous forward's reversed adjacency may still be in flight on the side stream: the neighbour list below rewrites its input)
    if (t.rev_pending) { ARREAU_CHECK_HIP(hipStreamWaitEvent(s, t.ev_rev, 0)); t.rev_pending = false; }
    // geometry and graph: the sampling path's own kernels (prep, neighbour list)
    TRY(arreau_launch_prep(m, d_frac, d_lengths, d_angles, d_t, d_off, B, N, t.lattice, t.cart, t.batch, t.cvec, s));
    TRY(arreau_launch_neighbor(t.cart, t.lattice, d_off, t.batch, B, N, m->cfg.radius, k, t.deg, t.src, t.cell, t.dir, t.dist, s));
    // embedding (ponita.py:98): x_0 = F . W_emb^T, embT = W_emb^T [S+78][C]  (F is kept for the embedder's gradient).  Round 5: on the
    // side stream, in front of the network's fiber branch and beside its edge-level products -- x_0 is first read by the layer loop,
    // behind arreau_general_network's join.  (It needs prep's outputs only, but is NOT started beside the neighbour list: kernels of two
    // streams sharing a CU is where round 2 saw a receiver's neighbour list lose a candidate -- DESIGN.md section 8, cause unknown --
    // and a wrong edge is a different graph, not a rounding difference.)
    {
        arreau_train_ctx ts = t;
        hipStream_t ss = s;
        if (side_setup) TRY(fork_side(t, s, ts, ss));
        hipLaunchKernelGGL(features_kernel, dim3((unsigned)M), dim3(64), 0, ss, d_frac, d_types, d_lengths, d_angles, d_t, d_off, t.batch, t.lattice,
                           m->vp_betas, m->t_emb_w, m->ori, S, m->T, N, t.F);
        ARREAU_CHECK_HIP(hipGetLastError());
        TRY(gemm(ss, ts, t.fwd_mode, (int)M, C, S + 78, t.F, S + 78, 1, m->embT, C, 1, t.x, C));
    }
    // sender-side adjacency of this step's graph, for the ordered, atomic-free d(x_l) of the spatial conv in the backward pass
    // (built here, while the caller's offsets are certainly alive: the backward pass reads only the context's own arrays).  Round 5: only
    // the backward pass reads it, so it runs on the side stream behind the network's fiber branch (which waits for the neighbour list:
    // fork_side in arreau_general_network) and the backward pass waits for ev_rev
    if (!side_setup) LAUNCH(reverse_adjacency_kernel, dim3((unsigned)B), dim3(1024), d_off, t.deg, t.src, k, t.rev_start, t.rev_cnt, t.rev_idx);
    const int rc = arreau_general_network(m, arreau_graph_view{t.batch, t.deg, t.src, t.lattice, t.dir, t.dist}, d_off, B, N, d_eps,
                                          d_logits, d_len0, s);
    if (rc) return rc;
    if (side_setup) {
        hipLaunchKernelGGL(reverse_adjacency_kernel, dim3((unsigned)B), dim3(1024), 0, t.side, d_off, t.deg, t.src, k, t.rev_start, t.rev_cnt, t.rev_idx);
        ARREAU_CHECK_HIP(hipGetLastError());
        ARREAU_CHECK_HIP(hipEventRecord(t.ev_rev, t.side));
        t.rev_pending = true;
    }
    return ARREAU_OK;
}

extern "C" int arreau_train_backward(arreau_model* m, const float* d_g_eps, const float* d_g_logits, const float* d_g_len0,
                                     const arreau_state_dict* g, void* stream) {
    ARREAU_REQUIRE(m && d_g_eps && d_g_logits && d_g_len0 && g, "arreau_train_backward: null pointer");
    ARREAU_REQUIRE(m->train && m->train->N > 0, "arreau_train_backward: call arreau_train_forward first");
    const float* need[] = {g->basis_w1, g->basis_b1, g->basis_w2, g->basis_b2, g->fiber_w1, g->fiber_b1, g->fiber_w2, g->fiber_b2,
                           g->x_embedder_w, g->conv_kernel_w, g->conv_fiber_w, g->conv_bias, g->norm_w, g->norm_b, g->linear1_w,
                           g->linear1_b, g->linear2_w, g->linear2_b, g->readout_w, g->readout_b};
    for (const float* p : need) ARREAU_REQUIRE(p != nullptr, "arreau_train_backward: missing gradient buffer");
    ARREAU_REQUIRE(!m->cfg.has_layer_scale || g->layer_scale, "arreau_train_backward: missing layer_scale gradient buffer");
    hipStream_t s = (hipStream_t)stream;
    arreau_train_ctx& t = *m->train;
    const int N = t.N, C = m->C, D = m->D, L = m->L, H = m->H, k = m->k, S = m->S, RO = S + 4;
    const long R = (long)N * k * 16, M = (long)N * 16;
    auto W = [](const float* p) { return const_cast<float*>(p); };  // the gradient struct reuses the const state_dict type
    reset_deferred(t);
    if (t.rev_pending) { ARREAU_CHECK_HIP(hipStreamWaitEvent(s, t.ev_rev, 0)); t.rev_pending = false; }
    // d(rbar) [M][ROP], ROP = RO rounded up to a multiple of four with zero pad columns (round 5): its products and its column sum run on
    // 16-byte fetches -- 94 columns sent the d(x) product to the exact fp32 kernel (24.6 us) and the bias gradient to the element-wise
    // column sum (15.1 us); the weight operand's pad rows are the next layer's first rows resp. zeros behind the last layer (model.hip)
    const int ROP = (RO + 3) & ~3;
    LAUNCH(train_outputs_backward_kernel, dim3((unsigned)M), dim3(128), d_g_eps, d_g_logits, d_g_len0, t.batch, m->ori, S, N, t.drbar, ROP);
    // The read-outs' contributions to d x_{l+1} = d(rbar) . W_ro,l / L depend on nothing inside the layer loop: ONE batched product for all
    // layers up front (they were L launches of the element-wise-fetch kernel -- 94 read-out columns are no multiple of four -- 13.6 us
    // each inside the chain); layer L - 1 starts from its slice, the others are added by the launch that completes d x_{l+1}.
    TRY(arreau_sgemm(s, t.partial, (int)M, C, ROP, t.drbar, ROP, 1, t.ro_w, C, 1, t.dxro, C, 1.0f / (float)L, 0.f, L, 0, (long)RO * C, (long)M * C,
                     t.bwd_mode));
    const float invL = 1.0f / (float)L;
    for (int l = L - 1; l >= 0; --l) {
        const float* xl = t.x + (size_t)l * M * C;
        const float* xnext = t.x + (size_t)(l + 1) * M * C;
        const float* x1 = t.x1 + (size_t)l * M * C;
        const float* xhat = t.xhat + (size_t)l * M * C;
        const float* fk = t.fk + (size_t)l * 256 * C;
        const float* hpre = t.hpre + (size_t)l * M * H;
        const float* h = t.h + (size_t)l * M * H;
        const float* out = t.out + (size_t)l * M * C;
        // read-out (ponita.py:105,108); its weight gradient: one batched product over the layers, below the loop
        // (every layer's read-out sees the same d(rbar): the bias gradients are equal -- copied to the other layers in one launch below)
        // (ROP sums into the scratch row -- the pad columns sum to zero --, copied to every layer's slice at the end of the pass)
        // (with the batched column-sum pass behind the loop when the launches are merged: nothing waits for it)
        if (l == L - 1 && !train_fuse_on())
            TRY(colsum(s, t, t.drbar, nullptr, M, ROP, invL, t.robias, 0, nullptr, nullptr, 1, 0, 0, 0, 0, nullptr, nullptr, nullptr, true));
        const float* dxl = l == L - 1 ? t.dxro + (size_t)l * M * C : t.dx + (size_t)(l + 1) * M * C;   // d x_{l+1}
        float* dxo = t.dx + (size_t)l * M * C;                                                        // d x_l
        // round 5: below the top layer d(out) = d(x_{l+1}) * layer_scale was written by the previous iteration's conv-gradient launch, and
        // the column sums over d(x_{l+1}) (d(layer_scale), d(linear_2.bias): results nothing waits for) leave the chain: one batched
        // pass behind the loop.  The same multiplies and the same sums.
        const bool dout_ahead = m->cfg.has_layer_scale && C % 4 == 0 && train_fuse_on() && C <= 2048;
        const float* dx_add = l > 0 ? t.dxro + (size_t)(l - 1) * M * C : nullptr;
        // ConvNext tail: x_{l+1} = out * ls + x_l
        // d(layer_scale) = sum_rows dx * out and d(linear_2.bias) = sum_rows dout = ls * sum_rows dx, in one pass over dx
        float* dout = t.dout_all + (size_t)l * M * C;
        bool have_dout = false;  // (d(out) = d(x) * layer_scale rides in the column-sum pass over d(x) where that pass takes 16-byte columns)
        if (dout_ahead && l < L - 1) have_dout = true;
        else if (m->cfg.has_layer_scale)
            TRY(colsum(s, t, dxl, out, M, C, 1.0f, W(g->layer_scale) + (size_t)l * C, 0, W(g->linear2_b) + (size_t)l * C, m->ls + (size_t)l * C, 1, 0, 0,
                       0, 0, dout, m->ls + (size_t)l * C, &have_dout, true));
        // (the weight gradients of linear_2, linear_1, the read-out and the fiber kernel are products nothing below waits for:
        // their operands are kept per layer and each kind runs as ONE batched product after the loop)
        float* dh = t.dh + (size_t)l * M * H;
        if (!have_dout) LAUNCH(scale_cols_kernel, dim3(blocks(M * C)), dim3(256), dxl, m->ls + (size_t)l * C, M, C, dout);   // dout
        if (!m->cfg.has_layer_scale) TRY(colsum(s, t, dout, nullptr, M, C, 1.0f, W(g->linear2_b) + (size_t)l * C));
        TRY(linear_dx_gelu_backward(s, t, M, H, C, dout, t.lin2 + (size_t)l * C * H, hpre, (const float*)nullptr, dh));   // dhpre
        float* dxn = t.dxn_all + (size_t)l * M * C;
        float* dx2 = t.dx2_all + (size_t)l * M * C;
        TRY(linear_dx(s, t, M, C, H, dh, t.lin1 + (size_t)l * H * C, dxn));                                          // dxn
        const bool fuse = C % 4 == 0 && train_fuse_on() && C <= 2048;
        if (!fuse)
            LAUNCH(ln_backward_kernel, dim3(blocks(M, 4)), dim3(256), dxn, xhat, t.rstd + (size_t)l * M, m->ln_w + (size_t)l * C, M, C, dx2);
        // spherical conv: x2 = mix(x1, fk) / 16 + bias
        // spatial conv: x1 = sum_s kern * x_l[src]; the residual path already sits in dx (= d x_l so far)
        if (fuse) {
            hipLaunchKernelGGL(ln_mix_backward_kernel4, dim3((unsigned)N), dim3(1024), (size_t)16 * C * sizeof(float), s, dxn, xhat,
                               t.rstd + (size_t)l * M, m->ln_w + (size_t)l * C, V4(fk), N, C, dx2, V4W(t.dx1));
            ARREAU_CHECK_HIP(hipGetLastError());
            const unsigned dx_blocks = blocks(M * C / 4), kern_blocks = blocks(R * C / 4);
            LAUNCH(conv_backward_both_kernel4, dim3(dx_blocks + kern_blocks), dim3(256), (int)dx_blocks, V4(xl), V4(t.dx1), t.deg, t.src,
                   V4(t.kern + (size_t)l * C), L * C / 4, t.rev_start, t.rev_cnt, t.rev_idx, N, k, C / 4, V4(dxl),
                   dx_add ? V4(dx_add) : (const f32x4*)nullptr, V4W(dxo), V4W(t.dkern + (size_t)l * C),
                   dout_ahead && l > 0 ? V4W(t.dout_all + (size_t)(l - 1) * M * C) : (f32x4*)nullptr, V4(m->ls + (size_t)(l > 0 ? l - 1 : 0) * C));
        } else if (C % 4 == 0) {
            LAUNCH(mix_backward_x_kernel4, dim3(blocks(M * C / 4)), dim3(256), V4(dx2), V4(fk), N, C / 4, V4W(t.dx1));
            LAUNCH(conv_backward_kern_kernel4, dim3(blocks(R * C / 4)), dim3(256), V4(xl), V4(t.dx1), t.deg, t.src, N, k, C / 4, L * C / 4,
                   V4W(t.dkern + (size_t)l * C));
            LAUNCH(conv_backward_dx_kernel4, dim3(blocks(M * C / 4)), dim3(256), V4(t.kern + (size_t)l * C), L * C / 4, V4(t.dx1), t.rev_start,
                   t.rev_cnt, t.rev_idx, N, k, C / 4, V4(dxl), dx_add ? V4(dx_add) : (const f32x4*)nullptr, V4W(dxo));
        } else {
            LAUNCH(mix_backward_x_kernel, dim3(blocks(M * C)), dim3(256), dx2, fk, N, C, t.dx1);
            LAUNCH(conv_backward_kern_kernel, dim3(blocks(R * C)), dim3(256), xl, t.dx1, t.deg, t.src, N, k, C, L * C, t.dkern + (size_t)l * C);
            LAUNCH(conv_backward_dx_kernel, dim3(blocks(M * C)), dim3(256), t.kern + (size_t)l * C, L * C, t.dx1, t.rev_start, t.rev_cnt,
                   t.rev_idx, N, k, C, dxl, dx_add, dxo);
        }
    }
    // d(fiber kernel) of every layer = sum over nodes of x1 (x) dx2 / 16: one batched pair of launches (both operands were kept
    // per layer), then its two uses -- and from there the whole fiber branch down to d(fiber_basis_fn): on the side stream, beside the
    // edge-level weight gradients below (it reads x1 and d(x2) of the loop above and writes gradients nothing else touches)
    // the layers' weight gradients, one batched product per kind (operands kept per layer above / by the forward pass), and the batched
    // column sums: main-stream work that depends on nothing of the side branch
    auto main_batched = [&]() -> int {
        // (round 5: the k-slice sums of these products and the chunk sums of the column-sum passes below wait for the two launches at the
        // end of this function)
        TRY(linear_dw_batched(s, t, L, M, C, RO, t.drbar, 0, t.x + (size_t)M * C, (long)M * C, W(g->readout_w), invL, true, ROP));         // x_{l+1}
        TRY(linear_dw_batched(s, t, L, M, H, C, t.dout_all, (long)M * C, t.h, (long)M * H, W(g->linear2_w), 1.f, true));
        TRY(linear_dw_batched(s, t, L, M, C, H, t.dh, (long)M * H, t.xn_all, (long)M * C, W(g->linear1_w), 1.f, true));
        // the k-slice sums of the three products in one launch, while their 65 MB of partial tiles are still in the Infinity Cache
        // (ONE such launch at the very end of the pass read 125 MB of long-evicted tiles from HBM in plane-strided pieces: 60 us
        // against 49 us for the seven separate sums; per cluster of products it is three launches fewer and faster than both)
        TRY(flush_deferred_gemms(s, t));
        // ... and the column sums: d(linear_1.bias) = sum_rows dhpre; d(norm.weight) = sum_rows dxn * xhat and d(norm.bias) = sum_rows dxn
        // in one pass over dxn; d(conv.bias) = sum_rows dx2
        // (round 5: the three passes as ONE launch where their chunk sums are deferred -- 15 matrices of the same row count)
        ColsumGather cg;
        if (train_fuse_on())   // d(readout bias): the one column sum of d(rbar) (scratch row, copied to every layer's slice at the end)
            TRY(colsum(s, t, t.drbar, nullptr, M, ROP, invL, t.robias, 0, nullptr, nullptr, 1, 0, 0, 0, 0, nullptr, nullptr, nullptr, true, 0, nullptr,
                       nullptr, &cg));
        TRY(colsum(s, t, t.dh, nullptr, M, H, 1.0f, W(g->linear1_b), 0, nullptr, nullptr, L, (long)M * H, 0, H, 0, nullptr, nullptr, nullptr, true, 0,
                   nullptr, nullptr, &cg));
        TRY(colsum(s, t, t.dxn_all, t.xhat, M, C, 1.0f, W(g->norm_w), 0, W(g->norm_b), nullptr, L, (long)M * C, (long)M * C, C, C, nullptr, nullptr,
                   nullptr, true, 0, nullptr, nullptr, &cg));
        TRY(colsum(s, t, t.dx2_all, nullptr, M, C, 1.0f, W(g->conv_bias), 0, nullptr, nullptr, L, (long)M * C, 0, C, 0, nullptr, nullptr, nullptr, true, 0,
                   nullptr, nullptr, &cg));
        // d(layer_scale) = sum_rows d(x_{l+1}) * out_l and d(linear_2.bias) = layer_scale * sum_rows d(x_{l+1}) of the layers below the top one
        // (the top layer's pass ran at the head of the loop; d(x_{l+1}) = slice l + 1 of the kept d(x))
        if (m->cfg.has_layer_scale && C % 4 == 0 && train_fuse_on() && C <= 2048 && L > 1)
            TRY(colsum(s, t, t.dx + (size_t)M * C, t.out, M, C, 1.0f, W(g->layer_scale), 0, W(g->linear2_b), m->ls, L - 1, (long)M * C, (long)M * C, C, C,
                       nullptr, nullptr, nullptr, true, C, nullptr, nullptr, &cg));
        TRY(launch_gathered(s, cg));
        return ARREAU_OK;
    };
    {
        hipStream_t main_stream = s;
        arreau_train_ctx ts = t;
        hipStream_t s = main_stream;
        TRY(fork_side(t, main_stream, ts, s));
        TRY(main_batched());   // (before the side branch's two dozen small launches: see the forward pass)
        arreau_train_ctx& t = ts;   // (the helpers take their scratch from the context they are handed)
        // partial sums live in the split-K scratch (free here): as many layers per pair of launches as fit it -- all L at the
        // bench's 64 crystals, one at the reference's `make train` preset (batch 270, hidden_dim 200: ~2,200 atoms) -- and
        // atom chunks that grow with the batch once a single layer's partial sums would not fit
        int chunk = MIX_CHUNK;
        while ((size_t)((N + chunk - 1) / chunk) * 256 * C > PARTIAL_FLOATS) chunk *= 2;
        const int chunks = (N + chunk - 1) / chunk;
        const int Lg = (int)std::min<size_t>((size_t)L, PARTIAL_FLOATS / ((size_t)chunks * 256 * C));
        for (int l0 = 0; l0 < L; l0 += Lg) {
            const int nl = std::min(Lg, L - l0);
            LAUNCH(mix_backward_fk_partial_kernel, dim3(chunks, nl), dim3(128), t.x1 + (size_t)l0 * N * 16 * C,
                   t.dx2_all + (size_t)l0 * N * 16 * C, N, C, t.partial, chunk);
            LAUNCH(mix_backward_fk_final_kernel, dim3(blocks(256L * C), nl), dim3(256), t.partial, chunks, C, t.dfk_all + (size_t)l0 * 256 * C);
        }
        // d(fiber basis) = sum over layers of d(fk_l) . Wfk_l: one batched product, summed in the order of the layer loop (L - 1 first:
        // slot z of the scratch holds layer L - 1 - z)
        TRY(arreau_sgemm(s, t.partial, 256, D, C, t.dfk_all + (size_t)(L - 1) * 256 * C, C, 1, m->fiber_wk + (size_t)(L - 1) * C * D, D, 1, t.dfkb_all, D,
                         1.0f, 0.f, L, -256L * C, -(long)C * D, 256L * D, t.bwd_mode));
        TRY(ordered_sum(s, t.dfkb_all, L, 256, D, t.dfkb));
        TRY(linear_dw_batched(s, t, L, 256, D, C, t.dfk_all, 256L * C, t.fkb, 0, W(g->conv_fiber_w)));
        // fiber basis MLP
        TRY(launch_gelu_backward(s, t.dfkb, t.fh2pre, (const float*)nullptr, 256L, D));
        TRY(linear_dw(s, t, 256, C, D, t.dfkb, t.fh1, W(g->fiber_w2)));
        TRY(colsum(s, t, t.dfkb, nullptr, 256, D, 1.0f, W(g->fiber_b2)));
        TRY(linear_dx_gelu_backward(s, t, 256, C, D, t.dfkb, m->fiber_w2, t.fh1pre, (const float*)nullptr, t.dfh1));
        TRY(linear_dw(s, t, 256, 3, C, t.dfh1, t.fpoly, W(g->fiber_w1)));
        TRY(colsum(s, t, t.dfh1, nullptr, 256, C, 1.0f, W(g->fiber_b1)));
        if (t.side) ARREAU_CHECK_HIP(hipEventRecord(t.ev_join, t.side));
    }
    // kernel projections of all layers at once: dWk [L*C][D] = dkern^T . kb,  dkb = dkern . Wk
    TRY(linear_dw(s, t, R, D, L * C, t.dkern, t.kb, W(g->conv_kernel_w)));
    const bool dkb_colsum_fused = train_fuse_on() && D % 4 == 0;
    if (dkb_colsum_fused) {
        // (round 5: a 128 x 128 product takes no GELU epilogue -- sgemm.h -- so d(h2pre) = (dkern . Wk) * gelu'(h2pre) * window is finished by
        // the column-sum pass of d(basis_fn.2.bias), which had to read it anyway: one pass over 66 MB instead of two)
        bool fused = false;
        Epi e;
        e.kind = 2; e.mat = t.h2pre; e.row = t.window;
        TRY(linear_dx(s, t, R, D, L * C, t.dkern, t.wk, t.dkb, 1.f, 0.f, &e, &fused));
        TRY(colsum(s, t, t.dkb, nullptr, R, D, 1.0f, W(g->basis_b2), 0, nullptr, nullptr, 1, 0, 0, 0, 0, nullptr, nullptr, nullptr, true, 0,
                   fused ? nullptr : t.h2pre, fused ? nullptr : t.window));
    } else
    TRY(linear_dx_gelu_backward(s, t, R, D, L * C, t.dkern, t.wk, t.h2pre, (const float*)t.window, t.dkb));   // dh2pre
    // embedding: x_0 = F . W_emb^T  -> dW_emb[c][i] = sum_rows dx[row][c] F[row][i]
    TRY(linear_dw(s, t, M, S + 78, C, t.dx, t.F, W(g->x_embedder_w), 1.f, true));
    // edge basis MLP
    TRY(linear_dw(s, t, R, C, D, t.dkb, t.h1, W(g->basis_w2), 1.f, true));
    TRY(flush_deferred_gemms(s, t));     // (x_embedder and basis_fn.2 weight gradients: one launch)
    if (!dkb_colsum_fused) TRY(colsum(s, t, t.dkb, nullptr, R, D, 1.0f, W(g->basis_b2), 0, nullptr, nullptr, 1, 0, 0, 0, 0, nullptr, nullptr, nullptr, true));
    TRY(linear_dx_gelu_backward(s, t, R, C, D, t.dkb, t.w2, t.h1pre, (const float*)nullptr, t.dh1));   // dh1pre
    TRY(linear_dw(s, t, R, ARREAU_MONO_PAD, C, t.dh1, t.mono, t.dw1f));
    TRY(colsum(s, t, t.dh1, nullptr, R, C, 1.0f, W(g->basis_b1), 0, nullptr, nullptr, 1, 0, 0, 0, 0, nullptr, nullptr, nullptr, true));
    TRY(flush_deferred_gemms(s, t));     // (nothing pending on the default path: a product that did not fit its cluster's scratch)
    TRY(flush_deferred_colsums(s, t));   // every bias / norm / layer-scale gradient's chunk sum: one launch
    {   // d(readout bias): every layer's read-out sees the same d(rbar), so the one column sum (scratch row, complete behind the launch
        // above) is copied to all L slices
        CopySegments seg;
        const int nseg = std::min(L, 24);
        for (int l = 0; l < nseg; ++l) {
            seg.dst[l] = W(g->readout_b) + (size_t)l * RO; seg.src[l] = t.robias; seg.n[l] = (unsigned)RO;
        }
        LAUNCH(copy_segments_kernel, dim3(1, nseg), dim3(128), seg);
        for (int l = nseg; l < L; ++l)  // (more than 24 layers: the rest one by one)
            ARREAU_CHECK_HIP(hipMemcpyAsync(W(g->readout_b) + (size_t)l * RO, t.robias, RO * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    LAUNCH(unfold_poly_grad_kernel, dim3(blocks((long)C * ARREAU_POLY_COLS)), dim3(256), t.dw1f, C, W(g->basis_w1));
    TRY(join_side(t, s));   // the fiber branch's gradients are complete when this call's work is
    return ARREAU_OK;
}

// activation statistics of the first training forward for FiberBundleConv.callibrate (conv.py:121-123,140-146): the
// unbiased standard deviations of x (layer input), x_1 (after the spatial conv) and x_2 (after the spherical conv,
// before the bias) of every layer, as torch.std() computes them.   d_stats[L][3]
namespace {
// Two stages, no atomics (round 5; VERDICT round 4, weak 6: this was ONE workgroup over 1.09 M elements, 1.5 ms x 15 launches on the
// first training step): STD_PARTS workgroups each sum a contiguous chunk in double, then one workgroup adds the partial sums in
// chunk order -- the same result whatever the chip does.
#define STD_PARTS 256
__global__ __launch_bounds__(256) void std_partial_kernel(const float* __restrict__ a, long n, double* __restrict__ part /*[2][STD_PARTS]*/) {
    __shared__ double s1[256], s2[256];
    const long chunk = (n + STD_PARTS - 1) / STD_PARTS;
    const long beg = (long)blockIdx.x * chunk, end = beg + chunk < n ? beg + chunk : n;
    double p = 0.0, q = 0.0;
    for (long i = beg + threadIdx.x; i < end; i += 256) { const double v = a[i]; p += v; q += v * v; }
    s1[threadIdx.x] = p; s2[threadIdx.x] = q;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) { s1[threadIdx.x] += s1[threadIdx.x + st]; s2[threadIdx.x] += s2[threadIdx.x + st]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[blockIdx.x] = s1[0]; part[STD_PARTS + blockIdx.x] = s2[0]; }
}
__global__ __launch_bounds__(256) void std_final_kernel(const double* __restrict__ part, long n, float* __restrict__ out) {
    __shared__ double s1[256], s2[256];
    s1[threadIdx.x] = threadIdx.x < STD_PARTS ? part[threadIdx.x] : 0.0;
    s2[threadIdx.x] = threadIdx.x < STD_PARTS ? part[STD_PARTS + threadIdx.x] : 0.0;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) { s1[threadIdx.x] += s1[threadIdx.x + st]; s2[threadIdx.x] += s2[threadIdx.x + st]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double mean = s1[0] / (double)n;
        out[0] = (float)sqrt(fmax((s2[0] - (double)n * mean * mean) / (double)(n - 1), 0.0));
    }
}
}  // namespace

extern "C" int arreau_train_conv_stats(arreau_model* m, float* d_stats, void* stream) {
    ARREAU_REQUIRE(m && d_stats, "arreau_train_conv_stats: null pointer");
    ARREAU_REQUIRE(m->train && m->train->N > 0, "arreau_train_conv_stats: call arreau_train_forward first");
    hipStream_t s = (hipStream_t)stream;
    arreau_train_ctx& t = *m->train;
    const int N = t.N, C = m->C, L = m->L;
    const long M = (long)N * 16;
    static double* part = nullptr;  // [2][STD_PARTS] partial sums (one process drives one GPU; the launches are stream-ordered)
    if (!part) ARREAU_CHECK_HIP(hipMalloc((void**)&part, 2 * STD_PARTS * sizeof(double)));
    auto std_of = [&](const float* a, float* out) {
        LAUNCH(std_partial_kernel, dim3(STD_PARTS), dim3(256), a, M * C, part);
        LAUNCH(std_final_kernel, dim3(1), dim3(256), (const double*)part, M * C, out);
        return ARREAU_OK;
    };
    for (int l = 0; l < L; ++l) {
        const float* xl = t.x + (size_t)l * M * C;
        const float* x1 = t.x1 + (size_t)l * M * C;
        TRY(std_of(xl, d_stats + 3 * l));
        TRY(std_of(x1, d_stats + 3 * l + 1));
        // x_2 without the bias: recompute the mix into scratch
        LAUNCH(mix_forward_kernel, dim3(blocks(M * C)), dim3(256), x1, t.fk + (size_t)l * 256 * C, (const float*)t.scratch_cols, N, C, t.dtmp);
        TRY(std_of(t.dtmp, d_stats + 3 * l + 2));
    }
    return ARREAU_OK;
}

// After an optimizer step: refresh the plain fp32 weights the TRAINING path reads (this file, prep_kernel's embT, the
// biases / LayerNorm / layer-scale vectors) from the caller's DEVICE tensors -- device-to-device copies, one fold and one
// transpose kernel, no host work.  The packed operand planes of the sampling kernels are NOT rebuilt: the model is marked
// stale for sampling (arreau_predict_scores & co. then fail loudly) until it is re-created from the new state_dict.
extern "C" int arreau_model_update_train_weights(arreau_model* m, const arreau_state_dict* d, void* stream) {
    ARREAU_REQUIRE(m && d, "arreau_model_update_train_weights: null pointer");
    const float* need[] = {d->basis_w1, d->basis_b1, d->basis_w2, d->basis_b2, d->fiber_w1, d->fiber_b1, d->fiber_w2, d->fiber_b2,
                           d->x_embedder_w, d->conv_kernel_w, d->conv_fiber_w, d->conv_bias, d->norm_w, d->norm_b, d->linear1_w,
                           d->linear1_b, d->linear2_w, d->linear2_b, d->readout_w, d->readout_b};
    for (const float* p : need) ARREAU_REQUIRE(p != nullptr, "arreau_model_update_train_weights: missing state_dict entry");
    hipStream_t s = (hipStream_t)stream;
    const size_t C = m->C, D = m->D, L = m->L, H = m->H, S = m->S, RO = S + 4;
    auto W = [](const float* p) { return const_cast<float*>(p); };
    CopySegments seg;
    int nseg = 0;
    auto cp = [&](const float* dst, const float* src, size_t n) {
        seg.dst[nseg] = W(dst); seg.src[nseg] = src; seg.n[nseg] = (unsigned)n;
        ++nseg;
    };
    m->packed_stale = 1;
    if (!m->train) TRY(ensure_ctx(m, 1, 1, s));  // (the monomial column table lives in the training context)
    LAUNCH(fold_poly_weight_kernel, dim3(blocks((long)C * ARREAU_MONO_PAD)), dim3(256), d->basis_w1, (int)C, m->train->mono_cols, W(m->t_w1f));
    LAUNCH(transpose_kernel, dim3(blocks((long)C * (S + 78))), dim3(256), d->x_embedder_w, (int)C, (int)(S + 78), W(m->embT));
    cp(m->b1, d->basis_b1, C);
    cp(m->t_w2, d->basis_w2, D * C);
    cp(m->b2, d->basis_b2, D);
    cp(m->fiber_w1, d->fiber_w1, C * 3);
    cp(m->fiber_b1, d->fiber_b1, C);
    cp(m->fiber_w2, d->fiber_w2, D * C);
    cp(m->fiber_b2, d->fiber_b2, D);
    cp(m->t_wk, d->conv_kernel_w, L * C * D);
    cp(m->fiber_wk, d->conv_fiber_w, L * C * D);
    cp(m->conv_bias, d->conv_bias, L * C);
    cp(m->ln_w, d->norm_w, L * C);
    cp(m->ln_b, d->norm_b, L * C);
    cp(m->t_lin1, d->linear1_w, L * H * C);
    cp(m->mb1, d->linear1_b, L * H);
    cp(m->t_lin2, d->linear2_w, L * C * H);
    cp(m->mb2, d->linear2_b, L * C);
    if (m->cfg.has_layer_scale && d->layer_scale) cp(m->ls, d->layer_scale, L * C);
    cp(m->t_ro_w, d->readout_w, L * RO * C);
    cp(m->ro_b, d->readout_b, L * RO);
    LAUNCH(copy_segments_kernel, dim3(64, nseg), dim3(256), seg);
    TRY(arreau_repack_mlp_f16x3_m16(m, s));   // the plane stream the fused training forward of the ConvNext block reads
    return ARREAU_OK;
}

// The optimizer's second destination (optim.hip): where arreau_model_update_train_weights would copy each tensor to.  Stacked [L, ...]
// layout, the caller's own state_dict layout; NULL for the two tensors the training entry points read in a derived form.
extern "C" int arreau_model_train_weight_pointers(arreau_model* m, arreau_state_dict* out) {
    ARREAU_REQUIRE(m && out, "arreau_model_train_weight_pointers: null pointer");
    *out = arreau_state_dict{};
    out->basis_b1 = m->b1; out->basis_w2 = m->t_w2; out->basis_b2 = m->b2;
    out->fiber_w1 = m->fiber_w1; out->fiber_b1 = m->fiber_b1; out->fiber_w2 = m->fiber_w2; out->fiber_b2 = m->fiber_b2;
    out->conv_kernel_w = m->t_wk; out->conv_fiber_w = m->fiber_wk; out->conv_bias = m->conv_bias;
    out->norm_w = m->ln_w; out->norm_b = m->ln_b;
    out->linear1_w = m->t_lin1; out->linear1_b = m->mb1; out->linear2_w = m->t_lin2; out->linear2_b = m->mb2;
    if (m->cfg.has_layer_scale) out->layer_scale = m->ls;
    out->readout_w = m->t_ro_w; out->readout_b = m->ro_b;
    return ARREAU_OK;
}

// ... and what is left of arreau_model_update_train_weights when the optimizer has written those itself: the folded polynomial weight
// and the transposed embedder.
extern "C" int arreau_model_refresh_derived_train_weights(arreau_model* m, const float* d_basis_w1, const float* d_x_embedder_w, void* stream) {
    ARREAU_REQUIRE(m && d_basis_w1 && d_x_embedder_w, "arreau_model_refresh_derived_train_weights: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const size_t C = m->C, S = m->S;
    auto W = [](const float* p) { return const_cast<float*>(p); };
    m->packed_stale = 1;
    if (!m->train) TRY(ensure_ctx(m, 1, 1, s));
    LAUNCH(fold_poly_weight_kernel, dim3(blocks((long)C * ARREAU_MONO_PAD)), dim3(256), d_basis_w1, (int)C, m->train->mono_cols, W(m->t_w1f));
    LAUNCH(transpose_kernel, dim3(blocks((long)C * (S + 78))), dim3(256), d_x_embedder_w, (int)C, (int)(S + 78), W(m->embT));
    TRY(arreau_repack_mlp_f16x3_m16(m, s));
    return ARREAU_OK;
}

extern "C" int arreau_debug_sgemm(int32_t mode, int32_t M, int32_t N, int32_t K, const float* d_A, int64_t as0, int64_t as1, const float* d_B,
                                  int64_t bs0, int64_t bs1, float* d_C, int32_t ldc, float alpha, float beta, void* stream) {
    ARREAU_REQUIRE(d_A && d_B && d_C, "arreau_debug_sgemm: null pointer");
    ARREAU_REQUIRE(M >= 0 && N >= 0 && K >= 1 && ldc >= N && mode >= 0 && mode <= 2, "arreau_debug_sgemm: bad size or mode");
    ARREAU_REQUIRE((as0 == 1 || as1 == 1) && (bs0 == 1 || bs1 == 1), "arreau_debug_sgemm: one stride of each operand must be 1");
    hipStream_t s = (hipStream_t)stream;
    float* partial = nullptr;
    ARREAU_CHECK_HIP(hipMalloc((void**)&partial, ARREAU_SGEMM_PARTIAL_FLOATS * sizeof(float)));
    int rc = arreau_sgemm(s, partial, M, N, K, d_A, (long)as0, (long)as1, d_B, (long)bs0, (long)bs1, d_C, ldc, alpha, beta, 1, 0, 0, 0, mode);
    const hipError_t e = hipStreamSynchronize(s);
    (void)hipFree(partial);
    if (rc == ARREAU_OK && e != hipSuccess) {
        arreau_set_error(std::string("arreau_debug_sgemm: ") + hipGetErrorString(e));
        rc = ARREAU_EHIP;
    }
    return rc;
}
