// Node-side kernels of the score network: feature assembly + embedding, the per-layer
// message/spherical-conv/ConvNext update, and the read-outs.
#include <stdlib.h>

#include "internal.h"
#include "readout_dev.h"
#include "embed_dev.h"
#include "prep_dev.h"

// ---------------------------------------------------------------------------------------------
// K0 prep: per crystal lattice, Cartesian coordinates, node->crystal map, and the part of the
// embedding that is shared by all atoms of a crystal.
//
// diffusion_loss.py:124-158 builds x = cat(one_hot(type) [S], t_emb(betas[t]) [64], n [1],
// lengths [3], angles [3], |lengths/n| [3]) per atom and ponita.py:98 applies x_embedder
// (Linear S+78 -> C, no bias) after replicating those scalars over the 16 orientations
// (position_orientation_graph.py:82-86).  Everything except the one-hot column and the 4 vector
// channels is constant inside a crystal, so it is reduced once per crystal to cvec[b][C].
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void prep_kernel(
    const float* __restrict__ frac, const float* __restrict__ lengths, const float* __restrict__ angles,
    const int32_t* __restrict__ tstep, const int32_t* __restrict__ offsets, const float* __restrict__ betas,
    const float* __restrict__ t_emb_w, const float* __restrict__ embT, int S, int C, int T,
    float* __restrict__ lattice, float* __restrict__ cart, int32_t* __restrict__ batch, float* __restrict__ cvec,
    int32_t* __restrict__ status, int32_t* __restrict__ t_next, int32_t* __restrict__ t_cur, int b0,
    int t_offset /* added to the timestep read (the sampling loop's first set-up: -1, see arreau_sample_loop) */) {
    __shared__ float feat[ARREAU_T_EMB_DIM + ARREAU_N_CRYSTAL_FEATS];
    // Sampling loop on device time (arreau_sample_loop): this workgroup is the only reader of its crystal's entry of
    // t_next in this launch; it publishes the timestep of the step in t_cur (read by the later kernels of the step) and
    // leaves the next one behind, so a captured step can be replayed without host-side bookkeeping.
    __shared__ int t_sh;
    const int b = b0 + blockIdx.x;
    if (threadIdx.x == 0) {
        if (t_next != nullptr) {
            t_sh = t_next[b];
            t_cur[b] = t_sh;
            t_next[b] = t_sh - 1;
        } else {
            t_sh = tstep[b];
        }
    }
    __syncthreads();
    __shared__ float Lm[9];
    const int first = offsets[b], n = offsets[b + 1] - first;
    const float* len = lengths + 3 * b;
    const float* ang = angles + 3 * b;
    if (threadIdx.x == 0) {
        float tmp[9];
        arreau_prep_cell(len, ang, tmp);
        for (int i = 0; i < 9; ++i) { Lm[i] = tmp[i]; lattice[9 * b + i] = tmp[i]; }
    }
    arreau_prep_cvec(t_sh + t_offset, n, len, ang, betas, t_emb_w, embT, S, C, T, feat, cvec + (size_t)b * C, status);  // (barriers inside)
    for (int a = threadIdx.x; a < n; a += blockDim.x) {
        const size_t i = (size_t)first + a;
        const float f0 = frac[3 * i], f1 = frac[3 * i + 1], f2 = frac[3 * i + 2];
#pragma unroll
        for (int j = 0; j < 3; ++j) cart[3 * i + j] = (f0 * Lm[j] + f1 * Lm[3 + j]) + f2 * Lm[6 + j];
        batch[i] = b;
    }
}

int arreau_launch_prep(const arreau_model* m, const float* frac, const float* lengths, const float* angles,
                       const int32_t* t, const int32_t* offsets, int B, int N, float* lattice, float* cart,
                       int32_t* batch, float* cvec, hipStream_t s, int32_t* t_next, int32_t* t_cur, NodeRange r, int t_offset) {
    const int b0 = r.b0, b1 = r.b1 < 0 ? B : r.b1;
    if (b1 <= b0) return ARREAU_OK;
    ARREAU_LAUNCH(prep_kernel, dim3(b1 - b0), dim3(128), 0, s, frac, lengths, angles, t, offsets, m->vp_betas,
                       m->t_emb_w, m->embT, m->S, m->C, m->T, lattice, cart, batch, cvec, m->status, t_next, t_cur, b0, t_offset);
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}

// ---------------------------------------------------------------------------------------------
// Embedding (body in embed_dev.h): x0[n][o][c] = embT[type_n][c] + cvec[b][c] + sum_v embT[S+74+v][c] * (vec[n][v] . ori[o])
// with vec[n] = (frac_n, lattice rows a, b, c)  (diffusion_loss.py:158; to_from_sphere.py:4-5).  Stand-alone launch: the
// teacher-forced and sliced paths; the sampler's step embeds inside the neighbour-list launch (graph.hip).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_kernel(
    const float* __restrict__ frac, const int32_t* __restrict__ types, const float* __restrict__ lattice,
    const int32_t* __restrict__ batch, const float* __restrict__ cvec, const float* __restrict__ ori,
    const float* __restrict__ embT, int S, int C, int n0, int N /* atoms n0 .. N-1 */, float* __restrict__ x0,
    int32_t* __restrict__ status) {
    arreau_embed_body(blockIdx.x * blockDim.x + threadIdx.x, frac, types, lattice, batch, cvec, ori, embT, S, C, n0, N, x0, status);
}

int arreau_launch_embed(const arreau_model* m, const float* frac, const int32_t* types, const float* lattice,
                        const int32_t* batch, const float* cvec, int N, float* x0, hipStream_t s, NodeRange r) {
    const int n0 = r.n0, n1 = r.n1 < 0 ? N : r.n1;
    if (n1 <= n0) return ARREAU_OK;
    const long long total = (long long)(n1 - n0) * (m->C / 4);
    if (total >= (1ll << 31)) {
        arreau_set_error("embed kernel: more than 2^31 (atom, channel group) pairs in one launch");
        return ARREAU_EINVAL;
    }
    ARREAU_LAUNCH(embed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, frac, types, lattice, batch,
                       cvec, m->ori, m->embT, m->S, m->C, n0, n1, x0, m->status);
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}

// ---------------------------------------------------------------------------------------------
// Embedding of caller-supplied node features (the inner operator seam, arreau_ponita_forward):
//   x0[n][o][c] = sum_{i < S+74} x[n][i] * W[c][i] + sum_v W[c][S+74+v] * (vec[n][v] . ori[o])
// i.e. x_embedder (ponita.py:70,98) applied to cat(scalar_to_sphere(x), vec_to_sphere(vec))
// (position_orientation_graph.py:82-86, to_from_sphere.py:4-8) for ANY x -- soft type vectors, a per-atom time
// embedding, whatever the caller assembled -- not only the one-hot/per-crystal form the sampler produces.
// Thread = (atom n, float4 column c4): the scalar part is one pass over the atom's S+74 features (x row broadcast
// from L1, weight rows coalesced over c4), shared by the 16 orientations.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_general_kernel(
    const float* __restrict__ x, const float* __restrict__ vec, const float* __restrict__ ori,
    const float* __restrict__ embT, int S, int C, int N, float* __restrict__ x0) {
    const int C4 = C / 4;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)N * C4) return;
    const int c4 = (int)(idx % C4);
    const int n = (int)(idx / C4);
    const int F = S + 74;
    const f32x4* e4 = reinterpret_cast<const f32x4*>(embT);
    const float* xr = x + (size_t)n * F;
    f32x4 base = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int i = 0; i < F; ++i) base += e4[(size_t)i * C4 + c4] * xr[i];
    float v[4][3];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int d = 0; d < 3; ++d) v[q][d] = vec[((size_t)n * 4 + q) * 3 + d];
    f32x4 ev[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) ev[q] = e4[(size_t)(F + q) * C4 + c4];
    f32x4* out = reinterpret_cast<f32x4*>(x0) + (size_t)n * ARREAU_ORI * C4 + c4;
#pragma unroll
    for (int o = 0; o < ARREAU_ORI; ++o) {
        const float ox = ori[3 * o], oy = ori[3 * o + 1], oz = ori[3 * o + 2];
        f32x4 acc = base;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc += ev[q] * ((v[q][0] * ox + v[q][1] * oy) + v[q][2] * oz);
        out[(size_t)o * C4] = acc;
    }
}

int arreau_launch_embed_general(const arreau_model* m, const float* x, const float* vec, int N, float* x0, hipStream_t s) {
    if (N == 0) return ARREAU_OK;
    const long long total = (long long)N * (m->C / 4);
    ARREAU_LAUNCH(embed_general_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, vec, m->ori, m->embT,
                       m->S, m->C, N, x0);
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}

// node -> crystal map from the CSR offsets (graph.batch of the reference, diffusion_loss.py:330-335)
__global__ void batch_index_kernel(const int32_t* __restrict__ offsets, int B, int32_t* __restrict__ batch) {
    const int b = blockIdx.x;
    for (int i = offsets[b] + threadIdx.x; i < offsets[b + 1]; i += blockDim.x) batch[i] = b;
}

int arreau_launch_batch_index(const int32_t* offsets, int B, int N, int32_t* batch, hipStream_t s) {
    if (N == 0 || B == 0) return ARREAU_OK;
    ARREAU_LAUNCH(batch_index_kernel, dim3(B), dim3(64), 0, s, offsets, B, batch);
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}

// ---------------------------------------------------------------------------------------------
// K4a: message passing + spherical convolution of one layer (HBM-bound; no MFMA).
//   x1[n,o,c] = sum_{slots s<deg[n]} K_l[(n,s),o,c] * x[src(n,s),o,c]      (conv.py:111,131-133 + PyG sum
//                                                                            aggregation onto the receiver)
//   x2[n,p,c] = sum_o x1[n,o,c] * FK_l[o,p,c] + bias[c]                     (FK holds the /O; conv.py:113-127)
// Persistent workgroups of 512 threads walk the nodes.  Thread (c, pq) keeps its slice of the layer's fiber
// kernel FK_l[0..15][4pq..4pq+3][c] in 64 registers for the whole launch, so FK (128 KB per layer) is read
// once per workgroup instead of once per node.  Per node: 16 rows x 32 float4-columns gather the k in-edge
// kernels (streamed from HBM, 64 KB per node) and the senders' features (L2 / Infinity Cache), products are
// rounded and summed in slot order like messages -> index_add_, the 16 x C tile goes through LDS for the
// depth-wise 16x16 orientation mix.  Two LDS tiles alternate so there is one barrier per node.
// ---------------------------------------------------------------------------------------------
#define CONV_LDS_STRIDE 132

#define CONV_GROUP 32
template <int C>
__global__ __launch_bounds__(512, 4) void conv_kernel(
    const float* __restrict__ kl,        // this layer's kernels [N*k*16][C]
    const int32_t* __restrict__ deg, const int32_t* __restrict__ src,
    const float* __restrict__ x_in,      // [N][16][C]
    const float* __restrict__ fk,        // [16(o)][16(p)][C]
    const float* __restrict__ conv_bias, int n0, int N /* receivers n0 .. n0 + N - 1 */, int k,
    float* __restrict__ x_conv)          // [N][16][C]
{
    static_assert(C == 128, "thread mapping assumes C = 128");
    __shared__ __attribute__((aligned(16))) float tile[2][16 * CONV_LDS_STRIDE];
    const int tid = threadIdx.x;
    const int c = tid & 127, pq = tid >> 7;  // conv role: channel, quarter of the output orientations
    const int c4 = tid & 31, o_row = tid >> 5;  // gather role: float4 column, orientation row
    float fkr[16][4];
#pragma unroll
    for (int o = 0; o < 16; ++o)
#pragma unroll
        for (int p = 0; p < 4; ++p) fkr[o][p] = fk[((size_t)o * 16 + (4 * pq + p)) * C + c];
    const float bias = conv_bias[c];

    // XCD-aware node order.  Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 shares an L2); a node's
    // neighbours are atoms of its own crystal, i.e. nearby node indices.  Blocks of CONV_GROUP consecutive nodes are
    // dealt to the XCDs, so the x rows an XCD gathers are (almost) only those of its own node blocks and are fetched
    // into that L2 once instead of into all eight.  With fewer than 8 workgroups the plain order is kept.
    const bool xcd_order = (gridDim.x & 7) == 0 && N >= 32 * CONV_GROUP;  // small batches: one receiver per workgroup first
    const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
    const int n_iter = xcd_order ? ((N + 8 * CONV_GROUP - 1) / (8 * CONV_GROUP)) * CONV_GROUP : N;  // local node slots per XCD
    int buf = 0;
    for (int m = xcd_order ? wg_in_xcd : (int)blockIdx.x; m < n_iter; m += xcd_order ? wgs_per_xcd : (int)gridDim.x) {
        const int nl = xcd_order ? ((m / CONV_GROUP) * 8 + xcd) * CONV_GROUP + (m % CONV_GROUP) : m;
        if (nl >= N) continue;  // workgroup-uniform
        const int n = n0 + nl;
        buf ^= 1;
        // ---- gather . multiply . ordered sum over the in-edges -------------------------------------
        const int nd = min(deg[n], k);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const size_t kbase = ((size_t)n * k * 16 + o_row) * C + 4 * c4;
        const int32_t* srow = src + (size_t)n * k;
#pragma unroll 1
        for (int s0 = 0; s0 < nd; s0 += 4) {
            // unconditional loads (unused slots are clamped to a valid address and dropped by the select below:
            // their kernel rows may be uninitialised, so they must not be multiplied in)
            f32x4 kv[4], xv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int s = min(s0 + i, k - 1);
                const int sn = max(srow[s], 0);
                kv[i] = *reinterpret_cast<const f32x4*>(kl + kbase + (size_t)s * 16 * C);
                xv[i] = *reinterpret_cast<const f32x4*>(x_in + ((size_t)sn * 16 + o_row) * C + 4 * c4);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool on = s0 + i < nd;  // product rounded, then added in edge order (messages -> index_add_)
                acc[0] = on ? __fadd_rn(acc[0], __fmul_rn(kv[i][0], xv[i][0])) : acc[0];
                acc[1] = on ? __fadd_rn(acc[1], __fmul_rn(kv[i][1], xv[i][1])) : acc[1];
                acc[2] = on ? __fadd_rn(acc[2], __fmul_rn(kv[i][2], xv[i][2])) : acc[2];
                acc[3] = on ? __fadd_rn(acc[3], __fmul_rn(kv[i][3], xv[i][3])) : acc[3];
            }
        }
        *reinterpret_cast<f32x4*>(&tile[buf][o_row * CONV_LDS_STRIDE + 4 * c4]) = acc;
        __syncthreads();  // tile[buf] complete; the previous node's readers of tile[buf^1] are long done
        // ---- depth-wise orientation mix ------------------------------------------------------------
        float out[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < 16; ++o) {
            const float xv = tile[buf][o * CONV_LDS_STRIDE + c];
#pragma unroll
            for (int p = 0; p < 4; ++p) out[p] += xv * fkr[o][p];
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) x_conv[((size_t)n * 16 + (4 * pq + p)) * C + c] = out[p] + bias;
    }
}

// ---------------------------------------------------------------------------------------------
// K4a, streamed form (k = 8): the same arithmetic as conv_kernel, but the receiver's K block (8 slots x 16
// orientations x C floats = 64 KiB, contiguous in HBM) is copied straight into LDS by LDS-DMA two receivers ahead
// -- no registers are tied up by bytes in flight, so a whole 64 KiB HBM request per receiver overlaps the arithmetic
// of the two receivers before it -- and the senders' x rows are requested (all 8 slots at once) two receivers
// ahead as well, into a second register set.  One workgroup of 512 threads per CU (LDS: 2 x 64 KiB K
// blocks + 2 tiles).  Loads and copies are inline asm with hand-counted vmcnt (loads, stores and DMA retire in issue
// order): at the top of receiver i the queue ends with [x(i+1): 8] [K(i+1) copy: 8 per wave] [4 stores of receiver
// i-1] -> vmcnt(20) retires what receiver i needs and leaves the rest in flight.
// ---------------------------------------------------------------------------------------------
// K is read exactly once, by one CU: the streaming hint keeps it from displacing the node features in L2 / Infinity Cache
// (measured at 256 x 20, tools/exp/sweep_libs.sh: conv 81.7 -> 71.5 us, the ConvNext kernel behind it 82 -> 79.5 us, the edge
// kernel of the next step 684 -> 700 us; sc0 / sc1 scopes change nothing).  -DARREAU_K_LOAD_POL=0: default policy.
#if !defined(ARREAU_K_LOAD_POL) || ARREAU_K_LOAD_POL == 1
#define ARREAU_K_LOAD_POLICY " nt"
#else
#define ARREAU_K_LOAD_POLICY ""
#endif
__device__ __forceinline__ void conv_glds16(const void* gsrc_lane, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ARREAU_K_LOAD_POLICY : : "v"(gsrc_lane), "s"(lds_dst) : "memory", "m0");
}
__device__ __forceinline__ f32x4 conv_load16(const void* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// stores as asm as well: the counted waits assume EXACTLY four store instructions per receiver
template <int IMM>
__device__ __forceinline__ void conv_store4(const float* base_uniform, unsigned lane_off, float v) {
    asm volatile("global_store_dword %0, %1, %2 offset:%3" : : "v"(lane_off), "v"(v), "s"(base_uniform), "n"(IMM) : "memory");
}
template <int C, bool K3 /* K as 3-byte floats (internal.h: "K stash format") */>
__global__ __launch_bounds__(512, 2) void conv_kernel_streamed(
    const float* __restrict__ kl,        // this layer's kernels [N*8*16][C] (fp32, or 3 bytes per value)
    const int32_t* __restrict__ deg, const int32_t* __restrict__ src,
    const float* __restrict__ x_in,      // [N][16][C]
    const float* __restrict__ fk,        // [16(o)][16(p)][C]
    const float* __restrict__ conv_bias, int n0, int N /* receivers n0 .. n0 + N - 1 (arrays are whole-batch, indices absolute) */,
    float* __restrict__ x_conv)          // [N][16][C]
{
    static_assert(C == 128, "thread mapping assumes C = 128");
    constexpr int K = 8;
    constexpr unsigned KBLOCK = K * 16 * C * (K3 ? 3 : 4);  // 64 KiB (48 KiB) per receiver
    constexpr int NDMA = KBLOCK / 8 / 1024;                 // 1 KiB copies per wave and receiver: 8 (6)
    __shared__ __attribute__((aligned(16))) float kbuf_s[2][KBLOCK / 4];
    __shared__ __attribute__((aligned(16))) float tile[2][16 * CONV_LDS_STRIDE];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int c = tid & 127, pq = tid >> 7;    // mix role: channel, quarter of the output orientations
    const int c4 = tid & 31, o_row = tid >> 5;  // gather role: float4 column, orientation row
    float fkr[16][4];
#pragma unroll
    for (int o = 0; o < 16; ++o)
#pragma unroll
        for (int p = 0; p < 4; ++p) fkr[o][p] = fk[((size_t)o * 16 + (4 * pq + p)) * C + c];
    float bias = conv_bias[c];
    // Consume the loads above here, outside the loop: hipcc then waits for them now, and not -- with the vmcnt(0) it
    // would have to use, since it cannot see the asm requests -- in front of their first use inside the loop.
#pragma unroll
    for (int o = 0; o < 16; ++o)
#pragma unroll
        for (int p = 0; p < 4; ++p) asm volatile("" : "+v"(fkr[o][p]));
    asm volatile("" : "+v"(bias));

    // XCD-aware receiver order (see conv_kernel); the i-th receiver of this workgroup, or -1
    const bool xcd_order = (gridDim.x & 7) == 0 && N >= 32 * CONV_GROUP;  // small batches: one receiver per workgroup first
    const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
    const int n_iter = xcd_order ? ((N + 8 * CONV_GROUP - 1) / (8 * CONV_GROUP)) * CONV_GROUP : N;
    const int m_step = xcd_order ? wgs_per_xcd : (int)gridDim.x;
    auto local_of = [&](int m) { return xcd_order ? ((m / CONV_GROUP) * 8 + xcd) * CONV_GROUP + (m % CONV_GROUP) : m; };
    auto node_of = [&](int m) { return n0 + local_of(m); };
    auto next_valid = [&](int m) {  // first local index >= m with a receiver inside the range, or n_iter
        while (m < n_iter && local_of(m) >= N) m += m_step;
        return m;
    };
    const unsigned kb0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)&kbuf_s[0][0]);
    auto copy_k = [&](int n, int b) {  // this wave's eighth of receiver n's K block -> kbuf_s[b]
        const char* g = reinterpret_cast<const char*>(kl) + (size_t)n * KBLOCK + (1024u * NDMA) * wave + 16u * lane;
#pragma unroll
        for (int i = 0; i < NDMA; ++i) conv_glds16(g + 1024 * i, kb0 + b * KBLOCK + (1024u * NDMA) * wave + 1024u * i);
    };
    f32x4 xv[2][K];  // x rows of the current and of the next receiver (static parity: the loop body is unrolled by two)
    auto load_x = [&](int n, auto par) {
        const int32_t* srow = src + (size_t)n * K;
#pragma unroll
        for (int s_ = 0; s_ < K; ++s_) {
            const int sn = max(srow[s_], 0);  // unused slots: any valid row, dropped by the select below
            xv[decltype(par)::value][s_] = conv_load16(x_in + ((size_t)sn * 16 + o_row) * C + 4 * c4);
        }
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;

    int m = next_valid(xcd_order ? wg_in_xcd : (int)blockIdx.x);
    if (m >= n_iter) return;
    int mn = next_valid(m + m_step);
    // prologue: queue = [x(0)] [K(0)] [x(1)] [K(1)]
    load_x(node_of(m), P0{});
    copy_k(node_of(m), 0);
    if (mn < n_iter) {
        load_x(node_of(mn), P1{});
        copy_k(node_of(mn), 1);
    }
    bool first = true;
    // One receiver.  PAR = parity of its position in this workgroup's sequence = its K buffer, tile and xv set.
    auto receiver = [&](auto par) {
        constexpr int PAR = decltype(par)::value;
        const int n = node_of(m);
        const int mnn = mn < n_iter ? next_valid(mn + m_step) : n_iter;  // the receiver after the next
        const int nd = min(deg[n], K);                                   // scalar load
        // Requests are issued after the second barrier of a receiver as [x(+2): 8] [K(+2): 8] and followed by its
        // 4 stores.  Receiver i needs x(i), K(i); younger than those are the requests for i+1 (16, if it exists) and
        // the stores of i-1 (4, unless i is the first): they stay in flight.
#ifdef ARREAU_DEBUG_WAIT_ALL  // debug build: every counted wait becomes vmcnt(0); outputs must not change
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
        if (mn < n_iter) {  // (8 x rows + NDMA K copies of the next receiver; + the 4 stores of the previous one)
            if (first) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 + NDMA) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 + NDMA + 4) : "memory");
        } else {
            if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
#endif
#pragma unroll
        for (int s_ = 0; s_ < K; ++s_) asm volatile("; landed %0" : "+v"(xv[PAR][s_]));  // the loaded values exist from here on
        // ("; landed" is the marker tools/isa_lint.py looks for: the build fails if the compiler touched a destination
        // register of an asm load between the load and this point)
        __syncthreads();  // every wave's share of K(n) has landed
        // ---- multiply . ordered sum over the in-edges -----------------------------------------------------
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* kb = &kbuf_s[PAR][K3 ? o_row * (C * 3 / 4) + 3 * c4 : o_row * C + 4 * c4];
#pragma unroll
        for (int s_ = 0; s_ < K; ++s_) {
            float kv[4];
            if constexpr (K3) {
                const unsigned* kq = reinterpret_cast<const unsigned*>(kb) + s_ * (16 * C * 3 / 4);
                arreau_unpack_k3(kq[0], kq[1], kq[2], kv);
            } else {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(kb + s_ * 16 * C);
                kv[0] = kf[0]; kv[1] = kf[1]; kv[2] = kf[2]; kv[3] = kf[3];
            }
            const bool on = s_ < nd;  // product rounded, then added in edge order (messages -> index_add_)
            acc[0] = on ? __fadd_rn(acc[0], __fmul_rn(kv[0], xv[PAR][s_][0])) : acc[0];
            acc[1] = on ? __fadd_rn(acc[1], __fmul_rn(kv[1], xv[PAR][s_][1])) : acc[1];
            acc[2] = on ? __fadd_rn(acc[2], __fmul_rn(kv[2], xv[PAR][s_][2])) : acc[2];
            acc[3] = on ? __fadd_rn(acc[3], __fmul_rn(kv[3], xv[PAR][s_][3])) : acc[3];
        }
        *reinterpret_cast<f32x4*>(&tile[PAR][o_row * CONV_LDS_STRIDE + 4 * c4]) = acc;
        __syncthreads();  // tile[PAR] complete; nobody reads kbuf_s[PAR] or xv[PAR] any more
        // ---- requests for the receiver after the next (same parity) -----------------------------------------
        if (mnn < n_iter) {
            load_x(node_of(mnn), par);
            copy_k(node_of(mnn), PAR);
        }
        // ---- depth-wise orientation mix ------------------------------------------------------------
        float out[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < 16; ++o) {
            const float xo = tile[PAR][o * CONV_LDS_STRIDE + c];
#pragma unroll
            for (int p = 0; p < 4; ++p) out[p] += xo * fkr[o][p];
        }
        {
            const float* xbase = x_conv + (size_t)n * 16 * C;  // wave-uniform
            const unsigned loff = 4u * ((4 * pq) * C + c);
            conv_store4<0>(xbase, loff, out[0] + bias);
            conv_store4<4 * C>(xbase, loff, out[1] + bias);
            conv_store4<8 * C>(xbase, loff, out[2] + bias);
            conv_store4<12 * C>(xbase, loff, out[3] + bias);
        }
        first = false;
        m = mn;
        mn = mnn;
    };
    while (m < n_iter) {
        receiver(P0{});
        if (m >= n_iter) break;
        receiver(P1{});
    }
    // (round 4, lint rule ldsdma-unwaited-exit: no LDS-DMA copy is left in flight when a wave ends -- the last copies of a ring
    // target a chunk nobody will read; the hardware's implicit wait at s_endpgm is not relied upon)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------
// K4b: ConvNext block of one layer on MFMA (convnext.py:25-32) + read-out partials (ponita.py:105-117).
// A 32-row tile (2 nodes x 16 orientations) is shared by a PAIR of waves, each owning half of the hidden
// units (finer tasks balance the 1024 SIMDs at small batches); the pair meets once, through LDS, to add
// the two partial outputs in a fixed order.  Each wave loads the rows of
// the conv output straight into the B-operand register layout (lane (h, j): row j, channels
// 32t + 8q + 4h + m), applies LayerNorm in registers (a row lives in lanes j and j+32), then walks the
// layer's weight stream quarter by quarter (quarter = 128 hidden units):
//     hid  = GELU(W1[quarter] . xn + b1[quarter])      out-tile-major, fragments prefetched through the ring
//     out += W2[:, quarter] . hid
// and finishes in registers:
//     x_out = (out + b2) * layer_scale + x_in ;  xbar_l[n][c] = mean_o x_out (16-lane butterfly) ;
//     vsum[n][o] (+)= w_vec_l . x_out[n,o,:] + b_vec_l
// ---------------------------------------------------------------------------------------------
template <int C, int H>
__global__ __launch_bounds__(256, 2) void mlp_kernel(
    const float* __restrict__ x_conv,    // [N][16][C]  conv output (pre-LayerNorm)
    const float* __restrict__ x_in, float* __restrict__ x_out,
    const float* __restrict__ ln_w, const float* __restrict__ ln_b,
    const float* __restrict__ mlp,       // this layer's stream: [4 quarters][W1 quarter | W2 quarter]
    const float* __restrict__ mb1, const float* __restrict__ mb2, const float* __restrict__ ls,
    const float* __restrict__ ro_wT,     // [C][S+4] this layer
    const float* __restrict__ ro_b,      // [S+4]
    int S, int N, int first_layer,
    float* __restrict__ xbar,            // [N][C] this layer
    float* __restrict__ vsum)            // [N][16]
{
    constexpr int TC = C / 32;          // in/out tiles of C
    constexpr int HQ = H / 4;           // hidden units per quarter
    constexpr int THQ = HQ / 32;        // hidden tiles per quarter
    constexpr int GA = TC * 4, GB = THQ * 4;
    constexpr int FA = THQ * GA, FB = TC * GB;  // groups in the W1 / W2 part of a quarter
    static_assert(FA % ARREAU_PF == 0 && FB % ARREAU_PF == 0, "ring phase must repeat");
    __shared__ __attribute__((aligned(16))) float part[2][32 * 132];
    const int lane = threadIdx.x & 63;
    const int h = lane >> 5, j = lane & 31;
    const int wave = threadIdx.x >> 6;
    const int tw = wave >> 1, half = wave & 1;  // tile within the workgroup, hidden half
    const long long tile = (long long)blockIdx.x * 2 + tw;
    const long long n_ll = 2 * tile + (j >> 4);
    const bool tile_live = 2 * tile < N;  // both waves of a pair agree; dead pairs still reach the barrier
    const bool valid = n_ll < N;
    const int n = valid ? (int)n_ll : N - 1;  // padding rows / dead tiles read a valid row and write nothing
    const int o = j & 15;

    // weight stream of this layer (this wave starts at its first quarter): start it first
    const float* sp0 = mlp + lane * 4;
    const float* sp = sp0 + (size_t)(2 * half) * (FA + FB) * 256;
    f32x4 ring[ARREAU_PF];
#pragma unroll
    for (int i = 0; i < ARREAU_PF; ++i) ring[i] = *reinterpret_cast<const f32x4*>(sp + (size_t)i * 256);

    // ---- load the row in B-operand layout and LayerNorm it (eps 1e-5, biased variance) ---------------
    const size_t rowoff = ((size_t)n * 16 + o) * C + 4 * h;
    f32x16 bx[TC][1];
    {
        const float* rowp = x_conv + rowoff;
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < TC; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(rowp + 32 * t + 8 * q);
                bx[t][0][4 * q] = v[0]; bx[t][0][4 * q + 1] = v[1]; bx[t][0][4 * q + 2] = v[2]; bx[t][0][4 * q + 3] = v[3];
                sum += (v[0] + v[1]) + (v[2] + v[3]);
            }
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * (1.0f / C);
        float sq = 0.f;
#pragma unroll
        for (int t = 0; t < TC; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d = bx[t][0][r] - mean;
                bx[t][0][r] = d;
                sq += d * d;
            }
        sq += __shfl_xor(sq, 32, 64);
        const float rstd = 1.0f / sqrtf(sq * (1.0f / C) + 1e-5f);
#pragma unroll
        for (int t = 0; t < TC; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(ln_w + 32 * t + 8 * q + 4 * h);
                const f32x4 be = *reinterpret_cast<const f32x4*>(ln_b + 32 * t + 8 * q + 4 * h);
#pragma unroll
                for (int m = 0; m < 4; ++m) bx[t][0][4 * q + m] = bx[t][0][4 * q + m] * rstd * g[m] + be[m];
            }
    }

    f32x16 acc_o[TC][1];
#pragma unroll
    for (int u = 0; u < TC; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[u][0][r] = 0.f;

#pragma unroll 1
    for (int w = 2 * half; w < 2 * half + 2; ++w) {
        if (!tile_live) break;
        const float* region_a = sp0 + (size_t)w * (FA + FB) * 256;
        const float* region_b = region_a + (size_t)FA * 256;
        // ---- hid = GELU(W1q . xn + b1q) ---------------------------------------------------------------
        f32x16 acc_h[THQ][1];
#pragma unroll
        for (int u = 0; u < THQ; ++u) {
            f32x16 acc[1];
            acc[0] = arreau_bias_tile(mb1 + w * HQ, u, h);
            __builtin_amdgcn_sched_barrier(0);
            arreau_stream_tile<GA, TC, 1>(acc, ring, region_a, u * GA, bx);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_h[u][0][r] = arreau_gelu(acc[0][r]);
        }
        // ---- out += W2[:, quarter] . hid ----------------------------------------------------------------
#pragma unroll
        for (int u = 0; u < TC; ++u) {
            __builtin_amdgcn_sched_barrier(0);
            arreau_stream_tile<GB, THQ, 1>(acc_o[u], ring, region_b, u * GB, acc_h);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- the upper-half wave hands its partial sum to its partner and retires ----------------------------
    if (half == 1) {
#pragma unroll
        for (int u = 0; u < TC; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = {acc_o[u][0][4 * q], acc_o[u][0][4 * q + 1], acc_o[u][0][4 * q + 2], acc_o[u][0][4 * q + 3]};
                *reinterpret_cast<f32x4*>(&part[tw][j * 132 + 32 * u + 8 * q + 4 * h]) = v;
            }
    }
    __syncthreads();
    if (half == 1 || !tile_live) return;
#pragma unroll
    for (int u = 0; u < TC; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&part[tw][j * 132 + 32 * u + 8 * q + 4 * h]);
#pragma unroll
            for (int m = 0; m < 4; ++m) acc_o[u][0][4 * q + m] += v[m];  // (quarters 0+1) + (quarters 2+3)
        }

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
            for (int m = 0; m < 4; ++m) xo[m] = (acc_o[u][0][4 * q + m] + b2v[m]) * lsv[m] + xi[m];
            if (valid) *reinterpret_cast<f32x4*>(x_out + rowoff + 32 * u + 8 * q) = xo;
            // vector read-out channel (column S of read_out_layers)
#pragma unroll
            for (int m = 0; m < 4; ++m) vdot += xo[m] * ro_wT[(size_t)(c0 + m) * (S + 4) + S];
            // mean over the node's 16 orientations: butterfly over lanes j&15 (same h, same node)
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

int arreau_launch_node_layer(const arreau_model* m, int layer, const float* kbuf, const int32_t* deg,
                             const int32_t* src, const float* x_in, float* x_conv, float* x_out, float* xbar,
                             float* vsum, int N, hipStream_t s, NodeRange r) {
    const int n0 = r.n0, n1 = r.n1 < 0 ? N : r.n1;
    const int Ng = n1 - n0;
    if (Ng <= 0) return ARREAU_OK;
    const bool whole = n0 == 0 && n1 == N;
    const int C = m->C, H = m->H, S = m->S;
    if (!(C == 128 && H == 512)) {
        arreau_set_error("node kernels: unsupported (hidden_dim, widening_factor)");
        return ARREAU_EINVAL;
    }
    const size_t layer_stride = (size_t)N * m->k * 16 * C;
    const size_t mlp_layer = (size_t)2 * H * C;  // floats of W1 + W2, packed
    const int conv_blocks = N < 512 ? N : 512;   // persistent: 2 workgroups of 512 threads per CU (a multiple of 8: XCD-aware order)
    // conv variant: 1 (default, k = 8 only) = streamed form (K blocks by LDS-DMA, one workgroup per CU); 0 = register form
    const bool basis_form = arreau_basis_form(m, N);  // whole-batch decision (one kbuf layout per evaluation)
    const int conv_variant = m->conv_variant == 2 ? 1 : m->conv_variant;  // (2 without the basis form = the streamed K pair)
    m->ran_conv = basis_form ? 2 : (conv_variant == 1 && m->k == 8) ? 1 : 0;
    if (arreau_small_layer_fusable(m, N, r)) {  // small launch: both halves of the layer in one kernel (bit-identical)
        m->ran_mlp = 3;
        return arreau_launch_small_layer(m, layer, kbuf, deg, src, x_in, x_out, xbar, vsum, N, s);
    }
    if (basis_form) {
        const int rc = arreau_launch_conv_proj(m, layer, kbuf, deg, src, x_in, x_conv, N, s, r);
        if (rc) return rc;
    } else if (conv_variant == 1 && m->k == 8) {
        int blocks = Ng < 256 ? Ng : 256;
        if (r.wg_cap > 0 && blocks > r.wg_cap) blocks = r.wg_cap;
        ARREAU_LAUNCH((conv_kernel_streamed<128, false>), dim3(blocks), dim3(512), 0, s, kbuf + (size_t)layer * layer_stride, deg,
                               src, x_in, m->fk + (size_t)layer * 16 * 16 * C, m->conv_bias + (size_t)layer * C, n0, Ng, x_conv);
    } else {
        int blocks = Ng < 512 ? Ng : 512;
        if (r.wg_cap > 0 && blocks > r.wg_cap) blocks = r.wg_cap;
        ARREAU_LAUNCH((conv_kernel<128>), dim3(whole ? conv_blocks : blocks), dim3(512), 0, s, kbuf + (size_t)layer * layer_stride,
                           deg, src, x_in, m->fk + (size_t)layer * 16 * 16 * C, m->conv_bias + (size_t)layer * C, n0, Ng, m->k, x_conv);
    }
    ARREAU_CHECK_HIP(hipGetLastError());
    // variant switch: 3 (default) = fp16x3 on 16x16x32 MFMAs (node_f16m.hip; needs weights that fit fp16); 1 = bf16x6 split-precision
    // MLP kernel (node_bf16.hip); 0 = fp32-MFMA kernel below  (2, the same arithmetic on 32x32x16 MFMAs, was removed in round 5)
    // 4 = always the small-launch form of 3 (one node per workgroup, node_f16m.hip: bit-identical; 3 picks it by size)
    const int mlp_variant = m->mlp_variant;
    if (mlp_variant == 4 && m->f16_ok) {
        m->ran_mlp = 3;
        return arreau_launch_mlp_f16x3_m16_split(m, layer, x_conv, x_in, x_out, xbar, vsum, N, s, r);
    }
    if (mlp_variant == 3 && m->f16_ok) {
        m->ran_mlp = 3;
        return arreau_launch_mlp_f16x3_m16(m, layer, x_conv, x_in, x_out, xbar, vsum, N, s, r);
    }
    if (!whole) {
        arreau_set_error("mlp kernel: range launches are implemented for the fp16x3 16x16x32 kernel only");
        return ARREAU_EINVAL;
    }
    if (mlp_variant >= 1) {
        m->ran_mlp = 1;
        return arreau_launch_mlp_bf16x6(m, layer, x_conv, x_in, x_out, xbar, vsum, N, s);
    }
    m->ran_mlp = 0;
    ARREAU_LAUNCH((mlp_kernel<128, 512>), dim3((N + 3) / 4), dim3(256), 0, s, x_conv, x_in, x_out,
                       m->ln_w + (size_t)layer * C, m->ln_b + (size_t)layer * C, m->mlp + (size_t)layer * mlp_layer,
                       m->mb1 + (size_t)layer * H, m->mb2 + (size_t)layer * C, m->ls + (size_t)layer * C,
                       m->ro_wT + (size_t)layer * C * (S + 4), m->ro_b + (size_t)layer * (S + 4), S, N,
                       layer == 0 ? 1 : 0, xbar + (size_t)layer * N * C, vsum);
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}

// ---------------------------------------------------------------------------------------------
// K5 read-outs (ponita.py:105-117,126-155).  sphere_to_scalar (mean over orientations) commutes
// with the per-layer Linear, so the scalar / global read-outs are taken of xbar_l = mean_o x_l:
//   logits[n][s] = (1/L) sum_l (W_l[s,:] . xbar_l[n] + b_l[s])
//   gs[n][g]     = (1/L) sum_l (W_l[S+1+g,:] . xbar_l[n] + b_l[S+1+g])
//   eps[n][d]    = (1/(L*O)) sum_o vsum[n][o] * ori[o][d]               (sphere_to_vec of the vector channel)
//   len0[b][g]   = sum_{n in b} gs[n][g]                                  (global_add_pool, atom order)
// Kernel 1: 8 atoms per workgroup, thread = output column, weights read once per 8 atoms (coalesced
// over columns).  Kernel 2: one thread per (crystal, g) sums its atoms in order (deterministic).
// ---------------------------------------------------------------------------------------------
#define RO_ATOMS 8
#define RO_COLS 128  // threads per layer group (>= S + 4)

// blockDim = RO_COLS * L: thread (l, s) reduces layer l's 128 channels for output column s and the tile's 8
// atoms; the L partial sums are then added in layer order (deterministic) by the l = 0 group.
__global__ __launch_bounds__(1024) void readout_nodes_kernel(
    const float* __restrict__ xbar,   // [L][N][C]
    const float* __restrict__ vsum,   // [N][16]
    const float* __restrict__ ro_wT,  // [L][C][S+4]
    const float* __restrict__ ro_b,   // [L][S+4]
    const float* __restrict__ ori, int S, int C, int L, int N, float* __restrict__ eps,
    float* __restrict__ logits, float* __restrict__ gs /*[N][3]*/, int32_t* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int RO = S + 4, LC = L * C;
    float* xs = smem;                          // [RO_ATOMS][L*C]
    float* part = smem + RO_ATOMS * LC;        // [L][RO_ATOMS][RO_COLS]
    const int n0 = blockIdx.x * RO_ATOMS;
    for (int i = threadIdx.x; i < RO_ATOMS * LC; i += blockDim.x) {
        const int a = i / LC, r = i - a * LC;
        const int l = r / C, c = r - l * C;
        const int n = n0 + a;
        xs[i] = n < N ? xbar[((size_t)l * N + n) * C + c] : 0.f;
    }
    __syncthreads();
    const int l = threadIdx.x / RO_COLS, s_out = threadIdx.x - l * RO_COLS;
    const float invL = 1.0f / (float)L;
    if (s_out < RO && s_out != S) {
        float acc[RO_ATOMS];
#pragma unroll
        for (int a = 0; a < RO_ATOMS; ++a) acc[a] = 0.f;
        const float* w = ro_wT + (size_t)l * C * RO + s_out;
        for (int c = 0; c < C; c += 4) {
            const float w0 = w[(size_t)c * RO], w1 = w[(size_t)(c + 1) * RO], w2 = w[(size_t)(c + 2) * RO],
                        w3 = w[(size_t)(c + 3) * RO];
#pragma unroll
            for (int a = 0; a < RO_ATOMS; ++a) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(&xs[a * LC + l * C + c]);
                acc[a] += x[0] * w0;
                acc[a] += x[1] * w1;
                acc[a] += x[2] * w2;
                acc[a] += x[3] * w3;
            }
        }
        const float bias = ro_b[l * RO + s_out];
#pragma unroll
        for (int a = 0; a < RO_ATOMS; ++a) part[(l * RO_ATOMS + a) * RO_COLS + s_out] = acc[a] + bias;
    } else if (l == 0 && RO_COLS - RO >= 3 * RO_ATOMS && s_out >= RO && s_out - RO < 3 * RO_ATOMS) {
        // vector channel on the idle columns of the first layer group: eps component d of atom a of the tile
        // (sphere_to_vec of the per-orientation dot products)
        const int a = (s_out - RO) / 3, d = (s_out - RO) - 3 * a;
        const size_t n = (size_t)n0 + a;
        if (n < (size_t)N) {
            float acc = 0.f;
            for (int o = 0; o < 16; ++o) acc += (vsum[n * 16 + o] * invL) * ori[3 * o + d];
            eps[n * 3 + d] = acc * (1.0f / 16.0f);
            if (!(fabsf(acc) < INFINITY)) atomicOr(status, ARREAU_STATUS_NONFINITE);
        }
    } else if (l == 0 && RO_COLS - RO < 3 * RO_ATOMS && s_out == S) {
        // (wide species tables leave too few idle columns: one thread walks the tile)
        for (int a = 0; a < RO_ATOMS; ++a) {
            const size_t n = (size_t)n0 + a;
            if (n >= (size_t)N) break;
            for (int d = 0; d < 3; ++d) {
                float acc = 0.f;
                for (int o = 0; o < 16; ++o) acc += (vsum[n * 16 + o] * invL) * ori[3 * o + d];
                eps[n * 3 + d] = acc * (1.0f / 16.0f);
                if (!(fabsf(acc) < INFINITY)) atomicOr(status, ARREAU_STATUS_NONFINITE);
            }
        }
    }
    __syncthreads();
    if (l == 0 && s_out < RO && s_out != S) {
#pragma unroll
        for (int a = 0; a < RO_ATOMS; ++a) {
            const size_t n = (size_t)n0 + a;
            if (n >= (size_t)N) break;
            float tot = 0.f;
            for (int ll = 0; ll < L; ++ll) tot += part[(ll * RO_ATOMS + a) * RO_COLS + s_out];
            tot *= invL;
            if (!(fabsf(tot) < INFINITY)) atomicOr(status, ARREAU_STATUS_NONFINITE);
            if (s_out < S) logits[n * S + s_out] = tot;
            else gs[n * 3 + (s_out - S - 1)] = tot;
        }
    }
}

// Read-out on the fp32 matrix pipe: workgroup = 32 atoms, wave l = layer l (readout_dev.h).  USPLIT = 1 (small launches):
// gridDim.y = ROT and a workgroup computes only output tile blockIdx.y (bit-identical).
template <int C, int ROT /* output tiles */, int USPLIT = 0>
__global__ __launch_bounds__(512) void readout_mfma_kernel(
    const float* __restrict__ xbar,     // [L][N][C]
    const float* __restrict__ vsum,     // [N][16]
    const float* __restrict__ ro_pack,  // [L][ROT][C/32][1024]
    const float* __restrict__ ro_b,     // [L][S+4]
    const float* __restrict__ ori, int S, int L, int Ntot /* batch size: strides xbar */, int nbeg,
    int N /* atoms nbeg .. N-1 */, float* __restrict__ eps, float* __restrict__ logits,
    float* __restrict__ gs /*[N][3]*/, int32_t* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) float part[];  // [L][ROT][64 lanes][16]
    arreau_readout_tile<C, ROT, USPLIT>(part, xbar, vsum, ro_pack, ro_b, ori, S, L, Ntot, nbeg + (int)blockIdx.x * 32, N,
                                        USPLIT ? (int)blockIdx.y : 0, eps, logits, gs, status);
}

__global__ void readout_crystals_kernel(const float* __restrict__ gs, const int32_t* __restrict__ offsets, int b0, int B,
                                        float* __restrict__ len0) {
    const int idx = 3 * b0 + blockIdx.x * blockDim.x + threadIdx.x;  // crystals b0 .. B-1
    if (idx >= 3 * B) return;
    const int b = idx / 3, g = idx - 3 * b;
    float acc = 0.f;
    for (int n = offsets[b]; n < offsets[b + 1]; ++n) acc += gs[(size_t)n * 3 + g];
    len0[idx] = acc;
}

// (Round 5: the 3-byte K stash of round 2 -- ARREAU_K3, the K pair's format when it ran at every size -- is gone with that mode:
// the K pair now only serves launches too small for the basis form, on an fp32 K buffer, bit-identical to the basis form.)
bool arreau_k3(const arreau_model*) { return false; }

// conv_variant 2 (default): basis form wherever it applies -- the fused shape, k = 8, fp16-representable weights, the
// split-precision edge kernel -- and a launch large enough for it to pay: measured at n = 20 (graph replay, one box), the
// basis form against the K pair on fp32 K: 960 receivers 0.436 / 0.421 ms per step, 1920: 0.655 / 0.656, 2560: 0.852 /
// 0.870, 3520: 1.059 / 1.078, 5120: 1.35 / 1.48 -- so it takes over from 2,000 receivers (ARREAU_BASIS_MIN_RECEIVERS
// moves the switch: the tests set 240 to run the basis form on small batches).  Below that the K pair runs on an fp32 K
// buffer (and at most 240 receivers: the tile-per-workgroup kernels); all of them evaluate the same numbers.
bool arreau_basis_form(const arreau_model* m, int receivers) {
    const char* e = getenv("ARREAU_BASIS_MIN_RECEIVERS");
    const int min_receivers = e ? atoi(e) : 2000;
    // (L >= 2: the stash -- 96 or 128 KiB per atom -- lives in the workspace region sized for L per-layer K buffers of 64 KiB per atom)
    return m->conv_variant == 2 && m->edge_variant == 4 && m->f16_ok && m->k == 8 && m->C == 128 && m->D == 256 && m->L >= 2 &&
           receivers > (min_receivers > 240 && !m->calibrating ? min_receivers : 240);
}

bool arreau_basis_fp8(const arreau_model* m) {
    static const int env = [] { const char* e = getenv("ARREAU_BASIS_FP8"); return e ? atoi(e) : -1; }();
    // (the environment overrides the model's calibration either way -- A/B, tests -- except while that calibration runs)
    return env >= 0 && !m->calibrating ? env != 0 : m->fp8_ok != 0;
}

bool arreau_cross_fp8(const arreau_model* m) {
    const char* e = getenv("ARREAU_CROSS_FP8");
    return (e == nullptr || atoi(e) != 0 || m->calibrating) && m->x8_ok && arreau_basis_fp8(m);
}

bool arreau_range_launches_supported(const arreau_model* m) {
    return m->edge_variant == 4 && (m->mlp_variant == 3 || m->mlp_variant == 4) && m->f16_ok && (m->conv_variant == 1 || m->conv_variant == 2) && m->k == 8 &&
           m->readout_variant == 1 && m->S + 4 <= 96 && m->L <= 8 && m->C == 128;
}

int arreau_launch_readout(const arreau_model* m, const float* xbar, const float* vsum, const int32_t* offsets, int B,
                          int N, float* gs, float* eps, float* logits, float* len0, hipStream_t s, NodeRange r) {
    if (B == 0) return ARREAU_OK;
    const int n0 = r.n0, n1 = r.n1 < 0 ? N : r.n1, b0 = r.b0, b1 = r.b1 < 0 ? B : r.b1;
    const bool whole = n0 == 0 && n1 == N;
    if (n1 > n0) {
        const size_t smem = ((size_t)RO_ATOMS * m->L * m->C + (size_t)m->L * RO_ATOMS * RO_COLS) * sizeof(float);
        if (m->L * RO_COLS > 1024 || m->S + 4 > RO_COLS) {
            arreau_set_error("readout kernel: num_layers * 128 threads must fit one workgroup");
            return ARREAU_EINVAL;
        }
        // read-out variant: 1 (default) = fp32-MFMA kernel (needs S + 4 <= 96, L <= 8); 0 = vector kernel
        const int ro_variant = m->readout_variant;
        if (ro_variant == 1 && m->S + 4 <= 96 && m->L <= 8 && m->C == 128) {
            const size_t smem_m = (size_t)m->L * 3 * 64 * 16 * sizeof(float);
            static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&readout_mfma_kernel<128, 3>),
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 3 * 64 * 16 * 4);
            ARREAU_CHECK_HIP(attr);
            static const hipError_t attr_s = hipFuncSetAttribute(reinterpret_cast<const void*>(&readout_mfma_kernel<128, 3, 1>),
                                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 3 * 64 * 16 * 4);
            ARREAU_CHECK_HIP(attr_s);
            const unsigned blocks32 = (unsigned)((n1 - n0 + 31) / 32);
            static const int split_env = [] { const char* e = getenv("ARREAU_READOUT_SPLIT"); return e ? atoi(e) : -1; }();
            // unsliced launch of fewer workgroups than the chip has CUs: one workgroup per output tile (measured, round 3: 160
            // workgroups -- 256 x 20 -- 18.7 us split against 21.6; 640 -- 1024 x 20 -- 75 against 65; 2048: 220 against 169)
            if (split_env >= 0 ? split_env != 0 : (whole && r.wg_cap == 0 && blocks32 < 256))
                ARREAU_LAUNCH((readout_mfma_kernel<128, 3, 1>), dim3(blocks32, 3), dim3(64 * m->L), smem_m, s, xbar, vsum,
                                   m->ro_pack, m->ro_b, m->ori, m->S, m->L, N, n0, n1, eps, logits, gs, m->status);
            else
                ARREAU_LAUNCH((readout_mfma_kernel<128, 3>), dim3(blocks32), dim3(64 * m->L), smem_m, s, xbar, vsum,
                                   m->ro_pack, m->ro_b, m->ori, m->S, m->L, N, n0, n1, eps, logits, gs, m->status);
        } else {
            if (!whole) {
                arreau_set_error("read-out: range launches are implemented for the MFMA kernel only");
                return ARREAU_EINVAL;
            }
            ARREAU_LAUNCH(readout_nodes_kernel, dim3((N + RO_ATOMS - 1) / RO_ATOMS), dim3(RO_COLS * m->L), smem, s, xbar, vsum,
                               m->ro_wT, m->ro_b, m->ori, m->S, m->C, m->L, N, eps, logits, gs, m->status);
        }
        ARREAU_CHECK_HIP(hipGetLastError());
    }
    if (b1 > b0 && len0 != nullptr)  // (len0 == nullptr: the caller pools the crystals itself -- the sampling loop's lattice update)
        ARREAU_LAUNCH(readout_crystals_kernel, dim3((3 * (b1 - b0) + 127) / 128), dim3(128), 0, s, gs, offsets, b0, b1, len0);
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}
