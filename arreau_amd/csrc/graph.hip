// Lattice algebra and the periodic-boundary neighbour list (HBM/latency-bound integer+fp32 work).
#include "internal.h"
#include "embed_dev.h"
#include "graph_dev.h"

// ---------------------------------------------------------------------------------------------
// lattice_from_params (diffusion/lattice_helpers.py:55-105): rows a, b, c of the cell.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void arreau_cell_from_params(const float* len, const float* ang, float* Lm) {
    const float a = len[0], b = len[1], c = len[2];
    const float ca = cosf(ang[0]), cb = cosf(ang[1]), cg = cosf(ang[2]);
    const float sa = sinf(ang[0]), sb = sinf(ang[1]);
    float val = (ca * cb - cg) / (sa * sb);
    val = fminf(fmaxf(val, -1.0f), 1.0f);  // abs_cap, lattice_helpers.py:38-51
    const float gs = acosf(val);
    Lm[0] = a * sb;             Lm[1] = 0.0f;               Lm[2] = a * cb;
    Lm[3] = -b * sa * cosf(gs); Lm[4] = b * sa * sinf(gs);  Lm[5] = b * ca;
    Lm[6] = 0.0f;               Lm[7] = 0.0f;               Lm[8] = c;
}

__global__ void lattice_from_params_kernel(const float* __restrict__ lengths, const float* __restrict__ angles,
                                           int B, float* __restrict__ lattice) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float Lm[9];
    arreau_cell_from_params(lengths + 3 * b, angles + 3 * b, Lm);
#pragma unroll
    for (int i = 0; i < 9; ++i) lattice[9 * b + i] = Lm[i];
}

extern "C" int arreau_lattice_from_params(const float* d_lengths, const float* d_angles, int32_t B,
                                          float* d_lattice, void* stream) {
    ARREAU_REQUIRE(d_lengths && d_angles && d_lattice && B >= 0, "arreau_lattice_from_params: bad argument");
    if (B == 0) return ARREAU_OK;
    ARREAU_LAUNCH(lattice_from_params_kernel, dim3((B + 127) / 128), dim3(128), 0, (hipStream_t)stream,
                       d_lengths, d_angles, B, d_lattice);
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}

// frac_to_cart_coords (diffusion/diffusion_helpers.py:223-230): x_j = sum_i frac_i * L[i][j]
__device__ __forceinline__ void arreau_frac_to_cart(const float* f, const float* Lm, float* x) {
#pragma unroll
    for (int j = 0; j < 3; ++j) x[j] = (f[0] * Lm[j] + f[1] * Lm[3 + j]) + f[2] * Lm[6 + j];
}

__global__ void frac_to_cart_kernel(const float* __restrict__ frac, const float* __restrict__ lattice,
                                    const int32_t* __restrict__ offsets, int B, int N, float* __restrict__ cart) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int b = arreau_find_crystal(offsets, B, i);
    float x[3];
    arreau_frac_to_cart(frac + 3 * i, lattice + 9 * b, x);
    cart[3 * i + 0] = x[0]; cart[3 * i + 1] = x[1]; cart[3 * i + 2] = x[2];
}

extern "C" int arreau_frac_to_cart(const float* d_frac, const float* d_lattice, const int32_t* d_off, int32_t B,
                                   int32_t N, float* d_cart, void* stream) {
    ARREAU_REQUIRE(d_frac && d_lattice && d_off && d_cart && B >= 1 && N >= 0, "arreau_frac_to_cart: bad argument");
    if (N == 0) return ARREAU_OK;
    ARREAU_LAUNCH(frac_to_cart_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_frac,
                       d_lattice, d_off, B, N, d_cart);
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}

// ---------------------------------------------------------------------------------------------
// K1: periodic radius graph with per-receiver top-k  (diffusion/diffusion_helpers.py:328-564).
// One wave per receiver atom.  Candidates are (sender j, image cell) with enumeration index
// c = 27*j_local + cell (the reference's order: sender-minor inside a receiver, 27 images in
// itertools.product((-1,0,1),repeat=3) order, :377-402).  A candidate survives when
// 1e-4 < d2 <= R^2 (:432-436).  The k smallest by (d2, c) are found by k rounds of a wave-wide
// arg-min over a key that orders (d2 bits, c) lexicographically (held as a double, see below; the
// wave minimum by DPP): deterministic, no atomics.  Slots are written in ascending c, i.e. the
// order the reference's masked_select leaves them in.  fp32 arithmetic mirrors the reference's
// op order without FMA contraction so that the selected set equals the fp32 CPU path's on
// tie-free input.  Per wave, LDS holds a copy of the crystal's positions (up to 128 atoms) and
// the table of the 27 image offsets (round 3: the kernel is bound by the vector work of the
// candidate evaluations and of the selection rounds, not by memory).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void neighbor_kernel(
    const float* __restrict__ cart, const float* __restrict__ lattice, const int32_t* __restrict__ offsets,
    const int32_t* __restrict__ batch, int B, int n0, int N, float r2, int k, int32_t* __restrict__ deg, int32_t* __restrict__ src,
    int32_t* __restrict__ cell, float* __restrict__ dir, float* __restrict__ dist) {
    arreau_neighbor_body(blockIdx.x, cart, lattice, offsets, batch, B, n0, N, r2, k, deg, src, cell, dir, dist);
}

// Round 3: the sampler's step builds the neighbour list and the embedded node features in ONE launch.  Both depend on prep_kernel
// only and not on each other; the embedding is bound by its HBM writes (16 rows of C floats per atom), the neighbour list by
// latency and VALU work, so side by side they take the time of the longer one.  The first `embed_blocks` workgroups embed.
// LOOP (sampling loop, round 3): no prep launch in the step.  `cart` holds the fractional coordinates (positions are formed by
// the waves that need them, with prep_kernel's expression); the lattice and the per-crystal embedding were left by the
// previous step's update launch (reverse_crystal_block) or by the one prep launch in front of the loop; and the first
// workgroups advance the device-side timestep of every crystal by one (`tick`: read only by the update launch).
template <bool LOOP = false>
__global__ __launch_bounds__(256) void neighbor_embed_kernel(
    unsigned embed_blocks, const float* __restrict__ cart, const float* __restrict__ lattice, const int32_t* __restrict__ offsets,
    const int32_t* __restrict__ batch, int B, int n0, int N, float r2, int k, int32_t* __restrict__ deg, int32_t* __restrict__ src,
    int32_t* __restrict__ cell, float* __restrict__ dir, float* __restrict__ dist, const float* __restrict__ frac,
    const int32_t* __restrict__ types, const float* __restrict__ cvec, const float* __restrict__ ori, const float* __restrict__ embT,
    int S, int C, float* __restrict__ x0, int32_t* __restrict__ status, int32_t* __restrict__ tick, int tick_b0, int tick_b1) {
    if constexpr (LOOP) {
        const int b = tick_b0 + (int)(blockIdx.x * blockDim.x + threadIdx.x);
        if (b < tick_b1) tick[b] -= 1;  // (nobody else reads or writes it in this launch)
    }
    if (blockIdx.x < embed_blocks) {
        arreau_embed_body(blockIdx.x * blockDim.x + threadIdx.x, frac, types, lattice, batch, cvec, ori, embT, S, C, n0, N, x0, status);
        return;
    }
    arreau_neighbor_body<LOOP>(blockIdx.x - embed_blocks, cart, lattice, offsets, batch, B, n0, N, r2, k, deg, src, cell, dir, dist);
}

int arreau_launch_neighbor(const float* cart, const float* lattice, const int32_t* offsets, const int32_t* batch,
                           int B, int N, float radius, int k, int32_t* deg, int32_t* src, int32_t* cell, float* dir, float* dist,
                           hipStream_t s, NodeRange r) {
    const int n0 = r.n0, n1 = r.n1 < 0 ? N : r.n1;
    if (n1 <= n0) return ARREAU_OK;
    const float r2 = (float)((double)radius * (double)radius);
    const int waves_per_block = 4;
    ARREAU_LAUNCH(neighbor_kernel, dim3((n1 - n0 + waves_per_block - 1) / waves_per_block), dim3(64 * waves_per_block),
                       0, s, cart, lattice, offsets, batch, B, n0, n1, r2, k, deg, src, cell, dir, dist);
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}

int arreau_launch_neighbor_embed(const arreau_model* m, const float* cart, const float* lattice, const int32_t* offsets,
                                 const int32_t* batch, int B, int N, int32_t* deg, int32_t* src, int32_t* cell, float* dir,
                                 float* dist, const float* frac, const int32_t* types, const float* cvec, float* x0, hipStream_t s,
                                 NodeRange r, int32_t* tick) {
    const int n0 = r.n0, n1 = r.n1 < 0 ? N : r.n1;
    if (n1 <= n0 && tick == nullptr) return ARREAU_OK;
    ARREAU_REQUIRE(batch != nullptr, "neighbour list + embedding: the atom -> crystal map is required");
    const float r2 = (float)((double)m->cfg.radius * (double)m->cfg.radius);
    const long long pairs = (long long)(n1 - n0) * (m->C / 4);
    if (pairs >= (1ll << 31)) {
        arreau_set_error("embed kernel: more than 2^31 (atom, channel group) pairs in one launch");
        return ARREAU_EINVAL;
    }
    const unsigned embed_blocks = (unsigned)((pairs + 255) / 256), nbr_blocks = (unsigned)((n1 - n0 + 3) / 4);
    if (tick != nullptr) {
        // sampling loop: positions from `frac` (the `cart` argument is not read), timesteps of crystals b0 .. b1-1 advanced
        const int b0 = r.b0, b1 = r.b1 < 0 ? B : r.b1;
        const unsigned tick_blocks = (unsigned)((b1 - b0 + 255) / 256);
        const unsigned grid = embed_blocks + nbr_blocks > tick_blocks ? embed_blocks + nbr_blocks : tick_blocks;  // (extra workgroups only tick)
        ARREAU_LAUNCH(neighbor_embed_kernel<true>, dim3(grid), dim3(256), 0, s, embed_blocks, frac, lattice, offsets, batch, B, n0, n1, r2,
                      m->k, deg, src, cell, dir, dist, frac, types, cvec, m->ori, m->embT, m->S, m->C, x0, m->status, tick, b0, b1);
    } else {
        ARREAU_LAUNCH(neighbor_embed_kernel<false>, dim3(embed_blocks + nbr_blocks), dim3(256), 0, s, embed_blocks, cart, lattice, offsets,
                      batch, B, n0, n1, r2, m->k, deg, src, cell, dir, dist, frac, types, cvec, m->ori, m->embT, m->S, m->C, x0, m->status,
                      (int32_t*)nullptr, 0, 0);
    }
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}

extern "C" int arreau_radius_graph_pbc(const float* d_cart, const float* d_lattice, const int32_t* d_off, int32_t B,
                                       int32_t N, float radius, int32_t k, int32_t* d_deg, int32_t* d_src,
                                       int32_t* d_cell, float* d_dir, float* d_dist, void* stream) {
    ARREAU_REQUIRE(d_cart && d_lattice && d_off && d_deg && d_src && d_cell && d_dir && d_dist,
                   "arreau_radius_graph_pbc: null pointer");
    ARREAU_REQUIRE(B >= 1 && N >= 0 && k >= 1 && k <= 64 && radius > 0.f, "arreau_radius_graph_pbc: bad size");
    return arreau_launch_neighbor(d_cart, d_lattice, d_off, nullptr, B, N, radius, k, d_deg, d_src, d_cell, d_dir, d_dist,
                                  (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// slot form <-> the reference's COO tuple
// ---------------------------------------------------------------------------------------------
__global__ void exclusive_scan_kernel(const int32_t* __restrict__ in, int n, int32_t* __restrict__ out) {
    // single workgroup, chunked Hillis-Steele; n+1 outputs (out[n] = total)
    __shared__ int32_t buf[1024];
    __shared__ int32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int idx = base + threadIdx.x;
        const int32_t v = idx < n ? in[idx] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int32_t add = threadIdx.x >= off ? buf[threadIdx.x - off] : 0;
            __syncthreads();
            buf[threadIdx.x] += add;
            __syncthreads();
        }
        if (idx < n) out[idx] = carry + buf[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += buf[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[n] = carry;
}

__global__ void compact_edges_kernel(const int32_t* __restrict__ deg, const int32_t* __restrict__ src,
                                     const int32_t* __restrict__ cell, const float* __restrict__ dir,
                                     const float* __restrict__ dist, int N, int k,
                                     const int32_t* __restrict__ eoff, int64_t* __restrict__ edge_index,
                                     float* __restrict__ cell_off, float* __restrict__ odist,
                                     float* __restrict__ odir) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * k) return;
    const int i = idx / k, s = idx - i * k;
    if (s >= deg[i]) return;
    const size_t e = (size_t)eoff[i] + s;
    const size_t cap = (size_t)N * k;
    edge_index[e] = src[idx];
    edge_index[cap + e] = i;
    const int ci = cell[idx];
    // the reference returns the NEGATED image cell (diffusion_helpers.py:551)
    cell_off[3 * e + 0] = -(float)(ci / 9 - 1);
    cell_off[3 * e + 1] = -(float)((ci / 3) % 3 - 1);
    cell_off[3 * e + 2] = -(float)(ci % 3 - 1);
    odist[e] = dist[idx];
    odir[3 * e + 0] = dir[3 * (size_t)idx + 0];
    odir[3 * e + 1] = dir[3 * (size_t)idx + 1];
    odir[3 * e + 2] = dir[3 * (size_t)idx + 2];
}

extern "C" int arreau_compact_edges(const int32_t* d_deg, const int32_t* d_src, const int32_t* d_cell,
                                    const float* d_dir, const float* d_dist, int32_t N, int32_t k,
                                    int32_t* d_edge_offsets, int64_t* d_edge_index, float* d_cell_offsets,
                                    float* d_out_dist, float* d_out_dir, void* stream) {
    ARREAU_REQUIRE(d_deg && d_src && d_cell && d_dir && d_dist && d_edge_offsets && d_edge_index && d_cell_offsets &&
                       d_out_dist && d_out_dir, "arreau_compact_edges: null pointer");
    ARREAU_REQUIRE(N >= 0 && k >= 1, "arreau_compact_edges: bad size");
    hipStream_t s = (hipStream_t)stream;
    ARREAU_LAUNCH(exclusive_scan_kernel, dim3(1), dim3(1024), 0, s, d_deg, N, d_edge_offsets);
    ARREAU_CHECK_HIP(hipGetLastError());
    if (N == 0) return ARREAU_OK;
    ARREAU_LAUNCH(compact_edges_kernel, dim3((N * k + 255) / 256), dim3(256), 0, s, d_deg, d_src, d_cell, d_dir,
                       d_dist, N, k, d_edge_offsets, d_edge_index, d_cell_offsets, d_out_dist, d_out_dir);
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}

__global__ void edges_to_slots_kernel(const int64_t* __restrict__ ei, const float* __restrict__ dist,
                                      const float* __restrict__ dir, long long E, int N, int k,
                                      int32_t* __restrict__ deg, int32_t* __restrict__ src,
                                      float* __restrict__ sdir, float* __restrict__ sdist,
                                      int32_t* __restrict__ status) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int64_t* recv = ei + E;
    const int64_t dst = recv[e];
    const int64_t sender = ei[e];
    if (dst < 0 || dst >= N || sender < 0 || sender >= N || (e > 0 && recv[e - 1] > dst)) {
        atomicExch(status, 1);
        return;
    }
    long long lo = 0, hi = e;  // first edge with recv == dst (list is receiver-sorted)
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (recv[mid] < dst) lo = mid + 1; else hi = mid;
    }
    const long long slot = e - lo;
    if (slot >= k) {
        atomicExch(status, 1);
        return;
    }
    const size_t o = (size_t)dst * k + (size_t)slot;
    src[o] = (int32_t)sender;
    sdist[o] = dist[e];
    sdir[3 * o + 0] = dir[3 * e + 0];
    sdir[3 * o + 1] = dir[3 * e + 1];
    sdir[3 * o + 2] = dir[3 * e + 2];
    if (e == E - 1 || recv[e + 1] != dst) deg[dst] = (int32_t)slot + 1;
}

extern "C" int arreau_edges_to_slots(const int64_t* d_edge_index, const float* d_dist, const float* d_dir, int64_t E,
                                     int32_t N, int32_t k, int32_t* d_deg, int32_t* d_src, float* d_slot_dir,
                                     float* d_slot_dist, int32_t* d_status, void* stream) {
    ARREAU_REQUIRE(d_deg && d_src && d_slot_dir && d_slot_dist && d_status, "arreau_edges_to_slots: null pointer");
    ARREAU_REQUIRE(E >= 0 && N >= 0 && k >= 1 && k <= ARREAU_MAX_K, "arreau_edges_to_slots: bad size");
    hipStream_t s = (hipStream_t)stream;
    ARREAU_CHECK_HIP(hipMemsetAsync(d_status, 0, sizeof(int32_t), s));
    if (N == 0) return ARREAU_OK;
    ARREAU_CHECK_HIP(hipMemsetAsync(d_deg, 0, sizeof(int32_t) * (size_t)N, s));
    ARREAU_CHECK_HIP(hipMemsetAsync(d_src, 0xFF, sizeof(int32_t) * (size_t)N * k, s));
    ARREAU_CHECK_HIP(hipMemsetAsync(d_slot_dir, 0, sizeof(float) * (size_t)N * k * 3, s));
    ARREAU_CHECK_HIP(hipMemsetAsync(d_slot_dist, 0, sizeof(float) * (size_t)N * k, s));
    if (E == 0) return ARREAU_OK;
    ARREAU_REQUIRE(d_edge_index && d_dist && d_dir, "arreau_edges_to_slots: null edge arrays");
    ARREAU_LAUNCH(edges_to_slots_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, s, d_edge_index, d_dist,
                       d_dir, (long long)E, N, k, d_deg, d_src, d_slot_dir, d_slot_dist, d_status);
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}
