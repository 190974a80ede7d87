// Internal declarations shared by the translation units of libarreau_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/arreau_hip.h"

#define ARREAU_NUM_ATTR 6      // inv1, inv2, dist, cos a, cos b, cos c   (transforms/invariants.py:85-88)
#define ARREAU_NUM_MONO 83     // distinct monomials of degree 1..3 in 6 variables (6 + 21 + 56)
#define ARREAU_MONO_PAD 96     // padded to 3 MFMA input tiles of 32
#define ARREAU_POLY_COLS 258   // 6 + 36 + 216 columns of PolynomialFeatures(3)  (nn/embedding.py:10-14)
#define ARREAU_ORI 16

struct arreau_train_ctx;
struct arreau_partition;
void arreau_partition_destroy(struct arreau_partition* p);
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

void arreau_set_error(const std::string& msg);

#define ARREAU_CHECK_HIP(expr)                                                              \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            arreau_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));           \
            return ARREAU_EHIP;                                                             \
        }                                                                                   \
    } while (0)

#define ARREAU_REQUIRE(cond, msg)                                                           \
    do {                                                                                    \
        if (!(cond)) {                                                                      \
            arreau_set_error(std::string(msg));                                            \
            return ARREAU_EINVAL;                                                           \
        }                                                                                   \
    } while (0)

// Packed weights resident in HBM.  All pointers are device pointers into `blob`.
struct arreau_model {
    arreau_config cfg;
    int S, C, D, L, O, W, H, k, T;
    float* blob;
    size_t blob_floats;
    const float* ori;        // [O][3]
    const float* w1p;        // basis layer 1, folded onto 83 monomials, MFMA-packed  out C, in 96
    const float* b1;         // [C]
    const float* w2p;        // basis layer 2, MFMA-packed                            out D, in C
    const float* b2;         // [D]
    const float* wkp;        // [L] conv.kernel.weight, MFMA-packed                   out C, in D
    const float* edge_bf16;  // w1 | w2 | wk_l as bf16x3 chunks (uint16 data), one chunk per output tile
    const float* edge_f16;   // w1 | w2 | wk_l as fp16x3 chunks (uint16 data, two planes), one chunk per output tile
    int f16_ok;              // 1 when every packed weight fits fp16 (|w| < 6e4): the fp16x3 kernels may be used
    const float* conv_x8;    // [L] fp8 e4m3 cross-product operands of conv.kernel.weight (conv_proj.hip, round 4; uint8 data)
    int x8_ok;               // 1 when 64 |w| <= 448 for every kernel weight AND the calibration allows it: the fp8 cross products may be used
    int x8_weights_ok;       // 64 |w| <= 448 for every kernel weight (what x8_ok may be switched back to)
    int fp8_ok;              // 1 when the fp8 (e4m3) residual plane of the basis stash passed the end-to-end calibration (model.hip: calibrate_message_formats)
    float calib_fp8, calib_x8;  // share of the parity bounds the formats used up on the calibration batch (fp8 residual plane; that + fp8 cross products); -1: not measured
    int train_full_range;    // 1: the training forward runs bf16x6 products even where the weights fit fp16 (arreau_model_set_variant(.., 1))
    int calibrating;         // 1 while that calibration runs: basis form at any launch size, formats from fp8_ok / x8_ok alone
    float edge_act_bound, node_act_bound;  // weight-derived bounds of the fp16 operands of the edge / ConvNext chains (model.hip)
    // Arithmetic / geometry variants requested for this model (defaults from ARREAU_*_VARIANT at create,
    // arreau_model_set_variant overrides) and what the last arreau_predict_scores actually launched.
    int edge_variant, mlp_variant, conv_variant, readout_variant;
    mutable int ran_edge, ran_mlp, ran_conv, ran_x8;
    // training (train_net.hip): plain row-major fp32 copies of the weights the sampling kernels hold only in packed
    // form, and the step's activation buffers (created on first use)
    const float *t_w1f, *t_w2, *t_wk, *t_lin1, *t_lin2, *t_ro_w;
    struct arreau_train_ctx* train;
    // crystal-aligned slices of the batch the score network may be run in, on separate streams (api.hip: set by
    // arreau_model_set_batch_layout; used when (B, N) of a call match and the default kernel set is selected)
    struct arreau_partition* part;
    int fused;               // 1: shape (C, D, W) = (128, 256, 4), the fused kernels' packed weights exist
    int packed_stale;        // 1 after arreau_model_update_train_weights: the sampling kernels' packed planes are out of date
    void* loop_stream;       // hipStream_t / hipEvent_t of arreau_sample_loop's graph mode (capture is not allowed on the
    void* loop_event;        //   legacy default stream callers usually pass); created on first use
    uint64_t graph_key[12];  // what the cached executable graph of arreau_sample_loop was captured for
    void* retired_graph;     // hipGraphExec_t of the last arreau_sample_loop (+ the stream it was launched on): destroyed,
    void* retired_stream;    //   after that stream has drained, by the next loop or by arreau_model_destroy
    int32_t* status;         // device word of sticky ARREAU_STATUS_* bits (written by the kernels with atomicOr)
    float* fk;               // [L][O(o)][O(p)][C] fiber kernels / O (written once by the precompute kernel)
    const float* conv_bias;  // [L][C]
    const float* ln_w;       // [L][C]
    const float* ln_b;       // [L][C]
    const float* mlp;        // [L][4 quarters][ linear_1 rows of the quarter (out H/4, in C) | linear_2 columns of the
                             //  quarter (out C, in H/4) ], MFMA-packed: one linear stream per (layer, quarter)
    const float* mlp_bf16;   // the same stream as bf16x3 chunks (uint16 data), 24 KiB per output tile
    const float* mlp_f16;    // the same stream as fp16x3 chunks (uint16 data), 16 KiB per output tile
    const float* mlp_f16m;   // ... as fp16x3 chunks for v_mfma_f32_16x16x32_f16 (node_f16m.hip)
    const float* mb1;        // [L][H]
    const float* mb2;        // [L][C]
    const float* ls;         // [L][C] layer_scale (ones when absent)
    const float* embT;       // [S+78][C] x_embedder.weight transposed
    const float* ro_wT;      // [L][C][S+4] read_out weight transposed
    const float* ro_b;       // [L][S+4]
    const float* ro_pack;    // [L][(S+4 padded to 32)/32][C/32][1024] read_out weights as fp32-MFMA fragment streams
    const float* ro_wv;      // [L][C] vector read-out weights (column S of read_out_layers), contiguous
    float ro_bv_host[16];    // [L] vector read-out bias (host copy, passed by value)
    const float* t_emb_w;    // [32]
    const float* ve_sigmas;  // [T+1]
    const float* vp_alpha_bars;  // [T+1]
    const float* vp_betas;   // [T+1]
    const float* q1t;        // [T][S][S]
    const float* qmats;      // [T][S][S]
    int qmats_absorbing;     // 1 when every Qbar_t is diagonal + last ("mask") column: the D3PM update skips the dense S x S read
    // raw fiber-basis MLP (only used by the one-time precompute kernel)
    const float* fiber_w1;   // [C][3]
    const float* fiber_b1;   // [C]
    const float* fiber_w2;   // [D][C]
    const float* fiber_b2;   // [D]
    const float* fiber_wk;   // [L][C][D] conv.fiber_kernel.weight
};

// ---------------------------------------------------------------------------------------------
// MFMA building block.  All dense layers are evaluated "transposed":
//     Y^T[out, row] = W[out, in] . X^T[in, row]
// with v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).  A 32x32 output tile lives in 16
// accumulator registers per lane: lane (h = lane>>5, j = lane&31) holds, in register r,
//     Y^T[32u + rho(r,h)][row j],   rho(r,h) = (r&3) + 8*(r>>2) + 4*h.
// Because the reduction index of the NEXT layer is this tile's row index, the accumulator is
// directly the B operand of the next MFMA chain (register r of tile t feeds the k-step whose two
// k values are rho(r,0) and rho(r,1)): activations never leave registers between layers.
// The A operand (weights) is pre-packed on the host so one 16-byte load per lane serves 4 k-steps:
//     P[u][t][q][lane][m] = W[out = 32u + (lane&31)][in = 32t + 8q + 4*(lane>>5) + m]
// (u: out tile, t: in tile, q: 0..3, m: 0..3) -- 1 KiB contiguous per wave-load.
// ---------------------------------------------------------------------------------------------
#define ARREAU_PACK_TILE_FLOATS 1024  // floats per (u,t) tile: 4 q * 64 lanes * 4

__device__ __forceinline__ f32x16 arreau_mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// Weights are consumed as ONE linear stream of 16-byte fragments per wave (1 KiB per wave-load).  Output
// tiles are produced one at a time (out-tile-major): for output tile u the k-groups g = 0..G-1 (G = 4 * input
// tiles) are contiguous, 256 floats apart, and tile u+1 follows.  The stream is prefetched ARREAU_PF groups
// ahead through a register ring, so L2 latency hides behind ARREAU_PF * 4 * NCB MFMAs (64 cycles each).
// NCB = 32-row column blocks sharing each fragment.  f0 (first group of this tile within `region`) must be a
// compile-time constant after unrolling so that ring slots are static registers.
#define ARREAU_PF 8

template <int G, int NIN, int NCB>
__device__ __forceinline__ void arreau_stream_tile(f32x16 (&acc)[NCB], f32x4 (&ring)[ARREAU_PF],
                                                   const float* __restrict__ region, const int f0,
                                                   const f32x16 (&b)[NIN][NCB]) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int f = f0 + g;
        const f32x4 a = ring[f % ARREAU_PF];
        ring[f % ARREAU_PF] = *reinterpret_cast<const f32x4*>(region + (size_t)(f + ARREAU_PF) * 256);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[cb] = arreau_mfma(a[m], b[g >> 2][cb][4 * (g & 3) + m], acc[cb]);
        }
    }
    // pin the issue order inside this scheduling region: one fragment prefetch, then its 4*NCB MFMAs
#pragma unroll
    for (int g = 0; g < G; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);        // VMEM read
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NCB, 0);  // MFMA
    }
}

// accumulator tile initialised with a bias: register r gets bias[32u + rho(r,h)]
__device__ __forceinline__ f32x16 arreau_bias_tile(const float* __restrict__ bias, int u, int h) {
    f32x16 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(bias + 32 * u + 8 * q + 4 * h);
        r[4 * q] = v[0]; r[4 * q + 1] = v[1]; r[4 * q + 2] = v[2]; r[4 * q + 3] = v[3];
    }
    return r;
}

// Branch-free erf (two polynomial pieces, both evaluated, one selected): the library erff branches on
// |x|, which splits the fully unrolled MFMA chains into hundreds of basic blocks and wrecks register
// allocation.  Coefficients: N. Juffa's single-precision erf (max error < 1 ulp with an accurate exp);
// v_exp_f32 adds about 2e-7 relative on the exp term, far inside the 1e-5 parity budget.
__device__ __forceinline__ float arreau_erf(float a) {
    const float t = fabsf(a);
    const float s = a * a;
    float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = fmaf(r, s, u);
    r = fmaf(r, t, -1.06777877e-1f);
    r = fmaf(r, t, -6.34846687e-1f);
    r = fmaf(r, t, -1.28717512e-1f);
    r = fmaf(r, t, -t);
    r = 1.0f - __builtin_amdgcn_exp2f(r * 1.44269504088896341f);
    r = copysignf(r, a);
    float p = -5.96761703e-4f;
    p = fmaf(p, s, 4.99119423e-3f);
    p = fmaf(p, s, -2.67681349e-2f);
    p = fmaf(p, s, 1.12819925e-1f);
    p = fmaf(p, s, -3.76125336e-1f);
    p = fmaf(p, s, 1.28379166e-1f);
    p = fmaf(p, a, a);
    return t > 0.927734375f ? r : p;
}

// exact (erf) GELU, torch.nn.GELU() default
__device__ __forceinline__ float arreau_gelu(float x) {
    return 0.5f * x * (1.0f + arreau_erf(x * 0.70710678118654752440f));
}

// Noise of one reverse step: either the caller's arrays (parity mode: the reference's draws injected) or drawn in the
// kernel from Philox4x32-10 keyed by (seed, timestep, draw kind, element) -- no RNG launches, no noise arrays in HBM.
struct StepNoiseSrc {
    const float* z_lattice;  // [B,3]  or null
    const float* z_frac;     // [N,3]  or null
    const float* u_types;    // [N,S]  or null
    uint64_t seed;           // used when the arrays are null
};

void arreau_train_ctx_destroy(struct arreau_train_ctx* t);
// edge_variant value that selects the shape-general fp32 network (train_net.hip) for the whole evaluation
#define ARREAU_VARIANT_GENERAL 5
inline bool arreau_general_path(const arreau_model* m) { return !m->fused || m->edge_variant == ARREAU_VARIANT_GENERAL; }
// The score network on the shape-general fp32 kernels (the training forward without its own graph construction): graph
// arrays and the embedded features x0 [N*16][C] (null: t.x already holds them) are the caller's; outputs as
// arreau_predict_scores.  Keeps every activation (a following arreau_train_backward is valid when keep_graph copies).
struct arreau_graph_view {
    const int32_t *batch, *deg, *src;
    const float *lattice, *dir, *dist;
};
int arreau_general_network(arreau_model* m, const arreau_graph_view& g, const int32_t* d_off, int B, int N, float* d_eps,
                           float* d_logits, float* d_len0, hipStream_t s);
float* arreau_general_x0(arreau_model* m, int N, int B, hipStream_t s);  // (re)sizes the context; returns its layer-0 feature buffer
void arreau_model_retire_graph(arreau_model* m, void* exec, void* stream);  // takes ownership; frees the previous one

// A launch over part of the batch: receivers / atoms n0 .. n1-1 (n1 < 0: all), crystals b0 .. b1-1, and for the persistent
// kernels a cap on the workgroup count (0: one per CU).  All arrays stay whole-batch arrays with absolute indices, so a
// range launch computes bit for bit what the whole-batch launch computes for those atoms: arreau_predict_scores uses it
// to run crystal-aligned slices of the batch on separate streams (api.hip).
struct NodeRange {
    int n0 = 0, n1 = -1, b0 = 0, b1 = -1, wg_cap = 0;
};

// Debug aid (api.hip, "uninitialised-state probe"): when a pollution pattern is set (arreau_debug_set_pollution), every
// kernel launch of the sampling path is preceded, on the same stream, by a kernel that fills every CU's LDS and vector
// registers with the pattern.  A kernel that reads LDS / registers it never wrote sees whatever the previous wave on its
// CU left there: alone on the GPU that is its own kernel's leftovers (the same every run), next to another stream's or
// process's kernels it is not -- the signature of the non-reproducibility DESIGN.md section 8 records.  With the probe the
// leftovers are under the test's control: outputs must be bit-identical for every pattern
// (tests/test_gpu_parity.py::test_outputs_do_not_depend_on_leftover_lds_or_registers).
extern unsigned arreau_debug_pollution;  // 0 = off (the product never sets it)
int arreau_debug_pollute(hipStream_t s);
#define ARREAU_LAUNCH(kernel, grid, block, smem, stream, ...)                      \
    do {                                                                           \
        if (arreau_debug_pollution) (void)arreau_debug_pollute(stream);            \
        hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__);        \
    } while (0)

// launchers implemented in the other translation units ------------------------------------------
int arreau_launch_fiber_precompute(arreau_model* m, hipStream_t s);
int arreau_launch_neighbor(const float* cart, const float* lattice, const int32_t* offsets, const int32_t* batch, int B, int N,
                           float radius, int k, int32_t* deg, int32_t* src, int32_t* cell, float* dir, float* dist,
                           hipStream_t s, NodeRange r = NodeRange());
// neighbour list + embedding of the sampler's node features in one launch (graph.hip; both read prep_kernel's outputs only)
int arreau_launch_neighbor_embed(const arreau_model* m, const float* cart, const float* lattice, const int32_t* offsets,
                                 const int32_t* batch, int B, int N, int32_t* deg, int32_t* src, int32_t* cell, float* dir,
                                 float* dist, const float* frac, const int32_t* types, const float* cvec, float* x0, hipStream_t s,
                                 NodeRange r = NodeRange(), int32_t* tick = nullptr /* sampling loop without a prep launch: see the kernel */);
int arreau_launch_prep(const arreau_model* m, const float* frac, const float* lengths, const float* angles,
                       const int32_t* t, const int32_t* offsets, int B, int N, float* lattice, float* cart,
                       int32_t* batch, float* cvec, hipStream_t s, int32_t* t_next = nullptr, int32_t* t_cur = nullptr,
                       NodeRange r = NodeRange(), int t_offset = 0);
int arreau_launch_reverse(const arreau_model* m, float* d_frac, int32_t* d_types, float* d_lengths, const float* d_angles,
                          const int32_t* d_t, const int32_t* d_off, int B, int N, const float* d_eps,
                          const float* d_logits, const float* d_len0, StepNoiseSrc noise, const int32_t* d_const_types,
                          float* d_lattice, hipStream_t s, const float* d_fixed_lengths = nullptr, NodeRange r = NodeRange(),
                          const float* d_gs_atoms = nullptr /* pool these per-atom read-outs into d_len0 first */,
                          const int32_t* d_batch = nullptr /* crystal index of each atom, if the caller has it */,
                          float* d_lattice_ws = nullptr, float* d_cvec_next = nullptr /* sampling loop: also prepare the next step
                          (workspace lattice + per-crystal embedding for timestep t - 1), see reverse_crystal_block */);
int arreau_launch_edge(const arreau_model* m, const float* dir, const float* dist, const int32_t* deg,
                       const int32_t* batch, const float* lattice, int N, float* kbuf, hipStream_t s, NodeRange r = NodeRange());
int arreau_launch_edge_bf16x6(const arreau_model* m, const float* dir, const float* dist, const int32_t* deg,
                              const int32_t* batch, const float* lattice, int N, float* kbuf, hipStream_t s);
int arreau_launch_edge_f16x3(const arreau_model* m, const float* dir, const float* dist, const int32_t* deg,
                             const int32_t* batch, const float* lattice, int N, float* kbuf, hipStream_t s, NodeRange r = NodeRange());
int arreau_launch_embed(const arreau_model* m, const float* frac, const int32_t* types, const float* lattice,
                        const int32_t* batch, const float* cvec, int N, float* x0, hipStream_t s, NodeRange r = NodeRange());
int arreau_launch_embed_general(const arreau_model* m, const float* x, const float* vec, int N, float* x0, hipStream_t s);
int arreau_launch_batch_index(const int32_t* offsets, int B, int N, int32_t* batch, hipStream_t s);
int arreau_launch_node_layer(const arreau_model* m, int layer, const float* kbuf, const int32_t* deg,
                             const int32_t* src, const float* x_in, float* x_conv, float* x_out, float* xbar,
                             float* vsum, int N, hipStream_t s, NodeRange r = NodeRange());
// ---- K stash format ------------------------------------------------------------------------------------------------
// The per-layer edge kernels K [L][N*k*16][C] are the only large intermediate of a sampling step.  With the split-precision
// edge kernel and the streamed conv kernel (the default pair) they are held as 3-BYTE floats: the top three bytes of the
// fp32 value after rounding to nearest on the magnitude (1 sign + 8 exponent + 15 stored significand bits), four values in
// three dwords, a row of C values in 3 C bytes.  Rounding K to 16 significand bits changes the network outputs by no more
// than the fp32 oracle's own rounding floor (tools/exp/k_precision_study.py, profiles/r02g_k_precision_study.txt); it takes
// a quarter off the stash's HBM traffic.  ARREAU_K3=0 keeps fp32.  Every other producer / consumer pair keeps fp32.
typedef unsigned int u32x3_k __attribute__((ext_vector_type(3)));
__device__ __forceinline__ u32x3_k arreau_pack_k3(float v0, float v1, float v2, float v3) {
    const unsigned r0 = __float_as_uint(v0) + 0x80u, r1 = __float_as_uint(v1) + 0x80u;
    const unsigned r2 = __float_as_uint(v2) + 0x80u, r3 = __float_as_uint(v3) + 0x80u;
    u32x3_k d;
    d[0] = __builtin_amdgcn_perm(r1, r0, 0x05030201u);  // r0.b1 r0.b2 r0.b3 r1.b1
    d[1] = __builtin_amdgcn_perm(r2, r1, 0x06050302u);  // r1.b2 r1.b3 r2.b1 r2.b2
    d[2] = __builtin_amdgcn_perm(r3, r2, 0x07060503u);  // r2.b3 r3.b1 r3.b2 r3.b3
    return d;
}
__device__ __forceinline__ void arreau_unpack_k3(unsigned d0, unsigned d1, unsigned d2, float (&v)[4]) {
    v[0] = __uint_as_float(d0 << 8);
    v[1] = __uint_as_float(__builtin_amdgcn_perm(d1, d0, 0x0504030cu));  // 0 d0.b3 d1.b0 d1.b1
    v[2] = __uint_as_float(__builtin_amdgcn_perm(d2, d1, 0x0403020cu));  // 0 d1.b2 d1.b3 d2.b0
    v[3] = __uint_as_float(d2 & 0xffffff00u);
}
// the decision, shared by the edge and the node-layer launchers (both read the same model fields)
bool arreau_k3(const arreau_model* m);
// Basis form (round 3, the default for launches above the small-launch sizes): the edge kernel stores the windowed basis
// planes instead of the L projected kernels, and each layer's message kernel projects them itself (conv_proj.hip) -- no K
// stash.  The edge launcher and the node-layer launcher take the decision from the same fields and the same receiver count.
bool arreau_basis_form(const arreau_model* m, int receivers);
// Numerics switch of the split-precision edge path: the residual plane of the windowed basis is rounded to fp8 e4m3 (11 + 4
// significand bits; default) -- what the basis form stores (3 bytes per value), applied by every fp16x3 edge kernel so that all
// launch sizes evaluate the same numbers.  Round 5: per MODEL -- arreau_model_create keeps it only if its calibration batch says the
// outputs hardly move (model.hip: calibrate_message_formats); ARREAU_BASIS_FP8=0 / 1 in the environment overrides either way
// (0: both planes fp16 everywhere, the round-2 arithmetic).
bool arreau_basis_fp8(const arreau_model* m);
// Round 4: the two cross products of the per-layer kernel projection (conv_proj.hip) on the fp8 matrix instruction (twice the fp16
// rate; operands e4m3: model.hip, pack_conv_cross_fp8).  Applies to the basis form with the fp8 residual plane; ARREAU_CROSS_FP8=0
// keeps three fp16 products (bit-identical to the K pair of the small launches).  Read per call (tests toggle it).
bool arreau_cross_fp8(const arreau_model* m);
void arreau_prof_conv(int end, hipStream_t s);  // api.hip: hipEvents around the message kernel while bench.py profiles
int arreau_launch_conv_proj(const arreau_model* m, int layer, const float* basis, const int32_t* deg, const int32_t* src,
                            const float* x_in, float* x_conv, int N, hipStream_t s, NodeRange r = NodeRange());

int arreau_launch_mlp_bf16x6(const arreau_model* m, int layer, const float* x_conv, const float* x_in, float* x_out,
                             float* xbar, float* vsum, int N, hipStream_t s);
bool arreau_mlp_train_forward_available(const arreau_model* m);
int arreau_launch_mlp_train_forward(const arreau_model* m, int layer, const float* x_conv, const float* x_in, float* x_out, float* xhat,
                                    float* rstd, float* xn, float* hpre, float* h, float* out, int N, hipStream_t s,
                                    // round 5: the spatial conv + spherical mix of the layer inside the same launch (kl = the layer's kernels,
                                    // rows kl_pitch floats apart; x1 receives the spatial conv's output; x_conv is not read) -- null: x_conv given
                                    const float* kl = nullptr, int kl_pitch = 0, const int32_t* deg = nullptr, const int32_t* src = nullptr,
                                    const float* fk = nullptr, float* x1 = nullptr);
int arreau_repack_mlp_f16x3_m16(arreau_model* m, hipStream_t s);
int arreau_launch_mlp_f16x3_m16(const arreau_model* m, int layer, const float* x_conv, const float* x_in, float* x_out,
                                float* xbar, float* vsum, int N, hipStream_t s, NodeRange r = NodeRange());
// one launch per layer for small unsliced launches (node_f16m.hip): conv_kernel_streamed<128, false> + the split MLP form
bool arreau_small_layer_fusable(const arreau_model* m, int N, NodeRange r);
int arreau_launch_small_layer(const arreau_model* m, int layer, const float* kbuf, const int32_t* deg, const int32_t* src,
                              const float* x_in, float* x_out, float* xbar, float* vsum, int Ntot, hipStream_t s);
int arreau_launch_mlp_f16x3_m16_split(const arreau_model* m, int layer, const float* x_conv, const float* x_in, float* x_out,
                                      float* xbar, float* vsum, int N, hipStream_t s, NodeRange r = NodeRange());
int arreau_launch_readout(const arreau_model* m, const float* xbar, const float* vsum, const int32_t* offsets,
                          int B, int N, float* gs, float* eps, float* logits, float* len0, hipStream_t s, NodeRange r = NodeRange());
bool arreau_range_launches_supported(const arreau_model* m);  // the default kernel set (fp16x3 edge + MLP, streamed conv, MFMA read-out)
