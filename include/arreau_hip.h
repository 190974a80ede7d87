/*
 * arreau_hip.h -- C ABI of libarreau_hip.so: the MI355X (gfx950) implementation of
 * arreau's reverse-diffusion sampling step.
 *
 * The reference (curtischong/arreau) is pure Python; its seams for this path are
 * Python calls.  Each entry point below names the reference interface it replaces
 * (paths relative to the reference checkout).  Conventions:
 *
 *   - all pointers named d_* are DEVICE pointers (HBM), contiguous row-major;
 *     h_* are HOST pointers.  float = fp32, index arrays = int32.
 *   - every launch function enqueues on `stream` (a hipStream_t passed as void*)
 *     and never synchronises, allocates or frees; scratch memory comes from a
 *     caller-owned workspace (arreau_workspace_bytes).
 *   - return value: 0 on success, a negative ARREAU_E* code otherwise;
 *     arreau_last_error() gives a message for the calling thread.
 *   - crystals are described CSR-style by d_crystal_offsets[B+1] (first atom of
 *     each crystal; atoms of one crystal are contiguous, as in the reference's
 *     `num_atoms` convention, diffusion/diffusion_loss.py:308,330-335).
 *   - the neighbour list is kept receiver-major in fixed-width slots:
 *     slot (i, s), s < deg[i] <= k, holds the s-th in-edge of receiver atom i in
 *     the reference's enumeration order (sender, image).  Unused slots hold
 *     src = -1, dir = 0, dist = 0.
 */
#ifndef ARREAU_HIP_H
#define ARREAU_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ARREAU_OK 0
#define ARREAU_EINVAL (-1)      /* bad argument / unsupported hyper-parameter */
#define ARREAU_EHIP (-2)        /* a HIP runtime call failed */
#define ARREAU_ECAPACITY (-3)   /* workspace too small / slot overflow */

/* Hyper-parameters read from the checkpoint's `args` (lightning_wrappers/diffusion.py:42-54,
 * 86-102; diffusion/diffusion_loss.py:70-72) plus S = len(z_table). */
typedef struct arreau_config {
    int32_t num_atomic_states;  /* S, including the mask state (last class)          */
    int32_t hidden_dim;         /* C  (args.hidden_dim)                              */
    int32_t basis_dim;          /* D  (args.basis_dim)                               */
    int32_t num_layers;         /* L  (args.layers)                                  */
    int32_t num_ori;            /* O  (args.num_ori), must be 16                     */
    int32_t widening_factor;    /* W  (args.widening_factor)                         */
    int32_t degree;             /* polynomial degree (args.degree), must be 3        */
    int32_t max_neighbors;      /* k  (args.max_neighbors), <= ARREAU_MAX_K          */
    int32_t num_timesteps;      /* T  (args.num_timesteps)                           */
    float radius;               /* R  (args.radius)                                  */
    int32_t has_layer_scale;    /* 0 when args.layer_scale == 0 (convnext.py:14-17)  */
} arreau_config;

#define ARREAU_MAX_K 8
#define ARREAU_T_EMB_DIM 64      /* lightning_wrappers/diffusion.py:23 */
#define ARREAU_N_CRYSTAL_FEATS 10 /* n, lengths(3), angles(3), |lengths/n|(3): diffusion_loss.py:139-149 */

/* The network's parameters and the diffusion buffers in the reference's state_dict
 * layout, as HOST fp32 arrays (SURVEY.md section 5 lists the keys).  Per-layer
 * tensors are stacked along a leading L axis. */
typedef struct arreau_state_dict {
    const float* basis_w1;      /* model.basis_fn.1.weight            [C, 258]      */
    const float* basis_b1;      /* model.basis_fn.1.bias              [C]           */
    const float* basis_w2;      /* model.basis_fn.3.weight            [D, C]        */
    const float* basis_b2;      /* model.basis_fn.3.bias              [D]           */
    const float* fiber_w1;      /* model.fiber_basis_fn.1.weight      [C, 3]        */
    const float* fiber_b1;      /* model.fiber_basis_fn.1.bias        [C]           */
    const float* fiber_w2;      /* model.fiber_basis_fn.3.weight      [D, C]        */
    const float* fiber_b2;      /* model.fiber_basis_fn.3.bias        [D]           */
    const float* x_embedder_w;  /* model.x_embedder.weight            [C, S+78]     */
    const float* conv_kernel_w; /* ...interaction_layers.i.conv.kernel.weight        [L, C, D] */
    const float* conv_fiber_w;  /* ...interaction_layers.i.conv.fiber_kernel.weight  [L, C, D] */
    const float* conv_bias;     /* ...interaction_layers.i.conv.bias                 [L, C]    */
    const float* norm_w;        /* ...interaction_layers.i.norm.weight               [L, C]    */
    const float* norm_b;        /* ...interaction_layers.i.norm.bias                 [L, C]    */
    const float* linear1_w;     /* ...interaction_layers.i.linear_1.weight           [L, W*C, C] */
    const float* linear1_b;     /* ...interaction_layers.i.linear_1.bias             [L, W*C]  */
    const float* linear2_w;     /* ...interaction_layers.i.linear_2.weight           [L, C, W*C] */
    const float* linear2_b;     /* ...interaction_layers.i.linear_2.bias             [L, C]    */
    const float* layer_scale;   /* ...interaction_layers.i.layer_scale               [L, C] (NULL if absent) */
    const float* readout_w;     /* model.read_out_layers.i.weight                    [L, S+4, C] */
    const float* readout_b;     /* model.read_out_layers.i.bias                      [L, S+4]  */
    const float* ori_grid;      /* PositionOrientationGraph.ori_grid_s2 (not in the state_dict) [O, 3] */
    const float* t_emb_w;       /* t_emb.gaussian_fourier_proj_w                     [32]      */
    const float* ve_sigmas;     /* diffusion_loss.pos_diffusion.sigmas               [T+1]     */
    const float* vp_alpha_bars; /* diffusion_loss.lattice_diffusion.alpha_bars       [T+1]     */
    const float* vp_betas;      /* diffusion_loss.lattice_diffusion.betas            [T+1]     */
    const float* q_one_step_transposed; /* diffusion_loss.d3pm.q_one_step_transposed [T, S, S] */
    const float* q_mats;        /* diffusion_loss.d3pm.q_mats                        [T, S, S] */
} arreau_state_dict;

typedef struct arreau_model arreau_model; /* opaque: packed weights resident in HBM */

const char* arreau_last_error(void);
const char* arreau_version(void);

/* Replaces PonitaFiberBundle.__init__ + load_state_dict (ponita/models/ponita.py:31-86) and the
 * buffer set-up of DiffusionLoss.__init__ (diffusion/diffusion_loss.py:68-93): folds the 258
 * polynomial columns onto their 83 distinct monomials, repacks every Linear for the MFMA
 * fragment order, uploads, and evaluates the input-independent fiber kernels
 * fiber_kernel(fiber_basis_fn(o_a . o_b)) (ponita.py:95, conv.py:113) once on the GPU.
 * Allocates device memory (the only entry point besides arreau_model_destroy that does).
 * Shapes: the fused sampling kernels exist for hidden_dim 128, basis_dim 256, widening_factor 4 (the shipped
 * checkpoint).  Any other shape with hidden_dim, basis_dim multiples of 4 and widening_factor * hidden_dim <= 1024
 * (e.g. the reference's `make train` preset hidden_dim = 200, Makefile:7) is accepted and runs every entry point --
 * scores, inner seam, sampling loop, training -- on the shape-general fp32 kernels (exact fp32 MFMA GEMMs +
 * element-wise kernels): same results, no fusion.  num_ori = 16, degree = 3, max_neighbors <= 8 are fixed. */
int arreau_model_create(const arreau_config* cfg, const arreau_state_dict* h_sd, void* stream,
                        arreau_model** out_model);
void arreau_model_destroy(arreau_model* model);
int arreau_model_config(const arreau_model* model, arreau_config* out_cfg);

/* Sticky condition bits collected on the device while kernels run (nothing is checked on the host per call, so
 * the launch functions stay asynchronous).  A set bit means results since the last reset must not be trusted:
 *   NONFINITE    a network output (eps, logits, pred_lengths_0) was inf/NaN -- this is how an activation beyond the
 *                fp16 range of the split-precision kernels (|v| >= 65520) surfaces: the planes overflow to inf and the
 *                value propagates as NaN instead of being clamped silently;
 *   BAD_TIMESTEP a timestep outside [0, T] (predict_scores) / [1, T] (reverse_step) was clamped;
 *   BAD_TYPE     an atom-type index outside [0, S) was clamped.
 * edge_kernel / mlp_kernel / conv_kernel name the kernel family the last arreau_predict_scores really launched
 * (edge: 0-2 fp32 MFMA, 3 bf16x6, 4 fp16x3; mlp: 0 fp32 MFMA, 1 bf16x6, 2 fp16x3 32x32x16, 3 fp16x3 16x16x32;
 * conv: 0 register form, 1 streamed form, 2 fused into the MLP kernel) -- e.g. 3/1 instead of 4/3 when a weight does not
 * fit fp16.  The reference has no counterpart (Python raises on bad indices: F.one_hot, tensor indexing). */
#define ARREAU_STATUS_NONFINITE 1
#define ARREAU_STATUS_BAD_TIMESTEP 2
#define ARREAU_STATUS_BAD_TYPE 4
typedef struct arreau_status {
    int32_t flags;
    int32_t edge_kernel;
    int32_t mlp_kernel;
    int32_t conv_kernel;
    int32_t basis_row_bytes; /* conv_kernel == 2: bytes of the stashed basis per (edge, orientation) row the kernels really used
                              * (768: fp16 plane + fp8 e4m3 residual plane; 1024: two fp16 planes -- ARREAU_BASIS_FP8=0, or a model whose
                              * calibration dropped the fp8 plane); 0 otherwise */
    int32_t conv_cross_fp8;  /* conv_kernel == 2: 1 when the layer projections ran their two cross products on the fp8 matrix instruction
                              * (round 4 default; e4m3 operands, twice the fp16 rate), 0 for three fp16 products (ARREAU_CROSS_FP8=0) */
    float edge_activation_bound; /* bounds, from the weights alone, of every fp16 operand of the split-precision edge chain */
    float node_activation_bound; /* (monomials, hidden units, basis) resp. ConvNext chain (LayerNorm output, hidden units): at most
                                  * 65504 = the fp16x3 kernels provably cannot overflow; up to 64 x that the library keeps them
                                  * and relies on NONFINITE (the host re-runs on bf16x6); beyond, the model starts on bf16x6 */
    float basis_fp8_share;   /* round 5: what arreau_model_create measured on its calibration batch -- the share of the parity bounds
                              * (1e-5 max(1, |eps|), 1e-5 max(1, |logits| / 8)) the fp8 residual plane of the stash, resp. that plane + */
    float cross_fp8_share;   /* the fp8 cross products used up against two fp16 planes + three fp16 products; a format above 0.1 is not
                              * used for this model (basis_row_bytes 1024 / conv_cross_fp8 0 then); -1: not measured */
} arreau_status;
/* Reads (and with reset != 0 clears) the status word; synchronises `stream`. */
int arreau_model_status(const arreau_model* model, arreau_status* out, int32_t reset, void* stream);
/* Round 5: selects the operand formats of the message path for this model (-1 keeps the current choice; 0 / 1): the fp8 (e4m3)
 * residual plane of the basis stash and the fp8 cross products of the layer projections (which need that plane and kernel
 * weights inside e4m3's range).  arreau_model_create chooses them from its calibration batch; the host clears both when an
 * evaluation came out non-finite with them on -- the hardware's fp8 conversion returns NaN for a basis value beyond 464 -- and
 * repeats it (HipEngine.checked, DiffusionLoss.sample).  No reference counterpart. */
int arreau_model_set_formats(arreau_model* model, int32_t basis_fp8, int32_t cross_fp8);
/* Selects the arithmetic of the dense kernels for this model (-1 keeps the current choice); the defaults come from
 * the environment (ARREAU_EDGE_VARIANT, ARREAU_MLP_VARIANT) at arreau_model_create.  Used by the parity report and
 * bench.py to time/compare the exact fp32-MFMA kernels against the default fp16x3 ones in one process.
 * mlp_variant 4 forces the small-launch form of the default ConvNext kernel (one node per workgroup, the layer's work dealt
 * to eight waves; bit-identical to variant 3, which picks it by itself for small launches) at every size (tests).
 * edge_variant 5 runs the whole score network on the shape-general fp32 GEMM kernels (what shapes without fused kernels
 * always use; such models accept no other value).  Those kernels read the plain weights
 * arreau_model_update_train_weights refreshes, so with variant 5 a model keeps sampling between optimiser steps. */
int arreau_model_set_variant(arreau_model* model, int32_t edge_variant, int32_t mlp_variant);

/* Optional: tell the library how the batch is laid out (HOST copy of d_crystal_offsets[B+1]) so that the score network
 * of subsequent arreau_predict_scores / arreau_sample_loop calls with the same (B, N) may run as `groups`
 * crystal-aligned slices on separate internal streams (forked from and joined to the caller's stream by events; no host
 * synchronisation).  Crystals are independent and every kernel takes a node range over whole-batch arrays, so
 * every slice computes what the unsliced run computes for its atoms; the slices drift into different phases and the
 * HBM-bound message-passing kernel of one overlaps the matrix-bound edge / MLP kernels of another (3-5 % at 256 x 20).
 * EXPERIMENT, refused by default (ARREAU_EINVAL for groups > 1 unless the environment holds ARREAU_ALLOW_MULTISTREAM=1):
 * on MI355X, with kernels of two streams or two processes sharing CUs, one crystal in a few runs came out different at
 * the 1e-8 .. 1e-4 level and the cause is not known, so a sliced multi-stream run is not
 * guaranteed identical to the unsliced one.  ARREAU_SLICE_EAGER=serial runs the slices' range launches one after another
 * on the caller's stream (bit-identical to the unsliced run; what the tests of the range launches use).
 * groups <= 0: the library's default (environment ARREAU_GROUPS, else 1 = off).  Used only with the default kernel set;
 * ignored otherwise. */
int arreau_model_set_batch_layout(arreau_model* model, const int32_t* h_crystal_offsets, int32_t B, int32_t groups);

/* Scratch for one step over at most max_atoms atoms / max_crystals crystals. */
size_t arreau_workspace_bytes(const arreau_config* cfg, int64_t max_atoms, int64_t max_crystals);

/* ---- small geometric operators ------------------------------------------------------------ */

/* lattice_from_params, diffusion/lattice_helpers.py:55-105.  d_lengths[B,3], d_angles[B,3]
 * (consumed as radians) -> d_lattice[B,3,3]. */
int arreau_lattice_from_params(const float* d_lengths, const float* d_angles, int32_t B,
                               float* d_lattice, void* stream);

/* frac_to_cart_coords, diffusion/diffusion_helpers.py:223-230. */
int arreau_frac_to_cart(const float* d_frac, const float* d_lattice, const int32_t* d_crystal_offsets,
                        int32_t B, int32_t N, float* d_cart, void* stream);

/* radius_graph_pbc(cart, lattice, num_atoms, radius, max_num_neighbors_threshold,
 * remove_self_edges=True), diffusion/diffusion_helpers.py:328-564, in slot form.
 * Outputs: d_deg[N]; d_src[N,k] (global sender index, -1 when unused); d_cell[N,k] image code
 * 0..26 in the reference's SUPERCELLS order (diffusion_helpers.py:10), -1 when unused;
 * d_dir[N,k,3] = pos_sender + image_offset - pos_receiver; d_dist[N,k].
 * Tie rule (the reference leaves it to an unstable sort): smaller d^2 first, then smaller
 * enumeration index (sender, image). */
int arreau_radius_graph_pbc(const float* d_cart, const float* d_lattice,
                            const int32_t* d_crystal_offsets, int32_t B, int32_t N,
                            float radius, int32_t k,
                            int32_t* d_deg, int32_t* d_src, int32_t* d_cell, float* d_dir,
                            float* d_dist, void* stream);

/* Slot form -> the reference's return tuple (edge_index[2,E] as (sender, receiver),
 * -unit_cell[E,3], dist[E], direction[E,3]; diffusion_helpers.py:548-555).
 * d_edge_offsets[N+1] is written (exclusive scan of deg); E = d_edge_offsets[N]. Outputs must
 * hold N*k entries. */
int arreau_compact_edges(const int32_t* d_deg, const int32_t* d_src, const int32_t* d_cell,
                         const float* d_dir, const float* d_dist, int32_t N, int32_t k,
                         int32_t* d_edge_offsets, int64_t* d_edge_index /*[2, N*k]*/,
                         float* d_cell_offsets /*[N*k,3]*/, float* d_out_dist, float* d_out_dir,
                         void* stream);

/* Receiver-sorted COO edges (edge_index[1] non-decreasing) -> slot form, for callers that bring
 * their own graph (PonitaFiberBundle.forward takes graph.edge_index, ponita.py:88-106).
 * d_status (int32, device) receives 1 when a receiver has more than k in-edges or the list is
 * not receiver-sorted. */
int arreau_edges_to_slots(const int64_t* d_edge_index /*[2,E]*/, const float* d_dist,
                          const float* d_dir, int64_t E, int32_t N, int32_t k,
                          int32_t* d_deg, int32_t* d_src, float* d_slot_dir, float* d_slot_dist,
                          int32_t* d_status, void* stream);

/* ---- the score network ---------------------------------------------------------------------- */

/* One evaluation of DiffusionLoss.predict_scores (diffusion/diffusion_loss.py:112-197):
 * feature assembly (:124-158), PBC neighbour list (:164-180) unless `use_given_edges`, and
 * PonitaFiberBundle.forward (ponita/models/ponita.py:88-123) with the read-outs of :126-155.
 *   d_frac[N,3], d_types[N] (class index), d_lengths[B,3], d_angles[B,3],
 *   d_t[B] timestep per crystal (the time feature is betas[t], diffusion_loss.py:126).
 * Outputs: d_eps[N,3] (pred_frac_eps_x), d_logits[N,S], d_len0[B,3] (pred_lengths_0).
 * When use_given_edges != 0 the slot arrays d_deg/d_src/d_dir/d_dist are inputs (teacher-forced
 * graph); otherwise they are outputs of the internal neighbour search and may be NULL to use
 * workspace storage.  Of a given graph only slots [0, min(d_deg[i], k)) of receiver i are read as
 * data: the rest may hold anything (NaN included) without changing an output bit. */
int arreau_predict_scores(const arreau_model* model,
                          const float* d_frac, const int32_t* d_types, const float* d_lengths,
                          const float* d_angles, const int32_t* d_t,
                          const int32_t* d_crystal_offsets, int32_t B, int32_t N,
                          int32_t use_given_edges,
                          int32_t* d_deg, int32_t* d_src, float* d_dir, float* d_dist,
                          float* d_eps, float* d_logits, float* d_len0,
                          void* d_workspace, size_t workspace_bytes, void* stream);

/* The inner operator seam: `model(batch)` at diffusion/diffusion_loss.py:183-189, i.e. PonitaFiberBundle.forward
 * (ponita/models/ponita.py:88-123) on the batch attributes the reference assembles at diffusion_loss.py:156-180:
 *   d_x[N, S+74]   batch.x   scalar node features (any values: the embedding is the general x . W^T, :98)
 *   d_vec[N,4,3]   batch.vec (fractional coordinate, then the three lattice rows; diffusion_loss.py:158)
 *   d_lattice[B,3,3] batch.lattice (only the edge cosine features read it, transforms/invariants.py:82-85)
 *   d_crystal_offsets[B+1]  CSR form of batch.batch / batch.num_atoms (atoms of a crystal contiguous)
 *   slot-form edges (arreau_edges_to_slots converts batch.edge_index / dists / inter_atom_direction).
 * Outputs: the reference's return tuple (ponita.py:123) without its None entries:
 *   d_logits[N,S] (output_scalar), d_vec_out[N,1,3] (output_vec), d_global_scalar[B,3] (global_add_pool, :152). */
int arreau_ponita_forward(const arreau_model* model, const float* d_x, const float* d_vec, const float* d_lattice,
                          const int32_t* d_crystal_offsets, int32_t B, int32_t N,
                          const int32_t* d_deg, const int32_t* d_src, const float* d_dir, const float* d_dist,
                          float* d_logits, float* d_vec_out, float* d_global_scalar,
                          void* d_workspace, size_t workspace_bytes, void* stream);

/* The four state updates of one loop iteration (diffusion/diffusion_loss.py:338-347):
 * VP_lattice.reverse_given_x0 on lengths with pred_lengths_0 * num_atoms
 * (diffusion_helpers.py:185-199), lattice_from_params, VE_pbc.reverse on the fractional
 * coordinates (diffusion_helpers.py:65-81) and D3PM.reverse on the atom types (d3pm.py:198-215).
 * Noise is supplied by the caller in the reference's draw order: d_z_lattice[B,3] ~ N(0,1),
 * d_z_frac[N,3] ~ N(0,1), d_u_types[N,S] ~ U[0,1).  State is updated in place; d_lattice[B,3,3]
 * receives the new cell. */
int arreau_reverse_step(const arreau_model* model,
                        float* d_frac, int32_t* d_types, float* d_lengths, const float* d_angles,
                        const int32_t* d_t, const int32_t* d_crystal_offsets, int32_t B, int32_t N,
                        const float* d_eps, const float* d_logits, const float* d_len0,
                        const float* d_z_lattice, const float* d_z_frac, const float* d_u_types,
                        float* d_lattice, void* stream);

/* The hot loop of DiffusionLoss.sample (diffusion/diffusion_loss.py:318-347) enqueued in ONE call: for timestep
 * t_start, t_start-1, ..., t_start-n_steps+1 (the reference runs T-1 .. 1): arreau_predict_scores, then the reverse
 * updates with the three draws of the step (diffusion_helpers.py:193-197, :79; d3pm.py:206) generated INSIDE the update
 * kernels from Philox4x32-10 keyed by (seed, timestep, draw, element) -- no RNG launches, no noise arrays, no host work
 * between steps; the timestep lives on the device.  State (d_frac, d_types, d_lengths) is updated in place, d_lattice
 * [B,3,3] receives the final cell.  d_const_types (may be NULL): species re-imposed after every step
 * (use_constant_atomic_symbols, lightning_wrappers/diffusion.py:231-236).  d_fixed_lengths[B,3] (may be NULL): the
 * same idea for the cell -- fixed-cell sampling, the given lengths re-imposed after every step (an extension; the
 * reference has no counterpart; bench.py uses it to keep the synthetic checkpoint's cells at the sampler's density).
 * use_graph != 0 captures one step into a
 * hipGraph and replays it (same trajectory; pays for itself only on small, launch-bound batches).
 * Does not synchronise. */
int arreau_sample_loop(arreau_model* model, float* d_frac, int32_t* d_types, float* d_lengths, const float* d_angles,
                       const int32_t* d_crystal_offsets, int32_t B, int32_t N, int32_t t_start, int32_t n_steps,
                       uint64_t seed, const int32_t* d_const_types, const float* d_fixed_lengths, float* d_lattice,
                       void* d_workspace, size_t workspace_bytes, int32_t use_graph, void* stream);

/* The sampler's in-kernel noise written out: d_out[i] = draw (seed, timestep, kind, element i) -- standard normal for
 * kind 0 (z_lattice) and 1 (z_frac), uniform [0,1) for kind 2 (u_types); d_raw[4 i .. 4 i + 3] (may be NULL) = the raw
 * Philox4x32-10 words of counter (i, timestep, kind, 0), key = seed.  Feeding these arrays to arreau_reverse_step
 * reproduces arreau_sample_loop's update bit for bit. */
int arreau_philox_fill(uint64_t seed, int32_t timestep, int32_t kind, int64_t n, float* d_out, uint32_t* d_raw,
                       void* stream);

/* ---- score-matching training loss, forward part (BASELINE config 5) ------------------------------------------ */

/* The forward-noising half of DiffusionLoss.__call__ (diffusion/diffusion_loss.py:222-234), random draws supplied by
 * the caller in the reference's order (z_frac: randn_like(frac_x0), diffusion_helpers.py:45; u_types: rand(N,S),
 * d3pm.py:141; z_lengths: randn_like(lengths), diffusion_helpers.py:158); d_t[B] in 1..T (diffusion_loss.py:213-216):
 *   VE_pbc.forward (diffusion_helpers.py:43-63, with min_distance_sqr_pbc :254-325 and cart_to_frac_coords :233-251)
 *     -> d_noisy_frac[N,3], d_target_eps[N,3] (the wrapped fractional noise, in [0,1));
 *   D3PM.get_xt / q_sample (d3pm.py:119-143) -> d_noisy_types[N];
 *   matrix_to_params (lattice_helpers.py:16-35) -> d_lengths[B,3], d_angles[B,3];
 *   VP_lattice.forward (diffusion_helpers.py:156-163) -> d_noisy_lengths[B,3].
 * d_inv_lattice[B,3,3] receives the inverse cells (scratch the caller owns). */
int arreau_diffusion_noise(const arreau_model* model, const float* d_frac0, const int32_t* d_types0,
                           const float* d_lattice0, const int32_t* d_t, const int32_t* d_crystal_offsets,
                           int32_t B, int32_t N, const float* d_z_frac, const float* d_u_types,
                           const float* d_z_lengths, float* d_noisy_frac, float* d_target_eps,
                           int32_t* d_noisy_types, float* d_noisy_lengths, float* d_lengths, float* d_angles,
                           float* d_inv_lattice, void* stream);

/* The three errors of DiffusionLoss.__call__ (diffusion_loss.py:250-274) from the network outputs on the noised state:
 * compute_frac_x_error (:95-110), D3PM.calculate_loss (d3pm.py:146-163: vb * 0.001 + cross entropy) and
 * mse(pred_lengths, lengths / num_atoms) (:264-267); loss weights 1, 1, 1 (:91-93).
 *   d_losses[6] = {loss, error_frac_x, error_atomic_type, error_lattice, vb, ce};  d_terms[N,3] = per-atom scratch.
 * Optional (may be NULL): the gradients of `loss` with respect to the network outputs -- d_grad_eps[N,3],
 * d_grad_logits[N,S], d_grad_lengths[B,3] -- the seeds of the backward pass (what autograd hands to the model in
 * lightning_wrappers/diffusion.py:108-118). */
int arreau_diffusion_losses(const arreau_model* model, const float* d_pred_eps, const float* d_target_eps,
                            const float* d_logits, const int32_t* d_types0, const int32_t* d_noisy_types,
                            const int32_t* d_t, const float* d_pred_lengths, const float* d_lengths,
                            const int32_t* d_crystal_offsets, int32_t B, int32_t N, float* d_terms,
                            float* d_losses, float* d_grad_eps, float* d_grad_logits, float* d_grad_lengths,
                            void* stream);

/* Training-mode evaluation of the score network on a (noised) batch -- same inputs and outputs as
 * arreau_predict_scores, fp32 throughout, every layer's activations kept for the backward pass (the sampling kernels
 * keep none).  PonitaFiberBundle.forward in train mode (ponita/models/ponita.py:88-123). */
int arreau_train_forward(arreau_model* model, const float* d_frac, const int32_t* d_types, const float* d_lengths,
                         const float* d_angles, const int32_t* d_t, const int32_t* d_crystal_offsets, int32_t B,
                         int32_t N, float* d_eps, float* d_logits, float* d_len0, void* stream);

/* Backward pass of the last arreau_train_forward: given d(loss)/d(eps, logits, len0) (arreau_diffusion_losses), the
 * gradient of every trainable tensor, written to the DEVICE arrays `d_grads` points to, in the state_dict layout
 * of arreau_state_dict (entries for buffers -- ori_grid, t_emb_w, schedules, D3PM matrices -- are ignored).  What
 * loss.backward() leaves in .grad in the reference's training_step (lightning_wrappers/diffusion.py:108-118). */
int arreau_train_backward(arreau_model* model, const float* d_grad_eps, const float* d_grad_logits,
                          const float* d_grad_len0, const arreau_state_dict* d_grads, void* stream);

/* After an optimizer step: refresh the fp32 weights the TRAINING entry points read (arreau_train_forward / backward)
 * from the caller's updated tensors -- `d_sd` holds DEVICE pointers in the state_dict layout (buffers ignored).  Only
 * device-to-device copies: no host repacking between steps.  The sampling kernels' packed operand planes are NOT
 * rebuilt; until the model is re-created from the new state_dict, arreau_predict_scores / arreau_ponita_forward /
 * arreau_sample_loop return ARREAU_EINVAL. */
int arreau_model_update_train_weights(arreau_model* model, const arreau_state_dict* d_sd, void* stream);

/* FiberBundleConv.callibrate's inputs (ponita/nn/conv.py:121-123,140-146) from the last arreau_train_forward:
 * d_stats[L][3] = unbiased std of x (layer input), x_1 (after the spatial conv), x_2 (after the spherical conv). */
int arreau_train_conv_stats(arreau_model* model, float* d_stats, void* stream);

/* Gradient clipping + Adam of the training step as two launches on the step's flat gradient buffer (no counterpart as a
 * function in the reference: `gradient_clip_val=0.5` of pl.Trainer, main_diffusion.py:297 = torch.nn.utils.clip_grad_norm_,
 * then torch.optim.Adam over the two parameter groups of configure_optimizers, lightning_wrappers/diffusion.py:152-218).
 * create: a table of `n_tensors` parameter tensors -- d_params[i] (device pointer, contiguous fp32, numel[i] elements), its
 * position flat_offset[i] in the flat gradient buffer of arreau_train_backward's caller (all the d_grads arrays are views of
 * one allocation of flat_len floats) and its parameter group; host arrays, copied.
 * step (t = args->step, counted from 1): norm = |flat gradient|_2 -> *d_norm_out (may be NULL); coefficient
 * min(max_norm / (norm + 1e-6), 1) (max_norm <= 0: none); per element torch's single-tensor Adam: g = grad * coef
 * (+ weight_decay * p), m += (1 - beta1)(g - m), v = beta2 v + (1 - beta2) g g,
 * p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps).  A non-finite norm makes g = 0 for the whole step.
 * d_exp_avg / d_exp_avg_sq: the moments, flat_len floats each, laid out like the gradient buffer (caller-owned: they are the
 * optimizer's state).  Reproducible bit for bit: the norm is a two-stage sum in a fixed order, no atomics.
 * d_mirrors (may be NULL, entries may be NULL): a second destination per tensor for the updated values -- the model's own fp32
 * copy of that tensor (arreau_model_train_weight_pointers), which makes arreau_model_update_train_weights' copies unnecessary;
 * arreau_model_refresh_derived_train_weights then rebuilds the two weights the training entry points read in a derived form
 * (the folded polynomial weight of basis_fn.1, the transposed embedder) from the caller's updated tensors. */
#define ARREAU_OPT_MAX_GROUPS 4
typedef struct arreau_optimizer arreau_optimizer;
typedef struct {   /* doubles: torch forms 1 - beta, 1 - beta^t and lr / (1 - beta1^t) from Python floats before anything is rounded to fp32 */
    int64_t step;
    double lr[ARREAU_OPT_MAX_GROUPS];
    double weight_decay[ARREAU_OPT_MAX_GROUPS];
    double beta1, beta2, eps;
    double max_norm;
} arreau_adam_args;
int arreau_optimizer_create(int32_t n_tensors, void* const* d_params, void* const* d_mirrors, const int64_t* numel,
                            const int64_t* flat_offset, const int32_t* group, int32_t n_groups, int64_t flat_len,
                            arreau_optimizer** out);
int arreau_optimizer_step(arreau_optimizer* opt, const float* d_flat_grad, float* d_exp_avg, float* d_exp_avg_sq,
                          const arreau_adam_args* args, float* d_norm_out, void* stream);
void arreau_optimizer_destroy(arreau_optimizer* opt);
/* DEVICE pointers of the model's own fp32 training weights, stacked [L, ...] in the state_dict layout (NULL: basis_w1,
 * x_embedder_w and every buffer entry). */
int arreau_model_train_weight_pointers(arreau_model* model, arreau_state_dict* out);
int arreau_model_refresh_derived_train_weights(arreau_model* model, const float* d_basis_w1, const float* d_x_embedder_w,
                                               void* stream);

/* The dense product every Linear of the training step runs through (no counterpart in the reference: torch.nn.functional.linear and
 * autograd's matmuls, ponita.py:65-66, conv.py:110-116, convnext.py:24-30), exposed so that the parity tests can call it directly:
 *   C[m][n] = alpha * sum_k A(m, k) B(k, n) + beta * C[m][n],   A(m, k) = d_A[m * as0 + k * as1],  B(k, n) = d_B[k * bs0 + n * bs1]
 * (strides in elements; one of each operand's strides must be 1).  `mode`: 0 = exact fp32 products (v_mfma_f32_32x32x2_f32),
 * 1 = fp16x3, 2 = bf16x6 (three / six 16-bit MFMA products per fp32 product; shapes the split kernel does not take run exact).
 * Device pointers; the split-K scratch is allocated and freed inside (a test hook, not a production entry point). */
int arreau_debug_sgemm(int32_t mode, int32_t M, int32_t N, int32_t K, const float* d_A, int64_t as0, int64_t as1, const float* d_B,
                       int64_t bs0, int64_t bs1, float* d_C, int32_t ldc, float alpha, float beta, void* stream);

/* Timing hook used by bench.py: records hipEvents around the dominant kernel of
 * arreau_predict_scores (the edge kernel) on the stream it is launched on.
 * enable=1 starts collecting; arreau_edge_kernel_time_ms returns the mean over the launches
 * recorded since then (synchronises the events) and the count. */
int arreau_profile_edge_kernel(int32_t enable);
int arreau_edge_kernel_time_ms(double* mean_ms, int64_t* launches);
/* The same for the per-layer message kernel of the default path since round 3 (conv_proj_kernel: kernel projection +
 * message passing + spherical convolution, ponita/nn/conv.py:110-127): L launches per evaluation. */
int arreau_conv_kernel_time_ms(double* mean_ms, int64_t* launches);

/* Debug aid, no reference counterpart: the "uninitialised-state probe".  pattern != 0: every kernel launch of the
 * sampling path is preceded (same stream) by a kernel that fills every CU's LDS and vector registers with `pattern`;
 * 0 switches it off.  A correct kernel's outputs do not depend on what the previous wave left on its CU, so results
 * must be bit-identical for every pattern (eager launches only; process-wide).  tests/test_gpu_parity.py uses it. */
int arreau_debug_set_pollution(uint32_t pattern);
/* The probe's positive control: one kernel per CU READS the whole LDS and 16 vector registers per lane without writing
 * them first and reports the fraction of words equal to `pattern` (close to 1 right after a pollution with it). */
int arreau_debug_leftover_fraction(uint32_t pattern, double* lds_fraction, double* reg_fraction, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ARREAU_HIP_H */
