// arreau_predict_scores: one evaluation of the score network on the current sampler state, plus the
// workspace carve-up and the hipEvent timing hook for the dominant (edge) kernel.
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "internal.h"

namespace {
struct Carver {
    char* base;
    size_t off = 0, cap;
    Carver(void* p, size_t c) : base((char*)p), cap(c) {}
    template <typename T>
    T* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* r = (T*)(base ? base + off : nullptr);
        off += n * sizeof(T);
        return r;
    }
};

struct Workspace {
    float *lattice, *cart, *cvec, *dir, *dist, *kbuf, *xa, *xb, *xc, *xbar, *vsum, *gs;
    float *eps, *logits, *len0;  // network outputs of the sampling loop (arreau_sample_loop)
    int32_t *batch, *deg, *src, *cell, *t_next, *t_cur;
    size_t bytes;
};

Workspace carve(const arreau_config* cfg, int64_t N, int64_t B, void* base, size_t cap) {
    Carver c(base, cap);
    Workspace w;
    const size_t C = cfg->hidden_dim, L = cfg->num_layers, k = cfg->max_neighbors, O = cfg->num_ori;
    w.lattice = c.take<float>(B * 9);
    w.cart = c.take<float>(N * 3);
    w.cvec = c.take<float>(B * C);
    w.batch = c.take<int32_t>(N);
    w.deg = c.take<int32_t>(N);
    w.src = c.take<int32_t>(N * k);
    w.cell = c.take<int32_t>(N * k);
    w.dir = c.take<float>(N * k * 3);
    w.dist = c.take<float>(N * k);
    w.kbuf = c.take<float>(L * N * k * O * C);
    w.xa = c.take<float>(N * O * C);
    w.xb = c.take<float>(N * O * C);
    w.xc = c.take<float>(N * O * C);
    w.xbar = c.take<float>(L * N * C);
    w.vsum = c.take<float>(N * O);
    w.gs = c.take<float>(N * 3);
    w.eps = c.take<float>(N * 3);
    w.logits = c.take<float>(N * (size_t)cfg->num_atomic_states);
    w.len0 = c.take<float>(B * 3);
    w.t_next = c.take<int32_t>(B);
    w.t_cur = c.take<int32_t>(B);
    w.bytes = (c.off + 255) & ~(size_t)255;
    return w;
}

constexpr int MAX_GROUPS = 8;
}  // namespace

// Crystal-aligned slices of a batch (host copy of the CSR offsets -> G ranges of about N / G atoms each) and the streams
// / events the slices run on.  Crystals are independent and every kernel of the network takes a node range over
// whole-batch arrays (NodeRange), so slice g computes bit for bit what the whole-batch launches compute for its atoms;
// what changes is WHEN: slices in different phases overlap the HBM-bound message-passing kernel of one with the
// matrix-bound edge / MLP kernels of another.
struct arreau_partition {
    int B = 0, N = 0, G = 1;
    int eager = 0;  // 1: also slice single evaluations / eager loops (fork-join per step; host-bound, for tests: ARREAU_SLICE_EAGER)
    int nb[MAX_GROUPS + 1], bb[MAX_GROUPS + 1];
    hipStream_t stream[MAX_GROUPS] = {};
    hipEvent_t fork = nullptr, join[MAX_GROUPS] = {}, stagger[MAX_GROUPS] = {};
    hipGraphExec_t exec[MAX_GROUPS] = {};  // per-slice step graphs of the pipelined sampling loop (cached on graph_key)
    uint64_t graph_key[12] = {};
};
void arreau_partition_destroy(arreau_partition* p) {
    if (!p) return;
    for (int g = 0; g < MAX_GROUPS; ++g) {
        if (p->stream[g]) { (void)hipStreamSynchronize(p->stream[g]); (void)hipStreamDestroy(p->stream[g]); }
        if (p->join[g]) (void)hipEventDestroy(p->join[g]);
        if (p->stagger[g]) (void)hipEventDestroy(p->stagger[g]);
        if (p->exec[g]) (void)hipGraphExecDestroy(p->exec[g]);
    }
    if (p->fork) (void)hipEventDestroy(p->fork);
    delete p;
}

extern "C" int arreau_model_set_batch_layout(arreau_model* m, const int32_t* h_off, int32_t B, int32_t groups) {
    ARREAU_REQUIRE(m && (B == 0 || h_off), "arreau_model_set_batch_layout: null pointer");
    static const int env_groups = [] { const char* e = getenv("ARREAU_GROUPS"); return e ? atoi(e) : 0; }();
    int G = groups > 0 ? groups : (env_groups > 0 ? env_groups : 1);
    G = G > MAX_GROUPS ? MAX_GROUPS : G;
    if (B <= 0 || G <= 1 || B < 2 * G) {  // nothing to slice
        if (m->part) m->part->G = 1, m->part->B = -1;
        return ARREAU_OK;
    }
    // Slices on SEPARATE STREAMS are an experiment, not a product mode: with kernels of two streams (or two processes)
    // sharing CUs, one crystal in a few runs came out different at the 1e-8 .. 1e-4 level, and the cause is not known
    // (DESIGN.md section 8 lists what was ruled out: wait-state hazards around the inline asm -- tools/isa_lint.py --,
    // miscounted waits, leftover LDS / register / workspace state).  The library refuses them unless the caller opts in
    // explicitly; ARREAU_SLICE_EAGER=serial (the slices' range launches one after another on ONE stream: what the range
    // launches compute, without any concurrency) stays available to the tests.
    const char* se = getenv("ARREAU_SLICE_EAGER");
    const bool serial_only = se != nullptr && strcmp(se, "serial") == 0;
    const char* allow = getenv("ARREAU_ALLOW_MULTISTREAM");
    ARREAU_REQUIRE(serial_only || (allow && atoi(allow) != 0),
                   "arreau_model_set_batch_layout: slices on separate streams are not reproducible on MI355X (DESIGN.md "
                   "section 8) and are disabled; ARREAU_ALLOW_MULTISTREAM=1 opts in to the experiment");
    arreau_partition* p = m->part ? m->part : new arreau_partition();
    m->part = p;
    const int N = h_off[B];
    p->B = B; p->N = N; p->G = G;
    // ARREAU_SLICE_EAGER: also slice single evaluations (tests).  "serial" = the slices' range launches one after another on
    // the caller's stream (what the range launches compute, without any concurrency); anything else = fork-join on the
    // slice streams.
    p->eager = se == nullptr ? 0 : (serial_only ? 2 : 1);
    p->bb[0] = 0; p->nb[0] = 0;
    int b = 0;
    for (int g = 1; g < G; ++g) {  // cut at the crystal boundary nearest to g N / G (at least one crystal per slice)
        const long long target = (long long)N * g / G;
        while (b < B - (G - g) && h_off[b + 1] <= target) ++b;
        if (b < B - (G - g) && b + 1 <= B && (target - h_off[b]) > (h_off[b + 1] - target)) ++b;
        if (b <= p->bb[g - 1]) b = p->bb[g - 1] + 1;
        p->bb[g] = b;
        p->nb[g] = h_off[b];
    }
    p->bb[G] = B; p->nb[G] = N;
    for (int g = 0; g < G; ++g) {
        if (!p->stream[g]) ARREAU_CHECK_HIP(hipStreamCreateWithFlags(&p->stream[g], hipStreamNonBlocking));
        if (!p->join[g]) ARREAU_CHECK_HIP(hipEventCreateWithFlags(&p->join[g], hipEventDisableTiming));
        if (!p->stagger[g]) ARREAU_CHECK_HIP(hipEventCreateWithFlags(&p->stagger[g], hipEventDisableTiming));
    }
    memset(p->graph_key, 0, sizeof(p->graph_key));  // a new layout invalidates the cached slice graphs
    if (!p->fork) ARREAU_CHECK_HIP(hipEventCreateWithFlags(&p->fork, hipEventDisableTiming));
    return ARREAU_OK;
}

namespace {
// hipEvent pairs around the edge kernel, on the stream it is launched on
struct EdgeProfile {
    std::mutex mu;
    bool enabled = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
} g_prof;
// the same for the per-layer message kernel of the basis form (conv_proj.hip), the dominant kernel since round 3
EdgeProfile g_prof_conv;
}  // namespace

// called by arreau_launch_conv_proj around its launch (internal.h)
void arreau_prof_conv(int end, hipStream_t s) {
    static thread_local hipEvent_t e0 = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_prof_conv.mu);
        if (!g_prof_conv.enabled) return;
    }
    if (!end) {
        if (hipEventCreate(&e0) != hipSuccess) { e0 = nullptr; return; }
        (void)hipEventRecord(e0, s);
        return;
    }
    if (!e0) return;
    hipEvent_t e1 = nullptr;
    if (hipEventCreate(&e1) == hipSuccess) {
        (void)hipEventRecord(e1, s);
        std::lock_guard<std::mutex> lock(g_prof_conv.mu);
        g_prof_conv.events.emplace_back(e0, e1);
    }
    e0 = nullptr;
}

extern "C" size_t arreau_workspace_bytes(const arreau_config* cfg, int64_t max_atoms, int64_t max_crystals) {
    if (!cfg || max_atoms < 0 || max_crystals < 0) return 0;
    return carve(cfg, max_atoms, max_crystals, nullptr, 0).bytes;
}

extern "C" int arreau_profile_edge_kernel(int32_t enable) {
    for (EdgeProfile* p : {&g_prof, &g_prof_conv}) {
        std::lock_guard<std::mutex> lock(p->mu);
        for (auto& ev : p->events) {
            (void)hipEventDestroy(ev.first);
            (void)hipEventDestroy(ev.second);
        }
        p->events.clear();
        p->enabled = enable != 0;
    }
    return ARREAU_OK;
}

extern "C" int arreau_conv_kernel_time_ms(double* mean_ms, int64_t* launches) {
    ARREAU_REQUIRE(mean_ms && launches, "arreau_conv_kernel_time_ms: null pointer");
    std::lock_guard<std::mutex> lock(g_prof_conv.mu);
    double tot = 0.0;
    int64_t n = 0;
    for (auto& ev : g_prof_conv.events) {
        ARREAU_CHECK_HIP(hipEventSynchronize(ev.second));
        float ms = 0.f;
        ARREAU_CHECK_HIP(hipEventElapsedTime(&ms, ev.first, ev.second));
        tot += ms;
        ++n;
    }
    *mean_ms = n ? tot / (double)n : 0.0;
    *launches = n;
    return ARREAU_OK;
}

extern "C" int arreau_edge_kernel_time_ms(double* mean_ms, int64_t* launches) {
    ARREAU_REQUIRE(mean_ms && launches, "arreau_edge_kernel_time_ms: null pointer");
    std::lock_guard<std::mutex> lock(g_prof.mu);
    double tot = 0.0;
    int64_t n = 0;
    for (auto& ev : g_prof.events) {
        ARREAU_CHECK_HIP(hipEventSynchronize(ev.second));
        float ms = 0.f;
        ARREAU_CHECK_HIP(hipEventElapsedTime(&ms, ev.first, ev.second));
        tot += ms;
        ++n;
    }
    *mean_ms = n ? tot / (double)n : 0.0;
    *launches = n;
    return ARREAU_OK;
}

namespace {
// the edge kernel, bracketed by hipEvents on its own stream when bench.py asked for its launch time
int run_edge_kernel(const arreau_model* m, const float* dir, const float* dist, const int32_t* deg, const Workspace& w,
                    int N, hipStream_t s, NodeRange r = NodeRange()) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool prof = false;
    {
        std::lock_guard<std::mutex> lock(g_prof.mu);
        prof = g_prof.enabled;
    }
    if (prof) {
        ARREAU_CHECK_HIP(hipEventCreate(&e0));
        ARREAU_CHECK_HIP(hipEventCreate(&e1));
        ARREAU_CHECK_HIP(hipEventRecord(e0, s));
    }
    const int rc = arreau_launch_edge(m, dir, dist, deg, w.batch, w.lattice, N, w.kbuf, s, r);
    if (prof) {
        ARREAU_CHECK_HIP(hipEventRecord(e1, s));
        std::lock_guard<std::mutex> lock(g_prof.mu);
        g_prof.events.emplace_back(e0, e1);
    }
    return rc;
}

// interaction layers (conv.py:105-129 + convnext.py:20-33) on the embedded features in w.xa, then the read-outs
int run_layers_and_readout(const arreau_model* m, const Workspace& w, const int32_t* deg, const int32_t* src,
                           const int32_t* d_off, int B, int N, float* d_eps, float* d_logits, float* d_len0,
                           hipStream_t s, NodeRange r = NodeRange()) {
    int rc;
    float* xin = w.xa;
    float* xout = w.xb;
    for (int l = 0; l < m->L; ++l) {
        if ((rc = arreau_launch_node_layer(m, l, w.kbuf, deg, src, xin, w.xc, xout, w.xbar, w.vsum, N, s, r))) return rc;
        float* tmp = xin; xin = xout; xout = tmp;
    }
    return arreau_launch_readout(m, w.xbar, w.vsum, d_off, B, N, w.gs, d_eps, d_logits, d_len0, s, r);
}

// The score network after prep_kernel: neighbour list (unless given), edge kernel, embedding, interaction layers,
// read-outs -- for the whole batch on `s`, or slice by slice on the partition's streams (forked from / joined to `s`
// by events, so the caller's stream order is kept and nothing synchronises with the host).
int run_network(const arreau_model* m, const Workspace& w, bool given, int32_t* deg, int32_t* src, float* dir, float* dist,
                const float* d_frac, const int32_t* d_types, const int32_t* d_off, int B, int N, float* d_eps,
                float* d_logits, float* d_len0, hipStream_t s) {
    int rc;
    if (arreau_general_path(m)) {
        // shape-general fp32 kernels (train_net.hip): any hidden_dim / basis_dim / widening, plain weights (never stale)
        arreau_model* mm = const_cast<arreau_model*>(m);  // the activation buffers are a cache owned by the model
        if (!given && (rc = arreau_launch_neighbor(w.cart, w.lattice, d_off, w.batch, B, N, m->cfg.radius, m->k, deg, src, w.cell, dir, dist, s)))
            return rc;
        float* x0 = arreau_general_x0(mm, N, B, s);
        if (!x0) return ARREAU_EHIP;
        if ((rc = arreau_launch_embed(m, d_frac, d_types, w.lattice, w.batch, w.cvec, N, x0, s))) return rc;
        return arreau_general_network(mm, arreau_graph_view{w.batch, deg, src, w.lattice, dir, dist}, d_off, B, N, d_eps, d_logits,
                                      d_len0, s);
    }
    arreau_partition* p = m->part;
    // Fork-join slicing of a single evaluation keeps the slices in lockstep (they start together and have the same work),
    // so it overlaps nothing and costs 4 G extra host calls per step: measured 1.70 / 2.59 ms per step at G = 2 / 4 against
    // 1.66 ms unsliced (256 x 20, eager).  It stays as the test vehicle of the range launches (ARREAU_SLICE_EAGER); the
    // sampling loop uses the pipelined form (arreau_sample_loop).
    const bool sliced = p && p->eager && p->G > 1 && p->B == B && p->N == N && arreau_range_launches_supported(m);
    if (!sliced) {
        if (!given) {
            // neighbour list and embedding side by side in one launch (both need prep_kernel's outputs only)
            if ((rc = arreau_launch_neighbor_embed(m, w.cart, w.lattice, d_off, w.batch, B, N, deg, src, w.cell, dir, dist, d_frac, d_types,
                                                   w.cvec, w.xa, s)))
                return rc;
            if ((rc = run_edge_kernel(m, dir, dist, deg, w, N, s))) return rc;
        } else {
            if ((rc = run_edge_kernel(m, dir, dist, deg, w, N, s))) return rc;
            if ((rc = arreau_launch_embed(m, d_frac, d_types, w.lattice, w.batch, w.cvec, N, w.xa, s))) return rc;
        }
        return run_layers_and_readout(m, w, deg, src, d_off, B, N, d_eps, d_logits, d_len0, s);
    }
    static const int n_cu = [] {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            return (int)prop.multiProcessorCount;
        return 256;
    }();
    static const int cap_env = [] { const char* e = getenv("ARREAU_GROUP_WGS"); return e ? atoi(e) : 0; }();
    const int cap = cap_env > 0 ? cap_env : (n_cu + p->G - 1) / p->G;
    static const int dbg_mode = [] { const char* e = getenv("ARREAU_DEBUG_SLICE_MODE"); return e ? atoi(e) : 0; }();
    if (dbg_mode) {
        // debugging aid: 1 = only the edge kernels run sliced (neighbour lists before, everything else after, on `s`);
        //                2 = neighbour list + edge kernel sliced, the rest whole-batch on `s`
        if (dbg_mode == 1 && !given &&
            (rc = arreau_launch_neighbor(w.cart, w.lattice, d_off, w.batch, B, N, m->cfg.radius, m->k, deg, src, w.cell, dir, dist, s)))
            return rc;
        ARREAU_CHECK_HIP(hipEventRecord(p->fork, s));
        for (int g = 0; g < p->G; ++g) {
            hipStream_t sg = p->stream[g];
            ARREAU_CHECK_HIP(hipStreamWaitEvent(sg, p->fork, 0));
            NodeRange r;
            r.n0 = p->nb[g]; r.n1 = p->nb[g + 1]; r.b0 = p->bb[g]; r.b1 = p->bb[g + 1]; r.wg_cap = cap;
            if (dbg_mode == 2 && !given &&
                (rc = arreau_launch_neighbor(w.cart, w.lattice, d_off, w.batch, B, N, m->cfg.radius, m->k, deg, src, w.cell, dir, dist, sg, r)))
                return rc;
            if ((rc = run_edge_kernel(m, dir, dist, deg, w, N, sg, r))) return rc;
            ARREAU_CHECK_HIP(hipEventRecord(p->join[g], sg));
            ARREAU_CHECK_HIP(hipStreamWaitEvent(s, p->join[g], 0));
        }
        if ((rc = arreau_launch_embed(m, d_frac, d_types, w.lattice, w.batch, w.cvec, N, w.xa, s))) return rc;
        return run_layers_and_readout(m, w, deg, src, d_off, B, N, d_eps, d_logits, d_len0, s);
    }
    const bool serial = p->eager == 2;
    if (!serial) ARREAU_CHECK_HIP(hipEventRecord(p->fork, s));
    for (int g = 0; g < p->G; ++g) {
        hipStream_t sg = serial ? s : p->stream[g];
        if (!serial) ARREAU_CHECK_HIP(hipStreamWaitEvent(sg, p->fork, 0));
        NodeRange r;
        r.n0 = p->nb[g]; r.n1 = p->nb[g + 1]; r.b0 = p->bb[g]; r.b1 = p->bb[g + 1]; r.wg_cap = cap;
        if (!given && (rc = arreau_launch_neighbor(w.cart, w.lattice, d_off, w.batch, B, N, m->cfg.radius, m->k, deg, src, w.cell, dir, dist, sg, r)))
            return rc;
        if ((rc = run_edge_kernel(m, dir, dist, deg, w, N, sg, r))) return rc;
        if ((rc = arreau_launch_embed(m, d_frac, d_types, w.lattice, w.batch, w.cvec, N, w.xa, sg, r))) return rc;
        if ((rc = run_layers_and_readout(m, w, deg, src, d_off, B, N, d_eps, d_logits, d_len0, sg, r))) return rc;
        if (!serial) {
            ARREAU_CHECK_HIP(hipEventRecord(p->join[g], sg));
            ARREAU_CHECK_HIP(hipStreamWaitEvent(s, p->join[g], 0));
        }
    }
    return ARREAU_OK;
}
}  // namespace

extern "C" int arreau_predict_scores(const arreau_model* m, const float* d_frac, const int32_t* d_types,
                                     const float* d_lengths, const float* d_angles, const int32_t* d_t,
                                     const int32_t* d_off, int32_t B, int32_t N, int32_t use_given_edges,
                                     int32_t* d_deg, int32_t* d_src, float* d_dir, float* d_dist, float* d_eps,
                                     float* d_logits, float* d_len0, void* d_workspace, size_t workspace_bytes,
                                     void* stream) {
    ARREAU_REQUIRE(m && d_frac && d_types && d_lengths && d_angles && d_t && d_off && d_eps && d_logits && d_len0,
                   "arreau_predict_scores: null pointer");
    ARREAU_REQUIRE(B >= 1 && N >= 0, "arreau_predict_scores: bad size");
    ARREAU_REQUIRE(!m->packed_stale || arreau_general_path(m),
                   "arreau_predict_scores: weights were updated for training only (arreau_model_update_train_weights); "
                   "re-create the model before sampling, or select the general path (arreau_model_set_variant(model, 5, -1))");
    ARREAU_REQUIRE(d_workspace != nullptr, "arreau_predict_scores: null workspace");
    Workspace w = carve(&m->cfg, N, B, d_workspace, workspace_bytes);
    if (w.bytes > workspace_bytes) {
        arreau_set_error("arreau_predict_scores: workspace too small");
        return ARREAU_ECAPACITY;
    }
    if (use_given_edges) {
        ARREAU_REQUIRE(d_deg && d_src && d_dir && d_dist, "arreau_predict_scores: teacher-forced edges missing");
    }
    int32_t* deg = d_deg ? d_deg : w.deg;
    int32_t* src = d_src ? d_src : w.src;
    float* dir = d_dir ? d_dir : w.dir;
    float* dist = d_dist ? d_dist : w.dist;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if ((rc = arreau_launch_prep(m, d_frac, d_lengths, d_angles, d_t, d_off, B, N, w.lattice, w.cart, w.batch, w.cvec, s)))
        return rc;
    return run_network(m, w, use_given_edges != 0, deg, src, dir, dist, d_frac, d_types, d_off, B, N, d_eps, d_logits, d_len0, s);
}

// PonitaFiberBundle.forward on the reference's own batch attributes (the inner operator seam).
extern "C" int arreau_ponita_forward(const arreau_model* m, const float* d_x, const float* d_vec, const float* d_lattice,
                                     const int32_t* d_off, int32_t B, int32_t N, const int32_t* d_deg,
                                     const int32_t* d_src, const float* d_dir, const float* d_dist, float* d_logits,
                                     float* d_vec_out, float* d_global_scalar, void* d_workspace,
                                     size_t workspace_bytes, void* stream) {
    ARREAU_REQUIRE(m && d_x && d_vec && d_lattice && d_off && d_deg && d_src && d_dir && d_dist && d_logits && d_vec_out &&
                       d_global_scalar, "arreau_ponita_forward: null pointer");
    ARREAU_REQUIRE(B >= 1 && N >= 0, "arreau_ponita_forward: bad size");
    ARREAU_REQUIRE(!m->packed_stale || arreau_general_path(m),
                   "arreau_ponita_forward: weights were updated for training only; re-create the model or select the general path");
    ARREAU_REQUIRE(d_workspace != nullptr, "arreau_ponita_forward: null workspace");
    Workspace w = carve(&m->cfg, N, B, d_workspace, workspace_bytes);
    if (w.bytes > workspace_bytes) {
        arreau_set_error("arreau_ponita_forward: workspace too small");
        return ARREAU_ECAPACITY;
    }
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if ((rc = arreau_launch_batch_index(d_off, B, N, w.batch, s))) return rc;
    w.lattice = const_cast<float*>(d_lattice);  // the edge kernel reads the caller's cells (cos features, invariants.py:82-85)
    if (arreau_general_path(m)) {
        arreau_model* mm = const_cast<arreau_model*>(m);
        float* x0 = arreau_general_x0(mm, N, B, s);
        if (!x0) return ARREAU_EHIP;
        if ((rc = arreau_launch_embed_general(m, d_x, d_vec, N, x0, s))) return rc;
        return arreau_general_network(mm, arreau_graph_view{w.batch, d_deg, d_src, d_lattice, d_dir, d_dist}, d_off, B, N,
                                      d_vec_out, d_logits, d_global_scalar, s);
    }
    if ((rc = run_edge_kernel(m, d_dir, d_dist, d_deg, w, N, s))) return rc;
    if ((rc = arreau_launch_embed_general(m, d_x, d_vec, N, w.xa, s))) return rc;
    return run_layers_and_readout(m, w, d_deg, d_src, d_off, B, N, d_vec_out, d_logits, d_global_scalar, s);
}

// ---------------------------------------------------------------------------------------------
// The hot loop of DiffusionLoss.sample (diffusion/diffusion_loss.py:318-347), enqueued in one call.
// ---------------------------------------------------------------------------------------------
namespace {
__global__ void fill_i32_kernel(int32_t* __restrict__ p, int32_t v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// One step of ONE slice of the batch, entirely on stream `s` (prep, network and updates restricted to the slice's crystals
// and atoms): slices are independent samplers that share the weights, so their chains need no synchronisation at all.
int enqueue_slice_step(const arreau_model* m, float* d_frac, int32_t* d_types, float* d_lengths, const float* d_angles,
                       const int32_t* d_off, int B, int N, uint64_t seed, const int32_t* d_const_types,
                       const float* d_fixed_lengths, float* d_lattice, const Workspace& w, hipStream_t s, NodeRange r,
                       hipEvent_t after_edge) {
    int rc;
    if ((rc = arreau_launch_prep(m, d_frac, d_lengths, d_angles, nullptr, d_off, B, N, w.lattice, w.cart, w.batch, w.cvec, s,
                                 w.t_next, w.t_cur, r)))
        return rc;
    if ((rc = arreau_launch_neighbor_embed(m, w.cart, w.lattice, d_off, w.batch, B, N, w.deg, w.src, w.cell, w.dir, w.dist, d_frac,
                                           d_types, w.cvec, w.xa, s, r)))
        return rc;
    if ((rc = run_edge_kernel(m, w.dir, w.dist, w.deg, w, N, s, r))) return rc;
    if (after_edge) ARREAU_CHECK_HIP(hipEventRecord(after_edge, s));
    // (the per-crystal pooling of the lattice read-out happens inside the lattice update: no launch of its own)
    if ((rc = run_layers_and_readout(m, w, w.deg, w.src, d_off, B, N, w.eps, w.logits, nullptr, s, r))) return rc;
    return arreau_launch_reverse(m, d_frac, d_types, d_lengths, d_angles, w.t_cur, d_off, B, N, w.eps, w.logits, w.len0,
                                 StepNoiseSrc{nullptr, nullptr, nullptr, seed}, d_const_types, d_lattice, s, d_fixed_lengths, r,
                                 w.gs, w.batch);
}

// The single-stream product loop runs WITHOUT a prep launch per step (round 3) when it can: the update launch of step i
// leaves the lattice and the per-crystal embedding of step i + 1 behind (reverse_crystal_block), the neighbour-list waves form
// the Cartesian positions themselves and advance the device-side timestep (neighbor_embed_kernel<LOOP>), and ONE prep launch
// in front of the loop (arreau_sample_loop) supplies the first step.  ARREAU_LOOP_PREP=1 keeps the prep launch in every step
// (A/B, tests: the two loops are bit-identical).
bool loop_without_prep(const arreau_model* m) {
    const char* e = getenv("ARREAU_LOOP_PREP");
    const arreau_partition* p = m->part;
    return !(e && atoi(e) != 0) && !arreau_general_path(m) && !(p && p->eager && p->G > 1);
}

int enqueue_sample_step(const arreau_model* m, float* d_frac, int32_t* d_types, float* d_lengths, const float* d_angles,
                        const int32_t* d_off, int B, int N, uint64_t seed, const int32_t* d_const_types,
                        const float* d_fixed_lengths, float* d_lattice, const Workspace& w, hipStream_t s, bool no_prep) {
    int rc;
    if (no_prep) {
        if ((rc = arreau_launch_neighbor_embed(m, nullptr, w.lattice, d_off, w.batch, B, N, w.deg, w.src, w.cell, w.dir, w.dist, d_frac,
                                               d_types, w.cvec, w.xa, s, NodeRange(), w.t_cur)))
            return rc;
        if ((rc = run_edge_kernel(m, w.dir, w.dist, w.deg, w, N, s))) return rc;
        if ((rc = run_layers_and_readout(m, w, w.deg, w.src, d_off, B, N, w.eps, w.logits, nullptr, s))) return rc;
        return arreau_launch_reverse(m, d_frac, d_types, d_lengths, d_angles, w.t_cur, d_off, B, N, w.eps, w.logits, w.len0,
                                     StepNoiseSrc{nullptr, nullptr, nullptr, seed}, d_const_types, d_lattice, s, d_fixed_lengths,
                                     NodeRange(), w.gs, w.batch, w.lattice, w.cvec);
    }
    if ((rc = arreau_launch_prep(m, d_frac, d_lengths, d_angles, nullptr, d_off, B, N, w.lattice, w.cart, w.batch, w.cvec, s,
                                 w.t_next, w.t_cur)))
        return rc;
    // fused kernels: the per-crystal pooling of the lattice read-out happens inside the lattice update (no launch of its own)
    const bool pool_in_update = !arreau_general_path(m);
    if ((rc = run_network(m, w, false, w.deg, w.src, w.dir, w.dist, d_frac, d_types, d_off, B, N, w.eps, w.logits,
                          pool_in_update ? nullptr : w.len0, s)))
        return rc;
    return arreau_launch_reverse(m, d_frac, d_types, d_lengths, d_angles, w.t_cur, d_off, B, N, w.eps, w.logits, w.len0,
                                 StepNoiseSrc{nullptr, nullptr, nullptr, seed}, d_const_types, d_lattice, s, d_fixed_lengths,
                                 NodeRange(), pool_in_update ? w.gs : nullptr, w.batch);
}
}  // namespace

extern "C" int arreau_sample_loop(arreau_model* m, float* d_frac, int32_t* d_types, float* d_lengths,
                                  const float* d_angles, const int32_t* d_off, int32_t B, int32_t N, int32_t t_start,
                                  int32_t n_steps, uint64_t seed, const int32_t* d_const_types,
                                  const float* d_fixed_lengths, float* d_lattice, void* d_workspace, size_t workspace_bytes, int32_t use_graph, void* stream) {
    ARREAU_REQUIRE(m && d_frac && d_types && d_lengths && d_angles && d_off && d_lattice, "arreau_sample_loop: null pointer");
    ARREAU_REQUIRE(B >= 1 && N >= 0 && n_steps >= 0, "arreau_sample_loop: bad size");
    ARREAU_REQUIRE(!m->packed_stale || arreau_general_path(m),
                   "arreau_sample_loop: weights were updated for training only; re-create the model or select the general path");
    ARREAU_REQUIRE(t_start <= m->T && t_start - n_steps >= 0, "arreau_sample_loop: timesteps t_start .. t_start-n_steps+1 must lie in 1..T");
    ARREAU_REQUIRE(d_workspace != nullptr, "arreau_sample_loop: null workspace");
    Workspace w = carve(&m->cfg, N, B, d_workspace, workspace_bytes);
    if (w.bytes > workspace_bytes) {
        arreau_set_error("arreau_sample_loop: workspace too small");
        return ARREAU_ECAPACITY;
    }
    if (n_steps == 0) return ARREAU_OK;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    // (slices on their own streams -- the opt-in experiment below -- keep the prep launch per step)
    const bool sliced_loop = use_graph && n_steps >= 3 && m->part && m->part->G > 1 && m->part->eager != 2 && m->part->B == B &&
                             m->part->N == N && arreau_range_launches_supported(m);
    const bool no_prep = !sliced_loop && loop_without_prep(m);
    if (no_prep) {
        // t_cur holds the timestep of the step in progress; every step's first launch advances it, so it starts one above
        ARREAU_LAUNCH(fill_i32_kernel, dim3((B + 255) / 256), dim3(256), 0, s, w.t_cur, t_start + 1, B);
        ARREAU_CHECK_HIP(hipGetLastError());
        if ((rc = arreau_launch_prep(m, d_frac, d_lengths, d_angles, w.t_cur, d_off, B, N, w.lattice, w.cart, w.batch, w.cvec, s, nullptr,
                                     nullptr, NodeRange(), -1)))
            return rc;
    } else {
        ARREAU_LAUNCH(fill_i32_kernel, dim3((B + 255) / 256), dim3(256), 0, s, w.t_next, t_start, B);
        ARREAU_CHECK_HIP(hipGetLastError());
    }
    if (!use_graph || n_steps < 3) {
        for (int i = 0; i < n_steps; ++i)
            if ((rc = enqueue_sample_step(m, d_frac, d_types, d_lengths, d_angles, d_off, B, N, seed, d_const_types, d_fixed_lengths, d_lattice, w, s, no_prep)))
                return rc;
        return ARREAU_OK;
    }
    // Pipelined slices: when the batch has a layout with G > 1 slices (arreau_model_set_batch_layout), every slice runs its
    // own chain of steps on its own stream -- one captured graph per slice, replayed n_steps - 1 times -- with NO
    // synchronisation between slices until the end of the loop.  The first steps are staggered (slice g starts when slice
    // g - 1 has finished its edge kernel), so the slices stay in different phases: while one is in its matrix-bound edge
    // kernel another streams its K blocks from HBM.  Every slice computes exactly what it computes in the whole-batch run.
    {
        arreau_partition* p = m->part;
        if (p && p->G > 1 && p->eager != 2 && p->B == B && p->N == N && arreau_range_launches_supported(m)) {
            static const int n_cu = [] {
                int dev = 0;
                hipDeviceProp_t prop;
                if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                    return (int)prop.multiProcessorCount;
                return 256;
            }();
            static const int cap_env = [] { const char* e = getenv("ARREAU_GROUP_WGS"); return e ? atoi(e) : 0; }();
            const int cap = cap_env > 0 ? cap_env : (n_cu + p->G - 1) / p->G;
            const uint64_t key[12] = {(uint64_t)d_frac, (uint64_t)d_types, (uint64_t)d_lengths, (uint64_t)d_angles, (uint64_t)d_off,
                                      ((uint64_t)(uint32_t)B << 32) | (uint32_t)N, seed, (uint64_t)d_const_types,
                                      (uint64_t)d_fixed_lengths, (uint64_t)d_lattice, (uint64_t)d_workspace,
                                      ((uint64_t)(uint32_t)p->G << 32) | (uint32_t)cap};
            const bool cached = p->exec[0] && memcmp(key, p->graph_key, sizeof(key)) == 0;
            ARREAU_CHECK_HIP(hipEventRecord(p->fork, s));
            NodeRange r[MAX_GROUPS];
            for (int g = 0; g < p->G; ++g) {
                r[g].n0 = p->nb[g]; r[g].n1 = p->nb[g + 1]; r[g].b0 = p->bb[g]; r[g].b1 = p->bb[g + 1]; r[g].wg_cap = cap;
                ARREAU_CHECK_HIP(hipStreamWaitEvent(p->stream[g], p->fork, 0));
                if (g > 0) ARREAU_CHECK_HIP(hipStreamWaitEvent(p->stream[g], p->stagger[g - 1], 0));
                // first step eagerly (stagger point + lazy module loading outside any capture)
                if ((rc = enqueue_slice_step(m, d_frac, d_types, d_lengths, d_angles, d_off, B, N, seed, d_const_types,
                                             d_fixed_lengths, d_lattice, w, p->stream[g], r[g], p->stagger[g])))
                    return rc;
            }
            if (!cached) {
                for (int g = 0; g < p->G; ++g) {
                    if (p->exec[g]) {
                        (void)hipStreamSynchronize(p->stream[g]);
                        (void)hipGraphExecDestroy(p->exec[g]);
                        p->exec[g] = nullptr;
                    }
                    hipGraph_t graph = nullptr;
                    ARREAU_CHECK_HIP(hipStreamBeginCapture(p->stream[g], hipStreamCaptureModeThreadLocal));
                    rc = enqueue_slice_step(m, d_frac, d_types, d_lengths, d_angles, d_off, B, N, seed, d_const_types,
                                            d_fixed_lengths, d_lattice, w, p->stream[g], r[g], nullptr);
                    hipError_t e = hipStreamEndCapture(p->stream[g], &graph);
                    if (rc) {
                        if (graph) (void)hipGraphDestroy(graph);
                        return rc;
                    }
                    ARREAU_CHECK_HIP(e);
                    e = hipGraphInstantiate(&p->exec[g], graph, nullptr, nullptr, 0);
                    (void)hipGraphDestroy(graph);
                    ARREAU_CHECK_HIP(e);
                }
                memcpy(p->graph_key, key, sizeof(key));
            }
            for (int i = 1; i < n_steps; ++i)
                for (int g = 0; g < p->G; ++g) ARREAU_CHECK_HIP(hipGraphLaunch(p->exec[g], p->stream[g]));
            for (int g = 0; g < p->G; ++g) {
                ARREAU_CHECK_HIP(hipEventRecord(p->join[g], p->stream[g]));
                ARREAU_CHECK_HIP(hipStreamWaitEvent(s, p->join[g], 0));
            }
            return ARREAU_OK;
        }
    }
    // One step captured into a hipGraph and replayed: the timestep lives on the device (prep_kernel advances it), the noise
    // is a function of (seed, timestep, element), so every replay is the next step of the same trajectory as the eager loop.
    // Capture is not allowed on the legacy default stream (which is what callers usually pass), so the loop runs on a
    // stream of the model's own, joined to the caller's stream by events on both sides: still no host synchronisation.
    if (!m->loop_stream) {
        hipStream_t ls = nullptr;
        ARREAU_CHECK_HIP(hipStreamCreateWithFlags(&ls, hipStreamNonBlocking));
        hipEvent_t ev = nullptr;
        ARREAU_CHECK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        m->loop_stream = (void*)ls;
        m->loop_event = (void*)ev;
    }
    hipStream_t user = s;
    hipEvent_t ev = (hipEvent_t)m->loop_event;
    s = (hipStream_t)m->loop_stream;
    ARREAU_CHECK_HIP(hipEventRecord(ev, user));
    ARREAU_CHECK_HIP(hipStreamWaitEvent(s, ev, 0));
    // The executable graph is kept with the model and reused while the next call names the same buffers, sizes and seed
    // (a sampler drawing sub-batch after sub-batch through the caching allocator does): capture + instantiation, about
    // 2 ms, are then paid once.  The timestep is not part of the graph (it lives in t_next / t_cur, set above).
    const uint64_t key[12] = {(uint64_t)d_frac, (uint64_t)d_types, (uint64_t)d_lengths, (uint64_t)d_angles, (uint64_t)d_off,
                              ((uint64_t)(uint32_t)B << 32) | (uint32_t)N, seed, (uint64_t)d_const_types, (uint64_t)d_fixed_lengths,
                              (uint64_t)d_lattice, (uint64_t)d_workspace,
                              ((uint64_t)(uint32_t)(m->edge_variant | (no_prep ? 0x10000 : 0) |
                                                    // which kernels a capture holds also depends on switches read per call
                                                    // (ARREAU_BASIS_MIN_RECEIVERS, ARREAU_FUSE_SMALL) and on the conv variant:
                                                    // a changed switch must not replay the stale graph
                                                    (arreau_basis_form(m, N) ? 0x20000 : 0) | (arreau_basis_fp8(m) ? 0x40000 : 0) | (arreau_cross_fp8(m) ? 0x400000 : 0) |
                                                    (arreau_small_layer_fusable(m, N, NodeRange()) ? 0x80000 : 0) |
                                                    ((m->conv_variant & 3) << 20)) << 32) | (uint32_t)m->mlp_variant};
    hipGraphExec_t exec = (hipGraphExec_t)m->retired_graph;
    int first_replay = 0;
    hipError_t e = hipSuccess;
    if (!exec || memcmp(key, m->graph_key, sizeof(key)) != 0) {
        // The first step runs eagerly (it also forces lazy module loading, which must not happen inside a capture).
        if ((rc = enqueue_sample_step(m, d_frac, d_types, d_lengths, d_angles, d_off, B, N, seed, d_const_types, d_fixed_lengths, d_lattice, w, s, no_prep)))
            return rc;
        first_replay = 1;
        hipGraph_t graph = nullptr;
        ARREAU_CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        rc = enqueue_sample_step(m, d_frac, d_types, d_lengths, d_angles, d_off, B, N, seed, d_const_types, d_fixed_lengths, d_lattice, w, s, no_prep);
        e = hipStreamEndCapture(s, &graph);
        if (rc) {
            if (graph) (void)hipGraphDestroy(graph);
            return rc;
        }
        ARREAU_CHECK_HIP(e);
        exec = nullptr;
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        ARREAU_CHECK_HIP(e);
        arreau_model_retire_graph(m, (void*)exec, (void*)s);  // takes ownership; frees the previous one after its stream drained
        memcpy(m->graph_key, key, sizeof(key));
    }
    for (int i = first_replay; i < n_steps; ++i) {
        e = hipGraphLaunch(exec, s);
        if (e != hipSuccess) break;
    }
    if (e == hipSuccess) e = hipEventRecord(ev, s);
    if (e == hipSuccess) e = hipStreamWaitEvent(user, ev, 0);  // the caller's stream continues after the loop
    ARREAU_CHECK_HIP(e);
    return ARREAU_OK;
}

// ---------------------------------------------------------------------------------------------
// Uninitialised-state probe (internal.h: ARREAU_LAUNCH).  One workgroup per CU takes the CU's whole LDS (160 KiB, so no
// second one fits beside it) and 16 waves x ~112 registers per lane = most of the four SIMDs' register files, writes the
// pattern everywhere and leaves.  The workgroups wait for each other (bounded: a clock limit, so the grid always drains)
// to make sure none of them is placed on a CU another one has already left.
// ---------------------------------------------------------------------------------------------
unsigned arreau_debug_pollution = 0;
namespace {
__global__ __launch_bounds__(1024) void pollute_kernel(unsigned pattern, unsigned* __restrict__ arrived, unsigned target) {
    extern __shared__ unsigned pl[];
    for (int i = threadIdx.x; i < 40960; i += 1024) pl[i] = pattern;
    float r[112];
#pragma unroll
    for (int i = 0; i < 112; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "v"(pattern));
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(arrived, 1u);
        const long long t0 = wall_clock64();  // 100 MHz
        while (__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && wall_clock64() - t0 < 20000) {}
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 112; ++i) asm volatile("" : : "v"(r[i]));
    if (pl[(threadIdx.x * 37) % 40960] != pattern) arrived[1] = 1;  // (keeps the LDS writes alive; never true)
}
unsigned* g_pollute_counter = nullptr;
unsigned g_pollute_launches = 0;
}  // namespace

int arreau_debug_pollute(hipStream_t s) {
    static const int n_cu = [] {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            return (int)prop.multiProcessorCount;
        return 256;
    }();
    if (!g_pollute_counter) {
        ARREAU_CHECK_HIP(hipMalloc(&g_pollute_counter, 8));
        ARREAU_CHECK_HIP(hipMemset(g_pollute_counter, 0, 8));
        ARREAU_CHECK_HIP(hipFuncSetAttribute((const void*)pollute_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    }
    ++g_pollute_launches;
    hipLaunchKernelGGL(pollute_kernel, dim3(n_cu), dim3(1024), 163840, s, arreau_debug_pollution, g_pollute_counter,
                       g_pollute_launches * (unsigned)n_cu);
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}

// Positive control of the probe: a kernel that READS what it never wrote -- every CU's whole LDS and 32 vector registers
// per lane -- and counts the words equal to `pattern`.  Launched through ARREAU_LAUNCH with the pollution switched on it
// must find (nearly) nothing else; the test asserts that, so a clean probe run means "no dependence on leftovers", not
// "the polluter did not reach them".
namespace {
__global__ __launch_bounds__(1024) void leftover_kernel(unsigned pattern, unsigned long long* __restrict__ counts) {
    extern __shared__ unsigned pl[];
    unsigned lds_hits = 0, reg_hits = 0;
    for (int i = threadIdx.x; i < 40960; i += 1024) lds_hits += pl[i] == pattern;
    asm volatile("" ::: "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109",
                 "v110", "v111");  // make the registers read below part of this wave's allocation
#define ARREAU_LEFTOVER_REG(n) { unsigned x; asm volatile("v_mov_b32 %0, v" #n : "=v"(x)); reg_hits += x == pattern; }
    ARREAU_LEFTOVER_REG(96) ARREAU_LEFTOVER_REG(97) ARREAU_LEFTOVER_REG(98) ARREAU_LEFTOVER_REG(99)
    ARREAU_LEFTOVER_REG(100) ARREAU_LEFTOVER_REG(101) ARREAU_LEFTOVER_REG(102) ARREAU_LEFTOVER_REG(103)
    ARREAU_LEFTOVER_REG(104) ARREAU_LEFTOVER_REG(105) ARREAU_LEFTOVER_REG(106) ARREAU_LEFTOVER_REG(107)
    ARREAU_LEFTOVER_REG(108) ARREAU_LEFTOVER_REG(109) ARREAU_LEFTOVER_REG(110) ARREAU_LEFTOVER_REG(111)
#undef ARREAU_LEFTOVER_REG
    atomicAdd(&counts[0], (unsigned long long)lds_hits);
    atomicAdd(&counts[1], (unsigned long long)reg_hits);
}
}  // namespace

extern "C" int arreau_debug_leftover_fraction(uint32_t pattern, double* lds_fraction, double* reg_fraction, void* stream) {
    ARREAU_REQUIRE(lds_fraction && reg_fraction, "arreau_debug_leftover_fraction: null pointer");
    int dev = 0, n_cu = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        n_cu = prop.multiProcessorCount;
    unsigned long long* d_counts = nullptr;
    ARREAU_CHECK_HIP(hipMalloc(&d_counts, 16));
    hipStream_t s = (hipStream_t)stream;
    ARREAU_CHECK_HIP(hipMemsetAsync(d_counts, 0, 16, s));
    ARREAU_CHECK_HIP(hipFuncSetAttribute((const void*)leftover_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    ARREAU_LAUNCH(leftover_kernel, dim3(n_cu), dim3(1024), 163840, s, pattern, d_counts);
    ARREAU_CHECK_HIP(hipGetLastError());
    unsigned long long h[2] = {0, 0};
    ARREAU_CHECK_HIP(hipMemcpyAsync(h, d_counts, 16, hipMemcpyDeviceToHost, s));
    ARREAU_CHECK_HIP(hipStreamSynchronize(s));
    ARREAU_CHECK_HIP(hipFree(d_counts));
    *lds_fraction = (double)h[0] / ((double)n_cu * 40960.0);
    *reg_fraction = (double)h[1] / ((double)n_cu * 1024.0 * 16.0);
    return ARREAU_OK;
}

extern "C" int arreau_debug_set_pollution(uint32_t pattern) {
    arreau_debug_pollution = pattern;
    return ARREAU_OK;
}
