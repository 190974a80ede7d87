// Gradient clipping + Adam for the training step (BASELINE configs[4]) as TWO launches on the step's flat gradient buffer.
//
// What it replaces: main_diffusion.py:297 (`gradient_clip_val=0.5`: torch.nn.utils.clip_grad_norm_ -- total 2-norm, coefficient
// max_norm / (norm + 1e-6) clamped to 1) followed by torch.optim.Adam over two parameter groups (lightning_wrappers/diffusion.py:
// 152-218: weight decay on Linear weights only).  Through torch that is a norm, nine scalar launches for the coefficient, a scale,
// the non-finite guard and the fused Adam's seven multi-tensor launches: 0.19 ms of device time in a 2.3 ms step
// (profiles/r05n_c5_step_launch_list_with_optimizer.txt), the decayed group's update alone 46 us on 30 workgroups.
//
// Here: `sqnorm_partial_kernel` sums the squares of the flat gradient in OPT_PARTS contiguous chunks (double), `adam_kernel` has every
// workgroup add those partial sums in chunk order (the same total everywhere, whatever the chip does), forms the clip coefficient
// and updates its 1,024 elements of one tensor.  The arithmetic per element is torch's (torch/optim/adam.py, _single_tensor_adam):
//     g  = grad * coef  (0 when the norm is not finite: such a step must not reach the moments, arreau_amd/train.py)
//     g += weight_decay * p
//     m  = m + (1 - beta1) (g - m)                      (torch.lerp)
//     v  = beta2 v + (1 - beta2) g g
//     p -= (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
// The moments live in two flat buffers of the CALLER (torch tensors: they are the optimizer's state_dict), laid out like the
// gradient buffer; parameters stay where the module keeps them (a table of device pointers, built once).
#include <stdint.h>

#include <vector>

#include "internal.h"

#define OPT_PARTS 256
#define OPT_CHUNK 1024

namespace {
struct OptChunk {
    float* p;          // first parameter element of the chunk
    float* mirror;     // second destination of the updated values (the engine's own fp32 copy of the tensor), or null
    uint32_t off;      // its position in the flat gradient / moment buffers
    uint32_t n;        // elements (<= OPT_CHUNK)
    uint32_t group;
    uint32_t pad;
};

__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, long n, double* __restrict__ part) {
    // workgroup b sums elements [b, b + 1) * chunk: float4 loads (the buffer is a torch allocation: 16-byte aligned; chunk is a multiple
    // of four), each thread's few values in double, then the workgroup's tree in a fixed order
    __shared__ double sh[256];
    const long chunk = ((n + OPT_PARTS - 1) / OPT_PARTS + 3) / 4 * 4;
    const long beg = (long)blockIdx.x * chunk, end = beg + chunk < n ? beg + chunk : n;
    double s = 0.0;
    const long n4 = beg < end ? (end - beg) / 4 : 0;
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4* g4 = reinterpret_cast<const f4*>(g + beg);
    for (long i = threadIdx.x; i < n4; i += 256) {
        const f4 v = g4[i];
        s += ((double)v[0] * v[0] + (double)v[1] * v[1]) + ((double)v[2] * v[2] + (double)v[3] * v[3]);
    }
    for (long i = beg + 4 * n4 + threadIdx.x; i < end; i += 256) { const double v = g[i]; s += v * v; }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

struct AdamHyper {
    float lr_over_bc1[ARREAU_OPT_MAX_GROUPS];   // lr / (1 - beta1^t)
    float weight_decay[ARREAU_OPT_MAX_GROUPS];
    float beta2, w1, w2, eps, sqrt_bc2;         // w = 1 - beta, formed in double like torch's host arithmetic; sqrt(1 - beta2^t)
    float max_norm;                             // <= 0: no clipping (the norm is still reported)
};

__global__ __launch_bounds__(256) void adam_kernel(const OptChunk* __restrict__ chunks, const float* __restrict__ grad,
                                                   float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq,
                                                   const double* __restrict__ part, AdamHyper h, float* __restrict__ norm_out) {
    __shared__ double sh[256];
    sh[threadIdx.x] = threadIdx.x < OPT_PARTS ? part[threadIdx.x] : 0.0;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    const float norm = (float)sqrt(sh[0]);
    if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) *norm_out = norm;
    float coef = 1.0f;
    if (h.max_norm > 0.f) coef = fminf(h.max_norm / (norm + 1e-6f), 1.0f);
    if (!isfinite(norm)) coef = 0.f;  // (0 * inf / NaN below would still poison the moments: the select is on the gradient)
    const bool dead = !isfinite(norm);
    const OptChunk c = chunks[blockIdx.x];
    const float step_size = h.lr_over_bc1[c.group], wd = h.weight_decay[c.group];
    const float w1 = h.w1, w2 = h.w2;
    for (uint32_t i = threadIdx.x; i < c.n; i += 256) {
        const float p = c.p[i];
        float g = dead ? 0.f : grad[c.off + i] * coef;
        if (wd != 0.f) g = fmaf(wd, p, g);
        float m = exp_avg[c.off + i], v = exp_avg_sq[c.off + i];
        m = fmaf(w1, g - m, m);
        v = h.beta2 * v + w2 * g * g;
        exp_avg[c.off + i] = m;
        exp_avg_sq[c.off + i] = v;
        const float denom = sqrtf(v) / h.sqrt_bc2 + h.eps;
        const float pn = p - step_size * (m / denom);
        c.p[i] = pn;
        if (c.mirror) c.mirror[i] = pn;
    }
}
}  // namespace

struct arreau_optimizer {
    OptChunk* d_chunks = nullptr;
    double* d_part = nullptr;
    int n_chunks = 0;
    int n_groups = 0;
    int64_t flat_len = 0;
};

extern "C" int arreau_optimizer_create(int32_t n_tensors, void* const* d_params, void* const* d_mirrors, const int64_t* numel,
                                       const int64_t* flat_offset, const int32_t* group, int32_t n_groups, int64_t flat_len,
                                       arreau_optimizer** out) {
    ARREAU_REQUIRE(out, "arreau_optimizer_create: null output");
    *out = nullptr;
    ARREAU_REQUIRE(n_tensors >= 1 && d_params && numel && flat_offset && group, "arreau_optimizer_create: null table");
    ARREAU_REQUIRE(n_groups >= 1 && n_groups <= ARREAU_OPT_MAX_GROUPS, "arreau_optimizer_create: bad group count");
    ARREAU_REQUIRE(flat_len >= 1 && flat_len < (1ll << 32), "arreau_optimizer_create: bad flat buffer length");
    std::vector<OptChunk> chunks;
    for (int i = 0; i < n_tensors; ++i) {
        ARREAU_REQUIRE(d_params[i] != nullptr && numel[i] >= 0, "arreau_optimizer_create: bad tensor");
        ARREAU_REQUIRE(flat_offset[i] >= 0 && flat_offset[i] + numel[i] <= flat_len, "arreau_optimizer_create: tensor outside the flat buffer");
        ARREAU_REQUIRE(group[i] >= 0 && group[i] < n_groups, "arreau_optimizer_create: bad group index");
        for (int64_t o = 0; o < numel[i]; o += OPT_CHUNK) {
            OptChunk c;
            c.p = static_cast<float*>(d_params[i]) + o;
            c.mirror = d_mirrors && d_mirrors[i] ? static_cast<float*>(d_mirrors[i]) + o : nullptr;
            c.off = (uint32_t)(flat_offset[i] + o);
            c.n = (uint32_t)std::min<int64_t>(OPT_CHUNK, numel[i] - o);
            c.group = (uint32_t)group[i];
            c.pad = 0;
            chunks.push_back(c);
        }
    }
    ARREAU_REQUIRE(!chunks.empty(), "arreau_optimizer_create: no elements");
    arreau_optimizer* o = new arreau_optimizer();
    o->n_chunks = (int)chunks.size();
    o->n_groups = n_groups;
    o->flat_len = flat_len;
    hipError_t e = hipMalloc(&o->d_chunks, chunks.size() * sizeof(OptChunk));
    if (e == hipSuccess) e = hipMalloc(&o->d_part, OPT_PARTS * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(o->d_chunks, chunks.data(), chunks.size() * sizeof(OptChunk), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        arreau_set_error(std::string("arreau_optimizer_create: ") + hipGetErrorString(e));
        if (o->d_chunks) (void)hipFree(o->d_chunks);
        if (o->d_part) (void)hipFree(o->d_part);
        delete o;
        return ARREAU_EHIP;
    }
    *out = o;
    return ARREAU_OK;
}

extern "C" void arreau_optimizer_destroy(arreau_optimizer* o) {
    if (!o) return;
    if (o->d_chunks) (void)hipFree(o->d_chunks);
    if (o->d_part) (void)hipFree(o->d_part);
    delete o;
}

extern "C" int arreau_optimizer_step(arreau_optimizer* o, const float* d_flat_grad, float* d_exp_avg, float* d_exp_avg_sq,
                                     const arreau_adam_args* a, float* d_norm_out, void* stream) {
    ARREAU_REQUIRE(o && d_flat_grad && d_exp_avg && d_exp_avg_sq && a, "arreau_optimizer_step: null pointer");
    ARREAU_REQUIRE((size_t)d_flat_grad % 16 == 0, "arreau_optimizer_step: the flat gradient buffer must be 16-byte aligned");
    ARREAU_REQUIRE(a->step >= 1, "arreau_optimizer_step: step counts from 1");
    ARREAU_REQUIRE(a->beta1 >= 0.0 && a->beta1 < 1.0 && a->beta2 >= 0.0 && a->beta2 < 1.0 && a->eps >= 0.0, "arreau_optimizer_step: bad hyper-parameters");
    hipStream_t s = (hipStream_t)stream;
    AdamHyper h;
    // bias corrections in double, as torch's host arithmetic (1 - beta ** step)
    const double bc1 = 1.0 - pow(a->beta1, (double)a->step), bc2 = 1.0 - pow(a->beta2, (double)a->step);
    for (int g = 0; g < ARREAU_OPT_MAX_GROUPS; ++g) {
        h.lr_over_bc1[g] = g < o->n_groups ? (float)(a->lr[g] / bc1) : 0.f;
        h.weight_decay[g] = g < o->n_groups ? (float)a->weight_decay[g] : 0.f;
    }
    h.beta2 = (float)a->beta2; h.eps = (float)a->eps;
    h.w1 = (float)(1.0 - a->beta1); h.w2 = (float)(1.0 - a->beta2);
    h.sqrt_bc2 = (float)sqrt(bc2);
    h.max_norm = (float)a->max_norm;
    ARREAU_LAUNCH(sqnorm_partial_kernel, dim3(OPT_PARTS), dim3(256), 0, s, d_flat_grad, (long)o->flat_len, o->d_part);
    ARREAU_LAUNCH(adam_kernel, dim3((unsigned)o->n_chunks), dim3(256), 0, s, o->d_chunks, d_flat_grad, d_exp_avg, d_exp_avg_sq, o->d_part, h,
                  d_norm_out);
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}
