// Node-side kernels of the score network: feature assembly + embedding, the per-layer
// message/spherical-conv/ConvNext update, and the read-outs.
#include "internal.h"

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
__device__ __forceinline__ void cell_from_params(const float* len, const float* ang, float* Lm) {
    const float a = len[0], b = len[1], c = len[2];
    const float ca = cosf(ang[0]), cb = cosf(ang[1]), cg = cosf(ang[2]);
    const float sa = sinf(ang[0]), sb = sinf(ang[1]);
    float val = (ca * cb - cg) / (sa * sb);
    val = fminf(fmaxf(val, -1.0f), 1.0f);
    const float gs = acosf(val);
    Lm[0] = a * sb;             Lm[1] = 0.0f;               Lm[2] = a * cb;
    Lm[3] = -b * sa * cosf(gs); Lm[4] = b * sa * sinf(gs);  Lm[5] = b * ca;
    Lm[6] = 0.0f;               Lm[7] = 0.0f;               Lm[8] = c;
}

__global__ __launch_bounds__(128) void prep_kernel(
    const float* __restrict__ frac, const float* __restrict__ lengths, const float* __restrict__ angles,
    const int32_t* __restrict__ tstep, const int32_t* __restrict__ offsets, const float* __restrict__ betas,
    const float* __restrict__ t_emb_w, const float* __restrict__ embT, int S, int C, int T,
    float* __restrict__ lattice, float* __restrict__ cart, int32_t* __restrict__ batch, float* __restrict__ cvec) {
    __shared__ float feat[ARREAU_T_EMB_DIM + ARREAU_N_CRYSTAL_FEATS];
    __shared__ float Lm[9];
    const int b = blockIdx.x;
    const int first = offsets[b], n = offsets[b + 1] - first;
    const float* len = lengths + 3 * b;
    const float* ang = angles + 3 * b;
    if (threadIdx.x == 0) {
        float tmp[9];
        cell_from_params(len, ang, tmp);
        for (int i = 0; i < 9; ++i) { Lm[i] = tmp[i]; lattice[9 * b + i] = tmp[i]; }
    }
    if (threadIdx.x < 32) {
        // GaussianFourierProjection of betas[t] (diffusion_helpers.py:23-25; diffusion_loss.py:126-127)
        int t = tstep[b];
        t = t < 0 ? 0 : (t > T ? T : t);
        const float proj = ((betas[t] * t_emb_w[threadIdx.x]) * 2.0f) * 3.14159265358979323846f;
        feat[threadIdx.x] = sinf(proj);
        feat[32 + threadIdx.x] = cosf(proj);
    } else if (threadIdx.x < 32 + ARREAU_N_CRYSTAL_FEATS) {
        const int i = threadIdx.x - 32;
        float v;
        if (i == 0) v = (float)n;
        else if (i < 4) v = len[i - 1];
        else if (i < 7) v = ang[i - 4];
        else v = fabsf(len[i - 7] / (float)n);
        feat[ARREAU_T_EMB_DIM + i] = v;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.0f;
        for (int i = 0; i < ARREAU_T_EMB_DIM + ARREAU_N_CRYSTAL_FEATS; ++i) acc += feat[i] * embT[(size_t)(S + i) * C + c];
        cvec[(size_t)b * C + c] = acc;
    }
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
                       int32_t* batch, float* cvec, hipStream_t s) {
    if (B == 0) return ARREAU_OK;
    hipLaunchKernelGGL(prep_kernel, dim3(B), dim3(128), 0, s, frac, lengths, angles, t, offsets, m->vp_betas,
                       m->t_emb_w, m->embT, m->S, m->C, m->T, lattice, cart, batch, cvec);
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}

// ---------------------------------------------------------------------------------------------
// Embedding: x0[n][o][c] = embT[type_n][c] + cvec[b][c] + sum_v embT[S+74+v][c] * (vec[n][v] . ori[o])
// with vec[n] = (frac_n, lattice rows a, b, c)  (diffusion_loss.py:158; to_from_sphere.py:4-5).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_kernel(
    const float* __restrict__ frac, const int32_t* __restrict__ types, const float* __restrict__ lattice,
    const int32_t* __restrict__ batch, const float* __restrict__ cvec, const float* __restrict__ ori,
    const float* __restrict__ embT, int S, int C, int N, float* __restrict__ x0) {
    const int C4 = C / 4;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)N * ARREAU_ORI * C4) return;
    const int c4 = (int)(idx % C4);
    const int o = (int)((idx / C4) % ARREAU_ORI);
    const int n = (int)(idx / ((long long)C4 * ARREAU_ORI));
    const int b = batch[n];
    int ty = types[n];
    ty = ty < 0 ? 0 : (ty >= S ? S - 1 : ty);
    const float ox = ori[3 * o], oy = ori[3 * o + 1], oz = ori[3 * o + 2];
    const float* Lm = lattice + 9 * (size_t)b;
    float proj[4];
    proj[0] = (frac[3 * (size_t)n] * ox + frac[3 * (size_t)n + 1] * oy) + frac[3 * (size_t)n + 2] * oz;
#pragma unroll
    for (int v = 0; v < 3; ++v) proj[1 + v] = (Lm[3 * v] * ox + Lm[3 * v + 1] * oy) + Lm[3 * v + 2] * oz;
    const f32x4* e4 = reinterpret_cast<const f32x4*>(embT);
    f32x4 acc = e4[(size_t)ty * C4 + c4] + reinterpret_cast<const f32x4*>(cvec)[(size_t)b * C4 + c4];
#pragma unroll
    for (int v = 0; v < 4; ++v) acc += e4[(size_t)(S + 74 + v) * C4 + c4] * proj[v];
    reinterpret_cast<f32x4*>(x0)[idx] = acc;
}

int arreau_launch_embed(const arreau_model* m, const float* frac, const int32_t* types, const float* lattice,
                        const int32_t* batch, const float* cvec, int N, float* x0, hipStream_t s) {
    if (N == 0) return ARREAU_OK;
    const long long total = (long long)N * ARREAU_ORI * (m->C / 4);
    hipLaunchKernelGGL(embed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, frac, types, lattice, batch,
                       cvec, m->ori, m->embT, m->S, m->C, N, x0);
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}

// ---------------------------------------------------------------------------------------------
// K4: one interaction layer for a tile of 2 nodes (32 (node, orientation) rows), 4 waves.
//   A  message + aggregate:  x1[n,o,c] = sum_{slots s<deg[n]} K_l[(n,s),o,c] * x[src(n,s),o,c]
//                            (conv.py:111,131-133 + PyG sum aggregation onto the receiver)
//   B  spherical conv:       x2[n,p,c] = sum_o x1[n,o,c] * FK_l[o,p,c] + bias[c]   (FK holds the /O; conv.py:113-127)
//   C  LayerNorm over c (eps 1e-5)                                                 (convnext.py:25)
//   D  MLP on MFMA: each wave owns one quarter of the hidden units:
//          hid = GELU(W1[quarter] . xn + b1[quarter]);  part = W2[:, quarter] . hid  (convnext.py:26-28)
//   E  deterministic cross-wave sum of the 4 partial outputs through LDS
//   F  x_out = (sum + b2) * layer_scale + x_in ; per-layer read-out partials:
//          xbar_l[n][c] = mean_o x_out ;  vsum[n][o] (+)= w_vec_l . x_out[n,o,:] + b_vec_l   (ponita.py:105-117)
// ---------------------------------------------------------------------------------------------
#define NODE_TILE_ROWS 32
#define NODE_LDS_STRIDE 132  // C + 4 floats: 16-byte aligned rows, conflict-free ds_read_b128 of 4 k-steps

template <int C, int H>
__global__ __launch_bounds__(256, 2) void node_layer_kernel(
    const float* __restrict__ kl,        // this layer's kernels [N*k*16][C]
    const int32_t* __restrict__ deg, const int32_t* __restrict__ src,
    const float* __restrict__ x_in, float* __restrict__ x_out,
    const float* __restrict__ fk,        // [16(o)][16(p)][C]
    const float* __restrict__ conv_bias, const float* __restrict__ ln_w, const float* __restrict__ ln_b,
    const float* __restrict__ m1p, const float* __restrict__ mb1, const float* __restrict__ m2p,
    const float* __restrict__ mb2, const float* __restrict__ ls,
    const float* __restrict__ ro_wT,     // [C][S+4] this layer
    const float* __restrict__ ro_b,      // [S+4]
    int S, int N, int k, int first_layer,
    float* __restrict__ xbar,            // [N][C] this layer
    float* __restrict__ vsum)            // [N][16]
{
    static_assert(C == 128, "tile mapping below assumes C = 128");
    constexpr int TC = C / 32;          // in/out tiles of C
    constexpr int HQ = H / 4;           // hidden units per wave
    constexpr int THQ = HQ / 32;        // hidden tiles per wave
    __shared__ __attribute__((aligned(16))) float xt[NODE_TILE_ROWS * NODE_LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) float red[NODE_TILE_ROWS * NODE_LDS_STRIDE];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n0 = 2 * blockIdx.x;

    // ---- A: gather + multiply + segmented sum (4 rows per thread, 4 channels per thread) --------
    {
        const int c4 = tid & 31, rr = tid >> 5;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = rr + 8 * i;
            const int n = n0 + (r >> 4), o = r & 15;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (n < N) {
                const int nd = min(deg[n], k);
                for (int s = 0; s < nd; ++s) {
                    const int sn = src[(size_t)n * k + s];
                    const f32x4 kv = *reinterpret_cast<const f32x4*>(kl + (((size_t)n * k + s) * 16 + o) * C + 4 * c4);
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(x_in + ((size_t)sn * 16 + o) * C + 4 * c4);
                    // product rounded, then added in edge order, like messages -> index_add_
                    acc[0] = __fadd_rn(acc[0], __fmul_rn(kv[0], xv[0]));
                    acc[1] = __fadd_rn(acc[1], __fmul_rn(kv[1], xv[1]));
                    acc[2] = __fadd_rn(acc[2], __fmul_rn(kv[2], xv[2]));
                    acc[3] = __fadd_rn(acc[3], __fmul_rn(kv[3], xv[3]));
                }
            }
            *reinterpret_cast<f32x4*>(&xt[r * NODE_LDS_STRIDE + 4 * c4]) = acc;
        }
    }
    __syncthreads();

    // ---- B: depth-wise spherical convolution ------------------------------------------------------
    {
        const int c = tid & 127, ph = tid >> 7;  // channel, half of the output orientations
        float acc[2][8];
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
            for (int p = 0; p < 8; ++p) acc[n2][p] = 0.f;
#pragma unroll 4
        for (int o = 0; o < 16; ++o) {
            const float xa = xt[o * NODE_LDS_STRIDE + c];
            const float xb = xt[(16 + o) * NODE_LDS_STRIDE + c];
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const float f = fk[((size_t)o * 16 + (8 * ph + p)) * C + c];
                acc[0][p] += xa * f;
                acc[1][p] += xb * f;
            }
        }
        const float bias = conv_bias[c];
        __syncthreads();  // every read of x1 is done before it is overwritten with x2
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
            for (int p = 0; p < 8; ++p) xt[(16 * n2 + 8 * ph + p) * NODE_LDS_STRIDE + c] = acc[n2][p] + bias;
    }
    __syncthreads();

    // ---- C: LayerNorm over channels, 8 rows per wave, 2 channels per lane --------------------------
    {
        const float g0 = ln_w[lane], g1 = ln_w[lane + 64], be0 = ln_b[lane], be1 = ln_b[lane + 64];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float* rowp = xt + (8 * wave + i) * NODE_LDS_STRIDE;
            const float v0 = rowp[lane], v1 = rowp[lane + 64];
            float sum = v0 + v1;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
            const float mean = sum * (1.0f / C);
            const float d0 = v0 - mean, d1 = v1 - mean;
            float sq = d0 * d0 + d1 * d1;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off, 64);
            const float rstd = 1.0f / sqrtf(sq * (1.0f / C) + 1e-5f);
            rowp[lane] = d0 * rstd * g0 + be0;
            rowp[lane + 64] = d1 * rstd * g1 + be1;
        }
    }
    __syncthreads();

    // ---- D: MLP, wave `wave` owns hidden units [wave*HQ, (wave+1)*HQ) ---------------------------------
    const int h = lane >> 5, j = lane & 31;
    f32x16 acc_o[TC];
    {
        f32x16 bx[TC];
#pragma unroll
        for (int t = 0; t < TC; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(&xt[j * NODE_LDS_STRIDE + 32 * t + 8 * q + 4 * h]);
                bx[t][4 * q] = v[0]; bx[t][4 * q + 1] = v[1]; bx[t][4 * q + 2] = v[2]; bx[t][4 * q + 3] = v[3];
            }
        f32x16 acc_h[THQ];
        arreau_bias_tiles<THQ>(acc_h, mb1 + wave * HQ, h);
        const float* w1 = m1p + (size_t)(wave * THQ) * TC * ARREAU_PACK_TILE_FLOATS;  // out tiles of this quarter
#pragma unroll
        for (int t = 0; t < TC; ++t)
            arreau_gemm_intile<THQ>(acc_h, w1 + (size_t)t * ARREAU_PACK_TILE_FLOATS, TC * ARREAU_PACK_TILE_FLOATS, bx[t], lane);
        arreau_gelu_tiles<THQ>(acc_h);
#pragma unroll
        for (int u = 0; u < TC; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_o[u][r] = 0.f;
        constexpr int TH = H / 32;
        const float* w2 = m2p + (size_t)(wave * THQ) * ARREAU_PACK_TILE_FLOATS;  // in tiles of this quarter
#pragma unroll
        for (int t = 0; t < THQ; ++t)
            arreau_gemm_intile<TC>(acc_o, w2 + (size_t)t * ARREAU_PACK_TILE_FLOATS, TH * ARREAU_PACK_TILE_FLOATS, acc_h[t], lane);
    }
    __syncthreads();  // all waves have read xn; xt and red are free

    // ---- E: (w0 + w2) -> red, (w1 + w3) -> xt, fixed order ------------------------------------------
    {
        float* buf = (wave & 1) ? xt : red;
        if (wave < 2) {
#pragma unroll
            for (int u = 0; u < TC; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v = {acc_o[u][4 * q], acc_o[u][4 * q + 1], acc_o[u][4 * q + 2], acc_o[u][4 * q + 3]};
                    *reinterpret_cast<f32x4*>(&buf[j * NODE_LDS_STRIDE + 32 * u + 8 * q + 4 * h]) = v;
                }
        }
        __syncthreads();
        if (wave >= 2) {
#pragma unroll
            for (int u = 0; u < TC; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4* p = reinterpret_cast<f32x4*>(&buf[j * NODE_LDS_STRIDE + 32 * u + 8 * q + 4 * h]);
                    f32x4 v = *p;
                    v[0] += acc_o[u][4 * q]; v[1] += acc_o[u][4 * q + 1]; v[2] += acc_o[u][4 * q + 2]; v[3] += acc_o[u][4 * q + 3];
                    *p = v;
                }
        }
        __syncthreads();
    }

    // ---- F: bias, layer scale, residual; write x_out; read-out partials ----------------------------
    {
        const int c4 = tid & 31, rr = tid >> 5;
        const f32x4 b2v = *reinterpret_cast<const f32x4*>(mb2 + 4 * c4);
        const f32x4 lsv = *reinterpret_cast<const f32x4*>(ls + 4 * c4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = rr + 8 * i;
            const int n = n0 + (r >> 4), o = r & 15;
            f32x4 v = *reinterpret_cast<const f32x4*>(&red[r * NODE_LDS_STRIDE + 4 * c4]) +
                      *reinterpret_cast<const f32x4*>(&xt[r * NODE_LDS_STRIDE + 4 * c4]);
            f32x4 xo = {0.f, 0.f, 0.f, 0.f};
            if (n < N) {
                const size_t g = ((size_t)n * 16 + o) * C + 4 * c4;
                const f32x4 xi = *reinterpret_cast<const f32x4*>(x_in + g);
                xo = (v + b2v) * lsv + xi;
                *reinterpret_cast<f32x4*>(x_out + g) = xo;
            }
            *reinterpret_cast<f32x4*>(&red[r * NODE_LDS_STRIDE + 4 * c4]) = xo;  // same thread read it above
        }
    }
    __syncthreads();
    {
        // mean over orientations (feeds the scalar / global read-outs, which commute with the mean)
        const int c = tid & 127, n2 = tid >> 7;
        const int n = n0 + n2;
        if (n < N) {
            float sum = 0.f;
#pragma unroll
            for (int o = 0; o < 16; ++o) sum += red[(16 * n2 + o) * NODE_LDS_STRIDE + c];
            xbar[(size_t)n * C + c] = sum * (1.0f / 16.0f);
        }
        // vector read-out channel (column S of read_out_layers): one dot product per (node, orientation)
        const float wv0 = ro_wT[(size_t)lane * (S + 4) + S], wv1 = ro_wT[(size_t)(lane + 64) * (S + 4) + S];
        const float bv = ro_b[S];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = 8 * wave + i;
            const int nn = n0 + (r >> 4), o = r & 15;
            float d = red[r * NODE_LDS_STRIDE + lane] * wv0 + red[r * NODE_LDS_STRIDE + lane + 64] * wv1;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) d += __shfl_xor(d, off, 64);
            if (lane == 0 && nn < N) {
                const size_t g = (size_t)nn * 16 + o;
                vsum[g] = (first_layer ? 0.0f : vsum[g]) + (d + bv);
            }
        }
    }
}

int arreau_launch_node_layer(const arreau_model* m, int layer, const float* kbuf, const int32_t* deg,
                             const int32_t* src, const float* x_in, float* x_out, float* xbar, float* vsum, int N,
                             hipStream_t s) {
    if (N == 0) return ARREAU_OK;
    const int C = m->C, H = m->H, L = m->L, S = m->S;
    (void)L;
    const size_t layer_stride = (size_t)N * m->k * 16 * C;
    const size_t m1_tile = (size_t)(H / 32) * (C / 32) * ARREAU_PACK_TILE_FLOATS;
    const size_t m2_tile = (size_t)(C / 32) * (H / 32) * ARREAU_PACK_TILE_FLOATS;
    if (C == 128 && H == 512) {
        hipLaunchKernelGGL((node_layer_kernel<128, 512>), dim3((N + 1) / 2), dim3(256), 0, s,
                           kbuf + (size_t)layer * layer_stride, deg, src, x_in, x_out,
                           m->fk + (size_t)layer * 16 * 16 * C, m->conv_bias + (size_t)layer * C,
                           m->ln_w + (size_t)layer * C, m->ln_b + (size_t)layer * C, m->m1p + layer * m1_tile,
                           m->mb1 + (size_t)layer * H, m->m2p + layer * m2_tile, m->mb2 + (size_t)layer * C,
                           m->ls + (size_t)layer * C, m->ro_wT + (size_t)layer * C * (S + 4),
                           m->ro_b + (size_t)layer * (S + 4), S, N, m->k, layer == 0 ? 1 : 0,
                           xbar + (size_t)layer * N * C, vsum);
    } else {
        arreau_set_error("node kernel: unsupported (hidden_dim, widening_factor)");
        return ARREAU_EINVAL;
    }
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

__global__ __launch_bounds__(128) void readout_nodes_kernel(
    const float* __restrict__ xbar,   // [L][N][C]
    const float* __restrict__ vsum,   // [N][16]
    const float* __restrict__ ro_wT,  // [L][C][S+4]
    const float* __restrict__ ro_b,   // [L][S+4]
    const float* __restrict__ ori, int S, int C, int L, int N, float* __restrict__ eps,
    float* __restrict__ logits, float* __restrict__ gs /*[N][3]*/) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [RO_ATOMS][L*C]
    const int n0 = blockIdx.x * RO_ATOMS;
    const int RO = S + 4, LC = L * C;
    for (int i = threadIdx.x; i < RO_ATOMS * LC; i += blockDim.x) {
        const int a = i / LC, r = i - a * LC;
        const int l = r / C, c = r - l * C;
        const int n = n0 + a;
        xs[i] = n < N ? xbar[((size_t)l * N + n) * C + c] : 0.f;
    }
    __syncthreads();
    const float invL = 1.0f / (float)L;
    for (int s_out = threadIdx.x; s_out < RO; s_out += blockDim.x) {
        if (s_out == S) {
            // vector channel: eps for the tile's atoms
            for (int a = 0; a < RO_ATOMS; ++a) {
                const size_t n = (size_t)n0 + a;
                if (n >= (size_t)N) break;
                for (int d = 0; d < 3; ++d) {
                    float acc = 0.f;
                    for (int o = 0; o < 16; ++o) acc += (vsum[n * 16 + o] * invL) * ori[3 * o + d];
                    eps[n * 3 + d] = acc * (1.0f / 16.0f);
                }
            }
            continue;
        }
        float tot[RO_ATOMS];
#pragma unroll
        for (int a = 0; a < RO_ATOMS; ++a) tot[a] = 0.f;
        for (int l = 0; l < L; ++l) {
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
            for (int a = 0; a < RO_ATOMS; ++a) tot[a] += acc[a] + bias;
        }
#pragma unroll
        for (int a = 0; a < RO_ATOMS; ++a) {
            const size_t n = (size_t)n0 + a;
            if (n < (size_t)N) {
                const float v = tot[a] * invL;
                if (s_out < S) logits[n * S + s_out] = v;
                else gs[n * 3 + (s_out - S - 1)] = v;
            }
        }
    }
}

__global__ void readout_crystals_kernel(const float* __restrict__ gs, const int32_t* __restrict__ offsets, int B,
                                        float* __restrict__ len0) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 3 * B) return;
    const int b = idx / 3, g = idx - 3 * b;
    float acc = 0.f;
    for (int n = offsets[b]; n < offsets[b + 1]; ++n) acc += gs[(size_t)n * 3 + g];
    len0[idx] = acc;
}

int arreau_launch_readout(const arreau_model* m, const float* xbar, const float* vsum, const int32_t* offsets, int B,
                          int N, float* gs, float* eps, float* logits, float* len0, hipStream_t s) {
    if (B == 0) return ARREAU_OK;
    if (N > 0) {
        const size_t smem = (size_t)RO_ATOMS * m->L * m->C * sizeof(float);
        hipLaunchKernelGGL(readout_nodes_kernel, dim3((N + RO_ATOMS - 1) / RO_ATOMS), dim3(128), smem, s, xbar, vsum,
                           m->ro_wT, m->ro_b, m->ori, m->S, m->C, m->L, N, eps, logits, gs);
        ARREAU_CHECK_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(readout_crystals_kernel, dim3((3 * B + 127) / 128), dim3(128), 0, s, gs, offsets, B, len0);
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}
