// Model construction: weight folding / MFMA packing (host), upload, one-time fiber-kernel
// evaluation on the GPU.  Replaces PonitaFiberBundle.__init__ + load_state_dict
// (ponita/models/ponita.py:31-86) for the sampling path.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "internal.h"

static thread_local std::string g_last_error;
void arreau_set_error(const std::string& msg) { g_last_error = msg; }
extern "C" const char* arreau_last_error(void) { return g_last_error.c_str(); }
extern "C" const char* arreau_version(void) { return "arreau_hip 0.1 (gfx950)"; }

// P[u][t][q][lane][m] = W[out = 32u + (lane&31)][in = 32t + 8q + 4*(lane>>5) + m]; W is row-major
// [out_dim][in_stride] (torch Linear layout); entries beyond (out_dim, in_dim) are zero.
static void pack_linear(const float* W, int out_dim, int in_dim, int in_stride, int out_pad, int in_pad,
                        float* P) {
    const int U = out_pad / 32, Tn = in_pad / 32;
    for (int u = 0; u < U; ++u)
        for (int t = 0; t < Tn; ++t)
            for (int q = 0; q < 4; ++q)
                for (int lane = 0; lane < 64; ++lane)
                    for (int m = 0; m < 4; ++m) {
                        int out = 32 * u + (lane & 31);
                        int in = 32 * t + 8 * q + 4 * (lane >> 5) + m;
                        float v = (out < out_dim && in < in_dim) ? W[(size_t)out * in_stride + in] : 0.0f;
                        P[((((size_t)u * Tn + t) * 4 + q) * 64 + lane) * 4 + m] = v;
                    }
}

// bf16x3 planes for the split-precision MFMA path.  Truncation split: w = w1 + w2 + w3 EXACTLY, each term
// holding 8 significant bits (bf16), so six bf16 products (a1b1, a1b2, a2b1, a1b3, a2b2, a3b1) reproduce the
// fp32 product to ~3 * 2^-24 relative while running on the 16x faster bf16 matrix pipe.
static inline uint16_t bf16_trunc_bits(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    return (uint16_t)(u >> 16);
}
static inline float bf16_bits_to_float(uint16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float x;
    memcpy(&x, &u, 4);
    return x;
}

// Chunked bf16x3 packing for v_mfma_f32_32x32x16_bf16.  One chunk = one 32-row output tile u:
//   Q[u][t][s][plane][lane][jj] = plane( W[out = 32u + (lane&31)][in = 32t + 16s + 8(jj>>2) + 4(lane>>5) + (jj&3)] )
// (t: 32-wide input tile, s: 16-deep k-step, jj = 0..7).  The k permutation is the one under which the
// previous layer's fp32 accumulator registers 8s..8s+7, converted pairwise to bf16, ARE the B fragment of
// k-step s (cdna_hip_programming.md, "An accumulator tile as the next MFMA's operand").
// 1 KiB per (u, t, s, plane) fragment; a chunk is 6 * (in_pad / 32) KiB.
static void pack_linear_bf16x3(const float* W, int out_dim, int in_dim, int in_stride, int out_pad, int in_pad,
                               uint16_t* Q) {
    const int U = out_pad / 32, Tn = in_pad / 32;
    for (int u = 0; u < U; ++u)
        for (int t = 0; t < Tn; ++t)
            for (int s = 0; s < 2; ++s)
                for (int lane = 0; lane < 64; ++lane)
                    for (int jj = 0; jj < 8; ++jj) {
                        const int out = 32 * u + (lane & 31);
                        const int in = 32 * t + 16 * s + 8 * (jj >> 2) + 4 * (lane >> 5) + (jj & 3);
                        const float w = (out < out_dim && in < in_dim) ? W[(size_t)out * in_stride + in] : 0.0f;
                        const uint16_t b1 = bf16_trunc_bits(w);
                        const float r1 = w - bf16_bits_to_float(b1);
                        const uint16_t b2 = bf16_trunc_bits(r1);
                        const float r2 = r1 - bf16_bits_to_float(b2);
                        const uint16_t b3 = bf16_trunc_bits(r2);
                        const uint16_t pl[3] = {b1, b2, b3};
                        for (int p = 0; p < 3; ++p)
                            Q[((((((size_t)u * Tn + t) * 2 + s) * 3 + p) * 64) + lane) * 8 + jj] = pl[p];
                    }
}

// fp16x3 planes (f16x3.h): w = w1 + w2 / 2^11 with w1 = f16(w), w2 = f16((w - w1) * 2^11), round to nearest even.
static inline uint16_t f32_to_f16_rne(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    const int32_t exp = (int32_t)((u >> 23) & 0xff) - 127 + 15;
    uint32_t man = u & 0x7fffffu;
    if (((u >> 23) & 0xff) == 0xff) return (uint16_t)(sign | 0x7c00u | (man ? 0x200u : 0));  // inf / nan
    if (exp >= 31) return (uint16_t)(sign | 0x7c00u);                                           // overflow -> inf
    if (exp <= 0) {                                                                              // subnormal / zero
        if (exp < -10) return (uint16_t)sign;
        man |= 0x800000u;
        const int shift = 14 - exp;  // 13 + (1 - exp)
        uint32_t half = man >> shift;
        const uint32_t rem = man & ((1u << shift) - 1), mid = 1u << (shift - 1);
        if (rem > mid || (rem == mid && (half & 1))) ++half;
        return (uint16_t)(sign | half);
    }
    uint32_t half = ((uint32_t)exp << 10) | (man >> 13);
    const uint32_t rem = man & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (half & 1))) ++half;  // may carry into the exponent: still correct
    return (uint16_t)(sign | half);
}
static inline float f16_to_f32(uint16_t hbits) {
    const uint32_t sign = (uint32_t)(hbits & 0x8000u) << 16;
    const uint32_t exp = (hbits >> 10) & 0x1f, man = hbits & 0x3ffu;
    uint32_t u;
    if (exp == 0) {
        if (man == 0) u = sign;
        else {
            int e = -1;
            uint32_t m = man;
            do { ++e; m <<= 1; } while (!(m & 0x400u));
            u = sign | ((uint32_t)(127 - 15 - e) << 23) | ((m & 0x3ffu) << 13);
        }
    } else if (exp == 31) u = sign | 0x7f800000u | (man << 13);
    else u = sign | ((exp - 15 + 127) << 23) | (man << 13);
    float x;
    memcpy(&x, &u, 4);
    return x;
}

// Chunked fp16x3 packing for v_mfma_f32_32x32x16_f16; same fragment geometry as pack_linear_bf16x3 with two
// planes:  Q[u][t][s][plane][lane][jj].  1 KiB per (u, t, s, plane) fragment, 4 * (in_pad / 32) KiB per output tile.
// Returns the largest |w| seen (the caller rejects weights that do not fit fp16).
static float pack_linear_f16x3(const float* W, int out_dim, int in_dim, int in_stride, int out_pad, int in_pad,
                               uint16_t* Q) {
    const int U = out_pad / 32, Tn = in_pad / 32;
    float wmax = 0.f;
    for (int u = 0; u < U; ++u)
        for (int t = 0; t < Tn; ++t)
            for (int s = 0; s < 2; ++s)
                for (int lane = 0; lane < 64; ++lane)
                    for (int jj = 0; jj < 8; ++jj) {
                        const int out = 32 * u + (lane & 31);
                        const int in = 32 * t + 16 * s + 8 * (jj >> 2) + 4 * (lane >> 5) + (jj & 3);
                        const float w = (out < out_dim && in < in_dim) ? W[(size_t)out * in_stride + in] : 0.0f;
                        wmax = fabsf(w) > wmax ? fabsf(w) : wmax;
                        const uint16_t h1 = f32_to_f16_rne(w);
                        const uint16_t h2 = f32_to_f16_rne((w - f16_to_f32(h1)) * 2048.0f);
                        const size_t base = ((((size_t)u * Tn + t) * 2 + s) * 2) * 512 + (size_t)lane * 8 + jj;
                        Q[base] = h1;
                        Q[base + 512] = h2;
                    }
    return wmax;
}

// The same weights for v_mfma_f32_16x16x32_f16 (the projection loop of edge_f16.hip): an output tile of 32 rows is two
// 16-row MFMA tiles mt, its K range is walked in 32-column blocks kb.  Fragment (kb, mt, plane) = 1 KiB at index
// (kb * 2 + mt) * 2 + plane of the chunk; lane (m = lane & 15, g = lane >> 4) holds the 8 halves e of output row
// 32 u + 16 mt + m whose input columns are, in the MFMA's k order 8 g + e,
//     32 kb + 16 (g & 1) + 8 (e >> 2) + 4 (g >> 1) + (e & 3)
// -- the order in which the activation planes of f16x3.h (k-step s = g & 1 of lane half h = g >> 1) land in the
// B operand after their relayout.
// `native` selects the k order of a chain that stays in 16x16 tiles (node_f16m.hip): the B operand of the next layer is
// then built in-lane from the accumulators of the two 16-row tiles mt of a 32-row chunk (register r of tile mt = row
// 16 mt + 4 g + r), i.e. half e = 4 mt + r of lane group g is input column 32 kb + 16 (e >> 2) + 4 g + (e & 3).
static float pack_linear_f16x3_m16(const float* W, int out_dim, int in_dim, int in_stride, uint16_t* Q, bool native = false) {
    const int U = out_dim / 32, KB = in_dim / 32;
    float wmax = 0.f;
    for (int u = 0; u < U; ++u)
        for (int kb = 0; kb < KB; ++kb)
            for (int mt = 0; mt < 2; ++mt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int g = lane >> 4;
                        const int out = 32 * u + 16 * mt + (lane & 15);
                        const int in = native ? 32 * kb + 16 * (e >> 2) + 4 * g + (e & 3)
                                              : 32 * kb + 16 * (g & 1) + 8 * (e >> 2) + 4 * (g >> 1) + (e & 3);
                        const float w = W[(size_t)out * in_stride + in];
                        wmax = fabsf(w) > wmax ? fabsf(w) : wmax;
                        const uint16_t h1 = f32_to_f16_rne(w);
                        const uint16_t h2 = f32_to_f16_rne((w - f16_to_f32(h1)) * 2048.0f);
                        const size_t base = (((size_t)u * KB + kb) * 2 + mt) * 2 * 512 + (size_t)lane * 8 + e;
                        Q[base] = h1;
                        Q[base + 512] = h2;
                    }
    return wmax;
}

// OCP fp8 e4m3fn, round to nearest even, saturating at +-448.  (v_cvt_scalef32_pk_fp8_f16 rounds the same way up to 464 and returns
// NaN beyond -- tools/exp/fp8_cvt_check.hip, measured on MI355X in round 5; the weights packed here are checked against 448 first.)
static inline uint8_t f32_to_e4m3_rne_sat(float x) {
    const uint8_t sign = signbit(x) ? 0x80 : 0;
    const float a = fabsf(x);
    if (!(a == a)) return 0x7f;
    if (a >= 448.0f) return sign | 0x7e;
    int e;
    const float f = frexpf(a, &e);  // a = f 2^e, f in [0.5, 1)
    const int E = e - 1;            // a = (2 f) 2^E
    if (a == 0.0f) return sign;
    if (E < -6) return sign | (uint8_t)nearbyintf(ldexpf(a, 9));  // subnormal steps of 2^-9 (8 = the smallest normal: same code)
    int mant = (int)nearbyintf((2.0f * f - 1.0f) * 8.0f), ex = E + 7;
    if (mant == 8) { mant = 0; ++ex; }
    const int code = (ex << 3) | mant;
    return sign | (uint8_t)(code > 0x7e ? 0x7e : code);
}

// Cross-product operands of conv_proj.hip's fp8 form (round 4).  The two cross products of the split scheme sit 2^-11 below
// the main one, so their operands need a handful of bits: both run as ONE v_mfma_scale_f32_16x16x128_f8f6f4 per pair of
// 32-wide k-blocks (twice the fp16 rate),  X = sum_k a1_8 b2_8 + a2_8 b1_8  with
//     a1_8 = e4m3(64 a1),  a2_8 = e4m3(64 a2),  b1_8 = e4m3(b1),  b2_8 = e4m3(b2) (the stash's residual plane as stored),
// i.e. X = 64 (a1 b2 + a2 b1) and K = a1 b1 + X / (64 * 2^11).  (The weights are scaled into the format's normal range; the basis
// is taken as it is -- |b1| up to 448 before the CORRECTION term saturates, values below 2^-9 drop out of it: their share of
// a product 2^-11 below the main one is beneath fp32 rounding.  profiles/r04_cross_precision_study.txt: outputs move by less than
// the fp32 oracle's own distance to fp64.)  Layout, per layer: [16-channel tile T = 0..7][k-block pair kp][plane h][lane] x 16
// bytes: plane 0 = a1_8 of k-blocks 2 kp (bytes 0-7) and 2 kp + 1 (bytes 8-15), plane 1 = a2_8 likewise, of output row
// 16 T + (lane & 15), in the k order of the stashed basis fragments (pack_linear_f16x3_m16, native).  Returns the largest |64 a1|
// (beyond 448 the operand would saturate: the caller then keeps the fp16 cross products).
static float pack_conv_cross_fp8(const float* W /*[C][D]*/, int C, int D, uint8_t* Q) {
    float amax = 0.f;
    for (int T = 0; T < C / 16; ++T)
        for (int kp = 0; kp < D / 64; ++kp)
            for (int h = 0; h < 2; ++h)  // k-block 2 kp + h
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int g = lane >> 4, kb = 2 * kp + h;
                        const int out = 16 * T + (lane & 15);
                        const int in = 32 * kb + 16 * (e >> 2) + 4 * g + (e & 3);
                        const float w = W[(size_t)out * D + in];
                        const float a1 = f16_to_f32(f32_to_f16_rne(w));
                        const float a2 = f16_to_f32(f32_to_f16_rne((w - a1) * 2048.0f));
                        amax = fmaxf(amax, fabsf(64.0f * a1));
                        const size_t plane0 = ((((size_t)T * (D / 64) + kp) * 2 + 0) * 64 + lane) * 16, plane1 = plane0 + 64 * 16;
                        Q[plane0 + 8 * h + e] = f32_to_e4m3_rne_sat(64.0f * a1);
                        Q[plane1 + 8 * h + e] = f32_to_e4m3_rne_sat(64.0f * a2);
                    }
    return amax;
}

// Monomial table: distinct monomials of degree 1..3 in 6 variables in the canonical order
// (i), (i<=j), (i<=j<=k), each lexicographic.  The device code (edge.hip) generates them in the
// same order.
struct Mono { int n; int idx[3]; };
static std::vector<Mono> monomials() {
    std::vector<Mono> v;
    for (int i = 0; i < 6; ++i) v.push_back({1, {i, 0, 0}});
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j) v.push_back({2, {i, j, 0}});
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j)
            for (int k = j; k < 6; ++k) v.push_back({3, {i, j, k}});
    return v;
}

// Fold basis_fn.1.weight [C][258] onto the 83 distinct monomials: the reference's feature vector
// (nn/embedding.py:10-14) holds x_i at column i, x_i x_j at 6 + 6i + j and x_i x_j x_k at
// 42 + 36i + 6j + k, so every permutation of a multiset is a separate column with its own weight;
// the sum of those weights (accumulated in double) multiplies the single monomial.
static void fold_poly_weight(const float* W, int C, std::vector<float>& Wf /*[C][96]*/) {
    auto mons = monomials();
    Wf.assign((size_t)C * ARREAU_MONO_PAD, 0.0f);
    for (int c = 0; c < C; ++c) {
        const float* row = W + (size_t)c * ARREAU_POLY_COLS;
        for (size_t mi = 0; mi < mons.size(); ++mi) {
            const Mono& mo = mons[mi];
            double acc = 0.0;
            if (mo.n == 1) {
                acc = row[mo.idx[0]];
            } else if (mo.n == 2) {
                int i = mo.idx[0], j = mo.idx[1];
                acc = row[6 + 6 * i + j];
                if (i != j) acc += row[6 + 6 * j + i];
            } else {
                int a[3] = {mo.idx[0], mo.idx[1], mo.idx[2]};
                // all distinct permutations of (a0,a1,a2)
                int perms[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
                int seen[6][3];
                int ns = 0;
                for (auto& p : perms) {
                    int t3[3] = {a[p[0]], a[p[1]], a[p[2]]};
                    bool dup = false;
                    for (int s = 0; s < ns; ++s)
                        if (seen[s][0] == t3[0] && seen[s][1] == t3[1] && seen[s][2] == t3[2]) dup = true;
                    if (dup) continue;
                    seen[ns][0] = t3[0]; seen[ns][1] = t3[1]; seen[ns][2] = t3[2];
                    ++ns;
                    acc += row[42 + 36 * t3[0] + 6 * t3[1] + t3[2]];
                }
            }
            Wf[(size_t)c * ARREAU_MONO_PAD + mi] = (float)acc;
        }
    }
}

struct BlobBuilder {
    std::vector<float> data;
    // every region starts 64-float (256 B) aligned so 16-byte vector loads are always aligned
    size_t reserve(size_t n) {
        size_t off = (data.size() + 63) / 64 * 64;
        data.resize(off + n, 0.0f);
        return off;
    }
    size_t put(const float* src, size_t n) {
        size_t off = reserve(n);
        memcpy(data.data() + off, src, n * sizeof(float));
        return off;
    }
};

// ---------------------------------------------------------------------------------------------------------------------------
// Precision safety of the cheap operand formats of the message path (round 5), measured on the model itself like the range
// bounds above are derived from it.  The per-layer kernel projection may read a basis stash whose residual plane is fp8
// (e4m3: 3 bytes per value instead of 4) and run its two cross products on fp8 (e4m3) operands.  With weights like the
// initialisers produce, both leave the outputs at the fp32 rounding floor; with HEAVY-TAILED weights (Student-t kernel / basis
// weights, a few rows fifty times the rest: profiles/r05_basis_q16_study.txt, second half; tests/helpers.py: make_heavy_tailed)
// the fp8 cross products cost 3.3e-6 on the logits -- eight times the exact fp32 kernels' distance to fp64, a third of the 1e-5
// parity target -- although the error of the projected kernels themselves is the same (a host-side calibration of K was written
// first and could not tell the two models apart: what differs is how far the network carries the error).  So the check runs END
// TO END, once, at arreau_model_create: one synthetic batch (16 crystals x 20 atoms, physical cells, t = T / 2, fixed seed)
// through the score network with two fp16 planes + three fp16 products (the round-2 arithmetic, operand error 2^-22), with the
// fp8 residual plane, and with the fp8 residual plane + fp8 cross products.  A format stays selected for this model only if it
// moves eps and the logits by at most ARREAU_CALIB_SHARE (a tenth) of their parity bounds (1e-5 max(1, |eps|), 1e-5 max(1,
// |logits| / 8)); a format that makes the batch non-finite (the hardware's fp8 conversion returns NaN beyond 464) is dropped too.
// The measured shares are reported (arreau_status.basis_fp8_share / cross_fp8_share; -1 = not measured).  ARREAU_BASIS_FP8 /
// ARREAU_CROSS_FP8 in the environment still override, ARREAU_CALIBRATE=0 skips the check (both formats on, the weights' range
// permitting), arreau_model_set_formats changes the selection afterwards.
// ---------------------------------------------------------------------------------------------------------------------------
#define ARREAU_CALIB_SHARE 0.1f
static int calibrate_message_formats(arreau_model* m, hipStream_t s) {
    const char* off = getenv("ARREAU_CALIBRATE");
    if ((off && atoi(off) == 0) || !m->fused || !m->f16_ok || m->edge_variant != 4 || m->conv_variant != 2 || m->k != 8 || m->L < 2 ||
        (m->mlp_variant != 3 && m->mlp_variant != 4))
        return ARREAU_OK;
    const int B = 16, n = 20, N = B * n, S = m->S;
    uint64_t rng = 0x9e3779b97f4a7c15ull;
    auto uni = [&]() {  // xorshift64*: deterministic, no library state
        rng ^= rng >> 12; rng ^= rng << 25; rng ^= rng >> 27;
        return (float)((double)((rng * 0x2545f4914f6cdd1dull) >> 11) / 9007199254740992.0);
    };
    std::vector<float> frac((size_t)N * 3), lengths((size_t)B * 3), angles((size_t)B * 3);
    std::vector<int32_t> types(N), tt(B, m->T / 2), offs(B + 1);
    for (auto& v : frac) v = uni();
    for (auto& v : lengths) v = 4.0f + 4.0f * uni();
    for (auto& v : angles) v = (70.0f + 40.0f * uni()) * 0.017453292519943295f;
    for (int i = 0; i < N; ++i) types[i] = i % 3 == 0 ? S - 1 : (int)(uni() * (float)S) % S;
    for (int b = 0; b <= B; ++b) offs[b] = b * n;
    const size_t ws_bytes = arreau_workspace_bytes(&m->cfg, N, B);
    const size_t f_in = (size_t)N * 3 + 6 * (size_t)B, out1 = (size_t)N * 3 + (size_t)N * S + 3 * (size_t)B;
    const size_t i_in = (size_t)N + 2 * (size_t)B + 1;
    char* dev = nullptr;
    const size_t bytes = ws_bytes + (f_in + 3 * out1 + i_in) * 4 + 4096;
    if (hipMalloc((void**)&dev, bytes) != hipSuccess) { (void)hipGetLastError(); return ARREAU_OK; }  // (no memory to spare: formats as they are)
    float* d_frac = (float*)dev; float* d_len = d_frac + (size_t)N * 3; float* d_ang = d_len + 3 * B;
    float* d_out = d_ang + 3 * B;
    int32_t* d_types = (int32_t*)(d_out + 3 * out1); int32_t* d_t = d_types + N; int32_t* d_off = d_t + B;
    char* d_ws = (char*)(((uintptr_t)(d_off + B + 1) + 255) & ~(uintptr_t)255);
    hipError_t e = hipMemcpyAsync(d_frac, frac.data(), frac.size() * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_len, lengths.data(), lengths.size() * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_ang, angles.data(), angles.size() * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_types, types.data(), types.size() * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_t, tt.data(), tt.size() * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_off, offs.data(), offs.size() * 4, hipMemcpyHostToDevice, s);
    int rc = e == hipSuccess ? ARREAU_OK : ARREAU_EHIP;
    const int x8_weights = m->x8_ok;
    m->calibrating = 1;  // the basis form whatever the launch size; the formats below, whatever the environment says
    const int modes[3][2] = {{0, 0}, {1, 0}, {1, 1}};  // {fp8 residual plane in the stash, fp8 cross products}
    for (int v = 0; v < 3 && rc == ARREAU_OK; ++v) {
        if (modes[v][1] && !x8_weights) break;
        m->fp8_ok = modes[v][0]; m->x8_ok = modes[v][1];
        float* o = d_out + (size_t)v * out1;
        rc = arreau_predict_scores(m, d_frac, d_types, d_len, d_ang, d_t, d_off, B, N, 0, nullptr, nullptr, nullptr, nullptr, o,
                                   o + (size_t)N * 3, o + (size_t)N * 3 + (size_t)N * S, d_ws, ws_bytes, (void*)s);
    }
    m->calibrating = 0;
    m->fp8_ok = 1; m->x8_ok = x8_weights;
    std::vector<float> h(3 * out1);
    int32_t flags = 0;
    if (rc == ARREAU_OK) {
        e = hipMemcpyAsync(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(&flags, m->status, 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemsetAsync(m->status, 0, 4, s);  // (whatever the synthetic batch raised is not the caller's)
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) rc = ARREAU_EHIP;
    }
    (void)hipStreamSynchronize(s);
    (void)hipFree(dev);
    m->ran_edge = m->ran_mlp = m->ran_conv = -1;
    m->ran_x8 = 0;
    if (rc != ARREAU_OK) { arreau_set_error(std::string("format calibration: ") + arreau_last_error()); return rc; }
    (void)flags;
    auto finite = [&](int v) {
        const float* q = h.data() + (size_t)v * out1;
        for (size_t i = 0; i < (size_t)N * 3 + (size_t)N * S; ++i)
            if (!(fabsf(q[i]) < INFINITY)) return false;
        return true;
    };
    if (!finite(0)) return ARREAU_OK;  // the reference arithmetic itself is non-finite on this batch: nothing to measure here
    auto share = [&](int v) {  // how much of the parity bounds variant v uses up, against the two-plane reference
        if (!finite(v)) return 1e9;
        const float* r = h.data();
        const float* q = h.data() + (size_t)v * out1;
        double me = 0.0, ml = 0.0, de = 0.0, dl = 0.0;
        for (size_t i = 0; i < (size_t)N * 3; ++i) { me = fmax(me, fabs((double)r[i])); de = fmax(de, fabs((double)q[i] - r[i])); }
        for (size_t i = (size_t)N * 3; i < (size_t)N * 3 + (size_t)N * S; ++i) { ml = fmax(ml, fabs((double)r[i])); dl = fmax(dl, fabs((double)q[i] - r[i])); }
        return fmax(de / (1e-5 * fmax(1.0, me)), dl / (1e-5 * fmax(1.0, ml / 8.0)));
    };
    m->calib_fp8 = (float)fmin(share(1), 1e9);
    m->fp8_ok = m->calib_fp8 <= ARREAU_CALIB_SHARE ? 1 : 0;
    if (x8_weights) {
        m->calib_x8 = (float)fmin(share(2), 1e9);
        m->x8_ok = m->fp8_ok && m->calib_x8 <= ARREAU_CALIB_SHARE ? 1 : 0;
    }
    if (getenv("ARREAU_VERBOSE_CALIBRATION"))
        fprintf(stderr, "[arreau_hip] message-path formats: share of the parity bounds used up on the calibration batch -- fp8 residual plane %.3f, "
                "+ fp8 cross products %.3f (limit %.2f): basis fp8 %d, fp8 cross %d\n", m->calib_fp8, m->calib_x8, ARREAU_CALIB_SHARE, m->fp8_ok, m->x8_ok);
    return ARREAU_OK;
}

extern "C" int arreau_model_create(const arreau_config* cfg, const arreau_state_dict* sd, void* stream,
                                   arreau_model** out_model) {
    ARREAU_REQUIRE(cfg && sd && out_model, "arreau_model_create: null argument");
    const int S = cfg->num_atomic_states, C = cfg->hidden_dim, D = cfg->basis_dim, L = cfg->num_layers;
    const int O = cfg->num_ori, W = cfg->widening_factor, T = cfg->num_timesteps, k = cfg->max_neighbors;
    const int H = W * C;
    ARREAU_REQUIRE(O == ARREAU_ORI, "unsupported num_ori (this build handles num_ori = 16)");
    ARREAU_REQUIRE(cfg->degree == 3, "unsupported polynomial degree (this build handles degree = 3)");
    // The fused kernels (edge_f16.hip, node*.hip) are instantiated for hidden_dim 128, basis_dim 256, widening 4 -- the
    // shipped checkpoint's shape.  Any other shape (the reference's `make train` preset is hidden_dim = 200) runs on the
    // shape-general fp32 kernels of train_net.hip (exact fp32 MFMA GEMMs + element-wise kernels): same results, no fusion.
    ARREAU_REQUIRE(C >= 4 && C % 4 == 0 && C <= 1024, "hidden_dim must be a multiple of 4 in 4..1024");
    ARREAU_REQUIRE(D >= 4 && D % 4 == 0 && D <= 1024, "basis_dim must be a multiple of 4 in 4..1024");
    ARREAU_REQUIRE(W >= 1 && H <= 1024, "widening_factor * hidden_dim must be at most 1024");
    const bool fused = C == 128 && D == 256 && W == 4;
    ARREAU_REQUIRE(k >= 1 && k <= ARREAU_MAX_K, "max_neighbors must be in 1..8");
    ARREAU_REQUIRE(S >= 2 && S <= 124, "num_atomic_states must be in 2..124");
    ARREAU_REQUIRE(L >= 1 && L <= 8 && T >= 2, "bad num_layers (1..8) / num_timesteps");
    const float* need[] = {sd->basis_w1, sd->basis_b1, sd->basis_w2, sd->basis_b2, sd->fiber_w1, sd->fiber_b1,
                           sd->fiber_w2, sd->fiber_b2, sd->x_embedder_w, sd->conv_kernel_w, sd->conv_fiber_w,
                           sd->conv_bias, sd->norm_w, sd->norm_b, sd->linear1_w, sd->linear1_b, sd->linear2_w,
                           sd->linear2_b, sd->readout_w, sd->readout_b, sd->ori_grid, sd->t_emb_w, sd->ve_sigmas,
                           sd->vp_alpha_bars, sd->vp_betas, sd->q_one_step_transposed, sd->q_mats};
    for (const float* p : need) ARREAU_REQUIRE(p != nullptr, "arreau_model_create: missing state_dict entry");

    // Range safety of the fp16x3 kernels (round 4): every fp16 OPERAND of the split-precision chains is bounded here from the
    // weights alone, over everything the inputs can be --
    //   edge chain: the 83 monomials (|pair invariants|, |cosines| <= 1, dist <= radius: R^(powers of dist)), the hidden units
    //     GELU(W1 mono + b1) <= sum_i |W1f[j,i]| bound_i + |b1_j|, the basis GELU(W2 h + b2) * window <= the same one layer on;
    //   node chain (per layer): the LayerNorm output (|xhat|_2 = sqrt(C)): sqrt(C - 1) |gamma_i| + |beta_i|, and the hidden units
    //     GELU(W1 xn + b1) <= sqrt(C) |W1[j,:] * gamma|_2 + |W1[j,:] . beta| + |b1_j|  (Cauchy-Schwarz).
    // A bound inside the fp16 range proves the kernels cannot overflow.  The L1 / L2 bounds are loose (every sign aligned), so
    // a bound up to 64 x the range keeps the fp16x3 kernels and relies on the sticky NONFINITE flag (an overflow is loud:
    // DiffusionLoss.sample then re-runs the batch on the full-range bf16x6 kernels); beyond that the model starts on bf16x6.
    float edge_bound = 0.f, node_bound = 0.f;
    {
        std::vector<float> w1f_b;
        fold_poly_weight(sd->basis_w1, C, w1f_b);
        auto mons = monomials();
        const double R = cfg->radius > 1.0f ? cfg->radius : 1.0f;
        double mono_max = 0.0, h1_max = 0.0, basis_max = 0.0;
        std::vector<double> mb(mons.size());
        for (size_t mi = 0; mi < mons.size(); ++mi) {
            double b = 1.0;
            for (int q = 0; q < mons[mi].n; ++q) b *= mons[mi].idx[q] == 2 ? R : 1.0;
            mb[mi] = b;
            mono_max = std::max(mono_max, b);
        }
        for (int j = 0; j < C; ++j) {
            double a = fabs((double)sd->basis_b1[j]);
            for (size_t mi = 0; mi < mons.size(); ++mi) a += fabs((double)w1f_b[(size_t)j * ARREAU_MONO_PAD + mi]) * mb[mi];
            h1_max = std::max(h1_max, a);
        }
        for (int j = 0; j < D; ++j) {
            double a = fabs((double)sd->basis_b2[j]);
            for (int i = 0; i < C; ++i) a += fabs((double)sd->basis_w2[(size_t)j * C + i]) * h1_max;
            basis_max = std::max(basis_max, a);
        }
        edge_bound = (float)std::max(mono_max, std::max(h1_max, basis_max));
        double nb = 0.0;
        for (int l = 0; l < L; ++l) {
            const float* g = sd->norm_w + (size_t)l * C;
            const float* be = sd->norm_b + (size_t)l * C;
            for (int i = 0; i < C; ++i) nb = std::max(nb, sqrt((double)C - 1.0) * fabs((double)g[i]) + fabs((double)be[i]));
            for (int j = 0; j < H; ++j) {
                const float* w = sd->linear1_w + ((size_t)l * H + j) * C;
                double n2 = 0.0, dot = 0.0;
                for (int i = 0; i < C; ++i) { n2 += (double)w[i] * g[i] * (double)w[i] * g[i]; dot += (double)w[i] * be[i]; }
                nb = std::max(nb, sqrt((double)C) * sqrt(n2) + fabs(dot) + fabs((double)sd->linear1_b[(size_t)l * H + j]));
            }
        }
        node_bound = (float)nb;
    }

    BlobBuilder bb;
    std::vector<float> tmp;
    const size_t off_ori = bb.put(sd->ori_grid, (size_t)O * 3);

    // Edge-kernel weight stream: w1 (folded, out C, in 96) | w2 (out D, in C) | wk_0..L-1 (out C, in D),
    // contiguous, followed by a pad the prefetch ring may read (never used).
    std::vector<float> w1f;
    fold_poly_weight(sd->basis_w1, C, w1f);
    const size_t n_w1 = (size_t)(C / 32) * (ARREAU_MONO_PAD / 32) * ARREAU_PACK_TILE_FLOATS;
    const size_t n_w2 = (size_t)(D / 32) * (C / 32) * ARREAU_PACK_TILE_FLOATS;
    const size_t wk_tile = (size_t)(C / 32) * (D / 32) * ARREAU_PACK_TILE_FLOATS;
    const size_t stream_pad = 16 * 256;
    const size_t off_w1p = bb.reserve(fused ? n_w1 + n_w2 + wk_tile * L + stream_pad : 0);
    const size_t off_w2p = off_w1p + n_w1;
    const size_t off_wkp = off_w2p + n_w2;
    if (fused) {
        pack_linear(w1f.data(), C, ARREAU_MONO_PAD, ARREAU_MONO_PAD, C, ARREAU_MONO_PAD, bb.data.data() + off_w1p);
        pack_linear(sd->basis_w2, D, C, C, D, C, bb.data.data() + off_w2p);
        for (int l = 0; l < L; ++l)
            pack_linear(sd->conv_kernel_w + (size_t)l * C * D, C, D, D, C, D, bb.data.data() + off_wkp + l * wk_tile);
    }
    const size_t off_b1 = bb.put(sd->basis_b1, C);
    const size_t off_b2 = bb.put(sd->basis_b2, D);
    // the same three matrices as bf16x3 chunks (one chunk per output tile): w1 | w2 | wk_0..L-1
    const size_t h_w1 = (size_t)(C / 32) * (ARREAU_MONO_PAD / 32) * 6 * 512;   // uint16 count
    const size_t h_w2 = (size_t)(D / 32) * (C / 32) * 6 * 512;
    const size_t h_wk = (size_t)(C / 32) * (D / 32) * 6 * 512;
    const size_t off_es16 = bb.reserve(fused ? (h_w1 + h_w2 + h_wk * L) / 2 + 64 : 0);
    if (fused) {
        uint16_t* q = reinterpret_cast<uint16_t*>(bb.data.data() + off_es16);
        pack_linear_bf16x3(w1f.data(), C, ARREAU_MONO_PAD, ARREAU_MONO_PAD, C, ARREAU_MONO_PAD, q);
        pack_linear_bf16x3(sd->basis_w2, D, C, C, D, C, q + h_w1);
        for (int l = 0; l < L; ++l)
            pack_linear_bf16x3(sd->conv_kernel_w + (size_t)l * C * D, C, D, D, C, D, q + h_w1 + h_w2 + l * h_wk);
    }

    // fp16x3 chunks of the same three matrices (edge_f16.hip): 4 KiB per input tile per output tile
    const size_t f_w1 = (size_t)(C / 32) * (ARREAU_MONO_PAD / 32) * 4 * 512;   // uint16 count
    const size_t f_w2 = (size_t)(D / 32) * (C / 32) * 4 * 512;
    const size_t f_wk = (size_t)(C / 32) * (D / 32) * 4 * 512;
    const size_t off_ef16 = bb.reserve(fused ? (f_w1 + f_w2 + f_wk * L) / 2 + 64 : 0);
    float wmax16 = 0.f;
    if (fused) {
        uint16_t* q = reinterpret_cast<uint16_t*>(bb.data.data() + off_ef16);
        wmax16 = fmaxf(wmax16, pack_linear_f16x3(w1f.data(), C, ARREAU_MONO_PAD, ARREAU_MONO_PAD, C, ARREAU_MONO_PAD, q));
        wmax16 = fmaxf(wmax16, pack_linear_f16x3_m16(sd->basis_w2, D, C, C, q + f_w1));  // K = re-laid layer-1 tiles
        for (int l = 0; l < L; ++l)
            wmax16 = fmaxf(wmax16, pack_linear_f16x3_m16(sd->conv_kernel_w + (size_t)l * C * D, C, D, D,
                                                         q + f_w1 + f_w2 + l * f_wk, true));  // K = basis, native order
    }
    // fp8 cross-product operands of the per-layer projection (conv_proj.hip): 64 KiB per layer
    const size_t x8_layer_floats = (size_t)(C / 16) * (D / 64) * 2 * 64 * 16 / 4;
    const size_t off_x8 = bb.reserve(fused ? x8_layer_floats * L : 0);
    float x8_amax = 0.f;
    if (fused)
        for (int l = 0; l < L; ++l)
            x8_amax = fmaxf(x8_amax, pack_conv_cross_fp8(sd->conv_kernel_w + (size_t)l * C * D, C, D,
                                                         reinterpret_cast<uint8_t*>(bb.data.data() + off_x8 + l * x8_layer_floats)));
    const size_t off_fk = bb.reserve((size_t)L * O * O * C);
    const size_t off_conv_bias = bb.put(sd->conv_bias, (size_t)L * C);
    const size_t off_ln_w = bb.put(sd->norm_w, (size_t)L * C);
    const size_t off_ln_b = bb.put(sd->norm_b, (size_t)L * C);

    // ConvNext MLP stream: per layer and per hidden quarter w, linear_1 rows [w*H/4, (w+1)*H/4) (out H/4, in C)
    // followed by linear_2 columns of the same quarter (out C, in H/4); + ring pad at the very end.
    const int HQ = H / 4;
    const size_t q1 = (size_t)(HQ / 32) * (C / 32) * ARREAU_PACK_TILE_FLOATS;
    const size_t q2 = (size_t)(C / 32) * (HQ / 32) * ARREAU_PACK_TILE_FLOATS;
    const size_t off_mlp = bb.reserve(fused ? (q1 + q2) * 4 * L + stream_pad : 0);
    for (int l = 0; l < L && fused; ++l)
        for (int w = 0; w < 4; ++w) {
            float* dst = bb.data.data() + off_mlp + ((size_t)l * 4 + w) * (q1 + q2);
            pack_linear(sd->linear1_w + ((size_t)l * H + (size_t)w * HQ) * C, HQ, C, C, HQ, C, dst);
            pack_linear(sd->linear2_w + (size_t)l * C * H + (size_t)w * HQ, C, HQ, H, C, HQ, dst + q1);
        }
    // the same per-quarter layout as bf16x3 chunks (one 24 KiB block per output tile; the kernel takes two per chunk)
    const size_t hq1 = (size_t)(HQ / 32) * (C / 32) * 6 * 512;   // uint16 count of W1 quarter
    const size_t hq2 = (size_t)(C / 32) * (HQ / 32) * 6 * 512;   // uint16 count of W2 quarter
    const size_t off_mlp16 = bb.reserve(fused ? (hq1 + hq2) * 4 * L / 2 + 64 : 0);
    for (int l = 0; l < L && fused; ++l)
        for (int w = 0; w < 4; ++w) {
            uint16_t* dst = reinterpret_cast<uint16_t*>(bb.data.data() + off_mlp16) + ((size_t)l * 4 + w) * (hq1 + hq2);
            pack_linear_bf16x3(sd->linear1_w + ((size_t)l * H + (size_t)w * HQ) * C, HQ, C, C, HQ, C, dst);
            pack_linear_bf16x3(sd->linear2_w + (size_t)l * C * H + (size_t)w * HQ, C, HQ, H, C, HQ, dst + hq1);
        }
    // ... and as fp16x3 chunks (16 KiB per output tile)
    const size_t fq1 = (size_t)(HQ / 32) * (C / 32) * 4 * 512;   // uint16 count of W1 quarter
    const size_t fq2 = (size_t)(C / 32) * (HQ / 32) * 4 * 512;   // uint16 count of W2 quarter
    const size_t off_mlpf16 = bb.reserve(fused ? (fq1 + fq2) * 4 * L / 2 + 64 : 0);
    for (int l = 0; l < L && fused; ++l)
        for (int w = 0; w < 4; ++w) {
            uint16_t* dst = reinterpret_cast<uint16_t*>(bb.data.data() + off_mlpf16) + ((size_t)l * 4 + w) * (fq1 + fq2);
            wmax16 = fmaxf(wmax16, pack_linear_f16x3(sd->linear1_w + ((size_t)l * H + (size_t)w * HQ) * C, HQ, C, C, HQ, C, dst));
            wmax16 = fmaxf(wmax16, pack_linear_f16x3(sd->linear2_w + (size_t)l * C * H + (size_t)w * HQ, C, HQ, H, C, HQ, dst + fq1));
        }
    // ... and as fp16x3 chunks for the 16x16x32 kernel (native k order), same sizes
    const size_t off_mlpf16m = bb.reserve(fused ? (fq1 + fq2) * 4 * L / 2 + 64 : 0);
    for (int l = 0; l < L && fused; ++l)
        for (int w = 0; w < 4; ++w) {
            uint16_t* dst = reinterpret_cast<uint16_t*>(bb.data.data() + off_mlpf16m) + ((size_t)l * 4 + w) * (fq1 + fq2);
            pack_linear_f16x3_m16(sd->linear1_w + ((size_t)l * H + (size_t)w * HQ) * C, HQ, C, C, dst, true);
            pack_linear_f16x3_m16(sd->linear2_w + (size_t)l * C * H + (size_t)w * HQ, C, HQ, H, dst + fq1, true);
        }
    const size_t off_mb1 = bb.put(sd->linear1_b, (size_t)L * H);
    const size_t off_mb2 = bb.put(sd->linear2_b, (size_t)L * C);

    tmp.assign((size_t)L * C, 1.0f);
    if (cfg->has_layer_scale && sd->layer_scale) memcpy(tmp.data(), sd->layer_scale, tmp.size() * sizeof(float));
    const size_t off_ls = bb.put(tmp.data(), tmp.size());

    const int IN = S + 78;
    tmp.assign((size_t)IN * C, 0.f);
    for (int c = 0; c < C; ++c)
        for (int i = 0; i < IN; ++i) tmp[(size_t)i * C + c] = sd->x_embedder_w[(size_t)c * IN + i];
    const size_t off_embT = bb.put(tmp.data(), tmp.size());

    const int RO = S + 4;
    tmp.assign((size_t)L * C * RO, 0.f);
    for (int l = 0; l < L; ++l)
        for (int s = 0; s < RO; ++s)
            for (int c = 0; c < C; ++c)
                tmp[((size_t)l * C + c) * RO + s] = sd->readout_w[((size_t)l * RO + s) * C + c];
    const size_t off_ro_wT = bb.put(tmp.data(), tmp.size());
    const size_t off_ro_b = bb.put(sd->readout_b, (size_t)L * RO);
    // the same weights as MFMA fragment streams (readout_mfma_kernel): per layer RO_PAD/32 output tiles x C/32 input
    // tiles of 1024 floats, + ARREAU_PF groups of slack behind the last layer for the prefetch ring
    const int RO_PAD = RO <= 96 ? 96 : ((RO + 31) / 32) * 32;  // the kernel is instantiated for three output tiles
    const size_t ro_layer = (size_t)(RO_PAD / 32) * (C / 32) * ARREAU_PACK_TILE_FLOATS;
    tmp.assign(fused ? (size_t)L * ro_layer + (size_t)ARREAU_PF * 256 : 0, 0.f);
    for (int l = 0; l < L && fused; ++l) pack_linear(sd->readout_w + (size_t)l * RO * C, RO, C, C, RO_PAD, C, tmp.data() + (size_t)l * ro_layer);
    const size_t off_ro_pack = bb.put(tmp.data(), tmp.size());
    tmp.assign((size_t)L * C, 0.f);
    for (int l = 0; l < L; ++l)
        for (int c = 0; c < C; ++c) tmp[(size_t)l * C + c] = sd->readout_w[((size_t)l * RO + S) * C + c];
    const size_t off_ro_wv = bb.put(tmp.data(), tmp.size());

    const size_t off_temb = bb.put(sd->t_emb_w, ARREAU_T_EMB_DIM / 2);
    const size_t off_ve = bb.put(sd->ve_sigmas, (size_t)T + 1);
    const size_t off_ab = bb.put(sd->vp_alpha_bars, (size_t)T + 1);
    const size_t off_be = bb.put(sd->vp_betas, (size_t)T + 1);
    const size_t off_q1t = bb.put(sd->q_one_step_transposed, (size_t)T * S * S);
    const size_t off_qm = bb.put(sd->q_mats, (size_t)T * S * S);

    const size_t off_fw1 = bb.put(sd->fiber_w1, (size_t)C * 3);
    const size_t off_fb1 = bb.put(sd->fiber_b1, C);
    const size_t off_fw2 = bb.put(sd->fiber_w2, (size_t)D * C);
    const size_t off_fb2 = bb.put(sd->fiber_b2, D);
    const size_t off_fwk = bb.put(sd->conv_fiber_w, (size_t)L * C * D);
    // plain row-major copies for the training path (train_net.hip)
    const size_t off_t_w1f = bb.put(w1f.data(), (size_t)C * ARREAU_MONO_PAD);
    const size_t off_t_w2 = bb.put(sd->basis_w2, (size_t)D * C);
    const size_t off_t_wk = bb.put(sd->conv_kernel_w, (size_t)L * C * D);
    const size_t off_t_lin1 = bb.put(sd->linear1_w, (size_t)L * H * C);
    const size_t off_t_lin2 = bb.put(sd->linear2_w, (size_t)L * C * H);
    // (+ four zero rows behind the last layer: the training step's d(x) product reads the read-out weights with the reduction padded to a
    // multiple of four -- its d(rbar) operand carries zero pad columns -- so layer L - 1 reads up to three rows past its own)
    const size_t off_t_ro_w = bb.reserve((size_t)L * RO * C + (size_t)4 * C);
    memcpy(bb.data.data() + off_t_ro_w, sd->readout_w, (size_t)L * RO * C * sizeof(float));
    const size_t off_status = bb.reserve(64);  // zero-initialised sticky status word (+ pad)

    arreau_model* m = new arreau_model();
    memset(m, 0, sizeof(*m));
    m->cfg = *cfg;
    m->S = S; m->C = C; m->D = D; m->L = L; m->O = O; m->W = W; m->H = H; m->k = k; m->T = T;
    m->blob_floats = bb.data.size();
    hipError_t e = hipMalloc((void**)&m->blob, m->blob_floats * sizeof(float));
    if (e != hipSuccess) {
        delete m;
        arreau_set_error(std::string("hipMalloc(model blob): ") + hipGetErrorString(e));
        return ARREAU_EHIP;
    }
    hipStream_t s = (hipStream_t)stream;
    e = hipMemcpyAsync(m->blob, bb.data.data(), m->blob_floats * sizeof(float), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);  // host staging buffer dies at scope exit
    if (e != hipSuccess) {
        (void)hipFree(m->blob);
        delete m;
        arreau_set_error(std::string("model upload: ") + hipGetErrorString(e));
        return ARREAU_EHIP;
    }
    float* b = m->blob;
    m->ori = b + off_ori; m->w1p = b + off_w1p; m->b1 = b + off_b1; m->w2p = b + off_w2p; m->b2 = b + off_b2;
    m->wkp = b + off_wkp; m->edge_bf16 = b + off_es16; m->edge_f16 = b + off_ef16; m->f16_ok = fused && wmax16 < 60000.0f ? 1 : 0; m->conv_x8 = b + off_x8; m->x8_ok = m->x8_weights_ok = fused && x8_amax <= 448.0f ? 1 : 0; m->fp8_ok = fused ? 1 : 0; m->calib_fp8 = m->calib_x8 = -1.0f; m->fk = b + off_fk; m->conv_bias = b + off_conv_bias; m->ln_w = b + off_ln_w;
    m->ln_b = b + off_ln_b; m->mlp = b + off_mlp; m->mlp_bf16 = b + off_mlp16; m->mlp_f16 = b + off_mlpf16; m->mlp_f16m = b + off_mlpf16m; m->mb1 = b + off_mb1; m->mb2 = b + off_mb2;
    m->ls = b + off_ls; m->embT = b + off_embT; m->ro_wT = b + off_ro_wT; m->ro_b = b + off_ro_b; m->ro_pack = b + off_ro_pack; m->ro_wv = b + off_ro_wv;
    for (int l = 0; l < L; ++l) m->ro_bv_host[l] = sd->readout_b[(size_t)l * RO + S];
    m->t_emb_w = b + off_temb; m->ve_sigmas = b + off_ve; m->vp_alpha_bars = b + off_ab; m->vp_betas = b + off_be;
    m->q1t = b + off_q1t; m->qmats = b + off_qm;
    {   // structure of the cumulative D3PM matrices (d3pm.py:33-54, forward_type "mask"): nonzeros only on the diagonal and in
        // the last column?  (A "uniform" chain, or any other buffer a checkpoint may hold, keeps the dense path.)
        int absorbing = 1;
        const float* q = sd->q_mats;
        for (size_t t = 0; t < (size_t)T && absorbing; ++t)
            for (int c = 0; c < S && absorbing; ++c)
                for (int s2 = 0; s2 < S - 1; ++s2)
                    if (s2 != c && q[(t * S + c) * S + s2] != 0.0f) { absorbing = 0; break; }
        if (getenv("ARREAU_D3PM_DENSE")) absorbing = 0;  // force the general path (tests)
        m->qmats_absorbing = absorbing;
    }
    m->fiber_w1 = b + off_fw1; m->fiber_b1 = b + off_fb1; m->fiber_w2 = b + off_fw2; m->fiber_b2 = b + off_fb2;
    m->fiber_wk = b + off_fwk;
    m->status = reinterpret_cast<int32_t*>(b + off_status);
    m->t_w1f = b + off_t_w1f; m->t_w2 = b + off_t_w2; m->t_wk = b + off_t_wk; m->t_lin1 = b + off_t_lin1;
    m->t_lin2 = b + off_t_lin2; m->t_ro_w = b + off_t_ro_w;
    auto env_int = [](const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; };
    m->edge_act_bound = edge_bound;
    m->node_act_bound = node_bound;
    // (see the bounds above: provably in range, or within the loose bound's slack and guarded by the NONFINITE flag -> fp16x3;
    // far outside -> the full-range bf16x6 kernels from the start.  The environment and arreau_model_set_variant override.)
    const float f16_slack = 64.0f * 65504.0f;
    m->edge_variant = env_int("ARREAU_EDGE_VARIANT", edge_bound <= f16_slack ? 4 : 3);
    m->mlp_variant = env_int("ARREAU_MLP_VARIANT", node_bound <= f16_slack ? 3 : 1);
    m->conv_variant = env_int("ARREAU_CONV_VARIANT", 2);  // 2: basis form + conv_proj.hip; 1: K stash + streamed conv; 0: register conv
    m->readout_variant = env_int("ARREAU_READOUT_VARIANT", 1);
    m->ran_edge = m->ran_mlp = m->ran_conv = -1;
    m->ran_x8 = 0;
    m->fused = fused ? 1 : 0;
    // ARREAU_GENERAL_PATH=1 (or edge variant 5): run the shape-general fp32 kernels also for the fused shape (cross-check)
    if (!fused || getenv("ARREAU_GENERAL_PATH")) m->edge_variant = ARREAU_VARIANT_GENERAL;

    int rc = arreau_launch_fiber_precompute(m, s);
    if (rc == ARREAU_OK) {
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) {
            arreau_set_error(std::string("fiber precompute: ") + hipGetErrorString(e));
            rc = ARREAU_EHIP;
        }
    }
    if (rc == ARREAU_OK) rc = calibrate_message_formats(m, s);
    if (rc != ARREAU_OK) {
        if (m->train) arreau_train_ctx_destroy(m->train);
        (void)hipFree(m->blob);
        delete m;
        return rc;
    }
    *out_model = m;
    return ARREAU_OK;
}

void arreau_model_retire_graph(arreau_model* m, void* exec, void* stream) {
    if (m->retired_graph) {
        (void)hipStreamSynchronize((hipStream_t)m->retired_stream);  // its launches have long finished; make it certain
        (void)hipGraphExecDestroy((hipGraphExec_t)m->retired_graph);
    }
    m->retired_graph = exec;
    m->retired_stream = stream;
}

extern "C" void arreau_model_destroy(arreau_model* model) {
    if (!model) return;
    arreau_model_retire_graph(model, nullptr, nullptr);
    arreau_partition_destroy(model->part);
    if (model->train) {
        (void)hipDeviceSynchronize();
        arreau_train_ctx_destroy(model->train);
    }
    if (model->loop_stream) {
        (void)hipStreamSynchronize((hipStream_t)model->loop_stream);
        (void)hipStreamDestroy((hipStream_t)model->loop_stream);
        (void)hipEventDestroy((hipEvent_t)model->loop_event);
    }
    if (model->blob) (void)hipFree(model->blob);
    delete model;
}

extern "C" int arreau_model_config(const arreau_model* model, arreau_config* out_cfg) {
    ARREAU_REQUIRE(model && out_cfg, "arreau_model_config: null argument");
    *out_cfg = model->cfg;
    return ARREAU_OK;
}

extern "C" int arreau_model_set_variant(arreau_model* model, int32_t edge_variant, int32_t mlp_variant) {
    ARREAU_REQUIRE(model, "arreau_model_set_variant: null model");
    ARREAU_REQUIRE(edge_variant >= -1 && edge_variant <= ARREAU_VARIANT_GENERAL && mlp_variant >= -1 && mlp_variant <= 4,
                   "arreau_model_set_variant: edge variant must be in 0..5, mlp variant in 0..4 (-1 keeps)");
    ARREAU_REQUIRE(model->fused || edge_variant < 0 || edge_variant == ARREAU_VARIANT_GENERAL,
                   "arreau_model_set_variant: this model's shape (hidden_dim, basis_dim, widening_factor) has no fused kernels; "
                   "only the general path (edge variant 5) is available");
    if (edge_variant >= 0) model->edge_variant = edge_variant;
    // (the training forward follows the ConvNext choice: bf16x6 = full-range products, fp16x3 = the default: train_net.hip)
    if (mlp_variant == 1) model->train_full_range = 1;
    else if (mlp_variant == 3) model->train_full_range = 0;
    if (mlp_variant >= 0) model->mlp_variant = mlp_variant;
    return ARREAU_OK;
}

extern "C" int arreau_model_set_formats(arreau_model* model, int32_t basis_fp8, int32_t cross_fp8) {
    ARREAU_REQUIRE(model, "arreau_model_set_formats: null model");
    if (basis_fp8 >= 0) model->fp8_ok = basis_fp8 != 0 && model->fused ? 1 : 0;
    if (cross_fp8 >= 0) model->x8_ok = cross_fp8 != 0 && model->x8_weights_ok ? 1 : 0;
    if (!model->fp8_ok) model->x8_ok = 0;  // the fp8 cross products read the stash's e4m3 residual plane as stored
    return ARREAU_OK;
}

extern "C" int arreau_model_status(const arreau_model* model, arreau_status* out, int32_t reset, void* stream) {
    ARREAU_REQUIRE(model && out, "arreau_model_status: null argument");
    hipStream_t s = (hipStream_t)stream;
    int32_t flags = 0;
    ARREAU_CHECK_HIP(hipMemcpyAsync(&flags, model->status, sizeof(flags), hipMemcpyDeviceToHost, s));
    if (reset) ARREAU_CHECK_HIP(hipMemsetAsync(model->status, 0, sizeof(int32_t), s));
    ARREAU_CHECK_HIP(hipStreamSynchronize(s));
    out->flags = flags;
    out->edge_kernel = model->ran_edge;
    out->mlp_kernel = model->ran_mlp;
    out->conv_kernel = model->ran_conv;
    out->basis_row_bytes = model->ran_conv == 2 ? (arreau_basis_fp8(model) ? 768 : 1024) : 0;
    out->conv_cross_fp8 = model->ran_conv == 2 ? model->ran_x8 : 0;
    out->edge_activation_bound = model->edge_act_bound;
    out->node_activation_bound = model->node_act_bound;
    out->basis_fp8_share = model->calib_fp8;
    out->cross_fp8_share = model->calib_x8;
    return ARREAU_OK;
}

// ---------------------------------------------------------------------------------------------
// One-time fiber kernels.  fiber_attr[o][p] = ori_o . ori_p (geometry/invariants.py:24);
// fiber_kernel_basis = fiber_basis_fn(fiber_attr) (ponita.py:66,95: PolynomialFeatures(3) ->
// Linear(3->C) -> GELU -> Linear(C->D) -> GELU); per layer fiber_kernel = Linear_{D->C}
// (conv.py:113) and the spherical conv divides by O (conv.py:115) -- folded in here.
// One workgroup per (o,p); runs once per model load, so it is written for clarity, not speed.
// ---------------------------------------------------------------------------------------------
__global__ void fiber_precompute_kernel(const float* __restrict__ ori, const float* __restrict__ w1,
                                        const float* __restrict__ b1, const float* __restrict__ w2,
                                        const float* __restrict__ b2, const float* __restrict__ wk,
                                        float* __restrict__ fk, int C, int D, int L, int O) {
    extern __shared__ float sm[];  // h1[C] then basis[D]
    float* h1 = sm;
    float* basis = sm + C;
    const int o = blockIdx.x / O, p = blockIdx.x % O;
    const float a = ori[o * 3 + 0] * ori[p * 3 + 0] + ori[o * 3 + 1] * ori[p * 3 + 1] + ori[o * 3 + 2] * ori[p * 3 + 2];
    const float poly[3] = {a, a * a, (a * a) * a};
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float v = b1[c];
        for (int i = 0; i < 3; ++i) v += w1[c * 3 + i] * poly[i];
        h1[c] = arreau_gelu(v);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float v = b2[d];
        for (int c = 0; c < C; ++c) v += w2[(size_t)d * C + c] * h1[c];
        basis[d] = arreau_gelu(v);
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < L * C; idx += blockDim.x) {
        const int l = idx / C, c = idx % C;
        const float* wrow = wk + ((size_t)l * C + c) * D;
        float v = 0.f;
        for (int d = 0; d < D; ++d) v += wrow[d] * basis[d];
        fk[(((size_t)l * O + o) * O + p) * C + c] = v / (float)O;
    }
}

int arreau_launch_fiber_precompute(arreau_model* m, hipStream_t s) {
    const size_t smem = (size_t)(m->C + m->D) * sizeof(float);
    hipLaunchKernelGGL(fiber_precompute_kernel, dim3(m->O * m->O), dim3(256), smem, s, m->ori, m->fiber_w1,
                       m->fiber_b1, m->fiber_w2, m->fiber_b2, m->fiber_wk, m->fk, m->C, m->D, m->L, m->O);
    ARREAU_CHECK_HIP(hipGetLastError());
    return ARREAU_OK;
}
