// Split-precision (fp16x3) building blocks shared by edge_f16.hip and node_f16.hip.
//
// Every fp32 product a*b is evaluated as three fp16 products on v_mfma_f32_32x32x16_f16 with fp32
// accumulation.  Operands are split into two fp16 planes (11 + 11 significant bits, round to nearest):
//     a = a1 + a2 / 2^11 ,   a1 = f16(a) ,   a2 = f16((a - a1) * 2^11)      (the residual is scaled so that it
//                                                                           stays a normal fp16 number)
//     a*b ~= a1 b1 + (a1 b2 + a2 b1) / 2^11                                  (dropped: a2 b2 / 2^22)
// `main` accumulates a1 b1, `cross` accumulates a1 b2 + a2 b1 and is folded in with one fma per element at the
// end of a tile.  Operand representation error 2^-22 relative (fp32: 2^-24); measured GEMM error equals the
// fp32-MFMA path's because accumulation rounding dominates ("Numerics").  fp16 range: operands must
// stay below 65504 in magnitude (LayerNorm outputs, GELU outputs and the geometric monomials of this model are
// O(1..100)); weights are checked on the host, activations are NOT clamped: one beyond the range becomes an inf
// plane, reaches the outputs as NaN and raises the sticky ARREAU_STATUS_NONFINITE flag (arreau_model_status).
// Half the matrix-pipe work and two thirds of the operand bytes of the bf16x6 scheme (bf16x6.h).
#pragma once
#include <hip/hip_fp16.h>

#include "internal.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define F16X3_SCALE 2048.0f
#define F16X3_INV_SCALE (1.0f / 2048.0f)

// Pair arithmetic of the epilogues.  Two forms of the same arithmetic (bit-identical results):
//   packed (-DARREAU_F32_PACKED): ext-vector pairs -> v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32
//   scalar (default): two independent fp32 instructions per pair.
#ifdef ARREAU_F32_PACKED
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f16x2 cvt_f16x2(f32x2 v) { return __builtin_convertvector(v, f16x2); }
__device__ __forceinline__ f32x2 cvt_f32x2(f16x2 h) { return __builtin_convertvector(h, f32x2); }
#else
struct f32x2 { float x, y; };
__device__ __forceinline__ f32x2 operator*(f32x2 a, f32x2 b) { return f32x2{a.x * b.x, a.y * b.y}; }
__device__ __forceinline__ f32x2 operator-(f32x2 a, f32x2 b) { return f32x2{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ f32x2 operator+(f32x2 a, f32x2 b) { return f32x2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return f32x2{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; }
__device__ __forceinline__ f16x2 cvt_f16x2(f32x2 v) {  // one v_cvt_pk_f16_f32 (round to nearest even)
    typedef float vf2 __attribute__((ext_vector_type(2)));
    return __builtin_convertvector(vf2{v.x, v.y}, f16x2);
}
__device__ __forceinline__ f32x2 cvt_f32x2(f16x2 h) { return f32x2{(float)h[0], (float)h[1]}; }
#endif
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }
#define F16X3_PAIR(v, i) (f32x2{(v)[2 * (i)], (v)[2 * (i) + 1]})

// GELU(x) = x Phi(x) = max(x, 0) - |x| h(|x|),  h(a) = 0.5 erfc(a / sqrt 2) = 2^Q(a)
// with Q a degree-6 polynomial: ONE transcendental and ten instructions per element --
//     a = min(|x|, 12);  Q by Horner (6 fma);  e = exp2(Q);  r = fma(-|x|, e, max(x, 0)).
// (Round 2, last session.  Removing parts of the kernels showed the previous form -- Abramowitz-Stegun 7.1.26: rcp, four
// fma, exp2, five more multiplies / fma, 14 instructions -- to cost 107 us of the 732 us edge kernel and 8 us of each
// 81 us ConvNext launch at 256 x 20, every instruction of it in full: DESIGN.md section 8.)  The coefficients are a
// weighted minimax fit of log2 h on [0, 10] (weight a h(a): what an error of Q does to GELU), tools/exp/gelu_fit.py:
// 5.1e-8 absolute in exact arithmetic, 2.8e-7 maximum / 4.5e-8 rms with fp32 Horner and a 1-ulp exp2 over |x| <= 14 and
// N(0, 1.5) samples (the previous form: 3.3e-7 / 7.4e-8).  Beyond a = 12, h a < 1e-27: the clamp keeps the even-degree
// polynomial from turning upwards.  Non-finite inputs stay loud: x = +-inf -> NaN / -inf, NaN -> NaN (nothing is
// clamped to the fp16 range: a value >= 65520 overflows its fp16 plane to inf, propagates to the network outputs and
// sets ARREAU_STATUS_NONFINITE in the read-out kernel).
// (The timing-only "remove a part" hooks of round 2 -- -DARREAU_EXP=<bits>, wrong results on purpose -- are no longer in the
// product sources: tools/exp/arreau_exp_hooks.patch holds them and tools/exp/build_exp.sh applies it to a scratch copy.)
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
    const float ax = fabsf(x.x), ay = fabsf(x.y);
    // (v_med3 / v_max / v_min co-execute with the partner wave's MFMAs, tools/exp/coexec.hip, coexec2.hip)
    const f32x2 a = f32x2{__builtin_amdgcn_fmed3f(ax, -INFINITY, 12.0f), __builtin_amdgcn_fmed3f(ay, -INFINITY, 12.0f)};
    f32x2 q = fma2(splat2(3.3092788558903113e-05f), a, splat2(-7.6921945996487589e-04f));
    q = fma2(q, a, splat2(8.0807156986535972e-03f));
    q = fma2(q, a, splat2(-5.3412102790454205e-02f));
    q = fma2(q, a, splat2(-4.5877097453124149e-01f));
    q = fma2(q, a, splat2(-1.1512017015627947e+00f));
    q = fma2(q, a, splat2(-9.9999306107072172e-01f));
    f32x2 r;
    r.x = fmaf(-ax, __builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_fmed3f(x.x, 0.0f, INFINITY));
    r.y = fmaf(-ay, __builtin_amdgcn_exp2f(q.y), __builtin_amdgcn_fmed3f(x.y, 0.0f, INFINITY));
    return r;
}
__device__ __forceinline__ float gelu_fast(float x) { return gelu_fast2(splat2(x)).x; }

// ---- fp16 planes of a 32x32 fp32 tile (B-operand form): registers 8s..8s+7 are the fragment of k-step s ------
struct Planes2 { u32x4 p[2][2]; };  // [plane][k-step s]: 8 fp16 per lane each

// Two planes of one register pair: v_cvt_pk_f16_f32 (round to nearest even) for the high plane; the residual plane
// straight from the mixed-precision fma, which reads the fp16 half in place and rounds its fp32 result to fp16:
//     lo = f16(v * 2^11 - 2^11 * f32(h1))     (v * 2^11 is exact, the fma's fp32 result is the exact scaled residual)
// -- one multiply + one v_fma_mix{lo,hi}_f16 per element instead of v_cvt_f32_f16, subtract, multiply and a second
// v_cvt_pk_f16_f32 per pair; the same bits.
template <bool CLAMP = true>
__device__ __forceinline__ void split_pair2(f32x2 v, unsigned& hi, unsigned& lo) {
    // (CLAMP is kept in the signature for the callers; no clamp is applied -- see gelu_fast2: overflow is loud)
    hi = __builtin_bit_cast(unsigned, cvt_f16x2(v));
    const f32x2 sc = v * splat2(F16X3_SCALE);
    const float neg_scale = -F16X3_SCALE;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "s"(neg_scale), "v"(sc.x));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "s"(neg_scale), "v"(sc.y));
}
// ACTIVATION planes (round 5): hi as above, the residual UNSCALED -- lo = f16(v - f32(hi)), one v_fma_mix*_f16 per value and no
// multiply by 2^11 (1.5 vector instructions per value instead of 2.5; every vector instruction of these kernels is paid in
// full beside the matrix work: DESIGN.md section 8).  The residual of |v| >= 1/4 is a normal fp16 number (11 bits: 2^-23 |v|);
// below that it is subnormal, step 2^-24: the pair then represents v to 3e-8 ABSOLUTE, which is what fp32 gives at 1/4 -- hidden
// units, LayerNorm outputs and monomials are consumed by dot products, where an absolute 3e-8 per term is the rounding floor
// (v_mfma_f32_*_f16 honours fp16 subnormals: tools/exp/coexec2.hip).  The product with a weight w = a1 + a2 / 2^11 becomes
//     main += a1 hi + a1 lo ,   cross += a2 hi   (folded by 2^-11 as before; dropped: a2 lo / 2^11)
// -- the same three matrix instructions, two of them on the main accumulator.  The BASIS planes keep the scaled e4m3 / fp16
// residual (split_pair_fp8 / split_pair2): they are stored, streamed and read by the fp8 cross products.
// OWNED: the asm's destination is first written by an instruction the compiler knows (a move), so that hipcc's own hazard handling
// separates it from a matrix instruction that wrote the same register just before -- hipcc counts no wait states for inline asm, and
// in one instantiation (the training forward with the conv inside, node_f16m.hip) its register allocator handed the asm the
// accumulator of a matrix instruction issued five instructions earlier (found by arreau_amd/_isa_lint.py: mfma-result).  One move
// per pair: only that instantiation pays it.
template <bool OWNED = false>
__device__ __forceinline__ void split_pair_act(f32x2 v, unsigned& hi, unsigned& lo) {
    hi = __builtin_bit_cast(unsigned, cvt_f16x2(v));
    const float neg_one = -1.0f;
    if constexpr (OWNED) {
        lo = hi;
        asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "s"(neg_one), "v"(v.x));
    } else
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "s"(neg_one), "v"(v.x));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "s"(neg_one), "v"(v.y));
}
// The planes of a BASIS value pair (round 5): hi as above; the residual goes straight from fp32 to e4m3 --
//     r = v - f32(hi)   (v_fma_mix_f32: reads the fp16 half in place; exact),   lo8 = e4m3(r * 2^11)   (v_cvt_scalef32_pk_fp8_f32,
//     whose scale operand DIVIDES by its power of two: tools/exp/fp8_scale_check.hip)
// -- 1.5 vector instructions per value where multiply + v_fma_mix*_f16 + v_cvt_scalef32_pk_fp8_f16 took 2.5 (the stash encoder
// of the edge kernel: 168 M basis values per step at 256 x 20).  One rounding instead of two (fp32 -> fp16 -> e4m3): every fp16x3
// path forms the basis residual with THIS function, so they still agree bit for bit.  WORD selects the half of `lo8` that
// receives the two bytes.
template <bool WORD>
__device__ __forceinline__ void split_pair_fp8(f32x2 v, unsigned& hi, unsigned& lo8) {
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    hi = __builtin_bit_cast(unsigned, cvt_f16x2(v));
    float r0, r1;
    const float neg_one = -1.0f;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "s"(neg_one), "v"(v.x));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "s"(neg_one), "v"(v.y));
    lo8 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(__builtin_bit_cast(s16x2, lo8), r0, r1, F16X3_INV_SCALE, WORD));
}
// the two e4m3 residuals of word WORD widened to fp16 (exact): the residual plane as the three-fp16-product kernels take it
template <bool WORD>
__device__ __forceinline__ unsigned widen_lo8(unsigned lo8) {
    return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(lo8, 1.0f, WORD));
}

__device__ __forceinline__ Planes2 split_tile2(const f32x16& x) {
    Planes2 r;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            unsigned hi, lo;
            split_pair_act(F16X3_PAIR(x, 4 * s + pp), hi, lo);
            r.p[0][s][pp] = hi;
            r.p[1][s][pp] = lo;
        }
    return r;
}
// Tile epilogue of a hidden layer: planes of GELU(v) * scale for a folded tile v = main + cross / 2^11.
__device__ __forceinline__ Planes2 gelu_split_folded2(const f32x16& v, float scale) {
    Planes2 r;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            unsigned hi, lo;
            split_pair_act(gelu_fast2(F16X3_PAIR(v, 4 * s + pp)) * splat2(scale), hi, lo);  // |scale| <= 1
            r.p[0][s][pp] = hi;
            r.p[1][s][pp] = lo;
        }
    return r;
}
__device__ __forceinline__ f32x16 fold_cross(const f32x16& mainacc, const f32x16& cross);
__device__ __forceinline__ Planes2 gelu_split_tile2(const f32x16& mainacc, const f32x16& cross, float scale) {
    return gelu_split_folded2(fold_cross(mainacc, cross), scale);
}

__device__ __forceinline__ f32x16 mfma_f16(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// One output tile over k-steps [KS0, KS1): main += a1 b1, cross += a1 b2 + a2 b1.  Chunk layout in LDS:
// fragment (ks, plane) at (ks * 2 + plane) * 64 + lane (16 bytes per lane).  The two weight planes of k-step
// ks+1 are read from LDS ahead of the three MFMAs of k-step ks; main and cross are independent chains.
template <int NIN, int KS0, int KS1, int PF = 1>
__device__ __forceinline__ void mma_range2(f32x16& mainacc, f32x16& cross, const u32x4* __restrict__ buf,
                                           const Planes2 (&b)[NIN], int lane) {
    // PF = how many k-steps the LDS fragment reads run ahead of their MFMAs (register ring of PF + 1 pairs)
    const u32x4* f = buf + lane;
    u32x4 c1[PF + 1], c2[PF + 1];
#pragma unroll
    for (int i = 0; i < PF; ++i)
        if (KS0 + i < KS1) {
            c1[i] = f[(size_t)(KS0 + i) * 128];
            c2[i] = f[(size_t)(KS0 + i) * 128 + 64];
        }
#pragma unroll
    for (int ks = KS0; ks < KS1; ++ks) {
        const int slot = (ks - KS0) % (PF + 1), nslot = (ks - KS0 + PF) % (PF + 1);
        if (ks + PF < KS1) {
            c1[nslot] = f[(size_t)(ks + PF) * 128];
            c2[nslot] = f[(size_t)(ks + PF) * 128 + 64];
        }
        const int t = ks >> 1, s = ks & 1;
        mainacc = mfma_f16(c1[slot], b[t].p[0][s], mainacc);
        cross = mfma_f16(c2[slot], b[t].p[0][s], cross);
        mainacc = mfma_f16(c1[slot], b[t].p[1][s], mainacc);  // (activation planes: the residual is unscaled)
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * (PF < KS1 - KS0 ? PF : KS1 - KS0), 0);  // the first PF k-steps' fragments
#pragma unroll
    for (int ks = KS0; ks < KS1; ++ks) {
        if (ks + PF < KS1) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // DS reads of k-step ks+PF ...
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                      // ... ahead of the MFMAs of k-step ks
    }
}

// The same stream cut into pieces (a barrier, a tile store ... between them) WITHOUT restarting the fragment
// prefetch: the reads of the first k-step of the next piece are issued by the piece before it, so their LDS latency
// is covered by MFMAs instead of being exposed after every cut.  NKS = k-steps of the whole chunk.
template <int NIN, int NKS>
struct MmaStream2 {
    const u32x4* f;
    u32x4 c1[2], c2[2];
    __device__ __forceinline__ void start(const u32x4* __restrict__ buf, int lane) {
        f = buf + lane;
        c1[0] = f[0];
        c2[0] = f[64];
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
    template <int KS0, int KS1>
    __device__ __forceinline__ void run(f32x16& mainacc, f32x16& cross, const Planes2 (&b)[NIN]) {
#pragma unroll
        for (int ks = KS0; ks < KS1; ++ks) {
            const int slot = ks & 1, nslot = slot ^ 1;
            if (ks + 1 < NKS) {
                c1[nslot] = f[(size_t)(ks + 1) * 128];
                c2[nslot] = f[(size_t)(ks + 1) * 128 + 64];
            }
            const int t = ks >> 1, s = ks & 1;
            mainacc = mfma_f16(c1[slot], b[t].p[0][s], mainacc);
            cross = mfma_f16(c2[slot], b[t].p[0][s], cross);
            mainacc = mfma_f16(c1[slot], b[t].p[1][s], mainacc);  // (activation planes: the residual is unscaled)
        }
#pragma unroll
        for (int ks = KS0; ks < KS1; ++ks) {
            if (ks + 1 < NKS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // DS reads of k-step ks+1 ...
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                     // ... ahead of the MFMAs of k-step ks
        }
    }
};

// ---- v_mfma_f32_16x16x32_f16 form of the same split product, for a wave's 32 rows as two 16-column blocks nb ------
// Under the chip's power limit this shape holds a higher clock than 32x32x16 (1875 vs 1572 MHz in bare streams:
// tools/exp/mfma_shape.hip) at the same cycles per FLOP, and each weight fragment serves both column blocks.
// Lane (c = lane & 15, g = lane >> 4): B operand = 8 halves of row c of the block (k order 8 g + e), accumulator
// register r = output row 4 g + r of the 16-row tile.
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4v mfma16_f16(const u32x4& a, const u32x4& b, const f32x4v& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
struct Acc16 { f32x4v m[2][2], x[2][2]; };  // [16-row tile mt][column block nb]: main and cross accumulators

// One chunk = 32 output rows x NKB k-blocks of 32: fragments (kb, mt, plane) at ((kb * 2 + mt) * 2 + plane) * 64 + lane.
// Step st = kb * 2 + mt: two fragment reads (both planes), six MFMAs.  The reads of step st+1 are issued ahead of the
// MFMAs of step st and run through cuts of the stream, as in MmaStream2.
template <int NKB, bool ACT = false /* B holds activation planes (unscaled residual): a1 b2 goes to the main accumulator */>
struct MmaStream16 {
    const u32x4* f;
    u32x4 a1[2], a2[2];
    __device__ __forceinline__ void start(const u32x4* __restrict__ buf, int lane) {
        f = buf + lane;
        a1[0] = f[0];
        a2[0] = f[64];
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
    template <int ST0, int ST1>
    __device__ __forceinline__ void run(Acc16& acc, const u32x4 (&b)[2][NKB][2]) {
        run<ST0, ST1>(acc.m, acc.x, b);
    }
    template <int ST0, int ST1>
    __device__ __forceinline__ void run(f32x4v (&am)[2][2], f32x4v (&ax)[2][2], const u32x4 (&b)[2][NKB][2]) {
#pragma unroll
        for (int st = ST0; st < ST1; ++st) {
            const int slot = st & 1, nslot = slot ^ 1;
            if (st + 1 < 2 * NKB) {
                a1[nslot] = f[(size_t)(st + 1) * 128];
                a2[nslot] = f[(size_t)(st + 1) * 128 + 64];
            }
            const int kb = st >> 1, mt = st & 1;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                am[mt][nb] = mfma16_f16(a1[slot], b[nb][kb][0], am[mt][nb]);
                if constexpr (ACT) {
                    ax[mt][nb] = mfma16_f16(a2[slot], b[nb][kb][0], ax[mt][nb]);
                    am[mt][nb] = mfma16_f16(a1[slot], b[nb][kb][1], am[mt][nb]);
                } else {
                    ax[mt][nb] = mfma16_f16(a1[slot], b[nb][kb][1], ax[mt][nb]);
                    ax[mt][nb] = mfma16_f16(a2[slot], b[nb][kb][0], ax[mt][nb]);
                }
            }
        }
#pragma unroll
        for (int st = ST0; st < ST1; ++st) {
            if (st + 1 < 2 * NKB) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
        }
    }
};

// Closed form of the above for register-starved kernels: steps [ST0, ST1) with their own fragment prefetch, nothing
// carried across the call (no fragment registers live over a barrier).  NB = 16-column blocks per wave (2 = the wave's
// 32 rows; 1 = a 16-row tile for batches too small to fill the chip with 32-row tiles -- every output row is computed by
// the same instruction sequence either way, so the two geometries agree bit for bit).
template <int NKB, int ST0, int ST1, int NB = 2>
__device__ __forceinline__ void mma16_range(f32x4v (&am)[2][NB], f32x4v (&ax)[2][NB], const u32x4* __restrict__ buf,
                                            const u32x4 (&b)[NB][NKB][2], int lane) {
    const u32x4* f = buf + lane;
    u32x4 a1[2], a2[2];
    auto frag = [&](u32x4& d, const u32x4& src) { d = src; };
    frag(a1[ST0 & 1], f[(size_t)ST0 * 128]);
    frag(a2[ST0 & 1], f[(size_t)ST0 * 128 + 64]);
#pragma unroll
    for (int st = ST0; st < ST1; ++st) {
        const int slot = st & 1, nslot = slot ^ 1;
        if (st + 1 < ST1) {
            frag(a1[nslot], f[(size_t)(st + 1) * 128]);
            frag(a2[nslot], f[(size_t)(st + 1) * 128 + 64]);
        }
        const int kb = st >> 1, mt = st & 1;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {  // (B = activation planes, unscaled residual: a1 b2 goes to the main accumulator)
            am[mt][nb] = mfma16_f16(a1[slot], b[nb][kb][0], am[mt][nb]);
            ax[mt][nb] = mfma16_f16(a2[slot], b[nb][kb][0], ax[mt][nb]);
            am[mt][nb] = mfma16_f16(a1[slot], b[nb][kb][1], am[mt][nb]);
        }
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
    for (int st = ST0; st < ST1; ++st) {
        if (st + 1 < ST1) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 3 * NB, 0);
    }
}

__device__ __forceinline__ f32x16 fold_cross(const f32x16& mainacc, const f32x16& cross) {
    f32x16 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const f32x2 v = fma2(F16X3_PAIR(cross, i), splat2(F16X3_INV_SCALE), F16X3_PAIR(mainacc, i));
        r[2 * i] = v.x;
        r[2 * i + 1] = v.y;
    }
    return r;
}

// LDS-DMA staging of a weight chunk (global_load_lds_dwordx4: each lane's 16 bytes go straight to
// LDS at wave-uniform base + 16 lane, no VGPR destination, no ds_write).  Wave `wave` of NW moves fragments
// f = NW i + wave; when NF is not a multiple of NW the surplus waves re-copy an earlier fragment (same bytes, same
// place) so that every wave issues the same number of DMA instructions.  hipcc does not count inline-asm memory
// operations: the caller drains them with dma_wait() before the barrier that publishes the chunk, and must keep
// ordinary global LOADS out of the span where a DMA is in flight (hipcc would wait vmcnt(0) for them).
__device__ __forceinline__ void glds16(const u32x4* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int NF, int NW>
__device__ __forceinline__ void dma_chunk(const u32x4* __restrict__ chunk, u32x4* slot, int wave, int lane) {
    constexpr int PER = (NF + NW - 1) / NW;
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)slot);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        int f = NW * i + wave;
        if (NF % NW != 0 && f >= NF) f -= NF % NW;  // wave-uniform; duplicate of a fragment another wave also copies
        glds16(chunk + (size_t)f * 64 + lane, base + (unsigned)f * 1024u);
    }
}
// Lean form for loops: fragment f = NW i + wave of the chunk is addressed as a wave-uniform 64-bit base (SGPR pair:
// chunk + 1024 f, scalar arithmetic) plus ONE per-lane 32-bit offset (16 lane) shared by all fragments, and M0 is
// written directly (it is dead everywhere else in these kernels): scalar work + 1 vector-memory instruction per
// KiB fragment, one VGPR in total.
template <int NF, int NW>
__device__ __forceinline__ void dma_chunk_lean(const void* chunk_uniform, unsigned lane16, int wave,
                                               unsigned slot_plus_wave /* LDS byte address of the slot + 1024 wave */) {
    static_assert(NF % NW == 0, "dma_chunk_lean: whole fragments per wave");
    const char* base = static_cast<const char*>(chunk_uniform) + 1024 * wave;
#pragma unroll
    for (int i = 0; i < NF / NW; ++i)
        asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     :
                     : "v"(lane16), "s"(base + 1024 * NW * i), "s"(slot_plus_wave), "n"(1024 * NW * i)
                     : "memory", "scc", "m0");  // s_add_u32 writes SCC; M0 is overwritten and not restored
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// vmcnt retires in issue order (loads, stores and LDS-DMA alike): all but the N youngest operations are done
// (-DARREAU_DEBUG_WAIT_ALL, the debug build of arreau_amd/build.py, turns every counted wait into vmcnt(0): the
// outputs of the two builds must be bit-identical -- tests/test_gpu_parity.py::test_counted_waits_match_full_waits)
template <int N>
__device__ __forceinline__ void dma_wait_but() {
#ifdef ARREAU_DEBUG_WAIT_ALL
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
