// Device code of the periodic-boundary neighbour list (K1), shared by the stand-alone kernels of graph.hip and by the per-crystal
// tail of the sampling step (tail.hip).  Test infrastructure does not include this file.
#pragma once
#include "internal.h"

__device__ __forceinline__ int arreau_find_crystal(const int32_t* __restrict__ offsets, int B, int i) {
    int lo = 0, hi = B;  // largest b with offsets[b] <= i
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (offsets[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

struct Cand { float dx, dy, dz, d2; };

// offset of periodic image `ci` (0..26, itertools.product((-1,0,1),repeat=3) order) = lattice^T @ cell (:391-393); products
// with -1/0/1 are exact
__device__ __forceinline__ void arreau_image_offset(int ci, const float* Lm, float* o) {
    const float cx = (float)(ci / 9 - 1), cy = (float)((ci / 3) % 3 - 1), cz = (float)(ci % 3 - 1);
    o[0] = __fadd_rn(__fadd_rn(__fmul_rn(cx, Lm[0]), __fmul_rn(cy, Lm[3])), __fmul_rn(cz, Lm[6]));
    o[1] = __fadd_rn(__fadd_rn(__fmul_rn(cx, Lm[1]), __fmul_rn(cy, Lm[4])), __fmul_rn(cz, Lm[7]));
    o[2] = __fadd_rn(__fadd_rn(__fmul_rn(cx, Lm[2]), __fmul_rn(cy, Lm[5])), __fmul_rn(cz, Lm[8]));
}

// candidate c = 27 * sender + image: `pos` holds the senders' positions (of the crystal when first == 0, else of the batch),
// `img` the wave's table of the 27 image offsets (LDS)
// frac_to_cart_coords of one atom (diffusion_helpers.py:223-230), the expression of prep_kernel / arreau_frac_to_cart
__device__ __forceinline__ float arreau_cart_component(const float* __restrict__ frac, const float* Lm, size_t atom, int d) {
    const float f0 = frac[3 * atom], f1 = frac[3 * atom + 1], f2 = frac[3 * atom + 2];
    return (f0 * Lm[d] + f1 * Lm[3 + d]) + f2 * Lm[6 + d];
}

template <bool FROM_FRAC = false>
__device__ __forceinline__ Cand arreau_candidate(const float* __restrict__ pos, int first, int c, const float* img,
                                                 float pix, float piy, float piz, const float* Lm = nullptr) {
    const int j = c / 27;
    const int ci = c - 27 * j;
    float pjv[3];
    if constexpr (FROM_FRAC) {  // `pos` holds fractional coordinates: form the sender's position here
#pragma unroll
        for (int d = 0; d < 3; ++d) pjv[d] = arreau_cart_component(pos, Lm, (size_t)(first + j), d);
    } else {
        const float* pj = pos + 3 * (size_t)(first + j);
        pjv[0] = pj[0]; pjv[1] = pj[1]; pjv[2] = pj[2];
    }
    const float* pj = pjv;
    const float* o = img + 3 * ci;
    Cand r;
    r.dx = __fsub_rn(__fadd_rn(pj[0], o[0]), pix);  // (pos2 + offset) - pos1  (:404-408)
    r.dy = __fsub_rn(__fadd_rn(pj[1], o[1]), piy);
    r.dz = __fsub_rn(__fadd_rn(pj[2], o[2]), piz);
    r.d2 = __fadd_rn(__fadd_rn(__fmul_rn(r.dx, r.dx), __fmul_rn(r.dy, r.dy)), __fmul_rn(r.dz, r.dz));
    return r;
}

// Wave-wide minimum of a double, the same value in every lane.  Round 3: the butterfly of __shfl_xor (two ds_bpermute round
// trips through the LDS crossbar per step, six steps, on the critical path of each of the k selection rounds) became DPP moves
// inside each row of 16 lanes -- quad_perm xor 1, xor 2, row_half_mirror, row_mirror -- then row_bcast15 / row_bcast31 across
// the rows and a read of lane 63.  An exact minimum either way: the selection is unchanged.
__device__ __forceinline__ double arreau_wave_min_f64(double v) {
#define ARREAU_DPP_MIN(ctrl, rows)                                                                              \
    {                                                                                                             \
        const int lo = __double2loint(v), hi = __double2hiint(v);                                                 \
        const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, ctrl, rows, 0xf, false);                              \
        const int hi2 = __builtin_amdgcn_update_dpp(hi, hi, ctrl, rows, 0xf, false);                              \
        v = fmin(v, __hiloint2double(hi2, lo2));                                                                  \
    }
    ARREAU_DPP_MIN(0xB1, 0xf)   // quad_perm [1,0,3,2]
    ARREAU_DPP_MIN(0x4E, 0xf)   // quad_perm [2,3,0,1]
    ARREAU_DPP_MIN(0x141, 0xf)  // row_half_mirror: the other quad of the 8
    ARREAU_DPP_MIN(0x140, 0xf)  // row_mirror: the other half of the row
    ARREAU_DPP_MIN(0x142, 0xa)  // row_bcast15 into rows 1 and 3
    ARREAU_DPP_MIN(0x143, 0xc)  // row_bcast31 into rows 2 and 3: lane 63 now holds the minimum of the wave
#undef ARREAU_DPP_MIN
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

template <bool FROM_FRAC = false /* `cart` holds FRACTIONAL coordinates; positions are formed here (sampling loop: no prep launch) */,
          int NW = 4 /* waves per workgroup: each has its own LDS areas */>
__device__ __forceinline__ void arreau_neighbor_receiver(
    int i /* receiver (wave-uniform) */, int wslot /* this wave's LDS areas: 0 .. NW-1 */, int lane,
    const float* __restrict__ cart, const float* __restrict__ lattice, const int32_t* __restrict__ offsets,
    const int32_t* __restrict__ batch /* [N] crystal of atom, or null */, int B, float r2, int k, int32_t* __restrict__ deg,
    int32_t* __restrict__ src, int32_t* __restrict__ cell, float* __restrict__ dir, float* __restrict__ dist) {
    // the crystal of the receiver: one load when the caller has the atom -> crystal map (prep_kernel writes it),
    // otherwise a binary search over the offsets (log2 B dependent loads)
    const int b = batch ? batch[i] : arreau_find_crystal(offsets, B, i);
    const int first = offsets[b];
    const int ncand = (offsets[b + 1] - first) * 27;
    float Lm[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) Lm[q] = lattice[9 * b + q];
    const float pix = FROM_FRAC ? arreau_cart_component(cart, Lm, (size_t)i, 0) : cart[3 * (size_t)i];
    const float piy = FROM_FRAC ? arreau_cart_component(cart, Lm, (size_t)i, 1) : cart[3 * (size_t)i + 1];
    const float piz = FROM_FRAC ? arreau_cart_component(cart, Lm, (size_t)i, 2) : cart[3 * (size_t)i + 2];
    // Round 3: the wave first copies the Cartesian positions of its crystal into LDS (crystals of up to NBR_LDS_ATOMS atoms;
    // wave-uniform test) and the candidates read them from there.  Before, every candidate evaluation waited for three global
    // loads under its own lane mask, a dozen dependent L2 round trips per receiver.  Same values, same arithmetic.
    // The 27 image offsets are likewise computed once per wave (lanes 0..26) into an LDS table instead of once per candidate:
    // the kernel is bound by the VALU work of the candidate evaluations.
    constexpr int NBR_LDS_ATOMS = 128;
    __shared__ float cpos[NW][3 * NBR_LDS_ATOMS];
    __shared__ float cimg[NW][27 * 3 + 3];
    float* mypos = cpos[wslot];
    float* myimg = cimg[wslot];
    const bool staged = ncand <= 27 * NBR_LDS_ATOMS;
    if (lane < 27) {
        float o[3];
        arreau_image_offset(lane, Lm, o);
        myimg[3 * lane] = o[0]; myimg[3 * lane + 1] = o[1]; myimg[3 * lane + 2] = o[2];
    }
    if (staged)
        for (int a = lane; a < ncand / 9; a += 64)  // ncand / 9 = 3 * atoms
            mypos[a] = FROM_FRAC ? arreau_cart_component(cart, Lm, (size_t)first + a / 3, a % 3) : cart[3 * (size_t)first + a];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    auto candidate = [&](int c) -> Cand {
        return staged ? arreau_candidate(mypos, 0, c, myimg, pix, piy, piz) : arreau_candidate<FROM_FRAC>(cart, first, c, myimg, pix, piy, piz, Lm);
    };

    // Crystals of up to 64 * NBR_KEYS / 27 = 28 atoms (wave-uniform test): every lane evaluates its candidates once
    // and keeps their keys in registers; each of the k selection rounds is then a scan of those keys.  Larger
    // crystals re-evaluate the candidates in every round (same keys, same selection).
    // Selection keys: (bits of d^2, enumeration index c) ordered lexicographically, held as the DOUBLE d2bits * 2^21 + c
    // (exact: d2bits < 2^32, c < 2^21), so that "smallest key above the last one" is a compare + select + v_min_f64 per
    // key and the wave minimum a v_min_f64 butterfly -- no 64-bit integer compares with their chains of scalar lane masks.
    constexpr int NBR_KEYS = 12;
    constexpr double KEY_NONE = 1.0e300;
    auto make_key = [](float d2, int c) { return (double)__float_as_uint(d2) * 2097152.0 + (double)c; };
    bool cached = ncand <= 64 * NBR_KEYS;
    int nkeys = ncand;  // wave-uniform: keys[q] beyond 64 q >= nkeys are KEY_NONE, the rounds skip them
    double keys[NBR_KEYS];
    if (cached) {
#pragma unroll
        for (int q = 0; q < NBR_KEYS; ++q) {
            const int c = lane + 64 * q;
            keys[q] = KEY_NONE;
            if (c < ncand) {
                const Cand cd = candidate(c);
                if (cd.d2 <= r2 && cd.d2 > 0.0001f) keys[q] = make_key(cd.d2, c);
            }
        }
    } else {
        // Larger crystals (round 2, last session: 1.29 ms per step at 1024 x 64 when each of the k selection rounds
        // re-evaluated all 27 n candidates): TWO passes over the candidates instead of k.
        //   pass 1: every lane keeps the smallest key among its candidates; the k-th smallest of these 64 lane minima is
        //           an upper bound T of the k-th smallest key overall (they are k distinct candidates);
        //   pass 2: the candidates with key <= T -- at least k, rarely many more -- are compacted into a per-wave LDS list
        //           (wave ballot + lane prefix count; the order in the list does not matter, the keys are unique) and
        //           become the register-resident key set of the rounds below.
        // Should more than 64 * NBR_LIST candidates pass (massive exact ties), the re-evaluating rounds remain.  Same keys,
        // same selection, same output as before.
        // (the list holds 64 * NBR_LIST keys: T leaves k .. a few dozen of them; a longer list would only cost LDS, i.e. resident
        // waves -- with 64 * NBR_KEYS entries the kernel was limited to 5 workgroups per CU)
        constexpr int NBR_LIST = 6;
        __shared__ double klist[NW][64 * NBR_LIST];
        double* mylist = klist[wslot];
        double lmin = KEY_NONE;
        for (int c = lane; c < ncand; c += 64) {
            const Cand cd = candidate(c);
            if (cd.d2 <= r2 && cd.d2 > 0.0001f) lmin = fmin(lmin, make_key(cd.d2, c));
        }
        double T = KEY_NONE, below = -1.0;
        for (int s = 0; s < k; ++s) {
            const double best = arreau_wave_min_f64(lmin > below ? lmin : KEY_NONE);
            T = best;
            if (best == KEY_NONE) break;  // wave-uniform: fewer than k lanes hold a candidate -> everything in range passes
            below = best;
        }
        int total = 0;  // wave-uniform
        for (int c0 = 0; c0 < ncand; c0 += 64) {
            const int c = c0 + lane;
            bool in = false;
            double key = KEY_NONE;
            if (c < ncand) {
                const Cand cd = candidate(c);
                if (cd.d2 <= r2 && cd.d2 > 0.0001f) {
                    key = make_key(cd.d2, c);
                    in = key <= T;
                }
            }
            const unsigned long long m = __ballot(in);
            const int pos = total + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (in && pos < 64 * NBR_LIST) mylist[pos] = key;
            total += __builtin_popcountll(m);
        }
        if (total <= 64 * NBR_LIST) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own list writes, before its reads
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < NBR_KEYS; ++q) keys[q] = q < NBR_LIST && lane + 64 * q < total ? mylist[lane + 64 * q] : KEY_NONE;
            cached = true;
            nkeys = total;
        }
    }
    double last = -1.0, mine = KEY_NONE;
    int count = 0;
    for (int s = 0; s < k; ++s) {
        double best = KEY_NONE;
        if (cached) {
#pragma unroll
            for (int q = 0; q < NBR_KEYS; ++q)
                if (64 * q < nkeys) best = fmin(best, keys[q] > last ? keys[q] : KEY_NONE);
        } else {
            for (int c = lane; c < ncand; c += 64) {
                const Cand cd = candidate(c);
                if (cd.d2 <= r2 && cd.d2 > 0.0001f) {
                    const double key = make_key(cd.d2, c);
                    best = fmin(best, key > last ? key : KEY_NONE);
                }
            }
        }
        best = arreau_wave_min_f64(best);
        if (best == KEY_NONE) break;  // wave-uniform: fewer than k candidates
        last = best;
        if (lane == s) mine = best;
        ++count;
    }
    // rank of my selection by enumeration index (output order of the reference)
    auto index_of = [](double key) { return (unsigned)((unsigned long long)key & 0x1fffffull); };
    const unsigned myc = index_of(mine);
    int rank = 0;
    for (int s = 0; s < count; ++s) {
        const unsigned oc = (unsigned)__builtin_amdgcn_readlane((int)myc, s);  // s is wave-uniform
        rank += (oc < myc) ? 1 : 0;
    }
    if (lane == 0) deg[i] = count;
    if (lane < k) {
        const size_t base = (size_t)i * k;
        if (lane < count) {
            const Cand cd = candidate((int)myc);
            const size_t o = base + rank;
            src[o] = first + (int)(myc / 27u);
            cell[o] = (int)(myc % 27u);
            dir[3 * o + 0] = cd.dx; dir[3 * o + 1] = cd.dy; dir[3 * o + 2] = cd.dz;
            dist[o] = __fsqrt_rn(cd.d2);
        } else {
            const size_t o = base + lane;  // lanes count..k-1 clear the unused slots
            src[o] = -1; cell[o] = -1;
            dir[3 * o + 0] = 0.f; dir[3 * o + 1] = 0.f; dir[3 * o + 2] = 0.f;
            dist[o] = 0.f;
        }
    }
}


// the form the stand-alone kernels use: workgroup `blk` of four waves, one receiver each
template <bool FROM_FRAC = false>
__device__ __forceinline__ void arreau_neighbor_body(
    unsigned blk /* workgroup of the neighbour part: four receivers */, const float* __restrict__ cart, const float* __restrict__ lattice, const int32_t* __restrict__ offsets,
    const int32_t* __restrict__ batch /* [N] crystal of atom, or null */, int B,
    int n0, int N /* receivers n0 .. N-1 */, float r2, int k, int32_t* __restrict__ deg, int32_t* __restrict__ src,
    int32_t* __restrict__ cell, float* __restrict__ dir, float* __restrict__ dist) {
    const int i = n0 + (int)((blk * (unsigned)blockDim.x + threadIdx.x) >> 6);
    if (i >= N) return;  // wave-uniform
    arreau_neighbor_receiver<FROM_FRAC, 4>(i, (threadIdx.x >> 6) & 3, threadIdx.x & 63, cart, lattice, offsets, batch, B, r2, k, deg, src, cell, dir, dist);
}
