// What does v_cvt_scalef32_pk_fp8_f16 (scale 1.0) do on gfx950?  Every fp16 bit pattern through the instruction and back
// (v_cvt_scalef32_pk_f16_fp8), compared on the host with OCP e4m3fn round-to-nearest-even with saturation to +-448.
//   hipcc --offload-arch=gfx950 -O2 tools/exp/fp8_cvt_check.hip -o tools/exp/_bin/fp8_cvt_check && tools/exp/_bin/fp8_cvt_check
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned short* out8, unsigned short* back16) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;  // fp16 pattern
    unsigned short bits = (unsigned short)i;
    _Float16 h = __builtin_bit_cast(_Float16, bits);
    h2 a = {h, h};
    s2 r = {0, 0};
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, a, 1.0f, false);
    unsigned u = __builtin_bit_cast(unsigned, r);
    out8[i] = (unsigned short)(u & 0xff);
    h2 b = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(u, 1.0f, false);
    back16[i] = __builtin_bit_cast(unsigned short, b[0]);
}
static float e4m3_to_float(unsigned v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x;
    if (e == 15 && m == 7) return NAN;
    if (e == 0) x = ldexpf((float)m, -9);
    else x = ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -x : x;
}
static unsigned float_to_e4m3_rne_sat(float f) {
    if (isnan(f)) return 0x7f;
    const unsigned s = signbit(f) ? 0x80 : 0;
    float a = fabsf(f);
    if (a >= 448.0f) return s | 0x7e;  // saturate (also inf)
    unsigned best = 0;
    float bd = 1e30f;
    for (unsigned v = 0; v < 0x7f; ++v) {
        const float d = fabsf(e4m3_to_float(v) - a);
        if (d < bd || (d == bd && (v & 1) == 0)) { bd = d; best = v; }
    }
    return s | best;
}
int main() {
    unsigned short *d8, *d16, h8[65536], h16[65536];
    hipMalloc(&d8, 131072); hipMalloc(&d16, 131072);
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d8, d16);
    hipMemcpy(h8, d8, 131072, hipMemcpyDeviceToHost); hipMemcpy(h16, d16, 131072, hipMemcpyDeviceToHost);
    int bad = 0, badback = 0;
    for (unsigned i = 0; i < 65536; ++i) {
        unsigned short bits = (unsigned short)i;
        _Float16 h; memcpy(&h, &bits, 2);
        const float f = (float)h;
        const unsigned want = float_to_e4m3_rne_sat(f);
        if (!isnan(f) && h8[i] != want) { if (bad < 12) printf("f16 %04x = %g -> fp8 %02x, e4m3 RNE(sat) %02x (%g vs %g)\n", i, f, h8[i], want, e4m3_to_float(h8[i]), e4m3_to_float(want)); ++bad; }
        _Float16 hb; memcpy(&hb, &h16[i], 2);
        if (!isnan(f) && (float)hb != e4m3_to_float(h8[i])) ++badback;
    }
    printf("fp16 -> fp8: %d of 65536 patterns differ from e4m3fn RNE with saturation; fp8 -> fp16 inexact for %d\n", bad, badback);
    return 0;
}
