// What does the scale operand of v_cvt_scalef32_pk_fp8_f32 do -- multiply or divide?  (MI355X; round 5)
//   hipcc --offload-arch=gfx950 -O2 -o fp8_scale_check tools/exp/fp8_scale_check.hip && ./fp8_scale_check
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
typedef short s2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, unsigned* out, float sc) {
    const float a = in[2 * threadIdx.x], b = in[2 * threadIdx.x + 1];
    s2 z = {0, 0};
    const s2 r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(z, a, b, sc, false);
    out[threadIdx.x] = __builtin_bit_cast(unsigned, r);
}
static float e4m3(unsigned c) {
    const unsigned s = c >> 7, e = (c >> 3) & 15, m = c & 7;
    if (e == 15 && m == 7) return NAN;
    const float x = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, (int)e - 7);
    return s ? -x : x;
}
int main() {
    float h[8] = {3.0f, -0.75f, 0.001953125f, 100.0f, 1e-3f, 17.0f, 500.0f, 0.3f};
    float* d; unsigned* o; unsigned ho[4];
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 16);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (float sc : {1.0f, 2.0f, 0.5f, 2048.0f, 1.0f / 2048.0f, 3.0f}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, d, o, sc);
        hipMemcpy(ho, o, 16, hipMemcpyDeviceToHost);
        printf("scale %g:", sc);
        for (int i = 0; i < 4; ++i) printf("  (%g, %g) -> (%g, %g)", h[2 * i], h[2 * i + 1], e4m3(ho[i] & 0xff), e4m3((ho[i] >> 8) & 0xff));
        printf("\n");
    }
    return 0;
}
