#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void glds16(const u32x4* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__global__ __launch_bounds__(64) void k(const u32x4* src, u32x4* out, int off_bytes) {
    extern __shared__ u32x4 lds[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 140 * 1024 / 16; i += 64) lds[i] = u32x4{0xdeadbeefu, 0, 0, 0};
    __syncthreads();
    unsigned base = (unsigned)(size_t)lds + off_bytes;
    base = __builtin_amdgcn_readfirstlane(base);
    glds16(src + lane, base);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[lane] = lds[off_bytes / 16 + lane];
    // also report where the data landed, if elsewhere
    for (int i = lane; i < 140 * 1024 / 16; i += 64)
        if (lds[i][0] != 0xdeadbeefu && (i < off_bytes / 16 || i >= off_bytes / 16 + 64)) out[64][0] = i * 16;
}
int main() {
    std::vector<u32x4> h(64);
    for (int i = 0; i < 64; ++i) h[i] = u32x4{(unsigned)i + 1, 7, 8, 9};
    u32x4 *d, *o;
    hipMalloc(&d, 1024); hipMalloc(&o, 2048);
    hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    int offs[] = {0, 32768, 61440, 65536, 66560, 98304, 131072, 139 * 1024};
    for (int off : offs) {
        hipMemset(o, 0, 2048);
        k<<<1, 64, 140 * 1024>>>(d, o, off);
        std::vector<u32x4> r(65);
        hipError_t e = hipMemcpy(r.data(), o, 65 * 16, hipMemcpyDeviceToHost);
        int ok = 1;
        for (int i = 0; i < 64; ++i) ok &= (r[i][0] == (unsigned)i + 1 && r[i][3] == 9);
        printf("off %6d: %s (err %d) stray-at %u\n", off, ok ? "OK" : "MISMATCH", (int)e, r[64][0]);
    }
}
