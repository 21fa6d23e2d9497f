// Store cache-policy probe (tools/probe: measurement only): write-only, 1:1 copy and 1:2.25 (24 -> 54 channel rows) streams with
// every aux (sc0 / nt / sc1) combination of buffer_store_dwordx4.   hipcc --offload-arch=gfx950 -O3 -o store_policy_probe store_policy_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float __attribute__((ext_vector_type(4))) f4;
typedef unsigned __attribute__((ext_vector_type(4))) u4;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
// each block owns a 1 GiB-safe window: base pointer advanced per block group (uniform), 32-bit offsets inside
template <int AUX, int RD>          // RD: float4 reads per 4 float4 writes (0 = write only, 4 = copy, 2 ~ 24->54)
__global__ __launch_bounds__(256) void k(const f4* __restrict__ x, f4* __restrict__ y, long n4) {
    const long i0 = ((long)blockIdx.x * 256 + threadIdx.x);
    const long stride = (long)gridDim.x * 256;
    f4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = f4{1.f, 2.f, 3.f, (float)u};
    const long seg = i0 / (1L << 24);          // 256 MiB segments keep offsets in 32 bits
    (void)seg;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long j = i0 + u * stride;
        if (u < RD && j < n4) v[u] = x[j];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long j = i0 + u * stride;
        if (j < n4) {
            const long base = (j >> 24) << 24;                     // wave-uniform only when the block does not straddle: use flat addressing through a per-lane descriptor-free path
            __amdgpu_buffer_rsrc_t r = rsrc(y + base, 0x10000000u);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v[u]), r, (int)((j - base) * 16), 0, AUX);
        }
    }
}
#define RUN(AUX, RD, name)                                                                           \
    do {                                                                                             \
        const unsigned grid = (unsigned)((n4 / 4 + 255) / 256);                                      \
        hipLaunchKernelGGL((k<AUX, RD>), dim3(grid), dim3(256), 0, 0, x, y, n4); hipDeviceSynchronize(); \
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0);             \
        for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((k<AUX, RD>), dim3(grid), dim3(256), 0, 0, x, y, n4); \
        hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5; \
        printf("%-22s aux=%2d  %7.3f ms  %7.1f GB/s\n", name, AUX, ms, (double)n4 * 16 * (1.0 + RD / 4.0) / 1e6 / ms); \
    } while (0)
#define ALL(RD, name) RUN(0, RD, name); RUN(1, RD, name); RUN(2, RD, name); RUN(3, RD, name); RUN(16, RD, name); RUN(17, RD, name); RUN(18, RD, name); RUN(19, RD, name)
int main() {
    const long n = 616562688L, n4 = n / 4;
    f4 *x, *y; hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMemset(x, 0, n * 4); hipMemset(y, 0, n * 4);
    ALL(0, "write only"); ALL(4, "copy 1:1"); ALL(2, "read 1 : write 2");
    return 0;
}
