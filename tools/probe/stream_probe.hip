// Streaming ceilings on one MI355X for the access shapes of the depthwise kernels (tools/probe: measurement only).
//   hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.hip && ./stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float __attribute__((ext_vector_type(4))) f4;

template <int UN, int NT>
__global__ __launch_bounds__(256) void copy_k(const f4* __restrict__ x, f4* __restrict__ y, long n4) {
    long i = ((long)blockIdx.x * UN) * 256 + threadIdx.x;
    f4 v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
        const long j = i + (long)u * 256;
        if (j < n4) v[u] = (NT & 1) ? __builtin_nontemporal_load(x + j) : x[j];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
        const long j = i + (long)u * 256;
        if (j < n4) { if (NT & 2) __builtin_nontemporal_store(v[u], y + j); else y[j] = v[u]; }
    }
}
// persistent: grid = k * 256 CUs, each block walks chunks
template <int UN, int NT>
__global__ __launch_bounds__(256) void copy_p(const f4* __restrict__ x, f4* __restrict__ y, long n4) {
    for (long base = (long)blockIdx.x * UN * 256; base < n4; base += (long)gridDim.x * UN * 256) {
        f4 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) { const long j = base + u * 256 + threadIdx.x; if (j < n4) v[u] = (NT & 1) ? __builtin_nontemporal_load(x + j) : x[j]; }
#pragma unroll
        for (int u = 0; u < UN; ++u) { const long j = base + u * 256 + threadIdx.x; if (j < n4) { if (NT & 2) __builtin_nontemporal_store(v[u], y + j); else y[j] = v[u]; } }
    }
}
// 4:1 read:write (stride-2 like): read 4 f4, write 1
template <int NT>
__global__ __launch_bounds__(256) void reduce4_k(const f4* __restrict__ x, f4* __restrict__ y, long n4out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4out) return;
    f4 a = x[i * 4], b = x[i * 4 + 1], c = x[i * 4 + 2], d = x[i * 4 + 3];
    f4 r = a + b + c + d;
    if (NT & 2) __builtin_nontemporal_store(r, y + i); else y[i] = r;
}
// read-only
__global__ __launch_bounds__(256) void read_k(const f4* __restrict__ x, float* out, long n4) {
    f4 s = {0, 0, 0, 0};
    for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < n4; j += (long)gridDim.x * 256) s += x[j];
    if (s.x + s.y + s.z + s.w == 123.456f) out[0] = 1.0f;
}
__global__ __launch_bounds__(256) void write_k(f4* y, long n4) {
    for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < n4; j += (long)gridDim.x * 256) y[j] = f4{1, 2, 3, 4};
}

#define TIME(name, bytes, ...)                                                                  \
    do {                                                                                        \
        __VA_ARGS__; hipDeviceSynchronize();                                                    \
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);                            \
        hipEventRecord(e0);                                                                     \
        for (int it = 0; it < 5; ++it) { __VA_ARGS__; }                                         \
        hipEventRecord(e1); hipEventSynchronize(e1);                                            \
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;                                    \
        printf("%-44s %8.3f ms  %8.1f GB/s\n", name, ms, (double)(bytes) / 1e6 / ms);           \
    } while (0)

int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 616562688L;   // floats: 8 x 24 x 256 x 112 x 112 (conv1_t)
    const long n4 = n / 4;
    f4 *x, *y; float* o;
    hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMalloc(&o, 4);
    hipMemset(x, 0, n * 4); hipMemset(y, 0, n * 4);
    char nm[128];
#define COPY(UN, NT) do { snprintf(nm, 128, "copy f4 x%d nt=%d (one pass grid)", UN, NT); \
    TIME(nm, 2.0 * n * 4, hipLaunchKernelGGL((copy_k<UN, NT>), dim3((unsigned)((n4 + 256L * UN - 1) / (256L * UN))), dim3(256), 0, 0, x, y, n4)); } while (0)
    COPY(1, 0); COPY(2, 0); COPY(4, 0); COPY(8, 0);
    COPY(1, 2); COPY(4, 2); COPY(4, 1); COPY(4, 3); COPY(8, 3);
#define COPYP(UN, NT, G) do { snprintf(nm, 128, "copy f4 x%d nt=%d persistent grid %d", UN, NT, G); \
    TIME(nm, 2.0 * n * 4, hipLaunchKernelGGL((copy_p<UN, NT>), dim3(G), dim3(256), 0, 0, x, y, n4)); } while (0)
    COPYP(4, 0, 1024); COPYP(4, 0, 2048); COPYP(4, 0, 4096); COPYP(8, 0, 2048); COPYP(4, 2, 2048); COPYP(2, 0, 4096); COPYP(2, 0, 8192);
    TIME("reduce 4:1", 1.25 * n * 4, hipLaunchKernelGGL((reduce4_k<0>), dim3((unsigned)((n4 / 4 + 255) / 256)), dim3(256), 0, 0, x, y, n4 / 4));
    TIME("reduce 4:1 nt store", 1.25 * n * 4, hipLaunchKernelGGL((reduce4_k<2>), dim3((unsigned)((n4 / 4 + 255) / 256)), dim3(256), 0, 0, x, y, n4 / 4));
    TIME("read only (grid 4096)", 1.0 * n * 4, hipLaunchKernelGGL(read_k, dim3(4096), dim3(256), 0, 0, x, o, n4));
    TIME("write only (grid 4096)", 1.0 * n * 4, hipLaunchKernelGGL(write_k, dim3(4096), dim3(256), 0, 0, y, n4));
    hipMemcpyAsync(y, x, n * 4, hipMemcpyDeviceToDevice, 0);
    TIME("hipMemcpy D2D", 2.0 * n * 4, hipMemcpyAsync(y, x, n * 4, hipMemcpyDeviceToDevice, 0));
    return 0;
}
