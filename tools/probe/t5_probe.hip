// Access-pattern probe for conv1_t (5x1x1 depthwise along t, 8 x 24 x 256 x 112 x 112 fp32): the t-marching persistent mapping
// of dwt5_fwd_stream_kernel against a FLAT mapping (one thread = TO consecutive output frames of one float4 position, blocks
// in memory order, each XCD walking one contiguous eighth of the tensor so that the temporal halo re-reads hit its own L2).
// Measurement only (tools/probe).   hipcc --offload-arch=gfx950 -O3 -o t5_probe t5_probe.hip && ./t5_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float __attribute__((ext_vector_type(4))) f4;
typedef unsigned __attribute__((ext_vector_type(4))) u4;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned total) {
    const unsigned q = total >> 3, r = total & 7u;
    const unsigned xcd = b & 7u, i = b >> 3;
    return xcd * q + (xcd < r ? xcd : r) + i;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

struct Args { const float* x; const float* w; float* y; double* s1; double* s2; int C, T, plane, TT, nchunks; };

// ---- the marching kernel (as csrc/dwt5.hip) -----------------------------------------------------------------------------
template <int PF>
__global__ __launch_bounds__(256) void march_k(const Args a) {
    constexpr int OOB = 0x7ffffff0, RING = 5 + PF;
    __shared__ float sh[8];
    const long nc = blockIdx.y;
    const int c = (int)(nc % a.C);
    const int chunk = blockIdx.x % a.nchunks, pc = blockIdx.x / a.nchunks;
    const int p = (pc * 256 + (int)threadIdx.x) * 4;
    const bool ok = p < a.plane;
    const int T = a.T, t0 = chunk * a.TT, t1 = min(t0 + a.TT, T), plane = a.plane;
    __amdgpu_buffer_rsrc_t rx = rsrc(a.x + nc * T * (long)plane, (unsigned)((long)T * plane * 4));
    __amdgpu_buffer_rsrc_t ry = rsrc(a.y + nc * T * (long)plane, (unsigned)((long)T * plane * 4));
    const int vx = ok ? p * 4 : OOB;
    float wk[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) wk[k] = __builtin_bit_cast(float, uni(__builtin_bit_cast(int, a.w[c * 5 + k])));
    auto ld = [&](int t) -> f4 {
        const bool tv = t >= 0 && t < T && t <= t1 + 1;
        return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rx, tv ? vx : OOB, tv ? t * plane * 4 : 0, 0));
    };
    f4 R[RING];
#pragma unroll
    for (int k = 0; k < RING - 1; ++k) R[k] = ld(t0 - 2 + k);
    float st1 = 0.f, st2 = 0.f;
    for (int tb = t0; tb < t1; tb += RING) {
#pragma unroll
        for (int j = 0; j < RING; ++j) {
            const int t = tb + j;
            const bool em = t < t1;
            R[(j + RING - 1) % RING] = ld(t + 2 + PF);
            f4 y = R[j % RING] * wk[0] + R[(j + 1) % RING] * wk[1] + R[(j + 2) % RING] * wk[2] + R[(j + 3) % RING] * wk[3] + R[(j + 4) % RING] * wk[4];
            u4 d = __builtin_bit_cast(u4, y);
            __builtin_amdgcn_raw_buffer_store_b128(d, ry, em ? vx : OOB, em ? t * plane * 4 : 0, 0);
            asm volatile("s_nop 1" : "+v"(d));
            const float m = (em && ok) ? 1.0f : 0.0f;
            const f4 ym = y * m;
            st1 += ym.x + ym.y + ym.z + ym.w;
            st2 += ym.x * y.x + ym.y * y.y + ym.z * y.z + ym.w * y.w;
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    st1 = wave_sum(st1); st2 = wave_sum(st2);
    if (lane == 0) { sh[wave] = st1; sh[4 + wave] = st2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&a.s1[nc], (double)(sh[0] + sh[1] + sh[2] + sh[3]));
        atomicAdd(&a.s2[nc], (double)(sh[4] + sh[5] + sh[6] + sh[7]));
    }
}

// ---- flat: thread = TO output frames x one float4; item = ((nc * T/TO + tg) * P4 + p4); blocks in memory order ------------
// REMAP: each XCD owns a contiguous eighth of the items.  STATS: 0 none, 1 block atomics, 2 per-block partials to scratch.
template <int TO, bool REMAP, int STATS, int NT, int ORDER = 0>
__global__ __launch_bounds__(256) void flat_k(const Args a, double* scratch) {
    constexpr int OOB = 0x7ffffff0;
    __shared__ float sh[8];
    unsigned L = REMAP ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    if (ORDER == 1) {            // scramble the order INSIDE each XCD's eighth (gridDim.x / 8 is even here; odd multiplier => bijection mod 2^k only: use q = grid/8 and a multiplier coprime to q)
        const unsigned q = gridDim.x >> 3, x = L / q, i = L - x * q;
        L = x * q + (unsigned)(((unsigned long long)i * 1000003ull) % q);
    }
    if (ORDER == 2) {            // p-major inside a (n, c): consecutive blocks walk t at one 4 KB column (the marching order, but flat)
        const unsigned bpn = (unsigned)(a.T / TO) * (unsigned)(a.plane >> 2) / 256u;      // blocks per (n, c)
        const unsigned tgs = (unsigned)(a.T / TO), ppb = bpn / tgs;                       // 49 / 4 is not whole: only exact when P4 % 256 == 0
        (void)ppb; (void)bpn;
    }
    const int P4 = a.plane >> 2, TG = a.T / TO;
    const unsigned per_nc = (unsigned)TG * P4;                    // a multiple of 256 for the probed shape
    const unsigned item0 = L * 256u;
    const int nc = uni((int)(item0 / per_nc));
    const unsigned in_nc = item0 - (unsigned)nc * per_nc + threadIdx.x;
    const int tg = in_nc / P4, p4 = in_nc - tg * P4;
    const int c = nc % a.C, T = a.T, plane = a.plane;
    const int t0 = tg * TO;
    __amdgpu_buffer_rsrc_t rx = rsrc(a.x + (long)nc * T * plane, (unsigned)((long)T * plane * 4));
    __amdgpu_buffer_rsrc_t ry = rsrc(a.y + (long)nc * T * plane, (unsigned)((long)T * plane * 4));
    float wk[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) wk[k] = __builtin_bit_cast(float, uni(__builtin_bit_cast(int, a.w[c * 5 + k])));
    f4 R[TO + 4];
#pragma unroll
    for (int k = 0; k < TO + 4; ++k) {
        const int t = t0 - 2 + k;
        const bool tv = t >= 0 && t < T;
        R[k] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rx, tv ? (t * plane + p4 * 4) * 4 : OOB, 0, 0));
    }
    float st1 = 0.f, st2 = 0.f;
#pragma unroll
    for (int j = 0; j < TO; ++j) {
        const f4 y = R[j] * wk[0] + R[j + 1] * wk[1] + R[j + 2] * wk[2] + R[j + 3] * wk[3] + R[j + 4] * wk[4];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, y), ry, ((t0 + j) * plane + p4 * 4) * 4, 0, NT ? 2 : 0);
        st1 += y.x + y.y + y.z + y.w;
        st2 += y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
    }
    if (STATS) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        st1 = wave_sum(st1); st2 = wave_sum(st2);
        if (lane == 0) { sh[wave] = st1; sh[4 + wave] = st2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const double v1 = (double)(sh[0] + sh[1] + sh[2] + sh[3]), v2 = (double)(sh[4] + sh[5] + sh[6] + sh[7]);
            if (STATS == 1) { atomicAdd(&a.s1[nc], v1); atomicAdd(&a.s2[nc], v2); }
            else { scratch[2 * (long)L] = v1; scratch[2 * (long)L + 1] = v2; }
        }
    }
}

__global__ void checksum_k(const float* y, long n, double* out) {
    double s = 0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) s += (double)y[i] * (double)((i % 97) + 1);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}

#define TIME(name, ...)                                                                         \
    do {                                                                                        \
        hipMemset(a.y, 0, n * 4); hipMemset(a.s1, 0, NC * 8); hipMemset(a.s2, 0, NC * 8);       \
        __VA_ARGS__; hipDeviceSynchronize();                                                    \
        double h1[4] = {0, 0, 0, 0}; hipMemset(cs, 0, 8);                                       \
        hipLaunchKernelGGL(checksum_k, dim3(4096), dim3(256), 0, 0, a.y, n, cs);                \
        hipMemcpy(&h1[0], cs, 8, hipMemcpyDeviceToHost); hipMemcpy(&h1[1], a.s1 + 5, 8, hipMemcpyDeviceToHost);  \
        hipMemcpy(&h1[2], a.s2 + 5, 8, hipMemcpyDeviceToHost);                                  \
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);                            \
        float best = 1e9f, tot = 0;                                                             \
        for (int it = 0; it < 6; ++it) {                                                        \
            hipEventRecord(e0); __VA_ARGS__; hipEventRecord(e1); hipEventSynchronize(e1);       \
            float ms; hipEventElapsedTime(&ms, e0, e1); tot += ms; if (ms < best) best = ms; }  \
        printf("%-52s avg %7.3f ms  best %7.3f ms  %7.1f GB/s   checksum %.6e  s1[5] %.6e s2[5] %.6e\n", name, tot / 6, best, \
               2.0 * n * 4 / 1e6 / (tot / 6), h1[0], h1[1], h1[2]);                              \
    } while (0)

int main() {
    const int N = 8, C = 24, T = 256, plane = 112 * 112;
    const long NC = (long)N * C, n = NC * T * plane;
    Args a; float* x; float* w; double* cs; double* scratch;
    hipMalloc(&x, n * 4); hipMalloc(&a.y, n * 4); hipMalloc(&w, C * 5 * 4); hipMalloc(&a.s1, NC * 8); hipMalloc(&a.s2, NC * 8); hipMalloc(&cs, 8);
    hipMalloc(&scratch, 16L * 4 * 1024 * 1024);
    {   // pseudo-random input on the host would take a while at 2.4 GB: fill by a kernel-free pattern through hipMemcpy of a 64 MB tile
        const long tile = 16L << 20; float* h = (float*)malloc(tile * 4);
        unsigned s = 12345u;
        for (long i = 0; i < tile; ++i) { s = s * 1664525u + 1013904223u; h[i] = (float)((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
        for (long o = 0; o < n; o += tile) hipMemcpy(x + o, h, (size_t)((n - o < tile ? n - o : tile) * 4), hipMemcpyHostToDevice);
        float hw[24 * 5]; for (int i = 0; i < C * 5; ++i) hw[i] = 0.1f * (float)((i * 7) % 11 - 5);
        hipMemcpy(w, hw, sizeof(hw), hipMemcpyHostToDevice); free(h);
    }
    a.x = x; a.w = w; a.C = C; a.T = T; a.plane = plane;
    for (int TT : {64, 32}) {
        a.TT = TT; a.nchunks = T / TT;
        char nm[96]; snprintf(nm, 96, "march TT=%d (dwt5_fwd_stream mapping)", TT);
        TIME(nm, hipLaunchKernelGGL(march_k<3>, dim3((unsigned)(((plane / 4 + 255) / 256) * a.nchunks), (unsigned)NC), dim3(256), 0, 0, a));
        snprintf(nm, 96, "march TT=%d PF=6", TT);
        TIME(nm, hipLaunchKernelGGL(march_k<6>, dim3((unsigned)(((plane / 4 + 255) / 256) * a.nchunks), (unsigned)NC), dim3(256), 0, 0, a));
        snprintf(nm, 96, "march TT=%d PF=10", TT);
        TIME(nm, hipLaunchKernelGGL(march_k<10>, dim3((unsigned)(((plane / 4 + 255) / 256) * a.nchunks), (unsigned)NC), dim3(256), 0, 0, a));
    }
#define FLAT(TO, REMAP, STATS, NT) do { char nm[96]; snprintf(nm, 96, "flat TO=%d remap=%d stats=%d nt=%d", TO, REMAP, STATS, NT); \
    TIME(nm, hipLaunchKernelGGL((flat_k<TO, REMAP, STATS, NT>), dim3((unsigned)(n / 4 / TO / 256)), dim3(256), 0, 0, a, scratch)); } while (0)
    FLAT(1, true, 1, 0); FLAT(1, false, 1, 0); FLAT(1, true, 0, 0); FLAT(1, true, 2, 0);
    FLAT(2, true, 1, 0); FLAT(2, false, 1, 0); FLAT(2, true, 0, 0); FLAT(2, true, 2, 0); FLAT(2, true, 2, 1);
    FLAT(4, true, 1, 0); FLAT(4, false, 1, 0); FLAT(4, true, 0, 0); FLAT(4, true, 2, 0); FLAT(4, true, 2, 1);
    FLAT(8, true, 1, 0); FLAT(8, true, 2, 0);
    FLAT(16, true, 1, 0); FLAT(16, true, 0, 0);
    { char nm[96]; snprintf(nm, 96, "flat TO=4 remap=1 stats=1 scrambled inside XCD");
      TIME(nm, hipLaunchKernelGGL((flat_k<4, true, 1, 0, 1>), dim3((unsigned)(n / 4 / 4 / 256)), dim3(256), 0, 0, a, scratch)); }
    { char nm[96]; snprintf(nm, 96, "flat TO=8 remap=1 stats=1 scrambled inside XCD");
      TIME(nm, hipLaunchKernelGGL((flat_k<8, true, 1, 0, 1>), dim3((unsigned)(n / 4 / 8 / 256)), dim3(256), 0, 0, a, scratch)); }
    return 0;
}
