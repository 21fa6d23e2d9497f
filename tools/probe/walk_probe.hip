// Probe: does the ORDER in which t-marching waves walk a (N*C, T, 56, 56) tensor decide the HBM rate of the depthwise forward kernels?
// The data movement of dw3d_cp_fwd_kernel<56, 1, ..> without its arithmetic: a wave owns a band of 4 rows, per frame step it reads the band's
// 6 input rows (two 16-byte loads per lane, ring of D frames ahead) and writes 4 output rows (one 16-byte store per lane).
//   mode 0: today's walk -- one (channel, t-chunk of TT = 52, band) item per wave, grid = all items (360 resident 52-frame streams);
//   mode 1: short chunks (TT = 10 + 2 halo frames), one item per wave, grid = all items: every wave pays its start-up round trip;
//   mode 2: PERSISTENT waves: the same short items in memory order, wave slot w takes items w, w + W, ..., and the frame ring runs on across the
//           item switch (the next item's first frames are in flight while the current one finishes): a dense moving window and no start-up bubbles.
//   hipcc --offload-arch=gfx950 -O3 -o walk_probe walk_probe.hip && ./walk_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f4 __attribute__((ext_vector_type(4)));
#define P 3136
#define NB 14
#define D 2
#define OOB 0x7fff0000

__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned total) {
    const unsigned q = total >> 3, r = total & 7u, xcd = b & 7u, i = b >> 3;
    return xcd * q + (xcd < r ? xcd : r) + i;
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

struct Item { int nc, t0, t1, band; };
__device__ __forceinline__ Item item_of(long idx, int nchunks, int TT, int T) {
    Item it;
    it.band = (int)(idx % NB); idx /= NB;
    const int chunk = (int)(idx % nchunks);
    it.nc = (int)(idx / nchunks);
    it.t0 = chunk * TT; it.t1 = min(it.t0 + TT, T);
    return it;
}

template <int MODE>
__global__ __launch_bounds__(256, 5) void walk(const float* x, float* y, int NC, int T, int TT, int nchunks, long items, long slots) {
    const int lane = threadIdx.x & 63, wv = uni(threadIdx.x >> 6);
    const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const long w = (long)L * 4 + wv;
    // per-lane offsets inside a plane: 6 input rows = 84 float4 (rows band*4-1 .. band*4+4, clamped by the range check), 4 output rows = 56 float4
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long idx = w; idx < items; idx += (MODE == 2 ? slots : items)) {       // modes 0 / 1: exactly one item
        const Item it = item_of(idx, nchunks, TT, T);
        const int band = uni(it.band), nc = uni(it.nc), t0 = uni(it.t0), t1 = uni(it.t1);
        const int row_lo = max(band * 4 - 1, 0), row_hi = min(band * 4 + 5, 56);
        const int nel = (row_hi - row_lo) * 56;
        int ldo[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) { const int e0 = (k * 64 + lane) * 4; ldo[k] = e0 < nel ? (row_lo * 56 + e0) * 4 : OOB; }
        const int sto = lane < 56 ? (band * 4 * 56 + lane * 4) * 4 : OOB;
        __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (long)nc * T * P), 0, (unsigned)((long)T * P * 4), 0x00020000);
        __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(y + (long)nc * T * P), 0, (unsigned)((long)T * P * 4), 0x00020000);
        // (mode 2 simplification: the ring of this item is primed here; the NEXT item's priming loads are issued right after this item's last
        // in-range fetch by running the loop below one item ahead -- see `pre` registers)
        f4 ring[D + 1][2];
        auto fetch = [&](int f, f4 (&dst)[2]) {
            const bool want = f >= 0 && f < T && f <= t1;
            const int so = uni(want ? f * P * 4 : 0);
#pragma unroll
            for (int k = 0; k < 2; ++k) dst[k] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rx, want ? ldo[k] : OOB, so, 0));
        };
        const int f_first = t0 - 1;
#pragma unroll
        for (int d = 0; d <= D; ++d) fetch(f_first + d, ring[d]);
        for (int f0 = f_first; f0 <= t1; f0 += D + 1) {
#pragma unroll
            for (int j = 0; j <= D; ++j) {
                const int f = f0 + j;
                const f4 a0 = ring[j][0], a1 = ring[j][1];
                fetch(f + D + 1, ring[j]);
                acc += a0 + a1;
                const int to = f - 1;
                const bool emit = to >= t0 && to < t1;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(unsigned __attribute__((ext_vector_type(4))), acc), ry, emit ? sto : OOB, uni(emit ? to * P * 4 : 0), 0);
            }
        }
    }
}

int main() {
    const int NC = 432, T = 256;
    const size_t n = (size_t)NC * T * P;
    float *x, *y;
    if (hipMalloc(&x, n * 4) != hipSuccess || hipMalloc(&y, n * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(x, 0, n * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const double gb = 2.0 * n * 4 / 1e9;
    for (int mode = 0; mode < 3; ++mode)
        for (int tt_i = 0; tt_i < (mode == 0 ? 1 : 4); ++tt_i) {
            const int TT = mode == 0 ? 52 : (tt_i == 0 ? 4 : tt_i == 1 ? 10 : tt_i == 2 ? 16 : 28);
            const int nchunks = (T + TT - 1) / TT;
            const long items = (long)NC * nchunks * NB;
            const long slots = 256L * 5 * 4;                                  // resident waves: 5 workgroups of 4 waves per CU
            const unsigned blocks = (unsigned)(mode == 2 ? slots / 4 : (items + 3) / 4);
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(walk<0>, dim3(blocks), dim3(256), 0, 0, x, y, NC, T, TT, nchunks, items, slots);
                if (mode == 1) hipLaunchKernelGGL(walk<1>, dim3(blocks), dim3(256), 0, 0, x, y, NC, T, TT, nchunks, items, slots);
                if (mode == 2) hipLaunchKernelGGL(walk<2>, dim3(blocks), dim3(256), 0, 0, x, y, NC, T, TT, nchunks, items, slots);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            printf("mode %d (%s) TT = %2d: %.3f ms = %.2f TB/s algorithmic (1:1 read:write, 8 x 54 x 256 x 56 x 56)\n", mode,
                   mode == 0 ? "one long item per wave" : mode == 1 ? "one short item per wave" : "persistent waves, short items in memory order", TT, best, gb / best);
        }
    return 0;
}
