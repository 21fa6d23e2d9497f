// X3D stem conv1_s (x3d_fine.py:210-215): Conv3d(3, 24, (1,3,3), stride (1,2,2), pad (0,1,1), bias=False), forward.
//
// HBM bound (one 3-channel frame in, one 24-channel quarter-resolution frame out), but as a gather-form implicit GEMM it
// runs at 2.6 TB/s: every lane fetches its 27 taps from global memory through a tap table.  Here a workgroup stages a band
// of input rows of one frame in LDS once (coalesced float4 rows, zero halo) and the im2col operand of the MFMA comes out of
// LDS:  Y[co][pos] = sum_k W[co][k] * X[k][pos],  k = (ci,kh,kw) -> v_mfma_f32_32x32x2 with the 32 x 28 weight operand
// resident in registers (14 k-pairs), lane <-> output position, so every store instruction writes full 128-byte lines.
#include "cfn_common.h"
#include <stdlib.h>

typedef float __attribute__((ext_vector_type(16))) st16;
typedef float __attribute__((ext_vector_type(4))) st4;

struct StemArgs {
    const float* x; const float* w; float* y;
    int N, Cout, T, Hi, Wi, Ho, Wo, RB, RIN, WPAD, bands;
};

template <int CI>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const StemArgs a) {
    extern __shared__ __attribute__((aligned(16))) float img[];      // [CI][RIN][WPAD], data at column 4, left halo at 3
    constexpr int KP = (CI * 9 + 1) / 2;                               // k-pairs
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, col = lane & 31;
    unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    const int band = L % a.bands; L /= a.bands;
    const int t = L % a.T;
    const int n = L / a.T;
    const int RIN = a.RIN, WPAD = a.WPAD, Wi = a.Wi, Hi = a.Hi, Wo = a.Wo, Ho = a.Ho;
    const int oh0 = band * a.RB, ih0 = 2 * oh0 - 1;

    // ---- stage the band: CI x RIN rows of Wi floats, float4 per thread, rows outside the image are zero -------------
    const int w4 = Wi >> 2, per_row = w4 + 1;                         // + one float4 slot that carries the left halo
    const int total = CI * RIN * per_row;
    // (all loads of a batch are issued before the first LDS write: a load -> store loop would pay one HBM round trip per
    // iteration)
    for (int e0 = tid; e0 < total; e0 += 256 * 8) {
        st4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * 256;
            const int rowid = e / per_row, c4 = e - rowid * per_row;
            const int ci = rowid / RIN, r = rowid - ci * RIN;
            const int ih = ih0 + r;
            v[u] = (st4){0.f, 0.f, 0.f, 0.f};
            if (e < total && c4 > 0 && ih >= 0 && ih < Hi)
                v[u] = *reinterpret_cast<const st4*>(a.x + (((long)n * CI + ci) * a.T + t) * (long)Hi * Wi + (long)ih * Wi + (c4 - 1) * 4);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * 256;
            const int rowid = e / per_row, c4 = e - rowid * per_row;
            if (e < total) *reinterpret_cast<st4*>(img + rowid * WPAD + c4 * 4) = v[u];   // c4 == 0: columns 0..3 (3 = halo of iw = -1) zero
        }
    }
    // weight operand: lane (co = col, k = 2s + half)
    float wreg[KP];
    int offk[KP];
#pragma unroll
    for (int s = 0; s < KP; ++s) {
        const int k = 2 * s + half;
        const bool kv = k < CI * 9;
        const int ci = k / 9, r9 = k - ci * 9, kh = r9 / 3, kw = r9 - kh * 3;
        wreg[s] = (kv && col < a.Cout) ? a.w[col * (CI * 9) + k] : 0.0f;
        offk[s] = kv ? (ci * RIN + kh) * WPAD + kw : 0;
    }
    __syncthreads();

    const int npos = min(a.RB, Ho - oh0) * Wo;                        // output rows of a band are contiguous in memory
    const long ybase = (((long)n * a.Cout) * a.T + t) * (long)Ho * Wo + (long)oh0 * Wo;
    const long cstride = (long)a.T * Ho * Wo;
    for (int tile = wave; tile * 32 < npos; tile += 4) {
        const int pos = tile * 32 + col;
        const bool valid = pos < npos;
        const int pc = valid ? pos : 0;
        const int ohl = pc / Wo, ow = pc - ohl * Wo;
        const float* lb = img + (2 * ohl) * WPAD + 2 * ow + 3;        // tap (kh, kw) of channel ci: + (ci*RIN + kh)*WPAD + kw
        st16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int s = 0; s < KP; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[s], lb[offk[s]], acc, 0, 0, 0);
        if (valid) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (r & 3) + 8 * (r >> 2) + 4 * half;
                if (co < a.Cout) a.y[ybase + co * cstride + pos] = acc[r];
            }
        }
    }
}

// -1 = shape not handled (the caller uses the implicit-GEMM path)
int stem_fwd_try_launch(const float* x, const float* w, float* y, int N, int Cimg, int Cout, int T, int Hi, int Wi, hipStream_t st) {
    if (Cimg != 3 || Cout > 32 || (Wi & 3) || (Hi & 1) || ((uintptr_t)x & 15)) return -1;
    { const char* e = getenv("CFN_STEM_OFF"); if (e && atoi(e)) return -1; }
    StemArgs a = {x, w, y, N, Cout, T, Hi, Wi, Hi / 2, Wi / 2};
    a.RB = (a.Ho % 8 == 0) ? 8 : 4;
    a.RIN = 2 * a.RB + 1;
    a.WPAD = Wi + 8;
    a.bands = cfn_cdiv(a.Ho, a.RB);
    const size_t lds = (size_t)3 * a.RIN * a.WPAD * sizeof(float);
    if (lds > 64 * 1024) return -1;
    const long blocks = (long)N * T * a.bands;
    if (blocks >= (1L << 31)) return -1;
    auto k = stem_fwd_kernel<3>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(256), lds, st, a);
    return cfn_check_launch("stem_conv_fwd");
}
