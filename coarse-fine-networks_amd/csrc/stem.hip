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

// ---- weight gradient of conv1_s at 224x224 (x3d_fine.py:210-215):  gw[co][(ci,kh,kw)] = sum over (n, t, oh, ow) of gy x im2col(x) ----
// HBM bound: gy (24 channels, 112x112) and x (3 channels, 224x224) are read once: 144 KB + 58 KB per frame; the product is ONE 24 x 27 tile
// whose K dimension (the output positions) is as long as the tensors.  The gather-form kernel (pw_wgrad_direct_kernel: every lane fetches
// its im2col operand from global memory) ran at 2.7 TB/s.  Here a persistent workgroup of 8 waves walks a contiguous run of (frame, band of
// 4 output rows) items: the band's 9 input rows x 3 channels and its 24 x 4 rows of gy are staged in LDS by coalesced float4 loads (the
// loads of the NEXT item are in flight while the current one is multiplied), wave w multiplies positions [56 w, 56 w + 56) of the band:
// 28 v_mfma_f32_32x32x2 with A = gy (row co, k = position) and B = x (column (ci, kh, kw), k = position) both ds_read_b32 with immediate
// offsets; fp32 accumulators per wave, reduced through LDS at the end, one fp64 atomic per element and workgroup.
// LDS: x image [3][9][228] (+8 floats per channel: the 27 operand columns then start in 27 different banks), data at column 4, column 3 =
// the zero halo of input column -1; gy image [24][450].
struct StemWgArgs {
    const float* gy; const float* x; double* gw;
    int N, T, items, per_block;
};

__global__ __launch_bounds__(512, 2) void stem_wgrad_kernel(const StemWgArgs a) {
    typedef float __attribute__((ext_vector_type(2))) st2;
    constexpr int WI = 224, WO = 112, RB = 4, RIN = 9, PITCH = 228, CIS = RIN * PITCH + 8, GP = 450, W4 = 56, G4 = 28;
    constexpr int P = WI * WI, PO = WO * WO, BANDS = WO / RB, OOB = 0x7fff0000;
    constexpr int XU = 3 * RIN * W4, NXU = (XU + 511) / 512;               // float4 units of x per item and thread
    constexpr int GU = 24 * RB * G4, NGU = (GU + 511) / 512;               // float4 units of gy per item and thread
    __shared__ __attribute__((aligned(16))) float img[3 * CIS];
    __shared__ __attribute__((aligned(16))) float gbuf[24 * GP];
    const int tid = threadIdx.x, lane = tid & 63, wave = cfn_uni(tid >> 6), h = lane >> 5, p = lane & 31;
    const int T = a.T;

    for (int i = tid; i < 3 * CIS; i += 512) img[i] = 0.0f;                // the halo column stays zero for good

    // staging units of this thread (the same for every item; the item adds a frame / row offset)
    int xo[NXU], xl[NXU], xr[NXU];
#pragma unroll
    for (int k = 0; k < NXU; ++k) {
        const int e = k * 512 + tid;
        const int rowid = e / W4, c4 = e - rowid * W4, ci = rowid / RIN, r = rowid - ci * RIN;
        const bool in = e < XU;
        xo[k] = in ? (int)(((long)ci * T * P + (long)r * WI + c4 * 4) * 4) : OOB;
        xl[k] = in ? ci * CIS + r * PITCH + 4 + c4 * 4 : -1;
        xr[k] = r;
    }
    int go[NGU], gl[NGU];
#pragma unroll
    for (int k = 0; k < NGU; ++k) {
        const int e = k * 512 + tid;
        const int co = e / (RB * G4), rem = e - co * (RB * G4);           // rem = row * 28 + c4: the band's rows are contiguous in memory
        const bool in = e < GU;
        go[k] = in ? (int)(((long)co * T * PO + rem * 4) * 4) : OOB;
        gl[k] = in ? co * GP + rem * 4 : -1;
    }
    // operands of this lane: A row co = p (rows 24-31 repeat row 23 and are dropped), B column (ci, kh, kw) = p (columns 27-31 repeat column 0)
    const int ohl = wave >> 1, ow0 = (wave & 1) * 56;
    const int colp = p < 27 ? p : 0;
    const int ci = colp / 9, kh = (colp - ci * 9) / 3, kw = colp - ci * 9 - kh * 3;
    const float* bp = img + ci * CIS + (2 * ohl + kh) * PITCH + 2 * ow0 + kw + 3 + 2 * h;
    const float* ap = gbuf + min(p, 23) * GP + ohl * WO + ow0 + h;

    st16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.0f;

    const int first = blockIdx.x * a.per_block, last = min(first + a.per_block, a.items);
    st4 fx[NXU], fg[NGU];
    auto fetch = [&](int item) {
        const bool on = item < last;
        const int band = item % BANDS, ft = item / BANDS;                  // ft = n * T + t
        const int n = ft / T, t = ft - n * T;
        const int ih0 = 2 * band * RB - 1;
        // (x: n's three channels are T * P apart; gy: n's 24 channels T * PO apart: one descriptor per tensor over the whole sample)
        __amdgpu_buffer_rsrc_t rx = cfn_rsrc(a.x + (long)n * 3 * T * P, (unsigned)((long)3 * T * P * 4));
        __amdgpu_buffer_rsrc_t rg = cfn_rsrc(a.gy + (long)n * 24 * T * PO, (unsigned)((long)24 * T * PO * 4));
        // (the scalar offset is not range checked and must not be negative: the band's first input row, -1 for band 0, goes into the vector offset)
        const int sx = cfn_uni(on ? t * P * 4 : 0), sg = cfn_uni(on ? (t * PO + band * RB * WO) * 4 : 0), rowoff = cfn_uni(ih0 * WI * 4);
#pragma unroll
        for (int k = 0; k < NXU; ++k)
            fx[k] = __builtin_bit_cast(st4, __builtin_amdgcn_raw_buffer_load_b128(rx, (on && xo[k] != OOB && ih0 + xr[k] >= 0) ? xo[k] + rowoff : OOB, sx, 0));
#pragma unroll
        for (int k = 0; k < NGU; ++k)
            fg[k] = __builtin_bit_cast(st4, __builtin_amdgcn_raw_buffer_load_b128(rg, on ? go[k] : OOB, sg, 0));
    };
    auto put = [&]() {                                                      // rows above the image were not read: zeros
#pragma unroll
        for (int k = 0; k < NXU; ++k)
            if (xl[k] >= 0) *reinterpret_cast<st4*>(img + xl[k]) = fx[k];
#pragma unroll
        for (int k = 0; k < NGU; ++k)
            if (gl[k] >= 0) {
                *reinterpret_cast<st2*>(gbuf + gl[k]) = (st2){fg[k].x, fg[k].y};
                *reinterpret_cast<st2*>(gbuf + gl[k] + 2) = (st2){fg[k].z, fg[k].w};
            }
    };
    __syncthreads();
    fetch(first);
    put();
    __syncthreads();
    for (int item = first; item < last; ++item) {
        fetch(item + 1);                                                    // in flight while this item is multiplied
#pragma unroll
        for (int j = 0; j < 28; j += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * j], bp[4 * j], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * j + 2], bp[4 * j + 4], acc1, 0, 0, 0);
        }
        __syncthreads();
        put();
        __syncthreads();
    }
    // reduce the 8 waves' tiles through LDS (the gy image is free now: 8 x 16 x 64 floats = 32 KB of its 43 KB)
#pragma unroll
    for (int r = 0; r < 16; ++r) gbuf[(wave * 16 + r) * 64 + lane] = acc0[r] + acc1[r];
    __syncthreads();
    for (int e = tid; e < 16 * 64; e += 512) {
        const int r = e >> 6, l = e & 63, hh = l >> 5, col = l & 31;
        const int co = (r & 3) + 8 * (r >> 2) + 4 * hh;
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < 8; ++w) s += gbuf[(w * 16 + r) * 64 + l];
        if (co < 24 && col < 27) cfn_add64(&a.gw[co * 27 + col], (double)s);
    }
}

// -1 = shape not handled (the caller uses the implicit-GEMM path); probe: 0 = handled, nothing launched
int stem_wgrad_try_launch(const float* gy, const float* x, double* gw, int N, int Cimg, int Cout, int T, int Hi, int Wi, hipStream_t st, bool probe) {
    if (Cimg != 3 || Cout != 24 || Hi != 224 || Wi != 224 || (((uintptr_t)x | (uintptr_t)gy) & 15)) return -1;
    { const char* e = getenv("CFN_STEM_WG_OFF"); if (e && atoi(e)) return -1; }
    if ((long)24 * T * 112 * 112 * 4 >= 0x7fff0000L) return -1;
    const long items = (long)N * T * 28;
    if (items >= (1L << 30)) return -1;
    if (probe) return 0;
    static int cus = 0;
    if (!cus) { int dev = 0; hipDeviceProp_t pr; cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256; }
    static const int bpc = getenv("CFN_STEM_WG_BPC") ? atoi(getenv("CFN_STEM_WG_BPC")) : 2;     // persistent workgroups per CU (8 x 256 x 224 x 224: 1: 726 us, 2: 655 us; gather-form kernel 1346)
    long blocks = (long)cus * (bpc > 0 ? bpc : 1);
    if (blocks > items) blocks = items;
    const long per = (items + blocks - 1) / blocks;
    blocks = (items + per - 1) / per;
    StemWgArgs a = {gy, x, gw, N, T, (int)items, (int)per};
    hipLaunchKernelGGL(stem_wgrad_kernel, dim3((unsigned)blocks), dim3(512), 0, st, a);
    return cfn_check_launch("stem_conv_wgrad");
}
