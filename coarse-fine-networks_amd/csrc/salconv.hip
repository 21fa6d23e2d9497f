// Grid Pool saliency convolutions (x3d_coarse.py:362-366, :379-381): Conv3d(24, 24, (3,3,3), stride (2,2,2), padding 1) on the
// layer-1 output (56x56 planes, then 28x28) -- forward, data gradient, weight gradient as LDS-tiled fp32-MFMA kernels.
//
// Until round 4 these ran as an im2col gather through the pointwise GEMM (every lane fetched its 648 taps from global memory
// through a tap table: 2.37 ms per launch for 0.69 GB at 8 clips x 256 frames = 0.29 TB/s).  The conv is 25 GFLOP against
// 0.69 GB (AI 36 flop/B): its floor is the fp32 matrix pipe (M = 24 of the 32 MFMA rows, 28 of 32 columns -> ~0.24 ms), not HBM.
//
// Forward:  Y[co][pos] = sum_k W[co][k] X[k][pos],  k = (ci, kt, kh, kw), on v_mfma_f32_32x32x2_f32.
//   * a workgroup (4 waves, one per SIMD, up to 512 VGPRs each) owns (sample, band of 4 tiles of output rows, chunk of TO output
//     frames) and marches over the input frames; every input frame is staged ONCE and consumed on the spot: an even frame 2 to
//     feeds temporal tap 1 of output frame to, an odd frame 2 to + 1 feeds tap 2 of frame to AND tap 0 of frame to + 1 from the
//     same LDS operand (two accumulator sets per tile);
//   * the contraction is split over the waves by INPUT CHANNEL (6 each): a wave stages only its own channels -- coalesced float4
//     buffer loads one frame ahead, prologue relu(A x + B) applied on the way, wave-private LDS image with a zero halo, no
//     workgroup barrier -- and keeps its 6 x 27 x 32 weight slice in 81 registers (MFMA A operand); the B operand of every MFMA
//     is one ds_read_b32 at lane base + immediate offset;
//   * once per output frame the four partial tiles meet in LDS (fixed summation order: bit-repeatable), wave j finishes tile j:
//     per-channel statistics in-lane (reduced once per chunk, fp64 atomics), stores of whole output rows.
#include "pw_common.h"
#include <stdlib.h>

typedef float __attribute__((ext_vector_type(16))) sv16;
typedef float __attribute__((ext_vector_type(4))) sv4;
typedef unsigned __attribute__((ext_vector_type(2))) cp_u2_t;

struct SalArgs {
    const float* x; const double* pa; const double* pb; const float* w; float* y; double* s1; double* s2;
    int N, Cout, T, To, Hi, Ho, bands, nchunks, TO;
};

#define SAL_CIN 24
#define SAL_CW 6          // input channels per wave

__device__ __forceinline__ void sal_wave_sync() {     // LDS ops of a wave run in order; only the compiler has to be told
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int WI, int NT, bool PRO>
__global__ __launch_bounds__(256, NT == 4 ? 1 : 2) void sal_fwd_kernel(const SalArgs a) {
    constexpr int WO = WI / 2, TR = 32 / WO, RB = NT * TR, RIN = 2 * RB + 1, PITCH = WI + 4, W4 = WI / 4;
    constexpr int UNITS = SAL_CW * RIN * W4, NLD = (UNITS + 63) / 64, IMG = SAL_CW * RIN * PITCH, OOB = 0x7fff0000;
    constexpr int SH = 4 / NT, R4 = NT;                 // waves that share the finishing of one tile, float4 register groups per wave
    static_assert(TR >= 1 && TR * WO <= 32 && WI % 4 == 0 && (NT == 1 || NT == 2 || NT == 4), "geometry");
    extern __shared__ __attribute__((aligned(16))) float smem[];          // 4 wave images | red[tile][wave][4][64] float4
    const int tid = threadIdx.x, lane = tid & 63, wv = cfn_uni(tid >> 6), half = lane >> 5, p = lane & 31;
    unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    const int band = cfn_uni((int)(L % a.bands)); L /= a.bands;
    const int chunk = cfn_uni((int)(L % a.nchunks));
    const int n = cfn_uni((int)(L / a.nchunks));
    const int T = a.T, Hi = a.Hi, Ho = a.Ho, To = a.To, Cout = a.Cout;
    const int to0 = chunk * a.TO, nto = min(a.TO, To - to0);
    const int oh0 = band * RB, ih0 = 2 * oh0 - 1;
    float* img = smem + wv * IMG;
    sv4* red = reinterpret_cast<sv4*>(smem + 4 * IMG);

    for (int i = lane; i < IMG; i += 64) img[i] = 0.0f;                   // halo column / rows outside the plane stay zero

    // loader units: float4 e of this wave's (6 channels x RIN rows x WI columns) slab of one frame
    int ldo[NLD], lo[NLD];
    float ua[PRO ? NLD : 1], ub[PRO ? NLD : 1];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int e = k * 64 + lane;
        const int rowid = e / W4, c4 = e - rowid * W4;
        const int ci = rowid / RIN, r = rowid - ci * RIN;
        const int ih = ih0 + r;
        const bool ok = e < UNITS && ih >= 0 && ih < Hi;
        ldo[k] = ok ? (((ci * T) * Hi + ih) * WI + c4 * 4) * 4 : OOB;
        lo[k] = (ci * RIN + r) * PITCH + 4 + c4 * 4;
        if (PRO) {
            const int cg = n * SAL_CIN + wv * SAL_CW + (ok ? ci : 0);
            ua[k] = (float)a.pa[cg]; ub[k] = (float)a.pb[cg];
        }
    }
    const long P = (long)Hi * WI;
    __amdgpu_buffer_rsrc_t rx = cfn_rsrc(a.x + ((long)n * SAL_CIN + wv * SAL_CW) * T * P, (unsigned)((long)SAL_CW * T * P * 4));
    sv4 fr[NLD];
    auto fetch = [&](int f) {
        const int so = cfn_uni((int)(f * P * 4));
#pragma unroll
        for (int k = 0; k < NLD; ++k) fr[k] = __builtin_bit_cast(sv4, __builtin_amdgcn_raw_buffer_load_b128(rx, ldo[k], so, 0));
    };
    auto stage = [&]() {
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            if (ldo[k] != OOB) {
                sv4 v = fr[k];
                if (PRO) {
                    v.x = fmaxf(fmaf(v.x, ua[k], ub[k]), 0.0f); v.y = fmaxf(fmaf(v.y, ua[k], ub[k]), 0.0f);
                    v.z = fmaxf(fmaf(v.z, ua[k], ub[k]), 0.0f); v.w = fmaxf(fmaf(v.w, ua[k], ub[k]), 0.0f);
                }
                *reinterpret_cast<sv4*>(img + lo[k]) = v;
            }
        }
    };

    // weights: MFMA A operand, lane (m = co = p, k = 2 s + half), s = (c2, kh, kw), input channel 6 wv + 2 c2 + half
    float wr[3][27];
#pragma unroll
    for (int kt = 0; kt < 3; ++kt)
#pragma unroll
        for (int s = 0; s < 27; ++s) {
            const int c2 = s / 9, r9 = s - c2 * 9;
            wr[kt][s] = p < Cout ? a.w[(long)p * (SAL_CIN * 27) + (wv * SAL_CW + 2 * c2 + half) * 27 + kt * 9 + r9] : 0.0f;
        }
    // B operand: lane (position p of a tile, k): image address = lb + ((2 c2) RIN + kh + 2 j TR) PITCH + kw
    const bool pv = p < TR * WO;
    const int pr = pv ? p / WO : 0, pc = pv ? p - pr * WO : 0;
    const float* lb = img + (half * RIN + 2 * pr) * PITCH + 2 * pc + 3;

    sv16 accA[NT], accB[NT];
    float ssum[4 * R4], ssq[4 * R4];
#pragma unroll
    for (int r = 0; r < 4 * R4; ++r) { ssum[r] = 0.0f; ssq[r] = 0.0f; }
    const sv16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // One staged frame against the temporal taps KA (accumulated into C) and KB (into Nx); -1: none.  A tap-0 product OPENS its
    // accumulator (C operand of the first MFMA = 0): accumulators are never zeroed or moved.
#define SAL_MMA(KA, KB, C, Nx)                                                                                           \
    do {                                                                                                                  \
        _Pragma("unroll") for (int s = 0; s < 27; ++s) {                                                                  \
            const int c2 = s / 9, kh = (s - c2 * 9) / 3, kw = s - c2 * 9 - kh * 3;                                        \
            _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                              \
                const float b = lb[((2 * c2) * RIN + kh + 2 * j * TR) * PITCH + kw];                                      \
                if (KA >= 0) C[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[KA < 0 ? 0 : KA][s], b, (KA == 0 && s == 0) ? zero16 : C[j], 0, 0, 0); \
                if (KB >= 0) Nx[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[KB < 0 ? 0 : KB][s], b, (KB == 0 && s == 0) ? zero16 : Nx[j], 0, 0, 0); \
            }                                                                                                             \
        }                                                                                                                 \
    } while (0)

    const int mt = wv / SH, mq = (wv % SH) * R4;                // this wave finishes register groups mq .. mq + R4 - 1 of tile mt
    const int oh = oh0 + mt * TR + pr;                          // output row this lane finishes
    const bool ov = pv && oh < Ho;
    const long PO = (long)Ho * WO;

    // output frame to: even frame 2 to (tap 1), odd frame 2 to + 1 (tap 2; tap 0 of frame to + 1 into Nx), reduction, epilogue
    auto step = [&](sv16 (&C)[NT], sv16 (&Nx)[NT], int i) __attribute__((always_inline)) {
        const int to = to0 + i, fo = 2 * to + 1;
        const bool more = i + 1 < nto;
        sal_wave_sync();
        stage();                                                // even frame (always inside the clip)
        if (fo < T) fetch(fo);
        sal_wave_sync();
        SAL_MMA(1, -1, C, Nx);
        if (fo < T) {
            sal_wave_sync();
            stage();
            if (more) fetch(fo + 1);
            sal_wave_sync();
            if (more) SAL_MMA(2, 0, C, Nx); else SAL_MMA(2, -1, C, Nx);
        }                                                       // fo >= T: to is the clip's last output frame
        // ---- the four channel-slice partials of every tile meet in LDS (fixed order: bit-repeatable) ------------------
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                red[((j * 4 + wv) * 4 + q) * 64 + lane] = (sv4){C[j][4 * q], C[j][4 * q + 1], C[j][4 * q + 2], C[j][4 * q + 3]};
        __syncthreads();
        sv4 tot[R4];
#pragma unroll
        for (int q = 0; q < R4; ++q) {
            const sv4* src = red + ((mt * 4) * 4 + mq + q) * 64 + lane;
            tot[q] = ((src[0] + src[4 * 64]) + src[2 * 4 * 64]) + src[3 * 4 * 64];
        }
        __syncthreads();
        float* yp = a.y + (((long)n * Cout) * To + to) * PO + (long)(oh0 + mt * TR) * WO + p;
#pragma unroll
        for (int q = 0; q < R4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = e + 8 * (mq + q) + 4 * half;
                const float v = tot[q][e];
                if (ov && co < Cout) {
                    yp[(long)co * To * PO] = v;
                    ssum[4 * q + e] += v; ssq[4 * q + e] = fmaf(v, v, ssq[4 * q + e]);
                }
            }
    };

    {   // halo frame 2 to0 - 1: tap 0 of the chunk's first output frame
        const int f = 2 * to0 - 1;
        if (f >= 0) { fetch(f); stage(); }
        fetch(2 * to0);
        sal_wave_sync();
        if (f >= 0) SAL_MMA(0, -1, accA, accB);
        else {
#pragma unroll
            for (int j = 0; j < NT; ++j) accA[j] = zero16;
        }
    }
    int i = 0;
    for (; i + 1 < nto; i += 2) { step(accA, accB, i); step(accB, accA, i + 1); }
    if (i < nto) step(accA, accB, i);
#undef SAL_MMA
    if (a.s1) {
#pragma unroll
        for (int r = 0; r < 4 * R4; ++r) {
            float s = ssum[r], q = ssq[r];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
            const int co = (r & 3) + 8 * (mq + (r >> 2)) + 4 * half;
            if (p == 0 && co < Cout) {
                cfn_add64(&a.s1[(long)n * Cout + co], (double)s);
                cfn_add64(&a.s2[(long)n * Cout + co], (double)q);
            }
        }
    }
}

// environment switches are read ONCE per process (getenv on every launch); INT_MIN = not set
static int sal_env_raw(const char* name) { const char* e = getenv(name); return e ? atoi(e) : -2147483647 - 1; }
#define SAL_ENV(name, dflt) ([&]() { static const int v_ = sal_env_raw(name); return v_ == -2147483647 - 1 ? (dflt) : v_; }())

// -1 = shape not handled (the caller runs the implicit GEMM)
int sal_fwd_try_launch(const float* x, const double* A, const double* B, int act, const float* w, float* y, double* sum,
                       double* sumsq, int N, int Cin, int Cout, int T, int Hi, int Wi, const int* g, hipStream_t st) {
    static const int want[9] = {3, 3, 3, 2, 2, 2, 1, 1, 1};
    for (int i = 0; i < 9; ++i) if (g[i] != want[i]) return -1;
    if (Cin != SAL_CIN || Cout > 32 || (Wi != 56 && Wi != 28) || (Hi & 1) || Hi < 2 || ((uintptr_t)x & 15)) return -1;
    if (A && act != CFN_ACT_RELU) return -1;
    if (!A && act != CFN_ACT_NONE) return -1;
    if (SAL_ENV("CFN_SAL_OFF", 0)) return -1;
    if ((long)SAL_CW * T * Hi * Wi * 4 >= 0x7fff0000L) return -1;
    SalArgs a = {x, A, B, w, y, sum, sumsq, N, Cout, T, (T - 1) / 2 + 1, Hi, Hi / 2};
    const int NT = SAL_ENV("CFN_SAL_NT", 1);
    if (NT != 1 && NT != 4) return -1;
    const int WO = Wi / 2, TR = 32 / WO, RB = NT * TR, RIN = 2 * RB + 1, PITCH = Wi + 4;
    a.bands = cfn_cdiv(a.Ho, RB);
    // chunk length: whole rounds of one workgroup per CU where possible (a 1.25-round grid costs a full second round)
    int cus = 256;
    const int per_cu = NT == 4 ? 1 : 2;              // resident workgroups per CU (LDS / registers)
    { int dev = 0; hipDeviceProp_t pr; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) cus = pr.multiProcessorCount; }
    int best = 4; double bestc = 1e30;
    for (int to = 2; to <= 16; ++to) {
        const long blocks = (long)N * a.bands * cfn_cdiv(a.To, to);
        const double rounds = (double)cfn_cdiv(blocks, (long)cus * per_cu);
        const double cost = rounds * (3.0 * to + 1.0);          // MFMA sets per block: 3 per output frame + the halo frame's one
        if (cost < bestc - 1e-9) { bestc = cost; best = to; }
    }
    a.TO = SAL_ENV("CFN_SAL_TO", best);
    if (a.TO < 1) a.TO = 1;
    a.nchunks = cfn_cdiv(a.To, a.TO);
    const long blocks = (long)N * a.bands * a.nchunks;
    if (blocks >= (1L << 31)) return -1;
    const size_t lds = ((size_t)4 * SAL_CW * RIN * PITCH + (size_t)NT * 4 * 16 * 64) * sizeof(float);
#define SAL_GO(WIV, NTV, PROV)                                                                                         \
    do {                                                                                                               \
        auto k = sal_fwd_kernel<WIV, NTV, PROV>;                                                                       \
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
        hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(256), lds, st, a);                                          \
    } while (0)
#define SAL_GO2(WIV, NTV) do { if (A) SAL_GO(WIV, NTV, true); else SAL_GO(WIV, NTV, false); } while (0)
#define SAL_GO3(WIV) do { if (NT == 1) SAL_GO2(WIV, 1); else SAL_GO2(WIV, 4); } while (0)
    if (Wi == 56) SAL_GO3(56); else SAL_GO3(28);
#undef SAL_GO3
#undef SAL_GO2
#undef SAL_GO
    return cfn_check_launch("sal_conv_fwd");
}


// ---------------------------------------------------------------------------------------------------------------------------
// Data gradient:  gx[ci][it][ih][iw] = act'(A x + B) A  sum_{co, taps hitting (it, ih, iw)} W[co][ci][kt][kh][kw] g'[co][to][oh][ow],
// g' = gy + gs[n,co] + 2 y gq[n,co];  gA += sum dz x,  gB += sum dz  (conventions of cfn_conv3d_dense_bwd_data).
//
// Gather form by stride-parity class (it, ih, iw) = (2u + pt, 2a + ph, 2c + pw): class (pt, ph, pw) sees nt(pt) nh(ph) nw(pw) taps
// (1 for an even coordinate: tap 1; 2 for an odd one: taps 0 and 2, tap 0 reaching the NEXT output index).  Per class an implicit
// GEMM  M = ci, N = 32 positions (one a-row x 28 c), K = (co, taps)  on v_mfma_f32_32x32x2_f32:
//   * a workgroup = 8 waves = 2 independent groups (different (sample, band, t-chunk)); the four waves of a group own the four
//     tile rows of a band and each computes ALL eight classes of its row -- full K per wave, so no cross-wave reduction, no
//     atomics on gx, results bit-repeatable; two waves per SIMD: one wave's epilogue runs under the other's MFMAs;
//   * the whole weight tensor sits in LDS once per workgroup as [tap][co pair][half][ci] (MFMA A operand = one ds_read_b32 at lane
//     base + immediate), g' of the group's band as [co][slot][row][col] with a zero row / column / frame behind the last one
//     (B operand, one read feeds the pw = 0 and pw = 1 classes); two slots hold output frames u and u + 1, frame u + 2 is
//     fetched during step u and takes frame u's slot behind a barrier;
//   * the pw = 0 / 1 classes of a tile leave together as 8-byte stores: every instruction writes whole 224-byte rows of gx.
// ---------------------------------------------------------------------------------------------------------------------------
struct SalBwdArgs {
    const float* gy; const float* y; const double* gs; const double* gq; const float* w; const float* x;
    const double* pa; const double* pb; float* gx; double* gA; double* gB; double* gw;
    int N, T, To, Hi, Ho, bands, nchunks, CH, ngroups;
};

typedef float __attribute__((ext_vector_type(2))) sv2;

// MFMAs of the class pair (PT, PH, pw = 0 | 1) of one tile: acc0 (pw = 0) and acc1 (pw = 1) are opened here
template <int GROWS, int GP, int PT, int PH, int SL>
__device__ __forceinline__ void sal_dg_pair(const float* lbA, const float* lbB, sv16& acc0, sv16& acc1) {
    const sv16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bool open = true;
#pragma unroll
    for (int it = 0; it < (PT ? 2 : 1); ++it) {
        const int kt = PT ? 2 * it : 1, dt = (PT && kt == 0) ? 1 : 0;
#pragma unroll
        for (int ih = 0; ih < (PH ? 2 : 1); ++ih) {
            const int kh = PH ? 2 * ih : 1, dh = (PH && kh == 0) ? 1 : 0;
#pragma unroll
            for (int c2 = 0; c2 < 12; ++c2) {
                const float* bp = lbB + ((4 * c2 + (SL ^ dt)) * GROWS + dh) * GP;
                const float b0 = bp[0], b1 = bp[1];
                const float w0 = lbA[((((kt * 3 + kh) * 3 + 0) * 12 + c2) * 2) * 24];
                const float w1 = lbA[((((kt * 3 + kh) * 3 + 1) * 12 + c2) * 2) * 24];
                const float w2 = lbA[((((kt * 3 + kh) * 3 + 2) * 12 + c2) * 2) * 24];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, b0, open ? zero16 : acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w2, b0, open ? zero16 : acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, b1, acc1, 0, 0, 0);
                open = false;
            }
        }
    }
}

template <int WI, bool PRO, bool HASY>
__global__ __launch_bounds__(512, 2) void sal_dgrad_kernel(const SalBwdArgs a) {
    constexpr int WO = WI / 2, WO2 = WO / 2, TRA = 32 / WO, BA = 4 * TRA, GROWS = BA + 1, GP = (WO + 4) & ~3;
    constexpr int GIMG = SAL_CIN * 2 * GROWS * GP, WL = 648 * 24, OOB = 0x7fff0000;
    constexpr int SU = SAL_CIN * GROWS * WO2, NSU = (SU + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];          // wl | gimg[2 groups] | gs, 2 gq [2 groups][2][24]
    float* wl = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = cfn_uni(tid >> 6), grp = wave >> 2, wg = wave & 3, gtid = tid & 255;
    const int h = lane >> 5, p = lane & 31;
    float* gimg = smem + WL + grp * GIMG;
    float* sgs = smem + WL + 2 * GIMG + grp * 48;
    const int T = a.T, To = a.To, Hi = a.Hi, Ho = a.Ho;
    const unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    int gidx = cfn_uni((int)(2 * L + grp));
    const bool active = gidx < a.ngroups;
    if (!active) gidx = 0;
    const int band = gidx % a.bands, chunk = (gidx / a.bands) % a.nchunks, n = gidx / (a.bands * a.nchunks);
    const int NU = (T + 1) / 2;                                  // input frame pairs
    const int u0 = chunk * a.CH, u1 = min(u0 + a.CH, NU);       // steps u0 .. u1 - 1 need g' frames u0 .. u1
    const int a0 = band * BA;
    const long P = (long)Hi * WI, PO = (long)Ho * WO;

    for (int i = tid; i < 2 * GIMG; i += 512) smem[WL + i] = 0.0f;
    for (int e = tid; e < WL; e += 512) {                       // w[co][ci][tap] -> wl[tap][co >> 1][co & 1][ci]
        const int tap = e % 27, r = e / 27, ci = r % 24, co = r / 24;
        wl[((tap * 12 + (co >> 1)) * 2 + (co & 1)) * 24 + ci] = a.w[e];
    }
    if (gtid < 24) {
        sgs[gtid] = a.gs ? (float)a.gs[(long)n * 24 + gtid] : 0.0f;
        sgs[24 + gtid] = (HASY && a.gq) ? 2.0f * (float)a.gq[(long)n * 24 + gtid] : 0.0f;
    }
    // g' staging units of this thread: float2 e of the band's (24 channels x GROWS rows x WO columns) slab of one output frame
    int goff[NSU], loff[NSU], gco[NSU];
#pragma unroll
    for (int k = 0; k < NSU; ++k) {
        const int e = k * 256 + gtid;
        const int co = e / (GROWS * WO2), rem = e - co * (GROWS * WO2), r = rem / WO2, c2i = rem - r * WO2;
        const bool ok = e < SU && a0 + r < Ho;
        goff[k] = ok ? (int)((((long)co * To * Ho + (a0 + r)) * WO + 2 * c2i) * 4) : OOB;
        loff[k] = (co * 2 * GROWS + r) * GP + 2 * c2i;
        gco[k] = ok ? co : 0;
    }
    __amdgpu_buffer_rsrc_t rgy = cfn_rsrc(a.gy + (long)n * 24 * To * PO, (unsigned)((long)24 * To * PO * 4));
    __amdgpu_buffer_rsrc_t ry = cfn_rsrc((HASY ? a.y : a.gy) + (long)n * 24 * To * PO, (unsigned)((long)24 * To * PO * 4));
    sv2 fg[NSU], fy[HASY ? NSU : 1];
    auto fetch = [&](int to) {                                  // to < To
        const int so = cfn_uni((int)(to * PO * 4));
#pragma unroll
        for (int k = 0; k < NSU; ++k) {
            fg[k] = __builtin_bit_cast(sv2, __builtin_amdgcn_raw_buffer_load_b64(rgy, goff[k], so, 0));
            if (HASY) fy[k] = __builtin_bit_cast(sv2, __builtin_amdgcn_raw_buffer_load_b64(ry, goff[k], so, 0));
        }
    };
    auto put = [&](int slot, bool real) {                       // real = false: the frame behind the clip's last one is zero
#pragma unroll
        for (int k = 0; k < NSU; ++k) {
            if (goff[k] != OOB) {
                sv2 v = {0.0f, 0.0f};
                if (real) {
                    const float s0 = sgs[gco[k]];
                    v = fg[k] + s0;
                    if (HASY) { const float q0 = sgs[24 + gco[k]]; v.x = fmaf(fy[k].x, q0, v.x); v.y = fmaf(fy[k].y, q0, v.y); }
                }
                *reinterpret_cast<sv2*>(gimg + loff[k] + slot * GROWS * GP) = v;
            }
        }
    };
    __syncthreads();                                            // zero fill, weights, gs / gq visible
    if (active) {
        if (u0 < To) fetch(u0);
        put(0, u0 < To);
        if (u0 + 1 < To) fetch(u0 + 1);
        put(1, u0 + 1 < To);
    }
    __syncthreads();

    // lane constants
    const bool pv = p < TRA * WO;
    const int ar = pv ? p / WO : 0, c = pv ? p - ar * WO : 0;
    const int arow = a0 + wg * TRA + ar;                        // a-row (output row index) of this lane's positions
    const bool okl = active && pv && arow < Ho;
    const float* lbA = wl + h * 24 + min(p, 23);
    const float* lbB = gimg + (h * 2 * GROWS + wg * TRA + ar) * GP + c;
    float ca[12], cb[12], sA[12], sB[12];
#pragma unroll
    for (int r = 0; r < 12; ++r) {
        const int ci = (r & 3) + 8 * (r >> 2) + 4 * h;
        ca[r] = PRO ? (float)a.pa[(long)n * 24 + ci] : 1.0f;
        cb[r] = PRO ? (float)a.pb[(long)n * 24 + ci] : 0.0f;
        sA[r] = 0.0f; sB[r] = 0.0f;
    }
    __amdgpu_buffer_rsrc_t rgx = cfn_rsrc(a.gx + (long)n * 24 * T * P, (unsigned)((long)24 * T * P * 4));
    __amdgpu_buffer_rsrc_t rxx = cfn_rsrc((PRO ? a.x : a.gx) + (long)n * 24 * T * P, (unsigned)((long)24 * T * P * 4));
    const int chs = cfn_uni((int)(T * P * 4));                  // channel stride in bytes

    auto finish = [&](int it, int ph, const sv2 (&xq)[12], const sv16& acc0, const sv16& acc1, int vo) {
#pragma unroll
        for (int r = 0; r < 12; ++r) {
            const int so = cfn_uni(((r & 3) + 8 * (r >> 2)) * chs);
            sv2 o = {acc0[r], acc1[r]};
            if (PRO) {
                const sv2 xv = xq[r];
                const float d0 = (okl && fmaf(xv.x, ca[r], cb[r]) > 0.0f) ? o.x : 0.0f, d1 = (okl && fmaf(xv.y, ca[r], cb[r]) > 0.0f) ? o.y : 0.0f;   // lanes without a position must not reach the sums
                sA[r] = fmaf(d0, xv.x, fmaf(d1, xv.y, sA[r]));
                sB[r] += d0 + d1;
                o = (sv2){d0 * ca[r], d1 * ca[r]};
            }
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(cp_u2_t, o), rgx, vo, so, 0);
        }
    };
#define SAL_DG_CLASS(PTV, PHV, SLV)                                                                                      \
    do {                                                                                                                  \
        const int it = 2 * u + PTV;                                                                                       \
        if (it < T) {                                                                                                     \
            const int vo = okl ? (int)(((long)4 * h * T * P + ((long)it * Hi + 2 * arow + PHV) * WI + 2 * c) * 4) : OOB;  \
            sv2 xq[12];                                                                                                   \
            if (PRO) {                                                                                                    \
                _Pragma("unroll") for (int r = 0; r < 12; ++r)                                                            \
                    xq[r] = __builtin_bit_cast(sv2, __builtin_amdgcn_raw_buffer_load_b64(rxx, vo, cfn_uni(((r & 3) + 8 * (r >> 2)) * chs), 0)); \
            }                                                                                                             \
            sv16 acc0, acc1;                                                                                              \
            sal_dg_pair<GROWS, GP, PTV, PHV, SLV>(lbA, lbB, acc0, acc1);                                                  \
            finish(it, PHV, xq, acc0, acc1, vo);                                                                          \
        }                                                                                                                 \
    } while (0)
#define SAL_DG_STEP(SLV)                                                                                                 \
    do {                                                                                                                  \
        const bool nxt = active && u + 1 < u1;                  /* a further step needs g' frame u + 2 */                 \
        if (nxt && u + 2 < To) fetch(u + 2);                                                                              \
        if (active && u < u1) {                                                                                           \
            SAL_DG_CLASS(0, 0, SLV); SAL_DG_CLASS(0, 1, SLV); SAL_DG_CLASS(1, 0, SLV); SAL_DG_CLASS(1, 1, SLV);           \
        }                                                                                                                 \
        __syncthreads();                                                                                                  \
        if (nxt) put(SLV, u + 2 < To);                                                                                    \
        __syncthreads();                                                                                                  \
    } while (0)
    for (int s = 0; s < a.CH; s += 2) {
        int u = u0 + s;
        SAL_DG_STEP(0);
        u = u0 + s + 1;
        SAL_DG_STEP(1);
    }
#undef SAL_DG_STEP
#undef SAL_DG_CLASS
    if (PRO && active) {
#pragma unroll
        for (int r = 0; r < 12; ++r) {
            float s = sA[r], q = sB[r];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
            const int ci = (r & 3) + 8 * (r >> 2) + 4 * h;
            if (p == 0) {
                cfn_add64(&a.gA[(long)n * 24 + ci], (double)s);
                cfn_add64(&a.gB[(long)n * 24 + ci], (double)q);
            }
        }
    }
}

// -1 = shape not handled
int sal_dgrad_try_launch(const float* gy, const float* y, const double* gs, const double* gq, const float* w, const float* x,
                         const double* A, const double* B, int act, float* gx, double* gA, double* gB, int N, int Cin, int Cout,
                         int T, int Hi, int Wi, const int* g, hipStream_t st) {
    static const int want[9] = {3, 3, 3, 2, 2, 2, 1, 1, 1};
    for (int i = 0; i < 9; ++i) if (g[i] != want[i]) return -1;
    if (Cin != SAL_CIN || Cout != 24 || (Wi != 56 && Wi != 28) || (Hi & 1) || Hi < 2) return -1;
    if (A && act != CFN_ACT_RELU) return -1;
    if (!A && act != CFN_ACT_NONE) return -1;
    if (SAL_ENV("CFN_SAL_OFF", 0) || SAL_ENV("CFN_SAL_DGRAD_OFF", 0)) return -1;
    if ((long)24 * T * Hi * Wi * 4 >= 0x7fff0000L) return -1;
    SalBwdArgs a = {};
    a.gy = gy; a.y = gq ? y : nullptr; a.gs = gs; a.gq = gq; a.w = w; a.x = x; a.pa = A; a.pb = B; a.gx = gx; a.gA = gA; a.gB = gB;
    a.N = N; a.T = T; a.To = (T - 1) / 2 + 1; a.Hi = Hi; a.Ho = Hi / 2;
    const int WO = Wi / 2, TRA = 32 / WO, BA = 4 * TRA, GROWS = BA + 1, GP = (WO + 4) & ~3;
    a.bands = cfn_cdiv(a.Ho, BA);
    const int NU = (T + 1) / 2;
    int cus = 256;
    { int dev = 0; hipDeviceProp_t pr; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) cus = pr.multiProcessorCount; }
    // chunk length (even): whole rounds of 2 groups per CU, the weight image (62 KB per workgroup) amortised over >= 8 steps where the clip allows
    int best = 2; double bestc = 1e30;
    for (int ch = 2; ch <= 64; ch += 2) {
        const long groups = (long)N * a.bands * cfn_cdiv(NU, ch);
        const double rounds = (double)cfn_cdiv(groups, 2L * cus);
        const double cost = rounds * (ch + 1.5);                // + the workgroup's start-up (weights, two frames) in steps
        if (cost < bestc - 1e-9) { bestc = cost; best = ch; }
    }
    a.CH = SAL_ENV("CFN_SAL_DG_CH", best);
    if (a.CH < 2) a.CH = 2;
    a.CH &= ~1;
    a.nchunks = cfn_cdiv(NU, a.CH);
    const long groups = (long)N * a.bands * a.nchunks;
    if (groups >= (1L << 30)) return -1;
    a.ngroups = (int)groups;
    const long blocks = (groups + 1) / 2;
    const size_t lds = ((size_t)648 * 24 + (size_t)2 * SAL_CIN * 2 * GROWS * GP + 96) * sizeof(float);
#define SAL_DG(WIV, PROV, YV)                                                                                          \
    do {                                                                                                               \
        auto k = sal_dgrad_kernel<WIV, PROV, YV>;                                                                      \
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
        hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(512), lds, st, a);                                          \
    } while (0)
#define SAL_DG2(WIV) do { if (A) { if (a.y) SAL_DG(WIV, true, true); else SAL_DG(WIV, true, false); }                  \
                          else { if (a.y) SAL_DG(WIV, false, true); else SAL_DG(WIV, false, false); } } while (0)
    if (Wi == 56) SAL_DG2(56); else SAL_DG2(28);
#undef SAL_DG2
#undef SAL_DG
    return cfn_check_launch("sal_conv_dgrad");
}


// ---------------------------------------------------------------------------------------------------------------------------
// Weight gradient:  gW[co][ci][kt][kh][kw] = sum_{n, to, oh, ow} g'[n][co][to][oh][ow] * a[n][ci][2 to - 1 + kt][2 oh - 1 + kh][2 ow - 1 + kw],
// a = relu(A x + B) (zero padded).  Implicit GEMM  M = co, N = the 648 columns (ci, tap), K = positions  on v_mfma_f32_32x32x2_f32:
//   * a PERSISTENT workgroup (8 waves, two per SIMD) walks over work items (sample, band of RB output rows, chunk of output
//     frames) and keeps its share of gW in registers across all of them: wave w owns the column tiles w, w + 8, w + 16 (21 tiles of
//     32 columns), 48 accumulator registers; one fp64 atomic per element and workgroup at the very end;
//   * per output frame the workgroup stages the two new input frames of its band (prologue applied, zero halo) into a ring of
//     three frame slots (the even frame's slot and the dead odd one) and the band's g' rows as [co][57]; operands of every MFMA:
//     A = g'[co = lane][position pair] (read once per wave, used by its three tiles), B = image at a per-lane tap base
//     + immediate position offset;
//   * loads for step to + 1 are in flight during the MFMAs of step to, written behind a barrier.
// ---------------------------------------------------------------------------------------------------------------------------
template <int WI, bool PRO, bool HASY>
__global__ __launch_bounds__(512, 2) void sal_wgrad_kernel(const SalBwdArgs a) {
    constexpr int WO = WI / 2, RB = 56 / WO, RIN = 2 * RB + 1, PITCH = WI + 4, W4 = WI / 4;
    constexpr int FR = SAL_CIN * RIN * PITCH, KQ = RB * WO, GPB = KQ + 1, OOB = 0x7fff0000;
    constexpr int XU = SAL_CIN * RIN * W4, NXU = (XU + 511) / 512;          // float4 units per frame and thread
    constexpr int GU = SAL_CIN * KQ / 2, NGU = (GU + 511) / 512;            // float2 units of g' per step and thread
    static_assert(KQ == 56 && WO % 2 == 0, "geometry");
    extern __shared__ __attribute__((aligned(16))) float smem[];            // img[3 slots][24][RIN][PITCH] | gbuf[24][57] | coefficient tables
    float* img = smem;
    float* gbuf = smem + 3 * FR;
    float* tab = gbuf + SAL_CIN * GPB;                                      // pa[24] pb[24] gs[24] 2gq[24]
    const int tid = threadIdx.x, lane = tid & 63, wave = cfn_uni(tid >> 6), h = lane >> 5, p = lane & 31;
    const int T = a.T, To = a.To, Hi = a.Hi, Ho = a.Ho;
    const long P = (long)Hi * WI, PO = (long)Ho * WO;

    for (int i = tid; i < 3 * FR; i += 512) img[i] = 0.0f;                  // the halo column stays zero for good

    // B operand bases of this wave's column tiles: column = (ci, kt, kh, kw); two slot patterns (parity of the output frame)
    int bb[2][3];
#pragma unroll
    for (int tt = 0; tt < 3; ++tt) {
        int col = (wave + 8 * tt) * 32 + p;
        if (col >= 648) col = 0;
        const int ci = col / 27, tap = col - ci * 27, kt = tap / 9, kh = (tap - kt * 9) / 3, kw = tap - kt * 9 - kh * 3;
        const int off = (ci * RIN + kh) * PITCH + kw + 3 + 2 * h;
        // frames 2 to - 1, 2 to, 2 to + 1 sit in slots (even to) 2, 0, 1 / (odd to) 1, 0, 2
        bb[0][tt] = (kt == 0 ? 2 : kt == 1 ? 0 : 1) * FR + off;
        bb[1][tt] = (kt == 0 ? 1 : kt == 1 ? 0 : 2) * FR + off;
    }
    const int ab = min(p, 23) * GPB + h;
    const bool t2 = wave + 16 < 21;                                         // waves 5-7 own two tiles only

    sv16 acc[3];
#pragma unroll
    for (int tt = 0; tt < 3; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tt][r] = 0.0f;

    const int items = a.ngroups;
    for (int item = blockIdx.x; item < items; item += gridDim.x) {
        const int band = item % a.bands, chunk = (item / a.bands) % a.nchunks, n = item / (a.bands * a.nchunks);
        const int to0 = chunk * a.CH, nst = min(a.CH, To - to0);
        const int oh0 = band * RB, ih0 = 2 * oh0 - 1;
        __syncthreads();                                                    // the previous item's reads are done
        if (tid < 24) {
            tab[tid] = PRO ? (float)a.pa[(long)n * 24 + tid] : 1.0f;
            tab[24 + tid] = PRO ? (float)a.pb[(long)n * 24 + tid] : 0.0f;
            tab[48 + tid] = a.gs ? (float)a.gs[(long)n * 24 + tid] : 0.0f;
            tab[72 + tid] = (HASY && a.gq) ? 2.0f * (float)a.gq[(long)n * 24 + tid] : 0.0f;
        }
        // staging units of this thread
        int xo[NXU], xl[NXU], xc[NXU];
#pragma unroll
        for (int k = 0; k < NXU; ++k) {
            const int e = k * 512 + tid;
            const int rowid = e / W4, c4 = e - rowid * W4, ci = rowid / RIN, r = rowid - ci * RIN, ih = ih0 + r;
            const bool in = e < XU;
            xo[k] = (in && ih >= 0 && ih < Hi) ? (int)((((long)ci * T * Hi + ih) * WI + c4 * 4) * 4) : OOB;
            xl[k] = in ? (ci * RIN + r) * PITCH + 4 + c4 * 4 : -1;
            xc[k] = in ? ci : 0;
        }
        int go[NGU], gl[NGU], gc[NGU];
#pragma unroll
        for (int k = 0; k < NGU; ++k) {
            const int e = k * 512 + tid;
            const int co = e / (KQ / 2), q2 = e - co * (KQ / 2);
            const bool in = e < GU;
            const int row = (2 * q2) / WO;
            go[k] = (in && oh0 + row < Ho) ? (int)((((long)co * To * Ho + oh0) * WO + 2 * q2) * 4) : OOB;
            gl[k] = in ? co * GPB + 2 * q2 : -1;
            gc[k] = in ? co : 0;
        }
        __amdgpu_buffer_rsrc_t rx = cfn_rsrc(a.x + (long)n * 24 * T * P, (unsigned)((long)24 * T * P * 4));
        __amdgpu_buffer_rsrc_t rgy = cfn_rsrc(a.gy + (long)n * 24 * To * PO, (unsigned)((long)24 * To * PO * 4));
        __amdgpu_buffer_rsrc_t ry = cfn_rsrc((HASY ? a.y : a.gy) + (long)n * 24 * To * PO, (unsigned)((long)24 * To * PO * 4));
        sv4 fx[2][NXU];
        sv2 fg[NGU], fy[HASY ? NGU : 1];
        auto fetch_x = [&](int which, int f) {                              // frames outside the clip read nothing
            const bool want = f >= 0 && f < T;
            const int so = cfn_uni(want ? (int)(f * P * 4) : 0);
#pragma unroll
            for (int k = 0; k < NXU; ++k) fx[which][k] = __builtin_bit_cast(sv4, __builtin_amdgcn_raw_buffer_load_b128(rx, want ? xo[k] : OOB, so, 0));
        };
        auto put_x = [&](int which, int slot, int f) {                      // every unit is written: rows / frames outside the clip are zero AFTER the prologue
            const bool fin = f >= 0 && f < T;
#pragma unroll
            for (int k = 0; k < NXU; ++k) {
                if (xl[k] >= 0) {
                    sv4 v = fx[which][k];
                    if (PRO) {
                        const float ua = tab[xc[k]], ub = tab[24 + xc[k]];
                        v.x = fmaxf(fmaf(v.x, ua, ub), 0.0f); v.y = fmaxf(fmaf(v.y, ua, ub), 0.0f);
                        v.z = fmaxf(fmaf(v.z, ua, ub), 0.0f); v.w = fmaxf(fmaf(v.w, ua, ub), 0.0f);
                    }
                    if (!fin || xo[k] == OOB) v = (sv4){0.f, 0.f, 0.f, 0.f};
                    *reinterpret_cast<sv4*>(img + slot * FR + xl[k]) = v;
                }
            }
        };
        auto fetch_g = [&](int to) {
            const int so = cfn_uni((int)(to * PO * 4));
#pragma unroll
            for (int k = 0; k < NGU; ++k) {
                fg[k] = __builtin_bit_cast(sv2, __builtin_amdgcn_raw_buffer_load_b64(rgy, go[k], so, 0));
                if (HASY) fy[k] = __builtin_bit_cast(sv2, __builtin_amdgcn_raw_buffer_load_b64(ry, go[k], so, 0));
            }
        };
        auto put_g = [&]() {
#pragma unroll
            for (int k = 0; k < NGU; ++k) {
                if (gl[k] >= 0) {
                    sv2 v = {0.0f, 0.0f};
                    if (go[k] != OOB) {
                        v = fg[k] + tab[48 + gc[k]];
                        if (HASY) { const float q0 = tab[72 + gc[k]]; v.x = fmaf(fy[k].x, q0, v.x); v.y = fmaf(fy[k].y, q0, v.y); }
                    }
                    gbuf[gl[k]] = v.x; gbuf[gl[k] + 1] = v.y;
                }
            }
        };
        __syncthreads();                                                    // coefficient tables visible
        // the chunk's first three frames (to0 is even: frame 2 to0 - 1 -> slot 2, 2 to0 -> slot 0, 2 to0 + 1 -> slot 1) and g'(to0)
        fetch_x(0, 2 * to0 - 1); fetch_x(1, 2 * to0);
        fetch_g(to0);
        put_x(0, 2, 2 * to0 - 1); put_x(1, 0, 2 * to0);
        fetch_x(0, 2 * to0 + 1);
        put_g();
        put_x(0, 1, 2 * to0 + 1);
        __syncthreads();
#define SAL_WG_STEP(PV)                                                                                                  \
        do {                                                                                                              \
            const int to = to0 + s + PV;                                                                                  \
            if (s + PV < nst) {                                             /* uniform over the workgroup */              \
                const bool more = s + PV + 1 < nst;                                                                       \
                if (more) { fetch_x(0, 2 * to + 2); fetch_x(1, 2 * to + 3); fetch_g(to + 1); }                            \
                _Pragma("unroll") for (int j = 0; j < KQ / 2; ++j) {                                                      \
                    const int q = 2 * j, r = q / WO, o = (2 * r) * PITCH + 2 * (q - r * WO);                              \
                    const float av = gbuf[ab + 2 * j];                                                                    \
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, img[bb[PV][0] + o], acc[0], 0, 0, 0);               \
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, img[bb[PV][1] + o], acc[1], 0, 0, 0);               \
                    if (t2) acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, img[bb[PV][2] + o], acc[2], 0, 0, 0);       \
                }                                                                                                         \
                __syncthreads();                                                                                          \
                if (more) { put_x(0, 0, 2 * to + 2); put_x(1, PV ? 1 : 2, 2 * to + 3); put_g(); }                         \
                __syncthreads();                                                                                          \
            }                                                                                                             \
        } while (0)
        for (int s = 0; s < nst; s += 2) { SAL_WG_STEP(0); SAL_WG_STEP(1); }
#undef SAL_WG_STEP
    }
    // one fp64 atomic per element and workgroup
#pragma unroll
    for (int tt = 0; tt < 3; ++tt) {
        const int col = (wave + 8 * tt) * 32 + p;
        if (wave + 8 * tt < 21 && col < 648) {
#pragma unroll
            for (int r = 0; r < 12; ++r) {
                const int co = (r & 3) + 8 * (r >> 2) + 4 * h;
                cfn_add64(&a.gw[(long)co * 648 + col], (double)acc[tt][r]);
            }
        }
    }
}

// -1 = shape not handled
int sal_wgrad_try_launch(const float* gy, const float* y, const double* gs, const double* gq, const float* x, const double* A,
                         const double* B, int act, double* gw, int N, int Cin, int Cout, int T, int Hi, int Wi, const int* g,
                         hipStream_t st) {
    static const int want[9] = {3, 3, 3, 2, 2, 2, 1, 1, 1};
    for (int i = 0; i < 9; ++i) if (g[i] != want[i]) return -1;
    if (Cin != SAL_CIN || Cout != 24 || (Wi != 56 && Wi != 28) || (Hi & 1) || Hi < 2 || ((uintptr_t)x & 15)) return -1;
    if (A && act != CFN_ACT_RELU) return -1;
    if (!A && act != CFN_ACT_NONE) return -1;
    if (SAL_ENV("CFN_SAL_OFF", 0) || SAL_ENV("CFN_SAL_WGRAD_OFF", 0)) return -1;
    if ((long)24 * T * Hi * Wi * 4 >= 0x7fff0000L) return -1;
    SalBwdArgs a = {};
    a.gy = gy; a.y = gq ? y : nullptr; a.gs = gs; a.gq = gq; a.x = x; a.pa = A; a.pb = B; a.gw = gw;
    a.N = N; a.T = T; a.To = (T - 1) / 2 + 1; a.Hi = Hi; a.Ho = Hi / 2;
    const int WO = Wi / 2, RB = 56 / WO, RIN = 2 * RB + 1, PITCH = Wi + 4;
    a.bands = cfn_cdiv(a.Ho, RB);
    int cus = 256;
    { int dev = 0; hipDeviceProp_t pr; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) cus = pr.multiProcessorCount; }
    // chunk length (even): items spread evenly over one persistent workgroup per CU; a chunk pays one extra (halo) frame
    int best = 2; double bestc = 1e30;
    for (int ch = 2; ch <= 64; ch += 2) {
        const long items = (long)N * a.bands * cfn_cdiv(a.To, ch);
        const double per = (double)cfn_cdiv(items, cus);
        const double cost = per * (ch + 0.75);
        if (cost < bestc - 1e-9) { bestc = cost; best = ch; }
    }
    a.CH = SAL_ENV("CFN_SAL_WG_CH", best);
    if (a.CH < 2) a.CH = 2;
    a.CH &= ~1;
    a.nchunks = cfn_cdiv(a.To, a.CH);
    const long items = (long)N * a.bands * a.nchunks;
    if (items >= (1L << 30)) return -1;
    a.ngroups = (int)items;
    const long blocks = items < cus ? items : cus;
    const size_t lds = ((size_t)3 * SAL_CIN * RIN * PITCH + (size_t)SAL_CIN * 57 + 96) * sizeof(float);
#define SAL_WG(WIV, PROV, YV)                                                                                          \
    do {                                                                                                               \
        auto k = sal_wgrad_kernel<WIV, PROV, YV>;                                                                      \
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
        hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(512), lds, st, a);                                          \
    } while (0)
#define SAL_WG2(WIV) do { if (A) { if (a.y) SAL_WG(WIV, true, true); else SAL_WG(WIV, true, false); }                  \
                          else { if (a.y) SAL_WG(WIV, false, true); else SAL_WG(WIV, false, false); } } while (0)
    if (Wi == 56) SAL_WG2(56); else SAL_WG2(28);
#undef SAL_WG2
#undef SAL_WG
    return cfn_check_launch("sal_conv_wgrad");
}
