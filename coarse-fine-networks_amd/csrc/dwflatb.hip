// Fused backward of the stride-1 depthwise 3x3x3 conv (data gradient AND weight gradient in one pass over gy, y, x), fp32 tensors --
// FLAT kernels (round 4), the backward counterpart of dwflat.hip (x3d_fine.py:89-97,171-201 in reverse; the fusion scheme of DESIGN section 3).
//
//   g'(t)      = gy + gs + 2 y gq                                                  (zero outside the clip)
//   da(t,r,c)  = sum_k w[k] g'(t - kt + 1, r - kh + 1, c - kw + 1)                 a = act(A x + B), the forward input after the prologue
//   gw[k]     += a(t,r,c) g'(t - kt + 1, r - kh + 1, c - kw + 1)                   -- the SAME g' element: one LDS read feeds both products
//   dz = act'(a) da;  gA += dz x;  gB += dz;  gx = A dz
// A work item is one (sample, channel, chunk of TO frames): every lane issues ALL its loads up front (gy and y for TO + 2 frames, x for TO),
// g' goes into an LDS image [TO + 2][rows][W], then the lane walks the g' frames once: per window value and temporal tap one FMA into the data-gradient accumulator of
// an output frame and one into the weight-gradient partial of the flipped tap.  27 weight-gradient partials per lane, transpose-reduced over
// the wave at the end of the item, fp64 atomics.
// dw3d_flat7_bwd_kernel: 7x7 planes (layer 4), a WAVE per item, a lane per position (49 of 64 lanes; the column-pair kernel of dwcpb.hip keeps
// 28 lanes busy on this plane and runs at 2.4 TB/s), no workgroup barrier.  The frames of an item are one contiguous run of floats that does
// not start on a 16-byte boundary: 4-byte loads, lane l takes floats l, l + 64, ...; g' AND x go through LDS (a lane's loads are not its
// position).
// hipcc-flags: -fno-slp-vectorize
// fp32 or bf16 tensors (cp_io.h: compiled a second time through dwflatb_bf16.hip; LDS images, accumulators and every reduction stay fp32 / fp64).
#include "cp_io.h"
#include <stdint.h>
#include <stdlib.h>

#ifdef DW_BF16
#define DwFlatBArgs H16N(DwFlatBArgs)
#endif
struct DwFlatBArgs {
    const cpe_t* gy; const cpe_t* y; const double* gs; const double* gq; const float* w; const cpe_t* x;
    const double* A; const double* B; cpe_t* gx; double* gA; double* gB; double* gw;
    int N, C, T, act, nchunks, subs;     // nchunks: wave items per (sample, channel); a wave item = subs consecutive chunks of TO frames
    long total;
};

typedef float __attribute__((ext_vector_type(4))) fb_f4;
typedef float __attribute__((ext_vector_type(2))) fb_p2;
typedef unsigned __attribute__((ext_vector_type(2))) fb_u2;

// 32 values x 64 lanes -> lane l holds the wave total of value (l >> 1) (32 shuffles; a wave sum per value would be 27 x 6)
__device__ __forceinline__ float fb_transpose_reduce(float (&v)[32], int lane) {
#pragma unroll
    for (int st = 0; st < 5; ++st) {                                     // lane bit 5 - st selects the half of the values it keeps
        const int half = 16 >> st, bit = 32 >> st;
        const bool up = (lane & bit) != 0;
#pragma unroll
        for (int k = 0; k < half; ++k) {
            const float send = up ? v[k] : v[k + half];
            const float keep = up ? v[k + half] : v[k];
            v[k] = keep + __shfl_xor(send, bit, 64);
        }
    }
    return v[0] + __shfl_xor(v[0], 1, 64);
}

template <int TO, bool HASY>
__global__ __launch_bounds__(256, 3) void dw3d_flat7_bwd_kernel(const DwFlatBArgs a) {
    constexpr int W = 7, P = 49, IR = 9, NF = TO + 2, FR = IR * W, OOB = 0x7fff0000;
    constexpr int NG = (NF * P + 63) / 64, NX = (TO * P + 63) / 64;      // 4-byte loads per lane: g' run, x run
    constexpr int WSZ = NF * FR + TO * P + 1;                            // per wave: g' image (zero row above and below every plane) | x | pad
    __shared__ float smem[4 * WSZ];
    const int lane = threadIdx.x & 63, wv = cfn_uni((int)(threadIdx.x >> 6));
    const unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    const long widx = cfn_uni((long)L * 4 + wv);
    if (widx >= a.total) return;                                         // whole waves only: no workgroup barrier below
    const int chunk = cfn_uni((int)(widx % a.nchunks));
    const long nc = cfn_uni((long)(widx / a.nchunks));
    const int c = cfn_uni((int)(nc % a.C));
    const int T = a.T;
    float* img = smem + wv * WSZ;
    float* ximg = img + NF * FR;

    __amdgpu_buffer_rsrc_t rg = cfn_rsrc(a.gy + nc * (long)T * P, (unsigned)((long)T * P * CP_ES));
    __amdgpu_buffer_rsrc_t ry = cfn_rsrc((HASY ? a.y : a.gy) + nc * (long)T * P, (unsigned)((long)T * P * CP_ES));
    __amdgpu_buffer_rsrc_t rx = cfn_rsrc(a.x + nc * (long)T * P, (unsigned)((long)T * P * CP_ES));
    __amdgpu_buffer_rsrc_t rd = cfn_rsrc(a.gx + nc * (long)T * P, (unsigned)((long)T * P * CP_ES));

    float wr[27];                                                        // flipped taps for the data gradient
#pragma unroll
    for (int j = 0; j < 27; ++j) wr[j] = cfn_uni(a.w[(long)c * 27 + 26 - j]);
    const bool hasA = a.A != nullptr;
    const float pa = cfn_uni(hasA ? (float)a.A[nc] : 1.0f);
    const float pb = cfn_uni(hasA ? (float)a.B[nc] : 0.0f);
    const float act_lo = (hasA && a.act == CFN_ACT_RELU) ? 0.0f : -__builtin_inff();   // none / ReLU only (the planner checks)
    const float gsv = cfn_uni(a.gs ? (float)a.gs[nc] : 0.0f);
    const float gqv = cfn_uni((HASY && a.gq) ? 2.0f * (float)a.gq[nc] : 0.0f);
    // zero rows 0 and 8 of every g' frame (never overwritten)
    for (int i = lane; i < NF * 2 * W; i += 64) {
        const int f = i / (2 * W), j = i - f * (2 * W);
        img[f * FR + (j >= W ? (IR - 1) * W + (j - W) : j)] = 0.0f;
    }
    float dwa[27];                                                       // weight-gradient partials: live over all the wave's chunks
#pragma unroll
    for (int j = 0; j < 27; ++j) dwa[j] = 0.0f;
    float st1 = 0.0f, st2 = 0.0f;

    for (int sub = 0; sub < a.subs; ++sub) {
    const int t0 = (chunk * a.subs + sub) * TO;
    if (t0 >= T) break;                                                  // wave uniform
    // run index i = l + 64 m: g' run starts at frame t0 - 1 (floats before the channel's first are not requested; floats behind its last are
    // out of the descriptor's range and read 0), x run at frame t0
    float Rg[NG], Ry[HASY ? NG : 1], Rx[NX];
    const int gstart = (t0 - 1) * P, xstart = t0 * P;
#pragma unroll
    for (int m = 0; m < NG; ++m) {
        const int i = lane + 64 * m;
        const int vo = (i < NF * P && gstart + i >= 0) ? (gstart + i) * CP_ES : OOB;
        Rg[m] = cp_ld1(rg, vo, 0);
        if (HASY) Ry[m] = cp_ld1(ry, vo, 0);
    }
#pragma unroll
    for (int m = 0; m < NX; ++m) {
        const int i = lane + 64 * m;
        Rx[m] = cp_ld1(rx, i < TO * P ? (xstart + i) * CP_ES : OOB, 0);
    }
    __builtin_amdgcn_wave_barrier();                                     // the previous chunk's LDS reads are done (in-order LDS; compiler fence)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int m = 0; m < NG; ++m) {
        const int i = lane + 64 * m;
        if (i < NF * P) {
            const int f = i / P, p = i - f * P, t = t0 - 1 + f;
            float v = Rg[m] + gsv;
            if (HASY) v = fmaf(Ry[m], gqv, v);
            img[f * FR + W + p] = (t >= 0 && t < T) ? v : 0.0f;         // g' is zero outside the clip
        }
    }
#pragma unroll
    for (int m = 0; m < NX; ++m) {
        const int i = lane + 64 * m;
        if (i < TO * P) ximg[i] = Rx[m];
    }
    // LDS operations of a wave run in order; only the compiler has to be told
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    if (lane < P) {
        const int r = lane / W, cc = lane - r * W;
        // window: image rows r .. r + 2 (image row = plane row + 1), columns cc - 1 .. cc + 1; the columns left of 0 / right of 6 are not
        // stored: valid address x 0
        const float* base = img + r * W + cc;
        const int eL = cc == 0 ? 0 : -1, eR = cc == W - 1 ? 0 : 1;
        const float mL = cc == 0 ? 0.0f : 1.0f, mR = cc == W - 1 ? 0.0f : 1.0f;
        float acc[TO], av[TO], xv[TO];
#pragma unroll
        for (int k = 0; k < NF; ++k) {
            if (k < TO) {                                                // output frame k enters the window
                acc[k] = 0.0f;
                xv[k] = ximg[k * P + lane];
                av[k] = t0 + k < T ? fmaxf(fmaf(xv[k], pa, pb), act_lo) : 0.0f;      // a is zero beyond the clip (act(B) is not)
            }
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const float* q = base + k * FR + kh * W;
                const float q0 = q[eL] * mL, q1 = q[0], q2 = q[eR] * mR;
#pragma unroll
                for (int kt = 0; kt < 3; ++kt) {
                    const int j = k - kt;                                // g' frame t0 - 1 + k is tap (kt, kh, .) of the flipped kernel for output frame t0 + j
                    if (j >= 0 && j < TO) {
                        const int tb = kt * 9 + kh * 3;
                        acc[j] = fmaf(wr[tb], q0, fmaf(wr[tb + 1], q1, fmaf(wr[tb + 2], q2, acc[j])));
                        dwa[26 - tb] = fmaf(av[j], q0, dwa[26 - tb]);
                        dwa[25 - tb] = fmaf(av[j], q1, dwa[25 - tb]);
                        dwa[24 - tb] = fmaf(av[j], q2, dwa[24 - tb]);
                    }
                }
            }
            if (k >= 2) {                                                // output frame k - 2 is complete
                const int j = k - 2, t = t0 + j;
                const bool emit = t < T;
                float v = acc[j];
                if (hasA) {                                              // wave uniform
                    const float dz = av[j] > act_lo ? v : 0.0f;          // act' of none / ReLU: a > 0 <=> z > 0
                    const float dm = emit ? dz : 0.0f;
                    st1 = fmaf(dm, xv[j], st1);
                    st2 += dm;
                    v = dz * pa;
                }
                cp_st1(v, rd, emit ? lane * CP_ES : OOB, cfn_uni(emit ? t * P * CP_ES : 0));
            }
        }
    }
    }   // sub
    // ---- reductions: gw (27 per channel), then gA / gB ----
    {
        float v[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = j < 27 ? dwa[j] : 0.0f;
        const float tot = fb_transpose_reduce(v, lane);
        const int idx = lane >> 1;
        if ((lane & 1) == 0 && idx < 27) cfn_add64(&a.gw[(long)c * 27 + idx], (double)tot);
    }
    if (hasA && a.gA) {
        st1 = cfn_wave_sum(st1); st2 = cfn_wave_sum(st2);
        if (lane == 0) { cfn_add64(&a.gA[nc], (double)st1); cfn_add64(&a.gB[nc], (double)st2); }
    }
}

// 14 -> 7 (stride 2, first block of layer 4): a WAVE per item, lane (o, j) < 49 owns output position (o, j) = the 2 x 2 input block rows
// 2o, 2o + 1, columns 2j, 2j + 1 (x and a stay in registers: two 8-byte loads / stores per frame); g' (7x7, 4-byte loads of the contiguous
// run) goes into an LDS image [TO + 2][8][8] with a zero row / column at the bottom / right.  Tap parity: of the 27 taps an even input row /
// column sees only the centre one, an odd one the two outer ones -- 9 (input position, g' element) products per temporal tap, each feeding
// the data gradient AND the weight gradient (dwcpb2.hip has the formulas):
//   (2o, 2j): w[1][1] G[o][j]                          (2o, 2j+1): w[1][0] G[o][j+1] + w[1][2] G[o][j]
//   (2o+1, 2j): w[0][1] G[o+1][j] + w[2][1] G[o][j]    (2o+1, 2j+1): w[0][0] G[o+1][j+1] + w[0][2] G[o+1][j] + w[2][0] G[o][j+1] + w[2][2] G[o][j]
// Before: dw3d_dgrad_s2_fast_kernel + dw3d_kernel<WGRAD> (x read twice), 431 + 303 us per step.
template <int TO, bool HASY>
__global__ __launch_bounds__(256, 3) void dw3d_flat14to7_bwd_kernel(const DwFlatBArgs a) {
    constexpr int WO = 7, PO = 49, WI = 14, PI = 196, NF = TO + 2, FR = 64, OOB = 0x7fff0000;
    constexpr int NG = (NF * PO + 63) / 64;                              // 4-byte loads per lane of the g' run
    constexpr int WSZ = NF * FR;
    __shared__ float smem[4 * WSZ];
    const int lane = threadIdx.x & 63, wv = cfn_uni((int)(threadIdx.x >> 6));
    const unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    const long widx = cfn_uni((long)L * 4 + wv);
    if (widx >= a.total) return;                                         // whole waves only: no workgroup barrier below
    const int chunk = cfn_uni((int)(widx % a.nchunks));
    const long nc = cfn_uni((long)(widx / a.nchunks));
    const int c = cfn_uni((int)(nc % a.C));
    const int T = a.T;
    float* img = smem + wv * WSZ;

    __amdgpu_buffer_rsrc_t rg = cfn_rsrc(a.gy + nc * (long)T * PO, (unsigned)((long)T * PO * CP_ES));
    __amdgpu_buffer_rsrc_t ry = cfn_rsrc((HASY ? a.y : a.gy) + nc * (long)T * PO, (unsigned)((long)T * PO * CP_ES));
    __amdgpu_buffer_rsrc_t rx = cfn_rsrc(a.x + nc * (long)T * PI, (unsigned)((long)T * PI * CP_ES));
    __amdgpu_buffer_rsrc_t rd = cfn_rsrc(a.gx + nc * (long)T * PI, (unsigned)((long)T * PI * CP_ES));

    float w9[3][9];                                                      // w[kt][kh][kw]
#pragma unroll
    for (int j = 0; j < 27; ++j) w9[j / 9][j % 9] = cfn_uni(a.w[(long)c * 27 + j]);
    const bool hasA = a.A != nullptr;
    const float pa = cfn_uni(hasA ? (float)a.A[nc] : 1.0f);
    const float pb = cfn_uni(hasA ? (float)a.B[nc] : 0.0f);
    const float act_lo = (hasA && a.act == CFN_ACT_RELU) ? 0.0f : -__builtin_inff();   // none / ReLU only (the planner checks)
    const float gsv = cfn_uni(a.gs ? (float)a.gs[nc] : 0.0f);
    const float gqv = cfn_uni((HASY && a.gq) ? 2.0f * (float)a.gq[nc] : 0.0f);
    // zero row 7 and column 7 of every g' frame (never overwritten): 15 floats per frame
    for (int i = lane; i < NF * 15; i += 64) {
        const int f = i / 15, j = i - f * 15;
        img[f * FR + (j < 8 ? 56 + j : (j - 8) * 8 + 7)] = 0.0f;
    }
    float dwa[27];                                                       // weight-gradient partials: live over all the wave's chunks
#pragma unroll
    for (int j = 0; j < 27; ++j) dwa[j] = 0.0f;
    float st1 = 0.0f, st2 = 0.0f;
    const bool on = lane < PO;
    const int o = lane / WO, jj = lane - o * WO;
    const int xo = on ? ((2 * o) * WI + 2 * jj) * CP_ES : OOB;               // byte offset of the block's first row in a frame of x / gx
    const float* gb = img + o * 8 + jj;

    for (int sub = 0; sub < a.subs; ++sub) {
        const int t0 = (chunk * a.subs + sub) * TO;
        if (t0 >= T) break;                                              // wave uniform
        float Rg[NG], Ry[HASY ? NG : 1];
        fb_p2 X0[TO], X1[TO];                                            // rows 2o / 2o + 1 of the block
        const int gstart = (t0 - 1) * PO;
#pragma unroll
        for (int m = 0; m < NG; ++m) {
            const int i = lane + 64 * m;
            const int vo = (i < NF * PO && gstart + i >= 0) ? (gstart + i) * CP_ES : OOB;
            Rg[m] = cp_ld1(rg, vo, 0);
            if (HASY) Ry[m] = cp_ld1(ry, vo, 0);
        }
#pragma unroll
        for (int j = 0; j < TO; ++j) {
            const bool tv = t0 + j < T;
            const int so = cfn_uni(tv ? (t0 + j) * PI * CP_ES : 0);
            X0[j] = cp_ld2(rx, tv ? xo : OOB, so);
            X1[j] = cp_ld2(rx, tv ? xo + WI * CP_ES : OOB, so);
        }
        __builtin_amdgcn_wave_barrier();                                 // the previous chunk's LDS reads are done (in-order LDS; compiler fence)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int m = 0; m < NG; ++m) {
            const int i = lane + 64 * m;
            if (i < NF * PO) {
                const int f = i / PO, p = i - f * PO, t = t0 - 1 + f;
                float v = Rg[m] + gsv;
                if (HASY) v = fmaf(Ry[m], gqv, v);
                img[f * FR + (p / WO) * 8 + p % WO] = (t >= 0 && t < T) ? v : 0.0f;   // g' is zero outside the clip
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (on) {
            float acc[TO][4], av[TO][4];                                 // [frame][(2o,2j), (2o,2j+1), (2o+1,2j), (2o+1,2j+1)]
#pragma unroll
            for (int k = 0; k < NF; ++k) {
                if (k < TO) {                                            // output frame k enters the window
                    const float m = t0 + k < T ? 1.0f : 0.0f;            // a is zero beyond the clip (act(B) is not)
                    const float xs[4] = {X0[k].x, X0[k].y, X1[k].x, X1[k].y};
#pragma unroll
                    for (int e = 0; e < 4; ++e) { acc[k][e] = 0.0f; av[k][e] = fmaxf(fmaf(xs[e], pa, pb), act_lo) * m; }
                }
                const float* q = gb + k * FR;
                const float g00 = q[0], g01 = q[1], g10 = q[8], g11 = q[9];
#pragma unroll
                for (int kr = 0; kr < 3; ++kr) {
                    const int j = k - kr, kt = 2 - kr;                   // da(t0 + j) takes w[kt] g'(t0 + j - kt + 1) = g' frame k
                    if (j >= 0 && j < TO) {
                        const float* w = w9[kt];
                        acc[j][0] = fmaf(w[4], g00, acc[j][0]);
                        acc[j][1] = fmaf(w[3], g01, fmaf(w[5], g00, acc[j][1]));
                        acc[j][2] = fmaf(w[1], g10, fmaf(w[7], g00, acc[j][2]));
                        acc[j][3] = fmaf(w[0], g11, fmaf(w[2], g10, fmaf(w[6], g01, fmaf(w[8], g00, acc[j][3]))));
                        float* d = dwa + kt * 9;
                        d[4] = fmaf(av[j][0], g00, d[4]);
                        d[3] = fmaf(av[j][1], g01, d[3]); d[5] = fmaf(av[j][1], g00, d[5]);
                        d[1] = fmaf(av[j][2], g10, d[1]); d[7] = fmaf(av[j][2], g00, d[7]);
                        d[0] = fmaf(av[j][3], g11, d[0]); d[2] = fmaf(av[j][3], g10, d[2]);
                        d[6] = fmaf(av[j][3], g01, d[6]); d[8] = fmaf(av[j][3], g00, d[8]);
                    }
                }
                if (k >= 2) {                                            // output frame k - 2 is complete
                    const int j = k - 2, t = t0 + j;
                    const bool emit = t < T;
                    float v[4] = {acc[j][0], acc[j][1], acc[j][2], acc[j][3]};
                    if (hasA) {                                          // wave uniform
                        const float xs[4] = {X0[j].x, X0[j].y, X1[j].x, X1[j].y};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float dz = av[j][e] > act_lo ? v[e] : 0.0f;     // act' of none / ReLU: a > 0 <=> z > 0
                            const float dm = emit ? dz : 0.0f;
                            st1 = fmaf(dm, xs[e], st1);
                            st2 += dm;
                            v[e] = dz * pa;
                        }
                    }
                    const int so = cfn_uni(emit ? t * PI * CP_ES : 0);
                    cp_st2((fb_p2){v[0], v[1]}, rd, emit ? xo : OOB, so);
                    cp_st2((fb_p2){v[2], v[3]}, rd, emit ? xo + WI * CP_ES : OOB, so);
                }
            }
        }
    }
    // ---- reductions: gw (27 per channel), then gA / gB ----
    {
        float v[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = (j < 27 && on) ? dwa[j] : 0.0f;
        const float tot = fb_transpose_reduce(v, lane);
        const int idx = lane >> 1;
        if ((lane & 1) == 0 && idx < 27) cfn_add64(&a.gw[(long)c * 27 + idx], (double)tot);
    }
    if (hasA && a.gA) {
        st1 = cfn_wave_sum(st1); st2 = cfn_wave_sum(st2);
        if (lane == 0) { cfn_add64(&a.gA[nc], (double)st1); cfn_add64(&a.gB[nc], (double)st2); }
    }
}

// stride 2: returns -1 when the shape is not handled (caller goes on to the wave / band kernels); H, W: INPUT plane
int CPN(dw_flatb_s2_try)(const cpe_t* gy, const cpe_t* y, const double* gs, const double* gq, const float* w, const cpe_t* x,
                    const double* A, const double* B, int act, cpe_t* gx, double* gA, double* gB, double* gw,
                    int N, int C, int T, int H, int W, hipStream_t st, bool probe) {
    static const int enabled = getenv("CFN_DW_FLATB") ? atoi(getenv("CFN_DW_FLATB")) : 24;      // 16 = 14 -> 7
    static const int subs_env = getenv("CFN_DW_FLATB_SUBS") ? atoi(getenv("CFN_DW_FLATB_SUBS")) : 0;
    if (H != 14 || W != 14 || !(enabled & 16)) return -1;
    if (A != nullptr && act != CFN_ACT_NONE && act != CFN_ACT_RELU) return -1;
    if ((long)T * H * W * CP_ES >= 0x7fff0000L) return -1;
    if ((((uintptr_t)x | (uintptr_t)gx) & (2 * CP_ES - 1)) != 0) return -1;
    const int TO = T >= 12 ? 8 : 4;
    const long nchunks = (T + TO - 1) / TO;
    const int subs = subs_env > 0 ? subs_env : 8;
    const long nch = (nchunks + subs - 1) / subs, items = (long)N * C * nch, blocks = (items + 3) / 4;
    if (blocks >= 0x7fffffffL) return -1;
    if (probe) return 0;
    DwFlatBArgs a = {gy, gq ? y : nullptr, gs, gq, w, x, A, B, gx, A ? gA : nullptr, A ? gB : nullptr, gw, N, C, T, act, (int)nch, subs, items};
#define CFN_FLATB_GO(...) hipLaunchKernelGGL((dw3d_flat14to7_bwd_kernel<__VA_ARGS__>), dim3((unsigned)blocks), dim3(256), 0, st, a)
    if (a.y) { if (TO == 8) CFN_FLATB_GO(8, true); else CFN_FLATB_GO(4, true); }
    else { if (TO == 8) CFN_FLATB_GO(8, false); else CFN_FLATB_GO(4, false); }
#undef CFN_FLATB_GO
    return cfn_check_launch("dwconv3d flat stride-2 backward");
}

// returns -1 when the shape is not handled (caller goes on to the wave / band kernels); probe: 0 = handled, nothing launched
int CPN(dw_flatb_try)(const cpe_t* gy, const cpe_t* y, const double* gs, const double* gq, const float* w, const cpe_t* x,
                 const double* A, const double* B, int act, cpe_t* gx, double* gA, double* gB, double* gw,
                 int N, int C, int T, int H, int W, hipStream_t st, bool probe) {
    // bit mask of the planes served: 8 = 7x7 (16 = 14 -> 7: dw_flatb_s2_try)
    static const int enabled = getenv("CFN_DW_FLATB") ? atoi(getenv("CFN_DW_FLATB")) : 24;
    static const int to_env = getenv("CFN_DW_FLATB_TO") ? atoi(getenv("CFN_DW_FLATB_TO")) : 0;
    if (H != W || H != 7 || !(enabled & 8)) return -1;
    if (A != nullptr && act != CFN_ACT_NONE && act != CFN_ACT_RELU) return -1;
    if ((long)T * H * W * CP_ES >= 0x7fff0000L) return -1;
    const int TO = to_env == 4 || to_env == 8 ? to_env : (T >= 12 ? 8 : 4);        // (16-frame items: 128 VGPRs + 670 spilled)
    // a wave takes `subs` consecutive chunks and reduces its 27 weight-gradient partials once (one chunk per wave: 55 k waves x 27 fp64 atomics per
    // launch made the kernel 2.6 x slower than the one it replaces)
    static const int subs_env = getenv("CFN_DW_FLATB_SUBS") ? atoi(getenv("CFN_DW_FLATB_SUBS")) : 0;
    const long nchunks = (T + TO - 1) / TO;
    const int subs = subs_env > 0 ? subs_env : 8;
    const long nch = (nchunks + subs - 1) / subs, items = (long)N * C * nch, blocks = (items + 3) / 4;
    if (blocks >= 0x7fffffffL) return -1;
    if (probe) return 0;
    DwFlatBArgs a = {gy, gq ? y : nullptr, gs, gq, w, x, A, B, gx, A ? gA : nullptr, A ? gB : nullptr, gw, N, C, T, act, (int)nch, subs, items};
#define CFN_FLATB_GO(...) hipLaunchKernelGGL((dw3d_flat7_bwd_kernel<__VA_ARGS__>), dim3((unsigned)blocks), dim3(256), 0, st, a)
    if (a.y) { if (TO == 8) CFN_FLATB_GO(8, true); else CFN_FLATB_GO(4, true); }
    else { if (TO == 8) CFN_FLATB_GO(8, false); else CFN_FLATB_GO(4, false); }
#undef CFN_FLATB_GO
    return cfn_check_launch("dwconv3d flat backward");
}
