// Fused backward of the STRIDE-2 depthwise 3x3x3 conv (112 -> 56, 56 -> 28, 28 -> 14), fp32 tensors: the column-pair wave kernel of dwcpb2.hip
// with ONE LDS image instead of three (round 4; the stride-1 counterpart is dwcpbx.hip).
//
// dwcpb2.hip stages g' at output resolution and a = act(A x + B) AND x at input resolution (4 x the size, both double buffered) because it
// forms the weight gradient from an a WINDOW and the g' centre.  Re-indexed by the position of a, every weight-gradient term is the product of
// one of the lane's OWN eight input positions with the very g' element the data gradient uses for the same tap: 9 products per output
// position and temporal tap feed both.  a and x are needed only at the lane's 2 x 4 input block, so they are loaded straight into registers
// (two 16-byte loads per frame) and only the small g' image goes through LDS.
//   (2o, 2j): w[1][1] G[o][j]                          (2o, 2j+1): w[1][0] G[o][j+1] + w[1][2] G[o][j]
//   (2o+1, 2j): w[0][1] G[o+1][j] + w[2][1] G[o][j]    (2o+1, 2j+1): w[0][0] G[o+1][j+1] + w[0][2] G[o+1][j] + w[2][0] G[o][j+1] + w[2][2] G[o][j]
// Step f: G(f) is in the image; output frame f + 1 - s (set s) takes the taps kt = 2 - s; frame f - 1 is complete afterwards.  A wave owns the
// weight-gradient terms of the a positions of ITS chunk (a is zero outside it).
// hipcc-flags: -fno-slp-vectorize
// fp32 or bf16 tensors (cp_io.h: compiled a second time through dwcpb2x_bf16.hip; the LDS image, accumulators and every reduction stay fp32 / fp64).
#include "cp_io.h"
#include <stdint.h>
#include <stdlib.h>

#ifdef DW_BF16
#define DwCpb2xArgs H16N(DwCpb2xArgs)
#endif

struct DwCpb2xArgs {
    const cpe_t* gy; const cpe_t* y; const double* gs; const double* gq; const float* w; const cpe_t* x;
    const double* A; const double* B; cpe_t* gx; double* gA; double* gB; double* gw;
    int N, C, T, act, TT, nchunks;
    long total_waves;
};

template <int WO, int RG, int OCC, bool HASY>       // WO: output width (square planes); one output row per lane, RG row groups
__global__ __launch_bounds__(256, OCC) void dw3d_cpx_bwd_s2_kernel(const DwCpb2xArgs a) {
    typedef float __attribute__((ext_vector_type(4))) f4;
    typedef float __attribute__((ext_vector_type(2))) p2;
    typedef unsigned __attribute__((ext_vector_type(4))) u4;
    constexpr int HO = WO, WI = 2 * WO, HI = 2 * HO, CP = WO / 2;
    constexpr int BR = RG, NB = (HO + BR - 1) / BR;   // output rows per band
    constexpr int GR = BR + 1, GP = (WO + 4 + 3) / 4 * 4;              // G image: rows (bottom halo), pitch (right halo, 16-byte rows)
    constexpr int IMGG = GR * GP;
    constexpr int NLG = (GR * WO / 4 + 1 + 63) / 64;   // float4 loads per lane and frame
    constexpr int PO = HO * WO, PI = HI * WI, OOB = 0x7fff0000;
    constexpr bool GROW4 = WO % 4 == 0;                // a float4 of the output-resolution tensors never straddles two rows
    static_assert(CP * RG <= 64 && WO % 2 == 0, "geometry");
    constexpr int WSZ = 2 * IMGG + 8;                  // per wave: G[2], dump slot
    __shared__ __attribute__((aligned(16))) float smem[4 * WSZ];

    const int lane = threadIdx.x & 63, wv = cfn_uni((int)(threadIdx.x >> 6));
    const unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    const long widx = cfn_uni((long)L * 4 + wv);
    if (widx >= a.total_waves) return;                // whole waves only: no barrier anywhere below
    const int band = cfn_uni((int)(widx % NB));
    const long rest = cfn_uni((long)(widx / NB));
    const int chunk = cfn_uni((int)(rest % a.nchunks));
    const long nc = cfn_uni((long)(rest / a.nchunks));
    const int c = cfn_uni((int)(nc % a.C));
    const int T = a.T, t0 = chunk * a.TT, t1 = min(t0 + a.TT, T);
    float* imG = smem + wv * WSZ;
    float* dump = imG + 2 * IMGG;

    float wr[27];
#pragma unroll
    for (int j = 0; j < 27; ++j) wr[j] = cfn_uni(a.w[(long)c * 27 + j]);
    const bool hasA = a.A != nullptr;
    const float pa = cfn_uni(hasA ? (float)a.A[nc] : 1.0f);
    const float pb = cfn_uni(hasA ? (float)a.B[nc] : 0.0f);
    const float act_lo = (hasA && a.act == CFN_ACT_RELU) ? 0.0f : -__builtin_inff();   // none / ReLU only (the planner checks)
    const float gsv = cfn_uni(a.gs ? (float)a.gs[nc] : 0.0f);
    const float gqv = cfn_uni((HASY && a.gq) ? 2.0f * (float)a.gq[nc] : 0.0f);

    for (int i = lane; i < WSZ; i += 64) imG[i] = 0.0f;           // halos (and everything else) zero; wave-private

    // loader: output-resolution rows band*BR .. (+ BR, the bottom halo)
    const int gr_lo = band * BR, gr_hi = min(band * BR + GR, HO);
    int ldg[NLG], lg0[NLG], lg1[GROW4 ? 1 : NLG];
    // float4s from the 16-byte aligned element at or below the band's first one (14-wide planes: a band can start in the middle of a
    // float4; the two leading elements then belong to the row above and go to the dump slot)
    const int eb = (gr_lo * WO) & ~3;
#pragma unroll
    for (int k = 0; k < NLG; ++k) {
        const int e0 = eb + (k * 64 + lane) * 4;
        const bool on = e0 < gr_hi * WO;
        ldg[k] = on ? e0 * CP_ES : OOB;
        const int r0 = e0 / WO - gr_lo, r2 = (e0 + 2) / WO - gr_lo;
        lg0[k] = (on && e0 >= gr_lo * WO) ? r0 * GP + e0 % WO : -1;
        if (!GROW4) lg1[k] = (on && e0 + 2 < gr_hi * WO) ? r2 * GP + (e0 + 2) % WO : -1;
    }
    // compute lane: output row o = band*BR + g, output columns 2cp, 2cp+1; input block rows 2o, 2o+1, columns 4cp .. 4cp+3
    const int g = lane / CP, cp = lane - g * CP;
    const bool act_lane = g < RG && band * BR + g < HO;
    const int gofs = act_lane ? g * GP + 2 * cp : 0;                       // G image: g'[o][2cp]
    const int xo = act_lane ? ((2 * (band * BR + g)) * WI + 4 * cp) * CP_ES : OOB;   // x / gx: row 2o, column 4cp

    __amdgpu_buffer_rsrc_t rgy = cfn_rsrc(a.gy + nc * (long)T * PO, (unsigned)((long)T * PO * CP_ES));
    __amdgpu_buffer_rsrc_t ryy = cfn_rsrc((HASY ? a.y : a.gy) + nc * (long)T * PO, (unsigned)((long)T * PO * CP_ES));
    __amdgpu_buffer_rsrc_t rx = cfn_rsrc(a.x + nc * (long)T * PI, (unsigned)((long)T * PI * CP_ES));
    __amdgpu_buffer_rsrc_t rd = cfn_rsrc(a.gx + nc * (long)T * PI, (unsigned)((long)T * PI * CP_ES));

    auto fetchG = [&](__amdgpu_buffer_rsrc_t r, int f, f4 (&dst)[NLG]) {     // unconditional: an unwanted frame reads zeros
        const bool want = f >= 0 && f < T && f <= t1;
        const int so = cfn_uni(want ? f * PO * CP_ES : 0);
#pragma unroll
        for (int k = 0; k < NLG; ++k) dst[k] = cp_ld4(r, want ? ldg[k] : OOB, so);
    };
    auto fetchX = [&](int f, f4 (&dst)[2]) {           // the lane's own 2 x 4 block of x(f); frames outside the chunk read zeros
        const bool want = f >= t0 && f < t1;
        const int so = cfn_uni(want ? f * PI * CP_ES : 0);
        dst[0] = cp_ld4(rx, want ? xo : OOB, so);
        dst[1] = cp_ld4(rx, want ? xo + WI * CP_ES : OOB, so);
    };
    auto stageG = [&](int f, const f4 (&sg)[NLG], const f4 (&sy)[NLG], float* im) {   // g' = gy + gs + 2 y gq, zero outside
        const float m = (f >= 0 && f < T && f <= t1) ? 1.0f : 0.0f;
#pragma unroll
        for (int k = 0; k < NLG; ++k) {
            f4 v = sg[k] + gsv;
            if (HASY) v += sy[k] * gqv;
            v *= m;
            if (GROW4) {
                *reinterpret_cast<f4*>(lg0[k] >= 0 ? im + lg0[k] : dump) = v;
            } else {
                *reinterpret_cast<p2*>(lg0[k] >= 0 ? im + lg0[k] : dump) = (p2){v.x, v.y};
                *reinterpret_cast<p2*>(lg1[k] >= 0 ? im + lg1[k] : dump + 4) = (p2){v.z, v.w};
            }
        }
    };
    auto wave_sync = [&]() {                          // LDS ops of a wave run in order; only the compiler has to be told
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    // set s: frame f + 1 - s;  [rr][cc]: block row (0, 1), block column (0..3)
    float acc[3][2][4], av[3][2][4], xv[3][2][4], dwa[27];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) { acc[s][e >> 2][e & 3] = 0.0f; av[s][e >> 2][e & 3] = 0.0f; xv[s][e >> 2][e & 3] = 0.0f; }
#pragma unroll
    for (int j = 0; j < 27; ++j) dwa[j] = 0.0f;
    float st1 = 0.0f, st2 = 0.0f;
    const float lane_m = act_lane ? 1.0f : 0.0f;

    // steps f = t0-1 .. t1.  Step j of a 2-step trip: G(f) is in imG[j & 1]; the rings hold gy / y / x of frame f+1 and are refilled right
    // after they were consumed.
    const int f_first = t0 - 1, f_last = t1;
    f4 rgG[NLG], rgY[NLG], rgX[2];
    {
        f4 fg[NLG], fy[NLG];
        fetchG(rgy, f_first, fg);
        if (HASY) fetchG(ryy, f_first, fy);
        fetchG(rgy, f_first + 1, rgG);
        if (HASY) fetchG(ryy, f_first + 1, rgY);
        fetchX(f_first + 1, rgX);
        wave_sync();
        stageG(f_first, fg, HASY ? fy : fg, imG);
    }
    for (int f0 = f_first; f0 <= f_last; f0 += 2) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int f = f0 + j;
            const int pg = j & 1, pq = (j + 1) & 1;                         // image of G(f); image that takes G(f+1)
            stageG(f + 1, rgG, HASY ? rgY : rgG, imG + pq * IMGG);
            fetchG(rgy, f + 2, rgG);
            if (HASY) fetchG(ryy, f + 2, rgY);
            {   // x(f+1) arrives: a(f+1) (zero outside the chunk: the loads were not issued there, and act(B) is not 0)
                const float m = (f + 1 >= t0 && f + 1 < t1) ? lane_m : 0.0f;
#pragma unroll
                for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = rgX[rr][e];
                        xv[0][rr][e] = x;
                        av[0][rr][e] = fmaxf(fmaf(x, pa, pb), act_lo) * m;
                    }
                fetchX(f + 2, rgX);
            }
            wave_sync();
            {
                const float* tg = imG + pg * IMGG + gofs;
                const float g00 = tg[0], g01 = tg[1], g02 = tg[2], g10 = tg[GP], g11 = tg[GP + 1], g12 = tg[GP + 2];
#pragma unroll
                for (int s = 0; s < 3; ++s) {                               // frame f+1-s takes the taps kt = 2 - s
                    const float* wk = wr + (2 - s) * 9;
                    float* dk = dwa + (2 - s) * 9;
                    const float (*aa)[4] = av[s];
                    // block of output column 2cp (input columns 4cp, 4cp+1)
                    acc[s][0][0] = fmaf(wk[4], g00, acc[s][0][0]);
                    acc[s][0][1] = fmaf(wk[3], g01, fmaf(wk[5], g00, acc[s][0][1]));
                    acc[s][1][0] = fmaf(wk[1], g10, fmaf(wk[7], g00, acc[s][1][0]));
                    acc[s][1][1] = fmaf(wk[0], g11, fmaf(wk[2], g10, fmaf(wk[6], g01, fmaf(wk[8], g00, acc[s][1][1]))));
                    // block of output column 2cp+1 (input columns 4cp+2, 4cp+3): one column to the right
                    acc[s][0][2] = fmaf(wk[4], g01, acc[s][0][2]);
                    acc[s][0][3] = fmaf(wk[3], g02, fmaf(wk[5], g01, acc[s][0][3]));
                    acc[s][1][2] = fmaf(wk[1], g11, fmaf(wk[7], g01, acc[s][1][2]));
                    acc[s][1][3] = fmaf(wk[0], g12, fmaf(wk[2], g11, fmaf(wk[6], g02, fmaf(wk[8], g01, acc[s][1][3]))));
                    // the same products with the lane's a in place of the weight: weight gradient
                    dk[4] = fmaf(aa[0][0], g00, fmaf(aa[0][2], g01, dk[4]));
                    dk[3] = fmaf(aa[0][1], g01, fmaf(aa[0][3], g02, dk[3]));
                    dk[5] = fmaf(aa[0][1], g00, fmaf(aa[0][3], g01, dk[5]));
                    dk[1] = fmaf(aa[1][0], g10, fmaf(aa[1][2], g11, dk[1]));
                    dk[7] = fmaf(aa[1][0], g00, fmaf(aa[1][2], g01, dk[7]));
                    dk[0] = fmaf(aa[1][1], g11, fmaf(aa[1][3], g12, dk[0]));
                    dk[2] = fmaf(aa[1][1], g10, fmaf(aa[1][3], g11, dk[2]));
                    dk[6] = fmaf(aa[1][1], g01, fmaf(aa[1][3], g02, dk[6]));
                    dk[8] = fmaf(aa[1][1], g00, fmaf(aa[1][3], g01, dk[8]));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- emit gx(f-1): complete in set 2 ----
            const int to = f - 1;
            const bool emit = to >= t0 && to < t1;                         // wave uniform
            const int so = cfn_uni(emit ? to * PI * CP_ES : 0);
            const float mf = emit ? lane_m : 0.0f;
            const int vo = emit ? xo : OOB;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                f4 v = {acc[2][rr][0], acc[2][rr][1], acc[2][rr][2], acc[2][rr][3]};
                if (hasA) {                                                // wave uniform
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float dz = av[2][rr][e] > act_lo ? v[e] : 0.0f;   // act' of none / ReLU: a > 0 <=> z > 0
                        const float dm = dz * mf;
                        st1 = fmaf(dm, xv[2][rr][e], st1);
                        st2 += dm;
                        v[e] = dz * pa;
                    }
                }
                cp_st4(v, rd, vo + rr * WI * CP_ES, so);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {                                  // rotate: frame f+1 becomes frame f of the next step
                const int rr = e >> 2, cc = e & 3;
                acc[2][rr][cc] = acc[1][rr][cc]; acc[1][rr][cc] = acc[0][rr][cc]; acc[0][rr][cc] = 0.0f;
                av[2][rr][cc] = av[1][rr][cc]; av[1][rr][cc] = av[0][rr][cc];
                xv[2][rr][cc] = xv[1][rr][cc]; xv[1][rr][cc] = xv[0][rr][cc];
            }
            asm volatile("" : "+v"(st1), "+v"(st2));
        }
    }
    // ---- reductions: gw (27 per channel) by transpose-reduce (see dwcpb.hip), then gA / gB ----
    {
        float v[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = (j < 27 && act_lane) ? dwa[j] : 0.0f;
#pragma unroll
        for (int st = 0; st < 5; ++st) {
            const int half = 16 >> st, bit = 32 >> st;
            const bool up = (lane & bit) != 0;
#pragma unroll
            for (int k = 0; k < half; ++k) {
                const float send = up ? v[k] : v[k + half];
                const float keep = up ? v[k + half] : v[k];
                v[k] = keep + __shfl_xor(send, bit, 64);
            }
        }
        const float tot = v[0] + __shfl_xor(v[0], 1, 64);
        const int idx = lane >> 1;
        if ((lane & 1) == 0 && idx < 27) cfn_add64(&a.gw[(long)c * 27 + idx], (double)tot);
    }
    if (hasA && a.gA) {
        st1 = cfn_wave_sum(st1); st2 = cfn_wave_sum(st2);
        if (lane == 0) { cfn_add64(&a.gA[nc], (double)st1); cfn_add64(&a.gB[nc], (double)st2); }
    }
}

// returns -1 when the shape is not handled (caller goes on to dwcpb2.hip); probe: 0 = handled, nothing launched.  H, W: input size.
int CPN(dw_cpb2x_try)(const cpe_t* gy, const cpe_t* y, const double* gs, const double* gq, const float* w, const cpe_t* x,
                 const double* A, const double* B, int act, cpe_t* gx, double* gA, double* gB, double* gw,
                 int N, int C, int T, int H, int W, hipStream_t st, bool probe) {
    // bit mask of the shapes served: 1 = 112->56, 2 = 56->28, 4 = 28->14
    static const int enabled = getenv("CFN_DW_CPB2X") ? atoi(getenv("CFN_DW_CPB2X")) : 7;
    static const int tt_env = getenv("CFN_DW_CPB2_TT") ? atoi(getenv("CFN_DW_CPB2_TT")) : 0;
    if (H != W || (H != 112 && H != 56 && H != 28)) return -1;
    if (!(enabled & (H == 112 ? 1 : H == 56 ? 2 : 4))) return -1;
    if (A != nullptr && act != CFN_ACT_NONE && act != CFN_ACT_RELU) return -1;      // act' from the sign of a: none / ReLU (every X3D conv2)
    if ((long)T * H * W * CP_ES >= 0x7fff0000L) return -1;
    if ((((uintptr_t)gy | (uintptr_t)x | (uintptr_t)gx | (uintptr_t)(y ? y : gy)) & (4 * CP_ES - 1)) != 0) return -1;
    if (probe) return 0;
    const bool hasy = y != nullptr && gq != nullptr;
    DwCpb2xArgs a = {gy, hasy ? y : nullptr, gs, hasy ? gq : nullptr, w, x, A, B, gx, A ? gA : nullptr, A ? gB : nullptr, gw, N, C, T, act, 0, 0, 0};
    const int NB = H == 112 ? 28 : H == 56 ? 7 : 2;
    const long units = (long)N * C * NB;
    long nch = (T + 26) / 52;
    if (nch < 1) nch = 1;
    while (units * nch < 2L * 256 * 12 && (T + nch) / (nch + 1) >= 16) ++nch;
    int TT = (int)((T + nch - 1) / nch);
    if (TT > 24) TT = 24;                                                           // (the sweep of dwcpb2.hip)
    if (tt_env > 0) TT = tt_env;
    if (TT > T) TT = T;
    a.TT = TT;
    a.nchunks = (T + TT - 1) / TT;
    a.total_waves = units * a.nchunks;
    const unsigned blocks = (unsigned)((a.total_waves + 3) / 4);
#define CFN_CPB2X_GO(...) do { if (hasy) hipLaunchKernelGGL((dw3d_cpx_bwd_s2_kernel<__VA_ARGS__, true>), dim3(blocks), dim3(256), 0, st, a); \
                               else hipLaunchKernelGGL((dw3d_cpx_bwd_s2_kernel<__VA_ARGS__, false>), dim3(blocks), dim3(256), 0, st, a); } while (0)
    if (H == 112) CFN_CPB2X_GO(56, 2, 3);
    else if (H == 56) CFN_CPB2X_GO(28, 4, 3);
    else CFN_CPB2X_GO(14, 7, 3);
#undef CFN_CPB2X_GO
    return cfn_check_launch("dwconv3d stride-2 column-pair fused backward (one image)");
}
