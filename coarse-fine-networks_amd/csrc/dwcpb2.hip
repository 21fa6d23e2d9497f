// Fused backward of the STRIDE-2 depthwise 3x3x3 conv (first block of every X3D stage: 112->56, 56->28, 28->14 at 224x224
// input; x3d_fine.py:89-97,171-201): data gradient AND weight gradient in one pass -- column-pair wave kernel, the stride-2
// counterpart of dwcpb.hip.
//
// Why: the two separate kernels (dw3d_dgrad_s2_fast_kernel, dw3d_kernel<WGRAD, 2>) read the input-resolution tensor x TWICE
// and write gx once: 3 passes at input resolution (plus gy / y at output resolution, a quarter of the size, twice).  Fused: x
// once, gx once.  They were the largest rows of the step's kernel table (5.2 + 3.5 ms of 103 ms, profiles/).
// A lane owns two horizontally adjacent output positions (o, j), (o, j+1) = the 2 x 4 input block rows 2o..2o+1, columns
// 4cp..4cp+3 (cp = j / 2).  Three wave-private LDS images per frame parity, no workgroup barrier:
//   G image  g'(f)  = gy + gs + 2 y gq   output resolution, right / bottom halo
//   A image  a(f-1) = act(A x + B)       input resolution, top / left halo (window -> weight gradient; block -> act')
//   X image  x(f-1)                      input resolution (block -> gA, gB)
// Data gradient (tap parity: an even input row / column sees only the centre tap, an odd one the two outer taps -- 27 FMAs
// per 2 x 2 input block and frame, as in dw3d_dgrad_s2_kernel):
//   gx(t)[2o  ][2j  ] = sum_kt w[kt][1][1] g'(t+1-kt)[o][j]
//   gx(t)[2o  ][2j+1] = sum_kt w[kt][1][0] g'[o][j+1] + w[kt][1][2] g'[o][j]
//   gx(t)[2o+1][2j  ] = sum_kt w[kt][0][1] g'[o+1][j] + w[kt][2][1] g'[o][j]
//   gx(t)[2o+1][2j+1] = sum_kt w[kt][0][0] g'[o+1][j+1] + w[kt][0][2] g'[o+1][j] + w[kt][2][0] g'[o][j+1] + w[kt][2][2] g'[o][j]
// with three rolling accumulator sets: frame f of g' feeds gx(f-1+kt).  Weight gradient: gw[kt][kh][kw] += g'(f-kt)[o][j] *
// a(f-1)[2o+kh-1][2j+kw-1] (the forward is y(t) = sum w[kt] a(t+kt-1)).  Even input sizes only (no bottom / right input halo);
// prologue activations other than none / ReLU and other planes keep the two band kernels (dw_cpb2_try returns -1).
// hipcc-flags: -fno-slp-vectorize
#include "cp_io.h"
#include <stdint.h>
#include <stdlib.h>

#ifdef DW_BF16
#define DwCpb2Args H16N(DwCpb2Args)
#endif
struct DwCpb2Args {
    const cpe_t* gy; const cpe_t* y; const double* gs; const double* gq; const float* w; const cpe_t* x;
    const double* A; const double* B; cpe_t* gx; double* gA; double* gB; double* gw;
    int N, C, T, act, TT, nchunks;
    long total_waves;
};

template <int WO, int RG, int OCC, bool HASY>       // WO: output width (square planes); one output row per lane, RG row groups
__global__ __launch_bounds__(256, OCC) void dw3d_cp_bwd_s2_kernel(const DwCpb2Args a) {
    typedef float __attribute__((ext_vector_type(4))) f4;
    typedef float __attribute__((ext_vector_type(2))) p2;
    typedef unsigned __attribute__((ext_vector_type(4))) u4;
    constexpr int HO = WO, WI = 2 * WO, HI = 2 * HO, CP = WO / 2;
    constexpr int BR = RG, NB = (HO + BR - 1) / BR;   // output rows per band
    constexpr int GR = BR + 1, GP = (WO + 4 + 3) / 4 * 4;              // G image: rows (bottom halo), pitch (right halo, 16-byte rows)
    constexpr int IR = 2 * BR + 1, XO = 4, PIT = WI + 8;              // A / X image: rows (top halo), column 0 at XO (left halo at XO-1)
    constexpr int IMGG = GR * GP, IMGA = IR * PIT;
    constexpr int NLG = (GR * WO / 4 + 1 + 63) / 64, NLA = (IR * WI / 4 + 63) / 64;   // float4 loads per lane and frame
    constexpr int PO = HO * WO, PI = HI * WI, OOB = 0x7fff0000;
    constexpr bool GROW4 = WO % 4 == 0;                // a float4 of the output-resolution tensors never straddles two rows
    static_assert(CP * RG <= 64 && WO % 2 == 0, "geometry");
    constexpr int WSZ = 2 * IMGG + 4 * IMGA + 8;       // per wave: G[2], A[2], X[2], dump slot
    __shared__ __attribute__((aligned(16))) float smem[4 * WSZ];

    const int lane = threadIdx.x & 63, wv = cfn_uni((int)(threadIdx.x >> 6));
    const unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    const long widx = cfn_uni((long)L * 4 + wv);      // wave-uniform by construction; stated for the compiler (see cfn_uni)
    if (widx >= a.total_waves) return;                // whole waves only: no barrier anywhere below
    const int band = cfn_uni((int)(widx % NB));
    const long rest = cfn_uni((long)(widx / NB));
    const int chunk = cfn_uni((int)(rest % a.nchunks));
    const long nc = cfn_uni((long)(rest / a.nchunks));
    const int c = cfn_uni((int)(nc % a.C));
    const int T = a.T, t0 = chunk * a.TT, t1 = min(t0 + a.TT, T);
    float* imG = smem + wv * WSZ;
    float* imA = imG + 2 * IMGG;
    float* imX = imA + 2 * IMGA;
    float* dump = imX + 2 * IMGA;

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

    // loaders: output-resolution rows band*BR .. (+ BR, the bottom halo), input-resolution rows 2 band BR - 1 .. (+ 2 BR)
    const int gr_lo = band * BR, gr_hi = min(band * BR + GR, HO);
    const int ar_lo = max(2 * band * BR - 1, 0), ar_hi = min(2 * band * BR - 1 + IR, HI);
    const int nela = (ar_hi - ar_lo) * WI;
    int ldg[NLG], lg0[NLG], lg1[GROW4 ? 1 : NLG], lda[NLA], la0[NLA];
    // output-resolution loader: float4s from the 16-byte aligned element at or below the band's first one (14-wide planes: a
    // band can start in the middle of a float4; the two leading elements then belong to the row above and go to the dump slot)
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
#pragma unroll
    for (int k = 0; k < NLA; ++k) {
        const int e0 = (k * 64 + lane) * 4;                                // WI % 4 == 0: a float4 stays in its row
        const bool on = e0 < nela;
        lda[k] = on ? (ar_lo * WI + e0) * CP_ES : OOB;
        la0[k] = on ? (ar_lo + e0 / WI - (2 * band * BR - 1)) * PIT + XO + e0 % WI : -1;
    }
    // compute lane: output row o = band*BR + g, output columns 2cp, 2cp+1; input block rows 2o, 2o+1, columns 4cp .. 4cp+3
    const int g = lane / CP, cp = lane - g * CP;
    const bool act_lane = g < RG && band * BR + g < HO;
    const int gofs = act_lane ? g * GP + 2 * cp : 0;                       // G image: g'[o][2cp]
    const int aofs = act_lane ? (2 * g) * PIT + (XO - 1) + 4 * cp : XO - 1;   // A / X image: row 2o-1, column 4cp-1 (idle lanes: aligned too)
    const int xo = act_lane ? ((2 * (band * BR + g)) * WI + 4 * cp) * CP_ES : OOB;   // gx: row 2o, column 4cp

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
    auto fetchX = [&](int f, f4 (&dst)[NLA]) {
        const bool want = f >= 0 && f < T && f <= t1;
        const int so = cfn_uni(want ? f * PI * CP_ES : 0);
#pragma unroll
        for (int k = 0; k < NLA; ++k) dst[k] = cp_ld4(rx, want ? lda[k] : OOB, so);
    };
    // branch-free staging: a loader lane without an element writes into the wave's dump slot
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
    auto stageAX = [&](int f, const f4 (&sx)[NLA], float* ia, float* ix) {   // a = act(A x + B) (zero outside the chunk's frames), x
        const float m = (f >= 0 && f < T && f >= t0 - 1 && f <= t1) ? 1.0f : 0.0f;
#pragma unroll
        for (int k = 0; k < NLA; ++k) {
            const f4 x = sx[k];
            f4 v;
            v.x = fmaxf(fmaf(x.x, pa, pb), act_lo) * m; v.y = fmaxf(fmaf(x.y, pa, pb), act_lo) * m;
            v.z = fmaxf(fmaf(x.z, pa, pb), act_lo) * m; v.w = fmaxf(fmaf(x.w, pa, pb), act_lo) * m;
            *reinterpret_cast<f4*>(la0[k] >= 0 ? ia + la0[k] : dump) = v;
            *reinterpret_cast<f4*>(la0[k] >= 0 ? ix + la0[k] : dump) = x;
        }
    };
    auto wave_sync = [&]() {                          // LDS ops of a wave run in order; only the compiler has to be told
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    // acc[s][rr][cc]: gx of frame f+1-s at block row rr (0, 1), block column cc (0..3)
    float acc[3][2][4], dwa[27];
    p2 gc[3];                                          // g'(f - kt) at the lane's two output positions
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        gc[s] = (p2){0.0f, 0.0f};
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[s][e >> 2][e & 3] = 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 27; ++j) dwa[j] = 0.0f;
    float st1 = 0.0f, st2 = 0.0f;
    const float lane_m = act_lane ? 1.0f : 0.0f;

    // steps f = t0-1 .. t1+1.  Step j of a 2-step trip: G(f) is in imG[j & 1], A / X(f-1) in im?[(j+1) & 1]; the rings hold
    // gy / y of frame f+1 and x of frame f and are refilled right after they were staged.
    const int f_first = t0 - 1, f_last = t1 + 1;
    f4 rgG[NLG], rgY[NLG], rgX[NLA];
    {
        f4 fg[NLG], fy[NLG];
        fetchG(rgy, f_first, fg);
        if (HASY) fetchG(ryy, f_first, fy);
        fetchG(rgy, f_first + 1, rgG);
        if (HASY) fetchG(ryy, f_first + 1, rgY);
        fetchX(f_first, rgX);
        wave_sync();
        stageG(f_first, fg, HASY ? fy : fg, imG);
    }
    for (int f0 = f_first; f0 <= f_last; f0 += 2) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int f = f0 + j;
            const int pg = j & 1, pq = (j + 1) & 1;                         // image of G(f); image of A / X(f-1)
            stageG(f + 1, rgG, HASY ? rgY : rgG, imG + pq * IMGG);
            fetchG(rgy, f + 2, rgG);
            if (HASY) fetchG(ryy, f + 2, rgY);
            stageAX(f, rgX, imA + pg * IMGA, imX + pg * IMGA);
            fetchX(f + 1, rgX);
            wave_sync();
            // ---- data gradient from g'(f): rows o, o+1, columns 2cp .. 2cp+2 --------------------------------------------
            const bool inchunk = f >= t0 && f < t1;                         // g'(f) is a weight-gradient term only inside the chunk
            const float cm = inchunk ? 1.0f : 0.0f;
            {
                const float* tg = imG + pg * IMGG + gofs;
                const float g00 = tg[0], g01 = tg[1], g02 = tg[2], g10 = tg[GP], g11 = tg[GP + 1], g12 = tg[GP + 2];
                gc[0] = (p2){g00 * cm, g01 * cm};
#pragma unroll
                for (int s = 0; s < 3; ++s) {                               // frame f+1-s takes the taps kt = 2 - s
                    const float* wk = wr + (2 - s) * 9;
                    // block of output column 2cp (input columns 4cp, 4cp+1): g[0] = g00, g[1] = g01, g[2] = g10, g[3] = g11
                    acc[s][0][0] = fmaf(wk[4], g00, acc[s][0][0]);
                    acc[s][0][1] = fmaf(wk[3], g01, fmaf(wk[5], g00, acc[s][0][1]));
                    acc[s][1][0] = fmaf(wk[1], g10, fmaf(wk[7], g00, acc[s][1][0]));
                    acc[s][1][1] = fmaf(wk[0], g11, fmaf(wk[2], g10, fmaf(wk[6], g01, fmaf(wk[8], g00, acc[s][1][1]))));
                    // block of output column 2cp+1 (input columns 4cp+2, 4cp+3): one column to the right
                    acc[s][0][2] = fmaf(wk[4], g01, acc[s][0][2]);
                    acc[s][0][3] = fmaf(wk[3], g02, fmaf(wk[5], g01, acc[s][0][3]));
                    acc[s][1][2] = fmaf(wk[1], g11, fmaf(wk[7], g01, acc[s][1][2]));
                    acc[s][1][3] = fmaf(wk[0], g12, fmaf(wk[2], g11, fmaf(wk[6], g02, fmaf(wk[8], g01, acc[s][1][3]))));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- weight gradient: A window of frame f-1 (rows 2o-1 .. 2o+1, columns 4cp-1 .. 4cp+3) ---------------------------
            float ac[2][4];
            {
                const float* tp = imA + pq * IMGA + aofs;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const float* q = tp + r * PIT;
                    const float qq[5] = {q[0], q[1], q[2], q[3], q[4]};
                    if (r >= 1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) ac[r - 1][e] = qq[e + 1];
                    }
#pragma unroll
                    for (int kt = 0; kt < 3; ++kt) {
                        const p2 gg = gc[kt];                              // g'(f - kt) at (o, 2cp), (o, 2cp+1)
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw)
                            dwa[kt * 9 + r * 3 + kw] = fmaf(gg.x, qq[kw], fmaf(gg.y, qq[kw + 2], dwa[kt * 9 + r * 3 + kw]));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- emit gx(f-1): complete in set 2 ----------------------------------------------------------------------------
            const int to = f - 1;
            const bool emit = to >= t0 && to < t1;                         // wave uniform
            const int so = cfn_uni(emit ? to * PI * CP_ES : 0);
            const float mf = emit ? lane_m : 0.0f;
            const int vo = emit ? xo : OOB;
            const float* tx = imX + pq * IMGA + aofs;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                f4 v = {acc[2][rr][0], acc[2][rr][1], acc[2][rr][2], acc[2][rr][3]};
                if (hasA) {                                                // wave uniform
                    const f4 xe = *reinterpret_cast<const f4*>(tx + (rr + 1) * PIT + 1);   // columns 4cp .. 4cp+3: 16-byte aligned
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float dz = ac[rr][e] > act_lo ? v[e] : 0.0f;  // act' of none / ReLU: a > 0 <=> z > 0
                        const float dm = dz * mf;
                        st1 = fmaf(dm, xe[e], st1);
                        st2 += dm;
                        v[e] = dz * pa;
                    }
                }
                cp_st4(v, rd, vo + rr * WI * CP_ES, so);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {                                  // rotate: frame f+1 becomes frame f of the next step
                acc[2][e >> 2][e & 3] = acc[1][e >> 2][e & 3]; acc[1][e >> 2][e & 3] = acc[0][e >> 2][e & 3]; acc[0][e >> 2][e & 3] = 0.0f;
            }
            gc[2] = gc[1]; gc[1] = gc[0];
            asm volatile("" : "+v"(st1), "+v"(st2));
        }
    }
    // ---- reductions: gw (27 per channel) by transpose-reduce (see dwcpb.hip), then gA / gB ---------------------------------
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

// returns -1 when the shape is not handled (caller uses the two band kernels); otherwise the launch status.  H, W: input size.
int CPN(dw_cpb2_try)(const cpe_t* gy, const cpe_t* y, const double* gs, const double* gq, const float* w, const cpe_t* x,
                const double* A, const double* B, int act, cpe_t* gx, double* gA, double* gB, double* gw,
                int N, int C, int T, int H, int W, hipStream_t st) {
    // bit mask of the shapes served: 1 = 112->56, 2 = 56->28, 4 = 28->14
    static const int enabled = getenv("CFN_DW_CPB2") ? atoi(getenv("CFN_DW_CPB2")) : 7;
    static const int tt_env = getenv("CFN_DW_CPB2_TT") ? atoi(getenv("CFN_DW_CPB2_TT")) : 0;
    if (H != W || (H != 112 && H != 56 && H != 28)) return -1;
    if (!(enabled & (H == 112 ? 1 : H == 56 ? 2 : 4))) return -1;
    if (A != nullptr && act != CFN_ACT_NONE && act != CFN_ACT_RELU) return -1;      // act' from the sign of a: none / ReLU (every X3D conv2)
    if ((long)T * H * W * CP_ES >= 0x7fff0000L) return -1;
    if ((((uintptr_t)gy | (uintptr_t)x | (uintptr_t)gx | (uintptr_t)(y ? y : gy)) & (4 * CP_ES - 1)) != 0) return -1;
    const bool hasy = y != nullptr && gq != nullptr;
    DwCpb2Args a = {gy, hasy ? y : nullptr, gs, hasy ? gq : nullptr, w, x, A, B, gx, A ? gA : nullptr, A ? gB : nullptr, gw, N, C, T, act, 0, 0, 0};
    const int NB = H == 112 ? 28 : H == 56 ? 7 : 2;
    // t-chunks of ~52 frames (a wave's fixed cost is worth a few frame steps, see dwcpb.hip); more chunks only while the grid
    // has fewer than ~2 rounds of the resident waves (12 per CU)
    const long units = (long)N * C * NB;
    long nch = (T + 26) / 52;
    if (nch < 1) nch = 1;
    while (units * nch < 2L * 256 * 12 && (T + nch) / (nch + 1) >= 16) ++nch;
    int TT = (int)((T + nch - 1) / nch);
    // same-box sweep at 8 clips x T = 256 (16 / 24 / 32 / 48 / 52 / 64 frames): 112->56 2.76 / 2.68 / 2.76 / 2.82 / 2.88 / 2.76 ms,
    // 56->28 1.36 / 1.38 / 1.40 / 1.44 / 1.40 / 1.41, 28->14 0.71 / 0.71 / 0.72 / 0.74 / 0.73 / 0.74
    if (TT > 24) TT = 24;
    if (tt_env > 0) TT = tt_env;
    if (TT > T) TT = T;
    a.TT = TT;
    a.nchunks = (T + TT - 1) / TT;
    a.total_waves = units * a.nchunks;
    const unsigned blocks = (unsigned)((a.total_waves + 3) / 4);
#define CFN_CPB2_GO(...) do { if (hasy) hipLaunchKernelGGL((dw3d_cp_bwd_s2_kernel<__VA_ARGS__, true>), dim3(blocks), dim3(256), 0, st, a); \
                              else hipLaunchKernelGGL((dw3d_cp_bwd_s2_kernel<__VA_ARGS__, false>), dim3(blocks), dim3(256), 0, st, a); } while (0)
    if (H == 112) CFN_CPB2_GO(56, 2, 3);
    else if (H == 56) CFN_CPB2_GO(28, 4, 3);
    else CFN_CPB2_GO(14, 7, 3);
#undef CFN_CPB2_GO
    return cfn_check_launch("dwconv3d stride-2 column-pair fused backward");
}

int CPN(dw_flatb_s2_try)(const cpe_t* gy, const cpe_t* y, const double* gs, const double* gq, const float* w, const cpe_t* x,
                         const double* A, const double* B, int act, cpe_t* gx, double* gA, double* gB, double* gw,
                         int N, int C, int T, int H, int W, hipStream_t st, bool probe);              // dwflatb.hip (both element types)
int CPN(dw_cpb2x_try)(const cpe_t* gy, const cpe_t* y, const double* gs, const double* gq, const float* w, const cpe_t* x,
                      const double* A, const double* B, int act, cpe_t* gx, double* gA, double* gB, double* gw,
                      int N, int C, int T, int H, int W, hipStream_t st, bool probe);                 // dwcpb2x.hip (both element types)
// C ABI (include/cfn_hip.h): data AND weight gradient of the stride-2 conv in one pass; -1 = not handled, call the two kernels
extern "C" int CPN(cfn_dwconv3d_bwd_fused_s2)(const cpe_t* gy, const cpe_t* y, const double* gsum, const double* gsumsq, const float* w,
                                         const cpe_t* x, const double* A, const double* B, int act, cpe_t* gx, double* gA, double* gB,
                                         double* gw, int N, int C, int T, int H, int W, void* stream) {
    CFN_REQUIRE(gy && w && x && gx && gw, "cfn_dwconv3d_bwd_fused_s2: null tensor");
    CFN_REQUIRE(N > 0 && C > 0 && T > 0 && H > 0 && W > 0, "cfn_dwconv3d_bwd_fused_s2: bad shape");
    CFN_REQUIRE((A == nullptr) == (B == nullptr), "cfn_dwconv3d_bwd_fused_s2: A/B mismatch");
    CFN_REQUIRE(A == nullptr || (gA != nullptr && gB != nullptr), "cfn_dwconv3d_bwd_fused_s2: prologue needs gA, gB");
    CFN_REQUIRE(gsumsq == nullptr || y != nullptr, "cfn_dwconv3d_bwd_fused_s2: gsumsq needs y");
    hipStream_t st = (hipStream_t)stream;
    if (CPN(dw_flatb_s2_try)(gy, y, gsum, gsumsq, w, x, A, B, act, gx, gA, gB, gw, N, C, T, H, W, st, true) == 0) {
        // 14 -> 7: flat wave kernel (dwflatb.hip)
        CfnProfScope prof(CFN_K_DWCONV_BWD, st, (double)CP_ES * N * C * T * (2.0 * H * W + 49.0 * (y ? 2 : 1)));
        return CPN(dw_flatb_s2_try)(gy, y, gsum, gsumsq, w, x, A, B, act, gx, gA, gB, gw, N, C, T, H, W, st, false);
    }
    if (CPN(dw_cpb2x_try)(gy, y, gsum, gsumsq, w, x, A, B, act, gx, gA, gB, gw, N, C, T, H, W, st, true) == 0) {
        // 112 -> 56, 56 -> 28, 28 -> 14: one LDS image, x / a in registers (dwcpb2x.hip)
        CfnProfScope prof(CFN_K_DWCONV_BWD, st, (double)CP_ES * N * C * T * (2.0 * H * W + (double)(H / 2) * (W / 2) * (y ? 2 : 1)));
        return CPN(dw_cpb2x_try)(gy, y, gsum, gsumsq, w, x, A, B, act, gx, gA, gB, gw, N, C, T, H, W, st, false);
    }
    if (H != W || (H != 112 && H != 56 && H != 28)) return -1;
    const double po = (double)(H / 2) * (W / 2);
    CfnProfScope prof(CFN_K_DWCONV_BWD, st, (double)CP_ES * N * C * T * (2.0 * H * W + po * (y ? 2 : 1)));
    return CPN(dw_cpb2_try)(gy, y, gsum, gsumsq, w, x, A, B, act, gx, gA, gB, gw, N, C, T, H, W, st);
}
