// Fused backward of the stride-1 depthwise 3x3x3 conv on the 56x56 / 28x28 / 14x14 planes, fp32 tensors: the column-pair wave kernel of
// dwcpb.hip with ONE LDS image instead of three (round 4).
//
// dwcpb.hip stages three images per frame -- g' (window -> data gradient), a = act(A x + B) (window -> weight gradient) and x (centre ->
// gA, gB) -- because it forms the weight gradient as  gw[k] += g'(t)[centre] * a(t + kt - 1)[window].  Re-indexed by the position of a,
//     gw[k] += a(t)[centre] * g'(t - kt + 1, r - kh + 1, c - kw + 1)
// is the product of the lane's OWN a value with the very g' element the data gradient reads for the same tap (da(t) = sum_k w[k] g'(...)):
// one window read feeds both, a and x are only ever needed at the lane's own positions, so they are loaded straight into registers (8-byte
// loads per row of the lane's column pair, no LDS).  Per frame step: one image written instead of three, (HS + 2) x 4 LDS dwords read
// instead of 2 (HS + 2) x 4 + 2 HS.  These kernels run at the board's power limit: what is not moved is not paid for.
// Step f: G(f) is in the LDS image; for kt = 0, 1, 2 the output frame tau = f + 1 - kt takes  acc[kt] += wflip[kt] G-window  and
// gw[flip(kt, ., .)] += a(tau) G-window; frame f - 1 is complete afterwards (act' epilogue, gA / gB sums, store).  a / x of frames
// f - 1, f, f + 1 and the three accumulator sets rotate in registers.  A wave owns the weight-gradient terms of the a positions of ITS chunk
// (a is zero outside it).
// hipcc-flags: -fno-slp-vectorize
// fp32 or bf16 tensors (cp_io.h: compiled a second time through dwcpbx_bf16.hip; the LDS image, accumulators and every reduction stay fp32 / fp64).
#include "cp_io.h"
#include <stdint.h>
#include <stdlib.h>

#ifdef DW_BF16
#define DwCpbxArgs H16N(DwCpbxArgs)
#endif

struct DwCpbxArgs {
    const cpe_t* gy; const cpe_t* y; const double* gs; const double* gq; const float* w; const cpe_t* x;
    const double* A; const double* B; cpe_t* gx; double* gA; double* gB; double* gw;
    int N, C, T, act, TT, nchunks;
    long total_waves;
};

template <int W, int HS, int RG, int D, int OCC, bool HASY>
__global__ __launch_bounds__(256, OCC) void dw3d_cpx_bwd_kernel(const DwCpbxArgs a) {
    typedef float __attribute__((ext_vector_type(4))) f4;
    typedef float __attribute__((ext_vector_type(2))) p2;
    typedef unsigned __attribute__((ext_vector_type(2))) u2;
    constexpr int H = W, CP = W / 2;
    constexpr int BR = RG * HS, NB = (H + BR - 1) / BR;   // output rows per band, bands per plane (the last may be ragged)
    constexpr int IR = BR + 2;                        // image rows (band + halo)
    constexpr int XO = 4, PIT = W + 8;                // plane column 0 sits at image column XO (16-byte aligned rows)
    constexpr int IMG = IR * PIT;
    constexpr int NLD = (IR * W / 4 + 63) / 64;       // float4 loads per lane, frame and tensor
    constexpr int P = H * W, OOB = 0x7fff0000;
    constexpr int U = 2;                              // steps per loop trip: image parity and the register rings are static
    constexpr bool ROW4 = W % 4 == 0;                 // a float4 never straddles two rows
    static_assert(H % HS == 0 && CP * RG <= 64 && U % D == 0 && W % 2 == 0, "geometry");
    constexpr int WSZ = 2 * IMG + 8;                  // per wave: two g' images + a dump slot for the loader lanes without an element
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

    float wr[27];                                     // flipped taps for the data gradient
#pragma unroll
    for (int j = 0; j < 27; ++j) wr[j] = cfn_uni(a.w[(long)c * 27 + 26 - j]);
    const bool hasA = a.A != nullptr;
    const float pa = cfn_uni(hasA ? (float)a.A[nc] : 1.0f);
    const float pb = cfn_uni(hasA ? (float)a.B[nc] : 0.0f);
    const float act_lo = (hasA && a.act == CFN_ACT_RELU) ? 0.0f : -__builtin_inff();   // none / ReLU only (the planner checks)
    const float gsv = cfn_uni(a.gs ? (float)a.gs[nc] : 0.0f);
    const float gqv = cfn_uni((HASY && a.gq) ? 2.0f * (float)a.gq[nc] : 0.0f);

    for (int i = lane; i < 2 * IMG; i += 64) imG[i] = 0.0f;       // halos (and everything else) zero; wave-private

    // loader of g': the band's valid rows row_lo .. row_hi-1 are one contiguous run of the plane (same for gy and y)
    const int row_lo = max(band * BR - 1, 0), row_hi = min(band * BR + BR + 1, H);
    const int nel = (row_hi - row_lo) * W;
    int ldo[NLD], lo0[NLD], lo1[ROW4 ? 1 : NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int e0 = (k * 64 + lane) * 4;
        const bool on = e0 < nel;
        const int r0 = row_lo + e0 / W - (band * BR - 1), c0 = e0 % W;
        ldo[k] = on ? (row_lo * W + e0) * CP_ES : OOB;
        lo0[k] = on ? r0 * PIT + XO + c0 : -1;
        if (!ROW4) {
            const int r2 = row_lo + (e0 + 2) / W - (band * BR - 1), c2 = (e0 + 2) % W;
            lo1[k] = on ? r2 * PIT + XO + c2 : -1;
        }
    }
    // compute lane: row group g, column pair cp
    const int g = lane / CP, cp = lane - g * CP;
    const bool act_lane = g < RG && band * BR + g * HS < H;        // H % HS == 0: a row group is valid as a whole
    const int tofs = act_lane ? (g * HS) * PIT + (XO - 1) + 2 * cp : 0;
    const int yo = act_lane ? ((band * BR + g * HS) * W + 2 * cp) * CP_ES : OOB;

    __amdgpu_buffer_rsrc_t rg = cfn_rsrc(a.gy + nc * (long)T * P, (unsigned)((long)T * P * CP_ES));
    __amdgpu_buffer_rsrc_t ry = cfn_rsrc((HASY ? a.y : a.gy) + nc * (long)T * P, (unsigned)((long)T * P * CP_ES));
    __amdgpu_buffer_rsrc_t rx = cfn_rsrc(a.x + nc * (long)T * P, (unsigned)((long)T * P * CP_ES));
    __amdgpu_buffer_rsrc_t rd = cfn_rsrc(a.gx + nc * (long)T * P, (unsigned)((long)T * P * CP_ES));

    auto fetch1 = [&](__amdgpu_buffer_rsrc_t r, int f, f4 (&dst)[NLD]) {     // unconditional: an unwanted frame reads zeros
        const bool want = f >= 0 && f < T && f <= t1;
        const int so = cfn_uni(want ? f * P * CP_ES : 0);
#pragma unroll
        for (int k = 0; k < NLD; ++k) dst[k] = cp_ld4(r, want ? ldo[k] : OOB, so);
    };
    auto fetchx = [&](int f, p2 (&dst)[HS]) {         // the lane's own column pair of x(f), HS rows; frames outside the chunk read zeros
        const bool want = f >= t0 && f < t1;
        const int so = cfn_uni(want ? f * P * CP_ES : 0);
#pragma unroll
        for (int i = 0; i < HS; ++i) dst[i] = cp_ld2(rx, want ? yo + i * W * CP_ES : OOB, so);
    };
    // branch-free staging: a loader lane without an element writes into the wave's dump slot
    float* dump = imG + 2 * IMG;
    auto stageG = [&](int f, const f4 (&sg)[NLD], const f4 (&sy)[NLD], float* im) {   // g' = gy + gs + 2 y gq, zero outside the clip
        const float m = (f >= 0 && f < T && f <= t1) ? 1.0f : 0.0f;
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            f4 v = sg[k] + gsv;
            if (HASY) v += sy[k] * gqv;
            v *= m;
            if (ROW4) {
                *reinterpret_cast<f4*>(lo0[k] >= 0 ? im + lo0[k] : dump) = v;
            } else {
                *reinterpret_cast<p2*>(lo0[k] >= 0 ? im + lo0[k] : dump) = (p2){v.x, v.y};
                *reinterpret_cast<p2*>(lo1[k] >= 0 ? im + lo1[k] : dump + 4) = (p2){v.z, v.w};
            }
        }
    };
    auto wave_sync = [&]() {                          // LDS ops of a wave run in order; only the compiler has to be told
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    p2 acc[3][HS], av[3][HS], xv[3][HS];              // [0]: frame f + 1, [1]: f, [2]: f - 1
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int i = 0; i < HS; ++i) { acc[s][i] = (p2){0.0f, 0.0f}; av[s][i] = (p2){0.0f, 0.0f}; xv[s][i] = (p2){0.0f, 0.0f}; }
    float dwa[27];
#pragma unroll
    for (int j = 0; j < 27; ++j) dwa[j] = 0.0f;
    p2 s1p = {0.0f, 0.0f}, s2p = {0.0f, 0.0f};
    const float lane_m = act_lane ? 1.0f : 0.0f;

    // steps f = t0-1 .. t1.  Step j of a trip: G(f) is in imG[j & 1]; ring slot (j+1) % D of the gy / y rings holds frame f+1, slot j % D of
    // the x ring frame f+1; each slot is refilled right after it was consumed.
    const int f_first = t0 - 1, f_last = t1;
    f4 rgG[D][NLD], rgY[HASY ? D : 1][NLD];
    p2 rgX[D][HS];
    {
        f4 fg[NLD], fy[NLD];
        fetch1(rg, f_first, fg);
        if (HASY) fetch1(ry, f_first, fy);
#pragma unroll
        for (int d = 1; d <= D; ++d) { fetch1(rg, f_first + d, rgG[d % D]); if (HASY) fetch1(ry, f_first + d, rgY[d % D]); }
#pragma unroll
        for (int d = 0; d < D; ++d) fetchx(f_first + 1 + d, rgX[d]);
        wave_sync();
        stageG(f_first, fg, HASY ? fy : fg, imG);
    }
    for (int f0 = f_first; f0 <= f_last; f0 += U) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int f = f0 + j;
            const int pg = j & 1, pq = (j + 1) & 1;                         // image of G(f); image that takes G(f+1)
            stageG(f + 1, rgG[(j + 1) % D], rgY[HASY ? (j + 1) % D : 0], imG + pq * IMG);
            fetch1(rg, f + 1 + D, rgG[(j + 1) % D]);
            if (HASY) fetch1(ry, f + 1 + D, rgY[(j + 1) % D]);
            {   // x(f+1) arrives: a(f+1) (zero outside the chunk: the loads were not issued there, and act(B) is not 0)
                const float m = (f + 1 >= t0 && f + 1 < t1) ? lane_m : 0.0f;
#pragma unroll
                for (int i = 0; i < HS; ++i) {
                    const p2 x = rgX[j % D][i];
                    xv[0][i] = x;
                    av[0][i] = (p2){fmaxf(fmaf(x.x, pa, pb), act_lo) * m, fmaxf(fmaf(x.y, pa, pb), act_lo) * m};
                }
                fetchx(f + 1 + D, rgX[j % D]);
            }
            wave_sync();
            {
                const float* tp = imG + pg * IMG + tofs;
#pragma unroll
                for (int r = 0; r < HS + 2; ++r) {
                    const float* q = tp + r * PIT;
                    const float q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
#pragma unroll
                    for (int i = 0; i < HS; ++i) {
                        const int kh = r - i;
                        if (kh >= 0 && kh < 3) {
#pragma unroll
                            for (int kt = 0; kt < 3; ++kt) {               // output frame f + 1 - kt: set kt
                                const int tb = kt * 9 + kh * 3;
                                const float w0 = wr[tb], w1 = wr[tb + 1], w2 = wr[tb + 2];
                                acc[kt][i].x = fmaf(w0, q0, fmaf(w1, q1, fmaf(w2, q2, acc[kt][i].x)));
                                acc[kt][i].y = fmaf(w0, q1, fmaf(w1, q2, fmaf(w2, q3, acc[kt][i].y)));
                                const p2 aa = av[kt][i];                    // wr[tb + kw] = w[26 - tb - kw]: the weight-gradient slot of this tap
                                dwa[26 - tb] = fmaf(aa.x, q0, fmaf(aa.y, q1, dwa[26 - tb]));
                                dwa[25 - tb] = fmaf(aa.x, q1, fmaf(aa.y, q2, dwa[25 - tb]));
                                dwa[24 - tb] = fmaf(aa.x, q2, fmaf(aa.y, q3, dwa[24 - tb]));
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- emit gx(f-1): complete in set 2 ----
            const int to = f - 1;
            const bool emit = to >= t0 && to < t1;                         // wave uniform
            const int so = cfn_uni(emit ? to * P * CP_ES : 0);
            const float mf = emit ? lane_m : 0.0f;
            const int vo = emit ? yo : OOB;
#pragma unroll
            for (int i = 0; i < HS; ++i) {
                p2 v = acc[2][i];
                if (hasA) {                                                // wave uniform
                    p2 dz;
                    dz.x = av[2][i].x > act_lo ? v.x : 0.0f;               // act' of none / ReLU: a > 0 <=> z > 0
                    dz.y = av[2][i].y > act_lo ? v.y : 0.0f;
                    const p2 dm = dz * mf;
                    s1p = __builtin_elementwise_fma(dm, xv[2][i], s1p);
                    s2p += dm;
                    v = dz * pa;
                }
                cp_st2(v, rd, vo + i * W * CP_ES, so);
            }
#pragma unroll
            for (int i = 0; i < HS; ++i) {                                 // rotate: frame f+1 becomes frame f of the next step
                acc[2][i] = acc[1][i]; acc[1][i] = acc[0][i]; acc[0][i] = (p2){0.0f, 0.0f};
                av[2][i] = av[1][i]; av[1][i] = av[0][i];
                xv[2][i] = xv[1][i]; xv[1][i] = xv[0][i];
            }
            asm volatile("" : "+v"(s1p), "+v"(s2p));
        }
    }
    // ---- reductions: gw (27 per channel), then gA / gB ----
    // transpose-reduce: 32 values x 64 lanes -> one total per lane pair in 32 shuffles; lane l ends with the total of value (l >> 1)
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
        const float st1 = cfn_wave_sum(s1p.x + s1p.y), st2 = cfn_wave_sum(s2p.x + s2p.y);
        if (lane == 0) { cfn_add64(&a.gA[nc], (double)st1); cfn_add64(&a.gB[nc], (double)st2); }
    }
}

// returns -1 when the shape is not handled (caller goes on to dwcpb.hip / the band kernels); probe: 0 = handled, nothing launched
int CPN(dw_cpbx_try)(const cpe_t* gy, const cpe_t* y, const double* gs, const double* gq, const float* w, const cpe_t* x,
                const double* A, const double* B, int act, cpe_t* gx, double* gA, double* gB, double* gw,
                int N, int C, int T, int H, int W, hipStream_t st, bool probe) {
    // bit mask of the planes served: 1 = 56x56, 2 = 28x28, 4 = 14x14
    static const int enabled = getenv("CFN_DW_CPBX") ? atoi(getenv("CFN_DW_CPBX")) : 7;
    static const int tt_env = getenv("CFN_DW_CPB_TT") ? atoi(getenv("CFN_DW_CPB_TT")) : 0;
    if (H != W || (H != 56 && H != 28 && H != 14)) return -1;
    if (!(enabled & (H == 56 ? 1 : H == 28 ? 2 : 4))) return -1;
    if (A != nullptr && act != CFN_ACT_NONE && act != CFN_ACT_RELU) return -1;      // act' from the sign of a: none / ReLU (every X3D conv2)
    if ((long)T * H * W * CP_ES >= 0x7fff0000L) return -1;
    if ((((uintptr_t)gy | (uintptr_t)x | (uintptr_t)gx | (uintptr_t)(y ? y : gy)) & (4 * CP_ES - 1)) != 0) return -1;
    if (probe) return 0;
    const bool hasy = y != nullptr && gq != nullptr;
    DwCpbxArgs a = {gy, hasy ? y : nullptr, gs, hasy ? gq : nullptr, w, x, A, B, gx, A ? gA : nullptr, A ? gB : nullptr, gw, N, C, T, act, 0, 0, 0};
    const int NB = H == 56 ? 14 : H == 28 ? 7 : 1;                                    // 14x14: the plane is one band
    // t-chunks as in dwcpb.hip: ~64 frames (a wave's fixed cost -- LDS clear, pipeline fill, the 27-value reduction -- is worth ~4 frame steps)
    const long units = (long)N * C * NB;
    long nch = (T + 32) / 64;
    if (nch < 1) nch = 1;
    while (units * nch < 2L * 256 * 12 && (T + nch) / (nch + 1) >= 16) ++nch;
    int TT = (int)((T + nch - 1) / nch);
    if (tt_env > 0) TT = tt_env;
    if (TT > T) TT = T;
    a.TT = TT;
    a.nchunks = (T + TT - 1) / TT;
    a.total_waves = units * a.nchunks;
    const unsigned blocks = (unsigned)((a.total_waves + 3) / 4);
#define CFN_CPBX_GO(...) do { if (hasy) hipLaunchKernelGGL((dw3d_cpx_bwd_kernel<__VA_ARGS__, true>), dim3(blocks), dim3(256), 0, st, a); \
                              else hipLaunchKernelGGL((dw3d_cpx_bwd_kernel<__VA_ARGS__, false>), dim3(blocks), dim3(256), 0, st, a); } while (0)
    // (deeper rings, 5 waves per SIMD on 28x28, one-row lanes on 56x56 / 14x14: all within the +-4 % run-to-run spread of these power-bound kernels)
    if (H == 56) CFN_CPBX_GO(56, 2, 2, 1, 3);
    else if (H == 28) CFN_CPBX_GO(28, 1, 4, 1, 4);
    else CFN_CPBX_GO(14, 2, 7, 2, 3);
#undef CFN_CPBX_GO
    return cfn_check_launch("dwconv3d column-pair fused backward (one image)");
}
