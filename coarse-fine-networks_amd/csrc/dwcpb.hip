// Fused backward of the stride-1 depthwise 3x3x3 conv (data gradient AND weight gradient in one pass over gy, y, x) on the
// square planes 56x56 / 28x28 / 14x14 / 7x7 (conv2 of X3D layers 1-4, x3d_fine.py:89-97,171-201), fp32 or bf16 tensors
// (cp_io.h) -- column-pair wave kernel, the backward counterpart of dwcp.hip.
//
// Why: run separately the two gradients move 7 tensor passes (dgrad: gy, y, x -> gx; wgrad: gy, y, x), fused 5.  The band
// kernel dw3d_bwd_fused_kernel (dwconv3d.hip) does fuse them, but with 7 / 4 output rows per lane it needs 211 / 163 VGPRs
// (2-3 waves per SIMD, one workgroup barrier per frame) and runs at 2.5-2.9 TB/s; it spills on 14x14, which therefore used the
// two separate kernels (profiles/r02_microbench_b8.txt).  dwcp.hip showed what these kernels respond to: independent waves,
// few rows per lane, as many waves per SIMD as the registers allow (here 3-4).  Same skeleton here: one WAVE per (sample, channel, t-chunk, row band), a lane owns
// two adjacent columns x HS (1-2) rows, three wave-private LDS images per frame parity, no workgroup barrier:
//   G image  g'(f)   = gy + gs + 2 y gq     window -> data gradient (flipped taps), centre -> weight gradient
//   A image  a(f-1)  = act(A x + B)         window -> weight gradient; centre > 0 = act' of the ReLU prologue
//   X image  x(f-1)                         centre -> the prologue-coefficient gradients (gA += dz x, gB += dz)
// At step f:  gx accumulators += flipped taps * G-window(f);  gw[kt] += g'(f-kt)[centre] * A-window(f-1) with the three g'
// centres and the three gx accumulator sets rotating in registers; output frame f-1 is finished with the act' epilogue.  The forward is
// y(t) = sum_kt w[kt] a(t+kt-1), so gw[kt] = sum_t g'(t) a(t+kt-1): with the A frame fa = f-1, t = f - kt.
// The 27 weight-gradient sums of a lane are scalar (54 registers as pairs would cost a wave per SIMD); they are reduced over
// the wave at the end and added with fp64 atomics like everywhere else.
// 7x7 (odd width): dword loads, the last column pair has one column, 28 of 64 lanes busy.
// Prologue activations other than none / ReLU and other planes keep the band kernels (dw_cpb_try returns -1).
// hipcc-flags: -fno-slp-vectorize
// (the SLP vectoriser re-pairs the scalar weight-gradient FMAs into v_pk_fma_f32 with ~300 pair-building moves per trip and
// pushes the kernel into spills; measured with and without)
#include "cp_io.h"
#include <stdint.h>
#include <stdlib.h>

#ifdef DW_BF16
#define DwCpbArgs H16N(DwCpbArgs)
#endif
struct DwCpbArgs {
    const cpe_t* gy; const cpe_t* y; const double* gs; const double* gq; const float* w; const cpe_t* x;
    const double* A; const double* B; cpe_t* gx; double* gA; double* gB; double* gw;
    int N, C, T, act, TT, nchunks;
    long total_waves;
};

template <int W, int HS, int RG, int D, int OCC, bool HASY>
__global__ __launch_bounds__(256, OCC) void dw3d_cp_bwd_kernel(const DwCpbArgs a) {
    typedef float __attribute__((ext_vector_type(4))) f4;
    typedef float __attribute__((ext_vector_type(2))) p2;
    typedef unsigned __attribute__((ext_vector_type(2))) u2;
    constexpr int H = W, CP = (W + 1) / 2;            // column pairs per row (odd width: the last pair has one column)
    constexpr int LV = W % 2 == 0 ? 4 : 1;            // elements per loader lane (odd width: frames are not 16-byte aligned)
    constexpr int BR = RG * HS, NB = (H + BR - 1) / BR;   // output rows per band, bands per plane (the last may be ragged)
    constexpr int IR = BR + 2;                        // image rows (band + halo)
    constexpr int XO = 4, PIT = W + 8;                // plane column 0 sits at image column XO (16-byte aligned rows)
    constexpr int IMG = IR * PIT;
    constexpr int NLD = (IR * W / LV + 63) / 64;      // loads per lane, frame and tensor
    constexpr int P = H * W, OOB = 0x7fff0000;
    constexpr int U = 2;                              // steps per loop trip: LDS image parity and the register ring are static;
                                                      // the accumulator sets are rotated with moves (24 of ~350 instructions per step:
                                                      // renaming them needs a 6-step trip, which hipcc allocates at 168 VGPRs + spills)
    constexpr bool ROW4 = W % 4 == 0;                 // a float4 never straddles two rows
    static_assert(H % HS == 0 && CP * RG <= 64 && U % D == 0, "geometry");
    constexpr int WSZ = 6 * IMG + 8;                   // per wave: six images + a dump slot for the loader lanes without an element
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
    float* imG = smem + wv * WSZ;                     // [2] g'
    float* imA = imG + 2 * IMG;                       // [2] act(A x + B)
    float* imX = imA + 2 * IMG;                       // [2] x

    float wr[27];                                     // flipped taps for the data gradient
#pragma unroll
    for (int j = 0; j < 27; ++j) wr[j] = cfn_uni(a.w[(long)c * 27 + 26 - j]);
    const bool hasA = a.A != nullptr;
    const float pa = cfn_uni(hasA ? (float)a.A[nc] : 1.0f);
    const float pb = cfn_uni(hasA ? (float)a.B[nc] : 0.0f);
    const float act_lo = (hasA && a.act == CFN_ACT_RELU) ? 0.0f : -__builtin_inff();   // none / ReLU only (the planner checks)
    const float gsv = cfn_uni(a.gs ? (float)a.gs[nc] : 0.0f);
    const float gqv = cfn_uni((HASY && a.gq) ? 2.0f * (float)a.gq[nc] : 0.0f);

    for (int i = lane; i < 6 * IMG; i += 64) imG[i] = 0.0f;       // halos (and everything else) zero; wave-private

    // loader: the band's valid input rows row_lo .. row_hi-1 are one contiguous run of the plane (same for gy, y, x)
    const int row_lo = max(band * BR - 1, 0), row_hi = min(band * BR + BR + 1, H);
    const int nel = (row_hi - row_lo) * W;
    int ldo[NLD], lo0[NLD], lo1[(ROW4 || LV == 1) ? 1 : NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int e0 = (k * 64 + lane) * LV;
        const bool on = e0 < nel;
        const int r0 = row_lo + e0 / W - (band * BR - 1), c0 = e0 % W;
        ldo[k] = on ? (row_lo * W + e0) * CP_ES : OOB;
        lo0[k] = on ? r0 * PIT + XO + c0 : -1;
        if (!ROW4 && LV == 4) {
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
        for (int k = 0; k < NLD; ++k) {
            if (LV == 4) dst[k] = cp_ld4(r, want ? ldo[k] : OOB, so);
            else dst[k] = (f4){cp_ld1(r, want ? ldo[k] : OOB, so), 0.0f, 0.0f, 0.0f};
        }
    };
    // branch-free staging: a loader lane without an element writes into the wave's dump slot (no exec-mask branches in the
    // frame loop: straight-line code schedules and allocates far better)
    float* dump = imG + 6 * IMG;
    auto put = [&](float* im, int k, f4 v) {
        if (LV == 1) {
            *(lo0[k] >= 0 ? im + lo0[k] : dump) = v.x;
        } else if (ROW4) {
            *reinterpret_cast<f4*>(lo0[k] >= 0 ? im + lo0[k] : dump) = v;
        } else {
            *reinterpret_cast<p2*>(lo0[k] >= 0 ? im + lo0[k] : dump) = (p2){v.x, v.y};
            *reinterpret_cast<p2*>(lo1[k] >= 0 ? im + lo1[k] : dump + 4) = (p2){v.z, v.w};
        }
    };
    auto stageG = [&](int f, const f4 (&sg)[NLD], const f4 (&sy)[NLD], float* im) {   // g' = gy + gs + 2 y gq, zero outside
        const float m = (f >= 0 && f < T && f <= t1) ? 1.0f : 0.0f;
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            f4 v = sg[k] + gsv;
            if (HASY) v += sy[k] * gqv;
            put(im, k, v * m);
        }
    };
    auto stageAX = [&](int f, const f4 (&sx)[NLD], float* ia, float* ix) {   // a = act(A x + B) (zero outside the chunk's frames), x
        const float m = (f >= 0 && f < T && f >= t0 - 1 && f <= t1) ? 1.0f : 0.0f;
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const f4 x = sx[k];
            f4 v;
            v.x = fmaxf(fmaf(x.x, pa, pb), act_lo) * m; v.y = fmaxf(fmaf(x.y, pa, pb), act_lo) * m;
            v.z = fmaxf(fmaf(x.z, pa, pb), act_lo) * m; v.w = fmaxf(fmaf(x.w, pa, pb), act_lo) * m;
            put(ia, k, v);
            put(ix, k, x);
        }
    };
    auto wave_sync = [&]() {                          // LDS ops of a wave run in order; only the compiler has to be told
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    p2 acc[3][HS], gc[3][HS];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int i = 0; i < HS; ++i) { acc[s][i] = (p2){0.0f, 0.0f}; gc[s][i] = (p2){0.0f, 0.0f}; }
    float dwa[27];
#pragma unroll
    for (int j = 0; j < 27; ++j) dwa[j] = 0.0f;
    p2 s1p = {0.0f, 0.0f}, s2p = {0.0f, 0.0f};
    const float lane_m = act_lane ? 1.0f : 0.0f;
    const bool col2 = 2 * cp + 1 < W;                              // the pair's second column exists (odd width: not in the last pair)
    const p2 lane_m2 = {lane_m, col2 ? lane_m : 0.0f};

    // steps f = t0-1 .. t1+1.  Step j of a trip: G(f) is in imG[j & 1], A / X(f-1) in im?[(j+1) & 1]; ring slot (j+1) % D of the
    // gy / y rings holds frame f+1, slot j % D of the x ring frame f; each slot is refilled right after it was staged.
    const int f_first = t0 - 1, f_last = t1 + 1;
    f4 rgG[D][NLD], rgY[HASY ? D : 1][NLD], rgX[D][NLD];
    {
        f4 fg[NLD], fy[NLD];
        fetch1(rg, f_first, fg);
        if (HASY) fetch1(ry, f_first, fy);
#pragma unroll
        for (int d = 1; d <= D; ++d) { fetch1(rg, f_first + d, rgG[d % D]); if (HASY) fetch1(ry, f_first + d, rgY[d % D]); }
#pragma unroll
        for (int d = 0; d < D; ++d) fetch1(rx, f_first + d, rgX[d]);
        wave_sync();
        stageG(f_first, fg, HASY ? fy : fg, imG);
    }
    for (int f0 = f_first; f0 <= f_last; f0 += U) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int f = f0 + j;
            const int pg = j & 1, pq = (j + 1) & 1;                         // image of G(f); image of A / X(f-1)
            stageG(f + 1, rgG[(j + 1) % D], rgY[HASY ? (j + 1) % D : 0], imG + pq * IMG);
            fetch1(rg, f + 1 + D, rgG[(j + 1) % D]);
            if (HASY) fetch1(ry, f + 1 + D, rgY[(j + 1) % D]);
            stageAX(f, rgX[j % D], imA + pg * IMG, imX + pg * IMG);
            fetch1(rx, f + D, rgX[j % D]);
            wave_sync();
            // ---- data gradient from the G window; g' centres of frame f for the weight gradient ------------------------
            const bool inchunk = f >= t0 && f < t1;                         // g'(f) is a weight-gradient term only inside the chunk
            const float cm = inchunk ? 1.0f : 0.0f;
            {
                const float* tp = imG + pg * IMG + tofs;
#pragma unroll
                for (int r = 0; r < HS + 2; ++r) {
                    const float* q = tp + r * PIT;
                    const float q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
                    if (r >= 1 && r <= HS) gc[0][r - 1] = (p2){q1 * cm, q2 * cm};
#pragma unroll
                    for (int i = 0; i < HS; ++i) {
                        const int kh = r - i;
                        if (kh >= 0 && kh < 3) {
#pragma unroll
                            for (int kt = 0; kt < 3; ++kt) {
                                const int sl = kt;                         // accumulator of output frame f + 1 - kt
                                // scalar FMAs with the weight as an SGPR operand: as (w, w) pairs for v_pk_fma_f32 the 27 taps
                                // take 54 SGPRs and the kernel starts spilling SGPRs
                                const float w0 = wr[kt * 9 + kh * 3 + 0], w1 = wr[kt * 9 + kh * 3 + 1], w2 = wr[kt * 9 + kh * 3 + 2];
                                acc[sl][i].x = fmaf(w0, q0, fmaf(w1, q1, fmaf(w2, q2, acc[sl][i].x)));
                                acc[sl][i].y = fmaf(w0, q1, fmaf(w1, q2, fmaf(w2, q3, acc[sl][i].y)));
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- weight gradient: A window of frame f-1 against g'(f), g'(f-1), g'(f-2) ---------------------------------
            p2 ac[HS];
            {
                const float* tp = imA + pq * IMG + tofs;
#pragma unroll
                for (int r = 0; r < HS + 2; ++r) {
                    const float* q = tp + r * PIT;
                    const float q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
                    const float qq[4] = {q0, q1, q2, q3};
                    if (r >= 1 && r <= HS) ac[r - 1] = (p2){q1, q2};
#pragma unroll
                    for (int i = 0; i < HS; ++i) {
                        const int kh = r - i;
                        if (kh >= 0 && kh < 3) {
#pragma unroll
                            for (int kt = 0; kt < 3; ++kt) {
                                const p2 gg = gc[kt][i];                    // g'(f - kt) at the lane's two columns of row i
#pragma unroll
                                for (int kw = 0; kw < 3; ++kw)
                                    dwa[kt * 9 + kh * 3 + kw] = fmaf(gg.x, qq[kw], fmaf(gg.y, qq[kw + 1], dwa[kt * 9 + kh * 3 + kw]));
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- emit gx(f-1): complete in set 2 ----------------------------------------------------------------------
            const int se = 2;
            const int to = f - 1;
            const bool emit = to >= t0 && to < t1;                         // wave uniform
            const int so = cfn_uni(emit ? to * P * CP_ES : 0);
            const p2 mf = emit ? lane_m2 : (p2){0.0f, 0.0f};
            const int vo = emit ? yo : OOB;
            const float* tx = imX + pq * IMG + tofs;
#pragma unroll
            for (int i = 0; i < HS; ++i) {
                p2 v = acc[se][i];
                if (hasA) {                                                // wave uniform
                    const p2 xe = {tx[(i + 1) * PIT + 1], tx[(i + 1) * PIT + 2]};
                    p2 dz;
                    dz.x = ac[i].x > act_lo ? v.x : 0.0f;                  // act' of none / ReLU: a > 0 <=> z > 0
                    dz.y = ac[i].y > act_lo ? v.y : 0.0f;
                    const p2 dm = dz * mf;
                    s1p = __builtin_elementwise_fma(dm, xe, s1p);
                    s2p += dm;
                    v = dz * pa;
                }
                if (LV == 4) {
                    cp_st2(v, rd, vo + i * W * CP_ES, so);
                } else {                                                   // odd width: rows are only 4-byte aligned
                    float v0 = v.x, v1 = v.y;
                    asm volatile("" : "+v"(v0), "+v"(v1));                 // (hipcc 7.2 otherwise stores v.x twice: it reuses v.y's register
                                                                           //  for the second address before the store has read it)
                    cp_st1(v0, rd, vo + i * W * CP_ES, so);
                    cp_st1(v1, rd, col2 ? vo + i * W * CP_ES + CP_ES : OOB, so);
                }
            }
#pragma unroll
            for (int i = 0; i < HS; ++i) {                                 // rotate: frame f+1 becomes frame f of the next step
                acc[2][i] = acc[1][i]; acc[1][i] = acc[0][i]; acc[0][i] = (p2){0.0f, 0.0f};
                gc[2][i] = gc[1][i]; gc[1][i] = gc[0][i];
            }
            // hipcc otherwise sinks the statistics of all U steps to the end of the trip and spills the rows they read
            asm volatile("" : "+v"(s1p), "+v"(s2p));
        }
    }
    // ---- reductions: gw (27 per channel), then gA / gB ---------------------------------------------------------------
    // transpose-reduce: 32 values x 64 lanes -> one total per lane pair in 16 + 8 + 4 + 2 + 1 + 1 = 32 shuffles (a wave sum per
    // value would be 27 x 6); at the end lane l holds the total of value (l >> 1), and 27 lanes issue ONE atomic instruction
    {
        float v[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = (j < 27 && act_lane) ? dwa[j] : 0.0f;
#pragma unroll
        for (int st = 0; st < 5; ++st) {                                   // lane bit 5 - st selects the half of the values it keeps
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
        const int idx = lane >> 1;                                         // value index: bits 5..1 of the lane, MSB first
        if ((lane & 1) == 0 && idx < 27) cfn_add64(&a.gw[(long)c * 27 + idx], (double)tot);
    }
    if (hasA && a.gA) {
        const float st1 = cfn_wave_sum(s1p.x + s1p.y), st2 = cfn_wave_sum(s2p.x + s2p.y);
        if (lane == 0) { cfn_add64(&a.gA[nc], (double)st1); cfn_add64(&a.gB[nc], (double)st2); }
    }
}

// returns -1 when the shape is not handled (caller goes on to the band kernels); otherwise the launch status
int CPN(dw_cpb_try)(const cpe_t* gy, const cpe_t* y, const double* gs, const double* gq, const float* w, const cpe_t* x,
               const double* A, const double* B, int act, cpe_t* gx, double* gA, double* gB, double* gw,
               int N, int C, int T, int H, int W, hipStream_t st, bool probe) {
    // bit mask of the planes served: 1 = 56x56, 2 = 28x28, 4 = 14x14, 8 = 7x7
    static const int enabled = getenv("CFN_DW_CPB") ? atoi(getenv("CFN_DW_CPB")) : 15;
    static const int tt_env = getenv("CFN_DW_CPB_TT") ? atoi(getenv("CFN_DW_CPB_TT")) : 0;
    if (H != W || (H != 56 && H != 28 && H != 14 && H != 7)) return -1;
    if (!(enabled & (H == 56 ? 1 : H == 28 ? 2 : H == 14 ? 4 : 8))) return -1;
    if (A != nullptr && act != CFN_ACT_NONE && act != CFN_ACT_RELU) return -1;      // act' from the sign of a: none / ReLU (every X3D conv2)
    if ((long)T * H * W * CP_ES >= 0x7fff0000L) return -1;
    if (H != 7 && (((uintptr_t)gy | (uintptr_t)x | (uintptr_t)gx | (uintptr_t)(y ? y : gy)) & (4 * CP_ES - 1)) != 0) return -1;
    if (probe) return 0;
    const bool hasy = y != nullptr && gq != nullptr;
    DwCpbArgs a = {gy, hasy ? y : nullptr, gs, hasy ? gq : nullptr, w, x, A, B, gx, A ? gA : nullptr, A ? gB : nullptr, gw, N, C, T, act, 0, 0, 0};
    const int NB = H == 56 ? 14 : H == 28 ? 7 : 1;                                    // 14x14 and 7x7: the plane is one band
    // t-chunks: >= ~6 rounds of the chip's resident waves (16 per CU)
    // t-chunks of ~52 frames: a wave's fixed cost (LDS clear, pipeline fill, the 27-value reduction) is worth ~4 frame steps
    // (measured, 8 clips x T=256, 56x56: chunks of 9 / 21 / 33 / 63 frames: 4.5 / 2.3 / 1.7 / 1.3 ms); more chunks only while
    // the grid has fewer than ~2 rounds of the resident waves (12 per CU)
    const long units = (long)N * C * NB;
    long nch = (T + 32) / 64;                                                       // (52 -> 64 frames: same-box sweep 16 / 24 / 32 / 48 / 52 / 64: 56x56 1.39 / 1.29 / 1.23 / 1.26 / 1.22 / 1.20 ms,
    if (nch < 1) nch = 1;                                                           //  28x28 0.76 / 0.70 / 0.64 / 0.65 / 0.64 / 0.61, 14x14 0.34 / 0.32 / 0.32 / 0.33 / 0.32 / 0.31, 7x7 0.37 / 0.34 / 0.32 / 0.33 / 0.31 / 0.30)
    while (units * nch < 2L * 256 * 12 && (T + nch) / (nch + 1) >= 16) ++nch;
    int TT = (int)((T + nch - 1) / nch);
    if (tt_env > 0) TT = tt_env;
    if (TT > T) TT = T;
    a.TT = TT;
    a.nchunks = (T + TT - 1) / TT;
    a.total_waves = units * a.nchunks;
    const unsigned blocks = (unsigned)((a.total_waves + 3) / 4);
#define CFN_CPB_GO(...) do { if (hasy) hipLaunchKernelGGL((dw3d_cp_bwd_kernel<__VA_ARGS__, true>), dim3(blocks), dim3(256), 0, st, a); \
                             else hipLaunchKernelGGL((dw3d_cp_bwd_kernel<__VA_ARGS__, false>), dim3(blocks), dim3(256), 0, st, a); } while (0)
    // (one row per lane fits 128 VGPRs = 4 waves per SIMD on 28x28 and 7x7: 0.67 -> 0.62 and 0.34 -> 0.31 ms; on 56x56 and
    //  14x14 a one-row variant at 4 waves measured 10-15 % SLOWER than two rows at 3 waves: more halo rows and LDS reads per output)
    if (H == 56) CFN_CPB_GO(56, 2, 2, 1, 3);
    else if (H == 28) CFN_CPB_GO(28, 1, 4, 1, 4);
    else if (H == 14) CFN_CPB_GO(14, 2, 7, 2, 3);
    else CFN_CPB_GO(7, 1, 7, 2, 4);                                                  // 4 column pairs x 7 rows = 28 lanes
#undef CFN_CPB_GO
    return cfn_check_launch("dwconv3d column-pair fused backward");
}
