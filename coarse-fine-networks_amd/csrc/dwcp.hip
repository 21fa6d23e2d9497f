// Depthwise 3x3x3 forward, stride 1 and 2, onto the even square output planes of X3D at 224x224 input (56x56, 28x28, 14x14:
// conv2 of layers 1-3 and the first block of layers 1-3; x3d_fine.py:89-97,171-201), fp32 or bf16 tensors (cp_io.h) --
// COLUMN-PAIR kernel: one WAVE per (sample, channel, t-chunk, row band), no
// workgroup barrier.
//
// Why: the band kernel of dwconv3d.hip gives a lane 7 vertically adjacent outputs of ONE column: 21 accumulators, operand
// pairs for the packed FMAs built with register moves, 109-248 VGPRs (2-4 waves per SIMD) and one workgroup barrier per
// frame; the wave-per-channel kernel of dwsmall.hip showed on 14x14 / 7x7 that independent waves and occupancy are what these
// kernels respond to.  Here a lane owns TWO ADJACENT COLUMNS x HS (1-2) rows:
//   * the two outputs of a row share every tap's weight, so each tap is ONE v_pk_fma_f32 on a natural register pair (inputs
//     (c+kw-1, c+kw) -- stride 2: (2c+kw-1, 2c+kw+1) -- straight out of LDS, the wave-uniform weight an SGPR operand): 27 packed FMAs per output pair and no
//     pair-building moves (issue rates measured with tools/probe/mfma_rate_probe.hip at 4 waves per SIMD: v_fma_f32 2.9,
//     v_pk_fma_f32 4.9 cycles per wave instruction; v_mfma_f32_4x4x1 at 9.8 cycles for 192 useful FMAs is no faster than
//     the vector ALU -- an outer-product MFMA formulation of this conv was built and measured 1.3-1.7x SLOWER);
//   * an input row is 4 (stride 2: 5) LDS dwords per lane for its 2 x 3 taps;
//   * the three rolling accumulator sets (temporal taps) are renamed instead of moved: the frame loop is unrolled U = 6 (12)
//     steps, a multiple of 3 (accumulator roles), 2 (LDS image parity) and D (register ring of prefetched frames);
//   * results leave as one 8-byte store per row; 93-97 VGPRs, 5 waves per SIMD, no barrier.
// A band's input rows are one contiguous run of the (n, c, t) plane: every frame is NLD coalesced float4 loads per lane
// (unconditional buffer loads: exact vmcnt waits), staged through a wave-private LDS image with a zero halo.  The four waves
// of a workgroup are the bands of one (n, c, t-chunk) (56x56) or neighbouring t-chunks of one channel: halo rows / frames
// come out of L2.  Shapes / activations outside this list use the band kernel (dw_cp_fwd_try returns -1).
#include "cp_io.h"
#include <stdint.h>
#include <stdlib.h>

#ifdef DW_BF16
#define DwCpArgs H16N(DwCpArgs)
#endif
struct DwCpArgs {
    const cpe_t* x; const double* A; const double* B; const float* w; cpe_t* y; double* s1; double* s2;
    int N, C, T, act, TT, nchunks;
    long total_waves;
};

template <int W, int S, int HS, int RG, int D, int OCC>     // W: OUTPUT width (square planes), S: spatial stride
__global__ __launch_bounds__(256, OCC) void dw3d_cp_fwd_kernel(const DwCpArgs a) {
    typedef float __attribute__((ext_vector_type(4))) f4;
    typedef float __attribute__((ext_vector_type(2))) p2;
    typedef unsigned __attribute__((ext_vector_type(2))) u2;
    typedef unsigned __attribute__((ext_vector_type(4))) u4;
    constexpr int H = W, CP = W / 2;                  // output plane, output column pairs per row
    constexpr int WI = W * S, HI = H * S;             // input plane (even input sizes: stride 2 needs no bottom / right halo)
    constexpr int BR = RG * HS, NB = (H + BR - 1) / BR;   // output rows per band, bands per plane (the last may be ragged)
    constexpr int IR = (BR - 1) * S + 3;              // image rows (the band's input rows + halo)
    constexpr int XO = 4, PIT = WI + 8;               // input column 0 sits at image column XO (16-byte aligned rows)
    constexpr int IMG = IR * PIT;
    constexpr int NLD = (IR * WI / 4 + 63) / 64;      // float4 loads per lane and frame
    constexpr int P = HI * WI, PO = H * W, OOB = 0x7fff0000;     // + row offsets stays below 2^31
    constexpr int U = D == 4 ? 12 : 6;                // steps per loop trip: multiple of 3, 2 and D
    constexpr bool ROW4 = WI % 4 == 0;                // a float4 never straddles two rows
    static_assert(H % HS == 0 && CP * RG <= 64 && U % D == 0 && W % 2 == 0, "geometry");
    __shared__ __attribute__((aligned(16))) float smem[4 * 2 * IMG];

    const int lane = threadIdx.x & 63, wv = cfn_uni((int)(threadIdx.x >> 6));
    const unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    // wave-uniform by construction; stated for the compiler (see cfn_uni)
    const long widx = cfn_uni((long)L * 4 + wv);
    if (widx >= a.total_waves) return;                // whole waves only: no barrier anywhere below
    const int band = cfn_uni((int)(widx % NB));
    const long rest = cfn_uni((long)(widx / NB));
    const int chunk = cfn_uni((int)(rest % a.nchunks));
    const long nc = cfn_uni((long)(rest / a.nchunks));
    const int c = cfn_uni((int)(nc % a.C));
    const int T = a.T, t0 = chunk * a.TT, t1 = min(t0 + a.TT, T);
    float* img = smem + wv * 2 * IMG;

    float wr[27];
#pragma unroll
    for (int j = 0; j < 27; ++j) wr[j] = cfn_uni(a.w[(long)c * 27 + j]);
    const float pa = cfn_uni(a.A ? (float)a.A[nc] : 1.0f);
    const float pb = cfn_uni(a.A ? (float)a.B[nc] : 0.0f);
    const float act_lo = a.act == CFN_ACT_RELU ? 0.0f : -__builtin_inff();      // none / ReLU only (the planner checks)

    for (int i = lane; i < 2 * IMG; i += 64) img[i] = 0.0f;       // halo (and everything else) zero; wave-private

    // loader: the band's valid input rows row_lo .. row_hi-1 are one contiguous run of the plane
    const int row_lo = max(band * BR * S - 1, 0), row_hi = min(band * BR * S - 1 + IR, HI);
    const int nel = (row_hi - row_lo) * WI;
    int ldo[NLD], lo0[NLD], lo1[ROW4 ? 1 : NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int e0 = (k * 64 + lane) * 4;
        const bool on = e0 < nel;
        const int r0 = row_lo + e0 / WI - (band * BR * S - 1), c0 = e0 % WI;
        ldo[k] = on ? (row_lo * WI + e0) * CP_ES : OOB;
        lo0[k] = on ? r0 * PIT + XO + c0 : 0;
        if (!ROW4) {
            const int r2 = row_lo + (e0 + 2) / WI - (band * BR * S - 1), c2 = (e0 + 2) % WI;
            lo1[k] = on ? r2 * PIT + XO + c2 : 0;
        }
    }
    // compute lane: row group g, column pair cp
    const int g = lane / CP, cp = lane - g * CP;
    const bool act_lane = g < RG && band * BR + g * HS < H;        // H % HS == 0: a row group is valid as a whole
    const float* tb = img + (act_lane ? (g * HS * S) * PIT + (XO - 1) + 2 * S * cp : 0);
    const int yo = act_lane ? ((band * BR + g * HS) * W + 2 * cp) * CP_ES : OOB;

    __amdgpu_buffer_rsrc_t rx = cfn_rsrc(a.x + nc * (long)T * P, (unsigned)((long)T * P * CP_ES));
    __amdgpu_buffer_rsrc_t ry = cfn_rsrc(a.y + nc * (long)T * PO, (unsigned)((long)T * PO * CP_ES));

    auto fetch = [&](int f, f4 (&dst)[NLD]) {        // unconditional: an unwanted frame reads nothing (zeros)
        const bool want = f >= 0 && f < T && f <= t1;
        const int so = cfn_uni(want ? f * P * CP_ES : 0);
#pragma unroll
        for (int k = 0; k < NLD; ++k) dst[k] = cp_ld4(rx, want ? ldo[k] : OOB, so);
    };
    auto stage = [&](int f, const f4 (&src)[NLD], float* im) {   // frames outside the clip are zero AFTER the prologue
        const float m = (f >= 0 && f < T) ? 1.0f : 0.0f;
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            if (ldo[k] != OOB) {
                f4 v = src[k];
                v.x = fmaxf(fmaf(v.x, pa, pb), act_lo) * m; v.y = fmaxf(fmaf(v.y, pa, pb), act_lo) * m;
                v.z = fmaxf(fmaf(v.z, pa, pb), act_lo) * m; v.w = fmaxf(fmaf(v.w, pa, pb), act_lo) * m;
                if (ROW4) {
                    *reinterpret_cast<f4*>(im + lo0[k]) = v;
                } else {
                    *reinterpret_cast<p2*>(im + lo0[k]) = (p2){v.x, v.y};
                    *reinterpret_cast<p2*>(im + lo1[k]) = (p2){v.z, v.w};
                }
            }
        }
    };
    auto wave_sync = [&]() {                          // LDS ops of a wave run in order; only the compiler has to be told
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    p2 acc[3][HS];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int i = 0; i < HS; ++i) acc[s][i] = (p2){0.0f, 0.0f};
    p2 s1p = {0.0f, 0.0f}, s2p = {0.0f, 0.0f};
    const float lane_m = act_lane ? 1.0f : 0.0f;

    // input frames t0-1 .. t1; step f consumes input frame f (LDS image f & 1 relative to the first frame) and finishes output
    // frame f-1.  Ring slot (j+1) % D holds frame f+1 at step j; it is refilled with frame f+1+D right after it was staged.
    const int f_first = t0 - 1, f_last = t1;
    f4 ring[D][NLD];
    {
        f4 first[NLD];
        fetch(f_first, first);
#pragma unroll
        for (int d = 1; d <= D; ++d) fetch(f_first + d, ring[d % D]);
        wave_sync();
        stage(f_first, first, img);
    }
    for (int f0 = f_first; f0 <= f_last; f0 += U) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int f = f0 + j;
            const int par = j & 1;
            stage(f + 1, ring[(j + 1) % D], img + (par ^ 1) * IMG);
            fetch(f + 1 + D, ring[(j + 1) % D]);
            wave_sync();
            const float* tp = tb + par * IMG;
            // input rows one ahead of their use in two static register sets; the scheduling barriers keep hipcc from hoisting
            // all HS + 2 rows' reads (and the operand pairs built from them) to the top of the step, which spills
            constexpr int NQ = S + 3, NR = (HS - 1) * S + 3;               // dwords per lane and input row; input rows per lane
            float qv[2][NQ];
#pragma unroll
            for (int e = 0; e < NQ; ++e) qv[0][e] = tp[e];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                if (r + 1 < NR) {
#pragma unroll
                    for (int e = 0; e < NQ; ++e) qv[(r + 1) & 1][e] = tp[(r + 1) * PIT + e];
                }
                __builtin_amdgcn_sched_barrier(0);
                const float* q = qv[r & 1];
                // tap kw of the output pair (c, c+1) reads input columns (S c + kw - 1, S (c+1) + kw - 1)
                const p2 v0 = {q[0], q[S]}, v1 = {q[1], q[1 + S]}, v2 = {q[2], q[2 + S]};
#pragma unroll
                for (int i = 0; i < HS; ++i) {
                    const int kh = r - i * S;
                    if (kh >= 0 && kh < 3) {
#pragma unroll
                        for (int kt = 0; kt < 3; ++kt) {
                            const int sl = (j + 1 - kt + 3) % 3;          // accumulator of output frame f + 1 - kt
                            const float w0 = wr[kt * 9 + kh * 3 + 0], w1 = wr[kt * 9 + kh * 3 + 1], w2 = wr[kt * 9 + kh * 3 + 2];
                            acc[sl][i] = __builtin_elementwise_fma((p2){w0, w0}, v0, __builtin_elementwise_fma((p2){w1, w1}, v1,
                                         __builtin_elementwise_fma((p2){w2, w2}, v2, acc[sl][i])));
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // output frame f-1 is complete in slot (j - 1) % 3
            const int se = (j + 2) % 3;
            const int to = f - 1;
            const bool emit = to >= t0 && to < t1;                         // wave uniform
            const int so = cfn_uni(emit ? to * PO * CP_ES : 0);
            const float mf = emit ? lane_m : 0.0f;
            const int vo = emit ? yo : OOB;                                // + i * W * 4 below: the instruction's immediate offset
#pragma unroll
            for (int i = 0; i < HS; ++i) {
                const p2 y = cp_rt2(acc[se][i]);                            // (bf16: statistics over the stored values)
                cp_st2(y, ry, vo + i * W * CP_ES, so);
                const p2 ym = y * mf;
                s1p += ym;
                s2p = __builtin_elementwise_fma(ym, y, s2p);
                acc[se][i] = (p2){0.0f, 0.0f};
            }
            // hipcc otherwise sinks the statistics of all U steps to the end of the trip and spills the rows they read
            asm volatile("" : "+v"(s1p), "+v"(s2p));
        }
    }
    if (a.s1) {
        const float st1 = cfn_wave_sum(s1p.x + s1p.y), st2 = cfn_wave_sum(s2p.x + s2p.y);
        if (lane == 0) { cfn_add64(&a.s1[nc], (double)st1); cfn_add64(&a.s2[nc], (double)st2); }
    }
}

// returns -1 when the shape is not handled (caller goes on to the other kernels); probe: 0 = handled, nothing launched;
// otherwise the launch status
int CPN(dw_cp_fwd_try)(const cpe_t* x, const double* A, const double* B, int act, const float* w, cpe_t* y, double* sum, double* sumsq,
                  int N, int C, int T, int Hi, int Wi, int stride, hipStream_t st, bool probe) {
    // bit mask of the shapes served: stride 1: 1 = 56x56, 2 = 28x28, 4 = 14x14; stride 2: 8 = 112->56, 16 = 56->28, 32 = 28->14
    static const int enabled = getenv("CFN_DW_CP") ? atoi(getenv("CFN_DW_CP")) : 63;
    static const int tt_env = getenv("CFN_DW_CP_TT") ? atoi(getenv("CFN_DW_CP_TT")) : 0;
    if (Hi != Wi || (stride != 1 && stride != 2)) return -1;
    const int Ho = stride == 1 ? Hi : Hi / 2;
    if ((stride == 2 && (Hi & 1)) || (Ho != 56 && Ho != 28 && Ho != 14)) return -1;
    const int bit = (Ho == 56 ? 1 : Ho == 28 ? 2 : 4) << (stride == 2 ? 3 : 0);
    if (!(enabled & bit)) return -1;
    if (act != CFN_ACT_NONE && act != CFN_ACT_RELU && A != nullptr) return -1;      // branch-free prologue: none / ReLU (every X3D conv2)
    if ((long)T * Hi * Wi * CP_ES >= 0x7fff0000L) return -1;
    if ((((uintptr_t)x | (uintptr_t)y) & (4 * CP_ES - 1)) != 0) return -1;
    if (probe) return 0;
    DwCpArgs a = {x, A, B, w, y, sum, sumsq, N, C, T, act, 0, 0, 0};
    // lanes = column pairs x row groups (few rows per lane = few accumulators = many resident waves, which is what these
    // kernels need: measured on 56x56, 8 clips x T=256: 7 rows per lane, 168 VGPRs, 3 waves per SIMD: 3.5 TB/s; 2 rows, 93
    // VGPRs, 5 waves: 5.1 TB/s; capped at 80 / 64 VGPRs the spills cost 1.6x / 3x)
    //   stride 1: 56x56: 28 x 2 lanes x 2 rows (14 bands of 4 rows); 28x28: 14 x 4 x 1 (7 bands); 14x14: 7 x 7 x 2 (the plane)
    //   stride 2: ->56: 28 x 2 x 2 (14 bands); ->28: 14 x 4 x 1 (7 bands); ->14: 7 x 7 x 1 (2 bands)
    // CFN_DW_CP_TALL=1: stride-1 56x56 / 28x28 with 8-row bands (10 input rows per 8 output rows instead of 6 per 4: the halo rows
    // are re-read by the neighbouring band, and re-reads cost the vector-memory path as much as first reads)
    static const int tall = getenv("CFN_DW_CP_TALL") ? atoi(getenv("CFN_DW_CP_TALL")) : 0;
    const bool tall56 = (tall & 1) && stride == 1 && Ho == 56, tall28 = (tall & 2) && stride == 1 && Ho == 28;
    const int NB = tall56 ? 7 : tall28 ? 4 : Ho == 56 ? 14 : Ho == 28 ? 7 : (stride == 2 ? 2 : 1);
    // t-chunks: >= ~6 rounds of the chip's resident waves (16 per CU) so that the tail of the last round stays small; the
    // chunk length + 2 halo frames is a multiple of the unrolled trip (6 or 12 steps) where T allows it
    const long units = (long)N * C * NB;
    const int U = (Ho == 14 && stride == 1) ? 12 : 6;
    long nch = (6L * 256 * 16 + units - 1) / units;
    if (nch < 1) nch = 1;
    int TT = (int)((T + nch - 1) / nch);
    TT = ((TT + 2 + U - 1) / U) * U - 2;                                            // TT + 2 = k * U
    if (TT < 16) TT = 16;                                                           // (measured at T = 16, 8 clips: chunks of 4 frames cost 15-40 % against one chunk of 16)
    // stride 2 (4:1 read:write, parked on memory rather than issue bound): shorter chunks = more, shorter-lived waves; same-box sweep
    // at 8 clips x T = 256 (16 / 22 / 28 / 34 / 52 frames): 112->56 1.409 / 1.449 / 1.457 / 1.452 / 1.434 ms, 56->28 0.689 / 0.705 /
    // 0.730 / 0.731 / 0.723, 28->14 0.357 / 0.365 / 0.380 / 0.374 / 0.377; the stride-1 planes are within +-2 % from 16 up
    if (stride == 2 && TT > 16) TT = 16;
    if (tt_env > 0) TT = tt_env;
    if (TT > T) TT = T;
    a.TT = TT;
    a.nchunks = (T + TT - 1) / TT;
    a.total_waves = units * a.nchunks;
    const unsigned blocks = (unsigned)((a.total_waves + 3) / 4);
#define CFN_CP_GO(...) hipLaunchKernelGGL((dw3d_cp_fwd_kernel<__VA_ARGS__>), dim3(blocks), dim3(256), 0, st, a)
    if (stride == 1) {
        if (tall56) CFN_CP_GO(56, 1, 4, 2, 2, 3); else if (tall28) CFN_CP_GO(28, 1, 2, 4, 2, 4);
        else if (Ho == 56) CFN_CP_GO(56, 1, 2, 2, 2, 4); else if (Ho == 28) CFN_CP_GO(28, 1, 1, 4, 2, 4); else CFN_CP_GO(14, 1, 2, 7, 4, 4);
    } else {
        if (Ho == 56) CFN_CP_GO(56, 2, 2, 2, 2, 4); else if (Ho == 28) CFN_CP_GO(28, 2, 1, 4, 2, 4); else CFN_CP_GO(14, 2, 1, 7, 2, 4);
    }
#undef CFN_CP_GO
    return cfn_check_launch("dwconv3d column-pair forward");
}
