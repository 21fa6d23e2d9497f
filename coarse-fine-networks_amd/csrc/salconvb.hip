// Grid Pool saliency convolutions, FORWARD, on the split-bf16 matrix pipe (round 6; x3d_coarse.py:362-366, :379-381: Conv3d(24, 24, (3,3,3),
// stride (2,2,2), padding 1) on 56 x 56 and 28 x 28 planes).
//
// salconv.hip's forward is an exact-fp32 MFMA kernel (v_mfma_f32_32x32x2_f32: 81 MFMAs x 64 cycles per wave, tile and output frame); it ran at
// 1.5 TB/s of its algorithmic bytes inside figure B (BENCH_r05 roofline_coarse: dense_fwd 1.58 ms for 2.37 GB).  Here the same contraction runs
// as in csrc/pws_kernel.h: fp32 tensors, every operand split on the fly into three bf16 terms, six v_mfma_f32_32x32x16_bf16 per k-block
// (a1 b1 + a1 b2 + a2 b1 + a2 b2 + a1 b3 + a3 b1, dropped terms <= 3 * 2^-27 |a b|: below fp32 rounding), fp32 accumulation --
// 24 MFMAs x 32 cycles per wave, tile and temporal tap.
//
//   * k order = (kh, kw, ci): 27 GROUPS of 8 input channels (tap = kh * 3 + kw, channel third), two groups per 16-deep k-block.  The LDS
//     image of a staged frame is CHANNELS LAST, img[term][row][col][24 ci] bf16 (48 B per pixel): a lane's B operand -- 8 consecutive k of one
//     output position -- is ONE ds_read_b128 per term, whatever the tap;
//   * the contraction is split over the 4 waves of a workgroup by group (7 + 7 + 7 + 6, padded to 4 k-blocks each); a wave keeps the
//     3 temporal taps x 4 k-blocks x 3 terms of its weight slice in 144 registers (MFMA A operand: lane = output channel);
//   * every input frame is staged ONCE by all 256 threads (a thread = 4 channels x 4 columns of one row: four 16-byte loads one frame ahead,
//     prologue relu(A x + B), split, three 8-byte LDS writes per pixel) into one of TWO image buffers -- one workgroup barrier per frame --
//     and is consumed on the spot by the temporal taps it feeds (even frame 2 to: tap 1 of output frame to; odd frame 2 to + 1: tap 2 of frame
//     to and tap 0 of frame to + 1), as in salconv.hip;
//   * once per output frame the four K-slice partial tiles meet in LDS in a fixed order (bit-repeatable), wave j finishes register group j:
//     statistics in-lane, stores of whole output rows -- salconv.hip's epilogue.
//
// Measured (8 clips x 256 frames, tools/salb_bench.py, profiles/r06_salb_bench.txt): conv1 56 -> 28 0.443-0.452 ms against 0.468 for the exact-fp32
// kernel, conv2 28 -> 14 0.059-0.071 against 0.074; max |dy| / max |y| 5e-7 between the two, both 3e-7 from fp64, bit-repeatable.  Far from the
// 2.7 x the MFMA counts promise, and the knock-out builds say why (tools/salb_knockouts.sh, profiles/r06_salb_knockouts.txt; conv1): frame loop +
// barriers alone 0.047 ms, + loads + staging 0.187, the MFMA phase ANOTHER 0.256 -- the parts add instead of overlapping.  A wave's 24-48 MFMAs
// per frame are ONE or two dependent accumulator chains (64 cycles issue to issue instead of 32: the even frames feed a single temporal tap) behind
// three ds_read_b128 with a full lgkmcnt wait per k-block, 256 VGPRs + 10-21 spilled leave no room for a second chain, and the barriers keep the
// two workgroups of a CU in step, so the other workgroup's staging does not hide it.  What would: 8 waves per workgroup (half the weight registers
// per wave: room for split accumulators and operand prefetch) with the next frame's staging pinned behind this frame's MFMAs (DESIGN 4.6's scheme)
// -- ~0.3 ms by the same arithmetic; not built: the whole family is 0.5 ms of a 56 ms coarse step.
#include "pws_kernel.h"

#ifndef SALB_KO
#define SALB_KO 0          // knock-out bit mask of the timing experiments (tools/variant_lib.sh): 1 MFMAs, 2 staging arithmetic + LDS writes, 4 global loads, 8 reduction + stores
#endif

typedef float __attribute__((ext_vector_type(16))) sb16;
typedef float __attribute__((ext_vector_type(4))) sb4;

struct SalBArgs {
    const float* x; const double* pa; const double* pb; const float* w; float* y; double* s1; double* s2;
    int N, Cout, T, To, Hi, Ho, bands, nchunks, TO;
};

#define SALB_CIN 24
#define SALB_PIX 48          // bytes per pixel and term: 24 channels x bf16
#define SALB_KB 4            // k-blocks (pairs of groups) per wave
#define SALB_GPW 7           // groups per wave (27 = 7 + 7 + 7 + 6)

template <int WI, bool PRO>
__global__ __launch_bounds__(256, 2) void sal_fwdb_kernel(const SalBArgs a) {
    constexpr int WO = WI / 2, TR = 32 / WO, RIN = 2 * TR + 1, PITCHC = WI + 1, W4 = WI / 4;      // image column = input column + 1 (zero column for iw = -1)
    constexpr int IMGT = RIN * PITCHC * SALB_PIX, IMGB = 3 * IMGT;                                 // bytes of one term image / of one frame buffer
    constexpr int UNITS = 6 * RIN * W4, OOB = 0x7fff0000;                                          // staging units: (channel quad, row, column quad)
    static_assert(UNITS <= 256 && TR >= 1 && TR * WO <= 32 && WI % 4 == 0 && IMGT % 16 == 0, "geometry");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];                           // 2 frame buffers | red[wave][4][64] float4
    const int tid = threadIdx.x, lane = tid & 63, wv = cfn_uni(tid >> 6), kg = lane >> 5, p = lane & 31;
    unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    const int band = cfn_uni((int)(L % a.bands)); L /= a.bands;
    const int chunk = cfn_uni((int)(L % a.nchunks));
    const int n = cfn_uni((int)(L / a.nchunks));
    const int T = a.T, Hi = a.Hi, Ho = a.Ho, To = a.To, Cout = a.Cout;
    const int to0 = chunk * a.TO, nto = min(a.TO, To - to0);
    const int oh0 = band * TR, ih0 = 2 * oh0 - 1;
    sb4* red = reinterpret_cast<sb4*>(smem + 2 * IMGB);

    for (int i = tid; i < 2 * IMGB / 16; i += 256) reinterpret_cast<u4v*>(smem)[i] = (u4v){0u, 0u, 0u, 0u};     // zero column, rows outside the plane

    // ---- staging role: channels 4 cq .. 4 cq + 3, image row sr, input columns 4 sq .. 4 sq + 3 ------------------------------------------------
    const int cq = tid % 6, srq = tid / 6, sr = srq / W4, sq = srq - sr * W4;
    const int sih = ih0 + sr;
    const bool sok = tid < UNITS && sih >= 0 && sih < Hi;
    const long P = (long)Hi * WI;
    __amdgpu_buffer_rsrc_t rx = cfn_rsrc(a.x + (long)n * SALB_CIN * T * P, (unsigned)((long)SALB_CIN * T * P * 4));
    int ldo[4];
    float ua[PRO ? 4 : 1], ub[PRO ? 4 : 1];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        ldo[c] = sok ? (int)((((long)(4 * cq + c) * T) * Hi + sih) * WI + sq * 4) * 4 : OOB;      // (< 2^31: the launcher checks 24 T P 4 bytes)
        if (PRO) {
            const int cg = n * SALB_CIN + 4 * cq + c;
            ua[c] = (float)a.pa[cg]; ub[c] = (float)a.pb[cg];
        }
    }
    const int sto = (sr * PITCHC + sq * 4 + 1) * SALB_PIX + cq * 8;                                 // byte offset of (row, column 4 sq, channel quad) in a term image
    sb4 fr[4];
    auto fetch = [&](int f) {
        if (SALB_KO & 4) return;
        const int so = cfn_uni((int)(f * P * 4));
#pragma unroll
        for (int c = 0; c < 4; ++c) fr[c] = __builtin_bit_cast(sb4, __builtin_amdgcn_raw_buffer_load_b128(rx, ldo[c], so, 0));
    };
    auto stage = [&](unsigned char* buf) {
        if (sok && !(SALB_KO & 2)) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {                                                           // column 4 sq + e: four channels -> 3 x 8 bytes
                float v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    v[c] = fr[c][e];
                    if (PRO) v[c] = fmaxf(fmaf(v[c], ua[c], ub[c]), 0.0f);
                }
                unsigned p01[3], p23[3];
                pws_split<3>(v[0], v[1], p01);
                pws_split<3>(v[2], v[3], p23);
#pragma unroll
                for (int s = 0; s < 3; ++s) *reinterpret_cast<u2v*>(buf + s * IMGT + sto + e * SALB_PIX) = (u2v){p01[s], p23[s]};
            }
        }
    };

    // ---- weights: MFMA A operand, lane (co = p, kg): k-block kb, slot s = 2 kb + kg -> group g = 7 wv + s = (tap, channel third) -------------------
    bf16x8 wr[3][SALB_KB][3];
    int lb[SALB_KB];                                                                                // B operand byte offsets of this lane (tile row 0 of the band)
    const bool pv = p < TR * WO;
    const int pr = pv ? p / WO : 0, pc = pv ? p - pr * WO : 0;
#pragma unroll
    for (int kb = 0; kb < SALB_KB; ++kb) {
        const int s = 2 * kb + kg, g = wv * SALB_GPW + s;
        const bool live = s < SALB_GPW && g < 27;
        const int tap = live ? g / 3 : 0, c8 = live ? g - tap * 3 : 0, kh = tap / 3, kw = tap - kh * 3;
        lb[kb] = ((2 * pr + kh) * PITCHC + 2 * pc + kw) * SALB_PIX + c8 * 16;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                v[i] = (live && p < Cout) ? a.w[(long)p * (SALB_CIN * 27) + (c8 * 8 + i) * 27 + kt * 9 + tap] : 0.0f;
            u4v t3[3];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                unsigned sp[3];
                pws_split<3>(v[2 * h], v[2 * h + 1], sp);
#pragma unroll
                for (int s3 = 0; s3 < 3; ++s3) t3[s3][h] = sp[s3];
            }
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) wr[kt][kb][s3] = __builtin_bit_cast(bf16x8, t3[s3]);
        }
    }

    sb16 accA, accB;
    float ssum[4], ssq[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { ssum[r] = 0.0f; ssq[r] = 0.0f; }
    const sb16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // One staged frame against the temporal taps KA (accumulated into C) and KB (into Nx); -1: none.  A tap-0 product OPENS its accumulator.
#define SALB_MMA(KA, KB, C, Nx, BUF)                                                                                                    \
    do {                                                                                                                                  \
        if (SALB_KO & 1) break;                                                                                                           \
        _Pragma("unroll") for (int kb = 0; kb < SALB_KB; ++kb) {                                                                         \
            bf16x8 Bt[3];                                                                                                                 \
            _Pragma("unroll") for (int s3 = 0; s3 < 3; ++s3) Bt[s3] = *reinterpret_cast<const bf16x8*>((BUF) + s3 * IMGT + lb[kb]);      \
            if (KA >= 0) {                                                                                                                \
                bool first = (KA == 0 && kb == 0);                                                                                        \
                pws_terms<3>([&](int sa, int sb) {                                                                                        \
                    C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[KA < 0 ? 0 : KA][kb][sa], Bt[sb], first ? zero16 : C, 0, 0, 0);       \
                    first = false;                                                                                                        \
                });                                                                                                                       \
            }                                                                                                                             \
            if (KB >= 0) {                                                                                                                \
                bool first = (KB == 0 && kb == 0);                                                                                        \
                pws_terms<3>([&](int sa, int sb) {                                                                                        \
                    Nx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[KB < 0 ? 0 : KB][kb][sa], Bt[sb], first ? zero16 : Nx, 0, 0, 0);     \
                    first = false;                                                                                                        \
                });                                                                                                                       \
            }                                                                                                                             \
        }                                                                                                                                 \
    } while (0)

    const int oh = oh0 + pr;                                                                        // output row of this lane's position
    const bool ov = pv && oh < Ho;
    const long PO = (long)Ho * WO;
    int cur = 0;                                                                                    // frame buffer the NEXT stage() writes
    auto nextbuf = [&]() { unsigned char* b = smem + cur * IMGB; cur ^= 1; return b; };

    // output frame to: even frame 2 to (tap 1), odd frame 2 to + 1 (tap 2; tap 0 of frame to + 1 into Nx), reduction, epilogue.
    // Buffers alternate: a buffer is rewritten two stages later, behind the barrier of the stage in between -- every wave has left its MFMAs by then.
    auto step = [&](sb16& C, sb16& Nx, int i) __attribute__((always_inline)) {
        const int to = to0 + i, fo = 2 * to + 1;
        const bool more = i + 1 < nto;
        {
            unsigned char* b = nextbuf();
            stage(b);                                                                               // even frame (always inside the clip)
            if (fo < T) fetch(fo);
            __syncthreads();
            SALB_MMA(1, -1, C, Nx, b);
        }
        if (fo < T) {
            unsigned char* b = nextbuf();
            stage(b);
            if (more) fetch(fo + 1);
            __syncthreads();
            if (more) SALB_MMA(2, 0, C, Nx, b); else SALB_MMA(2, -1, C, Nx, b);
        }                                                                                           // fo >= T: to is the clip's last output frame
        // ---- the four K-slice partials of the tile meet in LDS (fixed order: bit-repeatable); wave j finishes register group j ----------------
        if (SALB_KO & 8) return;
#pragma unroll
        for (int q = 0; q < 4; ++q) red[(wv * 4 + q) * 64 + lane] = (sb4){C[4 * q], C[4 * q + 1], C[4 * q + 2], C[4 * q + 3]};
        __syncthreads();
        const sb4* src = red + wv * 64 + lane;                                                      // group wv of wave 0, 1, 2, 3
        const sb4 tot = ((src[0] + src[4 * 64]) + src[2 * 4 * 64]) + src[3 * 4 * 64];
        __syncthreads();
        float* yp = a.y + (((long)n * Cout) * To + to) * PO + (long)oh0 * WO + p;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int co = e + 8 * wv + 4 * kg;
            const float v = tot[e];
            if (ov && co < Cout) {
                yp[(long)co * To * PO] = v;
                ssum[e] += v; ssq[e] = fmaf(v, v, ssq[e]);
            }
        }
    };

    {   // halo frame 2 to0 - 1: tap 0 of the chunk's first output frame
        const int f = 2 * to0 - 1;
        __syncthreads();                                                                            // the zero fill is complete
        if (f >= 0) {
            fetch(f);
            unsigned char* b = nextbuf();
            stage(b);
            fetch(2 * to0);
            __syncthreads();
            SALB_MMA(0, -1, accA, accB, b);
        } else {
            fetch(2 * to0);
            accA = zero16;
        }
    }
    int i = 0;
    for (; i + 1 < nto; i += 2) { step(accA, accB, i); step(accB, accA, i + 1); }
    if (i < nto) step(accA, accB, i);
#undef SALB_MMA
    if (a.s1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s = ssum[r], q = ssq[r];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
            const int co = r + 8 * wv + 4 * kg;
            if (p == 0 && co < Cout) {
                cfn_add64(&a.s1[(long)n * Cout + co], (double)s);
                cfn_add64(&a.s2[(long)n * Cout + co], (double)q);
            }
        }
    }
}

// -1 = shape not handled / switched off (the caller runs the exact-fp32 kernel of salconv.hip); same contract as sal_fwd_try_launch
int salb_fwd_try_launch(const float* x, const double* A, const double* B, int act, const float* w, float* y, double* sum, double* sumsq, int N, int Cin, int Cout,
                        int T, int Hi, int Wi, const int* g, hipStream_t st) {
    static const int want[9] = {3, 3, 3, 2, 2, 2, 1, 1, 1};
    for (int i = 0; i < 9; ++i) if (g[i] != want[i]) return -1;
    if (Cin != SALB_CIN || Cout > 32 || (Wi != 56 && Wi != 28) || (Hi & 1) || Hi < 2 || ((uintptr_t)x & 15)) return -1;
    if (A && act != CFN_ACT_RELU) return -1;
    if (!A && act != CFN_ACT_NONE) return -1;
    static const int on = getenv("CFN_SAL_BF16") ? atoi(getenv("CFN_SAL_BF16")) : 1;
    if (!on || pws_terms_now() != 6) return -1;                       // the arithmetic setting of the pointwise contractions governs this one as well
    if ((long)SALB_CIN * T * Hi * Wi * 4 >= 0x7fff0000L) return -1;
    SalBArgs a = {x, A, B, w, y, sum, sumsq, N, Cout, T, (T - 1) / 2 + 1, Hi, Hi / 2};
    const int WO = Wi / 2, TR = 32 / WO, RIN = 2 * TR + 1, PITCHC = Wi + 1;
    a.bands = cfn_cdiv(a.Ho, TR);
    // chunk length: whole rounds of two workgroups per CU where possible (a 1.25-round grid costs a full second round)
    int cus = 256;
    { int dev = 0; hipDeviceProp_t pr; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) cus = pr.multiProcessorCount; }
    int best = 4; double bestc = 1e30;
    for (int to = 2; to <= 16; ++to) {
        const long blocks = (long)N * a.bands * cfn_cdiv(a.To, to);
        const double rounds = (double)cfn_cdiv(blocks, (long)cus * 2);
        const double cost = rounds * (3.0 * to + 1.0);               // MFMA sets per block: 3 per output frame + the halo frame's one
        if (cost < bestc - 1e-9) { bestc = cost; best = to; }
    }
    static const int to_env = getenv("CFN_SALB_TO") ? atoi(getenv("CFN_SALB_TO")) : 0;
    a.TO = to_env > 0 ? to_env : best;
    a.nchunks = cfn_cdiv(a.To, a.TO);
    const long blocks = (long)N * a.bands * a.nchunks;
    if (blocks >= (1L << 31)) return -1;
    const size_t lds = (size_t)2 * 3 * RIN * PITCHC * SALB_PIX + (size_t)4 * 4 * 64 * 16;
#define SALB_GO(WIV, PROV)                                                                                              \
    do {                                                                                                                \
        auto k = sal_fwdb_kernel<WIV, PROV>;                                                                            \
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                \
        hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(256), lds, st, a);                                           \
    } while (0)
    if (Wi == 56) { if (A) SALB_GO(56, true); else SALB_GO(56, false); }
    else { if (A) SALB_GO(28, true); else SALB_GO(28, false); }
#undef SALB_GO
    return cfn_check_launch("sal_conv_fwd (split bf16)");
}
