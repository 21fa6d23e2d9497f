// Weight gradient of the pointwise contractions on FP32 tensors with SPLIT-BF16 arithmetic (see pwsplit.hip for the
// split; x3d_fine.py:100-105 conv1 / conv3 of layers 2-4):
//     gW[m][k] += sum_q G'[m][q] * a[k][q],   G' = gsc*gy + gs + 2 y gq,   a = act(A x + B).
// The contraction runs over POSITIONS, contiguous in both operands, and their order inside an MFMA k-block is free as long
// as both operands agree: lane (r = l & 31, kg = l >> 5) streams along ONE channel row and takes, of every 16-position
// step, the two aligned float4s at p0 + 4 kg and p0 + 8 + 4 kg (8 of the step's 16 positions = its 8 consecutive k of
// v_mfma_f32_32x32x16_bf16), straight from HBM into VGPRs (no LDS transpose, no barrier in the main loop), applies the
// prologue with per-lane coefficients, splits into NS bf16 terms and feeds 3 (NS = 2) or 6 (NS = 3) MFMAs per tile pair.
// A wave owns TM x TN tiles of 32x32 outputs over its own steps; the 8 waves of a workgroup take the steps of a strip
// interleaved (the pieces of every 128-byte line are consumed by neighbouring waves at about the same time) and are
// combined through LDS into one fp64 atomic per element per workgroup.
#include "pw_common.h"
#include <stdlib.h>
#include <map>
#include <mutex>
#include <utility>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

#define PWSW_WAVES 8
#define PWSW_OOB 0x7ffffff0

__device__ __forceinline__ float pwsw_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float pwsw_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ unsigned pwsw_pack(float lo, float hi) {
    const bf16x2 b = __builtin_convertvector((f2v){lo, hi}, bf16x2);
    return __builtin_bit_cast(unsigned, b);
}
template <int NS>
__device__ __forceinline__ void pwsw_split(float v0, float v1, unsigned (&p)[NS]) {
    p[0] = pwsw_pack(v0, v1);
#pragma unroll
    for (int s = 1; s < NS; ++s) {
        v0 -= pwsw_lo(p[s - 1]);
        v1 -= pwsw_hi(p[s - 1]);
        p[s] = pwsw_pack(v0, v1);
    }
}
template <int NS, class F>
__device__ __forceinline__ void pwsw_terms(F&& f) {
    if (NS == 3) { f(2, 0); f(0, 2); f(1, 1); }
    if (NS >= 2) { f(1, 0); f(0, 1); }
    f(0, 0);
}

struct WsArgs {
    const float* gy; const float* y; const double* gs; const double* gq; const double* gsc;
    const float* x; const double* pa; const double* pb;
    double* gw;
    int N, M, K, Q, act;
    int mgroups, kgroups, nstrips, mt32, kt32;
};

// balanced split of `tiles` into `groups` runs: run g covers [first, first + count)
__device__ __forceinline__ void pwsw_run(int tiles, int groups, int g, int& first, int& count) {
    const int base = tiles / groups, rem = tiles - base * groups;
    first = g * base + min(g, rem);
    count = base + (g < rem ? 1 : 0);
}

template <int TM, int TN, int ACT, bool HASY, int NS>
__global__ __launch_bounds__(64 * PWSW_WAVES) void pws_wgrad_kernel(const WsArgs a) {
    __shared__ float cw[PWSW_WAVES][32 * 33];
    const int tid = threadIdx.x, wave = cfn_uni(tid >> 6), lane = tid & 63, kg = lane >> 5, row = lane & 31;
    unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    // tile groups vary fastest: workgroups that stream the same positions (and re-read the same operand rows) are
    // neighbours on one XCD and find each other's lines in its L2
    const int kgi = L % a.kgroups; L /= a.kgroups;
    const int mgi = L % a.mgroups; L /= a.mgroups;
    const int strip = L % a.nstrips;
    const int n = L / a.nstrips;
    const int M = a.M, K = a.K, Q = a.Q;
    int mt0, mtn, kt0, ktn;
    pwsw_run(a.mt32, a.mgroups, mgi, mt0, mtn);   // <= TM row tiles
    pwsw_run(a.kt32, a.kgroups, kgi, kt0, ktn);   // <= TN column tiles
    const int m0 = mt0 * 32, k0 = kt0 * 32;

    // per-lane prologue coefficients (lane <-> channel row of each tile)
    float cs[TM], cq[TM], cz[TM], ca[TN], cb[TN];
    int om[TM], ok_[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + i * 32 + row;
        const bool ok = i < mtn && m < M;
        cs[i] = (ok && a.gs) ? (float)a.gs[(long)n * M + m] : 0.0f;
        cq[i] = (ok && HASY && a.gq) ? 2.0f * (float)a.gq[(long)n * M + m] : 0.0f;
        cz[i] = (ok && a.gsc) ? (float)a.gsc[(long)n * M + m] : 1.0f;
        om[i] = ok ? m * Q * 4 : PWSW_OOB;
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int k = k0 + i * 32 + row;
        const bool ok = i < ktn && k < K;
        ca[i] = (ok && a.pa) ? (float)a.pa[(long)n * K + k] : 1.0f;
        cb[i] = (ok && a.pb) ? (float)a.pb[(long)n * K + k] : 0.0f;
        ok_[i] = ok ? k * Q * 4 : PWSW_OOB;
    }
    const int row_bytes = Q * 4;
    __amdgpu_buffer_rsrc_t rg = cfn_rsrc(const_cast<float*>(a.gy + (long)n * M * Q), (unsigned)((long)M * row_bytes));
    __amdgpu_buffer_rsrc_t ry = cfn_rsrc(const_cast<float*>((HASY ? a.y : a.gy) + (long)n * M * Q), (unsigned)((long)M * row_bytes));
    __amdgpu_buffer_rsrc_t rx = cfn_rsrc(const_cast<float*>(a.x + (long)n * K * Q), (unsigned)((long)K * row_bytes));

    // steps of 16 positions; the workgroup owns a contiguous run of steps, its waves take them interleaved
    const int nst = (Q + 15) >> 4;
    const int per = (nst + a.nstrips - 1) / a.nstrips;
    const int sbeg = strip * per + wave, send = min(strip * per + per, nst);
    f16v acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) acc[i][jn] = (f16v)0.0f;

    auto ld4 = [&](__amdgpu_buffer_rsrc_t r, int voff) -> f4v {
        return __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
    };
    f4v rG[TM][2], rY[TM][2], rX[TN][2];
    // unconditional loads: a float4 beyond the row end (Q % 4 == 0: all inside or all outside) or of a dead row goes to an
    // out-of-range offset and reads 0
    auto load = [&](int s) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int q = s * 16 + 8 * e + 4 * kg;
            const bool inq = q < Q;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int vm = inq && om[i] != PWSW_OOB ? om[i] + q * 4 : PWSW_OOB;
                rG[i][e] = ld4(rg, vm);
                if (HASY) rY[i][e] = ld4(ry, vm);
            }
#pragma unroll
            for (int i = 0; i < TN; ++i) rX[i][e] = ld4(rx, inq && ok_[i] != PWSW_OOB ? ok_[i] + q * 4 : PWSW_OOB);
        }
    };
    if (sbeg < send) load(sbeg);
    for (int s = sbeg; s < send; s += PWSW_WAVES) {
        u4v Aop[TM][NS], Bop[TN][NS];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float vm = (s * 16 + 8 * e + 4 * kg < Q) ? 1.0f : 0.0f;     // masks the constant term beyond the row end
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float c0 = cs[i] * vm;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float g0 = fmaf(rG[i][e][2 * h], cz[i], c0), g1 = fmaf(rG[i][e][2 * h + 1], cz[i], c0);
                    if (HASY) { g0 = fmaf(rY[i][e][2 * h], cq[i], g0); g1 = fmaf(rY[i][e][2 * h + 1], cq[i], g1); }
                    unsigned p[NS];
                    pwsw_split<NS>(g0, g1, p);
#pragma unroll
                    for (int sp = 0; sp < NS; ++sp) Aop[i][sp][2 * e + h] = p[sp];
                }
            }
#pragma unroll
            for (int i = 0; i < TN; ++i) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float x0 = cfn_act<ACT>(fmaf(rX[i][e][2 * h], ca[i], cb[i])), x1 = cfn_act<ACT>(fmaf(rX[i][e][2 * h + 1], ca[i], cb[i]));
                    unsigned p[NS];
                    pwsw_split<NS>(x0, x1, p);
#pragma unroll
                    for (int sp = 0; sp < NS; ++sp) Bop[i][sp][2 * e + h] = p[sp];
                }
            }
        }
        if (s + PWSW_WAVES < send) load(s + PWSW_WAVES);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
                if (i < mtn && jn < ktn)
                    pwsw_terms<NS>([&](int sa, int sb) {
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Aop[i][sa]), __builtin_bit_cast(bf16x8, Bop[jn][sb]), acc[i][jn], 0, 0, 0);
                    });
    }

    // ---- combine the 8 waves tile by tile through LDS, one fp64 atomic per element per workgroup ----------
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            if (i < mtn && jn < ktn) {                           // workgroup uniform
#pragma unroll
                for (int r = 0; r < 16; ++r) cw[wave][((r & 3) + 8 * (r >> 2) + 4 * kg) * 33 + row] = acc[i][jn][r];
                __syncthreads();
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int el = tid + e * 64 * PWSW_WAVES;    // 1024 elements of the tile
                    const int ml = el >> 5, kl = el & 31;
                    float v = 0.0f;
#pragma unroll
                    for (int w = 0; w < PWSW_WAVES; ++w) v += cw[w][ml * 33 + kl];
                    const int m = m0 + i * 32 + ml, k = k0 + jn * 32 + kl;
                    if (m < M && k < K) cfn_add64(&a.gw[(long)m * K + k], (double)v);
                }
                __syncthreads();
            }
        }
}

template <int TM, int TN, int NS>
static int pwsw_launch(WsArgs& a, hipStream_t st) {
    a.mt32 = cfn_cdiv(a.M, 32); a.kt32 = cfn_cdiv(a.K, 32);
    a.mgroups = cfn_cdiv(a.mt32, TM); a.kgroups = cfn_cdiv(a.kt32, TN);
    const long groups = (long)a.N * a.mgroups * a.kgroups;
    static const int wg_env = getenv("CFN_PWSW_WGS") ? atoi(getenv("CFN_PWSW_WGS")) : 0;
    long strips = (wg_env > 0 ? wg_env : 512) / groups;
    if (strips < 1) strips = 1;
    const long nst = cfn_cdiv(a.Q, 16);
    if (strips > cfn_cdiv(nst, PWSW_WAVES * 4)) strips = cfn_cdiv(nst, PWSW_WAVES * 4);   // >= 4 steps per wave
    a.nstrips = (int)strips;
    const unsigned blocks = (unsigned)(groups * strips);
#define PWSW_GO(AV)                                                                                                        \
    do {                                                                                                                   \
        if (a.y) hipLaunchKernelGGL((pws_wgrad_kernel<TM, TN, AV, true, NS>), dim3(blocks), dim3(64 * PWSW_WAVES), 0, st, a); \
        else hipLaunchKernelGGL((pws_wgrad_kernel<TM, TN, AV, false, NS>), dim3(blocks), dim3(64 * PWSW_WAVES), 0, st, a);    \
    } while (0)
    switch (a.act) {
        case CFN_ACT_RELU: PWSW_GO(CFN_ACT_RELU); break;
        case CFN_ACT_SWISH: PWSW_GO(CFN_ACT_SWISH); break;
        default: PWSW_GO(CFN_ACT_NONE); break;
    }
#undef PWSW_GO
    return cfn_check_launch("pwconv_bwd_weight(split bf16)");
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS-staged variant (the one in use).  The direct kernel above reads 32-byte pieces of 32 different rows per load
// instruction; measured with the matrix pipe out of the way it moves 1.4-3.2 TB/s -- exactly what the fp32-MFMA kernel of the
// same access pattern reached, i.e. that pattern, not the MFMA rate, was the limit.  Here a workgroup (8 waves) owns a whole
// (row group x column group) block of gW and a strip of positions, and streams the strip in half-stages of 32 positions:
//   * every thread loads whole-line pieces (8 lanes x 16 B = one 128-byte line of a row; 8 rows per wave instruction) of gy, y
//     and x into one of TWO register sets, unconditionally (exact vmcnt waits; the other set stays in flight);
//   * G' = gsc gy + gs + 2 y gq and a = act(A x + B) are formed ONCE per element, split into NS bf16 terms and written to one
//     of TWO LDS buffers as [row][32 positions] bf16 images (80-byte pitch: 5 slots, odd => conflict-free ds_read_b128);
//   * one barrier per half-stage (two buffers: the writes of half-stage h+2 are separated from the reads of h by the barrier of
//     h+1); then the loads of half-stage h+2 are issued and every wave runs the MFMAs of ITS tiles (up to TPW 32x32 tiles, dealt
//     in row-major runs), reading both operands of a k-block as 16 contiguous bytes per lane;
//   * no cross-wave reduction: a wave owns its tiles for the whole strip and adds them to gW with one fp64 atomic per element.
// ---------------------------------------------------------------------------------------------------------------------
#define PWSS_P 32
#define PWSS_PITCH 80
#define PWSS_THREADS 512

struct WssArgs {      // gy, y, x: fp32 (ES = 4) or bf16 (ES = 2) tensors
    const void* gy; const void* y; const double* gs; const double* gq; const double* gsc;
    const void* x; const double* pa; const double* pb;
    double* gw;
    int N, M, K, Q, act;
    int mgroups, kgroups, nstrips, mt32, kt32;
    int mtg, ktg, per;      // row / column tiles per group (LDS images are 32*mtg / 32*ktg rows), positions per strip (multiple of 32)
    float* ws;              // null: one fp64 atomic per element and workgroup; else the workgroup's fp32 tiles go to ws (pws_wgrad_reduce_kernel adds them)
};

// ES = 2: bf16 tensors (the bf16 activation path, section 4b of DESIGN.md): the same staging with 8-byte loads of 4 positions and
// ONE bf16 term per operand (NS = 1: the operands are bf16 already; G' and the prologue are formed in fp32 and rounded once)
// BR x BC > 0 (round 5): a wave owns a BLOCK of BR x BC tiles instead of every 8th tile of the row-major order.  Dealt round robin, a wave's
// tiles share no operand: each of its 18 MFMAs per k-block (3 tiles x 6 terms) has its own pair of 16-byte LDS reads, 288 KB of LDS reads per
// half-stage and workgroup = 2,304 cycles of the LDS pipe next to 2,304 cycles of matrix pipe per SIMD -- and the multiply phase of a half-stage
// cannot hide behind anything (the barriers of section 4.1 keep every wave in the same phase).  A 1 x 3 (3 x 1) block reads its row (column)
// operand once per k-block and streams the other side: 12 reads per 18 MFMAs.
// PIPE (with blocks): the conversion of half-stage h + 1 runs in the SAME barrier interval as the MFMAs of half-stage h, cut into 12 SM pieces (a
// pair of elements through one stage of the split) that are pinned behind the MFMAs one by one -- a wave's MFMA chain is dependency paced (32+ cycles
// per instruction), and in the phase structure above that time hides nothing: convert -> barrier -> multiply costs the SUM of a VALU-bound and a
// matrix-bound phase (7,750 cycles per half-stage at layer 3 with 2,304 cycles of matrix pipe per SIMD in it).  One barrier per half-stage; the
// prologue coefficients sit in registers for the whole strip (no LDS read feeds the conversion: DESIGN 4.1).
template <int SM, int TPW, int ACT, bool HASY, int NS, int ES, int BR = 0, int BC = 0, bool PIPE = false>
__global__ __launch_bounds__(PWSS_THREADS) void pws_wgrad_staged_kernel(const WssArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = cfn_uni(tid >> 6), lane = tid & 63, kg = lane >> 5, r = lane & 31;
    unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    // tile groups vary fastest: workgroups that stream the same positions are neighbours on one XCD (shared rows hit its L2)
    const int kgi = L % a.kgroups; L /= a.kgroups;
    const int mgi = L % a.mgroups; L /= a.mgroups;
    const int strip = L % a.nstrips;
    const int n = L / a.nstrips;
    const int M = a.M, K = a.K, Q = a.Q;
    int mt0, mtn, kt0, ktn;
    pwsw_run(a.mt32, a.mgroups, mgi, mt0, mtn);
    pwsw_run(a.kt32, a.kgroups, kgi, kt0, ktn);
    const int m0 = mt0 * 32, k0 = kt0 * 32;
    const int GR = 32 * a.mtg, XR = 32 * a.ktg;                 // LDS image rows (padded: rows beyond the group / M / K hold zeros)
    const int gimg = GR * PWSS_PITCH, ximg = XR * PWSS_PITCH;
    const int bufb = NS * (gimg + ximg);
    unsigned char* buf0 = smem;                                 // [2][ G: NS x GR x 80 | X: NS x XR x 80 ]
    float4* cG = reinterpret_cast<float4*>(smem + 2 * bufb);    // [GR] (gs, 2 gq, gsc, -)
    float2* cX = reinterpret_cast<float2*>(cG + GR);            // [XR] (A, B)
    for (int i = tid; i < GR; i += PWSS_THREADS) {
        const int m = m0 + i;
        const bool ok = i < mtn * 32 && m < M;
        float4 c;
        c.x = (ok && a.gs) ? (float)a.gs[(long)n * M + m] : 0.0f;
        c.y = (ok && HASY && a.gq) ? 2.0f * (float)a.gq[(long)n * M + m] : 0.0f;
        c.z = (ok && a.gsc) ? (float)a.gsc[(long)n * M + m] : 1.0f;
        c.w = 0.0f;
        cG[i] = c;
    }
    for (int i = tid; i < XR; i += PWSS_THREADS) {
        const int k = k0 + i;
        const bool ok = i < ktn * 32 && k < K;
        cX[i] = float2{(ok && a.pa) ? (float)a.pa[(long)n * K + k] : 1.0f, (ok && a.pb) ? (float)a.pb[(long)n * K + k] : 0.0f};
    }
    __syncthreads();

    // this thread's slots: slot q = tid + 512 i -> (row q >> 3, float4 q & 7 of the 32-position half-stage)
    int offG[SM], offX[SM], ldsG[SM], ldsX[SM];
    const int c4 = tid & 7;
#pragma unroll
    for (int i = 0; i < SM; ++i) {
        const int row = (tid + PWSS_THREADS * i) >> 3;
        const bool okm = row < mtn * 32 && m0 + row < M, okk = row < ktn * 32 && k0 + row < K;
        offG[i] = okm ? ((m0 + row) * Q + c4 * 4) * ES : PWSW_OOB;
        offX[i] = okk ? ((k0 + row) * Q + c4 * 4) * ES : PWSW_OOB;
        ldsG[i] = row < GR ? row * PWSS_PITCH + c4 * 8 : -1;        // -1: beyond the image, nothing to write
        ldsX[i] = row < XR ? row * PWSS_PITCH + c4 * 8 : -1;
    }
    auto base = [&](const void* p, long rows) { return (char*)const_cast<void*>(p) + (long)n * rows * Q * ES; };
    __amdgpu_buffer_rsrc_t rg = cfn_rsrc(base(a.gy, M), (unsigned)((long)M * Q * ES));
    __amdgpu_buffer_rsrc_t ry = cfn_rsrc(base(HASY ? a.y : a.gy, M), (unsigned)((long)M * Q * ES));
    __amdgpu_buffer_rsrc_t rx = cfn_rsrc(base(a.x, K), (unsigned)((long)K * Q * ES));

    const int pbeg = strip * a.per, pend = min(pbeg + a.per, Q);
    const int nh = pend > pbeg ? (pend - pbeg + PWSS_P - 1) / PWSS_P : 0;

    // this wave's tiles: tiles wave, wave + 8, ... of the group's mtn x ktn (row major), or (BLK) the block (wave / WC, wave % WC) of BR x BC tiles
    constexpr bool BLK = BR > 0;
    static_assert(!BLK || BR * BC == TPW, "block = the wave's tile slots");
    int aoff[TPW], boff[TPW], tix[TPW];                           // tix: row-major tile index in the group (the workspace / gW slot)
    bool live[TPW];
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
        int ti, tj;
        if constexpr (BLK) {
            const int WC = (ktn + BC - 1) / BC;
            ti = (wave / WC) * BR + tt / BC; tj = (wave % WC) * BC + tt % BC;
            live[tt] = ti < mtn && tj < ktn;                          // wave uniform (a wave beyond the last block row owns nothing)
        } else {
            const int t = wave + (PWSS_THREADS / 64) * tt;            // tiles dealt round robin (groups of < 8 tiles still leave waves without one: `uneven` below)
            live[tt] = t < mtn * ktn;                                 // wave uniform
            ti = live[tt] ? t / ktn : 0; tj = live[tt] ? t - ti * ktn : 0;
        }
        if (!live[tt]) { ti = 0; tj = 0; }
        tix[tt] = ti * ktn + tj;
        aoff[tt] = (ti * 32 + r) * PWSS_PITCH + kg * 16;
        boff[tt] = NS * gimg + (tj * 32 + r) * PWSS_PITCH + kg * 16;
    }
    f16v acc[TPW];
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) acc[tt] = (f16v)0.0f;

    // prologue coefficients of this thread's slots, in REGISTERS for the whole strip.  Round 3 re-read them from the LDS tables inside every
    // conversion step; the ISA-level bisection of the round-3 race (profiles/r05_race_bisect.txt) ended at exactly that pattern: in a
    // wave that converts while the other wave of its SIMD multiplies, the packed FMA behind `ds_read_b64/b96 coefficient; s_waitcnt lgkmcnt(0)` saw
    // the high register of the returned pair as zero in lanes 48-63 (a = swish(A x) instead of swish(A x + B), g' without its 2 gq y term).
    // (SM = 4 variants -- 96 accumulator + 96 operand registers -- would spill 40-126 registers with 20 more live ones: they keep the table reads;
    // all their waves own tiles and convert in the same phase, see `uneven` below; round 6: what they read passes through cfn_settle, so no packed
    // instruction is the first reader of a just-returned LDS pair -- tools/pkfma_ldsret_scan.py finds no class-S site inside an MFMA loop any more)
    constexpr bool KREG = SM <= 2;
    float4 kG[KREG ? SM : 1];
    float2 kX[KREG ? SM : 1];
    if constexpr (KREG) {
#pragma unroll
        for (int i = 0; i < SM; ++i) {
            const int row = (tid + PWSS_THREADS * i) >> 3;
            kG[i] = row < GR ? cG[row] : float4{0.f, 0.f, 1.f, 0.f};
            kX[i] = row < XR ? cX[row] : float2{1.f, 0.f};
        }
    }
    f4v rG[2][SM], rY[2][SM], rX[2][SM];
    auto ld4 = [&](__amdgpu_buffer_rsrc_t rs, int vo, int so) -> f4v {
        if constexpr (ES == 4) {
            return __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0));
        } else {
            typedef unsigned u2v_ __attribute__((ext_vector_type(2)));
            const u2v_ d = __builtin_amdgcn_raw_buffer_load_b64(rs, vo, so, 0);
            const unsigned d0 = d.x, d1 = d.y;
            return (f4v){pwsw_lo(d0), pwsw_hi(d0), pwsw_lo(d1), pwsw_hi(d1)};
        }
    };
    auto issue = [&](int h, f4v (&g)[SM], f4v (&yy)[SM], f4v (&xx)[SM]) {
        const int p0 = pbeg + h * PWSS_P;
        const bool hv = h < nh;                                   // uniform; beyond the strip: nothing is fetched
        const bool pv = hv && p0 + c4 * 4 < pend;                 // a float4 whose first element is inside the strip is fetched (fp32: any dword address)
        const int so = hv ? p0 * ES : 0;
#pragma unroll
        for (int i = 0; i < SM; ++i) {
            g[i] = ld4(rg, pv ? offG[i] : PWSW_OOB, so);
            if (HASY) yy[i] = ld4(ry, pv ? offG[i] : PWSW_OOB, so);
            xx[i] = ld4(rx, pv ? offX[i] : PWSW_OOB, so);
        }
    };
    auto convert = [&](int h, unsigned char* buf, const f4v (&g)[SM], const f4v (&yy)[SM], const f4v (&xx)[SM]) {
        // element e of this thread's float4 is inside the strip (odd volumes -- 65 x 7 x 7 in the coarse stream: the float4 that straddles the
        // end of a row carries the head of the next row; whole float4s beyond the strip were not fetched)
        // Q % 4 == 0 (every fine-stream shape): a float4 is inside or outside as a whole, masking the constant term of the g' operand is enough
        // (loads beyond the strip returned zeros, a zero g' row kills whatever act(B) the x operand holds there)
        const int pe = pbeg + h * PWSS_P + c4 * 4;
        const bool ragged = (Q & 3) != 0;                           // uniform
        const float vm0 = (h < nh && pe < pend) ? 1.0f : 0.0f;
        float vm[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) vm[e] = (h < nh && pe + e < pend) ? 1.0f : 0.0f;
#pragma unroll
        for (int i = 0; i < SM; ++i) {
            if (ldsG[i] >= 0) {
                const float4 c = KREG ? kG[KREG ? i : 0] : cfn_settle3(cG[(tid + PWSS_THREADS * i) >> 3]);
                const float c0 = c.x * vm0;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = fmaf(g[i][e], c.z, c0);
                    if (HASY) v[e] = fmaf(yy[i][e], c.y, v[e]);
                }
                if (ragged) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = vm[e] != 0.0f ? v[e] : 0.0f;
                }
                unsigned p0[NS], p1[NS];
                pwsw_split<NS>(v[0], v[1], p0);
                pwsw_split<NS>(v[2], v[3], p1);
#pragma unroll
                for (int sp = 0; sp < NS; ++sp)
                    *reinterpret_cast<uint2*>(buf + sp * gimg + ldsG[i]) = uint2{p0[sp], p1[sp]};
            }
            if (ldsX[i] >= 0) {
                const float2 c = KREG ? kX[KREG ? i : 0] : cfn_settle(cX[(tid + PWSS_THREADS * i) >> 3]);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = cfn_act<ACT>(fmaf(xx[i][e], c.x, c.y));
                if (ragged) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = vm[e] != 0.0f ? v[e] : 0.0f;
                }
                unsigned p0[NS], p1[NS];
                pwsw_split<NS>(v[0], v[1], p0);
                pwsw_split<NS>(v[2], v[3], p1);
#pragma unroll
                for (int sp = 0; sp < NS; ++sp)
                    *reinterpret_cast<uint2*>(buf + NS * gimg + sp * ximg + ldsX[i]) = uint2{p0[sp], p1[sp]};
            }
        }
    };
    auto mfma = [&](const unsigned char* buf) {
        if constexpr (BLK) {
            // the operand of the block's short side is read once per k-block and kept, the long side streams past it
            constexpr bool HOLD_A = BR <= BC;
            constexpr int NH = HOLD_A ? BR : BC, NSTR = HOLD_A ? BC : BR;
#pragma unroll
            for (int kb = 0; kb < PWSS_P / 16; ++kb) {
                bf16x8 H[NH][NS];
#pragma unroll
                for (int i = 0; i < NH; ++i)
#pragma unroll
                    for (int sp = 0; sp < NS; ++sp)
                        H[i][sp] = HOLD_A ? *reinterpret_cast<const bf16x8*>(buf + sp * gimg + aoff[i * BC] + kb * 32)
                                          : *reinterpret_cast<const bf16x8*>(buf + sp * ximg + boff[i] + kb * 32);
#pragma unroll
                for (int j = 0; j < NSTR; ++j) {
                    bf16x8 S[NS];
#pragma unroll
                    for (int sp = 0; sp < NS; ++sp)
                        S[sp] = HOLD_A ? *reinterpret_cast<const bf16x8*>(buf + sp * ximg + boff[j] + kb * 32)
                                       : *reinterpret_cast<const bf16x8*>(buf + sp * gimg + aoff[j * BC] + kb * 32);
#pragma unroll
                    for (int i = 0; i < NH; ++i) {
                        const int tt = HOLD_A ? i * BC + j : j * BC + i;
                        if (live[tt]) {
#define PWSS_MMB(SA, SB) acc[tt] = HOLD_A ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(H[i][SA], S[SB], acc[tt], 0, 0, 0) \
                                          : __builtin_amdgcn_mfma_f32_32x32x16_bf16(S[SA], H[i][SB], acc[tt], 0, 0, 0)
                            if constexpr (NS == 3) { PWSS_MMB(2, 0); PWSS_MMB(0, 2); PWSS_MMB(1, 1); }
                            if constexpr (NS >= 2) { PWSS_MMB(1, 0); PWSS_MMB(0, 1); }
                            PWSS_MMB(0, 0);
#undef PWSS_MMB
                        }
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
            if (live[tt]) {
#pragma unroll
                for (int kb = 0; kb < PWSS_P / 16; ++kb) {
                    bf16x8 A[NS], B[NS];
#pragma unroll
                    for (int sp = 0; sp < NS; ++sp) {
                        A[sp] = *reinterpret_cast<const bf16x8*>(buf + sp * gimg + aoff[tt] + kb * 32);
                        B[sp] = *reinterpret_cast<const bf16x8*>(buf + sp * ximg + boff[tt] + kb * 32);
                    }
#define PWSS_MM(SA, SB) acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[SA], B[SB], acc[tt], 0, 0, 0)
                    if constexpr (NS == 3) { PWSS_MM(2, 0); PWSS_MM(0, 2); PWSS_MM(1, 1); }
                    if constexpr (NS >= 2) { PWSS_MM(1, 0); PWSS_MM(0, 1); }
                    PWSS_MM(0, 0);
#undef PWSS_MM
                }
            }
        }
    };

    unsigned char* buf1 = buf0 + bufb;
    if constexpr (PIPE) {
        static_assert(!PIPE || (BLK && NS == 3), "the pipelined loop is built on the block dealing and the 6-term product");
        float4 kG4[SM];
        float2 kX4[SM];
#pragma unroll
        for (int i = 0; i < SM; ++i) {
            const int row = (tid + PWSS_THREADS * i) >> 3;
            kG4[i] = row < GR ? cG[row] : float4{0.f, 0.f, 1.f, 0.f};
            kX4[i] = row < XR ? cX[row] : float2{1.f, 0.f};
        }
        constexpr bool HOLD_A = BR <= BC;
        constexpr int NH = HOLD_A ? BR : BC, NSTR = HOLD_A ? BC : BR;
        constexpr int NSUB = 12 * SM, NMF = 6 * TPW * (PWSS_P / 16);           // conversion pieces, MFMAs per half-stage
        // multiply half-stage `bufM` and convert half-stage hc (register set g / yy / xx) into bufC
        auto pipe_step = [&](const unsigned char* bufM, int hc, unsigned char* bufC, const f4v (&g)[SM], const f4v (&yy)[SM], const f4v (&xx)[SM]) __attribute__((always_inline)) {
            const int pe = pbeg + hc * PWSS_P + c4 * 4;
            const bool ragged = (Q & 3) != 0;                           // uniform
            const float vm0 = (hc < nh && pe < pend) ? 1.0f : 0.0f;
            float vm[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) vm[e] = (hc < nh && pe + e < pend) ? 1.0f : 0.0f;
            float v[4];
            unsigned p0[NS], p1[NS];
            // piece sidx: unit u = sidx / 6 (slot u >> 1, g' operand or x operand), pair q, stage st of the split; the unit's LDS writes behind its last piece
            auto sub = [&](int sidx) __attribute__((always_inline)) {
                const int u = sidx / 6, w = sidx % 6, q = w & 1, st = w >> 1, i = u >> 1;
                const bool isX = (u & 1) != 0;
                if (st == 0 && q == 0) {
                    if (!isX) {
                        const float4 c = kG4[i];
                        const float c0 = c.x * vm0;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = fmaf(g[i][e], c.z, c0);
                            if (HASY) v[e] = fmaf(yy[i][e], c.y, v[e]);
                        }
                    } else {
                        const float2 c = kX4[i];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = cfn_act<ACT>(fmaf(xx[i][e], c.x, c.y));
                    }
                    if (ragged) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = vm[e] != 0.0f ? v[e] : 0.0f;
                    }
                }
                unsigned (&pp)[NS] = q == 0 ? p0 : p1;
                if (st > 0) { v[2 * q] -= pwsw_lo(pp[st - 1]); v[2 * q + 1] -= pwsw_hi(pp[st - 1]); }
                pp[st] = pwsw_pack(v[2 * q], v[2 * q + 1]);
                if (w == 5) {
                    const int off = isX ? ldsX[i] : ldsG[i];
                    if (off >= 0) {
#pragma unroll
                        for (int sp = 0; sp < NS; ++sp)
                            *reinterpret_cast<uint2*>(bufC + (isX ? NS * gimg + sp * ximg : sp * gimg) + off) = uint2{p0[sp], p1[sp]};
                    }
                }
            };
            auto subs_of = [&](int m) __attribute__((always_inline)) {     // the pieces pinned behind MFMA m of the half-stage
#pragma unroll
                for (int sidx = NSUB * m / NMF; sidx < NSUB * (m + 1) / NMF; ++sidx) sub(sidx);
            };
#pragma unroll
            for (int kb = 0; kb < PWSS_P / 16; ++kb) {
                bf16x8 H[NH][NS];
#pragma unroll
                for (int i = 0; i < NH; ++i)
#pragma unroll
                    for (int sp = 0; sp < NS; ++sp)
                        H[i][sp] = HOLD_A ? *reinterpret_cast<const bf16x8*>(bufM + sp * gimg + aoff[i * BC] + kb * 32)
                                          : *reinterpret_cast<const bf16x8*>(bufM + sp * ximg + boff[i] + kb * 32);
#pragma unroll
                for (int j = 0; j < NSTR; ++j) {
                    bf16x8 S[NS];
#pragma unroll
                    for (int sp = 0; sp < NS; ++sp)
                        S[sp] = HOLD_A ? *reinterpret_cast<const bf16x8*>(bufM + sp * ximg + boff[j] + kb * 32)
                                       : *reinterpret_cast<const bf16x8*>(bufM + sp * gimg + aoff[j * BC] + kb * 32);
#pragma unroll
                    for (int i = 0; i < NH; ++i) {
                        const int tt = HOLD_A ? i * BC + j : j * BC + i;
                        const int m0f = ((kb * NSTR + j) * NH + i) * 6;    // first MFMA of this tile in the half-stage
                        constexpr int SA[6] = {2, 0, 1, 1, 0, 0}, SB[6] = {0, 2, 1, 0, 1, 0};
                        if (live[tt]) {
#pragma unroll
                            for (int tm = 0; tm < 6; ++tm) {
                                acc[tt] = HOLD_A ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(H[i][SA[tm]], S[SB[tm]], acc[tt], 0, 0, 0)
                                                 : __builtin_amdgcn_mfma_f32_32x32x16_bf16(S[SA[tm]], H[i][SB[tm]], acc[tt], 0, 0, 0);
                                subs_of(m0f + tm);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        } else {
#pragma unroll
                            for (int tm = 0; tm < 6; ++tm) subs_of(m0f + tm);
                        }
                    }
                }
            }
        };
        issue(0, rG[0], rY[0], rX[0]);
        issue(1, rG[1], rY[1], rX[1]);
        convert(0, buf0, rG[0], rY[0], rX[0]);
        issue(2, rG[0], rY[0], rX[0]);
        for (int h = 0; h < nh; h += 2) {
            __syncthreads();                                        // half-stage h is complete in buf0; buf1's readers have finished
            pipe_step(buf0, h + 1, buf1, rG[1], rY[1], rX[1]);      // h + 1 == nh: zeros (masked loads, masked constant term)
            issue(h + 3, rG[1], rY[1], rX[1]);
            __syncthreads();
            pipe_step(buf1, h + 2, buf0, rG[0], rY[0], rX[0]);
            issue(h + 4, rG[0], rY[0], rX[0]);
        }
    } else {
    // uneven: the tile count of the group is not a multiple of the wave count, i.e. in the last tile slot some waves multiply and some do not.  The
    // round-3 failure needed waves that convert while others multiply (with an extra barrier behind every multiply phase: 0 of 1,200 passes, same
    // binary otherwise); such groups (layer 2's 48 x 108 = 8 tiles is even; 48 x 96 = 6 tiles, the coarse stream's fusion convs) pay two more
    // barriers per pair of half-stages and keep every wave in the same phase.
    const bool uneven = BLK || (mtn * ktn) % (PWSS_THREADS / 64) != 0;   // workgroup uniform (blocks: a ragged grid leaves waves with fewer tiles)
    issue(0, rG[0], rY[0], rX[0]);
    issue(1, rG[1], rY[1], rX[1]);
    for (int h = 0; h < nh; h += 2) {
        convert(h, buf0, rG[0], rY[0], rX[0]);
        __syncthreads();
        issue(h + 2, rG[0], rY[0], rX[0]);
        mfma(buf0);
        if (uneven) __syncthreads();
        convert(h + 1, buf1, rG[1], rY[1], rX[1]);                // h + 1 == nh: zeros (masked loads, masked constant term)
        __syncthreads();
        issue(h + 3, rG[1], rY[1], rX[1]);
        mfma(buf1);
        if (uneven) __syncthreads();
    }
    }

    if (a.ws) {     // partial tiles of this workgroup: [group][n * nstrips + strip][tile][element e][lane]
        const int S = a.N * a.nstrips, TPG = a.mtg * a.ktg;
        float* wb = a.ws + (((size_t)(mgi * a.kgroups + kgi) * S + (size_t)n * a.nstrips + strip) * TPG) * 1024 + lane;
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
            if (live[tt]) {
                const int t = tix[tt];
#pragma unroll
                for (int e = 0; e < 16; ++e) wb[(size_t)t * 1024 + e * 64] = acc[tt][e];
            }
        }
        return;
    }
    // tile element e of lane (r, kg): row (e & 3) + 8 (e >> 2) + 4 kg, column r
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
        if (live[tt]) {
            const int t = tix[tt], ti = t / ktn, tj = t - ti * ktn;
            const int k = k0 + tj * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + ti * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg;
                if (m < M && k < K) cfn_add64(&a.gw[(long)m * K + k], (double)acc[tt][e]);
            }
        }
    }
}

// Second phase of the staged weight gradient when the workgroups' tiles went to the workspace: gW[m][k] += sum over the S = N x nstrips
// partial tiles in a FIXED order (fp64).  A workgroup = one 64-element row of a tile (element e, all lanes) x 4 slices of S; the fp64 atomics
// this replaces were 12-15 % of the first phase (6-11 M per launch, 256 workgroups onto the same M x K addresses), and the result no longer
// depends on the order in which workgroups finish.
__global__ __launch_bounds__(1024) void pws_wgrad_reduce_kernel(const WssArgs a) {
    // a workgroup = a quarter of a tile (256 floats = 64 float4) x 16 slices of S: every thread has ALL its (<= 16) 16-byte loads in flight at
    // once (the first version -- 64 floats x 4 slices per workgroup, 8 four-byte loads in flight -- took 16 us per launch, 0.8 ms per step)
    __shared__ double part[16][64][4];
    typedef float __attribute__((ext_vector_type(4))) rf4;
    const int l4 = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int TPG = a.mtg * a.ktg, S = a.N * a.nstrips;
    unsigned L = blockIdx.x;
    const int qt = L & 3; L >>= 2;
    const int t = L % TPG; L /= TPG;
    const int kgi = L % a.kgroups, mgi = L / a.kgroups;
    int mt0, mtn, kt0, ktn;
    pwsw_run(a.mt32, a.mgroups, mgi, mt0, mtn);
    pwsw_run(a.kt32, a.kgroups, kgi, kt0, ktn);
    if (t >= mtn * ktn) return;                                       // (whole workgroup)
    const float* wb = a.ws + (((size_t)(mgi * a.kgroups + kgi) * S) * TPG + t) * 1024 + qt * 256 + l4 * 4;
    const size_t sstride = (size_t)TPG * 1024;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int s0 = sg; s0 < S; s0 += 256) {                             // S <= 256 in practice: one trip
        rf4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = s0 + 16 * u < S ? *reinterpret_cast<const rf4*>(wb + (size_t)(s0 + 16 * u) * sstride) : (rf4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 16; ++u) { acc[0] += (double)v[u].x; acc[1] += (double)v[u].y; acc[2] += (double)v[u].z; acc[3] += (double)v[u].w; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) part[sg][l4][i] = acc[i];
    __syncthreads();
    if (sg == 0) {
        const int ti = t / ktn, tj = t - ti * ktn;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = qt * 256 + l4 * 4 + i, e = f >> 6, lane = f & 63, kg = lane >> 5, r = lane & 31;
            const int m = (mt0 + ti) * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg, k = (kt0 + tj) * 32 + r;
            double sum = 0.0;
#pragma unroll
            for (int g = 0; g < 16; ++g) sum += part[g][l4][i];
            if (m < a.M && k < a.K) a.gw[(long)m * a.K + k] += sum;
        }
    }
}

// workspace of the two-phase weight gradient: one grow-only device buffer per (device, stream) (the engine runs one stream per process; a second
// stream or a second device gets its own buffer).  No allocation while the stream is being captured into a graph: the launch then keeps the
// atomics.  A buffer that has been handed out is NEVER freed: a captured graph may have its address baked in; growing allocates a new
// buffer of at least twice the size and retires the old one (geometric growth bounds the retired bytes by the live buffer's size).
static float* pwss_workspace(size_t bytes, hipStream_t st) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, std::pair<float*, size_t>> bufs;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    std::lock_guard<std::mutex> lk(mu);
    auto& b = bufs[std::make_pair(dev, st)];
    if (b.second >= bytes) return b.first;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;
    const size_t want = bytes > 2 * b.second ? bytes : 2 * b.second;
    float* p = nullptr;
    if (hipMalloc(&p, want) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    b = {p, want};                                  // (the previous buffer, if any, stays allocated)
    return p;
}

static size_t pwss_lds(int mtg, int ktg, int NS) {
    return (size_t)2 * NS * 32 * (mtg + ktg) * PWSS_PITCH + (size_t)32 * mtg * 16 + (size_t)32 * ktg * 8;
}

template <int SM, int TPW, int NS, int ES = 4, int BR = 0, int BC = 0, bool PIPE = false>
static int pwss_launch(const WssArgs& a, unsigned blocks, size_t lds, hipStream_t st) {
#define PWSS_GO(AV, HY)                                                                                                    \
    do {                                                                                                                   \
        auto k = pws_wgrad_staged_kernel<SM, TPW, AV, HY, NS, ES, BR, BC, PIPE>;                                                           \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(k, dim3(blocks), dim3(PWSS_THREADS), lds, st, a);                                               \
    } while (0)
#define PWSS_ACT(HY)                                                                                                       \
    do {                                                                                                                   \
        if (a.act == CFN_ACT_RELU) PWSS_GO(CFN_ACT_RELU, HY);                                                              \
        else if (a.act == CFN_ACT_SWISH) PWSS_GO(CFN_ACT_SWISH, HY);                                                       \
        else PWSS_GO(CFN_ACT_NONE, HY);                                                                                    \
    } while (0)
    if (a.y) PWSS_ACT(true); else PWSS_ACT(false);
#undef PWSS_ACT
#undef PWSS_GO
    return cfn_check_launch("pwconv_bwd_weight(split bf16, staged)");
}

static int pwss_launch_any(const WssArgs& a, unsigned blocks, size_t lds, int tiles, int NS, hipStream_t st);

// -1 = shape not handled
// terms: 3 / 6 (fp32 tensors, split), 1 (bf16 tensors)
static int pwss_try(const void* gy, const void* y, const double* gs, const double* gq, const double* gsc, const void* x,
                    const double* pa, const double* pb, int act, double* gw, int N, int M, int K, int Q, int terms, hipStream_t st) {
    const int NS = terms == 1 ? 1 : (terms == 3 ? 2 : 3);
    WssArgs a = {gy, gq ? y : nullptr, gs, gq, gsc, x, pa, pb, gw, N, M, K, Q, act};
    a.mt32 = cfn_cdiv(M, 32); a.kt32 = cfn_cdiv(K, 32);
    // tile groups: least operand traffic  kgroups * M rows (x2 with y) + mgroups * K rows  under the limits of one workgroup:
    // <= 8 tiles (256 rows) per side, <= 48 tiles, both double-buffered LDS images within 160 KiB
    long best = -1;
    for (int mg = 1; mg <= a.mt32; ++mg)
        for (int kg = 1; kg <= a.kt32; ++kg) {
            const int mtg = cfn_cdiv(a.mt32, mg), ktg = cfn_cdiv(a.kt32, kg);
            if (mtg > 8 || ktg > 8 || mtg * ktg > 48 || pwss_lds(mtg, ktg, NS) > 160 * 1024) continue;
            const long cost = ((long)kg * M * (a.y ? 2 : 1) + (long)mg * K) * 64 + mg * kg;
            if (best < 0 || cost < best) { best = cost; a.mgroups = mg; a.kgroups = kg; a.mtg = mtg; a.ktg = ktg; }
        }
    if (best < 0) return -1;
    const long groups = (long)N * a.mgroups * a.kgroups;
    static const int wg_env = getenv("CFN_PWSS_WGS") ? atoi(getenv("CFN_PWSS_WGS")) : 0;
    long strips = (wg_env > 0 ? wg_env : 256) / groups;          // one 8-wave workgroup per CU
    if (strips < 1) strips = 1;
    const long maxs = cfn_cdiv(Q, 4 * PWSS_P);                    // >= 4 half-stages per strip
    if (strips > maxs) strips = maxs;
    a.per = cfn_cdiv(cfn_cdiv(Q, strips), PWSS_P) * PWSS_P;
    a.nstrips = cfn_cdiv(Q, a.per);
    const unsigned blocks = (unsigned)(groups * a.nstrips);
    const size_t lds = pwss_lds(a.mtg, a.ktg, NS);
    const int tiles = a.mtg * a.ktg;
    static const int ws_env = getenv("CFN_PWSS_WS") ? atoi(getenv("CFN_PWSS_WS")) : 1;     // 0: fp64 atomics from every workgroup
    a.ws = (ws_env && N * a.nstrips >= 8) ? pwss_workspace((size_t)blocks * tiles * 4096, st) : nullptr;
    const int rc1 = pwss_launch_any(a, blocks, lds, tiles, NS, st);
    if (rc1 != 0 || !a.ws) return rc1;
    hipLaunchKernelGGL(pws_wgrad_reduce_kernel, dim3((unsigned)(a.mgroups * a.kgroups * tiles * 4)), dim3(1024), 0, st, a);
    return cfn_check_launch("pwconv_bwd_weight(split bf16, staged) reduce");
}

static int pwss_launch_any(const WssArgs& a, unsigned blocks, size_t lds, int tiles, int NS, hipStream_t st) {
    if (NS == 1) {
        if (a.mtg <= 4 && a.ktg <= 4) return pwss_launch<2, 2, 1, 2>(a, blocks, lds, st);
        if (tiles <= 24) return pwss_launch<4, 3, 1, 2>(a, blocks, lds, st);
        return pwss_launch<4, 6, 1, 2>(a, blocks, lds, st);
    }
    // <= 8 tiles (layer 2: 48 x 108): ONE tile per wave -- with two per wave only four of the eight waves multiply while the other four only stage
    // operands a phase ahead of them: idle matrix pipes, and the configuration of the round-3 race (a staging-only wave's packed FMA read
    // a just-returned LDS coefficient as zero in lanes 48-63; the coefficients are registers now and uneven groups keep every wave in one phase)
    if (a.mtg <= 4 && a.ktg <= 4 && tiles <= 8) return NS == 2 ? pwss_launch<2, 1, 2>(a, blocks, lds, st) : pwss_launch<2, 1, 3>(a, blocks, lds, st);
    if (a.mtg <= 4 && a.ktg <= 4) return NS == 2 ? pwss_launch<2, 2, 2>(a, blocks, lds, st) : pwss_launch<2, 2, 3>(a, blocks, lds, st);
    if (tiles <= 24) {
        // blocks of 1 x 3 / 3 x 1 tiles per wave where the group's grid fits the 8 waves (layer 3: 7 x 3 / 3 x 7 tiles, layer 4: 4 x 6 / 3 x 7)
        static const int blk = getenv("CFN_PWSS_BLOCKS") ? atoi(getenv("CFN_PWSS_BLOCKS")) : 1;
        if (blk && NS == 3) {
            if (a.mtg * cfn_cdiv(a.ktg, 3) <= 8) return blk == 1 ? pwss_launch<4, 3, 3, 4, 1, 3, true>(a, blocks, lds, st) : pwss_launch<4, 3, 3, 4, 1, 3>(a, blocks, lds, st);
            if (cfn_cdiv(a.mtg, 3) * a.ktg <= 8) return blk == 1 ? pwss_launch<4, 3, 3, 4, 3, 1, true>(a, blocks, lds, st) : pwss_launch<4, 3, 3, 4, 3, 1>(a, blocks, lds, st);
        }
        return NS == 2 ? pwss_launch<4, 3, 2>(a, blocks, lds, st) : pwss_launch<4, 3, 3>(a, blocks, lds, st);
    }
    return NS == 2 ? pwss_launch<4, 6, 2>(a, blocks, lds, st) : pwss_launch<4, 6, 3>(a, blocks, lds, st);
}

extern "C" int cfn_pw_split_terms(int terms);

// contiguous pointwise conv (stride 1), M, K >= 48; -1 = not handled (caller falls through to the fp32-MFMA kernels)
int pws_wgrad_try_launch(const float* gy, const float* y, const double* gs, const double* gq, const double* gsc, const float* x,
                         const double* pa, const double* pb, int act, double* gw, int N, int M, int K, int Q, hipStream_t st) {
    const int terms = cfn_pw_split_terms(-1);
    if (terms == 0 || M < 48 || K < 48) return -1;
    const bool ragged = Q % 4 != 0;                               // rows on 4-byte boundaries: the staged kernel only (per-element strip masks)
    if (act != CFN_ACT_NONE && act != CFN_ACT_RELU && act != CFN_ACT_SWISH) return -1;
    if (((uintptr_t)gy | (uintptr_t)(y ? y : gy) | (uintptr_t)x) & 15) return -1;
    if ((long)M * Q * 4 >= (1L << 31) - 64 || (long)K * Q * 4 >= (1L << 31) - 64) return -1;
    static const int direct_env = getenv("CFN_PWSW_DIRECT") ? atoi(getenv("CFN_PWSW_DIRECT")) : 0;
    if (!direct_env || ragged) {
        const int rc = pwss_try(gy, y, gs, gq, gsc, x, pa, pb, act, gw, N, M, K, Q, terms, st);
        if (rc >= 0) return rc;
    }
    if (ragged) return -1;
    WsArgs a = {gy, gq ? y : nullptr, gs, gq, gsc, x, pa, pb, gw, N, M, K, Q, act};
    // tile group per wave: the candidate with the least operand traffic  kgroups * M rows (x2 with y) + mgroups * K rows
    const int mt = cfn_cdiv(M, 32), kt = cfn_cdiv(K, 32);
    static const int tm_env = getenv("CFN_PWSW_TM") ? atoi(getenv("CFN_PWSW_TM")) : 0;
    static const int tn_env = getenv("CFN_PWSW_TN") ? atoi(getenv("CFN_PWSW_TN")) : 0;
    const int cand[3][2] = {{2, 2}, {2, 3}, {3, 2}};
    int best = 0; long best_cost = -1;
    for (int c = 0; c < (a.y ? 2 : 3); ++c) {       // 3 row tiles with two row-side tensors (gy, y) spill
        const long cost = (long)cfn_cdiv(kt, cand[c][1]) * M * (a.y ? 2 : 1) + (long)cfn_cdiv(mt, cand[c][0]) * K;
        if (best_cost < 0 || cost < best_cost) { best = c; best_cost = cost; }
    }
    int TM = cand[best][0], TN = cand[best][1];
    if (tm_env && tn_env) { TM = tm_env; TN = tn_env; }
    if (terms == 3) {
        if (TM == 2 && TN == 2) return pwsw_launch<2, 2, 2>(a, st);
        if (TM == 2 && TN == 3) return pwsw_launch<2, 3, 2>(a, st);
        return pwsw_launch<3, 2, 2>(a, st);
    }
    if (TM == 2 && TN == 2) return pwsw_launch<2, 2, 3>(a, st);
    if (TM == 2 && TN == 3) return pwsw_launch<2, 3, 3>(a, st);
    return pwsw_launch<3, 2, 3>(a, st);
}

// bf16 tensors (cfn_pwconv_bwd_weight_bf16), M, K >= 48; -1 = not handled (the caller keeps its direct-operand kernel)
int pwss_wgrad_try_bf16(const uint16_t* gy, const uint16_t* y, const double* gs, const double* gq, const double* gsc, const uint16_t* x,
                        const double* pa, const double* pb, int act, double* gw, int N, int M, int K, int Q, hipStream_t st) {
    static const int off = getenv("CFN_PWB_WG_DIRECT") ? atoi(getenv("CFN_PWB_WG_DIRECT")) : 0;
    // few rows (layer 1: 54 + 24): a 32-position half-stage moves ~5 KB per barrier -- measured 0.60 vs 0.34-0.50 ms: declined
    if (off || Q % 4 != 0 || M < 48 || K < 48) return -1;
    if (act != CFN_ACT_NONE && act != CFN_ACT_RELU && act != CFN_ACT_SWISH) return -1;
    if (((uintptr_t)gy | (uintptr_t)(y ? y : gy) | (uintptr_t)x) & 7) return -1;
    if ((long)M * Q * 2 >= (1L << 31) - 64 || (long)K * Q * 2 >= (1L << 31) - 64) return -1;
    return pwss_try(gy, y, gs, gq, gsc, x, pa, pb, act, gw, N, M, K, Q, 1, st);
}
