// Pointwise (1x1x1) forward contraction on fp32 tensors with split-bf16 arithmetic (pwsplit.hip's 6-term product), REGISTER-RESIDENT
// WEIGHTS, two k slices per row tile: the deep-contraction / few-row shapes (x3d_fine.py:100-105 conv3 of layer 3: 216 -> 96 @14x14),
// which pws_kernel loses on (its VALU work grows with K) and which therefore ran on the fp32-MFMA pw_deep_kernel.
//
// pws_kernel gives a wave ALL output rows of its 32 positions and runs its phases -- load, activate + split, MFMAs, epilogue -- one
// after the other.  Here the roles are turned round (prototype and its ablations: tools/probe/pwreg_prototype):
//   * wave w = (row tile mt = w % 4, k slice ks = w / 4) keeps the weights of (mt, ks) in registers, fetched coalesced through LDS and
//     split once into 3 bf16 terms (7 k-blocks x 3 x 4 = 84 VGPRs at K = 216): no weight image in LDS, no LDS read per weight operand;
//   * the 8 waves share the ACTIVATIONS of a 32-position tile: wave w loads k-blocks w and w + 8 (loads two tiles ahead, static
//     register sets), applies the prologue, splits ONCE and publishes the three operand images in LDS (double buffered, ONE barrier
//     per tile); every wave reads its slice back as 16-byte MFMA operands, one k-block ahead of the MFMAs;
//   * the ks = 1 wave hands its 32x32 partial result to the ks = 0 wave of the same row tile through LDS (double buffered, picked up
//     behind the NEXT tile's barrier: no extra synchronisation); that wave adds it, takes the statistics in-lane (transposed result:
//     lane = output channel), puts the tile into its scratch and drains it -- 16-byte stores of whole 128-byte lines -- between the
//     MFMA groups of the next tile;
//   * two tiles per loop trip: LDS buffers, scratch halves and load sets are static (no register moves, no waterfall loops).
// One-slice mode (forward; K <= 112, 128 < M <= 256: layer-3 conv1, 96 -> 216; K <= 192 in row slabs of <= 8 row tiles for M <= 512: layer-4
// conv1, 192 -> 432 as 2 x 7 row tiles with 144 weight registers per wave): wave = row tile, the whole contraction in its registers, no partner:
// 96 -> 216 0.143-0.144 ms against 0.157-0.158 ms (pws_kernel), 192 -> 432 @7x7 0.131-0.133 ms against 0.217-0.221 ms (pw_deep_kernel).
// Two-slice shapes: stride 1, 128 < K <= 224 (an even number of k-blocks), 32 < M <= 128, Q % 4 == 0; forward (layer-3 conv3: 216 -> 96) and the
// data gradient without act' epilogue / compact shortcut gradient (layer-3 conv1: contraction over its 216 output channels, two staged
// tensors); one-slice shapes without slabs also WITH the act' epilogue (layer-3 conv3 data gradient, K = 96 -> 216 rows: the forward input
// of the tile is fetched in whole lines during the tile's MFMAs and crosses the wave's scratch into the lane = channel layout; 0.248 -> 0.234 ms
// against pws_kernel); everything else stays with pws_kernel / pw_deep_kernel.  Measured (8 clips x T = 256, @14x14, same box, against the
// fp32-MFMA pw_deep_kernel): forward 0.176-0.179 vs 0.218-0.219 ms, data gradient 0.214-0.215 vs 0.232-0.233 ms.
#include "pw_common.h"
#include <stdlib.h>
#include <type_traits>

typedef __bf16 bf16x8k __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2k __attribute__((ext_vector_type(2)));
typedef float f2k __attribute__((ext_vector_type(2)));
typedef unsigned u4k __attribute__((ext_vector_type(4)));

#define PWK_WAVES 8
#define PWK_OOB 0x40000000

__device__ __forceinline__ float pwk_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float pwk_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ unsigned pwk_pack(float lo, float hi) {
    const bf16x2k b = __builtin_convertvector((f2k){lo, hi}, bf16x2k);      // v_cvt_pk_bf16_f32 (round to nearest even)
    return __builtin_bit_cast(unsigned, b);
}
// 8 fp32 values -> three 16-byte operands (terms 1..3 of each value, 8 consecutive k)
__device__ __forceinline__ void pwk_split8(const float (&v)[8], u4k (&t)[3]) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        float a = v[2 * h], b = v[2 * h + 1];
        const unsigned p0 = pwk_pack(a, b);
        a -= pwk_lo(p0); b -= pwk_hi(p0);
        const unsigned p1 = pwk_pack(a, b);
        a -= pwk_lo(p1); b -= pwk_hi(p1);
        t[0][h] = p0; t[1][h] = p1; t[2][h] = pwk_pack(a, b);
    }
}

// MODE = PW_DGRAD (no act' epilogue, no compact shortcut gradient): the contraction runs over the conv's OUTPUT channels, w is (K, M) row
// major, the staged operand is g' = gsc gy + gs + 2 gq y (TWO: with the y term).
// KS = 1 (shallow contraction, many rows: K <= 112, 128 < M <= 256, layer-3 conv1 forward 96 -> 216): wave w = row tile w, the whole
// contraction in its registers, no partner.
template <int NKB, int KS, int MODE, int ACT, bool STATS, bool TWO>
__global__ __launch_bounds__(64 * PWK_WAVES) void pwk_kernel(const PwArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KP = 16 * NKB;
    constexpr int NKS = (NKB + KS - 1) / KS;                                // k-blocks per slice
    constexpr int PITCH = KP * 2 + 16;                                      // bytes per position row of one image: an odd number of 16-byte slots
    static_assert(((PITCH / 16) & 1) == 1, "conflict-free pitch");
    constexpr int IMG = 32 * PITCH;                                         // one term, 32 positions
    constexpr int NST = (NKB + PWK_WAVES - 1) / PWK_WAVES;                  // k-blocks a wave stages per tile (<= 2)
    static_assert(NST <= 2, "at most 16 k-blocks");
    const int tid = threadIdx.x, wave = cfn_uni(tid >> 6), lane = tid & 63, kg = lane >> 5, j = lane & 31;
    const int K = a.K, Mfull = a.M, Q = a.Q;
    const unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    // row slabs (one-slice forward with more than 8 row tiles: layer-4 conv1, 192 -> 432 as 2 x 7 row tiles) run side by side on the
    // same positions: the second slab's activation reads hit the XCD's L2
    const int slab = L % a.mtiles, Lr = L / a.mtiles;
    const int wg = Lr % a.nstrips, n = Lr / a.nstrips;
    const int m0 = slab * a.kres, M = min(a.kres, Mfull - m0);             // this workgroup's rows m0 .. m0 + M - 1
    const int nrt = (M + 31) >> 5;                                          // row tiles (<= 8 / KS)

    unsigned char* Bs = smem;                                               // [2 buffers][3 terms][32 positions][PITCH]
    float4* sP = reinterpret_cast<float4*>(Bs + 2 * 3 * IMG);              // [KP] prologue coefficients (FWD: A, B; DGRAD: gs, 2 gq, gsc)
    const int mt = KS == 2 ? (wave & 3) : wave, ks = KS == 2 ? (wave >> 2) : 0, row = mt * 32 + j;
    float* scr = reinterpret_cast<float*>(sP + KP) + mt * (2 * 32 * 36);    // owner (ks = 0) of row tile mt: [2 tiles][32 channels][36]
    float* red = reinterpret_cast<float*>(sP + KP) + (nrt + mt) * (2 * 32 * 36);   // partial results of the ks = 1 wave: [2 tiles][32][36]
    const bool has_rows = mt * 32 < M;                                      // wave uniform
    const bool owner = ks == 0;

    u4k Wr[NKS][3];
    {
        // this wave's 32 rows x its k slice of w through LDS (coalesced 16-byte loads, then each lane reads its row)
        constexpr int WC = NKS * 16;                                        // columns of the slice
        const int kbase = ks * WC;
        const bool vec = (a.Cin & 3) == 0 && (reinterpret_cast<uintptr_t>(a.w) & 15) == 0;
        if (MODE == PW_FWD) {                                               // w is (M, K): the wave's rows are contiguous runs along k
            constexpr int CH = NKS > 6 ? 6 : NKS, WCC = CH * 16;           // k-blocks per pass (8 waves x 32 rows x 100 floats = 100 KB of LDS)
            float* wtmp = reinterpret_cast<float*>(smem) + wave * (32 * (WCC + 4));
#pragma unroll
            for (int c0 = 0; c0 < NKS; c0 += CH) {
                for (int e = lane; e < 32 * (WCC / 4); e += 64) {
                    const int rr = e / (WCC / 4), k4 = (e - rr * (WCC / 4)) * 4, k = kbase + c0 * 16 + k4;
                    f4v v = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (mt * 32 + rr < M) {
                        const float* src = a.w + (long)(m0 + mt * 32 + rr) * a.Cin + k;
                        if (vec && k + 3 < K) v = *reinterpret_cast<const f4v*>(src);
                        else { if (k < K) v.x = src[0]; if (k + 1 < K) v.y = src[1]; if (k + 2 < K) v.z = src[2]; if (k + 3 < K) v.w = src[3]; }
                    }
                    *reinterpret_cast<f4v*>(wtmp + rr * (WCC + 4) + k4) = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int kbl = 0; kbl < CH; ++kbl) {
                    if (c0 + kbl < NKS) {
                        float v[8];
                        const float* p = wtmp + j * (WCC + 4) + kbl * 16 + kg * 8;
                        const f4v lo = *reinterpret_cast<const f4v*>(p), hi = *reinterpret_cast<const f4v*>(p + 4);
                        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
                        pwk_split8(v, Wr[c0 + kbl]);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        } else {                                                            // w is (K, M): 32 consecutive m per k row
            float* wtmp = reinterpret_cast<float*>(smem) + wave * (WC * 36);
            for (int e = lane; e < WC * 8; e += 64) {
                const int kk = e >> 3, m4 = (e & 7) * 4, k = kbase + kk, m = mt * 32 + m4;
                f4v v = {0.0f, 0.0f, 0.0f, 0.0f};
                if (k < K) {
                    const float* src = a.w + (long)k * a.Cin + m0 + m;
                    if (vec && m + 3 < M) v = *reinterpret_cast<const f4v*>(src);
                    else { if (m < M) v.x = src[0]; if (m + 1 < M) v.y = src[1]; if (m + 2 < M) v.z = src[2]; if (m + 3 < M) v.w = src[3]; }
                }
                *reinterpret_cast<f4v*>(wtmp + kk * 36 + m4) = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int kbl = 0; kbl < NKS; ++kbl) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = wtmp[(kbl * 16 + kg * 8 + i) * 36 + j];
                pwk_split8(v, Wr[kbl]);
            }
        }
    }
    __syncthreads();                                                        // the staging area is re-used: coefficients, activation buffers
    for (int k = tid; k < KP; k += 64 * PWK_WAVES) {
        float4 c = {1.0f, 0.0f, 1.0f, 0.0f};
        if (MODE == PW_FWD) {
            c.x = (k < K && a.pa) ? (float)a.pa[(long)n * K + k] : 1.0f;
            c.y = (k < K && a.pb) ? (float)a.pb[(long)n * K + k] : 0.0f;
        } else {
            c.x = (k < K && a.gs) ? (float)a.gs[(long)n * K + k] : 0.0f;
            c.y = (k < K && a.gq && a.src2) ? 2.0f * (float)a.gq[(long)n * K + k] : 0.0f;
            c.z = (k < K && a.gsc) ? (float)a.gsc[(long)n * K + k] : 1.0f;
        }
        sP[k] = c;
    }
    __syncthreads();

    __amdgpu_buffer_rsrc_t rs = cfn_rsrc(const_cast<float*>(a.src + (long)n * K * Q), (unsigned)((long)K * Q * 4));
    __amdgpu_buffer_rsrc_t rs2 = cfn_rsrc(const_cast<float*>((TWO ? a.src2 : a.src) + (long)n * K * Q), (unsigned)((long)K * Q * 4));
    const int mrows = max(min(32, M - mt * 32), 0);
    __amdgpu_buffer_rsrc_t rd = cfn_rsrc(a.dst + (long)n * Mfull * Q + (long)(m0 + mt * 32) * Q, (unsigned)((long)mrows * Q * 4));
    // DGRAD + STATS: act' epilogue.  dz = e act'(ea x + eb), out = dz ea, sums of dz x and dz per row; x = the conv's forward input
    // (rows = this kernel's output rows).  The lane's channel is fixed for the whole launch: its two coefficients live in registers.
    constexpr bool EPI = MODE == PW_DGRAD && STATS;
    __amdgpu_buffer_rsrc_t rx = cfn_rsrc(const_cast<float*>(EPI ? a.ex + (long)n * Mfull * Q + (long)(m0 + mt * 32) * Q : a.src), EPI ? (unsigned)((long)mrows * Q * 4) : 0u);
    const float cea = (EPI && row < M) ? (float)a.ea[(long)n * Mfull + m0 + row] : 1.0f;
    const float ceb = (EPI && row < M) ? (float)a.eb[(long)n * Mfull + m0 + row] : 0.0f;
    const int ntiles = (Q + 31) / 32, tstep = a.nstrips;
    const int lane_ld = kg * 8 * Q * 4 + j * 4;
    const int rd_off = j * PITCH + kg * 16;                                 // + kb * 32: A operand (row = position j, k = kb*16 + kg*8 + i)
    const int mrow = lane >> 3, mcol = 4 * (lane & 7);                      // memory-side role: row mrow + 8 s, 16 bytes at position mcol
    const int lane_mem = mrow * Q * 4 + mcol * 4;
    float ssum = 0.0f, qsum = 0.0f;

    // staging: wave w loads k-blocks w and w + 8 (position j, channels kb*16 + kg*8 + i), two tiles ahead
    // The row part of a load address is SCALAR and RUNNING (one s_add + one s_min per load): written as `row * Q * 4` per (k-block, i) the
    // compiler hoists all 16 products and their 16 lane predicates out of the tile loop -- 22-39 SGPRs spilled, 23 v_readlane per tile.
    // A row at or beyond K must not enter the scalar offset (the range check  voffset >= num_records - soffset  wraps), so the scalar row
    // is clamped to K - 1: the kg = 0 lanes of a padding row read the finite activations of row K - 1, which meet the zero-padded weight
    // columns k >= K; rows K .. K + 7 reached through the lane offset (8 kg) fail the range check by themselves and read as 0.
    const int so_cap = (K - 1) * Q * 4;
    auto issue = [&](int tile, float (&ld)[NST][8], float (&ld2)[TWO ? NST : 1][8]) {
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int kb = wave + PWK_WAVES * u;
            const bool live = (int)(kb < NKB) & (int)(tile < ntiles);       // wave uniform
            const int vo = ((int)live & (int)(tile * 32 + j < Q)) ? lane_ld : PWK_OOB;
            const int base = cfn_uni(live ? tile * 32 * 4 : 0);
            const int cap = cfn_uni(live ? so_cap + base : 0), q4 = cfn_uni(live ? Q * 4 : 0);
            int so = cfn_uni(live ? kb * 16 * Q * 4 + base : 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {                                   // channel 16 kb + i (+ 8 kg through the lane offset)
                const int sc = min(so, cap);
                ld[u][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, sc, 0));
                if (TWO) ld2[u][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs2, vo, sc, 0));
                so += q4;
            }
        }
    };
    auto wsync = [&]() {                                                   // LDS ops of a wave run in order; only the compiler is told
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    int pq0 = -1, last_buf = 0;                                             // (owner) the tile waiting in the scratch: first position, scratch half
    auto drain_read = [&](int pbuf, int sx) -> f4v { return *reinterpret_cast<const f4v*>(scr + pbuf * (32 * 36) + (mrow + 8 * sx) * 36 + mcol); };
    auto drain_store = [&](f4v v, int sx) {
        const bool ok = 8 * sx + mrow < mrows && pq0 + mcol < Q;
        cfn_bst128(__builtin_bit_cast(u4k, v), rd, (ok && pq0 >= 0) ? lane_mem + sx * 8 * Q * 4 : PWK_OOB, cfn_uni(pq0 >= 0 ? pq0 * 4 : 0));
    };
    f16v acc;
    f4v xe[EPI ? 4 : 1];                                                    // (EPI) forward input of the tile being multiplied, memory-side layout
    auto xe_issue = [&](int q0) {                                           // whole 128-byte lines: lane = (row mrow + 8 sx, 16 bytes at position mcol)
#pragma unroll
        for (int sx = 0; sx < 4; ++sx) {
            const bool ok = 8 * sx + mrow < mrows && q0 + mcol < Q;
            xe[sx] = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? lane_mem + sx * 8 * Q * 4 : PWK_OOB, cfn_uni(q0 * 4), 0));
        }
    };
    // owner: own accumulators + the partner's partial (scratch half PARV of `red`) -> statistics, transposed tile into scratch half PARV
    auto finish = [&](int PARV, int q0) {
        const bool full = q0 + 32 <= Q && mrows == 32;                      // wave uniform
        float* sb = scr + PARV * (32 * 36);
        const float* pr = red + PARV * (32 * 36);
        const bool chv = j < mrows;
        f4v s1v = {0.0f, 0.0f, 0.0f, 0.0f}, s2v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (EPI) {                                                          // x of this tile: memory-side registers -> scratch -> lane = channel
#pragma unroll
            for (int sx = 0; sx < 4; ++sx) *reinterpret_cast<f4v*>(sb + (mrow + 8 * sx) * 36 + mcol) = xe[sx];
            wsync();
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f4v o = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
            if (KS == 2) o += *reinterpret_cast<const f4v*>(pr + j * 36 + 8 * g + 4 * kg);
            const float gmask = (full || (chv && q0 + 8 * g + 4 * kg < Q)) ? 1.0f : 0.0f;
            if (EPI) {
                const f4v xs = *reinterpret_cast<const f4v*>(sb + j * 36 + 8 * g + 4 * kg);     // read before the cell is overwritten below
                f4v dz;
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) dz[e4] = o[e4] * cfn_act_grad<ACT>(fmaf(xs[e4], cea, ceb)) * gmask;
                s1v = __builtin_elementwise_fma(dz, xs, s1v); s2v += dz;
                o = dz * cea;
            } else if (STATS) {
                const f4v om = o * gmask;
                s1v += om; s2v = __builtin_elementwise_fma(om, om, s2v);
            }
            *reinterpret_cast<f4v*>(sb + j * 36 + 8 * g + 4 * kg) = o;
        }
        wsync();
        if (STATS) { ssum += (s1v.x + s1v.y) + (s1v.z + s1v.w); qsum += (s2v.x + s2v.y) + (s2v.z + s2v.w); }
        pq0 = q0; last_buf = PARV;
    };
    bool have = false; int hq0 = 0;                                         // (owner) a tile waits for its partner's partial behind the next barrier
    auto step = [&](auto par_tag, float (&lc)[NST][8], float (&lc2)[TWO ? NST : 1][8], int tile) {
        constexpr int PAR = decltype(par_tag)::value;
        unsigned char* buf = Bs + PAR * 3 * IMG;
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int kb = wave + PWK_WAVES * u;
            if (kb < NKB) {                                                 // activate, split once, publish
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float4 c = sP[kb * 16 + kg * 8 + i];
                    if (MODE == PW_FWD) v[i] = cfn_act<ACT>(fmaf(lc[u][i], c.x, c.y));
                    else { v[i] = fmaf(lc[u][i], c.z, c.x); if (TWO) v[i] = fmaf(lc2[u][i], c.y, v[i]); }
                }
                u4k t[3];
                pwk_split8(v, t);
#pragma unroll
                for (int s = 0; s < 3; ++s) *reinterpret_cast<u4k*>(buf + s * IMG + j * PITCH + (kb * 16 + kg * 8) * 2) = t[s];
            }
        }
        issue(tile + 2 * tstep, lc, lc2);                                   // two tiles ahead, into the sets just consumed
        __syncthreads();
        if (!has_rows) return;
        constexpr int pbuf = PAR ^ 1;
        if (owner && have) finish(pbuf, hq0);                               // the previous tile: its partner's partial was written before this barrier
        if (EPI && owner) xe_issue(cfn_uni(tile * 32));                     // this tile's forward input travels during its MFMAs
        f4v dv[4];
        bf16x8k AA[2][3];                                                   // the operands of k-block kbl + 1 are read before the MFMAs of kbl issue
        const int kb0 = ks * NKS;
#pragma unroll
        for (int s = 0; s < 3; ++s) AA[0][s] = *reinterpret_cast<const bf16x8k*>(buf + s * IMG + rd_off + kb0 * 32);
#pragma unroll
        for (int kbl = 0; kbl < NKS; ++kbl) {
            bf16x8k (&A)[3] = AA[kbl & 1];
            if (kbl + 1 < NKS) {
#pragma unroll
                for (int s = 0; s < 3; ++s) AA[(kbl + 1) & 1][s] = *reinterpret_cast<const bf16x8k*>(buf + s * IMG + rd_off + (kb0 + kbl + 1) * 32);
            }
            __builtin_amdgcn_sched_barrier(0);
#define PWK_MM(SA, SW) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[SA], __builtin_bit_cast(bf16x8k, Wr[kbl][SW]), acc, 0, 0, 0)
            if (kbl == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], __builtin_bit_cast(bf16x8k, Wr[kbl][2]), (f16v)0.0f, 0, 0, 0);
            else PWK_MM(0, 2);
            PWK_MM(2, 0); PWK_MM(1, 1);
            if (owner) {                                                    // drain the previous tile between the MFMA groups
                if (2 * kbl < 4) dv[2 * kbl] = drain_read(pbuf, 2 * kbl);
                if (2 * kbl >= 1 && 2 * kbl - 1 < 4) drain_store(dv[2 * kbl - 1], 2 * kbl - 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            PWK_MM(0, 1); PWK_MM(1, 0); PWK_MM(0, 0);
#undef PWK_MM
            if (owner) {
                if (2 * kbl + 1 < 4) dv[2 * kbl + 1] = drain_read(pbuf, 2 * kbl + 1);
                if (2 * kbl < 4) drain_store(dv[2 * kbl], 2 * kbl);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (owner) { have = true; hq0 = cfn_uni(tile * 32); }
        else {                                                              // hand the partial over (picked up behind the next barrier)
            float* pw = red + PAR * (32 * 36);
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f4v*>(pw + j * 36 + 8 * g + 4 * kg) = (f4v){acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        }
    };
    int tile = cfn_uni(wg), last_par = 0;
    float ldA[NST][8], ldB[NST][8], ldA2[TWO ? NST : 1][8], ldB2[TWO ? NST : 1][8];
    issue(tile, ldA, ldA2);
    issue(tile + tstep, ldB, ldB2);
    for (;;) {
        if (tile >= ntiles) break;
        step(std::integral_constant<int, 0>{}, ldA, ldA2, tile);
        last_par = 0;
        tile += tstep;
        if (tile >= ntiles) break;
        step(std::integral_constant<int, 1>{}, ldB, ldB2, tile);
        last_par = 1;
        tile += tstep;
    }
    __syncthreads();                                                        // the last partials are in place
    if (has_rows && owner && have) finish(last_par, hq0);
    if (has_rows && owner && pq0 >= 0) {                                    // the last tile leaves now
#pragma unroll
        for (int sx = 0; sx < 4; ++sx) drain_store(drain_read(last_buf, sx), sx);
    }
    if (STATS && a.s1 && has_rows && owner) {
        ssum += __shfl_xor(ssum, 32, 64);
        qsum += __shfl_xor(qsum, 32, 64);
        if (kg == 0 && row < M) {
            cfn_add64(&a.s1[(long)n * Mfull + m0 + row], (double)ssum);
            cfn_add64(&a.s2[(long)n * Mfull + m0 + row], (double)qsum);
        }
    }
}

template <int NKB, int KS>
static int pwk_go(const PwArgs& a, int mode, bool stats, unsigned blocks, size_t lds, hipStream_t st) {
#define PWK_GO(...)                                                                                                         \
    do {                                                                                                                    \
        auto k = pwk_kernel<NKB, KS, __VA_ARGS__>;                                                                          \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * PWK_WAVES), lds, st, a);                                              \
    } while (0)
    if (mode == PW_DGRAD && stats) {
        if constexpr (KS == 1 && NKB <= 7) {
            switch (a.act) {
                case CFN_ACT_RELU: if (a.src2) PWK_GO(PW_DGRAD, CFN_ACT_RELU, true, true); else PWK_GO(PW_DGRAD, CFN_ACT_RELU, true, false); break;
                case CFN_ACT_SWISH: if (a.src2) PWK_GO(PW_DGRAD, CFN_ACT_SWISH, true, true); else PWK_GO(PW_DGRAD, CFN_ACT_SWISH, true, false); break;
                default: if (a.src2) PWK_GO(PW_DGRAD, CFN_ACT_NONE, true, true); else PWK_GO(PW_DGRAD, CFN_ACT_NONE, true, false); break;
            }
        }
    } else if (mode == PW_DGRAD) {
        if (a.src2) PWK_GO(PW_DGRAD, CFN_ACT_NONE, false, true); else PWK_GO(PW_DGRAD, CFN_ACT_NONE, false, false);
    } else if (stats) {
        switch (a.act) {
            case CFN_ACT_RELU: PWK_GO(PW_FWD, CFN_ACT_RELU, true, false); break;
            case CFN_ACT_SWISH: PWK_GO(PW_FWD, CFN_ACT_SWISH, true, false); break;
            default: PWK_GO(PW_FWD, CFN_ACT_NONE, true, false); break;
        }
    } else {
        switch (a.act) {
            case CFN_ACT_RELU: PWK_GO(PW_FWD, CFN_ACT_RELU, false, false); break;
            case CFN_ACT_SWISH: PWK_GO(PW_FWD, CFN_ACT_SWISH, false, false); break;
            default: PWK_GO(PW_FWD, CFN_ACT_NONE, false, false); break;
        }
    }
#undef PWK_GO
    return cfn_check_launch("pwconv(split bf16, register-resident weights)");
}

// returns -1 when the shape is not handled.  DGRAD: only without the act' epilogue (stats == false) and without the compact shortcut gradient
int pwk_try_launch(PwArgs& a, int mode, bool stats, hipStream_t st) {
    static const int on = getenv("CFN_PWK") ? atoi(getenv("CFN_PWK")) : 15;             // bit 0: forward (two k slices), bit 1: data gradient, bit 2: forward (one slice), bit 3: data gradient with act' epilogue
    if (pws_terms_now() != 6 || a.stem || a.stride != 1 || a.acc) return -1;
    if (mode == PW_DGRAD && stats && !(a.ea && a.ex && a.s1)) return -1;
    if (a.Q & 3) return -1;
    const int nkb = cfn_cdiv(a.K, 16);
    int KS;
    if (a.K > 128 && a.K <= 224 && a.M > 32 && a.M <= 128 && !(nkb & 1)) KS = 2;        // two equal slices
    else if (a.K >= 48 && a.K <= (a.M > 256 ? 192 : 112) && a.M > 128 && a.M <= 512 && (mode == PW_FWD || stats)) KS = 1;   // deep + many rows: two slabs
    else return -1;
    if (mode == PW_DGRAD ? !(on & (stats ? 8 : 2)) : !(on & (KS == 2 ? 1 : 4))) return -1;
    if (mode == PW_DGRAD && stats && (KS != 1 || a.M > 256)) return -1;     // act' epilogue: one-slice shapes without slabs (192 -> 432 rows: 256 VGPRs + 63 spilled
                                                                            // dwords, 0.256 vs 0.260 ms for pw_deep_kernel: not instantiated)
    if (mode == PW_DGRAD && stats && (((uintptr_t)a.ex) & 15)) return -1;
    if (a.act != CFN_ACT_NONE && a.act != CFN_ACT_RELU && a.act != CFN_ACT_SWISH) return -1;
    if ((long)a.K * a.Q * 4 >= 0x3ffffff0L || (long)a.M * a.Q * 4 >= 0x3ffffff0L) return -1;
    if (((uintptr_t)a.src | (uintptr_t)a.dst | (uintptr_t)(a.src2 ? a.src2 : a.src)) & 15) return -1;
    const int KP = 16 * nkb, nrt = cfn_cdiv(a.M, 32), NKS = (nkb + KS - 1) / KS;
    size_t lds = (size_t)2 * 3 * 32 * (KP * 2 + 16) + (size_t)KP * 16 + (size_t)KS * (nrt > 8 ? cfn_cdiv(nrt, cfn_cdiv(nrt, 8)) : nrt) * 2 * 32 * 36 * 4;
    const size_t wtmp = (size_t)PWK_WAVES * (mode == PW_FWD ? 32 * ((NKS > 6 ? 6 : NKS) * 16 + 4) : NKS * 16 * 36) * 4;
    if (wtmp > lds) lds = wtmp;
    if (lds > 160 * 1024) return -1;
    PwArgs b = a;
    const int slabs = KS == 1 ? cfn_cdiv(nrt, 8) : 1, rts = cfn_cdiv(nrt, slabs);       // row tiles per slab (<= 8)
    b.mtiles = slabs; b.kres = KS == 1 ? 32 * rts : a.M;
    const int ntiles = cfn_cdiv(a.Q, 32);
    static const int wg_env = getenv("CFN_PWK_WGS") ? atoi(getenv("CFN_PWK_WGS")) : 0;
    long wgs = cfn_cdiv(wg_env > 0 ? wg_env : 256, (long)a.N * slabs);      // one workgroup per CU (128 / 192 / 256 / 384 / 512: 0.30 / 0.22 / 0.176 / 0.23 / 0.185 ms)
    if (wgs > ntiles) wgs = ntiles;
    if (wgs < 1) wgs = 1;
    b.nstrips = (int)wgs;
    const unsigned blocks = (unsigned)((long)a.N * wgs * slabs);
    if (KS == 2) {
        switch (nkb) {
            case 10: return pwk_go<10, 2>(b, mode, stats, blocks, lds, st);
            case 12: return pwk_go<12, 2>(b, mode, stats, blocks, lds, st);
            case 14: return pwk_go<14, 2>(b, mode, stats, blocks, lds, st);
            default: return -1;
        }
    }
    switch (nkb) {
        case 3: return pwk_go<3, 1>(b, mode, stats, blocks, lds, st);
        case 4: return pwk_go<4, 1>(b, mode, stats, blocks, lds, st);
        case 5: return pwk_go<5, 1>(b, mode, stats, blocks, lds, st);
        case 6: return pwk_go<6, 1>(b, mode, stats, blocks, lds, st);
        case 7: return pwk_go<7, 1>(b, mode, stats, blocks, lds, st);
        case 12: return pwk_go<12, 1>(b, mode, stats, blocks, lds, st);
        default: return -1;
    }
}
