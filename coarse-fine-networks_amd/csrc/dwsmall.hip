// Depthwise 3x3x3 forward on SMALL planes (14x14 and 7x7: X3D layers 3 and 4, stride 1) -- one WAVE per (sample, channel,
// t-chunk), no workgroup barrier.
//
// Why a second kernel: on these planes the band / t-chunk kernel of dwconv3d.hip packs 9 (14x14) or 73 (7x7) channels into a
// workgroup with per-lane channels, i.e. 27 weights + 21 accumulators + two prefetched frames in VGPRs (248 registers, two
// waves per SIMD) and one workgroup barrier per frame; the SQ counters show those launches 45 % parked with the VALU 36-46 %
// busy (profiles/r02_pmc_dw_valu.json): latency bound by occupancy.  Here the channel is wave uniform (27 weights and the
// prologue coefficients in SGPRs), a lane owns HS vertically adjacent outputs of one column (14x14: 14 columns x 4 groups of
// 4 rows = 56 lanes; 7x7: 49 lanes, one output each), the frame of the wave's channel is ONE coalesced load per lane (a
// float4 / a float), staged through a wave-private LDS image with zero halo, and three rolling accumulator sets carry the
// temporal taps.  43-91 VGPRs => 5-7 waves per SIMD, each an independent stream with a group of G frames in flight; waves
// never wait for each other.  Measured (8 clips, T=256): 7x7 142 -> 96 us (2.4 -> 3.6 TB/s), 14x14 203 -> 180 us (3.4 -> 3.85).
// What was learned on the way (ablations with the loads / stores / FMAs switched off): predicated loads make hipcc wait
// with vmcnt(0) (one HBM round trip per frame); a block index the compiler cannot prove wave uniform puts the buffer
// descriptors in VGPRs and wraps every buffer instruction in a waterfall loop; on 7x7 the per-frame bookkeeping (~90
// instructions against 27 FMAs per lane) was the limit, hence the frame groups; 14x14 is ~70 % VALU bound (lane waste of
// the 14-wide rows, 4.5 LDS reads per output).  LDS accesses of one wave execute in order, so the staged frame is visible to the wave's later reads
// without a barrier (a wave-level fence keeps the compiler from reordering them).
// fp32 or bf16 tensors (cp_io.h: compiled a second time through dwsmall_bf16.hip; LDS image, accumulators and statistics stay fp32).
#include "cp_io.h"
#include <stdlib.h>

#ifdef DW_BF16
#define DwSmallArgs H16N(DwSmallArgs)
#endif
struct DwSmallArgs {
    const cpe_t* x; const double* A; const double* B; const float* w; cpe_t* y; double* s1; double* s2;
    int N, C, T, act, TT, nchunks;
    long total_waves;
};

template <int PH, int HS>
__global__ __launch_bounds__(256, 4) void dw3d_small_fwd_kernel(const DwSmallArgs a) {
    unsigned long long* const det_keys = cfn_det_keys();
    constexpr int RG = (PH + HS - 1) / HS;            // row groups per column
    constexpr int ROWS = RG * HS + 2;                 // image rows incl. halo (and the unused rows of a ragged last group)
    constexpr int PIT = PH == 14 ? 20 : 9;            // row pitch in floats
    constexpr int XO = PH == 14 ? 2 : 1;              // column of plane column 0 (14x14: even, 8-byte aligned float2 writes)
    constexpr int IMG = ROWS * PIT;
    constexpr int P = PH * PH;
    constexpr int LV = PH == 14 ? 4 : 1;              // elements per loader lane
    constexpr int NLD = P / LV;                       // loader lanes (49)
    constexpr int DEPTH = PH == 14 ? 4 : 8;           // frames per fetch group (one float4 / one float per lane each); even
    constexpr int OOB = 0x7ffffff0;
    static_assert(P % LV == 0 && NLD <= 64, "plane must fit one load per lane");
    __shared__ __attribute__((aligned(16))) float smem[4 * 2 * IMG];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    // wave-uniform by construction; readfirstlane tells the compiler (otherwise the buffer descriptors built from it live in
    // VGPRs and every buffer instruction is wrapped in a waterfall loop).  Chunks of one channel are adjacent: halo frames
    // come from L2
    const long widx = __builtin_amdgcn_readfirstlane((int)(L * 4 + wv));
    if (widx >= a.total_waves) return;                // whole waves only: no barrier anywhere below
    // (the 64-bit divisions are expanded into vector code: state the uniformity of their results as well)
    const int chunk = cfn_uni((int)(widx % a.nchunks));
    const long nc = cfn_uni((long)(widx / a.nchunks));
    const int c = cfn_uni((int)(nc % a.C));
    const int T = a.T, t0 = chunk * a.TT, t1 = min(t0 + a.TT, T);
    float* img = smem + wv * 2 * IMG;

    // wave-uniform weights / coefficients -> SGPRs
    float wr[27];
#pragma unroll
    for (int j = 0; j < 27; ++j) wr[j] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.w[(long)c * 27 + j])));
    const float pa = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.A ? (float)a.A[nc] : 1.0f)));
    const float pb = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.A ? (float)a.B[nc] : 0.0f)));
    const float act_lo = a.act == CFN_ACT_RELU ? 0.0f : -__builtin_inff();      // none / ReLU only (the planner checks): branch free

    for (int i = lane; i < 2 * IMG; i += 64) img[i] = 0.0f;       // halo (and everything else) zero; wave-private

    // loader lane -> LDS offsets of its elements (14x14: two float2 halves, neither crosses a row: even width, even start)
    const bool ld_on = lane < NLD;
    const int e0 = lane * LV;
    const int lo0 = ld_on ? ((e0 / PH) + 1) * PIT + XO + (e0 % PH) : 0;
    const int lo1 = (ld_on && LV == 4) ? (((e0 + 2) / PH) + 1) * PIT + XO + ((e0 + 2) % PH) : 0;
    // compute lane: column cc, row group g
    const int g = lane / PH, cc = lane - g * PH;
    const bool act_lane = g < RG;
    const int row0 = g * HS;
    const float* tb = img + (act_lane ? row0 * PIT + (XO - 1) + cc : 0);

    typedef float __attribute__((ext_vector_type(4))) f4;
    typedef float __attribute__((ext_vector_type(2))) f2;
    // Every global access of the frame loop is an UNCONDITIONAL buffer load / store (an unwanted access gets an out-of-range
    // offset: loads return 0, stores are dropped): with no vector-memory instruction under a branch the compiler counts
    // them and waits with vmcnt(N) for exactly the frame it needs, so the DEPTH prefetched frames really stay in flight
    // (with predicated loads it waits with vmcnt(0) and every frame costs a full HBM round trip).
    __amdgpu_buffer_rsrc_t rx = cfn_rsrc(const_cast<cpe_t*>(a.x + nc * (long)T * P), (unsigned)((long)T * P * CP_ES));
    __amdgpu_buffer_rsrc_t ry = cfn_rsrc(a.y + nc * (long)T * P, (unsigned)((long)T * P * CP_ES));
    const int ldo = ld_on ? e0 * CP_ES : OOB;
    auto fetch = [&](int f) -> f4 {
        const bool want = f >= 0 && f < T && f <= t1;
        const int vo = want ? ldo : OOB, so = cfn_uni(want ? f * P * CP_ES : 0);
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (LV == 4) v = cp_ld4(rx, vo, so);
        else v.x = cp_ld1(rx, vo, so);
        return v;
    };
    auto stage = [&](int f, f4 v, float* im) {       // frames outside the clip are zero AFTER the prologue
        const bool fv = f >= 0 && f < T;
        if (!ld_on) return;
        const float m = fv ? 1.0f : 0.0f;             // frames outside the clip: zero AFTER the prologue
        v.x = fmaxf(fmaf(v.x, pa, pb), act_lo) * m;
        if (LV == 4) { v.y = fmaxf(fmaf(v.y, pa, pb), act_lo) * m; v.z = fmaxf(fmaf(v.z, pa, pb), act_lo) * m; v.w = fmaxf(fmaf(v.w, pa, pb), act_lo) * m; }
        if (LV == 4) {
            *reinterpret_cast<f2*>(im + lo0) = (f2){v.x, v.y};
            *reinterpret_cast<f2*>(im + lo1) = (f2){v.z, v.w};
        } else {
            im[lo0] = v.x;
        }
    };
    auto wave_sync = [&]() {                          // LDS ops of a wave run in order; only the compiler has to be told
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    // Accumulators as explicit register PAIRS so that the taps run on v_pk_fma_f32 (two FMAs per issue slot; hipcc leaves
    // this loop scalar): a01[i] = (out(f+1), out(f)) of row i share the input value and take the weight pair
    // (w[kt=0][tap], w[kt=1][tap]); the kt=2 sums of two ADJACENT rows share the input rows in the middle and take the pair
    // (w[2][kh][kw], w[2][kh-1][kw]).  108 -> 60 FMA instructions per 4-row strip and frame.
    typedef float __attribute__((ext_vector_type(2))) p2;
    constexpr int HP = (HS + 1) / 2;                  // row pairs
    p2 a01[HS], a2[HP];
#pragma unroll
    for (int i = 0; i < HS; ++i) a01[i] = (p2){0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < HP; ++i) a2[i] = (p2){0.0f, 0.0f};
    p2 w01[9], w2p[6];                                // wave uniform (SGPR pairs)
#pragma unroll
    for (int j = 0; j < 9; ++j) w01[j] = (p2){wr[j], wr[9 + j]};
#pragma unroll
    for (int j = 0; j < 6; ++j) w2p[j] = (p2){wr[18 + 3 + j], wr[18 + j]};      // (kh = 1 + j/3, kh - 1) at kw = j % 3
    float st1 = 0.0f, st2 = 0.0f;
    const int yo = act_lane ? (row0 * PH + cc) * CP_ES : OOB;

    // input frames t0-1 .. t1; frame f finishes output frame f-1.  Frames are fetched in GROUPS of G consecutive frames of the
    // channel (G back-to-back loads of one contiguous G*P*4-byte run: the DRAM page is still open for the next one) one
    // group ahead of their use, and the per-frame bookkeeping (queue rotation, loop control) is paid once per group.
    const int f_first = t0 - 1, f_last = t1;
    auto one_step = [&](int f, int par, f4 next_frame) {
        // stage frame f+1 into the other image, compute frame f from this one, emit output frame f-1
        stage(f + 1, next_frame, img + (par ^ 1) * IMG);
        wave_sync();
        const float* tp = tb + par * IMG;
        {   // frames outside the clip were staged as zeros: no branch needed
#pragma unroll
            for (int r = 0; r < HS + 2; ++r) {
                const float v0 = tp[r * PIT], v1 = tp[r * PIT + 1], v2 = tp[r * PIT + 2];
                const p2 b0 = {v0, v0}, b1 = {v1, v1}, b2 = {v2, v2};
#pragma unroll
                for (int i = 0; i < HS; ++i) {
                    const int kh = r - i;
                    if (kh >= 0 && kh < 3)
                        a01[i] = __builtin_elementwise_fma(w01[kh * 3 + 0], b0, __builtin_elementwise_fma(w01[kh * 3 + 1], b1, __builtin_elementwise_fma(w01[kh * 3 + 2], b2, a01[i])));
                }
#pragma unroll
                for (int ip = 0; ip < HP; ++ip) {      // rows (2 ip, 2 ip + 1): kh = r - 2 ip for the first, kh - 1 for the second
                    const int kh = r - 2 * ip;
                    const bool has2 = 2 * ip + 1 < HS;
                    if (kh == 0 || (kh == 1 && !has2) || (kh == 2 && !has2)) {
                        a2[ip].x = fmaf(wr[18 + kh * 3 + 0], v0, fmaf(wr[18 + kh * 3 + 1], v1, fmaf(wr[18 + kh * 3 + 2], v2, a2[ip].x)));
                    } else if (kh == 1 || kh == 2) {
                        a2[ip] = __builtin_elementwise_fma(w2p[(kh - 1) * 3 + 0], b0, __builtin_elementwise_fma(w2p[(kh - 1) * 3 + 1], b1, __builtin_elementwise_fma(w2p[(kh - 1) * 3 + 2], b2, a2[ip])));
                    } else if (kh == 3 && has2) {
                        a2[ip].y = fmaf(wr[18 + 6 + 0], v0, fmaf(wr[18 + 6 + 1], v1, fmaf(wr[18 + 6 + 2], v2, a2[ip].y)));
                    }
                }
            }
        }
        const int to = f - 1;
        const bool emit = to >= t0 && to < t1 && f <= f_last;
        const int so = cfn_uni(emit ? to * P * CP_ES : 0);
#pragma unroll
        for (int i = 0; i < HS; ++i) {
            const bool ok = emit && act_lane && row0 + i < PH;
            const float v = cp_rt1(ok ? ((i & 1) ? a2[i >> 1].y : a2[i >> 1].x) : 0.0f);      // (bf16: statistics over the stored values)
            cp_st1(v, ry, ok ? yo + i * PH * CP_ES : OOB, so);
            st1 += v;
            st2 = fmaf(v, v, st2);
        }
#pragma unroll
        for (int i = 0; i < HS; ++i) {
            if (i & 1) a2[i >> 1].y = a01[i].y; else a2[i >> 1].x = a01[i].y;
            a01[i] = (p2){0.0f, a01[i].x};
        }
    };
    constexpr int G = DEPTH;
    f4 cur[G], nxt[G];
    {
        const f4 first = fetch(f_first);
#pragma unroll
        for (int j = 0; j < G; ++j) cur[j] = fetch(f_first + 1 + j);       // frames staged during the first group
        wave_sync();
        stage(f_first, first, img);
    }
    for (int f0 = f_first; f0 <= f_last; f0 += 2 * G) {                    // two groups per trip: the register sets swap roles
#pragma unroll
        for (int j = 0; j < G; ++j) nxt[j] = fetch(f0 + G + 1 + j);
#pragma unroll
        for (int j = 0; j < G; ++j) one_step(f0 + j, j & 1, cur[j]);
        if (f0 + G > f_last) break;
#pragma unroll
        for (int j = 0; j < G; ++j) cur[j] = fetch(f0 + 2 * G + 1 + j);
#pragma unroll
        for (int j = 0; j < G; ++j) one_step(f0 + G + j, j & 1, nxt[j]);
    }
    if (a.s1) {
        st1 = cfn_wave_sum(st1);
        st2 = cfn_wave_sum(st2);
        if (lane == 0) { cfn_add64(&a.s1[nc], (double)st1, det_keys); cfn_add64(&a.s2[nc], (double)st2, det_keys); }
    }
}

// returns -1 when the shape is not handled (caller uses the band kernel); probe: 0 = handled, nothing launched; otherwise the
// launch status
int CPN(dw_small_fwd_try)(const cpe_t* x, const double* A, const double* B, int act, const float* w, cpe_t* y, double* sum, double* sumsq,
                     int N, int C, int T, int Hi, int Wi, int stride, hipStream_t st, bool probe) {
    static const int enabled = getenv("CFN_DW_SMALL") ? atoi(getenv("CFN_DW_SMALL")) : 1;
    if (!enabled || stride != 1 || Hi != Wi || (Hi != 14 && Hi != 7)) return -1;
    if (act != CFN_ACT_NONE && act != CFN_ACT_RELU && A != nullptr) return -1;      // branch-free prologue: none / ReLU (every X3D conv2)
    if ((long)T * Hi * Wi * CP_ES >= 0x7ffffff0L) return -1;
    if (Hi == 14 && (((uintptr_t)x & (4 * CP_ES - 1)) != 0)) return -1;
    if (probe) return 0;
    DwSmallArgs a = {x, A, B, w, y, sum, sumsq, N, C, T, act, 0, 0, 0};
    // t-chunks: ~2 rounds of the chip at 24 resident waves per CU (6 per SIMD), at least 8 frames per chunk (halo re-reads)
    const long planes = (long)N * C;
    long nch = (2L * 256 * 24 + planes - 1) / planes;
    if (nch < 1) nch = 1;
    int TT = (int)((T + nch - 1) / nch);
    if (TT < 8) TT = 8;
    if (TT > T) TT = T;
    a.TT = TT;
    a.nchunks = (T + TT - 1) / TT;
    a.total_waves = planes * a.nchunks;
    const unsigned blocks = (unsigned)((a.total_waves + 3) / 4);
    if (Hi == 14) hipLaunchKernelGGL((dw3d_small_fwd_kernel<14, 4>), dim3(blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((dw3d_small_fwd_kernel<7, 1>), dim3(blocks), dim3(256), 0, st, a);
    return cfn_check_launch("dwconv3d small-plane forward");
}
