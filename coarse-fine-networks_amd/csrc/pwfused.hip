// Fused backward of a stride-1 pointwise (1x1x1) convolution with few channels (Cin, Cout <= 64: X3D layer 1,
// x3d_fine.py:100-105 conv1 / conv3): data gradient AND weight gradient in one pass.
//
// These layers are HBM bound (8-19 flop/B).  Run as two kernels the backward moves 7 tensor passes
// (dgrad: gy, y, x -> gx; wgrad: gy, y, x); here gy, y, x leave HBM once and gx is written once: 4 passes.
//   G' = gsc*gy + gs + 2*y*gq                      (the gradient of the raw conv output, formed on load)
//   gW[co][ci] += sum_q G'[co][q] * act(A x + B)[ci][q]
//   da[ci][q]   = sum_co W[co][ci] * G'[co][q]  (+ acc on its lattice);  gx = da * act'(A x + B) * A
//   gA[n,ci] += sum_q dz*x,  gB[n,ci] += sum_q dz           (dz = da * act')
// A workgroup (4 waves) owns one sample and a strip of positions and walks it in stages of 64 positions: G' rows and
// the RAW x rows are staged in LDS as [channel][65] (float4 global loads one stage ahead in registers, double-buffered
// images, one barrier per stage).  Wave w owns positions 16w .. 16w+15 of a stage for BOTH products:
//   weight gradient: v_mfma_f32_32x32x2 with lane <-> channel row, k <-> position pair (conflict-free LDS reads, odd
//                    pitch); the prologue act(A x + B) is applied to the x operand as it is read (each element once);
//   data gradient:   v_mfma_f32_16x16x4 with the W^T operand resident in registers, B operand = G' rows of the wave's 16
//                    positions; the C layout (lane <-> position, 4 channel rows) feeds the act' epilogue (raw x from LDS),
//                    the statistics partials (kept per lane, reduced once at the end) and 64-byte row-segment stores.
#include "cfn_common.h"
#include <stdlib.h>

#include "pw_common.h"

typedef float __attribute__((ext_vector_type(4))) pf4;

#define PF_PT 64
#define PF_PITCH 65

struct PfArgs {
    const float* gy; const float* y; const double* gs; const double* gq; const double* gsc;
    const float* w;                       // (Cout, Cin) row major
    const float* x; const double* pa; const double* pb;
    float* gx; double* gA; double* gB; double* gw;
    const float* acc; int acc_s, acc_Ho, acc_Wo, Hi, Wi, T;
    int N, M, K, Q, nstrips, stages;      // M = Cout (rows of G'), K = Cin (rows of x)
};

// ACT < 0: the forward conv had no prologue (gx = da, no statistics)
template <int MTW, int NTW, int ACT>
__global__ __launch_bounds__(256, ACT < 0 ? 3 : 2) void pw_bwd_fused_kernel(const PfArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr bool EPI = ACT >= 0;
    constexpr int ACTV = EPI ? ACT : CFN_ACT_NONE;
    constexpr int BM = 32 * MTW, BN = 32 * NTW;
    constexpr int NG = BM / 16, NX = BN / 16;      // float4 per thread per stage (G rows / X rows)
    constexpr int NT16 = BN / 16, KS = BM / 4;      // data gradient: 16-row channel tiles, k-steps of 4 output channels
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, col = lane & 31;
    const int m16 = lane & 15, kq = lane >> 4;
    unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    const int strip = L % a.nstrips;
    const int n = L / a.nstrips;
    const int M = a.M, K = a.K, Q = a.Q;

    constexpr int IMG = (BM + BN) * PF_PITCH;          // one staged image: G' rows [BM][65] then raw x rows [BN][65]
    float* img0 = smem;                                // two images (double buffer)
    float* sCg = smem + 2 * IMG;                       // [BM][2]  (gs, 2gq)
    float* sCz = sCg + 2 * BM;                         // [BM]     gsc
    float* sCx = sCz + BM;                             // [BN][2]  (A, B)
    float* sSt = sCx + 2 * BN;                         // [4 waves][BN][2]  statistics of this workgroup
    for (int m = tid; m < BM; m += 256) {
        const bool ok = m < M;
        sCg[2 * m] = (ok && a.gs) ? (float)a.gs[(long)n * M + m] : 0.0f;
        sCg[2 * m + 1] = (ok && a.gq && a.y) ? 2.0f * (float)a.gq[(long)n * M + m] : 0.0f;
        sCz[m] = (ok && a.gsc) ? (float)a.gsc[(long)n * M + m] : 1.0f;
    }
    for (int k = tid; k < BN; k += 256) {
        const bool ok = EPI && k < K;
        sCx[2 * k] = ok ? (float)a.pa[(long)n * K + k] : 1.0f;
        sCx[2 * k + 1] = ok ? (float)a.pb[(long)n * K + k] : 0.0f;
    }
    // W^T operand of the data gradient, resident: lane (ci = t*16 + m16, co = 4s + kq)
    float wq[NT16][KS];
#pragma unroll
    for (int t = 0; t < NT16; ++t)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int ci = t * 16 + m16, co = 4 * s + kq;
            wq[t][s] = (ci < K && co < M) ? a.w[(long)co * K + ci] : 0.0f;
        }
    // prologue coefficients of this lane's x rows in the weight-gradient operand (row j*32 + col)
    float ca[NTW], cb[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int k = j * 32 + col;
        const bool ok = EPI && k < K;
        ca[j] = ok ? (float)a.pa[(long)n * K + k] : 1.0f;
        cb[j] = ok ? (float)a.pb[(long)n * K + k] : 0.0f;
    }
    f16v acc[MTW][NTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    float sa[NT16][4], sb[NT16][4];                    // per-lane partial statistics of rows t*16 + 4*kq + r
#pragma unroll
    for (int t = 0; t < NT16; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { sa[t][r] = 0.0f; sb[t][r] = 0.0f; }
    __syncthreads();

    const int lrow = tid >> 4, c4 = (tid & 15) * 4;          // 16 lanes cover one 64-position row segment
    // Every global access of the stage loop is an UNCONDITIONAL buffer load / store (unwanted ones get an out-of-range
    // offset: loads return 0, stores are dropped): the compiler can then count them and wait with vmcnt(N) for the
    // prefetched rows only, instead of vmcnt(0), which would also wait for the previous stage's gx stores to retire.
    constexpr int OOB = 0x7ffffff0;
    __amdgpu_buffer_rsrc_t rg = cfn_rsrc(const_cast<float*>(a.gy + (long)n * M * Q), (unsigned)((long)M * Q * 4));
    __amdgpu_buffer_rsrc_t ry = cfn_rsrc(const_cast<float*>((a.y ? a.y : a.gy) + (long)n * M * Q), a.y ? (unsigned)((long)M * Q * 4) : 0u);
    __amdgpu_buffer_rsrc_t rx = cfn_rsrc(const_cast<float*>(a.x + (long)n * K * Q), (unsigned)((long)K * Q * 4));
    __amdgpu_buffer_rsrc_t rd = cfn_rsrc(a.gx + (long)n * K * Q, (unsigned)((long)K * Q * 4));
    pf4 pg[NG], py[NG], px[NX];
    int vog[NG], vox[NX];                                    // byte offsets of this thread's row segments (position 0)
#pragma unroll
    for (int it = 0; it < NG; ++it) vog[it] = (it * 16 + lrow) < M ? ((it * 16 + lrow) * Q + c4) * 4 : OOB;
#pragma unroll
    for (int it = 0; it < NX; ++it) vox[it] = (it * 16 + lrow) < K ? ((it * 16 + lrow) * Q + c4) * 4 : OOB;
    auto prefetch = [&](int q0) {
        const bool inq = q0 + c4 < Q;
#pragma unroll
        for (int it = 0; it < NG; ++it) {
            const int vo = inq ? vog[it] : OOB;
            pg[it] = __builtin_bit_cast(pf4, __builtin_amdgcn_raw_buffer_load_b128(rg, vo, q0 * 4, 0));
            py[it] = __builtin_bit_cast(pf4, __builtin_amdgcn_raw_buffer_load_b128(ry, vo, q0 * 4, 0));
        }
#pragma unroll
        for (int it = 0; it < NX; ++it)
            px[it] = __builtin_bit_cast(pf4, __builtin_amdgcn_raw_buffer_load_b128(rx, inq ? vox[it] : OOB, q0 * 4, 0));
    };
    auto stage = [&](int q0, float* sG, float* sX) {
        const bool inq = q0 + c4 < Q;
#pragma unroll
        for (int it = 0; it < NG; ++it) {
            const int row = it * 16 + lrow;
            const float cs = cfn_settle(sCg[2 * row]), cq = cfn_settle(sCg[2 * row + 1]), cz = cfn_settle(sCz[row]);      // (an LDS pair in front of packed FMAs: DESIGN 4.1; found by the per-half scan)
            const bool ok = inq && (row < M);
            float* d = sG + row * PF_PITCH + c4;
            d[0] = ok ? fmaf(py[it].x, cq, fmaf(pg[it].x, cz, cs)) : 0.0f;
            d[1] = ok ? fmaf(py[it].y, cq, fmaf(pg[it].y, cz, cs)) : 0.0f;
            d[2] = ok ? fmaf(py[it].z, cq, fmaf(pg[it].z, cz, cs)) : 0.0f;
            d[3] = ok ? fmaf(py[it].w, cq, fmaf(pg[it].w, cz, cs)) : 0.0f;
        }
#pragma unroll
        for (int it = 0; it < NX; ++it) {
            const int row = it * 16 + lrow;
            float* d = sX + row * PF_PITCH + c4;             // raw x; rows >= K and positions >= Q were loaded as 0
            d[0] = px[it].x; d[1] = px[it].y; d[2] = px[it].z; d[3] = px[it].w;
        }
    };

    const int qbeg = strip * a.stages * PF_PT;
    const int nst = min(a.stages, (Q - qbeg + PF_PT - 1) / PF_PT);
    if (nst > 0) {
        prefetch(qbeg);
        stage(qbeg, img0, img0 + BM * PF_PITCH);
        if (nst > 1) prefetch(qbeg + PF_PT);
    }
    __syncthreads();
    const int pbase = wave * (PF_PT / 4);
    const int hw = a.Hi * a.Wi;
    const int acc_pitch4 = a.T * a.acc_Ho * a.acc_Wo * 4;   // bytes per channel of the compact gradient
    __amdgpu_buffer_rsrc_t racc = cfn_rsrc(const_cast<float*>(a.acc ? a.acc + (long)n * K * (acc_pitch4 / 4) : a.gy), a.acc ? (unsigned)((long)K * acc_pitch4) : 0u);
    for (int st = 0; st < nst; ++st) {
        const int q0 = qbeg + st * PF_PT;
        float* cur = img0 + (st & 1) * IMG;
        float* nxt = img0 + ((st + 1) & 1) * IMG;
        if (st + 1 < nst) {
            stage(q0 + PF_PT, nxt, nxt + BM * PF_PITCH);
            if (st + 2 < nst) prefetch(q0 + 2 * PF_PT);
        }
        const float* sG = cur;
        const float* sX = cur + BM * PF_PITCH;
        // ---- weight gradient: this wave's 16 positions ----------------------------------------------------
#pragma unroll
        for (int s = 0; s < PF_PT / 8; ++s) {
            const int p = pbase + 2 * s + half;
            float av[MTW], bv[NTW];
#pragma unroll
            for (int i = 0; i < MTW; ++i) av[i] = sG[(i * 32 + col) * PF_PITCH + p];
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const float xr = sX[(j * 32 + col) * PF_PITCH + p];
                bv[j] = EPI ? cfn_act<ACTV>(fmaf(xr, ca[j], cb[j])) : xr;
            }
#pragma unroll
            for (int i = 0; i < MTW; ++i)
#pragma unroll
                for (int j = 0; j < NTW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        // ---- data gradient of the same 16 positions ---------------------------------------------------------
        pf4 da[NT16];
#pragma unroll
        for (int t = 0; t < NT16; ++t) da[t] = (pf4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float b = sG[(4 * s + kq) * PF_PITCH + pbase + m16];
#pragma unroll
            for (int t = 0; t < NT16; ++t) da[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[t][s], b, da[t], 0, 0, 0);
        }
        const int q = q0 + pbase + m16;
        const bool qv = q < Q;
        int aoff = OOB;                                        // compact lattice byte offset of this lane's position
        if (a.acc && qv) {
            const int tq = q / hw, rq = q - tq * hw;
            const int hq = rq / a.Wi, wq_ = rq - hq * a.Wi;
            if (hq % a.acc_s == 0 && wq_ % a.acc_s == 0) aoff = (((tq * a.acc_Ho + hq / a.acc_s) * a.acc_Wo + wq_ / a.acc_s)) * 4;
        }
        float av4[NT16][4];
#pragma unroll
        for (int t = 0; t < NT16; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) av4[t][r] = 0.0f;
        if (a.acc) {                                           // workgroup uniform
            // per-lane part of the row (4*kq + r) goes into the VECTOR offset: a lane-dependent scalar offset would be
            // serialised by a waterfall loop
            const unsigned arow = (unsigned)aoff + (unsigned)(4 * kq) * (unsigned)acc_pitch4;
#pragma unroll
            for (int t = 0; t < NT16; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    av4[t][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        racc, (int)(aoff == OOB ? (unsigned)OOB : arow + (unsigned)r * (unsigned)acc_pitch4), t * 16 * acc_pitch4, 0));
        }
#pragma unroll
        for (int t = 0; t < NT16; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = t * 16 + 4 * kq + r;
                const bool ok = qv && ci < K;
                float v = da[t][r] + av4[t][r];
                if (EPI) {
                    const float xr = sX[ci * PF_PITCH + pbase + m16];
                    const float pa = sCx[2 * ci], pb = sCx[2 * ci + 1];
                    const float dz = ok ? v * cfn_act_grad<ACTV>(fmaf(xr, pa, pb)) : 0.0f;
                    sa[t][r] = fmaf(dz, xr, sa[t][r]);
                    sb[t][r] += dz;
                    v = dz * pa;
                }
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rd, ok ? (ci * Q + pbase + m16) * 4 : OOB, q0 * 4, 0);
            }
        }
        __syncthreads();
    }

    // ---- statistics: reduce over the 16 position lanes, then over the 4 waves in LDS, one fp64 atomic per row -------
    if (EPI) {
#pragma unroll
        for (int t = 0; t < NT16; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float u = sa[t][r], v = sb[t][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { u += __shfl_xor(u, o, 64); v += __shfl_xor(v, o, 64); }
                if (m16 == 0) {                                // per-wave slot, plain store (fixed summation order below)
                    const int ci = t * 16 + 4 * kq + r;
                    sSt[(wave * BN + ci) * 2] = u;
                    sSt[(wave * BN + ci) * 2 + 1] = v;
                }
            }
    }
    // ---- weight gradient: combine the 4 waves through LDS, one fp64 atomic per element per workgroup ---------
    constexpr int CWP = BN + 1;
    float* cw = smem + wave * (BM * CWP);      // [4][BM][BN+1] over the (idle) images
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                cw[ml * CWP + j * 32 + col] = acc[i][j][r];
            }
    __syncthreads();
    for (int e = tid; e < BM * BN; e += 256) {
        const int ml = e / BN, kl = e - ml * BN;
        const int o = ml * CWP + kl;
        const float v = (smem[o] + smem[BM * CWP + o]) + (smem[2 * BM * CWP + o] + smem[3 * BM * CWP + o]);
        if (ml < M && kl < K) cfn_add64(&a.gw[(long)ml * K + kl], (double)v);
    }
    if (EPI && tid < K) {
        const float u = (sSt[2 * tid] + sSt[(BN + tid) * 2]) + (sSt[(2 * BN + tid) * 2] + sSt[(3 * BN + tid) * 2]);
        const float v = (sSt[2 * tid + 1] + sSt[(BN + tid) * 2 + 1]) + (sSt[(2 * BN + tid) * 2 + 1] + sSt[(3 * BN + tid) * 2 + 1]);
        cfn_add64(&a.gA[(long)n * K + tid], (double)u);
        cfn_add64(&a.gB[(long)n * K + tid], (double)v);
    }
}

template <int MTW, int NTW>
static int pf_launch(const PfArgs& a, int act, bool epi, unsigned blocks, hipStream_t st) {
    constexpr int BM = 32 * MTW, BN = 32 * NTW;
    size_t lds = ((size_t)2 * (BM + BN) * PF_PITCH + 3 * BM + 10 * BN) * sizeof(float);
    const size_t lds_cw = (size_t)4 * BM * (BN + 1) * sizeof(float);
    if (lds_cw > lds) lds = lds_cw;
#define CFN_PF_GO(ACTV)                                                                                         \
    do {                                                                                                        \
        auto k = pw_bwd_fused_kernel<MTW, NTW, ACTV>;                                                           \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, st, a);                                             \
    } while (0)
    if (!epi) CFN_PF_GO(-1);
    else if (act == CFN_ACT_RELU) CFN_PF_GO(CFN_ACT_RELU);
    else if (act == CFN_ACT_SWISH) CFN_PF_GO(CFN_ACT_SWISH);
    else CFN_PF_GO(CFN_ACT_NONE);
#undef CFN_PF_GO
    return cfn_check_launch("pwconv_bwd_fused");
}

extern "C" int cfn_pwconv_bwd_fused(const float* gy, const float* y, const double* gsum, const double* gsumsq, const float* w,
                                    const float* x, const double* A, const double* B, int act, float* gx, double* gA, double* gB,
                                    double* gw, int N, int Cin, int Cout, int T, int Hi, int Wi, const float* acc, int acc_stride,
                                    const double* gscale, void* stream) {
    CFN_REQUIRE(gy && w && x && gx && gw, "cfn_pwconv_bwd_fused: null tensor");
    CFN_REQUIRE((A == nullptr) == (B == nullptr), "cfn_pwconv_bwd_fused: A/B mismatch");
    CFN_REQUIRE(A == nullptr || (gA && gB), "cfn_pwconv_bwd_fused: prologue needs gA, gB");
    CFN_REQUIRE(gsumsq == nullptr || y != nullptr, "cfn_pwconv_bwd_fused: gsumsq needs y");
    CFN_REQUIRE(acc == nullptr || acc_stride >= 1, "cfn_pwconv_bwd_fused: bad acc_stride");
    {   // layer-2 widths: the split-bf16 kernel of pwfuseds.hip (CFN_PWF_SPLIT: 0 off, 1 = default: the shapes without a prologue, 2 = all)
        const int rs = pwfs_try_launch(gy, y, gsum, gsumsq, w, x, A, B, act, gx, gA, gB, gw, N, Cin, Cout, T, Hi, Wi, acc, acc_stride, gscale, (hipStream_t)stream);
        if (rs != -1) return rs;
    }
    const long Ql = (long)T * Hi * Wi;
    if (Cin > 64 || Cout > 64 || (Cin > 32 && Cout > 32) || Ql % 4 != 0 || Ql >= (1L << 30)) return -1;
    if ((long)Cout * Ql * 4 >= 0x7ffffff0L || (long)Cin * Ql * 4 >= 0x7ffffff0L) return -1;   // 32-bit buffer offsets per sample
    if (A && act != CFN_ACT_NONE && act != CFN_ACT_RELU && act != CFN_ACT_SWISH) return -1;
    if ((((uintptr_t)gy | (uintptr_t)x | (uintptr_t)(y ? y : gy)) & 15) != 0) return -1;
    { const char* e = getenv("CFN_PWF_OFF"); if (e && atoi(e)) return -1; }
    PfArgs a = {};
    a.gy = gy; a.y = gsumsq ? y : nullptr; a.gs = gsum; a.gq = gsumsq; a.gsc = gscale; a.w = w; a.x = x; a.pa = A; a.pb = B;
    a.gx = gx; a.gA = gA; a.gB = gB; a.gw = gw;
    a.acc = acc; a.acc_s = acc ? acc_stride : 1; a.Hi = Hi; a.Wi = Wi; a.T = T;
    a.acc_Ho = (Hi - 1) / a.acc_s + 1; a.acc_Wo = (Wi - 1) / a.acc_s + 1;
    a.N = N; a.M = Cout; a.K = Cin; a.Q = (int)Ql;
    // >= 4 stages per workgroup
    const long nst = cfn_cdiv(Ql, PF_PT);
    // whole rounds of the chip: 2 workgroups per CU are resident with a prologue (3 without), so 1024 (1536) workgroups
    // are 2 full rounds; 640 (= 1.25 rounds) cost 4.56 instead of 3.83 ms on 24->54 @112
    const bool epi = A != nullptr;
    static const int wgs_env = getenv("CFN_PWF_WGS") ? atoi(getenv("CFN_PWF_WGS")) : 0;      // (measurement switch: workgroups per launch)
    long want = (wgs_env > 0 ? wgs_env : (epi ? 1024 : 1536)) / N;
    if (want < 1) want = 1;
    long stages = cfn_cdiv(nst, want);
    if (stages < 4) stages = 4;
    a.stages = (int)stages;
    a.nstrips = (int)cfn_cdiv(nst, stages);
    const unsigned blocks = (unsigned)((long)N * a.nstrips);
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_PWCONV_BWD, st, 4.0 * N * ((double)Cout * Ql * (a.y ? 2 : 1) + (double)Cin * Ql * 2));
    if (Cout > 32) return pf_launch<2, 1>(a, act, epi, blocks, st);
    if (Cin > 32) return pf_launch<1, 2>(a, act, epi, blocks, st);
    return pf_launch<1, 1>(a, act, epi, blocks, st);
}
