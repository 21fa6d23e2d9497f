// Weight gradient of the pointwise contractions for the MFMA-bound layers (M, K >= 48: X3D layers 2-4;
// x3d_fine.py:100-105 conv1/conv3):   gW[m][k] += sum_q G'[m][q] * a[k][q],   G' = gy + gs + 2 y gq,  a = act(A x + B).
//
// Both MFMA operands are indexed (channel = lane & 31, position pair = lane >> 5), i.e. a lane streams along ONE
// channel row.  Since the contraction runs over positions, their order inside a group of 8 is free: lane half h takes
// positions 4h..4h+3 of the group, so every lane loads one aligned float4 per channel tile straight from HBM into
// VGPRs (no LDS transpose, no workgroup barrier in the main loop), applies the load-time prologue with per-lane
// coefficients held in registers, and feeds 4 MFMA k-steps.  A wave owns up to 3x3 tiles of 32x32 outputs (144
// accumulator registers, 9 MFMAs per 9 loaded float4) over its own chunk of positions; the 8 waves of a workgroup share
// the output tile group and are combined through LDS before one fp64 atomic per element leaves the workgroup.
#include "pw_common.h"
#include <stdlib.h>

struct WdArgs {
    const float* gy; const float* y; const double* gs; const double* gq;
    const float* x; const double* pa; const double* pb;
    double* gw;
    int N, M, K, Q, act;
    const double* gsc;                 // per-(n,m) scale of gy (null = 1)
    int mgroups, kgroups, nstrips;     // output tile groups (<= 3 tiles of 32 each way), workgroups per (group, sample)
    int mt32, kt32;                    // tiles of 32 rows / cols in total
    // XMODE 1: pointwise conv with spatial stride (x row pitch Pin, position map); XMODE 2: dense conv, x rows are
    // im2col rows of a (N,Cimg,Ti,Hi,Wi) tensor
    int Pin, Hi, Wi, Ho, Wo, stride;
    int Cimg, kT, kH, kW, sT, sH, sW, pT, pH, pW, Ti;
};

#define WD_WAVES 8
#define WD_TMAX 3

// balanced split of `tiles` into `groups` runs: run g covers [first, first + count)
__device__ __forceinline__ void wd_split(int tiles, int groups, int g, int& first, int& count) {
    const int base = tiles / groups, rem = tiles - base * groups;
    first = g * base + min(g, rem);
    count = base + (g < rem ? 1 : 0);
}

__device__ __forceinline__ float wd_zfloor(float z) { return fabsf(z) < 1e-20f ? copysignf(1e-20f, z) : z; }

template <int ACT, int XMODE, int WD_TM, int WD_TN>
__global__ __launch_bounds__(64 * WD_WAVES) void pw_wgrad_direct_kernel(const WdArgs a) {
    __shared__ float cw[WD_WAVES][32 * 33];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, row = lane & 31;
    unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    // tile groups vary fastest: workgroups that stream the same positions (and re-read the same operand rows) are
    // neighbours on one XCD and find each other's lines in its L2
    const int kg = L % a.kgroups; L /= a.kgroups;
    const int mg = L % a.mgroups; L /= a.mgroups;
    const int strip = L % a.nstrips;
    const int n = L / a.nstrips;
    const int M = a.M, K = a.K, Q = a.Q;
    int mt0, mtn, kt0, ktn;
    wd_split(a.mt32, a.mgroups, mg, mt0, mtn);   // <= WD_TM row tiles
    wd_split(a.kt32, a.kgroups, kg, kt0, ktn);   // <= WD_TN column tiles
    const int m0 = mt0 * 32, k0 = kt0 * 32;

    // per-lane prologue coefficients (lane <-> channel row of each tile)
    // gy scale z (gsc): G' = z*gy + gs + 2*y*gq = z * (gy + gs/z + y*2gq/z): the loop runs on the bracket with pre-divided
    // coefficients and row m of the result is multiplied by z when it leaves the workgroup -- a resident per-lane z makes
    // the 3x3 variant spill (measured +20 % run time).  |z| is floored at 1e-20 (gamma == 0: the gy term then weighs
    // 1e-20 instead of 0, far below fp32 resolution of the other two terms)
    float cs[WD_TM], cq[WD_TM], ca[WD_TN], cb[WD_TN];
#pragma unroll
    for (int i = 0; i < WD_TM; ++i) {
        const int m = m0 + i * 32 + row;
        const bool ok = i < mtn && m < M;
        cs[i] = (ok && a.gs) ? (float)a.gs[(long)n * M + m] : 0.0f;
        cq[i] = (ok && a.gq && a.y) ? 2.0f * (float)a.gq[(long)n * M + m] : 0.0f;
        if (ok && a.gsc) {
            const float rz = 1.0f / wd_zfloor((float)a.gsc[(long)n * M + m]);
            cs[i] *= rz;
            cq[i] *= rz;
        }
    }
#pragma unroll
    for (int i = 0; i < WD_TN; ++i) {
        const int k = k0 + i * 32 + row;
        const bool okk = i < ktn && k < K && a.pa;
        const long ci = XMODE == 2 ? (long)n * a.Cimg + k / (a.kT * a.kH * a.kW) : (long)n * K + k;
        ca[i] = okk ? a.pa[ci] : 1.0f;
        cb[i] = okk ? a.pb[ci] : 0.0f;
    }
    // XMODE 2: this lane's im2col row of every column tile -> (channel offset, tap)
    int xbase[WD_TN], xkt[WD_TN], xkh[WD_TN], xkw[WD_TN];
    bool xrow[WD_TN];
    if (XMODE == 2) {
#pragma unroll
        for (int j = 0; j < WD_TN; ++j) {
            const int k = k0 + j * 32 + row;
            const int KV = a.kT * a.kH * a.kW;
            const int ci = k / KV, r = k - ci * KV;
            xkt[j] = r / (a.kH * a.kW);
            const int r2 = r - xkt[j] * a.kH * a.kW;
            xkh[j] = r2 / a.kW; xkw[j] = r2 - xkh[j] * a.kW;
            xrow[j] = j < ktn && k < K;
            xbase[j] = min(ci, a.Cimg - 1) * a.Pin;
        }
    }
    // buffer descriptors over one sample: rows >= M / K fall outside and read 0
    const int row_bytes = Q * 4;
    __amdgpu_buffer_rsrc_t rg = cfn_rsrc(const_cast<float*>(a.gy + (long)n * M * Q), M * row_bytes);
    __amdgpu_buffer_rsrc_t ry = cfn_rsrc(const_cast<float*>((a.y ? a.y : a.gy) + (long)n * M * Q), M * row_bytes);
    const long xn = XMODE == 2 ? (long)n * a.Cimg * a.Pin : (long)n * K * a.Pin;
    const unsigned xspan = (unsigned)((XMODE == 2 ? (long)a.Cimg : (long)K) * a.Pin * 4);
    __amdgpu_buffer_rsrc_t rx = cfn_rsrc(const_cast<float*>(a.x + xn), xspan);
    const bool has_y = a.y != nullptr;

    // the workgroup owns a contiguous run of 8-position groups; its waves take them interleaved (wave w: g0+w, g0+w+8,
    // ...) so that the four 32-byte pieces of every 128-byte line are consumed by neighbouring waves at about the same
    // time and the line is fetched from L2 once
    const int g8 = (Q + 7) / 8;
    const int per = (g8 + a.nstrips - 1) / a.nstrips;
    const int gbeg = strip * per + wave, gend = min(strip * per + per, g8);
    f16v acc[WD_TM][WD_TN];
#pragma unroll
    for (int i = 0; i < WD_TM; ++i)
#pragma unroll
        for (int j = 0; j < WD_TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    typedef int __attribute__((ext_vector_type(4))) i4v;
    auto ld4 = [&](__amdgpu_buffer_rsrc_t r, int voff) -> f4v {
        return __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
    };
    f4v rG[WD_TM], rY[WD_TM], rX[WD_TN];
    unsigned xmb[WD_TN] = {};   // XMODE 2: in-bounds bits of the 4 taps of each tile's raw float4
    auto load = [&](int g) {
        const int q = g * 8 + 4 * half;
        const bool inq = q < Q;                                  // Q % 4 == 0: a float4 is all inside or all outside
#pragma unroll
        for (int i = 0; i < WD_TM; ++i) {
            const int vm = inq && i < mtn ? ((m0 + i * 32 + row) * Q + q) * 4 : 0x7ffffff0;
            rG[i] = ld4(rg, vm);
            rY[i] = has_y ? ld4(ry, vm) : (f4v){0.f, 0.f, 0.f, 0.f};
        }
        if (XMODE == 0) {
#pragma unroll
            for (int i = 0; i < WD_TN; ++i) {
                const int vk = inq && i < ktn ? ((k0 + i * 32 + row) * Q + q) * 4 : 0x7ffffff0;
                rX[i] = ld4(rx, vk);
            }
        } else {
            // the 4 output positions q..q+3 sit in one output row (Wo % 4 == 0): decode once
            const int hw = a.Ho * a.Wo;
            const int qc = inq ? q : 0;
            const int to = qc / hw, rq = qc - to * hw;
            const int oh = rq / a.Wo, ow = rq - oh * a.Wo;
#pragma unroll
            for (int i = 0; i < WD_TN; ++i) {
                f4v v = {0.f, 0.f, 0.f, 0.f};
                if (XMODE == 1) {
                    if (inq && i < ktn) {
                        const int vo = ((k0 + i * 32 + row) * a.Pin + (to * a.Hi + oh * a.stride) * a.Wi + ow * a.stride) * 4;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vo + e * a.stride * 4, 0, 0));
                    }
                } else {
                    const int it = to * a.sT + xkt[i] - a.pT, ih = oh * a.sH + xkh[i] - a.pH;
                    const bool okr = inq && xrow[i] && it >= 0 && it < a.Ti && ih >= 0 && ih < a.Hi;
                    const int iw0 = ow * a.sW + xkw[i] - a.pW;
                    const int vo = (xbase[i] + (it * a.Hi + ih) * a.Wi + iw0) * 4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int iw = iw0 + e * a.sW;
                        const bool ok = okr && iw >= 0 && iw < a.Wi;
                        v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, ok ? vo + e * a.sW * 4 : 0x7ffffff0, 0, 0));
                        xmb[i] = e == 0 ? (ok ? 1u : 0u) : (xmb[i] | ((ok ? 1u : 0u) << e));
                    }
                }
                rX[i] = v;
            }
        }
        return inq;
    };
    bool inq = false;
    if (gbeg < gend) inq = load(gbeg);
    for (int g = gbeg; g < gend; g += WD_WAVES) {
        f4v G[WD_TM], X[WD_TN];
        const float vm = inq ? 1.0f : 0.0f;                      // masks the constant terms of an out-of-range group
#pragma unroll
        for (int i = 0; i < WD_TM; ++i) G[i] = rG[i] + rY[i] * cq[i] + cs[i] * vm;
#pragma unroll
        for (int i = 0; i < WD_TN; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xa = cfn_act<ACT>(fmaf(rX[i][e], ca[i], cb[i]));
                X[i][e] = XMODE == 2 ? (((xmb[i] >> e) & 1u) ? xa : 0.0f) : xa * vm;   // zero padding applies after the prologue
            }
        }
        if (g + WD_WAVES < gend) inq = load(g + WD_WAVES);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < WD_TM; ++i)
#pragma unroll
                for (int j = 0; j < WD_TN; ++j)
                    if (i < mtn && j < ktn) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(G[i][s], X[j][s], acc[i][j], 0, 0, 0);
    }

    // ---- combine the 8 waves tile by tile through LDS, one fp64 atomic per element per workgroup ----------
#pragma unroll
    for (int i = 0; i < WD_TM; ++i)
#pragma unroll
        for (int j = 0; j < WD_TN; ++j) {
            if (i < mtn && j < ktn) {                            // workgroup uniform
#pragma unroll
                for (int r = 0; r < 16; ++r) cw[wave][((r & 3) + 8 * (r >> 2) + 4 * half) * 33 + row] = acc[i][j][r];
                __syncthreads();
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int el = tid + e * 64 * WD_WAVES;      // 1024 elements of the tile
                    const int ml = el >> 5, kl = el & 31;
                    float v = 0.0f;
#pragma unroll
                    for (int w = 0; w < WD_WAVES; ++w) v += cw[w][ml * 33 + kl];
                    const int m = m0 + i * 32 + ml, k = k0 + j * 32 + kl;
                    if (m < M && k < K) {
                        if (a.gsc) v *= wd_zfloor((float)a.gsc[(long)n * M + m]);
                        cfn_add64(&a.gw[(long)m * K + k], (double)v);
                    }
                }
                __syncthreads();
            }
        }
}

template <int XMODE, int WD_TM, int WD_TN>
static int wd_launch(WdArgs& a, hipStream_t st) {
    constexpr int WD_T = WD_TM * WD_TN;                 // 1 = the single-tile variant
    a.mt32 = cfn_cdiv(a.M, 32); a.kt32 = cfn_cdiv(a.K, 32);
    a.mgroups = cfn_cdiv(a.mt32, WD_TM); a.kgroups = cfn_cdiv(a.kt32, WD_TN);
    const long groups = (long)a.N * a.mgroups * a.kgroups;
    // ~one workgroup per CU (measured: more, shorter strips lose); the single-tile variant is load bound and small:
    // four workgroups per CU
    // (re-measured with the round-2 kernels, 8 clips: 2x2 / 2x3 tiles everywhere lose 3-12 % against the shapes chosen below;
    //  two workgroups per CU gain 5-6 % for the 2x4 / 4x2 groups of layer 2 and nothing elsewhere)
    static const int wg_env = getenv("CFN_PWD_WGS") ? atoi(getenv("CFN_PWD_WGS")) : 0;
    long strips = (wg_env > 0 ? wg_env : (WD_T == 1 ? 1024 : (WD_T == 8 ? 512 : 256))) / groups;
    if (strips < 1) strips = 1;
    const long g8 = cfn_cdiv(a.Q, 8);
    if (strips > cfn_cdiv(g8, WD_WAVES * 4)) strips = cfn_cdiv(g8, WD_WAVES * 4);   // >= 4 position groups per wave
    a.nstrips = (int)strips;
    const unsigned blocks = (unsigned)(groups * strips);
    switch (a.act) {
        case CFN_ACT_RELU: hipLaunchKernelGGL((pw_wgrad_direct_kernel<CFN_ACT_RELU, XMODE, WD_TM, WD_TN>), dim3(blocks), dim3(64 * WD_WAVES), 0, st, a); break;
        case CFN_ACT_SWISH: hipLaunchKernelGGL((pw_wgrad_direct_kernel<CFN_ACT_SWISH, XMODE, WD_TM, WD_TN>), dim3(blocks), dim3(64 * WD_WAVES), 0, st, a); break;
        default: hipLaunchKernelGGL((pw_wgrad_direct_kernel<CFN_ACT_NONE, XMODE, WD_TM, WD_TN>), dim3(blocks), dim3(64 * WD_WAVES), 0, st, a); break;
    }
    return cfn_check_launch("pwconv_bwd_weight(direct)");
}

static bool wd_common_ok(const float* gy, const float* y, const float* x, int act, int M, int K, int Q) {
    if (Q % 4 != 0) return false;
    if (act != CFN_ACT_NONE && act != CFN_ACT_RELU && act != CFN_ACT_SWISH) return false;
    if (((uintptr_t)gy | (uintptr_t)(y ? y : gy)) & 15) return false;
    if ((long)M * Q * 4 >= (1L << 31) - 64) return false;
    const char* e = getenv("CFN_PWD_OFF");
    return !(e && atoi(e));
}

// contiguous pointwise conv (stride 1): M, K >= 48 (smaller layers are HBM bound and stay on the LDS-staged kernel)
int pwd_wgrad_try_launch(const float* gy, const float* y, const double* gs, const double* gq, const double* gsc, const float* x,
                         const double* pa, const double* pb, int act, double* gw, int N, int M, int K, int Q, hipStream_t st) {
    if (M < 48 || K < 48 || !wd_common_ok(gy, y, x, act, M, K, Q)) return -1;
    if (((uintptr_t)x & 15) || (long)K * Q * 4 >= (1L << 31) - 64) return -1;
    WdArgs a = {gy, y, gs, gq, x, pa, pb, gw, N, M, K, Q, act, gsc};
    a.Pin = Q;
    {   // balanced groups never exceed 2 tiles either way (e.g. 108 x 48): the 2x2 variant needs half the registers
        const int mt = cfn_cdiv(M, 32), kt = cfn_cdiv(K, 32);
        const int gm = cfn_cdiv(mt, cfn_cdiv(mt, 3)), gk = cfn_cdiv(kt, cfn_cdiv(kt, 3));
        // narrow x wide (e.g. 48 x 108): one 2x4 / 4x2 group instead of two 2x2 groups halves the re-reads of the operand
        // both groups share
        if (mt <= 2 && kt == 4) return wd_launch<0, 2, 4>(a, st);
        if (mt == 4 && kt <= 2) return wd_launch<0, 4, 2>(a, st);
        if (gm <= 2 && gk <= 2) return wd_launch<0, 2, 2>(a, st);
        // 7 row tiles x 3 column tiles (216 x 96): row groups of 3 leave a 1-tile group (7/9 of the MFMA slots used), groups
        // of 2 use 7/8 (measured 0.363 -> 0.332 ms; the transposed 3 x 7 case shows no difference and stays on 3x3)
        if (mt == 7 && kt == 3) return wd_launch<0, 2, 3>(a, st);
    }
    return wd_launch<0, 3, 3>(a, st);
}

// pointwise conv with spatial stride 2 (shortcut convs): gathered x operand
int pwd_wgrad_try_strided(const float* gy, const float* y, const double* gs, const double* gq, const double* gsc, const float* x,
                          const double* pa, const double* pb, int act, double* gw, int N, int M, int K, int T, int Hi, int Wi,
                          int stride, hipStream_t st) {
    const int Ho = (Hi - 1) / stride + 1, Wo = (Wi - 1) / stride + 1;
    const int Q = T * Ho * Wo;
    if (Wo % 4 != 0 || !wd_common_ok(gy, y, x, act, M, K, Q)) return -1;
    if ((long)K * T * Hi * Wi * 4 >= (1L << 31) - 64) return -1;
    WdArgs a = {gy, y, gs, gq, x, pa, pb, gw, N, M, K, Q, act, gsc};
    a.Pin = T * Hi * Wi; a.Hi = Hi; a.Wi = Wi; a.Ho = Ho; a.Wo = Wo; a.stride = stride;
    return wd_launch<1, 3, 3>(a, st);
}

// dense conv (stem 1x3x3 / Grid Pool saliency convs): x rows are im2col rows; geom = {kT,kH,kW,sT,sH,sW,pT,pH,pW}
int pwd_wgrad_try_dense(const float* gy, const float* y, const double* gs, const double* gq, const float* x, const double* pa,
                        const double* pb, int act, double* gw, int N, int M, int Cimg, int T, int Hi, int Wi, const int* g,
                        hipStream_t st) {
    const int To = (T + 2 * g[6] - g[0]) / g[3] + 1, Ho = (Hi + 2 * g[7] - g[1]) / g[4] + 1, Wo = (Wi + 2 * g[8] - g[2]) / g[5] + 1;
    const int K = Cimg * g[0] * g[1] * g[2];
    const long Ql = (long)To * Ho * Wo;
    if (To < 1 || Ho < 1 || Wo < 1 || Wo % 4 != 0 || Ql >= (1L << 30)) return -1;
    const int Q = (int)Ql;
    if (!wd_common_ok(gy, y, x, act, M, K, Q)) return -1;
    if ((long)Cimg * T * Hi * Wi * 4 >= (1L << 31) - 64) return -1;
    WdArgs a = {gy, y, gs, gq, x, pa, pb, gw, N, M, K, Q, act};
    a.Pin = T * Hi * Wi; a.Hi = Hi; a.Wi = Wi; a.Ho = Ho; a.Wo = Wo; a.stride = 1; a.Ti = T; a.Cimg = Cimg;
    a.kT = g[0]; a.kH = g[1]; a.kW = g[2]; a.sT = g[3]; a.sH = g[4]; a.sW = g[5]; a.pT = g[6]; a.pH = g[7]; a.pW = g[8];
    // one 32x32 tile per wave when the whole problem is one or two tiles (stem: 24 x 27): no idle tile slots
    if (cfn_cdiv(M, 32) * cfn_cdiv(K, 32) <= 2) return wd_launch<2, 1, 1>(a, st);
    return wd_launch<2, 3, 3>(a, st);
}
