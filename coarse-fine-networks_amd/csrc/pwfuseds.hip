// hipcc-flags: -fno-slp-vectorize
// Fused backward of a stride-1 pointwise (1x1x1) convolution at the layer-2 widths (48 / 108 channels, x3d_fine.py:100-105 conv1 / conv3 of
// res3): data gradient AND weight gradient in one pass over gy, y, x, on the bf16 matrix pipe with every fp32 operand split into three bf16
// terms (the 6-term product of pws_kernel.h: fp32-accurate).
//
// Same pass structure as pw_bwd_fused_kernel (pwfused.hip: layer 1, fp32 MFMA): a workgroup (4 waves) owns one sample and a strip of
// positions and walks it in stages of 64 positions; G' = gsc*gy + gs + 2*y*gq and the RAW x rows are staged in LDS as fp32 [channel][65]
// (float4 global loads one stage ahead in registers, double-buffered images, one barrier per stage); wave w owns positions 16w .. 16w+15 of a
// stage for BOTH products.  What differs is the arithmetic:
//   weight gradient: v_mfma_f32_32x32x16_bf16, k = the wave's 16 positions: a lane reads 8 consecutive positions of its G' row (x row) from
//                    the fp32 image, applies the prologue to x, and splits them into three packed-bf16 operands in registers;
//   data gradient:   v_mfma_f32_16x16x32_bf16, k = 32 output channels: the B operand is the lane's position, 8 consecutive G' rows, split in
//                    registers; the A operand W^T sits PRE-SPLIT in LDS (three [ci][co] bf16 images written once per workgroup) and is
//                    read 16 bytes at a time.  The C layout (lane <-> position, 4 channel rows) is the one pw_bwd_fused_kernel's epilogue
//                    works on: act' epilogue, statistics partials, compact shortcut gradient, 64-byte row-segment stores are the same code.
// Registers: 8 accumulator tiles of the weight gradient per wave (128) + operands: one wave per SIMD (launch bounds 256, 1); the matrix
// work per stage (3,072 cycles per SIMD) and the conversions are a third of the stage's HBM time, which is what has to be hidden.
#include "cfn_common.h"
#include <stdlib.h>
#include <map>
#include <mutex>

#include "pw_common.h"

typedef float __attribute__((ext_vector_type(4))) pf4;
typedef __bf16 pfs_bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 pfs_bf2 __attribute__((ext_vector_type(2)));
typedef float pfs_f2 __attribute__((ext_vector_type(2)));
typedef unsigned pfs_u4 __attribute__((ext_vector_type(4)));

#define PFS_PT 64
#define PFS_PITCH 65

__device__ __forceinline__ float pfs_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float pfs_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ unsigned pfs_pack(float lo, float hi) {
    const pfs_bf2 b = __builtin_convertvector((pfs_f2){lo, hi}, pfs_bf2);
    return __builtin_bit_cast(unsigned, b);
}
// (v0, v1) -> three packed bf16 pairs whose sum is (v0, v1) to fp32 accuracy
__device__ __forceinline__ void pfs_split3(float v0, float v1, unsigned (&p)[3]) {
    p[0] = pfs_pack(v0, v1);
    v0 -= pfs_lo(p[0]); v1 -= pfs_hi(p[0]);
    p[1] = pfs_pack(v0, v1);
    v0 -= pfs_lo(p[1]); v1 -= pfs_hi(p[1]);
    p[2] = pfs_pack(v0, v1);
}
// the six leading terms of (a0 + a1 + a2)(b0 + b1 + b2), smallest first (pws_kernel.h)
#define PFS_TERMS(F) F(2, 0) F(0, 2) F(1, 1) F(1, 0) F(0, 1) F(0, 0)

struct PfsArgs {
    const float* gy; const float* y; const double* gs; const double* gq; const double* gsc;
    const float* w;                       // (Cout, Cin) row major
    const float* x; const double* pa; const double* pb;
    float* gx; double* gA; double* gB; double* gw;
    const float* acc; int acc_s, acc_Ho, acc_Wo, Hi, Wi, T;
    int N, M, K, Q, nstrips, stages;      // M = Cout (rows of G'), K = Cin (rows of x)
    int dbg;                              // knock-outs for tools/pwfs_knockouts.sh (CFN_PWFS_DBG; 0 in the product): 1 weight gradient, 2 data gradient, 4 act' / statistics, 8 gx stores
};

// ACT < 0: the forward conv had no prologue (gx = da, no statistics)
// 8 waves, specialised by product: waves 0-3 own the weight gradient of positions 16 (w & 3) .. + 15 of every stage, waves 4-7 the data gradient,
// its epilogue and the statistics of the same positions.  Each SIMD then holds one wave of either kind -- the matrix instructions of one
// overlap the conversions / epilogue of the other -- and neither loop carries the other's registers (128 accumulators here, 32 statistics
// partials there).  All 512 threads stage.
template <int MTW, int NTW, int ACT>
__global__ __launch_bounds__(512, 1) void pw_bwd_fused_split_kernel(const PfsArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr bool EPI = ACT >= 0;
    constexpr int ACTV = EPI ? ACT : CFN_ACT_NONE;
    constexpr int BM = 32 * MTW, BN = 32 * NTW;
    constexpr int NG = BM / 16, NX = BN / 16;      // float4 per STAGING thread per stage (G rows / X rows): the 256 threads of the weight-gradient waves stage
    constexpr int NT16 = BN / 16, KS = BM / 32;     // data gradient: 16-row tiles of input channels, k-steps of 32 output channels
    constexpr bool ROWSPLIT = MTW >= NTW;           // weight gradient: the two tile halves are split along the longer side
    constexpr int MI = ROWSPLIT ? MTW / 2 : MTW, NJ = ROWSPLIT ? NTW : NTW / 2;
    constexpr int BMP = BM + 8;                     // W^T image: bf16 elements per input-channel row (272 / 144 bytes: 16-byte reads, rows 17 / 9 slots apart)
    const int tid = threadIdx.x, wave = cfn_uni((int)(tid >> 6)), lane = tid & 63, half = lane >> 5, col = lane & 31;
    const int m16 = lane & 15, kq = lane >> 4;
    unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    const int strip = L % a.nstrips;
    const int n = L / a.nstrips;
    const int M = a.M, K = a.K, Q = a.Q;

    constexpr int IMG = (BM + BN) * PFS_PITCH;         // one staged image: G' rows [BM][65] then raw x rows [BN][65]
    float* img0 = smem;                                // two images (double buffer)
    float* sCx = smem + 2 * IMG + 3 * BM;              // [BN][2]  (A, B) of the epilogue (3 BM floats in front stay unused: the layer-1 kernel keeps its G' coefficients there)
    float* sSt = sCx + 2 * BN;                         // [4 waves][BN][2]  statistics of this workgroup
    unsigned* sW = reinterpret_cast<unsigned*>(sSt + 8 * BN);      // [3 terms][BN][BMP / 2] packed bf16 pairs (co, co + 1) of W^T
    for (int k = tid; k < BN; k += 512) {
        const bool ok = EPI && k < K;
        sCx[2 * k] = ok ? (float)a.pa[(long)n * K + k] : 1.0f;
        sCx[2 * k + 1] = ok ? (float)a.pb[(long)n * K + k] : 0.0f;
    }
    for (int e = tid; e < BN * (BM / 2); e += 512) {   // W^T, split once per workgroup
        const int ci = e / (BM / 2), mp = e - ci * (BM / 2), co = 2 * mp;
        const float w0 = (ci < K && co < M) ? a.w[(long)co * K + ci] : 0.0f;
        const float w1 = (ci < K && co + 1 < M) ? a.w[(long)(co + 1) * K + ci] : 0.0f;
        unsigned p[3];
        pfs_split3(w0, w1, p);
#pragma unroll
        for (int s = 0; s < 3; ++s) sW[(s * BN + ci) * (BMP / 2) + mp] = p[s];
    }
    __syncthreads();

    // staging threads = waves 0-3 (the data-gradient waves carry 32-64 statistics registers instead of the prefetched rows).  A wave covers 4 rows x 16
    // float4 segments of a stage; lanes 0-31 take segments 0-7 of the four rows, lanes 32-63 segments 8-15: the four scalar LDS writes of a float4 then
    // hit 32 distinct banks per half wave (bank = row + 4 seg + e with the odd pitch; 2 rows x 16 segments put two lanes on every bank), and 8 lanes
    // still load one 128-byte line
    const int lrow = 4 * ((tid & 255) >> 6) + ((lane >> 3) & 3), c4 = ((lane & 7) + 8 * (lane >> 5)) * 4;
    // unconditional buffer loads / stores (unwanted ones get an out-of-range offset), as in pw_bwd_fused_kernel
    constexpr int OOB = 0x7ffffff0;
    __amdgpu_buffer_rsrc_t rg = cfn_rsrc(const_cast<float*>(a.gy + (long)n * M * Q), (unsigned)((long)M * Q * 4));
    __amdgpu_buffer_rsrc_t ry = cfn_rsrc(const_cast<float*>((a.y ? a.y : a.gy) + (long)n * M * Q), a.y ? (unsigned)((long)M * Q * 4) : 0u);
    __amdgpu_buffer_rsrc_t rx = cfn_rsrc(const_cast<float*>(a.x + (long)n * K * Q), (unsigned)((long)K * Q * 4));
    __amdgpu_buffer_rsrc_t rd = cfn_rsrc(a.gx + (long)n * K * Q, (unsigned)((long)K * Q * 4));
    pf4 pg[NG], py[NG], px[NX];
    int vog[NG], vox[NX];                                    // byte offsets of this thread's row segments (position 0)
    // the (gs, 2 gq, gsc) of this thread's G' rows live in REGISTERS, read from global memory once: a coefficient pair that a wide LDS read has just
    // returned must not be the operand of a packed FMA beside a matrix-bound wave (the round-3 race; the first build of this kernel,
    // which read them from an LDS table inside `stage`, reproduced it: one G' row wrong in lanes 48-63 in 1-7 of 100 launches)
    float rcs[NG], rcq[NG], rcz[NG];
#pragma unroll
    for (int it = 0; it < NG; ++it) {
        const int row = it * 16 + lrow;
        const bool ok = row < M;
        rcs[it] = (ok && a.gs) ? (float)a.gs[(long)n * M + row] : 0.0f;
        rcq[it] = (ok && a.gq && a.y) ? 2.0f * (float)a.gq[(long)n * M + row] : 0.0f;
        rcz[it] = (ok && a.gsc) ? (float)a.gsc[(long)n * M + row] : 1.0f;
    }
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
            const float cs = rcs[it], cq = rcq[it], cz = rcz[it];
            const bool ok = inq && (row < M);
            float* d = sG + row * PFS_PITCH + c4;
            d[0] = ok ? fmaf(py[it].x, cq, fmaf(pg[it].x, cz, cs)) : 0.0f;
            d[1] = ok ? fmaf(py[it].y, cq, fmaf(pg[it].y, cz, cs)) : 0.0f;
            d[2] = ok ? fmaf(py[it].z, cq, fmaf(pg[it].z, cz, cs)) : 0.0f;
            d[3] = ok ? fmaf(py[it].w, cq, fmaf(pg[it].w, cz, cs)) : 0.0f;
        }
#pragma unroll
        for (int it = 0; it < NX; ++it) {
            const int row = it * 16 + lrow;
            float* d = sX + row * PFS_PITCH + c4;            // raw x; rows >= K and positions >= Q were loaded as 0
            d[0] = px[it].x; d[1] = px[it].y; d[2] = px[it].z; d[3] = px[it].w;
        }
    };

    const int qbeg = strip * a.stages * PFS_PT;
    const int nst = min(a.stages, (Q - qbeg + PFS_PT - 1) / PFS_PT);
    if (nst > 0 && wave < 4) {
        prefetch(qbeg);
        stage(qbeg, img0, img0 + BM * PFS_PITCH);
        if (nst > 1) prefetch(qbeg + PFS_PT);
    }
    __syncthreads();
    const int pbase = (wave & 3) * (PFS_PT / 4);
    constexpr int CWP = 33, CWT = 32 * CWP;                 // the weight-gradient waves leave their tiles here after the last stage: [4 waves][MTW * NTW][32][33]

    if (wave < 4) {
        // ================= weight gradient: wave (th, ph) owns HALF the tiles (th) over positions 32 ph .. 32 ph + 31 of every stage (two k-blocks);
        // lane (row = col, k = 8 half + e).  64 accumulator registers per wave; the two position halves of a tile meet in LDS after the last stage
        const int th = wave & 1, ph = wave >> 1;
        const int i0 = ROWSPLIT ? th * MI : 0, j0 = ROWSPLIT ? 0 : th * NJ;
        float ca[NJ], cb[NJ];                                // prologue coefficients of this lane's x rows (row (j0 + j)*32 + col)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int k = (j0 + j) * 32 + col;
            const bool ok = EPI && k < K;
            ca[j] = ok ? (float)a.pa[(long)n * K + k] : 1.0f;
            cb[j] = ok ? (float)a.pb[(long)n * K + k] : 0.0f;
        }
        f16v acc[MI][NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        for (int st = 0; st < nst; ++st) {
            const int q0 = qbeg + st * PFS_PT;
            float* cur = img0 + (st & 1) * IMG;
            float* nxt = img0 + ((st + 1) & 1) * IMG;
            if (st + 1 < nst) {
                stage(q0 + PFS_PT, nxt, nxt + BM * PFS_PITCH);
                if (st + 2 < nst) prefetch(q0 + 2 * PFS_PT);
            }
            const float* sG = cur;
            const float* sX = cur + BM * PFS_PITCH;
            if (!(a.dbg & 1))
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int p0 = 32 * ph + 16 * kb + 8 * half;
                pfs_u4 Bf[NJ][3];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const float* r = sX + ((j0 + j) * 32 + col) * PFS_PITCH + p0;
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        float x0 = r[2 * h], x1 = r[2 * h + 1];
                        if (EPI) { x0 = cfn_act<ACTV>(fmaf(x0, ca[j], cb[j])); x1 = cfn_act<ACTV>(fmaf(x1, ca[j], cb[j])); }
                        unsigned p[3];
                        pfs_split3(x0, x1, p);
#pragma unroll
                        for (int s = 0; s < 3; ++s) Bf[j][s][h] = p[s];
                    }
                }
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const float* r = sG + ((i0 + i) * 32 + col) * PFS_PITCH + p0;
                    pfs_u4 Af[3];
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        unsigned p[3];
                        pfs_split3(r[2 * h], r[2 * h + 1], p);
#pragma unroll
                        for (int s = 0; s < 3; ++s) Af[s][h] = p[s];
                    }
#define PFS_WG(SA, SB) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pfs_bf8, Af[SA]), __builtin_bit_cast(pfs_bf8, Bf[j][SB]), acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) { PFS_TERMS(PFS_WG) }
#undef PFS_WG
                }
            }
            __syncthreads();
        }
        __syncthreads();                                     // (A) the data-gradient waves have left their statistics in sSt
        __syncthreads();                                     // (B) ... and they have been read: the LDS below sW's end is free
        float* cw = smem + wave * (MI * NJ * CWT);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) cw[(i * NJ + j) * CWT + ((r & 3) + 8 * (r >> 2) + 4 * half) * CWP + col] = acc[i][j][r];
    } else {
        // ================= data gradient of the same 16 positions: lane (position m16, k = 32 s + 8 kq + e), epilogue, statistics =================
        float sa[NT16][4], sb[NT16][4];                      // per-lane partial statistics of rows t*16 + 4*kq + r
#pragma unroll
        for (int t = 0; t < NT16; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { sa[t][r] = 0.0f; sb[t][r] = 0.0f; }
        const int hw = a.Hi * a.Wi;
        const int acc_pitch4 = a.T * a.acc_Ho * a.acc_Wo * 4;   // bytes per channel of the compact gradient
        __amdgpu_buffer_rsrc_t racc = cfn_rsrc(const_cast<float*>(a.acc ? a.acc + (long)n * K * (acc_pitch4 / 4) : a.gy), a.acc ? (unsigned)((long)K * acc_pitch4) : 0u);
        for (int st = 0; st < nst; ++st) {
            const int q0 = qbeg + st * PFS_PT;
            float* cur = img0 + (st & 1) * IMG;
            const float* sG = cur;
            const float* sX = cur + BM * PFS_PITCH;
            pf4 da[NT16];
#pragma unroll
            for (int t = 0; t < NT16; ++t) da[t] = (pf4){0.f, 0.f, 0.f, 0.f};
            if (!(a.dbg & 2))
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const float* r = sG + (32 * s + 8 * kq) * PFS_PITCH + pbase + m16;
                pfs_u4 Gf[3];
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    unsigned p[3];
                    pfs_split3(r[(2 * h) * PFS_PITCH], r[(2 * h + 1) * PFS_PITCH], p);
#pragma unroll
                    for (int u = 0; u < 3; ++u) Gf[u][h] = p[u];
                }
#pragma unroll
                for (int t = 0; t < NT16; ++t) {
                    const unsigned* wr = sW + ((t * 16 + m16) * BMP + 32 * s + 8 * kq) / 2;
                    pfs_u4 Wf[3];
#pragma unroll
                    for (int u = 0; u < 3; ++u) Wf[u] = *reinterpret_cast<const pfs_u4*>(wr + u * BN * (BMP / 2));
#define PFS_DG(SA, SB) da[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pfs_bf8, Wf[SA]), __builtin_bit_cast(pfs_bf8, Gf[SB]), da[t], 0, 0, 0);
                    PFS_TERMS(PFS_DG)
#undef PFS_DG
                    if (NT16 > 4 && (t & 1)) __builtin_amdgcn_sched_barrier(0);      // (two tiles' W^T operands in flight, not all sixteen)
                }
            }
            const int q = q0 + pbase + m16;
            const bool qv = q < Q;
            const int gvo = (qv && !(a.dbg & 8)) ? (4 * kq * Q + pbase + m16) * 4 : OOB;
            int aoff = OOB;                                        // compact lattice byte offset of this lane's position
            if (a.acc && qv) {
                const int tq = q / hw, rq = q - tq * hw;
                const int hq = rq / a.Wi, wq_ = rq - hq * a.Wi;
                if (hq % a.acc_s == 0 && wq_ % a.acc_s == 0) aoff = (((tq * a.acc_Ho + hq / a.acc_s) * a.acc_Wo + wq_ / a.acc_s)) * 4;
            }
            const unsigned arow = (unsigned)aoff + (unsigned)(4 * kq) * (unsigned)acc_pitch4;
            // the compact shortcut gradient of this lane's outputs: all loads up front where the registers allow it (<= 4 channel tiles: 16 values),
            // tile by tile otherwise; the per-lane part of the row goes into the VECTOR offset (a lane-dependent scalar offset would be a waterfall loop)
            constexpr bool AV_UPFRONT = NT16 <= 4;
            float av4[AV_UPFRONT ? NT16 : 1][4];
            if (AV_UPFRONT) {
#pragma unroll
                for (int t = 0; t < NT16; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) av4[AV_UPFRONT ? t : 0][r] = 0.0f;
                if (a.acc) {                                       // workgroup uniform
#pragma unroll
                    for (int t = 0; t < NT16; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            av4[AV_UPFRONT ? t : 0][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                racc, (int)(aoff == OOB ? (unsigned)OOB : arow + (unsigned)r * (unsigned)acc_pitch4), t * 16 * acc_pitch4, 0));
                }
            }
#pragma unroll
            for (int t = 0; t < NT16; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ci = t * 16 + 4 * kq + r;
                    float v = da[t][r];                            // exactly 0 for positions >= Q (G' = 0 there) and rows >= K (zero rows of W^T)
                    if (AV_UPFRONT) v += av4[AV_UPFRONT ? t : 0][r];
                    else if (a.acc)
                        v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                            racc, (int)(aoff == OOB ? (unsigned)OOB : arow + (unsigned)r * (unsigned)acc_pitch4), t * 16 * acc_pitch4, 0));
                    if (EPI && !(a.dbg & 4)) {
                        const float xr = sX[ci * PFS_PITCH + pbase + m16];
                        const float2 pab = cfn_settle(*reinterpret_cast<const float2*>(sCx + 2 * ci));      // (an LDS pair in front of FMAs: DESIGN 4.1)
                        const float pa = pab.x, pb = pab.y;
                        const float dz = v * cfn_act_grad<ACTV>(fmaf(xr, pa, pb));
                        sa[t][r] = fmaf(dz, xr, sa[t][r]);
                        sb[t][r] += dz;
                        v = dz * pa;
                    }
                    // ONE lane offset for all the wave's stores (row 4 kq, this position); the (t, r) part of the row is wave uniform and rides in the scalar
                    // offset; rows >= K fall outside the descriptor and are dropped by the hardware -- 32 per-output offsets kept live across the stage loop
                    // were what spilled the statistics registers
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rd, gvo, (q0 + (t * 16 + r) * Q) * 4, 0);
                }
                if (NT16 > 4) __builtin_amdgcn_sched_barrier(0);   // one channel tile's LDS reads in flight at a time: hoisting all 8 tiles' reads spills the 64 statistics registers
            }
            __syncthreads();
        }
        // ---- statistics: reduce over the 16 position lanes, then over the 4 data-gradient waves in LDS, one fp64 atomic per row -------
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
                        sSt[((wave & 3) * BN + ci) * 2] = u;
                        sSt[((wave & 3) * BN + ci) * 2 + 1] = v;
                    }
                }
        }
        __syncthreads();                                     // (A)
        const int t2 = tid - 256;
        if (EPI && t2 < K) {
            const float u = (sSt[2 * t2] + sSt[(BN + t2) * 2]) + (sSt[(2 * BN + t2) * 2] + sSt[(3 * BN + t2) * 2]);
            const float v = (sSt[2 * t2 + 1] + sSt[(BN + t2) * 2 + 1]) + (sSt[(2 * BN + t2) * 2 + 1] + sSt[(3 * BN + t2) * 2 + 1]);
            cfn_add64(&a.gA[(long)n * K + t2], (double)u);
            cfn_add64(&a.gB[(long)n * K + t2], (double)v);
        }
        __syncthreads();                                     // (B)
    }
    __syncthreads();                                         // the weight-gradient tiles of the four waves are in LDS
    // ---- weight gradient: a tile = wave (th, 0) + wave (th, 1); one fp64 atomic per element and workgroup ----
    constexpr int WT = MI * NJ;                              // tiles per wave
    for (int e = tid; e < 2 * WT * 1024; e += 512) {
        const int th = e / (WT * 1024), tile = (e >> 10) % WT, ml = (e >> 5) & 31, kl = e & 31, o = tile * CWT + ml * CWP + kl;
        const float v = smem[th * (WT * CWT) + o] + smem[(2 + th) * (WT * CWT) + o];
        const int i = tile / NJ, j = tile - i * NJ;
        const int gm = ((ROWSPLIT ? th * MI : 0) + i) * 32 + ml, gk = ((ROWSPLIT ? 0 : th * NJ) + j) * 32 + kl;
        if (gm < M && gk < K) cfn_add64(&a.gw[(long)gm * K + gk], (double)v);
    }
}

template <int MTW, int NTW>
static int pfs_launch(const PfsArgs& a, int act, bool epi, unsigned blocks, hipStream_t st) {
    constexpr int BM = 32 * MTW, BN = 32 * NTW, BMP = BM + 8;
    size_t lds = ((size_t)2 * (BM + BN) * PFS_PITCH + 3 * BM + 10 * BN) * sizeof(float) + (size_t)3 * BN * BMP * 2;
    const size_t lds_cw = (size_t)4 * (MTW * NTW / 2) * 32 * 33 * sizeof(float);   // the four weight-gradient waves' tiles, after the last stage
    if (lds_cw > lds) lds = lds_cw;
#define CFN_PFS_GO(ACTV)                                                                                        \
    do {                                                                                                        \
        auto k = pw_bwd_fused_split_kernel<MTW, NTW, ACTV>;                                                     \
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);        \
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), lds, st, a);                                             \
    } while (0)
    if (!epi) CFN_PFS_GO(-1);
    else if (act == CFN_ACT_RELU) CFN_PFS_GO(CFN_ACT_RELU);
    else if (act == CFN_ACT_SWISH) CFN_PFS_GO(CFN_ACT_SWISH);
    else CFN_PFS_GO(CFN_ACT_NONE);
#undef CFN_PFS_GO
    return cfn_check_launch("pwconv_bwd_fused (split bf16)");
}

// ---- layer 3 (conv1 of res4's blocks: 96 -> 216, no prologue): W^T pre-split is 129 KB and does not fit beside the images.  Same products, different budget: ----
//   * stages of 32 positions (two fp32 images of 320 rows x 33: 84 KB), W^T pre-split ONCE per launch into a global workspace (3 x [96][232] bf16) that the
//     data-gradient waves read 16 bytes at a time with buffer loads -- every workgroup reads the same 129 KB, so it lives in L2 (126 KB per wave and stage: a seventh
//     of the L2 rate a CU can draw);
//   * 4 weight-gradient waves + 3 data-gradient waves + 1 staging wave: wave w < 4 stages the G' rows and owns the row tiles i = w, w + 4 x all three column tiles over
//     ALL positions of a stage (96 accumulator registers, no cross-wave sum at the end); wave 4 + d (d < 3) owns the data gradient of input channels 32 d .. 32 d + 31 for
//     all 32 positions and keeps ITS third of W^T resident in 168 registers (read once per workgroup from the workspace); wave 7 stages the x rows.  (Builds with the W^T
//     operands streamed out of L2 inside the stage loop -- 6 + 2 and 4 + 4 waves -- ran at 0.56-0.57 ms against 0.45 for the separate kernels: seven dependent
//     load -> MFMA rounds per stage at two waves per SIMD.)
#define PF3_PT 32
#define PF3_PITCH 33
#define PF3_TP 36          // pitch of the data-gradient waves' output tiles (floats): 16-byte rows

struct Pf3Args {
    const float* gy; const float* y; const double* gs; const double* gq; const double* gsc;
    const float* x; const unsigned* wsplit;          // wsplit: [3 terms][BN][BMP / 2] packed bf16 pairs (co, co + 1) of W^T (pf3_presplit_kernel)
    float* gx; double* gw;
    const float* acc; int acc_s, acc_Ho, acc_Wo, Hi, Wi, T;      // compact shortcut gradient of a stage-first block (null otherwise), added on its stride lattice
    int N, M, K, Q, nstrips, stages;
    int dbg;                                         // knock-outs (CFN_PWFS_DBG, 0 in the product): 1 weight gradient, 2 data gradient, 8 gx stores
};

template <int MT, int NT>
__global__ __launch_bounds__(256) void pf3_presplit_kernel(const float* w, int M, int K, unsigned* ws) {
    constexpr int BM = 32 * MT, BN = 32 * NT, BMP = BM + 8;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= BN * (BMP / 2)) return;
    const int ci = e / (BMP / 2), mp = e - ci * (BMP / 2), co = 2 * mp;
    const float w0 = (ci < K && co < M) ? w[(long)co * K + ci] : 0.0f;
    const float w1 = (ci < K && co + 1 < M) ? w[(long)(co + 1) * K + ci] : 0.0f;
    unsigned p[3];
    pfs_split3(w0, w1, p);
#pragma unroll
    for (int s = 0; s < 3; ++s) ws[(s * BN + ci) * (BMP / 2) + mp] = p[s];
}

template <int MT, int NT>
__global__ __launch_bounds__(512, 1) void pw_bwd_fused_split3_kernel(const Pf3Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BM = 32 * MT, BN = 32 * NT, BMP = BM + 8;
    constexpr int NG = BM / 32, NX = BN / 32;                    // float4 per staging thread and stage: 32 rows x 8 float4 per pass of 256 threads
    constexpr int NT16 = BN / 16, KS = BM / 32;
    constexpr int NR = (MT + 3) / 4;                              // row tiles per weight-gradient wave (i = w + 4 n)
    constexpr int NDW = NT16 / 2;                                 // data-gradient waves: two 16-channel tiles each (3 at 96 input channels, 2 at 48-64)
    static_assert(NT16 % 2 == 0 && NDW <= 3, "at most three data-gradient waves beside the staging wave");
    const int tid = threadIdx.x, wave = cfn_uni((int)(tid >> 6)), lane = tid & 63, half = lane >> 5, col = lane & 31;
    const int m16 = lane & 15, kq = lane >> 4;
    unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    const int strip = L % a.nstrips;
    const int n = L / a.nstrips;
    const int M = a.M, K = a.K, Q = a.Q;
    constexpr int IMG = (BM + BN) * PF3_PITCH;
    float* img0 = smem;
    constexpr int OOB = 0x7ffffff0;
    const int qbeg = strip * a.stages * PF3_PT;
    const int nst = min(a.stages, (Q - qbeg + PF3_PT - 1) / PF3_PT);
    const int lrow = (tid & 255) >> 3, c4 = (tid & 7) * 4;           // 8 lanes cover one 32-position row segment (128 bytes); a half wave = 4 rows x 8 segments: 32 banks

    if (wave < 4) {
        // ================= waves 0-3: stage the G' rows; weight gradient of the row tiles i = w, w + 4 (x all column tiles) over all 32 positions =================
        __amdgpu_buffer_rsrc_t rg = cfn_rsrc(const_cast<float*>(a.gy + (long)n * M * Q), (unsigned)((long)M * Q * 4));
        __amdgpu_buffer_rsrc_t ry = cfn_rsrc(const_cast<float*>((a.y ? a.y : a.gy) + (long)n * M * Q), a.y ? (unsigned)((long)M * Q * 4) : 0u);
        pf4 pg[NG], py[NG];
        int vog[NG];
        float rcs[NG], rcq[NG], rcz[NG];                             // (gs, 2 gq, gsc) of this thread's G' rows: registers, read once
#pragma unroll
        for (int it = 0; it < NG; ++it) {
            const int row = it * 32 + lrow;
            const bool ok = row < M;
            rcs[it] = (ok && a.gs) ? (float)a.gs[(long)n * M + row] : 0.0f;
            rcq[it] = (ok && a.gq && a.y) ? 2.0f * (float)a.gq[(long)n * M + row] : 0.0f;
            rcz[it] = (ok && a.gsc) ? (float)a.gsc[(long)n * M + row] : 1.0f;
            vog[it] = ok ? (row * Q + c4) * 4 : OOB;
        }
        auto prefetch = [&](int q0) {
            const bool inq = q0 + c4 < Q;
#pragma unroll
            for (int it = 0; it < NG; ++it) {
                const int vo = inq ? vog[it] : OOB;
                pg[it] = __builtin_bit_cast(pf4, __builtin_amdgcn_raw_buffer_load_b128(rg, vo, q0 * 4, 0));
                py[it] = __builtin_bit_cast(pf4, __builtin_amdgcn_raw_buffer_load_b128(ry, vo, q0 * 4, 0));
            }
        };
        auto stage = [&](int q0, float* sG) {
            const bool inq = q0 + c4 < Q;
#pragma unroll
            for (int it = 0; it < NG; ++it) {
                const int row = it * 32 + lrow;
                const float cs = rcs[it], cq = rcq[it], cz = rcz[it];
                const bool ok = inq && (row < M);
                float* d = sG + row * PF3_PITCH + c4;
                d[0] = ok ? fmaf(py[it].x, cq, fmaf(pg[it].x, cz, cs)) : 0.0f;
                d[1] = ok ? fmaf(py[it].y, cq, fmaf(pg[it].y, cz, cs)) : 0.0f;
                d[2] = ok ? fmaf(py[it].z, cq, fmaf(pg[it].z, cz, cs)) : 0.0f;
                d[3] = ok ? fmaf(py[it].w, cq, fmaf(pg[it].w, cz, cs)) : 0.0f;
            }
        };
        if (nst > 0) {
            prefetch(qbeg);
            stage(qbeg, img0);
            if (nst > 1) prefetch(qbeg + PF3_PT);
        }
        __syncthreads();
        f16v acc[NR][NT];
#pragma unroll
        for (int t = 0; t < NR; ++t)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.0f;
        for (int st = 0; st < nst; ++st) {
            const int q0 = qbeg + st * PF3_PT;
            float* cur = img0 + (st & 1) * IMG;
            float* nxt = img0 + ((st + 1) & 1) * IMG;
            if (st + 1 < nst) {
                stage(q0 + PF3_PT, nxt);
                if (st + 2 < nst) prefetch(q0 + 2 * PF3_PT);
            }
            const float* sG = cur;
            const float* sX = cur + BM * PF3_PITCH;
            if (!(a.dbg & 1))
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int p0 = 16 * kb + 8 * half;
                pfs_u4 Af[NR][3];
#pragma unroll
                for (int t = 0; t < NR; ++t) {
                    const int i = wave + 4 * t;
                    if (i < MT) {                                    // wave uniform
                        const float* r = sG + (i * 32 + col) * PF3_PITCH + p0;
#pragma unroll
                        for (int h = 0; h < 4; ++h) {
                            unsigned p[3];
                            pfs_split3(r[2 * h], r[2 * h + 1], p);
#pragma unroll
                            for (int s = 0; s < 3; ++s) Af[t][s][h] = p[s];
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    pfs_u4 Bf[3];
                    const float* r = sX + (j * 32 + col) * PF3_PITCH + p0;
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        unsigned p[3];
                        pfs_split3(r[2 * h], r[2 * h + 1], p);
#pragma unroll
                        for (int s = 0; s < 3; ++s) Bf[s][h] = p[s];
                    }
                    // term major: consecutive MFMAs of a wave go to DIFFERENT accumulator tiles (the six terms of one tile are a dependent chain)
#define PF3_WG(SA, SB)                                                                                                                         \
                    _Pragma("unroll") for (int t = 0; t < NR; ++t)                                                                              \
                        if (wave + 4 * t < MT)                                                                                                  \
                            acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pfs_bf8, Af[t][SA]), __builtin_bit_cast(pfs_bf8, Bf[SB]), acc[t][j], 0, 0, 0);
                    PFS_TERMS(PF3_WG)
#undef PF3_WG
                }
            }
            __syncthreads();
        }
        // every tile has ONE owner: straight from the accumulators, one fp64 atomic per element and workgroup
#pragma unroll
        for (int t = 0; t < NR; ++t) {
            const int i = wave + 4 * t;
            if (i < MT) {
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int gm = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, gk = j * 32 + col;
                        const float v = acc[t][j][r];
                        if (gm < M && gk < K) cfn_add64(&a.gw[(long)gm * K + gk], (double)v);
                    }
            }
        }
    } else if (wave < 4 + NDW) {
        // ================= waves 4 .. 4 + NDW - 1: data gradient.  Wave d owns the channel tiles 2 d, 2 d + 1 (input channels 32 d .. 32 d + 31) for ALL 32 positions of a stage, and keeps
        // its part of W^T -- 2 tiles x 7 k-steps x 3 terms, 168 registers -- RESIDENT: read once from the pre-split workspace, nothing but LDS reads in the stage loop =========
        const int d = wave - 4;
        __amdgpu_buffer_rsrc_t rd = cfn_rsrc(a.gx + (long)n * K * Q, (unsigned)((long)K * Q * 4));
        __amdgpu_buffer_rsrc_t rw = cfn_rsrc(const_cast<unsigned*>(a.wsplit), (unsigned)((size_t)3 * BN * (BMP / 2) * 4));
        const int wlane = ((m16 * BMP + 8 * kq) / 2) * 4;            // byte offset of this lane's (ci = m16, co = 8 kq) in a term image
        pfs_u4 Wr[2][KS][3];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    Wr[t][s][u] = __builtin_bit_cast(pfs_u4, __builtin_amdgcn_raw_buffer_load_b128(rw, wlane, (((u * BN + (2 * d + t) * 16) * BMP + 32 * s) / 2) * 4, 0));
        const int hw = a.Hi * a.Wi;
        const int acc_pitch4 = a.T * a.acc_Ho * a.acc_Wo * 4;       // bytes per channel of the compact gradient
        __amdgpu_buffer_rsrc_t racc = cfn_rsrc(const_cast<float*>(a.acc ? a.acc + (long)n * K * (acc_pitch4 / 4) : a.gy), a.acc ? (unsigned)((long)K * acc_pitch4) : 0u);
        __syncthreads();
        for (int st = 0; st < nst; ++st) {
            const int q0 = qbeg + st * PF3_PT;
            const float* sG = img0 + (st & 1) * IMG;
            float av[2][2][4];                                       // the compact shortcut gradient of this lane's outputs: loads up front
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) av[pb][t][r] = 0.0f;
            if (a.acc) {                                             // workgroup uniform
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) {
                    const int q = q0 + 16 * pb + m16;
                    int aoff = OOB;
                    if (q < Q) {
                        const int tq = q / hw, rq = q - tq * hw;
                        const int hq = rq / a.Wi, wq_ = rq - hq * a.Wi;
                        if (hq % a.acc_s == 0 && wq_ % a.acc_s == 0) aoff = (((tq * a.acc_Ho + hq / a.acc_s) * a.acc_Wo + wq_ / a.acc_s)) * 4;
                    }
                    const unsigned arow = (unsigned)aoff + (unsigned)(4 * kq) * (unsigned)acc_pitch4;
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            av[pb][t][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                racc, (int)(aoff == OOB ? (unsigned)OOB : arow + (unsigned)r * (unsigned)acc_pitch4), (2 * d + t) * 16 * acc_pitch4, 0));
                }
            }
            pf4 da[2][2];                                            // [position block][tile]
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                for (int t = 0; t < 2; ++t) da[pb][t] = (pf4){0.f, 0.f, 0.f, 0.f};
            if (!(a.dbg & 2))
#pragma unroll
            for (int s = 0; s < KS; ++s) {
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) {
                    const float* r = sG + (32 * s + 8 * kq) * PF3_PITCH + 16 * pb + m16;
                    pfs_u4 Gf[3];
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        unsigned p[3];
                        pfs_split3(r[(2 * h) * PF3_PITCH], r[(2 * h + 1) * PF3_PITCH], p);
#pragma unroll
                        for (int u = 0; u < 3; ++u) Gf[u][h] = p[u];
                    }
#define PF3_DG(SA, SB)                                                                                                                         \
                    _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                                               \
                        da[pb][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pfs_bf8, Wr[t][s][SA]), __builtin_bit_cast(pfs_bf8, Gf[SB]), da[pb][t], 0, 0, 0);
                    PFS_TERMS(PF3_DG)
#undef PF3_DG
                }
            }
            // the wave's 32 channels x 32 positions go through a wave-private LDS tile and leave as 16-byte stores: 8 lanes write one 128-byte line of a row
            // (the C layout gives a lane one position and four ROWS: stored directly that is 16 four-byte stores per lane into 64-byte row segments)
            float* tile = smem + 2 * IMG + d * (32 * PF3_TP);
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = da[pb][t][r] + av[pb][t][r];  // (through a scalar: __builtin_bit_cast of a vector ELEMENT took element 0 for every r)
                        tile[(t * 16 + 4 * kq + r) * PF3_TP + 16 * pb + m16] = v;
                    }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            {
                const int seg = lane & 7, r0 = lane >> 3;                    // 8 lanes per row, 8 rows per pass
                const bool inq = q0 + 4 * seg < Q && !(a.dbg & 8);           // (Q is a multiple of 4: a float4 is inside or outside as a whole)
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const int row = ps * 8 + r0;
                    const pf4 v = *reinterpret_cast<const pf4*>(tile + row * PF3_TP + 4 * seg);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pfs_u4, v), rd, inq ? (r0 * Q + 4 * seg) * 4 : OOB, (q0 + (32 * d + ps * 8) * Q) * 4, 0);
                }
            }
            __syncthreads();
        }
    } else if (wave < 7) {
        // (with two data-gradient waves one wave has nothing to do but keep the barrier count)
        __syncthreads();
        for (int st = 0; st < nst; ++st) __syncthreads();
    } else {
        // ================= wave 7: stages the x rows (8 rows x 8 float4 per pass) =================
        __amdgpu_buffer_rsrc_t rx = cfn_rsrc(const_cast<float*>(a.x + (long)n * K * Q), (unsigned)((long)K * Q * 4));
        constexpr int NP = BN / 8;
        const int xrow = lane >> 3;                                  // 8 lanes per row segment
        pf4 px[NP];
        int vox[NP];
#pragma unroll
        for (int it = 0; it < NP; ++it) vox[it] = (it * 8 + xrow) < K ? ((it * 8 + xrow) * Q + c4) * 4 : OOB;
        auto prefetch = [&](int q0) {
            const bool inq = q0 + c4 < Q;
#pragma unroll
            for (int it = 0; it < NP; ++it)
                px[it] = __builtin_bit_cast(pf4, __builtin_amdgcn_raw_buffer_load_b128(rx, inq ? vox[it] : OOB, q0 * 4, 0));
        };
        auto stage = [&](float* sX) {
#pragma unroll
            for (int it = 0; it < NP; ++it) {
                float* dd = sX + (it * 8 + xrow) * PF3_PITCH + c4;
                dd[0] = px[it].x; dd[1] = px[it].y; dd[2] = px[it].z; dd[3] = px[it].w;
            }
        };
        if (nst > 0) {
            prefetch(qbeg);
            stage(img0 + BM * PF3_PITCH);
            if (nst > 1) prefetch(qbeg + PF3_PT);
        }
        __syncthreads();
        for (int st = 0; st < nst; ++st) {
            const int q0 = qbeg + st * PF3_PT;
            float* nxt = img0 + ((st + 1) & 1) * IMG;
            if (st + 1 < nst) {
                stage(nxt + BM * PF3_PITCH);
                if (st + 2 < nst) prefetch(q0 + 2 * PF3_PT);
            }
            __syncthreads();
        }
    }
}

// ---- layer 3, conv3 (216 -> 96 with the BN2 + swish prologue in front): the same budget with the roles of the sides exchanged.  G' has 96 rows (3 row tiles), x 216 (7 column
// tiles = 14 channel tiles of the data gradient); W^T pre-split is [3][224][104] bf16.  Waves 0-3 stage everything (G': 3 passes x 2 tensors, x: 7 passes) and own the column
// tiles w, w + 4 x all three row tiles (every x element goes through the prologue once); waves 4-7 own the data gradient of channel tiles 4 d .. 4 d + 3 for all 32
// positions with THEIR part of W^T resident (4 tiles x 3 k-steps x 3 terms: 144 registers), the act' epilogue and the statistics of those channels (no cross-wave sum).
struct Pf3eArgs {
    const float* gy; const float* y; const double* gs; const double* gq; const double* gsc;
    const float* x; const double* pa; const double* pb; const unsigned* wsplit;
    float* gx; double* gA; double* gB; double* gw;
    int N, M, K, Q, nstrips, stages;
};

template <int MT, int NT, int ACT>
__global__ __launch_bounds__(512, 1) void pw_bwd_fused_split3e_kernel(const Pf3eArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BM = 32 * MT, BN = 32 * NT, BMP = BM + 8;
    constexpr int NG = BM / 32, NX = BN / 32;                    // float4 per staging thread and stage: 32 rows x 8 float4 per pass of 256 threads
    constexpr int NT16 = BN / 16, KS = BM / 32;
    constexpr int NC = (NT + 3) / 4;                              // column tiles per weight-gradient wave (j = w + 4 n)
    constexpr int ND = 4;                                         // channel tiles per data-gradient wave
    static_assert(NT16 <= 4 * ND, "four data-gradient waves x four channel tiles");
    const int tid = threadIdx.x, wave = cfn_uni((int)(tid >> 6)), lane = tid & 63, half = lane >> 5, col = lane & 31;
    const int m16 = lane & 15, kq = lane >> 4;
    unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    const int strip = L % a.nstrips;
    const int n = L / a.nstrips;
    const int M = a.M, K = a.K, Q = a.Q;
    constexpr int IMG = (BM + BN) * PF3_PITCH;
    float* img0 = smem;
    float* sCx = smem + 2 * IMG;                                  // [BN][2] (A, B) of the epilogue
    constexpr int OOB = 0x7ffffff0;
    const int qbeg = strip * a.stages * PF3_PT;
    const int nst = min(a.stages, (Q - qbeg + PF3_PT - 1) / PF3_PT);
    for (int k = tid; k < BN; k += 512) {
        const bool ok = k < K;
        sCx[2 * k] = ok ? (float)a.pa[(long)n * K + k] : 1.0f;
        sCx[2 * k + 1] = ok ? (float)a.pb[(long)n * K + k] : 0.0f;
    }

    if (wave < 4) {
        // ================= waves 0-3: stage G' and x; weight gradient of the column tiles j = w, w + 4 (x all row tiles) over all 32 positions =================
        const int lrow = tid >> 3, c4 = (tid & 7) * 4;
        __amdgpu_buffer_rsrc_t rg = cfn_rsrc(const_cast<float*>(a.gy + (long)n * M * Q), (unsigned)((long)M * Q * 4));
        __amdgpu_buffer_rsrc_t ry = cfn_rsrc(const_cast<float*>((a.y ? a.y : a.gy) + (long)n * M * Q), a.y ? (unsigned)((long)M * Q * 4) : 0u);
        __amdgpu_buffer_rsrc_t rx = cfn_rsrc(const_cast<float*>(a.x + (long)n * K * Q), (unsigned)((long)K * Q * 4));
        pf4 pg[NG], py[NG], px[NX];
        int vog[NG], vox[NX];
        float rcs[NG], rcq[NG], rcz[NG];                             // (gs, 2 gq, gsc) of this thread's G' rows: registers
#pragma unroll
        for (int it = 0; it < NG; ++it) {
            const int row = it * 32 + lrow;
            const bool ok = row < M;
            rcs[it] = (ok && a.gs) ? (float)a.gs[(long)n * M + row] : 0.0f;
            rcq[it] = (ok && a.gq && a.y) ? 2.0f * (float)a.gq[(long)n * M + row] : 0.0f;
            rcz[it] = (ok && a.gsc) ? (float)a.gsc[(long)n * M + row] : 1.0f;
            vog[it] = ok ? (row * Q + c4) * 4 : OOB;
        }
#pragma unroll
        for (int it = 0; it < NX; ++it) vox[it] = (it * 32 + lrow) < K ? ((it * 32 + lrow) * Q + c4) * 4 : OOB;
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
                const int row = it * 32 + lrow;
                const float cs = rcs[it], cq = rcq[it], cz = rcz[it];
                const bool ok = inq && (row < M);
                float* d = sG + row * PF3_PITCH + c4;
                d[0] = ok ? fmaf(py[it].x, cq, fmaf(pg[it].x, cz, cs)) : 0.0f;
                d[1] = ok ? fmaf(py[it].y, cq, fmaf(pg[it].y, cz, cs)) : 0.0f;
                d[2] = ok ? fmaf(py[it].z, cq, fmaf(pg[it].z, cz, cs)) : 0.0f;
                d[3] = ok ? fmaf(py[it].w, cq, fmaf(pg[it].w, cz, cs)) : 0.0f;
            }
#pragma unroll
            for (int it = 0; it < NX; ++it) {
                float* d = sX + (it * 32 + lrow) * PF3_PITCH + c4;
                d[0] = px[it].x; d[1] = px[it].y; d[2] = px[it].z; d[3] = px[it].w;
            }
        };
        if (nst > 0) {
            prefetch(qbeg);
            stage(qbeg, img0, img0 + BM * PF3_PITCH);
            if (nst > 1) prefetch(qbeg + PF3_PT);
        }
        float ca[NC], cb[NC];                                        // prologue coefficients of this lane's x rows (row (w + 4 c) * 32 + col)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int k = (wave + 4 * c) * 32 + col;
            const bool ok = k < K;
            ca[c] = ok ? (float)a.pa[(long)n * K + k] : 1.0f;
            cb[c] = ok ? (float)a.pb[(long)n * K + k] : 0.0f;
        }
        __syncthreads();
        f16v acc[NC][MT];
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][i][r] = 0.0f;
        for (int st = 0; st < nst; ++st) {
            const int q0 = qbeg + st * PF3_PT;
            float* cur = img0 + (st & 1) * IMG;
            float* nxt = img0 + ((st + 1) & 1) * IMG;
            if (st + 1 < nst) {
                stage(q0 + PF3_PT, nxt, nxt + BM * PF3_PITCH);
                if (st + 2 < nst) prefetch(q0 + 2 * PF3_PT);
            }
            const float* sG = cur;
            const float* sX = cur + BM * PF3_PITCH;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int p0 = 16 * kb + 8 * half;
                pfs_u4 Bf[NC][3];
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const int j = wave + 4 * c;
                    if (j < NT) {                                    // wave uniform
                        const float* r = sX + (j * 32 + col) * PF3_PITCH + p0;
#pragma unroll
                        for (int h = 0; h < 4; ++h) {
                            const float x0 = cfn_act<ACT>(fmaf(r[2 * h], ca[c], cb[c])), x1 = cfn_act<ACT>(fmaf(r[2 * h + 1], ca[c], cb[c]));
                            unsigned p[3];
                            pfs_split3(x0, x1, p);
#pragma unroll
                            for (int s = 0; s < 3; ++s) Bf[c][s][h] = p[s];
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    pfs_u4 Af[3];
                    const float* r = sG + (i * 32 + col) * PF3_PITCH + p0;
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        unsigned p[3];
                        pfs_split3(r[2 * h], r[2 * h + 1], p);
#pragma unroll
                        for (int s = 0; s < 3; ++s) Af[s][h] = p[s];
                    }
#define PF3E_WG(SA, SB)                                                                                                                        \
                    _Pragma("unroll") for (int c = 0; c < NC; ++c)                                                                              \
                        if (wave + 4 * c < NT)                                                                                                  \
                            acc[c][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pfs_bf8, Af[SA]), __builtin_bit_cast(pfs_bf8, Bf[c][SB]), acc[c][i], 0, 0, 0);
                    PFS_TERMS(PF3E_WG)
#undef PF3E_WG
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int j = wave + 4 * c;
            if (j < NT) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int gm = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, gk = j * 32 + col;
                        const float v = acc[c][i][r];
                        if (gm < M && gk < K) cfn_add64(&a.gw[(long)gm * K + gk], (double)v);
                    }
            }
        }
    } else {
        // ================= waves 4-7: data gradient of the channel tiles 4 d .. 4 d + 3 for all 32 positions; W^T resident; act' epilogue; statistics =================
        const int d = wave - 4;
        const int tb = ND * d, tn = ND;                               // channel tiles 4 d .. 4 d + 3 (the last wave's tiles 14, 15 do not exist: zero weights, dropped stores)
        __amdgpu_buffer_rsrc_t rd = cfn_rsrc(a.gx + (long)n * K * Q, (unsigned)((long)K * Q * 4));
        __amdgpu_buffer_rsrc_t rw = cfn_rsrc(const_cast<unsigned*>(a.wsplit), (unsigned)((size_t)3 * BN * (BMP / 2) * 4));
        const int wlane = ((m16 * BMP + 8 * kq) / 2) * 4;
        pfs_u4 Wr[ND][KS][3];
#pragma unroll
        for (int t = 0; t < ND; ++t)
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    Wr[t][s][u] = (t < tn && tb + t < NT16) ? __builtin_bit_cast(pfs_u4, __builtin_amdgcn_raw_buffer_load_b128(rw, wlane, cfn_uni((((u * BN + (tb + t) * 16) * BMP + 32 * s) / 2) * 4), 0))
                                                      : (pfs_u4){0u, 0u, 0u, 0u};
        float sa[ND][4], sb[ND][4];                                  // per-lane partial statistics of rows (4 d + t) * 16 + 4 kq + r
#pragma unroll
        for (int t = 0; t < ND; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { sa[t][r] = 0.0f; sb[t][r] = 0.0f; }
        __syncthreads();
        for (int st = 0; st < nst; ++st) {
            const int q0 = qbeg + st * PF3_PT;
            const float* sG = img0 + (st & 1) * IMG;
            const float* sX = sG + BM * PF3_PITCH;
            pf4 da[2][ND];
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                for (int t = 0; t < ND; ++t) da[pb][t] = (pf4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) {
                    const float* r = sG + (32 * s + 8 * kq) * PF3_PITCH + 16 * pb + m16;
                    pfs_u4 Gf[3];
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        unsigned p[3];
                        pfs_split3(r[(2 * h) * PF3_PITCH], r[(2 * h + 1) * PF3_PITCH], p);
#pragma unroll
                        for (int u = 0; u < 3; ++u) Gf[u][h] = p[u];
                    }
#define PF3E_DG(SA, SB)                                                                                                                        \
                    _Pragma("unroll") for (int t = 0; t < ND; ++t)                                                                              \
                        da[pb][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pfs_bf8, Wr[t][s][SA]), __builtin_bit_cast(pfs_bf8, Gf[SB]), da[pb][t], 0, 0, 0);
                    PFS_TERMS(PF3E_DG)
#undef PF3E_DG
                }
            }
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                const int q = q0 + 16 * pb + m16;
                const int gvo = q < Q ? (4 * kq * Q + 16 * pb + m16) * 4 : OOB;
#pragma unroll
                for (int t = 0; t < ND; ++t) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ci = (tb + t) * 16 + 4 * kq + r;
                        const float v = da[pb][t][r];                // exactly 0 for positions >= Q and rows >= K
                        const float xr = sX[ci * PF3_PITCH + 16 * pb + m16];
                        const float2 pab = cfn_settle(*reinterpret_cast<const float2*>(sCx + 2 * ci));
                        const float dz = v * cfn_act_grad<ACT>(fmaf(xr, pab.x, pab.y));
                        sa[t][r] = fmaf(dz, xr, sa[t][r]);
                        sb[t][r] += dz;
                        const float o = dz * pab.x;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rd, gvo, (q0 + ((tb + t) * 16 + r) * Q) * 4, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();
        }
        // every channel has ONE owner wave: reduce over the 16 position lanes, one fp64 atomic per channel and workgroup
#pragma unroll
        for (int t = 0; t < ND; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float u = sa[t][r], v = sb[t][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { u += __shfl_xor(u, o, 64); v += __shfl_xor(v, o, 64); }
                const int ci = (tb + t) * 16 + 4 * kq + r;
                if (m16 == 0 && t < tn && ci < K) {
                    cfn_add64(&a.gA[(long)n * K + ci], (double)u);
                    cfn_add64(&a.gB[(long)n * K + ci], (double)v);
                }
            }
    }
}

static unsigned* pf3_workspace(size_t bytes, hipStream_t st) {
    // one grow-only buffer per (device, stream), never freed (a captured graph may hold the address); none is allocated while the stream is being captured
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, std::pair<unsigned*, size_t>> bufs;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    std::lock_guard<std::mutex> lk(mu);
    auto& b = bufs[std::make_pair(dev, st)];
    if (b.second >= bytes) return b.first;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;
    unsigned* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    b = {p, bytes};
    return p;
}

template <int NT>
static int pf3_launch(const float* gy, const float* y, const double* gsum, const double* gsumsq, const float* w, const float* x, float* gx, double* gw,
                      int N, int Cin, int Cout, int T, int Hi, int Wi, const float* acc, int acc_stride, const double* gscale, hipStream_t st) {
    constexpr int MT = 7, BM = 32 * MT, BN = 32 * NT, BMP = BM + 8;
    const long Ql = (long)T * Hi * Wi;
    const size_t wbytes = (size_t)3 * BN * (BMP / 2) * 4;
    unsigned* ws = pf3_workspace(wbytes, st);
    if (!ws) return -1;
    hipLaunchKernelGGL((pf3_presplit_kernel<MT, NT>), dim3((BN * (BMP / 2) + 255) / 256), dim3(256), 0, st, w, Cout, Cin, ws);
    Pf3Args a = {};
    a.gy = gy; a.y = gsumsq ? y : nullptr; a.gs = gsum; a.gq = gsumsq; a.gsc = gscale; a.x = x; a.wsplit = ws; a.gx = gx; a.gw = gw;
    a.acc = acc; a.acc_s = acc ? acc_stride : 1; a.Hi = Hi; a.Wi = Wi; a.T = T;
    a.acc_Ho = (Hi - 1) / a.acc_s + 1; a.acc_Wo = (Wi - 1) / a.acc_s + 1;
    a.N = N; a.M = Cout; a.K = Cin; a.Q = (int)Ql;
    { const char* e = getenv("CFN_PWFS_DBG"); a.dbg = e ? atoi(e) : 0; }
    const long nst = cfn_cdiv(Ql, PF3_PT);
    // whole rounds of the chip (one workgroup per CU is resident) and few of them: every workgroup pays the W^T load and M x K fp64 atomics once; measured at 96 -> 216,
    // 8 clips x 256 frames x 14 x 14: 256 / 512 / 1024 workgroups = 0.301 / 0.333 / 0.395 ms, 320 / 384 (partial rounds) 0.416 / 0.376.  One round; two where a workgroup
    // would otherwise walk more than ~100 stages (the first block of layer 3 at 28 x 28)
    static const int wgs_env = getenv("CFN_PWF3_WGS") ? atoi(getenv("CFN_PWF3_WGS")) : 0;
    const long wgs = wgs_env > 0 ? wgs_env : ((long)N * nst > 256L * 100 ? 512 : 256);
    long want = wgs / N;
    if (want < 1) want = 1;
    long stages = cfn_cdiv(nst, want);
    if (stages < 4) stages = 4;
    a.stages = (int)stages;
    a.nstrips = (int)cfn_cdiv(nst, stages);
    const unsigned blocks = (unsigned)((long)N * a.nstrips);
    const size_t lds = ((size_t)2 * (BM + BN) * PF3_PITCH + 3 * 32 * PF3_TP) * sizeof(float);      // two images + the data-gradient waves' output tiles
    CfnProfScope prof(CFN_K_PWCONV_BWD, st, 4.0 * N * ((double)Cout * Ql * (a.y ? 2 : 1) + (double)Cin * Ql * 2));
    auto k = pw_bwd_fused_split3_kernel<MT, NT>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), lds, st, a);
    return cfn_check_launch("pwconv_bwd_fused (split bf16, layer 3)");
}

static int pf3e_try_launch(const float* gy, const float* y, const double* gsum, const double* gsumsq, const float* w, const float* x, const double* A, const double* B,
                           int act, float* gx, double* gA, double* gB, double* gw, int N, int Cin, int Cout, long Ql, const double* gscale, hipStream_t st) {
    constexpr int MT = 3, NT = 7, BM = 32 * MT, BN = 32 * NT, BMP = BM + 8;
    const size_t wbytes = (size_t)3 * BN * (BMP / 2) * 4;
    unsigned* ws = pf3_workspace(wbytes, st);
    if (!ws) return -1;
    hipLaunchKernelGGL((pf3_presplit_kernel<MT, NT>), dim3((BN * (BMP / 2) + 255) / 256), dim3(256), 0, st, w, Cout, Cin, ws);
    Pf3eArgs a = {};
    a.gy = gy; a.y = gsumsq ? y : nullptr; a.gs = gsum; a.gq = gsumsq; a.gsc = gscale; a.x = x; a.pa = A; a.pb = B; a.wsplit = ws;
    a.gx = gx; a.gA = gA; a.gB = gB; a.gw = gw;
    a.N = N; a.M = Cout; a.K = Cin; a.Q = (int)Ql;
    const long nst = cfn_cdiv(Ql, PF3_PT);
    static const int wgs = getenv("CFN_PWF3_WGS") ? atoi(getenv("CFN_PWF3_WGS")) : 256;
    long want = wgs / N;
    if (want < 1) want = 1;
    long stages = cfn_cdiv(nst, want);
    if (stages < 4) stages = 4;
    a.stages = (int)stages;
    a.nstrips = (int)cfn_cdiv(nst, stages);
    const unsigned blocks = (unsigned)((long)N * a.nstrips);
    const size_t lds = ((size_t)2 * (BM + BN) * PF3_PITCH + 2 * BN) * sizeof(float);
    CfnProfScope prof(CFN_K_PWCONV_BWD, st, 4.0 * N * ((double)Cout * Ql * (a.y ? 2 : 1) + (double)Cin * Ql * 2));
#define CFN_PF3E_GO(ACTV)                                                                                       \
    do {                                                                                                        \
        auto k = pw_bwd_fused_split3e_kernel<MT, NT, ACTV>;                                                     \
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);        \
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), lds, st, a);                                             \
    } while (0)
    if (act == CFN_ACT_RELU) CFN_PF3E_GO(CFN_ACT_RELU);
    else if (act == CFN_ACT_SWISH) CFN_PF3E_GO(CFN_ACT_SWISH);
    else CFN_PF3E_GO(CFN_ACT_NONE);
#undef CFN_PF3E_GO
    return cfn_check_launch("pwconv_bwd_fused (split bf16, layer 3 conv3)");
}

// -1 = not handled (cfn_pwconv_bwd_fused goes on to its fp32 kernel / declines)
int pwfs_try_launch(const float* gy, const float* y, const double* gsum, const double* gsumsq, const float* w, const float* x, const double* A,
                    const double* B, int act, float* gx, double* gA, double* gB, double* gw, int N, int Cin, int Cout, int T, int Hi, int Wi,
                    const float* acc, int acc_stride, const double* gscale, hipStream_t st) {
    const char* env_on = getenv("CFN_PWF_SPLIT");      // read per call: tests and A/B harnesses switch it inside one process
    const int on = env_on ? atoi(env_on) : 1;          // default 1: the no-prologue shapes (measured in the step: -1.3 .. -1.9 ms; level 2 loses 0.5 ms of it)
    if (!on || pws_terms_now() != 6) return -1;
    // CFN_PWF_SPLIT: 1 = the shapes WITHOUT a prologue (conv1 of the layer-2 blocks, whose input is a materialised block output), 2 = also the
    // shapes with one (conv3: BN2 + swish in front)
    const bool wide_m = Cout > 64 && Cout <= 128 && Cin > 32 && Cin <= 64;          // conv1 of layer 2: 48 -> 108
    const bool thin_k = Cout > 64 && Cout <= 128 && Cin >= 16 && Cin <= 32;         // conv1 of the first block of layer 2: 24 -> 108
    const bool wide_k = Cin > 64 && Cin <= 128 && Cout > 32 && Cout <= 64;          // conv3 of layer 2: 108 -> 48
    // layer-1 widths (24 <-> 54), served by the fp32 kernel of pwfused.hip unless CFN_PWF_SPLIT >= 3 (measurement switch)
    const bool l1_m = on >= 3 && Cout > 32 && Cout <= 64 && Cin >= 16 && Cin <= 32;      // conv1 of layer 1: 24 -> 54
    const bool l1_k = on >= 3 && Cin > 32 && Cin <= 64 && Cout >= 16 && Cout <= 32;      // conv3 of layer 1: 54 -> 24
    // layer 3's conv1 (96 -> 216, no prologue, no shortcut gradient): the variant with W^T resident in the data-gradient waves' registers (CFN_PWF_L3=0 switches it off alone)
    static const int l3_on = getenv("CFN_PWF_L3") ? atoi(getenv("CFN_PWF_L3")) : 1;
    if (on >= 1 && l3_on && A == nullptr && Cout > 192 && Cout <= 224 && Cin > 32 && Cin <= 96) {
        const long Ql3 = (long)T * Hi * Wi;
        if (Ql3 % 4 == 0 && Ql3 < (1L << 30) && (long)Cout * Ql3 * 4 < 0x7ffffff0L && ((((uintptr_t)gy | (uintptr_t)x | (uintptr_t)(y ? y : gy)) & 15) == 0) &&
            (acc == nullptr || acc_stride >= 1)) {
            if (Cin > 64) return pf3_launch<3>(gy, y, gsum, gsumsq, w, x, gx, gw, N, Cin, Cout, T, Hi, Wi, acc, acc_stride, gscale, st);       // 96 -> 216: the blocks of layer 3
            return pf3_launch<2>(gy, y, gsum, gsumsq, w, x, gx, gw, N, Cin, Cout, T, Hi, Wi, acc, acc_stride, gscale, st);                      // 48 -> 216 @28: its first block
        }
    }
    // layer 3's conv3 (216 -> 96 behind BN2 + swish): the same structure with the sides exchanged (CFN_PWF_L3E: 0 off; while it is being measured: off unless set)
    const char* env_l3e = getenv("CFN_PWF_L3E");          // read per call (tests / A-B harnesses switch it inside one process)
    const int l3e_on = env_l3e ? atoi(env_l3e) : 0;
    if (on >= 1 && l3e_on && A != nullptr && acc == nullptr && Cout > 64 && Cout <= 96 && Cin > 192 && Cin <= 224) {
        const long Ql3 = (long)T * Hi * Wi;
        if (Ql3 % 4 == 0 && Ql3 < (1L << 30) && (long)Cin * Ql3 * 4 < 0x7ffffff0L && ((((uintptr_t)gy | (uintptr_t)x | (uintptr_t)(y ? y : gy)) & 15) == 0) &&
            (act == CFN_ACT_NONE || act == CFN_ACT_RELU || act == CFN_ACT_SWISH))
            return pf3e_try_launch(gy, y, gsum, gsumsq, w, x, A, B, act, gx, gA, gB, gw, N, Cin, Cout, Ql3, gscale, st);
    }
    if (!wide_m && !wide_k && !thin_k && !l1_m && !l1_k) return -1;
    if (A != nullptr && on < 2) return -1;
    const long Ql = (long)T * Hi * Wi;
    if (Ql % 4 != 0 || Ql >= (1L << 30)) return -1;
    if ((long)Cout * Ql * 4 >= 0x7ffffff0L || (long)Cin * Ql * 4 >= 0x7ffffff0L) return -1;   // 32-bit buffer offsets per sample
    if (A && act != CFN_ACT_NONE && act != CFN_ACT_RELU && act != CFN_ACT_SWISH) return -1;
    if ((((uintptr_t)gy | (uintptr_t)x | (uintptr_t)(y ? y : gy)) & 15) != 0) return -1;
    PfsArgs a = {};
    a.gy = gy; a.y = gsumsq ? y : nullptr; a.gs = gsum; a.gq = gsumsq; a.gsc = gscale; a.w = w; a.x = x; a.pa = A; a.pb = B;
    a.gx = gx; a.gA = gA; a.gB = gB; a.gw = gw;
    a.acc = acc; a.acc_s = acc ? acc_stride : 1; a.Hi = Hi; a.Wi = Wi; a.T = T;
    a.acc_Ho = (Hi - 1) / a.acc_s + 1; a.acc_Wo = (Wi - 1) / a.acc_s + 1;
    a.N = N; a.M = Cout; a.K = Cin; a.Q = (int)Ql;
    { const char* e = getenv("CFN_PWFS_DBG"); a.dbg = e ? atoi(e) : 0; }
    const long nst = cfn_cdiv(Ql, PFS_PT);
    // one workgroup per CU is resident: whole rounds of the chip, and few of them (a workgroup splits W^T once and ends with M x K fp64 atomics): measured, 48 -> 108 at
    // 8 clips x 256 frames: 256 / 512 / 768 / 1024 / 2048 workgroups = 0.495 / 0.471-0.481 / 0.491 / 0.504 / 0.553 ms; 320 (1.25 rounds) 0.619
    static const int wgs = getenv("CFN_PWFS_WGS") ? atoi(getenv("CFN_PWFS_WGS")) : 512;
    long want = wgs / N;
    if (want < 1) want = 1;
    long stages = cfn_cdiv(nst, want);
    if (stages < 4) stages = 4;
    a.stages = (int)stages;
    a.nstrips = (int)cfn_cdiv(nst, stages);
    const unsigned blocks = (unsigned)((long)N * a.nstrips);
    CfnProfScope prof(CFN_K_PWCONV_BWD, st, 4.0 * N * ((double)Cout * Ql * (a.y ? 2 : 1) + (double)Cin * Ql * 2));
    const bool epi = A != nullptr;
    if (wide_m) return pfs_launch<4, 2>(a, act, epi, blocks, st);
    if (thin_k) return pfs_launch<4, 1>(a, act, epi, blocks, st);
    if (l1_m) return pfs_launch<2, 1>(a, act, epi, blocks, st);
    if (l1_k) return pfs_launch<1, 2>(a, act, epi, blocks, st);
    return pfs_launch<2, 4>(a, act, epi, blocks, st);
}
