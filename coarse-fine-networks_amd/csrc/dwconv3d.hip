// Depthwise 3x3x3 convolution (X3D bottleneck conv2; x3d_fine.py:89-97, used at :117) for gfx950.
//
// One streaming skeleton, three modes:
//   DW_FWD    y = dwconv( act(A*x+B) ) , epilogue per-(n,c) sum / sum-of-squares of y
//   DW_DGRAD  (stride 1) da = dwconv_flipped( gy + gs + 2*y*gq ), gx = da*act'(A*x+B)*A,
//             epilogue per-(n,c) sum(dz*x), sum(dz)
//   DW_WGRAD  gw[c][27] += sum over outputs of (gy + gs + 2*y*gq) * act(A*x+B)[tap]
//
// Data layout: NCDHW fp32, each (n,c) volume is T contiguous H*W planes.
// A workgroup owns CG channels x one band of output rows x one chunk of TT output frames and
// marches along t: every input frame is read from HBM once (coalesced float4 over the
// contiguous plane), transformed by the load-time prologue, staged in LDS with a zero halo, and
// consumed by every thread for HS vertically adjacent outputs at one column (lanes run along w
// => conflict-free LDS reads, coalesced stores).  Three rolling accumulator sets carry the
// temporal taps, so each output frame is written once.  The next frame's global loads are in
// flight while the current one is computed (register prefetch, single LDS buffer).
// blockIdx is remapped so that the t-chunks / bands of one (n, channel group) run on one XCD and
// find their halo frames in that XCD's L2.
#include "cfn_common.h"
#include "h16.h"

// Element type of the activation tensors.  This file is compiled twice: as is (fp32 storage) and through
// dwconv3d_bf16.hip (#define DW_BF16: bf16 storage, identical fp32 arithmetic, entry points suffixed _bf16, argument
// structs renamed so that the kernel symbols differ).  Every global-memory access of a tensor goes through the helpers
// below; byte offsets use DW_ES.
#ifdef DW_BF16
typedef unsigned short dwe_t;
#define DW_ES 2
#define DWN(name) H16N(name)
#else
typedef float dwe_t;
#define DW_ES 4
#define DWN(name) name
#endif
typedef float __attribute__((ext_vector_type(4))) dw_f4;
typedef float __attribute__((ext_vector_type(2))) dw_f2;
#ifdef DW_BF16
__device__ __forceinline__ float dw_lo(unsigned u) { return h16_lo(u); }
__device__ __forceinline__ float dw_hi(unsigned u) { return h16_hi(u); }
__device__ __forceinline__ unsigned dw_pk(float a, float b) { return h16_pk(a, b); }
__device__ __forceinline__ float dw_ld(const dwe_t* p) { return dw_lo(*p); }
__device__ __forceinline__ void dw_st(dwe_t* p, float v) { *p = (unsigned short)(dw_pk(v, 0.0f) & 0xffffu); }
__device__ __forceinline__ void dw_st2(dwe_t* p, dw_f2 v) { *reinterpret_cast<unsigned*>(p) = dw_pk(v.x, v.y); }
__device__ __forceinline__ dw_f2 dw_ld2(const dwe_t* p) { const unsigned u = *reinterpret_cast<const unsigned*>(p); return (dw_f2){dw_lo(u), dw_hi(u)}; }
__device__ __forceinline__ dw_f4 dw_ld4(const dwe_t* p) { const uint2 u = *reinterpret_cast<const uint2*>(p); return (dw_f4){dw_lo(u.x), dw_hi(u.x), dw_lo(u.y), dw_hi(u.y)}; }
__device__ __forceinline__ float dw_bld1(__amdgpu_buffer_rsrc_t r, int vo, int so) { return dw_lo((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, vo, so, 0)); }
__device__ __forceinline__ dw_f2 dw_bld2(__amdgpu_buffer_rsrc_t r, int vo, int so) { const unsigned u = __builtin_amdgcn_raw_buffer_load_b32(r, vo, so, 0); return (dw_f2){dw_lo(u), dw_hi(u)}; }
__device__ __forceinline__ dw_f4 dw_bld4(__amdgpu_buffer_rsrc_t r, int vo, int so) {
    typedef unsigned __attribute__((ext_vector_type(2))) u2;
    const u2 u = __builtin_bit_cast(u2, __builtin_amdgcn_raw_buffer_load_b64(r, vo, so, 0));
    return (dw_f4){dw_lo(u.x), dw_hi(u.x), dw_lo(u.y), dw_hi(u.y)};
}
__device__ __forceinline__ void dw_bst1(float v, __amdgpu_buffer_rsrc_t r, int vo, int so) { __builtin_amdgcn_raw_buffer_store_b16((short)(dw_pk(v, 0.0f) & 0xffffu), r, vo, so, 0); }
__device__ __forceinline__ void dw_bst2(dw_f2 v, __amdgpu_buffer_rsrc_t r, int vo, int so) { __builtin_amdgcn_raw_buffer_store_b32(dw_pk(v.x, v.y), r, vo, so, 0); }
// value as the consumer will read it back (statistics are taken over stored values)
__device__ __forceinline__ float dw_rt(float v) { return dw_lo(dw_pk(v, 0.0f)); }
#else
__device__ __forceinline__ float dw_ld(const dwe_t* p) { return *p; }
__device__ __forceinline__ void dw_st(dwe_t* p, float v) { *p = v; }
__device__ __forceinline__ void dw_st2(dwe_t* p, dw_f2 v) { *reinterpret_cast<dw_f2*>(p) = v; }
__device__ __forceinline__ dw_f2 dw_ld2(const dwe_t* p) { return *reinterpret_cast<const dw_f2*>(p); }
__device__ __forceinline__ dw_f4 dw_ld4(const dwe_t* p) { return *reinterpret_cast<const dw_f4*>(p); }
__device__ __forceinline__ float dw_bld1(__amdgpu_buffer_rsrc_t r, int vo, int so) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, vo, so, 0)); }
__device__ __forceinline__ dw_f2 dw_bld2(__amdgpu_buffer_rsrc_t r, int vo, int so) { return __builtin_bit_cast(dw_f2, __builtin_amdgcn_raw_buffer_load_b64(r, vo, so, 0)); }
__device__ __forceinline__ dw_f4 dw_bld4(__amdgpu_buffer_rsrc_t r, int vo, int so) { return __builtin_bit_cast(dw_f4, __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0)); }
__device__ __forceinline__ void dw_bst1(float v, __amdgpu_buffer_rsrc_t r, int vo, int so) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, vo, so, 0); }
__device__ __forceinline__ void dw_bst2(dw_f2 v, __amdgpu_buffer_rsrc_t r, int vo, int so) {
    typedef int __attribute__((ext_vector_type(2))) i2v;
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i2v, v), r, vo, so, 0);
}
__device__ __forceinline__ float dw_rt(float v) { return v; }
#endif

enum { DW_FWD = 0, DW_DGRAD = 1, DW_WGRAD = 2 };
int DWN(dw_small_fwd_try)(const dwe_t* x, const double* A, const double* B, int act, const float* w, dwe_t* y, double* sum, double* sumsq,
                          int N, int C, int T, int Hi, int Wi, int stride, hipStream_t st, bool probe);   // dwsmall.hip (both element types)
int DWN(dw_flatb_try)(const dwe_t* gy, const dwe_t* y, const double* gs, const double* gq, const float* w, const dwe_t* x,
                      const double* A, const double* B, int act, dwe_t* gx, double* gA, double* gB, double* gw,
                      int N, int C, int T, int H, int W, hipStream_t st, bool probe);                  // dwflatb.hip (both element types)
int DWN(dw_cpbx_try)(const dwe_t* gy, const dwe_t* y, const double* gs, const double* gq, const float* w, const dwe_t* x,
                     const double* A, const double* B, int act, dwe_t* gx, double* gA, double* gB, double* gw,
                     int N, int C, int T, int H, int W, hipStream_t st, bool probe);                   // dwcpbx.hip (both element types)
#ifndef DW_BF16
int dw_flat_fwd_try(const float* x, const double* A, const double* B, int act, const float* w, float* y, double* sum, double* sumsq,
                    int N, int C, int T, int Hi, int Wi, int stride, hipStream_t st, bool probe);    // dwflat.hip
#endif
// column-pair wave kernels (dwcp.hip, dwcpb.hip; compiled for both element types like this file: cp_io.h)
int DWN(dw_cp_fwd_try)(const dwe_t* x, const double* A, const double* B, int act, const float* w, dwe_t* y, double* sum, double* sumsq,
                       int N, int C, int T, int Hi, int Wi, int stride, hipStream_t st, bool probe);
int DWN(dw_cpb_try)(const dwe_t* gy, const dwe_t* y, const double* gs, const double* gq, const float* w, const dwe_t* x,
                    const double* A, const double* B, int act, dwe_t* gx, double* gA, double* gB, double* gw,
                    int N, int C, int T, int H, int W, hipStream_t st, bool probe);

struct DwArgs {
    const dwe_t* src;    // FWD/WGRAD: x raw (N,C,T,Hi,Wi)      DGRAD: gy (N,C,T,H,W)
    const dwe_t* src2;   // DGRAD: y (raw conv output) for the 2*y*gq term, may be null
    const double* A;      // per-(n,c) prologue scale of the forward input (null = identity)
    const double* B;
    const double* gs;    // DGRAD/WGRAD: d loss / d sum(y)   per (n,c), may be null
    const double* gq;    // DGRAD/WGRAD: d loss / d sum(y^2) per (n,c), may be null
    const float* w;      // (C,27)
    dwe_t* dst;          // FWD: y     DGRAD: gx
    const dwe_t* xin;    // DGRAD: forward input x raw (for act' and the A/B gradients)
    const dwe_t* gy;     // WGRAD: upstream gradient at output resolution
    const dwe_t* yout;   // WGRAD: raw conv output (for the 2*y*gq term), may be null
    double* s1;          // FWD: sum(y)      DGRAD: sum(dz*x)     WGRAD: gw (C,27) accumulators
    double* s2;          // FWD: sum(y*y)    DGRAD: sum(dz)
    int N, C, T, Hi, Wi, Ho, Wo, act;
    int TT, nchunks, CG, ngroups, GB, nbands, IPCb, IPCp, RIN, WP, XO;   // IPCp: thread slots per channel (>= IPCb)
};

// contiguous-segment sum inside a wave: lanes with equal key form runs; the first lane of each
// run ends up with the run total.
__device__ __forceinline__ float seg_wave_sum(float v, int key, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float ov = __shfl_down(v, o, 64);
        const int ok = __shfl_down(key, o, 64);
        if (lane + o < 64 && ok == key) v += ov;
    }
    return v;
}

// UNIW: every wave works on a single channel (thread slots per channel padded to a multiple of 64), so the 27
// weights live in SGPRs; together with the 128-VGPR cap this lets two 7-wave workgroups share a CU.
// DEPTH 2 (float4 loaders with <= 2 loads per thread): two input frames are in flight per workgroup.
template <int MODE, int S, int HS, int VEC, int MAXLD, bool UNIW>
__global__ __launch_bounds__(UNIW ? 256 : 512, UNIW ? (MAXLD == 2 ? (MODE == DW_FWD ? 4 : 3) : (MAXLD == 4 ? 3 : 2)) : 2) void dw3d_kernel(const DwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int HSIN = (HS - 1) * S + 3;
    // VEC: 4 = rows are whole float4s; 1 = scalar loader; 2 = FLAT: the row width is only even, but a plane is a whole
    // number of float4s and the band covers it: the loader walks the contiguous plane in float4s and writes each as two
    // float2 halves (neither crosses a row: even width, even start)
    constexpr bool FLAT = VEC == 2;
    constexpr int LV = VEC == 1 ? 1 : 4;
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;

    unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    const int chunk = L % a.nchunks; L /= a.nchunks;
    const int band = L % a.nbands;   L /= a.nbands;
    const int grp = L % a.ngroups;
    const int n = L / a.ngroups;
    const int c0 = grp * a.CG;
    const int ncg = min(a.CG, a.C - c0);
    const int t0 = chunk * a.TT, t1 = min(t0 + a.TT, a.T);
    const int RIN = a.RIN, WP = a.WP, XO = a.XO;
    const int Hi = a.Hi, Wi = a.Wi, Ho = a.Ho, Wo = a.Wo, T = a.T, C = a.C;
    const int hin0 = band * a.GB * HS * S - 1;                 // input row held by LDS row 0
    const int row_lo = max(hin0, 0), row_hi = min(hin0 + RIN, Hi);
    const int per_ch = (row_hi - row_lo) * Wi;                 // floats per channel per frame
    const int total_ld = ncg * per_ch;
    const long plane_i = (long)Hi * Wi, plane_o = (long)Ho * Wo;

    const int bufsz = a.CG * RIN * WP;                         // one frame image (all CG channels) incl. zero halo
    float* buf = smem;                                         // two images: frame parity selects one
    float* sA = buf + 2 * bufsz;
    float* sB = sA + a.CG;
    float* sR = sB + a.CG;                                     // reduction scratch: 27*CG floats

    for (int i = tid; i < 2 * bufsz; i += nthr) buf[i] = 0.0f;
    // reduction scratch: one slot set per WAVE (plain stores, summed in wave order: an LDS float atomic per wave would make
    // the fp32 sum depend on the arrival order and the results differ from run to run)
    const int nwv = nthr >> 6, wv = tid >> 6;
    constexpr int KJ = MODE == DW_WGRAD ? 27 : 2;          // values per (wave, channel) slot
    for (int i = tid; i < nwv * a.CG * KJ; i += nthr) sR[i] = 0.0f;
    if (tid < a.CG) {
        if (MODE == DW_DGRAD) {   // staged tensor is gy + gs + y * 2gq
            const bool ok = tid < ncg;
            sA[tid] = (ok && a.gs) ? (float)a.gs[(long)n * C + c0 + tid] : 0.0f;
            sB[tid] = (ok && a.gq && a.src2) ? 2.0f * (float)a.gq[(long)n * C + c0 + tid] : 0.0f;
        } else {                  // staged tensor is act(A * x + B)
            const bool ok = tid < ncg && a.A != nullptr;
            sA[tid] = ok ? a.A[(long)n * C + c0 + tid] : 1.0f;
            sB[tid] = ok ? a.B[(long)n * C + c0 + tid] : 0.0f;
        }
    }

    // ---- loader bookkeeping (frame invariant) -------------------------------------------------
    int rel[MAXLD];      // element offset from the (n, c0, frame) base, -1 = nothing to load
    int lofs[MAXLD];     // (channel_local << 16) | LDS float offset
    int lofs2[FLAT ? MAXLD : 1];   // FLAT: LDS float offset of elements 2, 3
#pragma unroll
    for (int k = 0; k < MAXLD; ++k) {
        const int e = (k * nthr + tid) * LV;
        if (e < total_ld) {
            const int cl = e / per_ch, off = e - cl * per_ch;
            const int r = off / Wi, col = off - r * Wi;
            rel[k] = (int)((long)cl * T * plane_i) + row_lo * Wi + off;
            lofs[k] = (cl << 16) | ((cl * RIN + (row_lo - hin0) + r) * WP + XO + col);
            if (FLAT) {
                const int r2 = (off + 2) / Wi, col2 = off + 2 - r2 * Wi;
                lofs2[k] = (cl * RIN + (row_lo - hin0) + r2) * WP + XO + col2;
            }
        } else {
            rel[k] = -1;
            lofs[k] = 0;
            if (FLAT) lofs2[k] = 0;
        }
    }

    // ---- compute-thread identity ----------------------------------------------------------------
    const int IPCb = a.IPCb, IPCp = a.IPCp;
    const int c_slot = tid / IPCp, item = tid - c_slot * IPCp;
    const bool active = c_slot < ncg && item < IPCb;
    const int c_local = c_slot < ncg ? c_slot : 0;
    const int gl = active ? item / Wo : 0;
    const int wo = active ? item - gl * Wo : 0;
    const int c = c0 + c_local;
    const int hrow0 = (band * a.GB + gl) * HS;                 // first output row of this thread
    const float* tb = buf + (c_local * RIN + gl * HS * S) * WP + (XO - 1) + wo * S;
    const long nc = (long)n * C + c;

    // ---- buffer descriptors over this (sample, channel group): every global access of the frame loop is an
    // UNCONDITIONAL buffer load / store whose per-lane offset is out of range when the access is not wanted (loads return
    // 0, stores are dropped).  With no vector-memory instruction under a branch the compiler can count them and waits
    // with vmcnt(N) for exactly the frame it needs instead of vmcnt(0), so the frames prefetched behind it stay in flight.
    // (measured: wins for the float4 forward and the data gradient; the scalar-loader forward and the weight gradient,
    // whose consumer needs the youngest loads anyway, are faster with plain predicated accesses)
    constexpr bool UNC = MODE == DW_DGRAD || (MODE == DW_FWD && LV == 4);
    constexpr int OOB = 0x7ffffff0;
    const long gi0 = ((long)n * C + c0) * T * plane_i, go0 = ((long)n * C + c0) * T * plane_o;
    const unsigned span_i = (unsigned)((long)ncg * T * plane_i * DW_ES), span_o = (unsigned)((long)ncg * T * plane_o * DW_ES);
    __amdgpu_buffer_rsrc_t rs1 = cfn_rsrc(const_cast<dwe_t*>(a.src + gi0), span_i);
    __amdgpu_buffer_rsrc_t rs2 = cfn_rsrc(const_cast<dwe_t*>((MODE == DW_DGRAD && a.src2 ? a.src2 : a.src) + gi0), span_i);
    // per-thread operands / results at output resolution: FWD/DGRAD dst, DGRAD xin, WGRAD gy / yout
    const dwe_t* po1 = MODE == DW_WGRAD ? a.gy : (MODE == DW_DGRAD && a.xin ? a.xin : a.src);
    const dwe_t* po2 = MODE == DW_WGRAD && a.yout ? a.yout : po1;
    __amdgpu_buffer_rsrc_t ro1 = cfn_rsrc(const_cast<dwe_t*>(po1 + go0), span_o);
    __amdgpu_buffer_rsrc_t ro2 = cfn_rsrc(const_cast<dwe_t*>(po2 + go0), span_o);
    __amdgpu_buffer_rsrc_t rd = cfn_rsrc((MODE == DW_WGRAD ? const_cast<dwe_t*>(a.src) : a.dst) + (MODE == DW_WGRAD ? gi0 : go0), MODE == DW_WGRAD ? 0u : span_o);
    const int ovo = active ? (int)(((long)c_local * T * plane_o + (long)hrow0 * Wo + wo) * DW_ES) : OOB;   // this thread's first output
    int relb[MAXLD];
#pragma unroll
    for (int k = 0; k < MAXLD; ++k) relb[k] = rel[k] >= 0 ? rel[k] * DW_ES : OOB;

    float wr[27];
    if (MODE != DW_WGRAD) {
#pragma unroll
        for (int j = 0; j < 27; ++j) {
            const float wv = a.w[(long)c * 27 + (MODE == DW_DGRAD ? 26 - j : j)];
            wr[j] = UNIW ? __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, wv))) : wv;
        }
    }
    // stats-gradient terms of the incoming gradient (DGRAD: on the LDS-staged tensor; WGRAD: on gy)
    float gs_c = 0.0f, gq2_c = 0.0f;
    if (MODE == DW_WGRAD && active) {
        if (a.gs) gs_c = (float)a.gs[nc];
        if (a.gq) gq2_c = 2.0f * (float)a.gq[nc];
    }
    // DGRAD epilogue coefficients (prologue of the forward input)
    float eA = 1.0f, eB = 0.0f;
    if (MODE == DW_DGRAD && active && a.A) { eA = a.A[nc]; eB = a.B[nc]; }

    float acc[3][HS];      // FWD/DGRAD rolling accumulators: [0]=frame+1, [1]=frame, [2]=frame-1
    float dwa[27];         // WGRAD partial weight gradients
    float gro[3][HS];      // WGRAD rolling gradients: [0]=g(frame+1) [1]=g(frame) [2]=g(frame-1)
#pragma unroll
    for (int i = 0; i < HS; ++i) { acc[0][i] = acc[1][i] = acc[2][i] = 0.0f; gro[0][i] = gro[1][i] = gro[2][i] = 0.0f; }
#pragma unroll
    for (int j = 0; j < 27; ++j) dwa[j] = 0.0f;
    float st1 = 0.0f, st2 = 0.0f;

    typedef float __attribute__((ext_vector_type(4))) f4;
    constexpr int DEPTH = (MODE != DW_WGRAD && LV == 4) ? 2 : 1;   // frames in flight in registers
    f4 pfA[MAXLD], pfA2[MAXLD], pfB[DEPTH == 2 ? MAXLD : 1], pfB2[DEPTH == 2 ? MAXLD : 1];
    const bool two_src = (MODE == DW_DGRAD) && a.src2 != nullptr;

    const int f_last_ = t1;                                   // last input frame of this chunk
    const bool simple_act = a.act == CFN_ACT_NONE || a.act == CFN_ACT_RELU;
    const float act_lo = a.act == CFN_ACT_RELU ? 0.0f : -__builtin_inff();
    auto frame_valid = [&](int f) { return f >= 0 && f < T; };
    typedef int __attribute__((ext_vector_type(4))) i4;
    auto prefetch = [&](int f, f4* pf, f4* pf2) {             // UNC: always issues MAXLD loads per tensor
        const bool fvd = frame_valid(f);
        if (!UNC) {
            if (!fvd || f > f_last_) return;
            const long base = gi0 + (long)f * plane_i;
#pragma unroll
            for (int k = 0; k < MAXLD; ++k) {
                if (rel[k] >= 0) {
                    if (LV == 4) pf[k] = dw_ld4(a.src + base + rel[k]);
                    else pf[k].x = dw_ld(a.src + base + rel[k]);
                }
            }
            return;
        }
        const int so = fvd ? f * (int)plane_i * DW_ES : 0;
#pragma unroll
        for (int k = 0; k < MAXLD; ++k) {
            const int vo = fvd ? relb[k] : OOB;
            if (LV == 4) {
                pf[k] = dw_bld4(rs1, vo, so);
                if (MODE == DW_DGRAD && two_src) pf2[k] = dw_bld4(rs2, vo, so);
            } else {
                pf[k].x = dw_bld1(rs1, vo, so);
                if (MODE == DW_DGRAD && two_src) pf2[k].x = dw_bld1(rs2, vo, so);
            }
        }
    };
    auto stage = [&](const f4* pf, const f4* pf2, float* img) {   // registers -> LDS with the load-time prologue
#pragma unroll
        for (int k = 0; k < MAXLD; ++k) {
            if (rel[k] >= 0) {
                const int cl = lofs[k] >> 16, lo = lofs[k] & 0xffff;
                f4 v = pf[k];
                const float pa = sA[cl], pb = sB[cl];
                if (MODE == DW_DGRAD) {
                    if (two_src) {
                        v.x = fmaf(pf2[k].x, pb, v.x + pa);
                        if (LV == 4) {
                            v.y = fmaf(pf2[k].y, pb, v.y + pa);
                            v.z = fmaf(pf2[k].z, pb, v.z + pa);
                            v.w = fmaf(pf2[k].w, pb, v.w + pa);
                        }
                    } else {
                        v.x += pa;
                        if (LV == 4) { v.y += pa; v.z += pa; v.w += pa; }
                    }
                } else if (simple_act) {   // none / ReLU (every X3D conv2): branch-free max against -inf or 0
                    v.x = fmaxf(fmaf(v.x, pa, pb), act_lo);
                    if (LV == 4) {
                        v.y = fmaxf(fmaf(v.y, pa, pb), act_lo);
                        v.z = fmaxf(fmaf(v.z, pa, pb), act_lo);
                        v.w = fmaxf(fmaf(v.w, pa, pb), act_lo);
                    }
                } else {
                    v.x = cfn_act_rt(fmaf(v.x, pa, pb), a.act);
                    if (LV == 4) {
                        v.y = cfn_act_rt(fmaf(v.y, pa, pb), a.act);
                        v.z = cfn_act_rt(fmaf(v.z, pa, pb), a.act);
                        v.w = cfn_act_rt(fmaf(v.w, pa, pb), a.act);
                    }
                }
                typedef float __attribute__((ext_vector_type(2))) f2;
                if (FLAT) {
                    *reinterpret_cast<f2*>(img + lo) = f2{v.x, v.y};
                    *reinterpret_cast<f2*>(img + lofs2[k]) = f2{v.z, v.w};
                } else if (LV == 4) *reinterpret_cast<f4*>(img + lo) = v;
                else img[lo] = v.x;
            }
        }
    };
    // WGRAD: upstream gradient of output frame t for this thread's HS outputs.  The loads only land in registers
    // (raw gy / y); the gy + gs + 2*y*gq arithmetic happens one frame later in take_g, so no wait sits behind the loads
    float gnn[HS], gny[HS];   // raw gy / y of frame+2
    bool gn_ok = false;
    auto load_g = [&](int t) {
        gn_ok = t >= t0 && t < t1 && active;
        if (gn_ok) {
            const long o = (nc * T + t) * plane_o + (long)hrow0 * Wo + wo;
#pragma unroll
            for (int i = 0; i < HS; ++i) {
                gnn[i] = dw_ld(a.gy + o + (long)i * Wo);
                if (a.yout) gny[i] = dw_ld(a.yout + o + (long)i * Wo);
            }
        }
    };
    auto take_g = [&](float (&g)[HS]) {
#pragma unroll
        for (int i = 0; i < HS; ++i) {
            float v = gnn[i] + gs_c;
            if (a.yout) v = fmaf(gny[i], gq2_c, v);
            g[i] = gn_ok ? v : 0.0f;
        }
    };

    __syncthreads();   // zero fill + sA/sB visible

    const int f_first = t0 - 1, f_last = t1;   // input frames t0-1 .. t1 (inclusive)
    if (MODE == DW_WGRAD) load_g(t0);   // consumed (rotated in) at the start of the first step
    // prime the pipeline: frame f_first staged in image 0, the next DEPTH frames in flight in registers
    prefetch(f_first, pfA, pfA2);
    if (DEPTH == 2) prefetch(f_first + 1, pfB, pfB2);
    if (UNC || frame_valid(f_first)) stage(pfA, pfA2, buf);
    prefetch(f_first + DEPTH, pfA, pfA2);
    __syncthreads();

    float xen[HS];   // DGRAD: forward input of the next frame to be emitted
#pragma unroll
    for (int i = 0; i < HS; ++i) xen[i] = 0.0f;
    // one frame step (ONE barrier): stage frame f+1 from `nx` into the other LDS image, refill `nx` with frame
    // f+1+DEPTH, compute frame f from image `par`, emit output frame f-1
    // Inside a step every consumer of loads issued one step earlier comes BEFORE any new load is issued: the loads sit
    // under (uniform) conditions, so the compiler can only wait with vmcnt(0) -- a wait placed after a fresh prefetch
    // would expose a full HBM round trip in every frame (measured: 60-75 % of the step).
    auto step = [&](int f, int par, f4* nx, f4* nx2) {
        const bool fv = frame_valid(f);
        const int to = f - 1;                                  // output frame completed by this step
        const bool emit = (to >= t0 && to < t1) && active;
        // ---- consume ------------------------------------------------------------------------------------------
        float xe[HS];
        if (MODE == DW_DGRAD && a.A) {
#pragma unroll
            for (int i = 0; i < HS; ++i) xe[i] = xen[i];
        }
        if (MODE == DW_WGRAD) {
#pragma unroll
            for (int i = 0; i < HS; ++i) { gro[2][i] = gro[1][i]; gro[1][i] = gro[0][i]; }
            take_g(gro[0]);                                    // g(f+1), loaded during the previous step
        }
        if (UNC || (f + 1 <= f_last && frame_valid(f + 1))) stage(nx, nx2, buf + (par ^ 1) * bufsz);   // UNC: junk frames are never read
        // ---- issue (unconditional) ----------------------------------------------------------------------------
        prefetch(f + 1 + DEPTH, nx, nx2);
        if (MODE == DW_DGRAD && a.A) {
            const bool want = to + 1 >= t0 && to + 1 < t1;
            const int vo = want ? ovo : OOB, so = want ? (to + 1) * (int)plane_o * DW_ES : 0;
#pragma unroll
            for (int i = 0; i < HS; ++i) xen[i] = dw_bld1(ro1, vo + i * Wo * DW_ES, so);
        }
        if (MODE == DW_WGRAD) load_g(f + 2);
        const float* tbp = tb + par * bufsz;

        if (fv && active) {
#pragma unroll
            for (int r = 0; r < HSIN; ++r) {
                const float v0 = tbp[r * WP], v1 = tbp[r * WP + 1], v2 = tbp[r * WP + 2];
#pragma unroll
                for (int i = 0; i < HS; ++i) {
                    const int kh = r - i * S;
                    if (kh >= 0 && kh < 3) {
                        if (MODE == DW_WGRAD) {
#pragma unroll
                            for (int kt = 0; kt < 3; ++kt) {
                                dwa[kt * 9 + kh * 3 + 0] = fmaf(gro[kt][i], v0, dwa[kt * 9 + kh * 3 + 0]);
                                dwa[kt * 9 + kh * 3 + 1] = fmaf(gro[kt][i], v1, dwa[kt * 9 + kh * 3 + 1]);
                                dwa[kt * 9 + kh * 3 + 2] = fmaf(gro[kt][i], v2, dwa[kt * 9 + kh * 3 + 2]);
                            }
                        } else {
#pragma unroll
                            for (int kt = 0; kt < 3; ++kt)
                                acc[kt][i] = fmaf(wr[kt * 9 + kh * 3 + 0], v0,
                                             fmaf(wr[kt * 9 + kh * 3 + 1], v1,
                                             fmaf(wr[kt * 9 + kh * 3 + 2], v2, acc[kt][i])));
                        }
                    }
                }
            }
        }

        if (MODE != DW_WGRAD) {
            // the scalar offset depends on wave-uniform values only (a per-lane soffset makes hipcc wrap every store in a
            // waterfall loop); the lane's own validity is in ovo
            const bool emit_u = to >= t0 && to < t1;
            const int vo = emit_u ? ovo : OOB, so = emit_u ? to * (int)plane_o * DW_ES : 0;
            const float em = emit ? 1.0f : 0.0f;
#pragma unroll
            for (int i = 0; i < HS; ++i) {
                float v = acc[2][i];
                if (MODE == DW_FWD) {
                    v = dw_rt(v);
                    st1 = fmaf(v, em, st1);
                    st2 = fmaf(v * em, v, st2);
                } else if (a.A) {
                    const float z = fmaf(xe[i], eA, eB);
                    const float dz = v * cfn_act_grad_rt(z, a.act) * em;
                    st1 = fmaf(dz, xe[i], st1);
                    st2 += dz;
                    v = dz * eA;
                }
                if (UNC) dw_bst1(v, rd, vo + i * Wo * DW_ES, so);
                else if (emit) dw_st(a.dst + go0 + (long)c_local * T * plane_o + (long)to * plane_o + (long)hrow0 * Wo + wo + (long)i * Wo, v);
            }
#pragma unroll
            for (int i = 0; i < HS; ++i) { acc[2][i] = acc[1][i]; acc[1][i] = acc[0][i]; acc[0][i] = 0.0f; }
        }
        __syncthreads();   // image par fully consumed, image par^1 fully written
    };
    // frame f_first+1 sits in set B (DEPTH 2) or A (DEPTH 1); sets alternate with the frame parity
    for (int f = f_first; f <= f_last; f += 2) {
        step(f, 0, DEPTH == 2 ? pfB : pfA, DEPTH == 2 ? pfB2 : pfA2);
        if (f + 1 <= f_last) step(f + 1, 1, pfA, pfA2);
    }

    // ---- block reductions -> one fp64 atomic per (channel, value) per block -----------------
    const int key = active ? c_local : -1 - (tid >> 6);
    const int prev_key = __shfl_up(key, 1, 64);   // all lanes take part in the shuffle
    const bool head = active && (lane == 0 || prev_key != key);
    if (MODE == DW_WGRAD) {
#pragma unroll
        for (int j = 0; j < 27; ++j) {
            const float r = seg_wave_sum(dwa[j], key, lane);
            if (head) sR[(wv * a.CG + c_local) * 27 + j] = r;
        }
        __syncthreads();
        for (int i = tid; i < ncg * 27; i += nthr) {
            float v = 0.0f;
            for (int w = 0; w < nwv; ++w) v += sR[w * a.CG * 27 + i];
            cfn_add64(&a.s1[(long)c0 * 27 + i], (double)v);
        }
    } else if (a.s1 != nullptr) {
        const float r1 = seg_wave_sum(st1, key, lane);
        const float r2 = seg_wave_sum(st2, key, lane);
        if (head) { sR[(wv * a.CG + c_local) * KJ] = r1; sR[(wv * a.CG + c_local) * KJ + 1] = r2; }
        __syncthreads();
        if (tid < ncg) {
            float v1 = 0.0f, v2 = 0.0f;
            for (int w = 0; w < nwv; ++w) { v1 += sR[(w * a.CG + tid) * KJ]; v2 += sR[(w * a.CG + tid) * KJ + 1]; }
            cfn_add64(&a.s1[(long)n * C + c0 + tid], (double)v1);
            cfn_add64(&a.s2[(long)n * C + c0 + tid], (double)v2);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Fused backward of the stride-1 depthwise conv: data gradient AND weight gradient in one pass.  Both read gy, y and
// x; run separately they move 7 tensor passes (dgrad: gy, y, x -> gx; wgrad: gy, y, x), fused 4.  Same band / t-chunk
// skeleton as dw3d_kernel with TWO staged streams that are one frame apart:
//   G image = g'(f) = gy + gs + 2 y gq   (window -> data gradient, centre -> weight gradient)
//   A image = a(f-1) = act(A x + B)      (window -> weight gradient)
// At step f:  gx accumulators += flipped taps * G-window(f);  gw[kt] += g'(f-kt)[centre] * A-window(f-1), the three g'
// centres rolling in registers; output frame f-1 is finished with the act' epilogue (raw x of that frame prefetched
// one step ahead).  DW_WGRAD's t_out = fa - kt + 1 pairing with fa = f-1.
// ---------------------------------------------------------------------------------------------
struct DwFusedArgs {
    const dwe_t* gy; const dwe_t* y; const double* gs; const double* gq; const float* w; const dwe_t* x;
    const double* A; const double* B; dwe_t* gx; double* gA; double* gB; double* gw;
    int N, C, T, H, W, act;
    int TT, nchunks, CG, ngroups, GB, nbands, IPCb, IPCp, RIN, WP, XO;
};

template <int HS, int VEC, int MAXLD, bool UNIW>
__global__ __launch_bounds__(UNIW ? 256 : 512, 2) void dw3d_bwd_fused_kernel(const DwFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int HSIN = HS + 2;
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;
    unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    const int chunk = L % a.nchunks; L /= a.nchunks;
    const int band = L % a.nbands;   L /= a.nbands;
    const int grp = L % a.ngroups;
    const int n = L / a.ngroups;
    const int c0 = grp * a.CG;
    const int ncg = min(a.CG, a.C - c0);
    const int t0 = chunk * a.TT, t1 = min(t0 + a.TT, a.T);
    const int RIN = a.RIN, WP = a.WP, XO = a.XO;
    const int H = a.H, W = a.W, T = a.T, C = a.C;
    const int hin0 = band * a.GB * HS - 1;
    const int row_lo = max(hin0, 0), row_hi = min(hin0 + RIN, H);
    const int per_ch = (row_hi - row_lo) * W;
    const int total_ld = ncg * per_ch;
    const long plane = (long)H * W;

    const int bufsz = a.CG * RIN * WP;
    float* bufG = smem;                    // 2 images
    float* bufA = smem + 2 * bufsz;        // 2 images
    float* sGs = bufA + 2 * bufsz;         // [CG] gs, [CG] 2gq, [CG] A, [CG] B
    float* sGq = sGs + a.CG;
    float* sPa = sGq + a.CG;
    float* sPb = sPa + a.CG;
    float* sR = sPb + a.CG;                // 27*CG (+ reuse for the 2*CG statistics)
    for (int i = tid; i < 4 * bufsz; i += nthr) smem[i] = 0.0f;
    const int nwv = nthr >> 6, wv = tid >> 6;             // per-wave reduction slots, see dw3d_kernel
    for (int i = tid; i < nwv * a.CG * 27; i += nthr) sR[i] = 0.0f;
    if (tid < a.CG) {
        const bool ok = tid < ncg;
        const long nci = (long)n * C + c0 + tid;
        sGs[tid] = (ok && a.gs) ? (float)a.gs[nci] : 0.0f;
        sGq[tid] = (ok && a.gq && a.y) ? 2.0f * (float)a.gq[nci] : 0.0f;
        sPa[tid] = (ok && a.A) ? (float)a.A[nci] : 1.0f;
        sPb[tid] = (ok && a.A) ? (float)a.B[nci] : 0.0f;
    }
    int rel[MAXLD], lofs[MAXLD];
#pragma unroll
    for (int k = 0; k < MAXLD; ++k) {
        const int e = (k * nthr + tid) * VEC;
        if (e < total_ld) {
            const int cl = e / per_ch, off = e - cl * per_ch;
            const int r = off / W, col = off - r * W;
            rel[k] = (int)((long)cl * T * plane) + row_lo * W + off;
            lofs[k] = (cl << 16) | ((cl * RIN + (row_lo - hin0) + r) * WP + XO + col);
        } else {
            rel[k] = -1;
            lofs[k] = 0;
        }
    }
    const int IPCb = a.IPCb, IPCp = a.IPCp;
    const int c_slot = tid / IPCp, item = tid - c_slot * IPCp;
    const bool active = c_slot < ncg && item < IPCb;
    const int c_local = c_slot < ncg ? c_slot : 0;
    const int gl = active ? item / W : 0;
    const int wo = active ? item - gl * W : 0;
    const int c = c0 + c_local;
    const int hrow0 = (band * a.GB + gl) * HS;
    const int tofs = (c_local * RIN + gl * HS) * WP + (XO - 1) + wo;
    const long nc = (long)n * C + c;

    float wr[27];
#pragma unroll
    for (int j = 0; j < 27; ++j) {
        const float wv = a.w[(long)c * 27 + 26 - j];             // flipped taps for the data gradient
        wr[j] = UNIW ? __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, wv))) : wv;
    }
    float eA = 1.0f, eB = 0.0f;
    if (active && a.A) { eA = (float)a.A[nc]; eB = (float)a.B[nc]; }

    float acc[3][HS], dwa[27], gro[3][HS], xen[HS];
#pragma unroll
    for (int i = 0; i < HS; ++i) { acc[0][i] = acc[1][i] = acc[2][i] = 0.0f; gro[0][i] = gro[1][i] = gro[2][i] = 0.0f; xen[i] = 0.0f; }
#pragma unroll
    for (int j = 0; j < 27; ++j) dwa[j] = 0.0f;
    float st1 = 0.0f, st2 = 0.0f;

    typedef float __attribute__((ext_vector_type(4))) f4;
    f4 pfG[MAXLD], pfY[MAXLD], pfX[MAXLD];
    const bool has_y = a.y != nullptr;
    const long gbase = ((long)n * C + c0) * T * plane;
    auto fvalid = [&](int f) { return f >= 0 && f < T; };
    auto prefetchG = [&](int f) {
        if (!fvalid(f) || f > t1) return;
        const long base = gbase + (long)f * plane;
#pragma unroll
        for (int k = 0; k < MAXLD; ++k) {
            if (rel[k] >= 0) {
                if (VEC == 4) {
                    pfG[k] = dw_ld4(a.gy + base + rel[k]);
                    if (has_y) pfY[k] = dw_ld4(a.y + base + rel[k]);
                } else {
                    pfG[k].x = dw_ld(a.gy + base + rel[k]);
                    if (has_y) pfY[k].x = dw_ld(a.y + base + rel[k]);
                }
            }
        }
    };
    auto prefetchX = [&](int f) {
        if (!fvalid(f) || f > t1) return;
        const long base = gbase + (long)f * plane;
#pragma unroll
        for (int k = 0; k < MAXLD; ++k) {
            if (rel[k] >= 0) {
                if (VEC == 4) pfX[k] = dw_ld4(a.x + base + rel[k]);
                else pfX[k].x = dw_ld(a.x + base + rel[k]);
            }
        }
    };
    auto stageG = [&](float* img) {
#pragma unroll
        for (int k = 0; k < MAXLD; ++k) {
            if (rel[k] >= 0) {
                const int cl = lofs[k] >> 16, lo = lofs[k] & 0xffff;
                const float ps = sGs[cl], pq = sGq[cl];
                f4 v = pfG[k];
                v.x = has_y ? fmaf(pfY[k].x, pq, v.x + ps) : v.x + ps;
                if (VEC == 4) {
                    v.y = has_y ? fmaf(pfY[k].y, pq, v.y + ps) : v.y + ps;
                    v.z = has_y ? fmaf(pfY[k].z, pq, v.z + ps) : v.z + ps;
                    v.w = has_y ? fmaf(pfY[k].w, pq, v.w + ps) : v.w + ps;
                    *reinterpret_cast<f4*>(img + lo) = v;
                } else {
                    img[lo] = v.x;
                }
            }
        }
    };
    auto stageA = [&](float* img) {
#pragma unroll
        for (int k = 0; k < MAXLD; ++k) {
            if (rel[k] >= 0) {
                const int cl = lofs[k] >> 16, lo = lofs[k] & 0xffff;
                const float pa = sPa[cl], pb = sPb[cl];
                f4 v = pfX[k];
                v.x = cfn_act_rt(fmaf(v.x, pa, pb), a.act);
                if (VEC == 4) {
                    v.y = cfn_act_rt(fmaf(v.y, pa, pb), a.act);
                    v.z = cfn_act_rt(fmaf(v.z, pa, pb), a.act);
                    v.w = cfn_act_rt(fmaf(v.w, pa, pb), a.act);
                    *reinterpret_cast<f4*>(img + lo) = v;
                } else {
                    img[lo] = v.x;
                }
            }
        }
    };

    __syncthreads();
    const int f_first = t0 - 1, f_last = t1 + 1;
    // prime: G(f_first) and A(f_first - 1) in images 0 (the latter is never paired: all its t_out lie outside)
    prefetchG(f_first);
    if (fvalid(f_first)) stageG(bufG);
    prefetchG(f_first + 1);
    prefetchX(f_first);                                       // A(f_first) is staged during the first step
    __syncthreads();

    for (int f = f_first, par = 0; f <= f_last; ++f, par ^= 1) {
        const bool fv = fvalid(f) && f <= t1;                  // G(f) staged and meaningful
        const bool av = fvalid(f - 1) && f - 1 >= t0 - 1 && f - 1 <= t1 && f > f_first;   // A(f-1) staged
        const int to = f - 1;
        const bool emit = (to >= t0 && to < t1) && active;
        // ---- consume ------------------------------------------------------------------------------------------
        float xe[HS];
#pragma unroll
        for (int i = 0; i < HS; ++i) xe[i] = xen[i];
        if (f + 1 <= t1 && fvalid(f + 1)) stageG(bufG + (par ^ 1) * bufsz);
        if (f <= t1 && fvalid(f)) stageA(bufA + (par ^ 1) * bufsz);
        // ---- issue --------------------------------------------------------------------------------------------
        prefetchG(f + 2);
        prefetchX(f + 1);
        if (a.A && to + 1 >= t0 && to + 1 < t1 && active) {
            const long o = (nc * T + to + 1) * plane + (long)hrow0 * W + wo;
#pragma unroll
            for (int i = 0; i < HS; ++i) xen[i] = dw_ld(a.x + o + (long)i * W);
        }
        // ---- data gradient from the G window, g' centres for the weight gradient ----------------------------
#pragma unroll
        for (int i = 0; i < HS; ++i) { gro[2][i] = gro[1][i]; gro[1][i] = gro[0][i]; gro[0][i] = 0.0f; }
        if (fv && active) {
            const float* tg = bufG + par * bufsz + tofs;
            const bool inchunk = f >= t0 && f < t1;            // g'(f) is a weight-gradient term only inside the chunk
#pragma unroll
            for (int r = 0; r < HSIN; ++r) {
                const float v0 = tg[r * WP], v1 = tg[r * WP + 1], v2 = tg[r * WP + 2];
                if (r >= 1 && r <= HS) gro[0][r - 1] = inchunk ? v1 : 0.0f;
#pragma unroll
                for (int i = 0; i < HS; ++i) {
                    const int kh = r - i;
                    if (kh >= 0 && kh < 3) {
#pragma unroll
                        for (int kt = 0; kt < 3; ++kt)
                            acc[kt][i] = fmaf(wr[kt * 9 + kh * 3 + 0], v0,
                                         fmaf(wr[kt * 9 + kh * 3 + 1], v1,
                                         fmaf(wr[kt * 9 + kh * 3 + 2], v2, acc[kt][i])));
                    }
                }
            }
        }
        // ---- weight gradient: A window of frame f-1 against g'(f), g'(f-1), g'(f-2) -------------------------
        if (av && active) {
            const float* ta = bufA + par * bufsz + tofs;
#pragma unroll
            for (int r = 0; r < HSIN; ++r) {
                const float v0 = ta[r * WP], v1 = ta[r * WP + 1], v2 = ta[r * WP + 2];
#pragma unroll
                for (int i = 0; i < HS; ++i) {
                    const int kh = r - i;
                    if (kh >= 0 && kh < 3) {
#pragma unroll
                        for (int kt = 0; kt < 3; ++kt) {
                            dwa[kt * 9 + kh * 3 + 0] = fmaf(gro[kt][i], v0, dwa[kt * 9 + kh * 3 + 0]);
                            dwa[kt * 9 + kh * 3 + 1] = fmaf(gro[kt][i], v1, dwa[kt * 9 + kh * 3 + 1]);
                            dwa[kt * 9 + kh * 3 + 2] = fmaf(gro[kt][i], v2, dwa[kt * 9 + kh * 3 + 2]);
                        }
                    }
                }
            }
        }
        // ---- emit gx(to) ----------------------------------------------------------------------------------------
        if (emit) {
            const long o = (nc * T + to) * plane + (long)hrow0 * W + wo;
#pragma unroll
            for (int i = 0; i < HS; ++i) {
                float v = acc[2][i];
                if (a.A) {
                    const float dz = v * cfn_act_grad_rt(fmaf(xe[i], eA, eB), a.act);
                    st1 = fmaf(dz, xe[i], st1);
                    st2 += dz;
                    v = dz * eA;
                }
                dw_st(a.gx + o + (long)i * W, v);
            }
        }
#pragma unroll
        for (int i = 0; i < HS; ++i) { acc[2][i] = acc[1][i]; acc[1][i] = acc[0][i]; acc[0][i] = 0.0f; }
        __syncthreads();
    }

    // ---- reductions: gw (27 per channel), then gA / gB ---------------------------------------------------------
    const int key = active ? c_local : -1 - (tid >> 6);
    const int prev_key = __shfl_up(key, 1, 64);
    const bool head = active && (lane == 0 || prev_key != key);
#pragma unroll
    for (int j = 0; j < 27; ++j) {
        const float r = seg_wave_sum(dwa[j], key, lane);
        if (head) sR[(wv * a.CG + c_local) * 27 + j] = r;
    }
    __syncthreads();
    for (int i = tid; i < ncg * 27; i += nthr) {
        float v = 0.0f;
        for (int w = 0; w < nwv; ++w) v += sR[w * a.CG * 27 + i];
        cfn_add64(&a.gw[(long)c0 * 27 + i], (double)v);
    }
    if (a.A && a.gA) {
        __syncthreads();
        for (int i = tid; i < nwv * a.CG * 27; i += nthr) sR[i] = 0.0f;
        __syncthreads();
        const float r1 = seg_wave_sum(st1, key, lane);
        const float r2 = seg_wave_sum(st2, key, lane);
        if (head) { sR[(wv * a.CG + c_local) * 27] = r1; sR[(wv * a.CG + c_local) * 27 + 1] = r2; }
        __syncthreads();
        if (tid < ncg) {
            float v1 = 0.0f, v2 = 0.0f;
            for (int w = 0; w < nwv; ++w) { v1 += sR[(w * a.CG + tid) * 27]; v2 += sR[(w * a.CG + tid) * 27 + 1]; }
            cfn_add64(&a.gA[(long)n * C + c0 + tid], (double)v1);
            cfn_add64(&a.gB[(long)n * C + c0 + tid], (double)v2);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// stride-2 data gradient (first block of every stage).  A thread owns one position (i,j) of the
// (Ho x Wo) gradient plane = the 2x2 input block (2i..2i+1, 2j..2j+1) and marches along t holding
// the 2x2 neighbourhood g'[f][i..i+1][j..j+1] of three consecutive gradient frames in registers,
// so every gradient element is fetched once per thread (neighbours come from L1), x is read once
// and gx written once with 8-byte accesses.  Tap parity: even input row <- kh=1 only, odd row <-
// kh in {0,2}; same along w  =>  27 FMAs per 2x2 block per frame.
// ---------------------------------------------------------------------------------------------
struct DwS2Args {
    const dwe_t* gy; const dwe_t* y; const double* gs; const double* gq; const float* w;
    const dwe_t* x; const double* A; const double* B; dwe_t* gx; double* gA; double* gB;
    int C, T, Hi, Wi, Ho, Wo, act, TT, nchunks, pblocks;
    int PBLK;            // positions per workgroup (the plane is split evenly over pblocks workgroups)
    int PB, CPB, NC;     // small planes: CPB channels of PB = Ho*Wo positions share a workgroup
};

// PACKED: several small-plane channels per workgroup (per-lane channel, weights in VGPRs); otherwise the channel is
// workgroup uniform and its 27 weights and coefficients live in SGPRs
template <bool PACKED>
__global__ __launch_bounds__(256) void dw3d_dgrad_s2_kernel(const DwS2Args a) {
    const int Ho = a.Ho, Wo = a.Wo, Hi = a.Hi, Wi = a.Wi, T = a.T;
    const int chunk = blockIdx.x % a.nchunks, pb = blockIdx.x / a.nchunks;
    // big planes: one (n,c) per blockIdx.y, 256 positions per workgroup; small planes (<= 128 positions): CPB channels
    // per workgroup so that the lanes stay busy (a 7x7 plane would fill 49 of 256 threads)
    const int cslot = PACKED ? threadIdx.x / a.PB : 0;
    const int p = PACKED ? threadIdx.x - cslot * a.PB : (threadIdx.x < a.PBLK ? pb * a.PBLK + threadIdx.x : Ho * Wo);
    const int by = blockIdx.y + blockIdx.z * gridDim.y;
    const int ncr = PACKED ? by * a.CPB + cslot : by;
    const bool ok = (!PACKED || (cslot < a.CPB && ncr < a.NC)) && p < Ho * Wo;
    const int nc = (!PACKED || ncr < a.NC) ? ncr : a.NC - 1, c = nc % a.C;
    const int i = ok ? p / Wo : 0, j = ok ? p - i * Wo : 0;
    const bool i1 = i + 1 < Ho, j1 = j + 1 < Wo;              // neighbours inside the gradient plane
    const bool r1 = 2 * i + 1 < Hi, c1 = 2 * j + 1 < Wi;      // odd row / column inside the input plane
    const float gsv = a.gs ? (float)a.gs[nc] : 0.0f;
    const float gqv = (a.gq && a.y) ? 2.0f * (float)a.gq[nc] : 0.0f;
    float w[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) w[k] = a.w[c * 27 + k];
    const float pa = a.A ? a.A[nc] : 1.0f, pb2 = a.A ? a.B[nc] : 0.0f;
    const long po = (long)Ho * Wo, pi = (long)Hi * Wi;
    const dwe_t* gyb = a.gy + (long)nc * T * po + (long)i * Wo + j;
    const dwe_t* yb = a.y ? a.y + (long)nc * T * po + (long)i * Wo + j : nullptr;

    const int t0 = chunk * a.TT, t1 = min(t0 + a.TT, T);
    // g'[f] at (i,j) (i,j+1) (i+1,j) (i+1,j+1), zero outside.  Software pipelined: ld_raw only issues the loads
    // (frame t+2 while frame t is computed), fin applies gy + gs + 2*y*gq when the frame is consumed.
    auto ld_raw = [&](int f, float (&g)[4], float (&y)[4]) -> bool {
        const bool v = ok && f >= 0 && f < T;
        if (v) {
            const long o = (long)f * po;
            g[0] = dw_ld(gyb + o);
            if (j1) g[1] = dw_ld(gyb + o + 1);
            if (i1) g[2] = dw_ld(gyb + o + Wo);
            if (i1 && j1) g[3] = dw_ld(gyb + o + Wo + 1);
            if (yb) {
                y[0] = dw_ld(yb + o);
                if (j1) y[1] = dw_ld(yb + o + 1);
                if (i1) y[2] = dw_ld(yb + o + Wo);
                if (i1 && j1) y[3] = dw_ld(yb + o + Wo + 1);
            }
        }
        return v;
    };
    auto fin = [&](bool v, const float (&g)[4], const float (&y)[4], float (&out)[4]) {
        const bool in[4] = {v, v && j1, v && i1, v && i1 && j1};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float r = g[k] + gsv;
            if (yb) r = fmaf(y[k], gqv, r);
            out[k] = in[k] ? r : 0.0f;
        }
    };
    typedef float __attribute__((ext_vector_type(2))) f2;
    const bool pair = (Wi & 1) == 0;           // even rows => (2i, 2j) 8-byte aligned, column 2j+1 always inside
    auto ld_x = [&](int t, float (&xv)[4]) {   // forward input of the 2x2 block (only needed with a prologue)
        if (!(a.A && ok && t < t1)) return;
        const long o = ((long)nc * T + t) * pi + (long)(2 * i) * Wi + 2 * j;
        if (pair) {
            const f2 u = dw_ld2(a.x + o);
            xv[0] = u.x; xv[1] = u.y;
            if (r1) { const f2 d = dw_ld2(a.x + o + Wi); xv[2] = d.x; xv[3] = d.y; }
        } else {
            xv[0] = dw_ld(a.x + o);
            if (c1) xv[1] = dw_ld(a.x + o + 1);
            if (r1) xv[2] = dw_ld(a.x + o + Wi);
            if (r1 && c1) xv[3] = dw_ld(a.x + o + Wi + 1);
        }
    };
    float G[3][4], rg[4], ry[4], xn[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) rg[k] = ry[k] = xn[k] = 0.0f;
    { const bool v = ld_raw(t0 - 1, rg, ry); fin(v, rg, ry, G[1]); }
    { const bool v = ld_raw(t0, rg, ry); fin(v, rg, ry, G[2]); }
    bool rv = ld_raw(t0 + 1, rg, ry);
    ld_x(t0, xn);
    float s1 = 0.0f, s2 = 0.0f;
    for (int t = t0; t < t1; ++t) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { G[0][k] = G[1][k]; G[1][k] = G[2][k]; }
        fin(rv, rg, ry, G[2]);                 // frame t+1 (loaded one iteration ago)
        float xv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) xv[k] = xn[k];
        rv = ld_raw(t + 2, rg, ry);
        ld_x(t + 1, xn);
        // gx[t] = sum_kt W[kt] * g'[t+1-kt]  -> frame slot 2-kt
        float o00 = 0.f, o01 = 0.f, o10 = 0.f, o11 = 0.f;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            const float* g = G[2 - kt];
            const float* wk = w + kt * 9;
            o00 = fmaf(wk[4], g[0], o00);                                          // kh=1,kw=1
            o01 = fmaf(wk[3], g[1], fmaf(wk[5], g[0], o01));                       // kh=1; kw=0 -> j+1, kw=2 -> j
            o10 = fmaf(wk[1], g[2], fmaf(wk[7], g[0], o10));                       // kw=1; kh=0 -> i+1, kh=2 -> i
            o11 = fmaf(wk[0], g[3], fmaf(wk[2], g[2], fmaf(wk[6], g[1], fmaf(wk[8], g[0], o11))));
        }
        if (ok) {
            const long o = ((long)nc * T + t) * pi + (long)(2 * i) * Wi + 2 * j;
            float v[4] = {o00, o01, o10, o11};
            const bool in[4] = {true, c1, r1, r1 && c1};
            if (a.A) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (!in[k]) continue;
                    const float dz = v[k] * cfn_act_grad_rt(fmaf(xv[k], pa, pb2), a.act);
                    s1 = fmaf(dz, xv[k], s1);
                    s2 += dz;
                    v[k] = dz * pa;
                }
            }
            if (pair) {
                dw_st2(a.gx + o, (f2){v[0], v[1]});
                if (r1) dw_st2(a.gx + o + Wi, (f2){v[2], v[3]});
            } else {
                dw_st(a.gx + o, v[0]);
                if (c1) dw_st(a.gx + o + 1, v[1]);
                if (r1) dw_st(a.gx + o + Wi, v[2]);
                if (r1 && c1) dw_st(a.gx + o + Wi + 1, v[3]);
            }
        }
    }
    if (a.A && a.gA) {
        __shared__ float sh[8];
        const int lane = threadIdx.x & 63;
        if (PACKED) {      // per-channel runs inside each wave -> one fp64 atomic pair per run
            const int key = ok ? cslot : -1 - (int)(threadIdx.x >> 6);
            const int prev_key = __shfl_up(key, 1, 64);
            const bool head = ok && (lane == 0 || prev_key != key);
            const float q1 = seg_wave_sum(s1, key, lane);
            const float q2 = seg_wave_sum(s2, key, lane);
            if (head) {
                cfn_add64(&a.gA[nc], (double)q1);
                cfn_add64(&a.gB[nc], (double)q2);
            }
        } else {           // one channel per workgroup: ONE atomic pair per workgroup (same-address fp64 atomics serialise)
            s1 = cfn_wave_sum(s1); s2 = cfn_wave_sum(s2);
            if (lane == 0) { sh[threadIdx.x >> 6] = s1; sh[4 + (threadIdx.x >> 6)] = s2; }
            __syncthreads();
            if (threadIdx.x == 0) {
                cfn_add64(&a.gA[nc], (double)(sh[0] + sh[1] + sh[2] + sh[3]));
                cfn_add64(&a.gB[nc], (double)(sh[4] + sh[5] + sh[6] + sh[7]));
            }
        }
    }
}

// Fast variant of the stride-2 data gradient for the training configuration (even input width, forward prologue and
// the sum-of-squares term present): same arithmetic, but every global access of the frame loop is an unconditional
// buffer load / store (out-of-range offset = not wanted), so the compiler waits with exact vmcnt values and the loads
// of frame t+2 stay in flight while frame t is finished.
template <bool PACKED>
__global__ __launch_bounds__(256) void dw3d_dgrad_s2_fast_kernel(const DwS2Args a) {
    const int Ho = a.Ho, Wo = a.Wo, Hi = a.Hi, Wi = a.Wi, T = a.T;
    const int chunk = blockIdx.x % a.nchunks, pb = blockIdx.x / a.nchunks;
    const int cslot = PACKED ? threadIdx.x / a.PB : 0;
    const int p = PACKED ? threadIdx.x - cslot * a.PB : (threadIdx.x < a.PBLK ? pb * a.PBLK + threadIdx.x : Ho * Wo);
    const int by = blockIdx.y + blockIdx.z * gridDim.y;
    const int nc0 = PACKED ? by * a.CPB : by;            // first (n,c) of this workgroup
    const int ncr = nc0 + cslot;
    const bool ok = (!PACKED || (cslot < a.CPB && ncr < a.NC)) && p < Ho * Wo;
    const int nc = (!PACKED || ncr < a.NC) ? ncr : a.NC - 1, c = nc % a.C;
    const int i = ok ? p / Wo : 0, j = ok ? p - i * Wo : 0;
    const bool i1 = i + 1 < Ho, j1 = j + 1 < Wo;
    const bool r1 = 2 * i + 1 < Hi;
    const float gsv = a.gs ? (float)a.gs[nc] : 0.0f;
    const float gqv = 2.0f * (float)a.gq[nc];
    float w[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) w[k] = a.w[c * 27 + k];
    const float pa = a.A[nc], pb2 = a.B[nc];
    const int po = cfn_uni(Ho * Wo), pi = cfn_uni(Hi * Wi);   // (hipcc merges Ho*Wo with the divergent definition of p)
    constexpr int OOB = 0x7ffffff0;
    const int nch = PACKED ? min(a.CPB, a.NC - nc0) : 1;
    __amdgpu_buffer_rsrc_t rgy = cfn_rsrc(const_cast<dwe_t*>(a.gy + (long)nc0 * T * po), (unsigned)((long)nch * T * po * DW_ES));
    __amdgpu_buffer_rsrc_t ry = cfn_rsrc(const_cast<dwe_t*>(a.y + (long)nc0 * T * po), (unsigned)((long)nch * T * po * DW_ES));
    __amdgpu_buffer_rsrc_t rx = cfn_rsrc(const_cast<dwe_t*>(a.x + (long)nc0 * T * pi), (unsigned)((long)nch * T * pi * DW_ES));
    __amdgpu_buffer_rsrc_t rgx = cfn_rsrc(a.gx + (long)nc0 * T * pi, (unsigned)((long)nch * T * pi * DW_ES));
    const int gb = (cslot * T * po + i * Wo + j) * DW_ES;
    const int go[4] = {ok ? gb : OOB, ok && j1 ? gb + DW_ES : OOB, ok && i1 ? gb + Wo * DW_ES : OOB, ok && i1 && j1 ? gb + (Wo + 1) * DW_ES : OOB};
    const int xb = (cslot * T * pi + 2 * i * Wi + 2 * j) * DW_ES;
    const int xo[2] = {ok ? xb : OOB, ok && r1 ? xb + Wi * DW_ES : OOB};
    const int t0 = chunk * a.TT, t1 = min(t0 + a.TT, T);

    typedef float __attribute__((ext_vector_type(2))) f2;
    typedef int __attribute__((ext_vector_type(2))) i2;
    auto ld_raw = [&](int f, float (&g)[4], float (&y)[4]) -> bool {
        const bool fv = f >= 0 && f < T;
        const int so = fv ? f * po * DW_ES : 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int vo = fv ? go[k] : OOB;
            g[k] = dw_bld1(rgy, vo, so);
            y[k] = dw_bld1(ry, vo, so);
        }
        return ok && fv;
    };
    auto fin = [&](bool v, const float (&g)[4], const float (&y)[4], float (&out)[4]) {
        const bool in[4] = {v, v && j1, v && i1, v && i1 && j1};
#pragma unroll
        for (int k = 0; k < 4; ++k) out[k] = in[k] ? fmaf(y[k], gqv, g[k] + gsv) : 0.0f;
    };
    auto ld_x = [&](int t, float (&xv)[4]) {
        const bool want = t < t1;
        const int so = want ? t * pi * DW_ES : 0;
        const f2 u = dw_bld2(rx, want ? xo[0] : OOB, so);
        const f2 d = dw_bld2(rx, want ? xo[1] : OOB, so);
        xv[0] = u.x; xv[1] = u.y; xv[2] = d.x; xv[3] = d.y;
    };
    float G[3][4], rg[4], ryv[4], xn[4];
    { const bool v = ld_raw(t0 - 1, rg, ryv); fin(v, rg, ryv, G[1]); }
    { const bool v = ld_raw(t0, rg, ryv); fin(v, rg, ryv, G[2]); }
    bool rv = ld_raw(t0 + 1, rg, ryv);
    ld_x(t0, xn);
    float s1 = 0.0f, s2 = 0.0f;
    const float m0 = ok ? 1.0f : 0.0f, m1 = ok && r1 ? 1.0f : 0.0f;      // statistics masks of the two rows
    for (int t = t0; t < t1; ++t) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { G[0][k] = G[1][k]; G[1][k] = G[2][k]; }
        fin(rv, rg, ryv, G[2]);
        float xv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) xv[k] = xn[k];
        rv = ld_raw(t + 2, rg, ryv);
        ld_x(t + 1, xn);
        float o00 = 0.f, o01 = 0.f, o10 = 0.f, o11 = 0.f;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            const float* g = G[2 - kt];
            const float* wk = w + kt * 9;
            o00 = fmaf(wk[4], g[0], o00);
            o01 = fmaf(wk[3], g[1], fmaf(wk[5], g[0], o01));
            o10 = fmaf(wk[1], g[2], fmaf(wk[7], g[0], o10));
            o11 = fmaf(wk[0], g[3], fmaf(wk[2], g[2], fmaf(wk[6], g[1], fmaf(wk[8], g[0], o11))));
        }
        float v[4] = {o00, o01, o10, o11};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float dz = v[k] * cfn_act_grad_rt(fmaf(xv[k], pa, pb2), a.act) * (k < 2 ? m0 : m1);
            s1 = fmaf(dz, xv[k], s1);
            s2 += dz;
            v[k] = dz * pa;
        }
        const int so = t * pi * DW_ES;
        dw_bst2((f2){v[0], v[1]}, rgx, xo[0], so);
        dw_bst2((f2){v[2], v[3]}, rgx, xo[1], so);
    }
    if (a.gA) {
        __shared__ float sh[8];
        const int lane = threadIdx.x & 63;
        if (PACKED) {
            const int key = ok ? cslot : -1 - (int)(threadIdx.x >> 6);
            const int prev_key = __shfl_up(key, 1, 64);
            const bool head = ok && (lane == 0 || prev_key != key);
            const float q1 = seg_wave_sum(s1, key, lane);
            const float q2 = seg_wave_sum(s2, key, lane);
            if (head) {
                cfn_add64(&a.gA[nc], (double)q1);
                cfn_add64(&a.gB[nc], (double)q2);
            }
        } else {
            s1 = cfn_wave_sum(s1); s2 = cfn_wave_sum(s2);
            if (lane == 0) { sh[threadIdx.x >> 6] = s1; sh[4 + (threadIdx.x >> 6)] = s2; }
            __syncthreads();
            if (threadIdx.x == 0) {
                cfn_add64(&a.gA[nc], (double)(sh[0] + sh[1] + sh[2] + sh[3]));
                cfn_add64(&a.gB[nc], (double)(sh[4] + sh[5] + sh[6] + sh[7]));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side: geometry heuristics + dispatch
// ---------------------------------------------------------------------------------------------
struct DwPlan { int HS, VEC, MAXLD, threads; bool UNIW; size_t lds; unsigned blocks; };

static thread_local int g_force_hs = 0;   // fused backward: explicit strip height
static int pick_hs(int Ho, int mode, int S) {
    if (g_force_hs > 0 && Ho % g_force_hs == 0) return g_force_hs;
    // stride 2 (forward / weight gradient): one output row per thread.  A 7-row strip needs a 15-row input window per
    // thread and leaves too few threads per plane (measured at 4 clips: 112->56 fwd 1.15 -> 0.67 ms, wgrad 1.17 -> 0.77)
    if (S == 2) return 1;
    // small planes (<= 14 rows), data gradient: short strips keep the register footprint low (more waves per CU;
    // measured faster); everything else: 7-row strips minimise LDS reads per output
    if (Ho <= 14 && mode == DW_DGRAD) return (Ho % 2 == 0) ? 2 : 1;
    if (Ho % 7 == 0) return 7;
    if (Ho % 4 == 0) return 4;
    if (Ho % 2 == 0) return 2;
    return 1;
}

static int dw_plan_impl(DwArgs& a, int S, int mode, DwPlan& pl, bool allow_flat) {
    a.Ho = (a.Hi + 2 - 3) / S + 1;
    a.Wo = (a.Wi + 2 - 3) / S + 1;
    const int HS = pick_hs(a.Ho, mode, S);
    const int G = a.Ho / HS;
    if (a.Wo > 512) return cfn_fail(CFN_ERR_UNSUPPORTED, "dwconv3d: output width %d > 512 not supported", a.Wo);
    // loader: whole-float4 rows (4); else, when the width is even and the plane a whole number of float4s, float4s over
    // the contiguous plane (2, needs the band to cover the plane); else scalar (1)
    const bool aligned16 = ((uintptr_t)a.src % 16 == 0) && (mode != DW_DGRAD || a.src2 == nullptr || (uintptr_t)a.src2 % 16 == 0);
    int VEC = (a.Wi % 4 == 0) ? 4 : ((allow_flat && a.Wi % 2 == 0 && (a.Hi * a.Wi) % 4 == 0 && aligned16) ? 2 : 1);
    int GB;
    for (;; ) {
        const int LVh = VEC == 1 ? 1 : 4;
        a.XO = VEC == 4 ? 4 : (VEC == 2 ? 2 : 1);
        a.WP = VEC == 4 ? a.Wi + 8 : (VEC == 2 ? a.Wi + 4 : a.Wi + 2);
        // rows per band: largest divisor GB of G with GB*Wo <= 512 threads, LDS <= 64 KiB, loader capacity
        GB = G;
        for (;; ) {
            while (G % GB) --GB;
            const int rin = (GB * HS - 1) * S + 3;
            const long ipcb = (long)GB * a.Wo;
            const long thr = (ipcb + 63) / 64 * 64;
            const bool fits = ipcb <= 256 && (long)rin * a.WP * 8 <= 60 * 1024 && (long)rin * a.Wi <= (long)LVh * 8 * thr;
            if (fits || GB == 1) break;
            --GB;
        }
        if (VEC == 2 && GB != G) { VEC = 1; continue; }   // the flat loader needs the whole plane in one band
        break;
    }
    const int LVh = VEC == 1 ? 1 : 4;
    a.GB = GB;
    a.nbands = G / GB;
    a.RIN = (GB * HS - 1) * S + 3;
    a.IPCb = GB * a.Wo;
    if (a.IPCb > 512 || (long)a.RIN * a.WP * 8 > 150 * 1024)
        return cfn_fail(CFN_ERR_UNSUPPORTED, "dwconv3d: plane %dx%d does not fit the LDS band scheme", a.Hi, a.Wi);
    // thread slots per channel: pad to whole waves when that costs <= 15 % idle lanes (=> wave-uniform channel,
    // weights in SGPRs)
    const int ipc64 = (a.IPCb + 63) / 64 * 64;
    pl.UNIW = a.IPCb >= 64 && ipc64 <= 256 && (ipc64 - a.IPCb) * 100 <= 15 * ipc64;
    a.IPCp = pl.UNIW ? ipc64 : a.IPCb;
    // UNIW: 4-wave workgroups (several fit on a CU, barriers stay cheap)
    // small planes: up to 8 waves; the data gradient (most registers, two staged tensors) runs better as 4-wave
    // workgroups, several of which fit on a CU (measured at 4 clips: 14x14 0.31 -> 0.25 ms, 7x7 0.25 -> 0.16 ms)
    // 4-wave workgroups (two per CU at this variant's register budget) beat one 8-wave workgroup everywhere except the
    // 7x7 forward (measured, 8 clips: 56->28 s2 fwd 0.80 -> 0.70 ms, 28->14 s2 0.42 -> 0.35, 14x14 wgrad 0.47 -> 0.45)
    const int wg = (pl.UNIW || mode != DW_FWD || a.Ho * a.Wo > 64 || S == 2) ? 256 : 512;
    int CG = wg / a.IPCp;
    if (CG < 1) CG = 1;
    if (CG > a.C) CG = a.C;
    if (CG > 128) CG = 128;
    while (CG > 1 && ((long)CG * a.RIN * a.WP * 8 > 48 * 1024)) --CG;
    int threads;
    for (;; --CG) {   // loader capacity: LV*8 elements per thread per frame
        threads = (CG * a.IPCp + 63) / 64 * 64;
        if ((long)CG * a.RIN * a.Wi <= (long)LVh * 8 * threads || CG == 1) break;
    }
    if ((long)CG * a.RIN * a.Wi > (long)LVh * 8 * threads)
        return cfn_fail(CFN_ERR_UNSUPPORTED, "dwconv3d: loader capacity exceeded for plane %dx%d", a.Hi, a.Wi);
    if ((long)CG * a.T * a.Hi * a.Wi * DW_ES >= 0x7ffffff0L)
        return cfn_fail(CFN_ERR_UNSUPPORTED, "dwconv3d: %d channels x %d frames of a %dx%d plane exceed the 2 GiB buffer range", CG, a.T, a.Hi, a.Wi);
    a.CG = CG;
    a.ngroups = cfn_cdiv(a.C, CG);
    const long per_thread = ((long)CG * a.RIN * a.Wi + (long)LVh * threads - 1) / ((long)LVh * threads);
    const int MAXLD = (LVh == 4 && per_thread <= 2) ? 2 : (per_thread <= 4 ? 4 : 8);
    // frames per chunk: as long as possible while keeping >= ~6 workgroups per CU in the grid
    const long planes = (long)a.N * a.ngroups * a.nbands;
    // Whole rounds: resident workgroups per CU follow from the register budget of the variant (launch bounds),
    // so size the t-chunks such that the grid fills R full rounds of the chip (a 1.1-round grid costs 2 rounds).
    const int per_cu = !pl.UNIW ? (threads > 256 ? 1 : (mode == DW_DGRAD ? 4 : 2)) : (MAXLD == 8 ? 2 : (MAXLD == 4 ? 3 : (mode == DW_FWD ? 4 : 3)));
    const long slots = 256L * per_cu;
    int TT = a.T;
    for (int R = 1; R <= 8; ++R) {
        long nch = slots * R / planes;
        if (nch < 1) continue;
        if (nch > a.T) nch = a.T;
        TT = cfn_cdiv(a.T, nch);
        if (TT <= 40) break;
    }
    if (TT < 8) TT = 8;
    if (TT > a.T) TT = a.T;
    a.TT = TT;
    a.nchunks = cfn_cdiv(a.T, TT);
    pl.HS = HS; pl.VEC = VEC; pl.MAXLD = MAXLD; pl.threads = threads;
    pl.lds = ((size_t)2 * CG * a.RIN * a.WP + 2 * CG + (mode == DW_WGRAD ? 27 : 2) * CG * (threads / 64)) * sizeof(float);
    pl.blocks = (unsigned)(planes * a.nchunks);
    {   // CFN_DW_PLAN_DEBUG=1: print the launch plan (geometry audits)
        static const int dbg = getenv("CFN_DW_PLAN_DEBUG") ? atoi(getenv("CFN_DW_PLAN_DEBUG")) : 0;
        if (dbg) fprintf(stderr, "dw_plan mode %d S %d C %d %dx%d: HS %d VEC %d MAXLD %d UNIW %d thr %d CG %d TT %d chunks %d blocks %u per_cu %d lds %zu\n",
                         mode, S, a.C, a.Hi, a.Wi, HS, VEC, MAXLD, (int)pl.UNIW, threads, CG, TT, a.nchunks, pl.blocks, per_cu, pl.lds);
    }
    return CFN_OK;
}

static int dw_plan(DwArgs& a, int S, int mode, DwPlan& pl) {
    int rc = dw_plan_impl(a, S, mode, pl, true);
    // the flat loader exists for per-lane channels and <= 4 loads per thread only
    if (rc == CFN_OK && pl.VEC == 2 && (pl.UNIW || pl.MAXLD == 8)) rc = dw_plan_impl(a, S, mode, pl, false);
    return rc;
}

template <int MODE, int S, int HS>
static int dw_launch_hs(const DwArgs& a, const DwPlan& pl, hipStream_t st) {
#define CFN_DW_GO(VEC, MAXLD, UW)                                                                                      \
    do {                                                                                                               \
        auto k = dw3d_kernel<MODE, S, HS, VEC, MAXLD, UW>;                                                             \
        if (pl.lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds); \
        hipLaunchKernelGGL(k, dim3(pl.blocks), dim3(pl.threads), pl.lds, st, a);                                       \
    } while (0)
    if (pl.UNIW) {
        if (pl.VEC == 4 && pl.MAXLD == 2) CFN_DW_GO(4, 2, true);
        else if (pl.VEC == 4 && pl.MAXLD == 4) CFN_DW_GO(4, 4, true);
        else if (pl.VEC == 4) CFN_DW_GO(4, 8, true);
        else if (pl.MAXLD == 4) CFN_DW_GO(1, 4, true);
        else CFN_DW_GO(1, 8, true);
    } else {
        if (pl.VEC == 2 && pl.MAXLD == 2) CFN_DW_GO(2, 2, false);
        else if (pl.VEC == 2) CFN_DW_GO(2, 4, false);
        else if (pl.VEC == 4 && pl.MAXLD == 2) CFN_DW_GO(4, 2, false);
        else if (pl.VEC == 4 && pl.MAXLD == 4) CFN_DW_GO(4, 4, false);
        else if (pl.VEC == 4) CFN_DW_GO(4, 8, false);
        else if (pl.MAXLD == 4) CFN_DW_GO(1, 4, false);
        else CFN_DW_GO(1, 8, false);
    }
#undef CFN_DW_GO
    return cfn_check_launch("dwconv3d");
}

template <int MODE, int S>
static int dw_launch(const DwArgs& a, const DwPlan& pl, hipStream_t st) {
    switch (pl.HS) {
        case 7: return dw_launch_hs<MODE, S, 7>(a, pl, st);
        case 4: return dw_launch_hs<MODE, S, 4>(a, pl, st);
        case 2: return dw_launch_hs<MODE, S, 2>(a, pl, st);
        default: return dw_launch_hs<MODE, S, 1>(a, pl, st);
    }
}

static double dw_bytes(const DwArgs& a, int tensors_in, int tensors_out) {
    return (double)DW_ES * a.N * a.C * a.T * ((double)tensors_in * a.Hi * a.Wi + (double)tensors_out * a.Ho * a.Wo);
}

extern "C" int DWN(cfn_dwconv3d_fwd)(const dwe_t* x, const double* A, const double* B, int act, const float* w, dwe_t* y,
                                double* sum, double* sumsq, int N, int C, int T, int Hi, int Wi, int stride,
                                void* stream) {
    CFN_REQUIRE(x && w && y, "cfn_dwconv3d_fwd: null tensor");
    CFN_REQUIRE(N > 0 && C > 0 && T > 0 && Hi > 0 && Wi > 0, "cfn_dwconv3d_fwd: bad shape");
    CFN_REQUIRE(stride == 1 || stride == 2, "cfn_dwconv3d_fwd: stride must be 1 or 2 (got %d)", stride);
    CFN_REQUIRE((A == nullptr) == (B == nullptr), "cfn_dwconv3d_fwd: A and B must both be given or both null");
    CFN_REQUIRE((sum == nullptr) == (sumsq == nullptr), "cfn_dwconv3d_fwd: sum and sumsq go together");
    DwArgs a = {};
    a.src = x; a.A = A; a.B = B; a.act = act; a.w = w; a.dst = y; a.s1 = sum; a.s2 = sumsq;
    a.N = N; a.C = C; a.T = T; a.Hi = Hi; a.Wi = Wi;
    hipStream_t st = (hipStream_t)stream;
#ifndef DW_BF16
    if (dw_flat_fwd_try(x, A, B, act, w, y, sum, sumsq, N, C, T, Hi, Wi, stride, st, true) == 0) {
        // output planes 56x56 / 28x28 / 14x14, stride 1 and 2, fp32: flat kernels, every load of a work item up front (dwflat.hip)
        const double po_ = stride == 1 ? (double)Hi * Wi : ((Hi - 1) / 2 + 1.0) * ((Wi - 1) / 2 + 1.0);
        CfnProfScope prof(CFN_K_DWCONV_FWD, st, (double)DW_ES * N * C * T * ((double)Hi * Wi + po_) + 4.0 * C * 27);
        return dw_flat_fwd_try(x, A, B, act, w, y, sum, sumsq, N, C, T, Hi, Wi, stride, st, false);
    }
#endif
    if (DWN(dw_cp_fwd_try)(x, A, B, act, w, y, sum, sumsq, N, C, T, Hi, Wi, stride, st, true) == 0) {
        // output planes 56x56 / 28x28 / 14x14, stride 1 and 2: column-pair wave kernel (dwcp.hip)
        const double po_ = stride == 1 ? (double)Hi * Wi : ((Hi - 1) / 2 + 1.0) * ((Wi - 1) / 2 + 1.0);
        CfnProfScope prof(CFN_K_DWCONV_FWD, st, (double)DW_ES * N * C * T * ((double)Hi * Wi + po_) + 4.0 * C * 27);
        return DWN(dw_cp_fwd_try)(x, A, B, act, w, y, sum, sumsq, N, C, T, Hi, Wi, stride, st, false);
    }
    if (DWN(dw_small_fwd_try)(x, A, B, act, w, y, sum, sumsq, N, C, T, Hi, Wi, stride, st, true) == 0) {
        // 14x14 / 7x7 stride 1: wave-per-channel kernel (dwsmall.hip)
        CfnProfScope prof(CFN_K_DWCONV_FWD, st, (double)DW_ES * N * C * T * 2.0 * Hi * Wi + 4.0 * C * 27);
        return DWN(dw_small_fwd_try)(x, A, B, act, w, y, sum, sumsq, N, C, T, Hi, Wi, stride, st, false);
    }
    DwPlan pl;
    int rc = dw_plan(a, stride, DW_FWD, pl);
    if (rc) return rc;
    CfnProfScope prof(CFN_K_DWCONV_FWD, st, (double)DW_ES * N * C * T * ((double)Hi * Wi + (double)a.Ho * a.Wo) + 4.0 * C * 27);
    return stride == 1 ? dw_launch<DW_FWD, 1>(a, pl, st) : dw_launch<DW_FWD, 2>(a, pl, st);
}

extern "C" int DWN(cfn_dwconv3d_bwd_data)(const dwe_t* gy, const dwe_t* y, const double* gsum, const double* gsumsq,
                                     const float* w, const dwe_t* x, const double* A, const double* B, int act,
                                     dwe_t* gx, double* gA, double* gB, int N, int C, int T, int Hi, int Wi,
                                     int stride, void* stream) {
    CFN_REQUIRE(gy && w && gx, "cfn_dwconv3d_bwd_data: null tensor");
    CFN_REQUIRE(stride == 1 || stride == 2, "cfn_dwconv3d_bwd_data: stride must be 1 or 2");
    CFN_REQUIRE((A == nullptr) == (B == nullptr), "cfn_dwconv3d_bwd_data: A/B mismatch");
    CFN_REQUIRE(A == nullptr || (x != nullptr && gA != nullptr && gB != nullptr), "cfn_dwconv3d_bwd_data: prologue needs x, gA, gB");
    CFN_REQUIRE(gsumsq == nullptr || y != nullptr, "cfn_dwconv3d_bwd_data: gsumsq needs y");
    hipStream_t st = (hipStream_t)stream;
    const int Ho = (Hi + 2 - 3) / stride + 1, Wo = (Wi + 2 - 3) / stride + 1;
    CfnProfScope prof(CFN_K_DWCONV_BWD, st, 4.0 * N * C * T * ((double)Hi * Wi * (A ? 2 : 1) + (double)Ho * Wo * (y ? 2 : 1)));
    if (stride == 2) {
        DwS2Args a = {};
        a.gy = gy; a.y = gsumsq ? y : nullptr; a.gs = gsum; a.gq = gsumsq; a.w = w; a.x = x; a.A = A; a.B = B; a.gx = gx;
        a.gA = gA; a.gB = gB; a.C = C; a.T = T; a.Hi = Hi; a.Wi = Wi; a.Ho = Ho; a.Wo = Wo; a.act = act;
        a.NC = N * C;
        a.PB = Ho * Wo;
        a.CPB = a.PB <= 128 ? 256 / a.PB : 1;
        a.pblocks = cfn_cdiv((long)Ho * Wo, 256);
        // even split of the plane: 28x28 = 784 positions are 4 x 196, not 3 x 256 + a 16-thread straggler that streams all
        // frames for 2 % of the work
        a.PBLK = (a.PB % a.pblocks == 0 && (a.PB / a.pblocks) % 4 == 0) ? a.PB / a.pblocks : 256;   // (uneven splits measured slower)
        const int ygrid = cfn_cdiv(a.NC, a.CPB);
        unsigned gy_, gz_;
        CFN_REQUIRE(cfn_split_nc(ygrid, gy_, gz_), "cfn_dwconv3d_bwd_data: N*C exceeds grid.y");
        int TT = 64;
        while (TT > 8 && (long)ygrid * a.pblocks * cfn_cdiv(T, TT) < 2048) TT >>= 1;
        a.TT = TT > T ? T : TT;
        a.nchunks = cfn_cdiv(T, a.TT);
        const bool fast = (Wi % 2 == 0) && A && a.y && a.gq && gA && (long)a.CPB * T * Hi * Wi * DW_ES < 0x7ffffff0L;
        const dim3 grid(a.pblocks * a.nchunks, gy_, gz_);
        if (fast && a.CPB > 1) hipLaunchKernelGGL(dw3d_dgrad_s2_fast_kernel<true>, grid, dim3(256), 0, st, a);
        else if (fast) hipLaunchKernelGGL(dw3d_dgrad_s2_fast_kernel<false>, grid, dim3(256), 0, st, a);
        else if (a.CPB > 1) hipLaunchKernelGGL(dw3d_dgrad_s2_kernel<true>, grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(dw3d_dgrad_s2_kernel<false>, grid, dim3(256), 0, st, a);
        return cfn_check_launch("dwconv3d_bwd_data_s2");
    }
    DwArgs a = {};
    a.src = gy; a.src2 = y; a.gs = gsum; a.gq = gsumsq; a.w = w; a.xin = x; a.A = A; a.B = B; a.act = act;
    a.dst = gx; a.s1 = A ? gA : nullptr; a.s2 = A ? gB : nullptr;
    a.N = N; a.C = C; a.T = T; a.Hi = Hi; a.Wi = Wi;
    DwPlan pl;
    int rc = dw_plan(a, 1, DW_DGRAD, pl);
    if (rc) return rc;
    return dw_launch<DW_DGRAD, 1>(a, pl, st);
}

extern "C" int DWN(cfn_dwconv3d_bwd_weight)(const dwe_t* gy, const dwe_t* y, const double* gsum, const double* gsumsq,
                                       const dwe_t* x, const double* A, const double* B, int act, double* gw, int N,
                                       int C, int T, int Hi, int Wi, int stride, void* stream) {
    CFN_REQUIRE(gy && x && gw, "cfn_dwconv3d_bwd_weight: null tensor");
    CFN_REQUIRE(stride == 1 || stride == 2, "cfn_dwconv3d_bwd_weight: stride must be 1 or 2");
    CFN_REQUIRE((A == nullptr) == (B == nullptr), "cfn_dwconv3d_bwd_weight: A/B mismatch");
    CFN_REQUIRE(gsumsq == nullptr || y != nullptr, "cfn_dwconv3d_bwd_weight: gsumsq needs y");
    DwArgs a = {};
    a.src = x; a.A = A; a.B = B; a.act = act; a.gy = gy; a.yout = gsumsq ? y : nullptr; a.gs = gsum; a.gq = gsumsq;
    a.s1 = gw;
    a.N = N; a.C = C; a.T = T; a.Hi = Hi; a.Wi = Wi;
    DwPlan pl;
    int rc = dw_plan(a, stride, DW_WGRAD, pl);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_DWCONV_WGRAD, st, 4.0 * N * C * T * ((double)Hi * Wi + (double)a.Ho * a.Wo * (a.yout ? 2 : 1)));
    return stride == 1 ? dw_launch<DW_WGRAD, 1>(a, pl, st) : dw_launch<DW_WGRAD, 2>(a, pl, st);
}

// fused data + weight gradient (stride 1, big planes: wave-uniform channels, float4 rows); -1 = use the two kernels
template <int HS>
static int dwf_launch_hs(const DwFusedArgs& f, const DwPlan& pl, size_t lds, hipStream_t st) {
#define CFN_DWF_GO(VECV, ML, UW)                                                                                       \
    do {                                                                                                               \
        auto k = dw3d_bwd_fused_kernel<HS, VECV, ML, UW>;                                                              \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(k, dim3(pl.blocks), dim3(pl.threads), lds, st, f);                                          \
    } while (0)
    if (pl.MAXLD == 2) CFN_DWF_GO(4, 2, true); else CFN_DWF_GO(4, 4, true);   // measured: no gain on 14x14 / 7x7 planes
#undef CFN_DWF_GO
    return cfn_check_launch("dwconv3d_bwd_fused");
}

extern "C" int DWN(cfn_dwconv3d_bwd_fused)(const dwe_t* gy, const dwe_t* y, const double* gsum, const double* gsumsq,
                                      const float* w, const dwe_t* x, const double* A, const double* B, int act, dwe_t* gx,
                                      double* gA, double* gB, double* gw, int N, int C, int T, int H, int W, void* stream) {
    CFN_REQUIRE(gy && w && x && gx && gw, "cfn_dwconv3d_bwd_fused: null tensor");
    CFN_REQUIRE((A == nullptr) == (B == nullptr), "cfn_dwconv3d_bwd_fused: A/B mismatch");
    CFN_REQUIRE(A == nullptr || (gA != nullptr && gB != nullptr), "cfn_dwconv3d_bwd_fused: prologue needs gA, gB");
    CFN_REQUIRE(gsumsq == nullptr || y != nullptr, "cfn_dwconv3d_bwd_fused: gsumsq needs y");
    if (DWN(dw_flatb_try)(gy, y, gsum, gsumsq, w, x, A, B, act, gx, gA, gB, gw, N, C, T, H, W, (hipStream_t)stream, true) == 0) {
        // 7x7: flat wave kernel, a lane per position (dwflatb.hip)
        CfnProfScope prof(CFN_K_DWCONV_BWD, (hipStream_t)stream, (double)DW_ES * N * C * T * (double)H * W * (y ? 4 : 3));
        return DWN(dw_flatb_try)(gy, y, gsum, gsumsq, w, x, A, B, act, gx, gA, gB, gw, N, C, T, H, W, (hipStream_t)stream, false);
    }
    if (DWN(dw_cpbx_try)(gy, y, gsum, gsumsq, w, x, A, B, act, gx, gA, gB, gw, N, C, T, H, W, (hipStream_t)stream, true) == 0) {
        // 56x56 / 28x28 / 14x14: column-pair wave kernel with one LDS image, x / a in registers (dwcpbx.hip)
        CfnProfScope prof(CFN_K_DWCONV_BWD, (hipStream_t)stream, (double)DW_ES * N * C * T * (double)H * W * (y ? 4 : 3));
        return DWN(dw_cpbx_try)(gy, y, gsum, gsumsq, w, x, A, B, act, gx, gA, gB, gw, N, C, T, H, W, (hipStream_t)stream, false);
    }
    if (DWN(dw_cpb_try)(gy, y, gsum, gsumsq, w, x, A, B, act, gx, gA, gB, gw, N, C, T, H, W, (hipStream_t)stream, true) == 0) {
        // 56x56 / 28x28 / 14x14: column-pair wave kernel (dwcpb.hip)
        CfnProfScope prof(CFN_K_DWCONV_BWD, (hipStream_t)stream, (double)DW_ES * N * C * T * (double)H * W * (y ? 4 : 3));
        return DWN(dw_cpb_try)(gy, y, gsum, gsumsq, w, x, A, B, act, gx, gA, gB, gw, N, C, T, H, W, (hipStream_t)stream, false);
    }
    DwArgs a = {};
    a.N = N; a.C = C; a.T = T; a.Hi = H; a.Wi = W;
    DwPlan pl;
    // 4-row strips first (164 VGPRs -> 3 waves per SIMD; measured 56x56: 1.92 ms against 2.18 ms with 7 rows and 2.35 ms
    // for the two separate kernels), 7-row strips when 4 rows do not give wave-uniform channels (28x28)
    int rc = CFN_ERR_UNSUPPORTED;
    bool ok = false;
    for (int hs : {4, 0}) {
        g_force_hs = hs;
        rc = dw_plan(a, 1, DW_WGRAD, pl);
        g_force_hs = 0;
        if (rc == CFN_OK && pl.UNIW && pl.VEC == 4 && pl.MAXLD != 8 && (pl.HS == 4 || pl.HS == 7)) { ok = true; break; }
    }
    if (!ok) return -1;                            // small planes keep the two separate kernels
    DwFusedArgs f = {gy, gsumsq ? y : nullptr, gsum, gsumsq, w, x, A, B, gx, A ? gA : nullptr, A ? gB : nullptr, gw,
                     N, C, T, H, W, act};
    f.TT = a.TT; f.nchunks = a.nchunks; f.CG = a.CG; f.ngroups = a.ngroups; f.GB = a.GB; f.nbands = a.nbands;
    f.IPCb = a.IPCb; f.IPCp = a.IPCp; f.RIN = a.RIN; f.WP = a.WP; f.XO = a.XO;
    const size_t lds = ((size_t)4 * a.CG * a.RIN * a.WP + 4 * a.CG + 27 * a.CG * (pl.threads / 64)) * sizeof(float);
    if (lds > 150 * 1024) return -1;
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_DWCONV_BWD, st, (double)DW_ES * N * C * T * (double)H * W * (y ? 4 : 3));
    switch (pl.HS) {
        case 7: return dwf_launch_hs<7>(f, pl, lds, st);
        case 4: return dwf_launch_hs<4>(f, pl, lds, st);
        default: return -1;
    }
}
