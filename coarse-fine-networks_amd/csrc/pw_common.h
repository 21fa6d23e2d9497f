// Shared declarations of the pointwise-contraction kernels (pwconv.hip, pwdeep.hip).
#pragma once
#include "cfn_common.h"

typedef float __attribute__((ext_vector_type(16))) f16v;
typedef float __attribute__((ext_vector_type(4))) f4v;

enum { PW_FWD = 0, PW_DGRAD = 1 };
#define PW_KC 32
#define PW_RED_PITCH 33

struct PwArgs {
    const float* src;    // FWD: x raw (N,K,Pin)          DGRAD: gy (N,K,Q)
    const float* src2;   // DGRAD: y raw (N,K,Q) for the 2*y*gq term (may be null)
    const double* pa;     // FWD: prologue A[n,k] (null = identity)
    const double* pb;
    const double* gs;    // DGRAD: d/d sum(y)   [n,k]  (may be null)
    const double* gq;    // DGRAD: d/d sum(y^2) [n,k]  (may be null)
    const double* gsc;   // DGRAD: per-(n,k) scale of the incoming gradient: g' = gsc*gy + gs + 2*y*gq  (null = 1)
    const float* w;      // (Cout, Cin) row major
    float* dst;          // FWD: y (N,M,Q)                DGRAD: gx (N,M,Pin)
    const float* ex;     // DGRAD: forward input x raw (N,M,Pin) (needed when ea != null)
    const double* ea;     // DGRAD: forward prologue A[n,m] (null = identity => gx = da)
    const double* eb;
    // DGRAD: optional compact gradient (N, M, T, acc_Ho, acc_Wo) of a second, spatially strided consumer of the same
    // input (the stride-s shortcut conv of a stage's first block): added to W^T g' on the lattice h % s == w % s == 0
    // BEFORE the act' epilogue, so the zero-filled sparse tensor and the dense add kernel never exist
    const float* acc; int acc_s, acc_Ho, acc_Wo;
    double* s1;          // FWD: sum(y) [n,m]             DGRAD: sum(dz*x) [n,m]
    double* s2;          // FWD: sum(y^2)                 DGRAD: sum(dz)
    int N, M, K, Q, Pin, Hi, Wi, Ho, Wo, stride, act;
    int Cin;             // row pitch of w
    int mtiles, nstrips, tpb, kres, Kpad;   // kres: weight rows resident in LDS per pass (multiple of 8)
    // stem != 0: dense convolution as an implicit GEMM -- the B operand is the im2col view of a (N,Cimg,Ti,Hi,Wi)
    // tensor for a (kT,kH,kW) kernel with strides (sT,sH,sW) and zero padding (pT,pH,pW); K = Cimg*kT*kH*kW
    int stem, Cimg, kT, kH, kW, sT, sH, sW, pT, pH, pW, Ti, To;
};

#define PW_UNIT 8        // input channels per pipelined unit (4 MFMA k-steps)
#define PW_KRES_MAX 512  // weight rows kept in LDS at once (K > 512 streams the weights in chunks)

__device__ __forceinline__ float pw_bload(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}


// pwdeep.hip: returns -1 when the shape is not handled by the deep kernel (caller falls through to pw_gemm_kernel),
// otherwise the launch status.  mode: PW_FWD / PW_DGRAD; stats: epilogue statistics / act' epilogue present.
int pwd_try_launch(PwArgs& a, int mode, bool stats, hipStream_t st);

// pwsplit.hip / pwsplitw.hip: the same contractions with split-bf16 arithmetic (fp32 tensors, operands split into 2-3 bf16
// terms on load, 3 or 6 bf16 MFMAs per k-block); tried first, -1 = shape not handled or the split is switched off
int pws_try_launch(PwArgs& a, int mode, bool stats, hipStream_t st);
// pwregk.hip: split-bf16 forward with register-resident weights, two k slices per row tile (128 < K <= 224, M <= 128); tried first
int pwk_try_launch(PwArgs& a, int mode, bool stats, hipStream_t st);
// pwstream.hip: pws_kernel with the weights pre-split into a workspace and STREAMED through LDS (K >= 400: layer 4); tried after pwk_try_launch
int pwt_try_launch(PwArgs& a, int mode, bool stats, hipStream_t st);
int pws_terms_now();     // 0 = fp32 MFMA kernels, 3 / 6 = bf16 MFMAs per k-block (cfn_pw_split_terms / CFN_PW_SPLIT)
int pws_wgrad_try_launch(const float* gy, const float* y, const double* gs, const double* gq, const double* gsc, const float* x,
                         const double* pa, const double* pb, int act, double* gw, int N, int M, int K, int Q, hipStream_t st);

// the LDS-staged weight gradient of pwsplitw.hip on bf16 tensors (one bf16 term per operand); -1 = not handled
int pwss_wgrad_try_bf16(const uint16_t* gy, const uint16_t* y, const double* gs, const double* gq, const double* gsc, const uint16_t* x,
                        const double* pa, const double* pb, int act, double* gw, int N, int M, int K, int Q, hipStream_t st);

// pwwgrad.hip: direct-operand weight gradient for M, K >= 48 (stride 1); -1 = shape not handled.
// gsc: per-(n,m) scale of gy (null = 1)
int pwd_wgrad_try_launch(const float* gy, const float* y, const double* gs, const double* gq, const double* gsc, const float* x,
                         const double* pa, const double* pb, int act, double* gw, int N, int M, int K, int Q, hipStream_t st);
int pwd_wgrad_try_strided(const float* gy, const float* y, const double* gs, const double* gq, const double* gsc, const float* x,
                          const double* pa, const double* pb, int act, double* gw, int N, int M, int K, int T, int Hi, int Wi,
                          int stride, hipStream_t st);
int pwd_wgrad_try_dense(const float* gy, const float* y, const double* gs, const double* gq, const float* x, const double* pa,
                        const double* pb, int act, double* gw, int N, int M, int Cimg, int T, int Hi, int Wi, const int* g,
                        hipStream_t st);

// stem.hip: LDS-tiled forward of the 1x3x3 stride-(1,2,2) stem conv (Cimg == 3, Cout <= 32, Wi % 4 == 0, Hi even); -1 = not handled
int stem_fwd_try_launch(const float* x, const float* w, float* y, int N, int Cimg, int Cout, int T, int Hi, int Wi, hipStream_t st);
int stem_wgrad_try_launch(const float* gy, const float* x, double* gw, int N, int Cimg, int Cout, int T, int Hi, int Wi, hipStream_t st, bool probe);

// salconv.hip: LDS-tiled fp32-MFMA kernels for the Grid Pool saliency convs (Cin == 24, Cout <= 32, 3x3x3, stride 2, pad 1, input
// planes 56 or 28 wide, even height); -1 = not handled
int sal_fwd_try_launch(const float* x, const double* A, const double* B, int act, const float* w, float* y, double* sum,
                       double* sumsq, int N, int Cin, int Cout, int T, int Hi, int Wi, const int* g, hipStream_t st);
// salconvb.hip: the forward of the same shapes on the split-bf16 matrix pipe (6 bf16 MFMAs per k-block, fp32-accurate); -1 = not handled / switched off
// pwfuseds.hip: one-pass (data + weight gradient) backward of the layer-2 pointwise convs on the split-bf16 matrix pipe; -1 = not handled
int pwfs_try_launch(const float* gy, const float* y, const double* gsum, const double* gsumsq, const float* w, const float* x, const double* A,
                    const double* B, int act, float* gx, double* gA, double* gB, double* gw, int N, int Cin, int Cout, int T, int Hi, int Wi,
                    const float* acc, int acc_stride, const double* gscale, hipStream_t st);
int salb_fwd_try_launch(const float* x, const double* A, const double* B, int act, const float* w, float* y, double* sum,
                        double* sumsq, int N, int Cin, int Cout, int T, int Hi, int Wi, const int* g, hipStream_t st);
int sal_wgrad_try_launch(const float* gy, const float* y, const double* gs, const double* gq, const float* x, const double* A,
                         const double* B, int act, double* gw, int N, int Cin, int Cout, int T, int Hi, int Wi, const int* g,
                         hipStream_t st);
