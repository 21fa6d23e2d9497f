// bf16-storage variants of the streaming kernels around the convolutions (block tail, spatial pooling): same contracts
// as elementwise.hip, tensors in HBM are bf16, all arithmetic and every per-(n,c) reduction is fp32 / fp64.
// A thread moves 16 bytes (8 elements) per access.
#include "cfn_common.h"
#include <stdint.h>

typedef unsigned u4v __attribute__((ext_vector_type(4)));
// element kind of the 2-byte tensors: bf16, or IEEE half when compiled through streamf16.hip (h16.h)
#include "h16.h"
#define bn_add_relu_fwd_bf16_kernel H16N(bn_add_relu_fwd_h_kernel)
#define bn_add_relu_bwd_g_bf16_kernel H16N(bn_add_relu_bwd_g_h_kernel)
#define pool_hw_fwd_bf16_kernel H16N(pool_hw_fwd_h_kernel)
#define pool_hw_bwd_bf16_kernel H16N(pool_hw_bwd_h_kernel)
typedef float f2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float sb_lo(unsigned u) { return h16_lo(u); }
__device__ __forceinline__ float sb_hi(unsigned u) { return h16_hi(u); }
__device__ __forceinline__ unsigned sb_pack(float lo, float hi) { return h16_pk(lo, hi); }
__device__ __forceinline__ float sb_ld1(const uint16_t* p) { return h16_lo((unsigned)p[0]); }
__device__ __forceinline__ void sb_st1(uint16_t* p, float v) { p[0] = (uint16_t)(sb_pack(v, 0.0f) & 0xffffu); }

template <int NV>
__device__ __forceinline__ void sb_block_sum(float (&v)[NV], float* sh /* [NV*4] */) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = cfn_wave_sum(v[i]);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int i = 0; i < NV; ++i) sh[i * 4 + wave] = v[i];
    __syncthreads();
    if (threadIdx.x == 0)
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = sh[i * 4] + sh[i * 4 + 1] + sh[i * 4 + 2] + sh[i * 4 + 3];
}

#define SB_ITEMS 4      // 16-byte accesses per thread: 32 elements -> one 32-bit ReLU mask word per thread

// ---- out = relu( A*y + B + (Ar*res + Br) ), x3d_fine.py:167-173; mask bit 8k+e = element e of the k-th access is > 0 ----
template <bool VEC>
__global__ __launch_bounds__(256) void bn_add_relu_fwd_bf16_kernel(const uint16_t* __restrict__ y, const double* __restrict__ A,
                                                                   const double* __restrict__ B, const uint16_t* __restrict__ res,
                                                                   const double* __restrict__ Ar, const double* __restrict__ Br,
                                                                   uint16_t* __restrict__ out, unsigned* __restrict__ mask, long vol) {
    const long nc = blockIdx.y + (long)blockIdx.z * gridDim.y;
    const float a = A[nc], b = B[nc] + (Br ? Br[nc] : 0.0f), ar = Ar ? Ar[nc] : 1.0f;
    const long base = nc * vol;
    unsigned mw = 0;
    if (VEC) {
        long i = ((long)blockIdx.x * 256 * SB_ITEMS + threadIdx.x) * 8;
#pragma unroll
        for (int k = 0; k < SB_ITEMS; ++k, i += 256 * 8) {
            if (i >= vol) break;
            const u4v yv = *reinterpret_cast<const u4v*>(y + base + i);
            const u4v rv = *reinterpret_cast<const u4v*>(res + base + i);
            u4v o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lo = fmaxf(fmaf(sb_lo(yv[e]), a, fmaf(sb_lo(rv[e]), ar, b)), 0.f);
                const float hi = fmaxf(fmaf(sb_hi(yv[e]), a, fmaf(sb_hi(rv[e]), ar, b)), 0.f);
                o[e] = sb_pack(lo, hi);
                // the mask records the sign of the ROUNDED output (what the next layer sees); rounding keeps the sign
                mw |= ((lo > 0.f ? 1u : 0u) | (hi > 0.f ? 2u : 0u)) << (8 * k + 2 * e);
            }
            *reinterpret_cast<u4v*>(out + base + i) = o;
        }
        if (mask) mask[(nc * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = mw;
    } else {
        for (long i = (long)blockIdx.x * 256 * SB_ITEMS * 8 + threadIdx.x; i < vol && i < ((long)blockIdx.x + 1) * 256 * SB_ITEMS * 8; i += 256)
            sb_st1(out + base + i, fmaxf(fmaf(sb_ld1(y + base + i), a, fmaf(sb_ld1(res + base + i), ar, b)), 0.f));
    }
}

// g = (gout [+ gout2]) * (out > 0) written once (see cfn_bn_add_relu_bwd_g);  gA += sum g*y;  gB += sum g;  gAr += sum g*res
template <bool VEC>
__global__ __launch_bounds__(256) void bn_add_relu_bwd_g_bf16_kernel(const uint16_t* __restrict__ gout, const uint16_t* __restrict__ gout2,
                                                                     const uint16_t* __restrict__ out, const unsigned* __restrict__ mask,
                                                                     const uint16_t* __restrict__ y, const uint16_t* __restrict__ res,
                                                                     uint16_t* __restrict__ g_out, double* __restrict__ gA,
                                                                     double* __restrict__ gB, double* __restrict__ gAr, long vol) {
    __shared__ float sh[12];
    const long nc = blockIdx.y + (long)blockIdx.z * gridDim.y;
    const long base = nc * vol;
    float acc[3] = {0.f, 0.f, 0.f};
    if (VEC) {
        long i = ((long)blockIdx.x * 256 * SB_ITEMS + threadIdx.x) * 8;
        const unsigned mw = mask ? mask[(nc * gridDim.x + blockIdx.x) * 256 + threadIdx.x] : 0u;
#pragma unroll
        for (int k = 0; k < SB_ITEMS; ++k, i += 256 * 8) {
            if (i >= vol) break;
            const u4v go = *reinterpret_cast<const u4v*>(gout + base + i);
            u4v go2 = {0u, 0u, 0u, 0u}, ov = {0u, 0u, 0u, 0u}, rv = {0u, 0u, 0u, 0u};
            if (gout2) go2 = *reinterpret_cast<const u4v*>(gout2 + base + i);
            if (!mask) ov = *reinterpret_cast<const u4v*>(out + base + i);
            if (gAr) rv = *reinterpret_cast<const u4v*>(res + base + i);
            const u4v yv = *reinterpret_cast<const u4v*>(y + base + i);
            u4v g;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bool plo, phi;
                if (mask) { plo = (mw >> (8 * k + 2 * e)) & 1u; phi = (mw >> (8 * k + 2 * e + 1)) & 1u; }
                else { plo = sb_lo(ov[e]) > 0.f; phi = sb_hi(ov[e]) > 0.f; }
                float lo = sb_lo(go[e]), hi = sb_hi(go[e]);
                if (gout2) { lo += sb_lo(go2[e]); hi += sb_hi(go2[e]); }
                g[e] = sb_pack(plo ? lo : 0.f, phi ? hi : 0.f);
                lo = sb_lo(g[e]); hi = sb_hi(g[e]);                    // reductions over the values the consumers will read
                acc[0] += lo * sb_lo(yv[e]) + hi * sb_hi(yv[e]);
                acc[1] += lo + hi;
                if (gAr) acc[2] += lo * sb_lo(rv[e]) + hi * sb_hi(rv[e]);
            }
            *reinterpret_cast<u4v*>(g_out + base + i) = g;
        }
    } else {
        for (long i = (long)blockIdx.x * 256 * SB_ITEMS * 8 + threadIdx.x; i < vol && i < ((long)blockIdx.x + 1) * 256 * SB_ITEMS * 8; i += 256) {
            float g = sb_ld1(out + base + i) > 0.f ? sb_ld1(gout + base + i) + (gout2 ? sb_ld1(gout2 + base + i) : 0.f) : 0.f;
            sb_st1(g_out + base + i, g);
            g = sb_ld1(g_out + base + i);
            acc[0] = fmaf(g, sb_ld1(y + base + i), acc[0]);
            acc[1] += g;
            if (gAr) acc[2] = fmaf(g, sb_ld1(res + base + i), acc[2]);
        }
    }
    sb_block_sum<3>(acc, sh);
    if (threadIdx.x == 0) {
        cfn_add64(&gA[nc], (double)acc[0]);
        cfn_add64(&gB[nc], (double)acc[1]);
        if (gAr) cfn_add64(&gAr[nc], (double)acc[2]);
    }
}

static bool sb_vec_ok(long vol, const void* p0, const void* p1, const void* p2) {
    return vol % 8 == 0 && ((reinterpret_cast<uintptr_t>(p0) | reinterpret_cast<uintptr_t>(p1) | reinterpret_cast<uintptr_t>(p2)) & 15) == 0;
}

extern "C" long H16N(cfn_bn_add_relu_mask_words)(long NC, long vol) {
    if (vol % 8 != 0) return 0;
    return NC * cfn_cdiv(vol, 256L * SB_ITEMS * 8) * 256;
}

extern "C" int H16N(cfn_bn_add_relu_fwd)(const uint16_t* y, const double* A, const double* B, const uint16_t* res, const double* Ar,
                                        const double* Br, uint16_t* out, int* mask, long NC, long vol, void* stream) {
    CFN_REQUIRE(y && A && B && res && out, "cfn_bn_add_relu_fwd_" H16_NAME ": null tensor");
    CFN_REQUIRE((Ar == nullptr) == (Br == nullptr), "cfn_bn_add_relu_fwd_" H16_NAME ": Ar/Br mismatch");
    unsigned gy_, gz_;
    CFN_REQUIRE(cfn_split_nc(NC, gy_, gz_), "cfn_bn_add_relu_fwd_" H16_NAME ": N*C = %ld exceeds grid.y", NC);
    const bool vec = sb_vec_ok(vol, y, res, out);
    CFN_REQUIRE(mask == nullptr || vec, "cfn_bn_add_relu_fwd_" H16_NAME ": the bit mask needs the vector path (volume %% 8 == 0, 16-byte aligned)");
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_ELEMWISE, st, 6.0 * NC * vol);
    const dim3 grid((unsigned)cfn_cdiv(vol, 256L * SB_ITEMS * 8), gy_, gz_);
    if (vec) hipLaunchKernelGGL(bn_add_relu_fwd_bf16_kernel<true>, grid, dim3(256), 0, st, y, A, B, res, Ar, Br, out, (unsigned*)mask, vol);
    else hipLaunchKernelGGL(bn_add_relu_fwd_bf16_kernel<false>, grid, dim3(256), 0, st, y, A, B, res, Ar, Br, out, (unsigned*)nullptr, vol);
    return cfn_check_launch("bn_add_relu_fwd " H16_NAME);
}

extern "C" int H16N(cfn_bn_add_relu_bwd_g)(const uint16_t* gout, const uint16_t* gout2, const uint16_t* out, const int* mask,
                                          const uint16_t* y, const uint16_t* res, uint16_t* g, double* gA, double* gB, double* gAr,
                                          long NC, long vol, void* stream) {
    CFN_REQUIRE(gout && y && g && gA && gB, "cfn_bn_add_relu_bwd_g_" H16_NAME ": null tensor");
    CFN_REQUIRE((out != nullptr) != (mask != nullptr), "cfn_bn_add_relu_bwd_g_" H16_NAME ": give exactly one of out / mask");
    CFN_REQUIRE(gAr == nullptr || res != nullptr, "cfn_bn_add_relu_bwd_g_" H16_NAME ": gAr needs res");
    unsigned gy_, gz_;
    CFN_REQUIRE(cfn_split_nc(NC, gy_, gz_), "cfn_bn_add_relu_bwd_g_" H16_NAME ": N*C = %ld exceeds grid.y", NC);
    const bool vec = sb_vec_ok(vol, gout, y, g) && sb_vec_ok(vol, gout2, out, res);
    CFN_REQUIRE(mask == nullptr || vec, "cfn_bn_add_relu_bwd_g_" H16_NAME ": the bit mask needs the vector path");
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_ELEMWISE, st, (6.0 + (gout2 ? 2.0 : 0.0) + (out ? 2.0 : 0.125) + (gAr ? 2.0 : 0.0)) * NC * vol);
    const dim3 grid((unsigned)cfn_cdiv(vol, 256L * SB_ITEMS * 8), gy_, gz_);
    if (vec) hipLaunchKernelGGL(bn_add_relu_bwd_g_bf16_kernel<true>, grid, dim3(256), 0, st, gout, gout2, out, (const unsigned*)mask, y, res, g, gA, gB, gAr, vol);
    else hipLaunchKernelGGL(bn_add_relu_bwd_g_bf16_kernel<false>, grid, dim3(256), 0, st, gout, gout2, out, (const unsigned*)nullptr, y, res, g, gA, gB, gAr, vol);
    return cfn_check_launch("bn_add_relu_bwd_g " H16_NAME);
}

// ---- adaptive spatial mean of act(A x + B), x bf16 -> pooled fp32 (the head / feature tower leave the bf16 domain here:
// everything after the pooling is T*OH*OW small), x3d_fine.py:255,366 / :345-363 ----
__device__ __forceinline__ int sb_ap_start(int o, int O, int S) { return (o * S) / O; }
__device__ __forceinline__ int sb_ap_end(int o, int O, int S) { return ((o + 1) * S + O - 1) / O; }

__global__ __launch_bounds__(256) void pool_hw_fwd_bf16_kernel(const uint16_t* __restrict__ x, const double* __restrict__ A,
                                                               const double* __restrict__ B, int act, float* __restrict__ out, int T,
                                                               int H, int W, int OH, int OW) {
    const long nc = blockIdx.y + (long)blockIdx.z * gridDim.y;
    const long ovol = (long)T * OH * OW;
    const long o = (long)blockIdx.x * 256 + threadIdx.x;
    if (o >= ovol) return;
    const float a = A ? A[nc] : 1.0f, b = A ? B[nc] : 0.0f;
    const int ow = (int)(o % OW), oh = (int)((o / OW) % OH), t = (int)(o / ((long)OW * OH));
    const int h0 = sb_ap_start(oh, OH, H), h1 = sb_ap_end(oh, OH, H), w0 = sb_ap_start(ow, OW, W), w1 = sb_ap_end(ow, OW, W);
    const uint16_t* p = x + (nc * T + t) * (long)H * W;
    float s = 0.f;
    for (int i = h0; i < h1; ++i)
        for (int j = w0; j < w1; ++j) s += cfn_act_rt(fmaf(sb_ld1(p + i * W + j), a, b), act);
    out[nc * ovol + o] = s / (float)((h1 - h0) * (w1 - w0));
}

__global__ __launch_bounds__(256) void pool_hw_bwd_bf16_kernel(const float* __restrict__ gout, const uint16_t* __restrict__ x,
                                                               const double* __restrict__ A, const double* __restrict__ B, int act,
                                                               uint16_t* __restrict__ gx, double* __restrict__ gA, double* __restrict__ gB,
                                                               int T, int H, int W, int OH, int OW) {
    __shared__ float sh[8];
    const long nc = blockIdx.y + (long)blockIdx.z * gridDim.y;
    const long vol = (long)T * H * W;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    float acc[2] = {0.f, 0.f};
    if (i < vol) {
        const float a = A ? A[nc] : 1.0f, b = A ? B[nc] : 0.0f;
        const int w = (int)(i % W), h = (int)((i / W) % H), t = (int)(i / ((long)W * H));
        const float* gp = gout + (nc * T + t) * (long)OH * OW;
        float g = 0.f;
        for (int oh = 0; oh < OH; ++oh) {
            const int hs = sb_ap_start(oh, OH, H), he = sb_ap_end(oh, OH, H);
            if (h < hs || h >= he) continue;
            for (int ow = 0; ow < OW; ++ow) {
                const int ws = sb_ap_start(ow, OW, W), we = sb_ap_end(ow, OW, W);
                if (w >= ws && w < we) g += gp[oh * OW + ow] / (float)((he - hs) * (we - ws));
            }
        }
        const float xv = sb_ld1(x + nc * vol + i);
        const float dz = g * cfn_act_grad_rt(fmaf(xv, a, b), act);
        acc[0] = dz * xv;
        acc[1] = dz;
        sb_st1(gx + nc * vol + i, dz * a);
    }
    if (gA) {
        sb_block_sum<2>(acc, sh);
        if (threadIdx.x == 0) { cfn_add64(&gA[nc], (double)acc[0]); cfn_add64(&gB[nc], (double)acc[1]); }
    }
}

extern "C" int H16N(cfn_pool_hw_fwd)(const uint16_t* x, const double* A, const double* B, int act, float* out, long NC, int T, int H,
                                    int W, int OH, int OW, void* stream) {
    CFN_REQUIRE(x && out, "cfn_pool_hw_fwd_" H16_NAME ": null tensor");
    CFN_REQUIRE((A == nullptr) == (B == nullptr), "cfn_pool_hw_fwd_" H16_NAME ": A/B mismatch");
    unsigned gy_, gz_;
    CFN_REQUIRE(cfn_split_nc(NC, gy_, gz_), "cfn_pool_hw_fwd_" H16_NAME ": N*C = %ld exceeds grid.y", NC);
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_ELEMWISE, st, 2.0 * NC * T * H * W);
    hipLaunchKernelGGL(pool_hw_fwd_bf16_kernel, dim3((unsigned)cfn_cdiv((long)T * OH * OW, 256), gy_, gz_), dim3(256), 0, st, x, A, B, act, out,
                       T, H, W, OH, OW);
    return cfn_check_launch("pool_hw_fwd " H16_NAME);
}

extern "C" int H16N(cfn_pool_hw_bwd)(const float* gout, const uint16_t* x, const double* A, const double* B, int act, uint16_t* gx,
                                    double* gA, double* gB, long NC, int T, int H, int W, int OH, int OW, void* stream) {
    CFN_REQUIRE(gout && x && gx, "cfn_pool_hw_bwd_" H16_NAME ": null tensor");
    CFN_REQUIRE((A == nullptr) == (B == nullptr), "cfn_pool_hw_bwd_" H16_NAME ": A/B mismatch");
    CFN_REQUIRE(A == nullptr || (gA && gB), "cfn_pool_hw_bwd_" H16_NAME ": prologue needs gA, gB");
    unsigned gy_, gz_;
    CFN_REQUIRE(cfn_split_nc(NC, gy_, gz_), "cfn_pool_hw_bwd_" H16_NAME ": N*C = %ld exceeds grid.y", NC);
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_ELEMWISE, st, 4.0 * NC * T * H * W);
    hipLaunchKernelGGL(pool_hw_bwd_bf16_kernel, dim3((unsigned)cfn_cdiv((long)T * H * W, 256), gy_, gz_), dim3(256), 0, st, gout, x, A, B, act, gx,
                       A ? gA : nullptr, A ? gB : nullptr, T, H, W, OH, OW);
    return cfn_check_launch("pool_hw_bwd " H16_NAME);
}
