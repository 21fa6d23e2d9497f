// Coarse-stream specific kernels:
//   * data gradient of the dense (Grid Pool saliency) convolutions, gather form by stride-parity class;
//   * the Multi-stage Fusion temporal-alignment gather of RewightLayer (x3d_coarse.py:199-226), evaluated at the
//     fine features' native 7x7 resolution: the reference first up-samples them with adaptive_max_pool2d
//     (7 -> 56/28/14: each output cell copies exactly one input cell) and materialises a
//     (B,C,T',K,h,w) product; every quantity is constant over the (h/7 x w/7) blocks, so the 7x7
//     result up-sampled is identical (measured 3e-8).
#include "cfn_common.h"
#include <stdlib.h>

// salconv.hip
int sal_dgrad_try_launch(const float* gy, const float* y, const double* gs, const double* gq, const float* w, const float* x,
                         const double* A, const double* B, int act, float* gx, double* gA, double* gB, int N, int Cin, int Cout,
                         int T, int Hi, int Wi, const int* g, hipStream_t st);

// ---------------------------------------------------------------------------------------------------------
// gx[n,ci,it,ih,iw] = act'(A x + B) * A * sum_{co, taps hitting (it,ih,iw)} W[co,ci,kt,kh,kw] * g'[n,co,to,oh,ow]
// with g' = gy + gs[n,co] + 2 y gq[n,co];  gA += sum dz*x, gB += sum dz.
// ---------------------------------------------------------------------------------------------------------
struct DenseBwdArgs {
    const float* gy; const float* y; const double* gs; const double* gq; const float* w;
    const float* x; const double* A; const double* B; float* gx; double* gA; double* gB;
    int Cin, Cout, Ti, Hi, Wi, To, Ho, Wo, kT, kH, kW, sT, sH, sW, pT, pH, pW, act;
};

// Threads are grouped by stride-parity class (blockIdx.z = (it%sT, ih%sH, iw%sW)): inside a class every lane sees the
// same set of contributing taps (kt = k0, k0+sT, ...), so there is no divergence and no modulo test in the loops, and
// consecutive lanes read consecutive output columns of g'.  Weights of this input channel sit in LDS.
__global__ __launch_bounds__(256) void conv3d_dense_bwd_data_kernel(const DenseBwdArgs a) {
    __shared__ float sh[8];
    extern __shared__ float sg[];            // gs[Cout] | 2gq[Cout] | w[Cout][KV] of this input channel
    const int nci = blockIdx.y, n = nci / a.Cin, ci = nci - n * a.Cin;
    const int KV = a.kT * a.kH * a.kW;
    float* sw = sg + 2 * a.Cout;
    for (int co = threadIdx.x; co < a.Cout; co += 256) {
        sg[co] = a.gs ? (float)a.gs[(long)n * a.Cout + co] : 0.0f;
        sg[a.Cout + co] = (a.gq && a.y) ? 2.0f * (float)a.gq[(long)n * a.Cout + co] : 0.0f;
    }
    for (int e = threadIdx.x; e < a.Cout * KV; e += 256) {
        const int co = e / KV, tap = e - co * KV;
        sw[e] = a.w[((long)co * a.Cin + ci) * KV + tap];
    }
    __syncthreads();
    // parity class and the lattice of input positions it owns
    int cls = blockIdx.z;
    const int cw = cls % a.sW; cls /= a.sW;
    const int ch = cls % a.sH;
    const int ct = cls / a.sH;
    const int nw = (a.Wi - cw + a.sW - 1) / a.sW, nh = (a.Hi - ch + a.sH - 1) / a.sH, nt = (a.Ti - ct + a.sT - 1) / a.sT;
    const long pin = (long)a.Ti * a.Hi * a.Wi, po = (long)a.To * a.Ho * a.Wo;
    const long pc = (long)blockIdx.x * 256 + threadIdx.x;
    const bool ok = nw > 0 && nh > 0 && nt > 0 && pc < (long)nt * nh * nw;
    float s1 = 0.f, s2 = 0.f;
    if (ok) {
        const int jw = (int)(pc % nw), jh = (int)((pc / nw) % nh), jt = (int)(pc / ((long)nw * nh));
        const int iw = cw + jw * a.sW, ih = ch + jh * a.sH, it = ct + jt * a.sT;
        // first tap of each axis that lands on an output sample: (i + p - k) % s == 0
        const int kt0 = (ct + a.pT) % a.sT, kh0 = (ch + a.pH) % a.sH, kw0 = (cw + a.pW) % a.sW;
        float da = 0.f;
        for (int kt = kt0; kt < a.kT; kt += a.sT) {
            const int to = (it + a.pT - kt) / a.sT;
            if (it + a.pT - kt < 0 || to >= a.To) continue;
            for (int kh = kh0; kh < a.kH; kh += a.sH) {
                const int oh = (ih + a.pH - kh) / a.sH;
                if (ih + a.pH - kh < 0 || oh >= a.Ho) continue;
                for (int kw = kw0; kw < a.kW; kw += a.sW) {
                    const int ow = (iw + a.pW - kw) / a.sW;
                    if (iw + a.pW - kw < 0 || ow >= a.Wo) continue;
                    const long oq = ((long)to * a.Ho + oh) * a.Wo + ow;
                    const int tap = (kt * a.kH + kh) * a.kW + kw;
                    const float* gyp = a.gy + (long)n * a.Cout * po + oq;
                    const float* yp = a.y ? a.y + (long)n * a.Cout * po + oq : nullptr;
#pragma unroll 4
                    for (int co = 0; co < a.Cout; ++co) {
                        float g = gyp[(long)co * po] + sg[co];
                        if (yp) g = fmaf(yp[(long)co * po], sg[a.Cout + co], g);
                        da = fmaf(sw[co * KV + tap], g, da);
                    }
                }
            }
        }
        const long o = (long)nci * pin + ((long)it * a.Hi + ih) * a.Wi + iw;
        if (a.A) {
            const float xa = a.A[nci], xb = a.B[nci], xv = a.x[o];
            const float dz = da * cfn_act_grad_rt(fmaf(xv, xa, xb), a.act);
            s1 = dz * xv; s2 = dz;
            a.gx[o] = dz * xa;
        } else {
            a.gx[o] = da;
        }
    }
    if (a.A && a.gA) {
        s1 = cfn_wave_sum(s1); s2 = cfn_wave_sum(s2);
        if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = s1; sh[4 + (threadIdx.x >> 6)] = s2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            cfn_add64(&a.gA[nci], (double)(sh[0] + sh[1] + sh[2] + sh[3]));
            cfn_add64(&a.gB[nci], (double)(sh[4] + sh[5] + sh[6] + sh[7]));
        }
    }
}

// All-input-channels variant (Cin <= 32, weights of the whole layer <= 64 KiB): a thread owns ONE input position of
// its parity class and accumulates the gradient of every input channel in registers, so g' = gy + gs + 2 y gq is
// loaded once per (output channel, tap) instead of once per input channel as well (24x fewer loads for the 24-channel
// saliency convs).  Weights live in LDS as [co][tap][ci] (ci fastest: one ds_read_b128 feeds 4 channels).
template <int CI>
__global__ __launch_bounds__(256) void conv3d_dense_bwd_data_allci_kernel(const DenseBwdArgs a) {
    extern __shared__ float sg[];            // gs[Cout] | 2gq[Cout] | w[Cout][KV][CI] | red[4 waves][2*CI]
    const int n = blockIdx.y;
    const int KV = a.kT * a.kH * a.kW;
    float* sw = sg + 2 * a.Cout;
    float* red = sw + a.Cout * KV * CI;
    for (int co = threadIdx.x; co < a.Cout; co += 256) {
        sg[co] = a.gs ? (float)a.gs[(long)n * a.Cout + co] : 0.0f;
        sg[a.Cout + co] = (a.gq && a.y) ? 2.0f * (float)a.gq[(long)n * a.Cout + co] : 0.0f;
    }
    for (int e = threadIdx.x; e < a.Cout * KV * CI; e += 256) {
        const int ci = e % CI, r = e / CI, tap = r % KV, co = r / KV;
        sw[e] = ci < a.Cin ? a.w[((long)co * a.Cin + ci) * KV + tap] : 0.0f;
    }
    for (int e = threadIdx.x; e < 8 * CI; e += 256) red[e] = 0.0f;   // per-wave slots: fixed summation order
    __syncthreads();
    int cls = blockIdx.z;
    const int cw = cls % a.sW; cls /= a.sW;
    const int ch = cls % a.sH;
    const int ct = cls / a.sH;
    const int nw = (a.Wi - cw + a.sW - 1) / a.sW, nh = (a.Hi - ch + a.sH - 1) / a.sH, nt = (a.Ti - ct + a.sT - 1) / a.sT;
    const long pin = (long)a.Ti * a.Hi * a.Wi, po = (long)a.To * a.Ho * a.Wo;
    const long pc = (long)blockIdx.x * 256 + threadIdx.x;
    const bool ok = nw > 0 && nh > 0 && nt > 0 && pc < (long)nt * nh * nw;
    float da[CI];
#pragma unroll
    for (int i = 0; i < CI; ++i) da[i] = 0.0f;
    long opos = 0;
    if (ok) {
        const int jw = (int)(pc % nw), jh = (int)((pc / nw) % nh), jt = (int)(pc / ((long)nw * nh));
        const int iw = cw + jw * a.sW, ih = ch + jh * a.sH, it = ct + jt * a.sT;
        opos = ((long)it * a.Hi + ih) * a.Wi + iw;
        const int kt0 = (ct + a.pT) % a.sT, kh0 = (ch + a.pH) % a.sH, kw0 = (cw + a.pW) % a.sW;
        for (int kt = kt0; kt < a.kT; kt += a.sT) {
            const int to = (it + a.pT - kt) / a.sT;
            if (it + a.pT - kt < 0 || to >= a.To) continue;
            for (int kh = kh0; kh < a.kH; kh += a.sH) {
                const int oh = (ih + a.pH - kh) / a.sH;
                if (ih + a.pH - kh < 0 || oh >= a.Ho) continue;
                for (int kw = kw0; kw < a.kW; kw += a.sW) {
                    const int ow = (iw + a.pW - kw) / a.sW;
                    if (iw + a.pW - kw < 0 || ow >= a.Wo) continue;
                    const long oq = ((long)to * a.Ho + oh) * a.Wo + ow;
                    const int tap = (kt * a.kH + kh) * a.kW + kw;
                    const float* gyp = a.gy + (long)n * a.Cout * po + oq;
                    const float* yp = a.y ? a.y + (long)n * a.Cout * po + oq : nullptr;
                    for (int co = 0; co < a.Cout; ++co) {
                        float g = gyp[(long)co * po] + sg[co];
                        if (yp) g = fmaf(yp[(long)co * po], sg[a.Cout + co], g);
                        const float* wp = sw + (co * KV + tap) * CI;
#pragma unroll
                        for (int i = 0; i < CI; ++i) da[i] = fmaf(wp[i], g, da[i]);
                    }
                }
            }
        }
    }
    // epilogue per input channel; the per-(n,ci) sums go through wave reductions and one LDS atomic per wave
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < CI; ++i) {
        float s1 = 0.f, s2 = 0.f;
        if (i < a.Cin) {
            const long nci = (long)n * a.Cin + i;
            if (ok) {
                const long o = nci * pin + opos;
                if (a.A) {
                    const float xa = a.A[nci], xb = a.B[nci], xv = a.x[o];
                    const float dz = da[i] * cfn_act_grad_rt(fmaf(xv, xa, xb), a.act);
                    s1 = dz * xv; s2 = dz;
                    a.gx[o] = dz * xa;
                } else {
                    a.gx[o] = da[i];
                }
            }
            if (a.A && a.gA) {
                s1 = cfn_wave_sum(s1); s2 = cfn_wave_sum(s2);
                if (lane == 0) { red[(threadIdx.x >> 6) * 2 * CI + 2 * i] = s1; red[(threadIdx.x >> 6) * 2 * CI + 2 * i + 1] = s2; }
            }
        }
    }
    if (a.A && a.gA) {
        __syncthreads();
        for (int i = threadIdx.x; i < a.Cin; i += 256) {
            cfn_add64(&a.gA[(long)n * a.Cin + i], (double)((red[2 * i] + red[2 * CI + 2 * i]) + (red[4 * CI + 2 * i] + red[6 * CI + 2 * i])));
            cfn_add64(&a.gB[(long)n * a.Cin + i], (double)((red[2 * i + 1] + red[2 * CI + 2 * i + 1]) + (red[4 * CI + 2 * i + 1] + red[6 * CI + 2 * i + 1])));
        }
    }
}

extern "C" int cfn_conv3d_dense_bwd_data(const float* gy, const float* y, const double* gsum, const double* gsumsq,
                                         const float* w, const float* x, const double* A, const double* B, int act, float* gx,
                                         double* gA, double* gB, int N, int Cin, int Cout, int T, int Hi, int Wi,
                                         const int* geom, void* stream) {
    CFN_REQUIRE(gy && w && gx && geom, "cfn_conv3d_dense_bwd_data: null tensor");
    CFN_REQUIRE((A == nullptr) == (B == nullptr), "cfn_conv3d_dense_bwd_data: A/B mismatch");
    CFN_REQUIRE(A == nullptr || (x && gA && gB), "cfn_conv3d_dense_bwd_data: prologue needs x, gA, gB");
    CFN_REQUIRE(gsumsq == nullptr || y != nullptr, "cfn_conv3d_dense_bwd_data: gsumsq needs y");
    CFN_REQUIRE((long)N * Cin <= 65535, "cfn_conv3d_dense_bwd_data: N*Cin exceeds grid.y");
    DenseBwdArgs a = {};
    a.gy = gy; a.y = gsumsq ? y : nullptr; a.gs = gsum; a.gq = gsumsq; a.w = w; a.x = x; a.A = A; a.B = B; a.gx = gx;
    a.gA = gA; a.gB = gB; a.Cin = Cin; a.Cout = Cout; a.Ti = T; a.Hi = Hi; a.Wi = Wi; a.act = act;
    a.kT = geom[0]; a.kH = geom[1]; a.kW = geom[2]; a.sT = geom[3]; a.sH = geom[4]; a.sW = geom[5];
    a.pT = geom[6]; a.pH = geom[7]; a.pW = geom[8];
    a.To = (T + 2 * a.pT - a.kT) / a.sT + 1;
    a.Ho = (Hi + 2 * a.pH - a.kH) / a.sH + 1;
    a.Wo = (Wi + 2 * a.pW - a.kW) / a.sW + 1;
    const long pin = (long)T * Hi * Wi;
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_STEM, st, 4.0 * N * ((double)Cin * pin * 2 + (double)Cout * a.To * a.Ho * a.Wo));
    {   // LDS-tiled MFMA kernel (salconv.hip) for the Grid Pool saliency shapes
        const int rs = sal_dgrad_try_launch(gy, a.y, gsum, gsumsq, w, x, A, B, act, gx, gA, gB, N, Cin, Cout, T, Hi, Wi, geom, st);
        if (rs >= 0) return rs;
    }
    const int ncls = a.sT * a.sH * a.sW;
    const long pcls = (long)cfn_cdiv(T, a.sT) * cfn_cdiv(Hi, a.sH) * cfn_cdiv(Wi, a.sW);     // largest class
    {   // all-input-channels variant when the whole layer's weights fit in LDS
        const int CI = Cin <= 8 ? 8 : (Cin <= 24 ? 24 : 32);
        const size_t lds_all = ((size_t)2 * Cout + (size_t)Cout * a.kT * a.kH * a.kW * CI + 8 * CI) * sizeof(float);
        if (Cin <= 32 && lds_all <= 64 * 1024 && ncls <= 64 && N <= 65535) {
            const dim3 grid(cfn_cdiv(pcls, 256), N, ncls);
#define CFN_DENSE_GO(CIV)                                                                                                 \
            do {                                                                                                           \
                auto k = conv3d_dense_bwd_data_allci_kernel<CIV>;                                                          \
                if (lds_all > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_all); \
                hipLaunchKernelGGL(k, grid, dim3(256), lds_all, st, a);                                                    \
            } while (0)
            if (CI == 8) CFN_DENSE_GO(8); else if (CI == 24) CFN_DENSE_GO(24); else CFN_DENSE_GO(32);
#undef CFN_DENSE_GO
            return cfn_check_launch("conv3d_dense_bwd_data(all channels)");
        }
    }
    const size_t lds = ((size_t)2 * Cout + (size_t)Cout * a.kT * a.kH * a.kW) * sizeof(float);
    CFN_REQUIRE(ncls <= 64 && lds <= 60 * 1024, "cfn_conv3d_dense_bwd_data: stride / weight slice too large");
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)conv3d_dense_bwd_data_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(conv3d_dense_bwd_data_kernel, dim3(cfn_cdiv(pcls, 256), N * Cin, ncls), dim3(256), lds, st, a);
    return cfn_check_launch("conv3d_dense_bwd_data");
}

// ---------------------------------------------------------------------------------------------------------
// Gaussian temporal alignment (Gaussian.forward, x3d_coarse.py:256-286).  One thread per (row r of b2 = B*crops, knot k):
//   mu = (tl + st) / ratio,  tl = gx[r,k]*tx (grid mode) or k,  st = meta[b,0] + meta[b,3]*crop  (:264-266)
//   f[t] = exp(-((t - mu)^2 / (2 std^2 + 1e-16))),  std = sum(mask[b,:]) / 8,   GX[r,t,k] = f[t] / (max_t f + 1e-16)
// same operation order as the reference's tensor expression; backward = autograd of it (the max passes its
// gradient to the first arg-max frame, like torch.max(dim)).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gauss_f(int t, float mu, float den) {
    const float d = (float)t - mu;
    return expf(-((d * d) / den));
}

__global__ __launch_bounds__(256) void gauss_align_fwd_kernel(const long* __restrict__ meta, const float* __restrict__ mask,
                                                              const float* __restrict__ gx, float tx, float ratio,
                                                              float* __restrict__ GX, int crops, int Tf, int K, int total) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int k = e % K, r = e / K, b = r / crops, crop = r - b * crops;
    float msum = 0.f;
    for (int t = 0; t < Tf; ++t) msum += mask[(long)b * Tf + t];
    const float std = 0.125f * msum;
    const float den = 2.0f * (std * std) + 1e-16f;
    const float st = (float)meta[b * 4] + (float)meta[b * 4 + 3] * (float)crop;
    const float tl = gx ? gx[(long)r * K + k] * tx : (float)k;
    const float mu = (tl + st) / ratio;
    float m = 0.f;
    for (int t = 0; t < Tf; ++t) m = fmaxf(m, gauss_f(t, mu, den));
    const float dn = m + 1e-16f;
    for (int t = 0; t < Tf; ++t) GX[((long)r * Tf + t) * K + k] = gauss_f(t, mu, den) / dn;
}

__global__ __launch_bounds__(256) void gauss_align_bwd_kernel(const float* __restrict__ gGX, const long* __restrict__ meta,
                                                              const float* __restrict__ mask, const float* __restrict__ gx,
                                                              float tx, float ratio, float* __restrict__ ggx, int crops, int Tf,
                                                              int K, int total) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int k = e % K, r = e / K, b = r / crops, crop = r - b * crops;
    float msum = 0.f;
    for (int t = 0; t < Tf; ++t) msum += mask[(long)b * Tf + t];
    const float std = 0.125f * msum;
    const float den = 2.0f * (std * std) + 1e-16f;
    const float st = (float)meta[b * 4] + (float)meta[b * 4 + 3] * (float)crop;
    const float mu = (gx[(long)r * K + k] * tx + st) / ratio;
    float m = -1.f;
    int tm = 0;
    for (int t = 0; t < Tf; ++t) { const float f = gauss_f(t, mu, den); if (f > m) { m = f; tm = t; } }
    const float dn = m + 1e-16f;
    // y_t = f_t / dn:  d/df_t = g_t / dn ;  d/dm = -sum_t g_t f_t / dn^2 (lands on frame tm) ;  df_t/dmu = f_t * 2 (t - mu) / den
    float gm = 0.f, gmu = 0.f;
    for (int t = 0; t < Tf; ++t) {
        const float g = gGX[((long)r * Tf + t) * K + k], f = gauss_f(t, mu, den);
        gm -= g * f;
        gmu += (g / dn) * f * (2.0f * ((float)t - mu) / den);
    }
    gmu += (gm / (dn * dn)) * m * (2.0f * ((float)tm - mu) / den);
    ggx[(long)r * K + k] = gmu / ratio * tx;
}

extern "C" int cfn_gauss_align_fwd(const long* meta, const float* mask, const float* gx, double tx, double ratio, float* GX,
                                   int B, int crops, int Tf, int K, void* stream) {
    CFN_REQUIRE(meta && mask && GX, "cfn_gauss_align_fwd: null tensor");
    CFN_REQUIRE(B > 0 && crops > 0 && Tf > 0 && K > 0 && ratio != 0.0, "cfn_gauss_align_fwd: bad sizes");
    const int total = B * crops * K;
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_FUSION, st, 4.0 * B * crops * (double)Tf * K);
    hipLaunchKernelGGL(gauss_align_fwd_kernel, dim3(cfn_cdiv(total, 256)), dim3(256), 0, st, meta, mask, gx, (float)tx, (float)ratio,
                       GX, crops, Tf, K, total);
    return cfn_check_launch("gauss_align_fwd");
}

extern "C" int cfn_gauss_align_bwd(const float* gGX, const long* meta, const float* mask, const float* gx, double tx, double ratio,
                                   float* ggx, int B, int crops, int Tf, int K, void* stream) {
    CFN_REQUIRE(gGX && meta && mask && gx && ggx, "cfn_gauss_align_bwd: null tensor");
    const int total = B * crops * K;
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_FUSION, st, 4.0 * B * crops * (double)Tf * K);
    hipLaunchKernelGGL(gauss_align_bwd_kernel, dim3(cfn_cdiv(total, 256)), dim3(256), 0, st, gGX, meta, mask, gx, (float)tx,
                       (float)ratio, ggx, crops, Tf, K, total);
    return cfn_check_launch("gauss_align_bwd");
}

// ---------------------------------------------------------------------------------------------------------
// fusion gather.  x (B,C,Tf,P) fine features (shared by the `crops` rows r = b*crops + j of a video), at_raw (B,Tf,P)
// attention logits (at = sigmoid(at_raw + at_bias[0]), x3d_coarse.py:219), GX (B*crops,Tf,K) Gaussian alignment, mask (B,Tf):
//   w[r,t,k,p] = at[b,t,p] * GX[r,t,k] * mask[b,t];  den[r,k,p] = sum_t w + 1e-6;  z[r,c,k,p] = sum_t x[b,c,t,p]*w / den
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fusion_gather_fwd_kernel(const float* __restrict__ x, const float* __restrict__ at_raw,
                                                                const float* __restrict__ at_bias, const float* __restrict__ GX,
                                                                const float* __restrict__ mask, float* __restrict__ z,
                                                                float* __restrict__ den, int crops, int C, int Tf, int K, int P,
                                                                long total) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int p = (int)(e % P), k = (int)((e / P) % K), c = (int)((e / ((long)P * K)) % C);
    const long r = e / ((long)P * K * C), b = r / crops;
    const float* xp = x + ((b * C + c) * Tf) * (long)P + p;
    const float* ap = at_raw + b * Tf * (long)P + p;
    const float* gp = GX + r * Tf * (long)K + k;
    const float* mp = mask + b * Tf;
    const float ab = at_bias ? at_bias[0] : 0.0f;
    float num = 0.f, d = 0.f;
    for (int t = 0; t < Tf; ++t) {
        const float a = 1.0f / (1.0f + expf(-(ap[(long)t * P] + ab)));
        const float w = a * (gp[(long)t * K] * mp[t]);
        num = fmaf(xp[(long)t * P], w, num);
        d += w;
    }
    d += 1e-6f;
    z[e] = num / d;
    if (c == 0) den[(r * K + k) * (long)P + p] = d;
}

// REGISTER-TILED forward (round 4).  The kernel above gives a thread ONE output and walks Tf with three loads and an expf per step: the
// sigmoid of at is recomputed C x K times, the step of x3d_coarse at T = 256 spent 2.5 ms here (0.1 TB/s of 240 MB).  Per (row r, position p)
// the op is a small matrix product  Z_p (C x K) = X_p (C x Tf) . W_p (Tf x K),  W_p[t][k] = at[b,t,p] GX[r,t,k] mask[b,t]:  here a thread owns
// position p (the lane dimension: every load is a contiguous run over p) and a CT x KT block of (c, k): per t one sigmoid, KT broadcast
// loads of GX, CT loads of x, CT x KT FMAs.  The sum over t runs in the same order with the same roundings as above (bit-identical z / den).
template <int CT, int KT>
__global__ __launch_bounds__(256) void fusion_gather_fwd_tiled_kernel(const float* __restrict__ x, const float* __restrict__ at_raw,
                                                                      const float* __restrict__ at_bias, const float* __restrict__ GX,
                                                                      const float* __restrict__ mask, float* __restrict__ z,
                                                                      float* __restrict__ den, int crops, int C, int Tf, int K, int P,
                                                                      int kgroups, int cblocks) {
    const int G = 256 / P;                                             // channel groups per workgroup
    const int cg = threadIdx.x / P, p = threadIdx.x - cg * P;
    if (cg >= G) return;
    unsigned L = blockIdx.x;
    const int cb = L % cblocks; L /= cblocks;
    const int kg = L % kgroups;
    const long r = L / kgroups, b = r / crops;
    const int c0 = (cb * G + cg) * CT, k0 = kg * KT;
    if (c0 >= C) return;
    const float ab = at_bias ? at_bias[0] : 0.0f;
    const float* ap = at_raw + b * Tf * (long)P + p;
    const float* gp = GX + r * Tf * (long)K + k0;
    const float* mp = mask + b * Tf;
    const float* xp[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i) xp[i] = x + ((b * C + min(c0 + i, C - 1)) * Tf) * (long)P + p;
    float num[CT][KT], d[KT];
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        d[j] = 0.0f;
#pragma unroll
        for (int i = 0; i < CT; ++i) num[i][j] = 0.0f;
    }
    for (int t = 0; t < Tf; ++t) {
        const float a = 1.0f / (1.0f + expf(-(ap[(long)t * P] + ab)));
        const float m = mp[t];
        float w[KT], xv[CT];
#pragma unroll
        for (int j = 0; j < KT; ++j) w[j] = a * ((k0 + j < K ? gp[(long)t * K + j] : 0.0f) * m);
#pragma unroll
        for (int i = 0; i < CT; ++i) xv[i] = xp[i][(long)t * P];
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            d[j] += w[j];
#pragma unroll
            for (int i = 0; i < CT; ++i) num[i][j] = fmaf(xv[i], w[j], num[i][j]);
        }
    }
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        if (k0 + j < K) {
            const float dd = d[j] + 1e-6f;
#pragma unroll
            for (int i = 0; i < CT; ++i)
                if (c0 + i < C) z[((r * C + c0 + i) * K + k0 + j) * (long)P + p] = num[i][j] / dd;
            if (c0 == 0) den[(r * K + k0 + j) * (long)P + p] = dd;
        }
    }
}

// REGISTER-TILED dw (see the one-output kernel below for the formula): a thread owns position p and a TT x KT block of (t, k) and walks the
// channels: per c KT loads of gz and z, TT loads of x, TT x KT (subtract, FMA) pairs -- the same order and roundings per output.
template <int TT, int KT>
__global__ __launch_bounds__(256) void fusion_gather_bwd_w_tiled_kernel(const float* __restrict__ gz, const float* __restrict__ z,
                                                                        const float* __restrict__ den, const float* __restrict__ x,
                                                                        float* __restrict__ dw, int crops, int C, int Tf, int K, int P,
                                                                        int kgroups, int tgroups, int rows) {
    const int G = 256 / P;
    const int g = threadIdx.x / P, p = threadIdx.x - g * P;
    if (g >= G) return;
    const long tiles = (long)kgroups * tgroups;
    const long tile = (long)blockIdx.x * G + g;                        // over (r, tg, kg)
    const long r = tile / tiles, b = r / crops;
    if (r >= rows) return;
    const int tg = (int)((tile - r * tiles) / kgroups), kg = (int)((tile - r * tiles) % kgroups);
    const int t0 = tg * TT, k0 = kg * KT;
    float acc[TT][KT];
#pragma unroll
    for (int i = 0; i < TT; ++i)
#pragma unroll
        for (int j = 0; j < KT; ++j) acc[i][j] = 0.0f;
    for (int c = 0; c < C; ++c) {
        const float* gzp = gz + ((r * C + c) * K + k0) * (long)P + p;
        const float* zp = z + ((r * C + c) * K + k0) * (long)P + p;
        const float* xq = x + ((b * C + c) * Tf + t0) * (long)P + p;
        float gv[KT], zv[KT], xv[TT];
#pragma unroll
        for (int j = 0; j < KT; ++j) { const bool ok = k0 + j < K; gv[j] = ok ? gzp[(long)j * P] : 0.0f; zv[j] = ok ? zp[(long)j * P] : 0.0f; }
#pragma unroll
        for (int i = 0; i < TT; ++i) xv[i] = t0 + i < Tf ? xq[(long)i * P] : 0.0f;
#pragma unroll
        for (int i = 0; i < TT; ++i)
#pragma unroll
            for (int j = 0; j < KT; ++j) acc[i][j] = fmaf(gv[j], xv[i] - zv[j], acc[i][j]);
    }
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        if (k0 + j < K) {
            const float id = 1.0f / den[(r * K + k0 + j) * (long)P + p];
#pragma unroll
            for (int i = 0; i < TT; ++i)
                if (t0 + i < Tf) dw[((r * Tf + t0 + i) * (long)K + k0 + j) * P + p] = acc[i][j] * id;
        }
    }
}

// gx[b,c,t,p] = at[b,t,p] mask[b,t] sum_{crop} sum_k (gz/den)[r,c,k,p] * GX[r,t,k]
__global__ __launch_bounds__(256) void fusion_gather_bwd_x_kernel(const float* __restrict__ gz, const float* __restrict__ den,
                                                                  const float* __restrict__ at_raw, const float* __restrict__ at_bias,
                                                                  const float* __restrict__ GX, const float* __restrict__ mask,
                                                                  float* __restrict__ gx, int crops, int C, int Tf, int K, int P,
                                                                  long total) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int p = (int)(e % P), t = (int)((e / P) % Tf), c = (int)((e / ((long)P * Tf)) % C);
    const long b = e / ((long)P * Tf * C);
    const float ab = at_bias ? at_bias[0] : 0.0f;
    const float a0 = mask[b * Tf + t] / (1.0f + expf(-(at_raw[(b * Tf + t) * (long)P + p] + ab)));
    float acc = 0.f;
    for (int j = 0; j < crops; ++j) {
        const long r = b * crops + j;
        const float* gp = GX + (r * Tf + t) * (long)K;
        const float* zp = gz + ((r * C + c) * K) * (long)P + p;
        const float* dp = den + r * K * (long)P + p;
        for (int k = 0; k < K; ++k) acc = fmaf(zp[(long)k * P] / dp[(long)k * P], gp[k], acc);
    }
    gx[e] = acc * a0;
}

// dw[r,t,k,p] = sum_c (gz/den)[r,c,k,p] * (x[b,c,t,p] - z[r,c,k,p])      (d num and d den together)
__global__ __launch_bounds__(256) void fusion_gather_bwd_w_kernel(const float* __restrict__ gz, const float* __restrict__ z,
                                                                  const float* __restrict__ den, const float* __restrict__ x,
                                                                  float* __restrict__ dw, int crops, int C, int Tf, int K, int P,
                                                                  long total) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int p = (int)(e % P), k = (int)((e / P) % K), t = (int)((e / ((long)P * K)) % Tf);
    const long r = e / ((long)P * K * Tf), b = r / crops;
    const float id = 1.0f / den[(r * K + k) * (long)P + p];
    float acc = 0.f;
    for (int c = 0; c < C; ++c) {
        const long o = ((r * C + c) * K + k) * (long)P + p;
        acc = fmaf(gz[o], x[((b * C + c) * Tf + t) * (long)P + p] - z[o], acc);
    }
    dw[e] = acc * id;
}

// One workgroup per (b, t): the two contractions of dw that autograd of `at * GX * mask` needs, in a fixed order:
//   g_at_raw[b,t,p] = at (1 - at) mask[b,t] * sum_{crop,k} dw[r,t,k,p] GX[r,t,k]          (through the sigmoid)
//   gGX[r,t,k]      = mask[b,t] * sum_p dw[r,t,k,p] at[b,t,p]
__global__ __launch_bounds__(256) void fusion_gather_bwd_reduce_kernel(const float* __restrict__ dw, const float* __restrict__ at_raw,
                                                                       const float* __restrict__ at_bias, const float* __restrict__ GX,
                                                                       const float* __restrict__ mask, float* __restrict__ gat,
                                                                       float* __restrict__ gGX, int crops, int Tf, int K, int P) {
    extern __shared__ float sat[];     // at[b,t,:]
    const long b = blockIdx.x / Tf;
    const int t = blockIdx.x % Tf;
    const float mk = mask[b * Tf + t];
    const float ab = at_bias ? at_bias[0] : 0.0f;
    for (int p = threadIdx.x; p < P; p += 256) sat[p] = 1.0f / (1.0f + expf(-(at_raw[(b * Tf + t) * (long)P + p] + ab)));
    __syncthreads();
    if (gat) {
        for (int p = threadIdx.x; p < P; p += 256) {
            float acc = 0.f;
            for (int j = 0; j < crops; ++j) {
                const long r = b * crops + j;
                const float* dp = dw + ((r * Tf + t) * (long)K) * P + p;
                const float* gp = GX + (r * Tf + t) * (long)K;
                for (int k = 0; k < K; ++k) acc = fmaf(dp[(long)k * P], gp[k], acc);
            }
            const float a = sat[p];
            gat[(b * Tf + t) * (long)P + p] = acc * mk * a * (1.0f - a);
        }
    }
    if (gGX) {
        for (int e = threadIdx.x; e < crops * K; e += 256) {
            const int j = e / K, k = e - j * K;
            const long r = b * crops + j;
            const float* dp = dw + ((r * Tf + t) * (long)K + k) * P;
            float acc = 0.f;
            for (int p = 0; p < P; ++p) acc = fmaf(dp[p], sat[p], acc);
            gGX[(r * Tf + t) * (long)K + k] = acc * mk;
        }
    }
}

extern "C" int cfn_fusion_gather_fwd(const float* x, const float* at_raw, const float* at_bias, const float* GX, const float* mask,
                                     float* z, float* den, int B, int crops, int C, int Tf, int K, int P, void* stream) {
    CFN_REQUIRE(x && at_raw && GX && mask && z && den, "cfn_fusion_gather_fwd: null tensor");
    CFN_REQUIRE(crops >= 1, "cfn_fusion_gather_fwd: crops must be >= 1");
    const long total = (long)B * crops * C * K * P;
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_FUSION, st, 4.0 * B * ((double)C * Tf * P + (double)crops * C * K * P));
    static const int tiled = getenv("CFN_FUSION_TILED") ? atoi(getenv("CFN_FUSION_TILED")) : 1;
    if (tiled && P <= 256) {
        // k tile: 13 (K = 65: T = 256), 9 (K = 17: T = 64) or 8 -- the one that wastes the fewest slots
        const int G = 256 / P, CT = 4;
        const int w13 = cfn_cdiv(K, 13) * 13 - K, w9 = cfn_cdiv(K, 9) * 9 - K, w8 = cfn_cdiv(K, 8) * 8 - K;
        const int KT = (w13 <= w9 && w13 <= w8) ? 13 : (w9 <= w8 ? 9 : 8);
        const int kgroups = cfn_cdiv(K, KT), cblocks = cfn_cdiv(C, G * CT);
        const long blocks = (long)B * crops * kgroups * cblocks;
        if (blocks < 0x7fffffffL) {
            if (KT == 13) hipLaunchKernelGGL((fusion_gather_fwd_tiled_kernel<4, 13>), dim3((unsigned)blocks), dim3(256), 0, st, x, at_raw, at_bias, GX, mask, z, den, crops, C, Tf, K, P, kgroups, cblocks);
            else if (KT == 9) hipLaunchKernelGGL((fusion_gather_fwd_tiled_kernel<4, 9>), dim3((unsigned)blocks), dim3(256), 0, st, x, at_raw, at_bias, GX, mask, z, den, crops, C, Tf, K, P, kgroups, cblocks);
            else hipLaunchKernelGGL((fusion_gather_fwd_tiled_kernel<4, 8>), dim3((unsigned)blocks), dim3(256), 0, st, x, at_raw, at_bias, GX, mask, z, den, crops, C, Tf, K, P, kgroups, cblocks);
            return cfn_check_launch("fusion_gather_fwd");
        }
    }
    hipLaunchKernelGGL(fusion_gather_fwd_kernel, dim3(cfn_cdiv(total, 256)), dim3(256), 0, st, x, at_raw, at_bias, GX, mask, z, den,
                       crops, C, Tf, K, P, total);
    return cfn_check_launch("fusion_gather_fwd");
}

extern "C" int cfn_fusion_gather_bwd(const float* gz, const float* z, const float* den, const float* x, const float* at_raw,
                                     const float* at_bias, const float* GX, const float* mask, float* gx, float* gat, float* gGX,
                                     float* dw, int B, int crops, int C, int Tf, int K, int P, void* stream) {
    CFN_REQUIRE(gz && z && den && x && at_raw && GX && mask && dw, "cfn_fusion_gather_bwd: null tensor");
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_FUSION, st, 4.0 * B * ((double)C * Tf * P * 2 + (double)crops * C * K * P * 2));
    if (gx) {
        const long total = (long)B * C * Tf * P;
        hipLaunchKernelGGL(fusion_gather_bwd_x_kernel, dim3(cfn_cdiv(total, 256)), dim3(256), 0, st, gz, den, at_raw, at_bias, GX, mask,
                           gx, crops, C, Tf, K, P, total);
    }
    if (gat || gGX) {
        const long total = (long)B * crops * Tf * K * P;
        static const int tiled = getenv("CFN_FUSION_TILED") ? atoi(getenv("CFN_FUSION_TILED")) : 1;
        const int G = P <= 256 ? 256 / P : 0;
        const int kgroups = cfn_cdiv(K, 8), tgroups = cfn_cdiv(Tf, 8);
        const long nblk = G ? cfn_cdiv((long)B * crops * kgroups * tgroups, G) : 0;
        if (tiled && G && nblk < 0x7fffffffL)
            hipLaunchKernelGGL((fusion_gather_bwd_w_tiled_kernel<8, 8>), dim3((unsigned)nblk), dim3(256), 0, st, gz, z, den, x, dw,
                               crops, C, Tf, K, P, kgroups, tgroups, B * crops);
        else
        hipLaunchKernelGGL(fusion_gather_bwd_w_kernel, dim3(cfn_cdiv(total, 256)), dim3(256), 0, st, gz, z, den, x, dw, crops, C, Tf, K,
                           P, total);
        hipLaunchKernelGGL(fusion_gather_bwd_reduce_kernel, dim3(B * Tf), dim3(256), P * sizeof(float), st, dw, at_raw, at_bias, GX,
                           mask, gat, gGX, crops, Tf, K, P);
    }
    return cfn_check_launch("fusion_gather_bwd");
}
