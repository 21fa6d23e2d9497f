// Coarse-stream specific kernels:
//   * data gradient of the dense (Grid Pool saliency) convolutions, gather form by stride-parity class;
//   * the Multi-stage Fusion temporal-alignment gather of RewightLayer (x3d_coarse.py:199-226), evaluated at the
//     fine features' native 7x7 resolution: the reference first up-samples them with adaptive_max_pool2d
//     (7 -> 56/28/14: each output cell copies exactly one input cell) and materialises a
//     (B,C,T',K,h,w) product; every quantity is constant over the (h/7 x w/7) blocks, so the 7x7
//     result up-sampled is identical (measured 3e-8).
#include "cfn_common.h"

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
            atomicAdd(&a.gA[nci], (double)(sh[0] + sh[1] + sh[2] + sh[3]));
            atomicAdd(&a.gB[nci], (double)(sh[4] + sh[5] + sh[6] + sh[7]));
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
            atomicAdd(&a.gA[(long)n * a.Cin + i], (double)((red[2 * i] + red[2 * CI + 2 * i]) + (red[4 * CI + 2 * i] + red[6 * CI + 2 * i])));
            atomicAdd(&a.gB[(long)n * a.Cin + i], (double)((red[2 * i + 1] + red[2 * CI + 2 * i + 1]) + (red[4 * CI + 2 * i + 1] + red[6 * CI + 2 * i + 1])));
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
// fusion gather.  x (B,C,Tf,P) fine features, at (B,Tf,P) attention, gm (B,Tf,K) = Gaussian alignment * mask.
//   w[b,t,k,p] = at[b,t,p] * gm[b,t,k];  den[b,k,p] = sum_t w + 1e-6;  z[b,c,k,p] = sum_t x*w / den
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fusion_gather_fwd_kernel(const float* __restrict__ x, const float* __restrict__ at,
                                                                const float* __restrict__ gm, float* __restrict__ z,
                                                                float* __restrict__ den, int C, int Tf, int K, int P, long total) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int p = (int)(e % P), k = (int)((e / P) % K), c = (int)((e / ((long)P * K)) % C);
    const long b = e / ((long)P * K * C);
    const float* xp = x + ((b * C + c) * Tf) * (long)P + p;
    const float* ap = at + b * Tf * (long)P + p;
    const float* gp = gm + b * Tf * (long)K + k;
    float num = 0.f, d = 0.f;
    for (int t = 0; t < Tf; ++t) {
        const float w = ap[(long)t * P] * gp[(long)t * K];
        num = fmaf(xp[(long)t * P], w, num);
        d += w;
    }
    d += 1e-6f;
    z[e] = num / d;
    if (c == 0) den[(b * K + k) * (long)P + p] = d;
}

// gx[b,c,t,p] = sum_k (gz/den)[b,c,k,p] * w[b,t,k,p]
__global__ __launch_bounds__(256) void fusion_gather_bwd_x_kernel(const float* __restrict__ gz, const float* __restrict__ den,
                                                                  const float* __restrict__ at, const float* __restrict__ gm,
                                                                  float* __restrict__ gx, int C, int Tf, int K, int P, long total) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int p = (int)(e % P), t = (int)((e / P) % Tf), c = (int)((e / ((long)P * Tf)) % C);
    const long b = e / ((long)P * Tf * C);
    const float a0 = at[(b * Tf + t) * (long)P + p];
    const float* gp = gm + (b * Tf + t) * (long)K;
    const float* zp = gz + ((b * C + c) * K) * (long)P + p;
    const float* dp = den + b * K * (long)P + p;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc = fmaf(zp[(long)k * P] / dp[(long)k * P], gp[k], acc);
    gx[e] = acc * a0;
}

// dw[b,t,k,p] = sum_c (gz/den)[b,c,k,p] * (x[b,c,t,p] - z[b,c,k,p])      (d num and d den together)
__global__ __launch_bounds__(256) void fusion_gather_bwd_w_kernel(const float* __restrict__ gz, const float* __restrict__ z,
                                                                  const float* __restrict__ den, const float* __restrict__ x,
                                                                  float* __restrict__ dw, int C, int Tf, int K, int P, long total) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int p = (int)(e % P), k = (int)((e / P) % K), t = (int)((e / ((long)P * K)) % Tf);
    const long b = e / ((long)P * K * Tf);
    const float id = 1.0f / den[(b * K + k) * (long)P + p];
    float acc = 0.f;
    for (int c = 0; c < C; ++c) {
        const long o = ((b * C + c) * K + k) * (long)P + p;
        acc = fmaf(gz[o], x[((b * C + c) * Tf + t) * (long)P + p] - z[o], acc);
    }
    dw[e] = acc * id;
}

extern "C" int cfn_fusion_gather_fwd(const float* x, const float* at, const float* gm, float* z, float* den, int B, int C,
                                     int Tf, int K, int P, void* stream) {
    CFN_REQUIRE(x && at && gm && z && den, "cfn_fusion_gather_fwd: null tensor");
    const long total = (long)B * C * K * P;
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_FUSION, st, 4.0 * B * ((double)C * Tf * P + (double)C * K * P));
    hipLaunchKernelGGL(fusion_gather_fwd_kernel, dim3(cfn_cdiv(total, 256)), dim3(256), 0, st, x, at, gm, z, den, C, Tf, K, P, total);
    return cfn_check_launch("fusion_gather_fwd");
}

extern "C" int cfn_fusion_gather_bwd(const float* gz, const float* z, const float* den, const float* x, const float* at,
                                     const float* gm, float* gx, float* dw, int B, int C, int Tf, int K, int P, void* stream) {
    CFN_REQUIRE(gz && z && den && x && at && gm && dw, "cfn_fusion_gather_bwd: null tensor");
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_FUSION, st, 4.0 * B * ((double)C * Tf * P * 2 + (double)C * K * P * 2));
    if (gx) {
        const long total = (long)B * C * Tf * P;
        hipLaunchKernelGGL(fusion_gather_bwd_x_kernel, dim3(cfn_cdiv(total, 256)), dim3(256), 0, st, gz, den, at, gm, gx, C, Tf, K, P, total);
    }
    const long total = (long)B * Tf * K * P;
    hipLaunchKernelGGL(fusion_gather_bwd_w_kernel, dim3(cfn_cdiv(total, 256)), dim3(256), 0, st, gz, z, den, x, dw, C, Tf, K, P, total);
    return cfn_check_launch("fusion_gather_bwd");
}
