// SubBatchNorm3d statistics -> prologue coefficients, fused with the squeeze-excite gate.
//
// The convolution kernels deliver fp64 per-(n,c) sums of their raw output y (sum, sum of squares).
// This single small kernel turns them into the per-(n,c) affine (A,B) the next kernel applies at load
// time, reproducing SubBatchNorm3d.forward (x3d_fine.py:51-62: batch_norm over split groups + shared
// affine), the running-statistics update of nn.BatchNorm3d (momentum, unbiased variance) and, for the
// even-indexed bottlenecks, the SE branch (x3d_fine.py:157-163: global mean -> fc1 -> relu -> fc2 ->
// sigmoid -> scale), whose global average pool of bn2(y) equals A*mean(y)+B.  The backward kernel is the
// exact adjoint (statistics are differentiated: gsum/gsumsq flow back into the producing conv).
// One workgroup; all arithmetic that feeds the normalisation is fp64.  The SE matrices are staged in LDS and the
// matrix-vector products are wave cooperative, so the kernel is a handful of dependent memory latencies long.
#include "cfn_common.h"
#include <stdlib.h>

struct BnFoldArgs {
    const double* s; const double* q;       // (N,C) sums of y, y*y over `count` positions (training; SE needs s too)
    const float* gamma; const float* beta;  // (C) or null (affine=False)
    float* run_mean; float* run_var;        // training: split_bn buffers (S*C) updated in place; eval: bn buffers (C)
    long* nbt;                              // num_batches_tracked (training) or null
    int training, N, C, S, Wd;
    int stage;                              // SE matrices staged in LDS (0: they do not fit -- X3D-XL -- and are read through L2)
    double count, pool_count, eps, momentum;
    const float* w1; const float* b1; const float* w2; const float* b2;   // SE (Wd > 0): fc1 (Wd,C), fc2 (C,Wd)
    double* A; double* B;                     // (N,C) outputs (gated when SE)
    double* mean; double* rstd;             // (S,C) saved
    float* A0; float* B0; float* gate; float* hbuf; float* pooled;   // SE saved: (N,C),(N,C),(N,C),(N,Wd),(N,C)
};

#define BNF_NB 8   // samples per squeeze-excite pass (LDS tables are sized for this many; 8: a batch of 8 clips is ONE pass of the backward)

// stage the two SE matrices in LDS: w1s[j*C + c] (rows contiguous), w2s[c*(Wd+1) + j] (odd pitch => lanes along c
// hit distinct banks)
__device__ __forceinline__ void bnf_stage_se(const float* w1, const float* w2, float* w1s, float* w2s, int C, int Wd) {
    for (int e = threadIdx.x; e < C * Wd; e += blockDim.x) {
        w1s[e] = w1[e];
        const int c = e / Wd, j = e - c * Wd;
        w2s[c * (Wd + 1) + j] = w2[e];
    }
}

__global__ __launch_bounds__(1024) void bn_fold_fwd_kernel(const BnFoldArgs a) {
    const int tid = threadIdx.x, nthr = blockDim.x, N = a.N, C = a.C, S = a.training ? a.S : 1, G = N / S, Wd = a.Wd;
    extern __shared__ float sh[];      // [w1s[Wd*C] | w2s[C*(Wd+1)]] | pooled[NB*C] | h[NB*Wd]
    const bool st = Wd > 0 && a.stage;
    float* sp = sh + (st ? Wd * C + C * (Wd + 1) : 0);
    float* shh = sp + BNF_NB * C;
    const float* w1s = st ? sh : a.w1;
    const float* w2s = st ? sh + Wd * C : a.w2;
    const int w2p = st ? Wd + 1 : Wd;                  // row pitch of w2s
    if (st) bnf_stage_se(a.w1, a.w2, sh, sh + Wd * C, C, Wd);
    // (1) statistics per (split group, channel); without a squeeze-excite gate the launch is several small workgroups (one element per thread:
    // the fp64 divide / sqrt chain of an element is ~2 us, a single workgroup looping over 8 x 432 elements was 10 us on the critical path)
    // With a gate and several workgroups (one per SAMPLE: the gate of a sample depends on nothing but its own statistics row) a workgroup
    // computes the statistics of its sample's split group only; the group's first sample writes them.
    const bool ps = Wd > 0 && gridDim.x > 1;
    const int gown = ps ? (int)(blockIdx.x % S) : 0;
    const int gtid = Wd > 0 ? tid : (int)(blockIdx.x * blockDim.x) + tid, gn = Wd > 0 ? nthr : (int)(gridDim.x * blockDim.x);
    for (int e = ps ? gown * C + tid : gtid; e < (ps ? (gown + 1) * C : S * C); e += gn) {
        const bool wr = !ps || (int)blockIdx.x == gown;
        const int g = e / C, c = e - g * C;
        double mean, var;
        if (a.training) {
            double ss = 0.0, qq = 0.0;
            for (int i = 0; i < G; ++i) { ss += a.s[(long)(i * S + g) * C + c]; qq += a.q[(long)(i * S + g) * C + c]; }
            const double cnt = a.count * G;
            mean = ss / cnt;
            var = qq / cnt - mean * mean;
            if (var < 0.0) var = 0.0;
            const double unb = var * (cnt / (cnt > 1.0 ? cnt - 1.0 : 1.0));
            if (wr) {
                a.run_mean[e] = (float)((1.0 - a.momentum) * (double)a.run_mean[e] + a.momentum * mean);
                a.run_var[e] = (float)((1.0 - a.momentum) * (double)a.run_var[e] + a.momentum * unb);
            }
        } else {
            mean = (double)a.run_mean[c];
            var = (double)a.run_var[c];
        }
        const double rstd = 1.0 / sqrt(var + a.eps);
        if (wr) { a.mean[e] = mean; a.rstd[e] = rstd; }
        const double ga = a.gamma ? (double)a.gamma[c] : 1.0, be = a.beta ? (double)a.beta[c] : 0.0;
        const float av = (float)(ga * rstd), bv = (float)(be - mean * ga * rstd);
        for (int i = 0; i < G; ++i) {
            if (ps && i * S + g != (int)blockIdx.x) continue;            // this workgroup's sample only
            const long o = (long)(i * S + g) * C + c;
            if (Wd > 0) { a.A0[o] = av; a.B0[o] = bv; } else { a.A[o] = av; a.B[o] = bv; }
        }
    }
    if (a.training && a.nbt && gtid == 0 && (!ps || blockIdx.x == 0)) a.nbt[0] += 1;
    if (Wd <= 0) return;
    __syncthreads();
    // (2) squeeze-excite gate, BNF_NB samples per pass; dot products are wave cooperative (lanes along the long axis)
    const int wave = tid >> 6, lane = tid & 63, nwaves = nthr >> 6;
    for (int n0 = ps ? (int)blockIdx.x : 0; n0 < (ps ? (int)blockIdx.x + 1 : N); n0 += BNF_NB) {
        const int nb = ps ? 1 : min(BNF_NB, N - n0);
        for (int e = tid; e < nb * C; e += nthr) {
            const long o = (long)n0 * C + e;
            const float pv = (float)(a.s[o] / a.pool_count) * a.A0[o] + a.B0[o];
            sp[e] = pv;
            a.pooled[o] = pv;
        }
        __syncthreads();
        for (int p = wave; p < nb * Wd; p += nwaves) {          // h[n][j] = relu(b1[j] + w1[j,:] . pooled[n,:])
            const int nl = p / Wd, j = p - nl * Wd;
            float acc = 0.0f;
            for (int c = lane; c < C; c += 64) acc = fmaf(w1s[j * C + c], sp[nl * C + c], acc);
            acc = cfn_wave_sum(acc);
            if (lane == 0) {
                acc = fmaxf(acc + a.b1[j], 0.0f);
                shh[p] = acc;
                a.hbuf[(long)(n0 + nl) * Wd + j] = acc;
            }
        }
        __syncthreads();
        for (int e = tid; e < nb * C; e += nthr) {               // gate[n][c] = sigmoid(b2[c] + w2[c,:] . h[n,:])
            const int nl = e / C, c = e - nl * C;
            float acc = a.b2[c];
            for (int j = 0; j < Wd; ++j) acc = fmaf(w2s[c * w2p + j], shh[nl * Wd + j], acc);
            const float gt = 1.0f / (1.0f + expf(-acc));
            const long o = (long)n0 * C + e;
            a.gate[o] = gt;
            a.A[o] = a.A0[o] * gt;
            a.B[o] = a.B0[o] * gt;
        }
        __syncthreads();
    }
}

struct BnFoldBwdArgs {
    const double* gA; const double* gB;       // (N,C) incoming gradients of the outputs
    const double* s;                        // (N,C) (SE)
    const float* gamma;
    const double* mean; const double* rstd; // (S,C)
    const float* A0; const float* B0; const float* gate; const float* hbuf; const float* pooled;
    const float* w1; const float* w2;
    int training, N, C, S, Wd;
    int stage;                              // SE matrices staged in LDS (0: they do not fit -- X3D-XL -- and are read through L2)
    double count, pool_count;
    double* gs; double* gq;                 // (N,C) outputs (training) or null
    float* ggamma; float* gbeta;            // (C) outputs or null
    float* gw1; float* gb1; float* gw2; float* gb2;   // SE parameter gradients (overwritten)
    double* tA; double* tB;                   // (N,C) scratch: gradients w.r.t. the un-gated A0/B0
};

__global__ __launch_bounds__(1024) void bn_fold_bwd_kernel(const BnFoldBwdArgs a) {
    const int tid = threadIdx.x, nthr = blockDim.x, N = a.N, C = a.C, S = a.training ? a.S : 1, G = N / S, Wd = a.Wd;
    extern __shared__ float sh[];      // w1s[Wd*C] | w2s[C*(Wd+1)] | z[NB*C] | pooled[NB*C] | gz1[NB*Wd] | h[NB*Wd]
    // (2') squeeze-excite adjoint -> gradients w.r.t. A0, B0, s (through pooled) and the SE parameters
    if (Wd > 0) {
        const bool st = a.stage != 0;
        const float* w1s = st ? sh : a.w1;
        const float* w2s = st ? sh + Wd * C : a.w2;
        const int w2p = st ? Wd + 1 : Wd;
        float* gz2 = sh + (st ? Wd * C + C * (Wd + 1) : 0);
        float* pl = gz2 + BNF_NB * C;
        float* gz1 = pl + BNF_NB * C;
        float* shh = gz1 + BNF_NB * Wd;
        if (st) bnf_stage_se(a.w1, a.w2, sh, sh + Wd * C, C, Wd);
        const int wave = tid >> 6, lane = tid & 63, nwaves = nthr >> 6;
        for (int n0 = 0; n0 < N; n0 += BNF_NB) {
            const int nb = min(BNF_NB, N - n0);
            const bool first = n0 == 0;
            for (int e = tid; e < nb * Wd; e += nthr) shh[e] = a.hbuf[(long)n0 * Wd + e];
            for (int e = tid; e < nb * C; e += nthr) {
                const long o = (long)n0 * C + e;
                const float gt = a.gate[o];
                const float ggate = a.gA[o] * a.A0[o] + a.gB[o] * a.B0[o];
                gz2[e] = ggate * gt * (1.0f - gt);
                pl[e] = a.pooled[o];
                a.tA[o] = a.gA[o] * gt;
                a.tB[o] = a.gB[o] * gt;
            }
            __syncthreads();
            for (int e = tid; e < C * Wd; e += nthr) {            // gw2[c][j] += sum_n gz2[n][c] * h[n][j]
                const int c = e / Wd, j = e - c * Wd;
                float acc = 0.0f;
                for (int nl = 0; nl < nb; ++nl) acc = fmaf(gz2[nl * C + c], shh[nl * Wd + j], acc);
                a.gw2[e] = first ? acc : a.gw2[e] + acc;
            }
            for (int c = tid; c < C; c += nthr) {
                float acc = 0.0f;
                for (int nl = 0; nl < nb; ++nl) acc += gz2[nl * C + c];
                a.gb2[c] = first ? acc : a.gb2[c] + acc;
            }
            for (int p = wave; p < nb * Wd; p += nwaves) {       // gz1[n][j] = relu'(h) * gz2[n,:] . w2[:,j]
                const int nl = p / Wd, j = p - nl * Wd;
                float acc = 0.0f;
                for (int c = lane; c < C; c += 64) acc = fmaf(gz2[nl * C + c], w2s[c * w2p + j], acc);
                acc = cfn_wave_sum(acc);
                if (lane == 0) gz1[p] = shh[p] > 0.0f ? acc : 0.0f;
            }
            __syncthreads();
            for (int e = tid; e < Wd * C; e += nthr) {            // gw1[j][c] += sum_n gz1[n][j] * pooled[n][c]
                const int j = e / C, c = e - j * C;
                float acc = 0.0f;
                for (int nl = 0; nl < nb; ++nl) acc = fmaf(gz1[nl * Wd + j], pl[nl * C + c], acc);
                a.gw1[e] = first ? acc : a.gw1[e] + acc;
            }
            for (int j = tid; j < Wd; j += nthr) {
                float acc = 0.0f;
                for (int nl = 0; nl < nb; ++nl) acc += gz1[nl * Wd + j];
                a.gb1[j] = first ? acc : a.gb1[j] + acc;
            }
            for (int e = tid; e < nb * C; e += nthr) {            // gradient of pooled[n][c]
                const int nl = e / C, c = e - nl * C;
                float gp = 0.0f;
                for (int j = 0; j < Wd; ++j) gp = fmaf(gz1[nl * Wd + j], w1s[j * C + c], gp);
                const long o = (long)n0 * C + e;
                const float sm = (float)(a.s[o] / a.pool_count);
                a.tA[o] += gp * sm;
                a.tB[o] += gp;
                if (a.gs) a.gs[o] = (double)(gp * a.A0[o]) / a.pool_count;     // direct path s -> pooled
            }
            __syncthreads();
        }
    }
    const double* dA = Wd > 0 ? a.tA : a.gA;
    const double* dB = Wd > 0 ? a.tB : a.gB;
    // (1') batch-norm adjoint per channel (no gate: several small workgroups, one channel per thread)
    const int gtid = Wd > 0 ? tid : (int)(blockIdx.x * blockDim.x) + tid, gn = Wd > 0 ? nthr : (int)(gridDim.x * blockDim.x);
    for (int c = gtid; c < C; c += gn) {
        double ggam = 0.0, gbet = 0.0;
        for (int g = 0; g < S; ++g) {
            double ga = 0.0, gb = 0.0;
            for (int i = 0; i < G; ++i) { ga += dA[(long)(i * S + g) * C + c]; gb += dB[(long)(i * S + g) * C + c]; }
            const double mean = a.mean[g * C + c], rstd = a.rstd[g * C + c];
            const double gam = a.gamma ? (double)a.gamma[c] : 1.0;
            ggam += (ga - gb * mean) * rstd;
            gbet += gb;
            if (a.training && a.gs) {
                const double cnt = a.count * G;
                const double g_rstd = (ga - gb * mean) * gam;
                const double g_var = g_rstd * (-0.5 * rstd * rstd * rstd);
                const double g_mean = -gb * gam * rstd - 2.0 * mean * g_var;
                for (int i = 0; i < G; ++i) {
                    const long o = (long)(i * S + g) * C + c;
                    a.gs[o] = (Wd > 0 ? a.gs[o] : 0.0) + g_mean / cnt;
                    a.gq[o] = g_var / cnt;
                }
            }
        }
        if (a.ggamma) { a.ggamma[c] = (float)ggam; a.gbeta[c] = (float)gbet; }
    }
}

extern "C" int cfn_bn_fold_fwd(const double* s, const double* q, const float* gamma, const float* beta, float* run_mean,
                               float* run_var, long* nbt, int training, int N, int C, int S, double count, double eps,
                               double momentum, const float* w1, const float* b1, const float* w2, const float* b2, int Wd,
                               double pool_count, double* A, double* B, double* mean, double* rstd, float* A0, float* B0,
                               float* gate, float* hbuf, float* pooled, void* stream) {
    CFN_REQUIRE(A && B && mean && rstd && run_mean && run_var, "cfn_bn_fold_fwd: null tensor");
    CFN_REQUIRE(!training || (s && q), "cfn_bn_fold_fwd: training needs sum / sumsq");
    CFN_REQUIRE(N > 0 && C > 0 && S > 0, "cfn_bn_fold_fwd: bad sizes");
    CFN_REQUIRE(!training || N % S == 0, "cfn_bn_fold_fwd: batch %d not divisible by %d splits", N, S);   // eval: no split groups
    CFN_REQUIRE(Wd <= 0 || (w1 && b1 && w2 && b2 && s && A0 && B0 && gate && hbuf && pooled), "cfn_bn_fold_fwd: SE needs its tensors");
    const size_t lds_small = Wd > 0 ? BNF_NB * (size_t)(C + Wd) * sizeof(float) : 0;
    const size_t lds_full = Wd > 0 ? lds_small + ((size_t)Wd * C + (size_t)C * (Wd + 1)) * sizeof(float) : 0;
    const int stage = lds_full <= 150 * 1024;
    const size_t lds = stage ? lds_full : lds_small;
    BnFoldArgs a = {s, q, gamma, beta, run_mean, run_var, nbt, training, N, C, S, Wd, stage, count, pool_count, eps, momentum,
                    w1, b1, w2, b2, A, B, mean, rstd, A0, B0, gate, hbuf, pooled};
    CFN_REQUIRE(lds <= 150 * 1024, "cfn_bn_fold_fwd: per-sample SE tables (C=%d, width=%d) exceed LDS", C, Wd);
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)bn_fold_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int Se = training ? S : 1;
    static const int per_sample = getenv("CFN_BNFOLD_PS") ? atoi(getenv("CFN_BNFOLD_PS")) : 1;     // gate: one workgroup per sample (0: one for all)
    hipLaunchKernelGGL(bn_fold_fwd_kernel, dim3(Wd > 0 ? (per_sample ? N : 1) : cfn_cdiv((long)Se * C, 64)), dim3(Wd > 0 ? 1024 : 64), lds, (hipStream_t)stream, a);
    return cfn_check_launch("bn_fold_fwd");
}

extern "C" int cfn_bn_fold_bwd(const double* gA, const double* gB, const double* s, const float* gamma, const double* mean,
                               const double* rstd, const float* A0, const float* B0, const float* gate, const float* hbuf,
                               const float* pooled, const float* w1, const float* w2, int training, int N, int C, int S,
                               int Wd, double count, double pool_count, double* gs, double* gq, float* ggamma, float* gbeta,
                               float* gw1, float* gb1, float* gw2, float* gb2, double* tA, double* tB, void* stream) {
    CFN_REQUIRE(gA && gB && mean && rstd, "cfn_bn_fold_bwd: null tensor");
    CFN_REQUIRE(Wd <= 0 || (s && A0 && B0 && gate && hbuf && pooled && w1 && w2 && gw1 && gb1 && gw2 && gb2 && tA && tB),
                "cfn_bn_fold_bwd: SE needs its tensors");
    CFN_REQUIRE(!training || ((gs == nullptr) == (gq == nullptr)), "cfn_bn_fold_bwd: gs/gq mismatch");   // eval + SE: gs alone
    const size_t lds_small = Wd > 0 ? 2 * BNF_NB * (size_t)(C + Wd) * sizeof(float) : 0;
    const size_t lds_full = Wd > 0 ? lds_small + ((size_t)Wd * C + (size_t)C * (Wd + 1)) * sizeof(float) : 0;
    const int stage = lds_full <= 150 * 1024;
    const size_t lds = stage ? lds_full : lds_small;
    BnFoldBwdArgs a = {gA, gB, s, gamma, mean, rstd, A0, B0, gate, hbuf, pooled, w1, w2, training, N, C, S, Wd, stage, count,
                       pool_count, gs, gq, ggamma, gbeta, gw1, gb1, gw2, gb2, tA, tB};
    CFN_REQUIRE(lds <= 150 * 1024, "cfn_bn_fold_bwd: per-sample SE tables (C=%d, width=%d) exceed LDS", C, Wd);
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)bn_fold_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(bn_fold_bwd_kernel, dim3(Wd > 0 ? 1 : cfn_cdiv(C, 64)), dim3(Wd > 0 ? 1024 : 64), lds, (hipStream_t)stream, a);
    return cfn_check_launch("bn_fold_bwd");
}
