// SubBatchNorm3d statistics -> prologue coefficients, fused with the squeeze-excite gate.
//
// The convolution kernels deliver fp64 per-(n,c) sums of their raw output y (sum, sum of squares).
// This single small kernel turns them into the per-(n,c) affine (A,B) the next kernel applies at load
// time, reproducing SubBatchNorm3d.forward (x3d_fine.py:51-62: batch_norm over split groups + shared
// affine), the running-statistics update of nn.BatchNorm3d (momentum, unbiased variance) and, for the
// even-indexed bottlenecks, the SE branch (x3d_fine.py:157-163: global mean -> fc1 -> relu -> fc2 ->
// sigmoid -> scale), whose global average pool of bn2(y) equals A*mean(y)+B.  The backward kernel is the
// exact adjoint (statistics are differentiated: gsum/gsumsq flow back into the producing conv).
// One workgroup; all arithmetic that feeds the normalisation is fp64.
#include "cfn_common.h"

struct BnFoldArgs {
    const double* s; const double* q;       // (N,C) sums of y, y*y over `count` positions (training; SE needs s too)
    const float* gamma; const float* beta;  // (C) or null (affine=False)
    float* run_mean; float* run_var;        // training: split_bn buffers (S*C) updated in place; eval: bn buffers (C)
    long* nbt;                              // num_batches_tracked (training) or null
    int training, N, C, S, Wd;
    double count, pool_count, eps, momentum;
    const float* w1; const float* b1; const float* w2; const float* b2;   // SE (Wd > 0): fc1 (Wd,C), fc2 (C,Wd)
    float* A; float* B;                     // (N,C) outputs (gated when SE)
    double* mean; double* rstd;             // (S,C) saved
    float* A0; float* B0; float* gate; float* hbuf; float* pooled;   // SE saved: (N,C),(N,C),(N,C),(N,Wd),(N,C)
};

__global__ __launch_bounds__(256) void bn_fold_fwd_kernel(const BnFoldArgs a) {
    const int tid = threadIdx.x, N = a.N, C = a.C, S = a.training ? a.S : 1, G = N / S;
    // (1) statistics per (split group, channel)
    for (int e = tid; e < S * C; e += 256) {
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
            a.run_mean[e] = (float)((1.0 - a.momentum) * (double)a.run_mean[e] + a.momentum * mean);
            a.run_var[e] = (float)((1.0 - a.momentum) * (double)a.run_var[e] + a.momentum * unb);
        } else {
            mean = (double)a.run_mean[c];
            var = (double)a.run_var[c];
        }
        const double rstd = 1.0 / sqrt(var + a.eps);
        a.mean[e] = mean;
        a.rstd[e] = rstd;
        const double ga = a.gamma ? (double)a.gamma[c] : 1.0, be = a.beta ? (double)a.beta[c] : 0.0;
        const float av = (float)(ga * rstd), bv = (float)(be - mean * ga * rstd);
        for (int i = 0; i < G; ++i) {
            const long o = (long)(i * S + g) * C + c;
            if (a.Wd > 0) { a.A0[o] = av; a.B0[o] = bv; } else { a.A[o] = av; a.B[o] = bv; }
        }
    }
    if (a.training && a.nbt && tid == 0) a.nbt[0] += 1;
    if (a.Wd <= 0) return;
    __syncthreads();
    // (2) squeeze-excite gate per sample
    extern __shared__ float sh[];      // pooled[C] | h[Wd]
    float* sp = sh;
    float* shh = sh + C;
    for (int n = 0; n < N; ++n) {
        for (int c = tid; c < C; c += 256) {
            const long o = (long)n * C + c;
            const float pv = (float)(a.s[o] / a.pool_count) * a.A0[o] + a.B0[o];
            sp[c] = pv;
            a.pooled[o] = pv;
        }
        __syncthreads();
        for (int j = tid; j < a.Wd; j += 256) {
            float acc = a.b1[j];
            for (int c = 0; c < C; ++c) acc = fmaf(a.w1[(long)j * C + c], sp[c], acc);
            acc = fmaxf(acc, 0.0f);
            shh[j] = acc;
            a.hbuf[(long)n * a.Wd + j] = acc;
        }
        __syncthreads();
        for (int c = tid; c < C; c += 256) {
            float acc = a.b2[c];
            for (int j = 0; j < a.Wd; ++j) acc = fmaf(a.w2[(long)c * a.Wd + j], shh[j], acc);
            const float gt = 1.0f / (1.0f + expf(-acc));
            const long o = (long)n * C + c;
            a.gate[o] = gt;
            a.A[o] = a.A0[o] * gt;
            a.B[o] = a.B0[o] * gt;
        }
        __syncthreads();
    }
}

struct BnFoldBwdArgs {
    const float* gA; const float* gB;       // (N,C) incoming gradients of the outputs
    const double* s;                        // (N,C) (SE)
    const float* gamma;
    const double* mean; const double* rstd; // (S,C)
    const float* A0; const float* B0; const float* gate; const float* hbuf; const float* pooled;
    const float* w1; const float* w2;
    int training, N, C, S, Wd;
    double count, pool_count;
    double* gs; double* gq;                 // (N,C) outputs (training) or null
    float* ggamma; float* gbeta;            // (C) outputs or null
    float* gw1; float* gb1; float* gw2; float* gb2;   // SE parameter gradients (zero-filled by caller)
    float* tA; float* tB;                   // (N,C) scratch: gradients w.r.t. the un-gated A0/B0
};

__global__ __launch_bounds__(256) void bn_fold_bwd_kernel(const BnFoldBwdArgs a) {
    const int tid = threadIdx.x, N = a.N, C = a.C, S = a.training ? a.S : 1, G = N / S;
    extern __shared__ float sh[];      // gz2[C] | gz1[Wd] | h[Wd]
    float* gz2 = sh;
    float* gz1 = sh + C;
    float* shh = gz1 + (a.Wd > 0 ? a.Wd : 0);
    // (2') squeeze-excite adjoint -> gradients w.r.t. A0, B0, s (through pooled)
    if (a.Wd > 0) {
        for (int n = 0; n < N; ++n) {
            for (int j = tid; j < a.Wd; j += 256) shh[j] = a.hbuf[(long)n * a.Wd + j];
            for (int c = tid; c < C; c += 256) {
                const long o = (long)n * C + c;
                const float gt = a.gate[o];
                const float ggate = a.gA[o] * a.A0[o] + a.gB[o] * a.B0[o];
                const float z = ggate * gt * (1.0f - gt);
                gz2[c] = z;
                a.tA[o] = a.gA[o] * gt;
                a.tB[o] = a.gB[o] * gt;
                atomicAdd(&a.gb2[c], z);
            }
            __syncthreads();
            for (int e = tid; e < C * a.Wd; e += 256) {         // gw2[c][j] += gz2[c] * h[j]
                const int c = e / a.Wd, j = e - c * a.Wd;
                a.gw2[e] += gz2[c] * shh[j];
            }
            for (int j = tid; j < a.Wd; j += 256) {
                float acc = 0.0f;
                for (int c = 0; c < C; ++c) acc = fmaf(gz2[c], a.w2[(long)c * a.Wd + j], acc);
                acc = shh[j] > 0.0f ? acc : 0.0f;
                gz1[j] = acc;
                a.gb1[j] += acc;
            }
            __syncthreads();
            for (int e = tid; e < a.Wd * C; e += 256) {         // gw1[j][c] += gz1[j] * pooled[c]
                const int j = e / C, c = e - j * C;
                a.gw1[e] += gz1[j] * a.pooled[(long)n * C + c];
            }
            for (int c = tid; c < C; c += 256) {
                float gp = 0.0f;
                for (int j = 0; j < a.Wd; ++j) gp = fmaf(gz1[j], a.w1[(long)j * C + c], gp);
                const long o = (long)n * C + c;
                const float sm = (float)(a.s[o] / a.pool_count);
                a.tA[o] += gp * sm;
                a.tB[o] += gp;
                if (a.gs) a.gs[o] = (double)(gp * a.A0[o]) / a.pool_count;     // direct path s -> pooled
            }
            __syncthreads();
        }
    }
    const float* dA = a.Wd > 0 ? a.tA : a.gA;
    const float* dB = a.Wd > 0 ? a.tB : a.gB;
    // (1') batch-norm adjoint per channel
    for (int c = tid; c < C; c += 256) {
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
                    a.gs[o] = (a.Wd > 0 ? a.gs[o] : 0.0) + g_mean / cnt;
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
                               double pool_count, float* A, float* B, double* mean, double* rstd, float* A0, float* B0,
                               float* gate, float* hbuf, float* pooled, void* stream) {
    CFN_REQUIRE(A && B && mean && rstd && run_mean && run_var, "cfn_bn_fold_fwd: null tensor");
    CFN_REQUIRE(!training || (s && q), "cfn_bn_fold_fwd: training needs sum / sumsq");
    CFN_REQUIRE(N > 0 && C > 0 && S > 0 && N % S == 0, "cfn_bn_fold_fwd: batch %d not divisible by %d splits", N, S);
    CFN_REQUIRE(Wd <= 0 || (w1 && b1 && w2 && b2 && s && A0 && B0 && gate && hbuf && pooled), "cfn_bn_fold_fwd: SE needs its tensors");
    BnFoldArgs a = {s, q, gamma, beta, run_mean, run_var, nbt, training, N, C, S, Wd, count, pool_count, eps, momentum,
                    w1, b1, w2, b2, A, B, mean, rstd, A0, B0, gate, hbuf, pooled};
    const size_t lds = (size_t)(C + (Wd > 0 ? Wd : 0)) * sizeof(float);
    hipLaunchKernelGGL(bn_fold_fwd_kernel, dim3(1), dim3(256), lds, (hipStream_t)stream, a);
    return cfn_check_launch("bn_fold_fwd");
}

extern "C" int cfn_bn_fold_bwd(const float* gA, const float* gB, const double* s, const float* gamma, const double* mean,
                               const double* rstd, const float* A0, const float* B0, const float* gate, const float* hbuf,
                               const float* pooled, const float* w1, const float* w2, int training, int N, int C, int S,
                               int Wd, double count, double pool_count, double* gs, double* gq, float* ggamma, float* gbeta,
                               float* gw1, float* gb1, float* gw2, float* gb2, float* tA, float* tB, void* stream) {
    CFN_REQUIRE(gA && gB && mean && rstd, "cfn_bn_fold_bwd: null tensor");
    CFN_REQUIRE(Wd <= 0 || (s && A0 && B0 && gate && hbuf && pooled && w1 && w2 && gw1 && gb1 && gw2 && gb2 && tA && tB),
                "cfn_bn_fold_bwd: SE needs its tensors");
    CFN_REQUIRE((gs == nullptr) == (gq == nullptr), "cfn_bn_fold_bwd: gs/gq mismatch");
    BnFoldBwdArgs a = {gA, gB, s, gamma, mean, rstd, A0, B0, gate, hbuf, pooled, w1, w2, training, N, C, S, Wd, count,
                       pool_count, gs, gq, ggamma, gbeta, gw1, gb1, gw2, gb2, tA, tB};
    const size_t lds = (size_t)(C + 2 * (Wd > 0 ? Wd : 0)) * sizeof(float);
    hipLaunchKernelGGL(bn_fold_bwd_kernel, dim3(1), dim3(256), lds, (hipStream_t)stream, a);
    return cfn_check_launch("bn_fold_bwd");
}
