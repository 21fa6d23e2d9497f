// Grid Pool / Grid Unpool temporal resampler (x3d_coarse.py:355-451) and Interp1d (interp1d.py:8-147).
//
// The reference resamples with a 5-D F.grid_sample whose h/w coordinates are the pixel centres, so
// the op is a 2-tap linear interpolation along t at  i_t = ((2(cdf-0.5)+1)/2)*(T-1)  (ATen
// grid_sampler_unnormalize, align_corners=True, zeros padding).  The integer frame index
// i0 = floor(i_t) is reproduced bit-exactly: same fp32 operation order, FMA contraction disabled
// in the index helpers.  Kernels are pure streaming (two input planes -> one output plane).
#include "cfn_common.h"

typedef float __attribute__((ext_vector_type(4))) f4v;

__device__ __forceinline__ void grid_time_coord(float cdf, int Tin, int& i0, float& w0, float& w1) {
#pragma clang fp contract(off)
    const float c = (cdf - 0.5f) * 2.0f;                       // x3d_coarse.py:394
    const float it = ((c + 1.0f) / 2.0f) * (float)(Tin - 1);   // ATen unnormalize, align_corners
    const float fl = floorf(it);
    i0 = (int)fl;
    w1 = it - fl;
    w0 = (fl + 1.0f) - it;
}

__global__ void grid_time_index_kernel(const float* __restrict__ cdf, int n, int Tin, int* __restrict__ i0,
                                       float* __restrict__ w1) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int k; float a, b;
    grid_time_coord(cdf[i], Tin, k, a, b);
    i0[i] = k;
    if (w1) w1[i] = b;
}

// out[b,c,k,p] = w0*x[b,c,i0,p] + w1*x[b,c,i0+1,p]   (frames outside [0,Tin) read as zero)
template <int VEC>
__global__ __launch_bounds__(256) void time_sample_fwd_kernel(const float* __restrict__ x, const float* __restrict__ cdf,
                                                              float* __restrict__ out, int C, int Tin, int K, long P) {
    const long bc = blockIdx.z;
    const int k = blockIdx.y, b = (int)(bc / C);
    int i0; float w0, w1;
    grid_time_coord(cdf[(long)b * K + k], Tin, i0, w0, w1);
    const bool ok0 = i0 >= 0 && i0 < Tin, ok1 = i0 + 1 >= 0 && i0 + 1 < Tin;
    const long p = ((long)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (p >= P) return;
    const float* x0 = x + (bc * Tin + i0) * P + p;
    float* o = out + (bc * K + k) * P + p;
    if (VEC == 4) {
        f4v a = {0.f, 0.f, 0.f, 0.f}, c = {0.f, 0.f, 0.f, 0.f};
        if (ok0) a = *reinterpret_cast<const f4v*>(x0);
        if (ok1) c = *reinterpret_cast<const f4v*>(x0 + P);
        *reinterpret_cast<f4v*>(o) = a * w0 + c * w1;
    } else {
        const float a = ok0 ? x0[0] : 0.f, c = ok1 ? x0[P] : 0.f;
        o[0] = a * w0 + c * w1;
    }
}

// gx[b,c,t,p] = sum_k [i0(k)==t] w0(k) g[k] + [i0(k)+1==t] w1(k) g[k]   (gather, deterministic)
// The taps that land on frame t are found ONCE per workgroup (wave 0: one k per lane, ordered compaction by ballot + prefix count into LDS:
// the order of the sum is the order of k, whatever the launch) instead of by every thread for every k (65 coordinate computations per
// float4 of output: 830 us at 8 x 24 x 256 x 56 x 56); the threads then add the 0-3 listed taps.
template <int VEC>
__global__ __launch_bounds__(256) void time_sample_bwd_x_kernel(const float* __restrict__ g, const float* __restrict__ cdf,
                                                                float* __restrict__ gx, int C, int Tin, int K, long P) {
    extern __shared__ float tl[];                      // [K] weights | [K] k indices (as int)
    __shared__ int ntap;
    int* tk = reinterpret_cast<int*>(tl + K);
    const long bc = blockIdx.z;
    const int t = blockIdx.y, b = (int)(bc / C);
    if (threadIdx.x < 64) {
        int base = 0;
        for (int k0 = 0; k0 < K; k0 += 64) {
            const int k = k0 + threadIdx.x;
            float wsel = 0.f;
            bool hit = false;
            if (k < K) {
                int i0; float w0, w1;
                grid_time_coord(cdf[(long)b * K + k], Tin, i0, w0, w1);
                if (i0 == t) { wsel = w0; hit = true; } else if (i0 + 1 == t) { wsel = w1; hit = true; }
            }
            const unsigned long long m = __ballot(hit);
            if (hit) {
                const int pos = base + __popcll(m & ((1ull << threadIdx.x) - 1ull));
                tl[pos] = wsel; tk[pos] = k;
            }
            base += __popcll(m);
        }
        if (threadIdx.x == 0) ntap = base;
    }
    __syncthreads();
    const long p = ((long)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (p >= P) return;
    f4v acc = {0.f, 0.f, 0.f, 0.f};
    const int n = ntap;
    for (int j = 0; j < n; ++j) {
        const float wsel = tl[j];
        const float* gp = g + (bc * K + tk[j]) * P + p;
        if (VEC == 4) acc += *reinterpret_cast<const f4v*>(gp) * wsel;
        else acc.x = fmaf(gp[0], wsel, acc.x);
    }
    float* o = gx + (bc * Tin + t) * P + p;
    if (VEC == 4) *reinterpret_cast<f4v*>(o) = acc; else o[0] = acc.x;
}

// gcdf[b,k] += (Tin-1) * sum_{c,p} g[b,c,k,p] * (x[i0+1] - x[i0])      (d i_t / d cdf = Tin-1)
__global__ __launch_bounds__(256) void time_sample_bwd_cdf_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                                  const float* __restrict__ cdf, double* __restrict__ gcdf,
                                                                  int C, int Tin, int K, long P, int cchunk) {
    __shared__ float sh[4];
    const int b = blockIdx.z, k = blockIdx.y;
    int i0; float w0, w1;
    grid_time_coord(cdf[(long)b * K + k], Tin, i0, w0, w1);
    const bool ok0 = i0 >= 0 && i0 < Tin, ok1 = i0 + 1 >= 0 && i0 + 1 < Tin;
    const int c_lo = blockIdx.x * cchunk, c_hi = min(c_lo + cchunk, C);
    float acc = 0.f;
    for (int c = c_lo; c < c_hi; ++c) {
        const long bc = (long)b * C + c;
        const float* gp = g + (bc * K + k) * P;
        const float* xp = x + (bc * Tin + i0) * P;
        for (long p = threadIdx.x; p < P; p += 256) {
            const float d = (ok1 ? xp[P + p] : 0.f) - (ok0 ? xp[p] : 0.f);
            acc = fmaf(gp[p], d, acc);
        }
    }
    acc = cfn_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) cfn_add64(&gcdf[(long)b * K + k], (double)(sh[0] + sh[1] + sh[2] + sh[3]) * (double)(Tin - 1));
}

// ---- Interp1d ----------------------------------------------------------------------------------
// one thread per query: ind = clamp(#(x < q) - 1, 0, N-2)   (searchsorted 'left' - 1, interp1d.py:100-110)
__global__ void interp1d_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ q,
                                    float* __restrict__ ynew, long* __restrict__ ind, int B, int N, int Pq, int xrow,
                                    int yrow, int qrow) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * Pq) return;
    const int b = i / Pq, j = i - b * Pq;
    const float* xr = x + (long)(xrow ? b : 0) * N;
    const float* yr = y + (long)(yrow ? b : 0) * N;
    const float qv = q[(long)(qrow ? b : 0) * Pq + j];
    int lo = 0, hi = N;                     // first index with x[idx] >= qv
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (xr[mid] < qv) lo = mid + 1; else hi = mid; }
    int id = lo - 1;
    id = id < 0 ? 0 : (id > N - 2 ? N - 2 : id);
    const float eps = 1.1920928955078125e-07f;   // torch.finfo(float32).eps, interp1d.py:37
    const float slope = (yr[id + 1] - yr[id]) / (eps + (xr[id + 1] - xr[id]));
    ynew[i] = yr[id] + slope * (qv - xr[id]);
    if (ind) ind[i] = id;
}

// gradients of ynew = y0 + (y1-y0)/(eps+x1-x0) * (q-x0) with ind constant.  Gather form: one thread per
// OUTPUT element walks the queries in index order, so every sum has a fixed order (a scatter with fp32 atomics made the
// Grid Pool gradients differ from run to run); the tensors are (B,N)/(B,Pq)-sized (tiny).  A broadcast x / y row (xrow /
// yrow == 0) collects the queries of all B rows.
__global__ void interp1d_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ y,
                                    const float* __restrict__ q, const long* __restrict__ ind, float* __restrict__ gx,
                                    float* __restrict__ gy, float* __restrict__ gq, int B, int N, int Pq, int xrow, int yrow,
                                    int qrow) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    auto terms = [&](int b, int j, float& gv, float& w1, float& t, float& slope) -> int {
        const long xo = (long)(xrow ? b : 0) * N, yo = (long)(yrow ? b : 0) * N;
        const int id = (int)ind[(long)b * Pq + j];
        const float x0 = x[xo + id], x1 = x[xo + id + 1], y0 = y[yo + id], y1 = y[yo + id + 1];
        const float qv = q[(long)(qrow ? b : 0) * Pq + j];
        const float den = 1.1920928955078125e-07f + (x1 - x0);
        const float dy = y1 - y0, dq = qv - x0;
        slope = dy / den;
        gv = g[(long)b * Pq + j];
        w1 = dq / den;
        t = dy * dq / (den * den);
        return id;
    };
    // knot gradients: element (row r, knot k) <- queries whose bin is k (left knot) or k-1 (right knot)
    if (i < B * N) {
        const int r = i / N, k = i - r * N;
        if (gy && (yrow || r == 0)) {
            float acc = 0.0f;
            for (int b = yrow ? r : 0; b < (yrow ? r + 1 : B); ++b)
                for (int j = 0; j < Pq; ++j) {
                    float gv, w1, t, sl;
                    const int id = terms(b, j, gv, w1, t, sl);
                    if (id == k) acc += gv * (1.0f - w1);
                    else if (id + 1 == k) acc += gv * w1;
                }
            gy[(long)r * N + k] = acc;
        }
        if (gx && (xrow || r == 0)) {
            float acc = 0.0f;
            for (int b = xrow ? r : 0; b < (xrow ? r + 1 : B); ++b)
                for (int j = 0; j < Pq; ++j) {
                    float gv, w1, t, sl;
                    const int id = terms(b, j, gv, w1, t, sl);
                    if (id == k) acc += gv * (t - sl);
                    else if (id + 1 == k) acc += gv * (-t);
                }
            gx[(long)r * N + k] = acc;
        }
    }
    // query gradients: one term per (b, j); a broadcast query row sums over b in order
    if (gq && i < (qrow ? B : 1) * Pq) {
        const int r = i / Pq, j = i - r * Pq;
        float acc = 0.0f;
        for (int b = qrow ? r : 0; b < (qrow ? r + 1 : B); ++b) {
            float gv, w1, t, sl;
            terms(b, j, gv, w1, t, sl);
            acc += gv * sl;
        }
        gq[(long)r * Pq + j] = acc;
    }
}

// ---- temporal linear resize: align_corners=True (F.interpolate 'linear' x3d_coarse.py:725 and the t-axis of
// 'trilinear' :449 when h,w keep their size) or half-pixel centres (align_corners=False: the loss upsampling of
// train_coarse_fineFEAT.py:226) -- ATen's area_pixel_compute_scale / _source_index in the same operation order
__device__ __forceinline__ void resize_src(int j, int Kin, int Lout, int ac, int& i0, int& i1, float& l0, float& l1) {
#pragma clang fp contract(off)
    float src;
    if (ac) {
        const float scale = Lout > 1 ? (float)(Kin - 1) / (float)(Lout - 1) : 0.0f;
        src = scale * (float)j;
    } else {
        const float scale = (float)Kin / (float)Lout;
        src = fmaf(scale, (float)j + 0.5f, -0.5f);   // ATen's CPU build contracts this expression into one FMA (checked
        if (src < 0.0f) src = 0.0f;                  // against F.interpolate: 2e-7 with, 1e-5 without)
    }
    i0 = (int)src;
    i1 = i0 + (i0 < Kin - 1 ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.0f - l1;
}

// one thread per output element (bc, j, p), p fastest
__global__ __launch_bounds__(256) void time_resize_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int Kin,
                                                              int Lout, long P, long total, int ac) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const long p = e % P, r = e / P;
    const int j = (int)(r % Lout);
    const long bc = r / Lout;
    int i0, i1; float l0, l1;
    resize_src(j, Kin, Lout, ac, i0, i1, l0, l1);
    const float* xp = x + bc * Kin * P + p;
    out[e] = l0 * xp[(long)i0 * P] + l1 * xp[(long)i1 * P];
}

// one thread per input element (bc, k, p): gathers the few outputs j whose source interval touches k
// (src(j) = j*(Kin-1)/(Lout-1) in [k-1, k+1]  =>  j in [(k-1)/scale, (k+1)/scale], checked exactly)
__global__ __launch_bounds__(256) void time_resize_bwd_kernel(const float* __restrict__ g, float* __restrict__ gx, int Kin,
                                                              int Lout, long P, long total, int ac) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const long p = e % P, r = e / P;
    const int k = (int)(r % Kin);
    const long bc = r / Kin;
    int jlo = 0, jhi = Lout - 1;
    if (!ac) {              // src(j) = (j + 0.5) * Kin / Lout - 0.5 (clamped at 0) within [k-1, k+1]
        const double inv = (double)Lout / (double)Kin;
        jlo = k == 0 ? 0 : (int)floor((k - 0.5) * inv - 0.5) - 1;
        jhi = (int)ceil((k + 1.5) * inv - 0.5) + 1;
        if (jlo < 0) jlo = 0;
        if (jhi > Lout - 1) jhi = Lout - 1;
    } else if (Kin > 1 && Lout > 1) {
        const double inv = (double)(Lout - 1) / (double)(Kin - 1);
        jlo = (int)floor((k - 1) * inv) - 1;
        jhi = (int)ceil((k + 1) * inv) + 1;
        if (jlo < 0) jlo = 0;
        if (jhi > Lout - 1) jhi = Lout - 1;
    }
    float acc = 0.f;
    for (int j = jlo; j <= jhi; ++j) {
        int i0, i1; float l0, l1;
        resize_src(j, Kin, Lout, ac, i0, i1, l0, l1);
        if (i0 != k && i1 != k) continue;
        const float gv = g[(bc * Lout + j) * P + p];
        if (i0 == k) acc = fmaf(gv, l0, acc);
        if (i1 == k) acc = fmaf(gv, l1, acc);
    }
    gx[e] = acc;
}

// ---------------------------------------------------------------------------------------------
extern "C" int cfn_grid_time_index(const float* cdf, int n, int Tin, int* i0, float* w1, void* stream) {
    CFN_REQUIRE(cdf && i0 && n > 0 && Tin > 0, "cfn_grid_time_index: bad argument");
    hipLaunchKernelGGL(grid_time_index_kernel, dim3(cfn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, cdf, n, Tin, i0, w1);
    return cfn_check_launch("grid_time_index");
}

#define CFN_GRID_CHECK(BC, Y) CFN_REQUIRE((BC) > 0 && (BC) <= 65535 && (Y) > 0 && (Y) <= 65535, "grid dims exceed 65535 (%ld, %ld)", (long)(BC), (long)(Y))

extern "C" int cfn_time_sample_fwd(const float* x, const float* cdf, float* out, int B, int C, int Tin, int K, long P,
                                   void* stream) {
    CFN_REQUIRE(x && cdf && out, "cfn_time_sample_fwd: null tensor");
    CFN_GRID_CHECK((long)B * C, K);
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_GRIDPOOL, st, 4.0 * B * C * K * P * 3);
    if (P % 4 == 0) hipLaunchKernelGGL(time_sample_fwd_kernel<4>, dim3(cfn_cdiv(P, 1024), K, B * C), dim3(256), 0, st, x, cdf, out, C, Tin, K, P);
    else hipLaunchKernelGGL(time_sample_fwd_kernel<1>, dim3(cfn_cdiv(P, 256), K, B * C), dim3(256), 0, st, x, cdf, out, C, Tin, K, P);
    return cfn_check_launch("time_sample_fwd");
}

extern "C" int cfn_time_sample_bwd(const float* g, const float* x, const float* cdf, float* gx, double* gcdf, int B, int C,
                                   int Tin, int K, long P, void* stream) {
    CFN_REQUIRE(g && cdf, "cfn_time_sample_bwd: null tensor");
    CFN_GRID_CHECK((long)B * C, Tin > K ? Tin : K);
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_GRIDPOOL_BWD, st, 4.0 * B * C * P * ((double)K * 2 + Tin));
    if (gx) {
        if (P % 4 == 0) hipLaunchKernelGGL(time_sample_bwd_x_kernel<4>, dim3(cfn_cdiv(P, 1024), Tin, B * C), dim3(256), 2 * K * sizeof(float), st, g, cdf, gx, C, Tin, K, P);
        else hipLaunchKernelGGL(time_sample_bwd_x_kernel<1>, dim3(cfn_cdiv(P, 256), Tin, B * C), dim3(256), 2 * K * sizeof(float), st, g, cdf, gx, C, Tin, K, P);
    }
    if (gcdf) {
        CFN_REQUIRE(x != nullptr, "cfn_time_sample_bwd: gcdf needs x");
        int cchunk = 1;
        while ((long)cchunk * P < 16384 && cchunk < C) cchunk *= 2;
        hipLaunchKernelGGL(time_sample_bwd_cdf_kernel, dim3(cfn_cdiv(C, cchunk), K, B), dim3(256), 0, st, g, x, cdf, gcdf, C, Tin, K, P, cchunk);
    }
    return cfn_check_launch("time_sample_bwd");
}

extern "C" int cfn_interp1d_fwd(const float* x, const float* y, const float* xnew, float* ynew, long* ind, int B, int N,
                                int Pq, int xrow, int yrow, int qrow, void* stream) {
    CFN_REQUIRE(x && y && xnew && ynew, "cfn_interp1d_fwd: null tensor");
    CFN_REQUIRE(N >= 2, "cfn_interp1d_fwd: need at least 2 knots (got %d)", N);
    hipLaunchKernelGGL(interp1d_fwd_kernel, dim3(cfn_cdiv((long)B * Pq, 256)), dim3(256), 0, (hipStream_t)stream, x, y, xnew, ynew, ind, B, N, Pq, xrow, yrow, qrow);
    return cfn_check_launch("interp1d_fwd");
}

// gx (B or 1, N), gy (B or 1, N), gq (B or 1, Pq) are fully overwritten; any of them may be null
extern "C" int cfn_interp1d_bwd(const float* g, const float* x, const float* y, const float* xnew, const long* ind, float* gx,
                                float* gy, float* gq, int B, int N, int Pq, int xrow, int yrow, int qrow, void* stream) {
    CFN_REQUIRE(g && x && y && xnew && ind, "cfn_interp1d_bwd: null tensor");
    const long work = (long)B * (N > Pq ? N : Pq);
    hipLaunchKernelGGL(interp1d_bwd_kernel, dim3(cfn_cdiv(work, 256)), dim3(256), 0, (hipStream_t)stream, g, x, y, xnew, ind, gx, gy, gq, B, N, Pq, xrow, yrow, qrow);
    return cfn_check_launch("interp1d_bwd");
}

extern "C" int cfn_time_resize_fwd(const float* x, float* out, long BC, int Kin, int Lout, long P, int align_corners,
                                   void* stream) {
    CFN_REQUIRE(x && out && Kin > 0 && Lout > 0 && BC > 0 && P > 0, "cfn_time_resize_fwd: bad argument");
    const long total = BC * Lout * P;
    hipLaunchKernelGGL(time_resize_fwd_kernel, dim3(cfn_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, out, Kin, Lout, P, total, align_corners);
    return cfn_check_launch("time_resize_fwd");
}

extern "C" int cfn_time_resize_bwd(const float* g, float* gx, long BC, int Kin, int Lout, long P, int align_corners,
                                   void* stream) {
    CFN_REQUIRE(g && gx && Kin > 0 && Lout > 0 && BC > 0 && P > 0, "cfn_time_resize_bwd: bad argument");
    const long total = BC * Kin * P;
    hipLaunchKernelGGL(time_resize_bwd_kernel, dim3(cfn_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, g, gx, Kin, Lout, P, total, align_corners);
    return cfn_check_launch("time_resize_bwd");
}

// ---------------------------------------------------------------------------------------------------------
// saliency logits -> CDF knots (GridPoolLayer.forward x3d_coarse.py:384-392), one thread per row:
//   q = 1 - sigmoid(0.5 (g + bias));  p = q / (sum q + 1e-16);  cdf = [0, cumsum(p)]
// fp32 element arithmetic in the reference's operation order (contraction off), cumsum accumulated in fp64 and rounded
// per knot like ATen's CPU cumsum; the row sum is a plain left-to-right fp32 sum (ATen's vectorised CPU sum order depends
// on the host's vector ISA and cannot be a contract).  Backward = autograd of the same expression.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float cdf_q(float g) {
#pragma clang fp contract(off)
    return 1.0f - 1.0f / (1.0f + expf(-(g * 0.5f)));
}

__global__ void grid_cdf_fwd_kernel(const float* __restrict__ g, const float* __restrict__ bias, float* __restrict__ cdf, int B, int Kin) {
#pragma clang fp contract(off)
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const float bs = bias ? bias[0] : 0.0f;
    const float* gr = g + (long)b * Kin;
    float S = 0.0f;
    for (int i = 0; i < Kin; ++i) S += cdf_q(gr[i] + bs);
    const float dn = S + 1e-16f;
    double c = 0.0;
    float* o = cdf + (long)b * (Kin + 1);
    o[0] = 0.0f;
    for (int i = 0; i < Kin; ++i) {
        c += (double)(cdf_q(gr[i] + bs) / dn);
        o[i + 1] = (float)c;
    }
}

__global__ void grid_cdf_bwd_kernel(const float* __restrict__ gc, const float* __restrict__ g, const float* __restrict__ bias,
                                    float* __restrict__ gg, int B, int Kin) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const float bs = bias ? bias[0] : 0.0f;
    const float* gr = g + (long)b * Kin;
    const float* gcr = gc + (long)b * (Kin + 1);
    float S = 0.0f;
    for (int i = 0; i < Kin; ++i) S += cdf_q(gr[i] + bs);
    const float dn = S + 1e-16f;
    // gp_i = sum_{k > i} gc_k (suffix sums);  dot = sum_j gp_j q_j
    double suf = 0.0, dot = 0.0;
    for (int i = Kin - 1; i >= 0; --i) {
        suf += (double)gcr[i + 1];
        dot += suf * (double)cdf_q(gr[i] + bs);
    }
    suf = 0.0;
    const double idn = 1.0 / (double)dn;
    for (int i = Kin - 1; i >= 0; --i) {
        suf += (double)gcr[i + 1];
        const float q = cdf_q(gr[i] + bs), s = 1.0f - q;
        const double gq = suf * idn - dot * idn * idn;
        gg[(long)b * Kin + i] = (float)(gq * (double)(-0.5f * s * (1.0f - s)));
    }
}

extern "C" int cfn_grid_cdf_fwd(const float* g, const float* bias, float* cdf, int B, int Kin, void* stream) {
    CFN_REQUIRE(g && cdf && B > 0 && Kin > 0, "cfn_grid_cdf_fwd: bad arguments");
    hipLaunchKernelGGL(grid_cdf_fwd_kernel, dim3(cfn_cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, g, bias, cdf, B, Kin);
    return cfn_check_launch("grid_cdf_fwd");
}

extern "C" int cfn_grid_cdf_bwd(const float* gcdf, const float* g, const float* bias, float* gg, int B, int Kin, void* stream) {
    CFN_REQUIRE(gcdf && g && gg && B > 0 && Kin > 0, "cfn_grid_cdf_bwd: bad arguments");
    hipLaunchKernelGGL(grid_cdf_bwd_kernel, dim3(cfn_cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, gcdf, g, bias, gg, B, Kin);
    return cfn_check_launch("grid_cdf_bwd");
}

// ---------------------------------------------------------------------------------------------------------
// Fixed temporal pooling of the coarse stream, t_pool = 'avg' | 'max' (nn.AvgPool3d / nn.MaxPool3d((4,1,1), stride (4,1,1)),
// x3d_coarse.py:489-492, applied at :640-643): out[bc,k,p] = mean / max of x[bc, R k .. R k + R - 1, p], K = floor(T / R).
// bwd: avg spreads g / R; max routes g to the first maximal frame (ATen's tie rule); frames beyond R*K get zero.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void time_pool_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int mode, int Tin, int K,
                                                            int R, long P) {
    const long bc = blockIdx.z;
    const int k = blockIdx.y;
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const float* xp = x + (bc * Tin + (long)k * R) * P + p;
    float v = xp[0];
    for (int i = 1; i < R; ++i) v = mode == 0 ? v + xp[(long)i * P] : fmaxf(v, xp[(long)i * P]);
    out[(bc * K + k) * P + p] = mode == 0 ? v / (float)R : v;
}

__global__ __launch_bounds__(256) void time_pool_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x, float* __restrict__ gx,
                                                            int mode, int Tin, int K, int R, long P) {
    const long bc = blockIdx.z;
    const int t = blockIdx.y;
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const int k = t / R;
    float r = 0.0f;
    if (k < K) {
        const float gv = g[(bc * K + k) * P + p];
        if (mode == 0) r = gv / (float)R;
        else {
            const float* xp = x + (bc * Tin + (long)k * R) * P + p;
            int arg = 0;
            float m = xp[0];
            for (int i = 1; i < R; ++i) { const float v = xp[(long)i * P]; if (v > m) { m = v; arg = i; } }
            r = (t - k * R) == arg ? gv : 0.0f;
        }
    }
    gx[(bc * Tin + t) * P + p] = r;
}

extern "C" int cfn_time_pool_fwd(const float* x, float* out, int mode, long BC, int Tin, int R, long P, void* stream) {
    CFN_REQUIRE(x && out && (mode == 0 || mode == 1) && R >= 1 && Tin >= R, "cfn_time_pool_fwd: bad arguments");
    const int K = Tin / R;
    CFN_GRID_CHECK(BC, K);
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_GRIDPOOL, st, 4.0 * BC * P * ((double)K * R + K));
    hipLaunchKernelGGL(time_pool_fwd_kernel, dim3(cfn_cdiv(P, 256), K, (unsigned)BC), dim3(256), 0, st, x, out, mode, Tin, K, R, P);
    return cfn_check_launch("time_pool_fwd");
}

extern "C" int cfn_time_pool_bwd(const float* g, const float* x, float* gx, int mode, long BC, int Tin, int R, long P, void* stream) {
    CFN_REQUIRE(g && gx && (mode == 0 || (mode == 1 && x)) && R >= 1 && Tin >= R, "cfn_time_pool_bwd: bad arguments");
    const int K = Tin / R;
    CFN_GRID_CHECK(BC, Tin);
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_GRIDPOOL_BWD, st, 4.0 * BC * P * ((double)Tin * (mode ? 2 : 1) + K));
    hipLaunchKernelGGL(time_pool_bwd_kernel, dim3(cfn_cdiv(P, 256), Tin, (unsigned)BC), dim3(256), 0, st, g, x, gx, mode, Tin, K, R, P);
    return cfn_check_launch("time_pool_bwd");
}
