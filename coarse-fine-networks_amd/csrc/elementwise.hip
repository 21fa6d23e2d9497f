// Streaming per-(n,c) kernels around the convolutions: everything SubBatchNorm3d / ReLU / Swish /
// residual add / spatial average pooling does in the reference as separate eager passes
// (x3d_fine.py:51-62, :74-86, :151, :164, :172-173, :345-366) is folded here into single passes
// with a per-(n,c) affine  z = A[n,c]*x + B[n,c]  (A,B carry batch-norm scale/shift, the BN affine
// and the squeeze-excite gate) and fp64 per-(n,c) reduction outputs for the backward of that affine.
// All kernels: grid = (chunks of the (n,c) volume, N*C), float4 when the volume allows it.
#include "cfn_common.h"
#include <stdlib.h>

typedef float __attribute__((ext_vector_type(4))) f4v;

// block-wide sum of up to 4 values -> thread 0 ; 256 threads
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* sh /* [NV*4] */) {
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

#define EW_ITEMS 8   // float4 (or scalars) per thread

// ---- out = relu( A*y + B + (Ar*res + Br) ) --------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void bn_add_relu_fwd_kernel(const float* __restrict__ y, const double* __restrict__ A,
                                                              const double* __restrict__ B, const float* __restrict__ res,
                                                              const double* __restrict__ Ar, const double* __restrict__ Br,
                                                              float* __restrict__ out, unsigned* __restrict__ mask, long vol) {
    const long nc = blockIdx.y + (long)blockIdx.z * gridDim.y;      // (y, z) = exact factorisation of N*C (cfn_split_nc)
    const float a = A[nc], b = B[nc] + (Br ? Br[nc] : 0.0f), ar = Ar ? Ar[nc] : 1.0f;
    const long base = nc * vol;
    long i = ((long)blockIdx.x * 256 * EW_ITEMS + threadIdx.x) * VEC;
    unsigned mw = 0;   // bit 4k+e: element e of this thread's k-th float4 is positive (VEC 4 only)
#pragma unroll
    for (int k = 0; k < EW_ITEMS; ++k, i += 256 * VEC) {
        if (i >= vol) break;
        if (VEC == 4) {
            const f4v yv = *reinterpret_cast<const f4v*>(y + base + i);
            const f4v rv = *reinterpret_cast<const f4v*>(res + base + i);
            f4v o;
            o.x = fmaxf(fmaf(yv.x, a, fmaf(rv.x, ar, b)), 0.f); o.y = fmaxf(fmaf(yv.y, a, fmaf(rv.y, ar, b)), 0.f);
            o.z = fmaxf(fmaf(yv.z, a, fmaf(rv.z, ar, b)), 0.f); o.w = fmaxf(fmaf(yv.w, a, fmaf(rv.w, ar, b)), 0.f);
            *reinterpret_cast<f4v*>(out + base + i) = o;
            mw |= ((o.x > 0.f ? 1u : 0u) | (o.y > 0.f ? 2u : 0u) | (o.z > 0.f ? 4u : 0u) | (o.w > 0.f ? 8u : 0u)) << (4 * k);
        } else {
            out[base + i] = fmaxf(fmaf(y[base + i], a, fmaf(res[base + i], ar, b)), 0.f);
        }
    }
    if (VEC == 4 && mask) mask[(nc * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = mw;
}

// backward of the tail WITHOUT the per-channel scales:  g = (gout [+ gout2]) * (out > 0)  is written once and serves as
// both d/dy (consumer applies A: cfn_pwconv_bwd_* `gscale`) and d/dres (identity shortcut: exact; conv shortcut: `gscale`
// = Ar).  The ReLU mask comes from the forward's bit mask (1/32 of a tensor pass) or, without it, from `out`.
//   gA += sum g*y;  gB += sum g;  gAr += sum g*res (when gAr != null)
template <int VEC>
__global__ __launch_bounds__(256) void bn_add_relu_bwd_g_kernel(const float* __restrict__ gout, const float* __restrict__ gout2,
                                                                const float* __restrict__ out, const unsigned* __restrict__ mask,
                                                                const float* __restrict__ y, const float* __restrict__ res,
                                                                float* __restrict__ g_out, double* __restrict__ gA,
                                                                double* __restrict__ gB, double* __restrict__ gAr, long vol) {
    __shared__ float sh[12];
    const long nc = blockIdx.y + (long)blockIdx.z * gridDim.y;      // (y, z) = exact factorisation of N*C (cfn_split_nc)
    const long base = nc * vol;
    float acc[3] = {0.f, 0.f, 0.f};
    long i = ((long)blockIdx.x * 256 * EW_ITEMS + threadIdx.x) * VEC;
    const unsigned mw = (VEC == 4 && mask) ? mask[(nc * gridDim.x + blockIdx.x) * 256 + threadIdx.x] : 0u;
#pragma unroll
    for (int k = 0; k < EW_ITEMS; ++k, i += 256 * VEC) {
        if (i >= vol) break;
        if (VEC == 4) {
            f4v go = *reinterpret_cast<const f4v*>(gout + base + i);
            if (gout2) go += *reinterpret_cast<const f4v*>(gout2 + base + i);
            const f4v yv = *reinterpret_cast<const f4v*>(y + base + i);
            unsigned m4;
            if (mask) m4 = (mw >> (4 * k)) & 15u;
            else {
                const f4v ov = *reinterpret_cast<const f4v*>(out + base + i);
                m4 = (ov.x > 0.f ? 1u : 0u) | (ov.y > 0.f ? 2u : 0u) | (ov.z > 0.f ? 4u : 0u) | (ov.w > 0.f ? 8u : 0u);
            }
            f4v g;
            g.x = (m4 & 1u) ? go.x : 0.f; g.y = (m4 & 2u) ? go.y : 0.f;
            g.z = (m4 & 4u) ? go.z : 0.f; g.w = (m4 & 8u) ? go.w : 0.f;
            acc[0] += g.x * yv.x + g.y * yv.y + g.z * yv.z + g.w * yv.w;
            acc[1] += g.x + g.y + g.z + g.w;
            if (gAr) {
                const f4v rv = *reinterpret_cast<const f4v*>(res + base + i);
                acc[2] += g.x * rv.x + g.y * rv.y + g.z * rv.z + g.w * rv.w;
            }
            *reinterpret_cast<f4v*>(g_out + base + i) = g;
        } else {
            const float g = out[base + i] > 0.f ? gout[base + i] + (gout2 ? gout2[base + i] : 0.f) : 0.f;
            acc[0] = fmaf(g, y[base + i], acc[0]);
            acc[1] += g;
            if (gAr) acc[2] = fmaf(g, res[base + i], acc[2]);
            g_out[base + i] = g;
        }
    }
    block_sum<3>(acc, sh);
    if (threadIdx.x == 0) {
        cfn_add64(&gA[nc], (double)acc[0]);
        cfn_add64(&gB[nc], (double)acc[1]);
        if (gAr) cfn_add64(&gAr[nc], (double)acc[2]);
    }
}

// FLAT variants of the two kernels above (float4 tensors; round 4): the loops above are one load -> wait -> store round trip per iteration
// (hipcc keeps every iteration behind its `break`, and on gfx950 the wait for a load also waits for the store before it: 8 serial HBM round
// trips per thread, hidden only by occupancy: 5.3-5.5 TB/s).  Here a thread issues ALL its EW_ITEMS loads of every tensor up front (unconditional
// buffer loads, out-of-range offsets beyond the channel's volume), and the workgroups are dealt in memory order with each XCD walking one
// contiguous eighth (cfn_xcd_remap) -- the pattern of dwt5_fwd_flat_kernel.  Same mask words, same per-workgroup reduction.
struct EwFlatArgs {
    const float* y; const double* A; const double* B; const float* res; const double* Ar; const double* Br; float* out; unsigned* mask;
    const float* gout; const float* gout2; const float* outr; const unsigned* maskr; float* g; double* gA; double* gB; double* gAr;
    long vol; unsigned nchunks;
};

__global__ __launch_bounds__(256) void bn_add_relu_fwd_flat_kernel(const EwFlatArgs a) {
    constexpr int OOB = 0x7ffffff0;
    const unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    const long nc = cfn_uni((int)(L / a.nchunks));
    const unsigned chunk = cfn_uni(L - (unsigned)nc * a.nchunks);
    const float ca = a.A[nc], cb = a.B[nc] + (a.Br ? a.Br[nc] : 0.0f), car = a.Ar ? a.Ar[nc] : 1.0f;
    const unsigned bytes = (unsigned)(a.vol * 4);
    __amdgpu_buffer_rsrc_t ry = cfn_rsrc(a.y + nc * a.vol, bytes), rr = cfn_rsrc(a.res + nc * a.vol, bytes), ro = cfn_rsrc(a.out + nc * a.vol, bytes);
    const long i0 = ((long)chunk * 256 * EW_ITEMS + threadIdx.x) * 4;
    f4v yv[EW_ITEMS], rv[EW_ITEMS];
    int off[EW_ITEMS];
#pragma unroll
    for (int k = 0; k < EW_ITEMS; ++k) {
        const long i = i0 + (long)k * 1024;
        off[k] = i < a.vol ? (int)(i * 4) : OOB;
        yv[k] = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(ry, off[k], 0, 0));
        rv[k] = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(rr, off[k], 0, 0));
    }
    unsigned mw = 0;
#pragma unroll
    for (int k = 0; k < EW_ITEMS; ++k) {
        f4v o;
        o.x = fmaxf(fmaf(yv[k].x, ca, fmaf(rv[k].x, car, cb)), 0.f); o.y = fmaxf(fmaf(yv[k].y, ca, fmaf(rv[k].y, car, cb)), 0.f);
        o.z = fmaxf(fmaf(yv[k].z, ca, fmaf(rv[k].z, car, cb)), 0.f); o.w = fmaxf(fmaf(yv[k].w, ca, fmaf(rv[k].w, car, cb)), 0.f);
        cfn_bst128(__builtin_bit_cast(unsigned __attribute__((ext_vector_type(4))), o), ro, off[k], 0);
        if (off[k] != OOB) mw |= ((o.x > 0.f ? 1u : 0u) | (o.y > 0.f ? 2u : 0u) | (o.z > 0.f ? 4u : 0u) | (o.w > 0.f ? 8u : 0u)) << (4 * k);
    }
    if (a.mask) a.mask[((unsigned long)nc * a.nchunks + chunk) * 256 + threadIdx.x] = mw;
}

__global__ __launch_bounds__(256) void bn_add_relu_bwd_g_flat_kernel(const EwFlatArgs a) {
    constexpr int OOB = 0x7ffffff0;
    __shared__ float sh[12];
    const unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    const long nc = cfn_uni((int)(L / a.nchunks));
    const unsigned chunk = cfn_uni(L - (unsigned)nc * a.nchunks);
    const unsigned bytes = (unsigned)(a.vol * 4);
    const bool has2 = a.gout2 != nullptr, hasm = a.maskr != nullptr, hasr = a.gAr != nullptr;     // workgroup uniform
    // (an absent second gradient gets a descriptor of 0 bytes: its loads return zeros without touching memory)
    __amdgpu_buffer_rsrc_t rg = cfn_rsrc(a.gout + nc * a.vol, bytes), rg2 = cfn_rsrc((has2 ? a.gout2 : a.gout) + nc * a.vol, has2 ? bytes : 0u);
    __amdgpu_buffer_rsrc_t ry = cfn_rsrc(a.y + nc * a.vol, bytes), ro = cfn_rsrc((hasm ? a.gout : a.outr) + nc * a.vol, bytes);
    __amdgpu_buffer_rsrc_t rr = cfn_rsrc((hasr ? a.res : a.gout) + nc * a.vol, bytes), rd = cfn_rsrc(a.g + nc * a.vol, bytes);
    const long i0 = ((long)chunk * 256 * EW_ITEMS + threadIdx.x) * 4;
    const unsigned mw = hasm ? a.maskr[((unsigned long)nc * a.nchunks + chunk) * 256 + threadIdx.x] : 0u;
    f4v go[EW_ITEMS], g2v[EW_ITEMS], yv[EW_ITEMS];
    int off[EW_ITEMS];
#pragma unroll
    for (int k = 0; k < EW_ITEMS; ++k) {
        const long i = i0 + (long)k * 1024;
        off[k] = i < a.vol ? (int)(i * 4) : OOB;
        go[k] = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(rg, off[k], 0, 0));
        g2v[k] = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(rg2, off[k], 0, 0));
        yv[k] = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(ry, off[k], 0, 0));
    }
    float acc[3] = {0.f, 0.f, 0.f};
    // the rare tensors in a second round (registers): `out` when there is no bit mask, the residual for gAr (conv shortcut: first block of a layer)
#pragma unroll
    for (int k = 0; k < EW_ITEMS; ++k) {
        f4v ov, rv;
        const f4v g2 = g2v[k];
        if (!hasm) ov = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(ro, off[k], 0, 0));
        if (hasr) rv = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(rr, off[k], 0, 0));
        const f4v gs = go[k] + g2;
        unsigned m4;
        if (hasm) m4 = (mw >> (4 * k)) & 15u;
        else m4 = (ov.x > 0.f ? 1u : 0u) | (ov.y > 0.f ? 2u : 0u) | (ov.z > 0.f ? 4u : 0u) | (ov.w > 0.f ? 8u : 0u);
        f4v g;
        g.x = (m4 & 1u) ? gs.x : 0.f; g.y = (m4 & 2u) ? gs.y : 0.f;
        g.z = (m4 & 4u) ? gs.z : 0.f; g.w = (m4 & 8u) ? gs.w : 0.f;
        acc[0] += g.x * yv[k].x + g.y * yv[k].y + g.z * yv[k].z + g.w * yv[k].w;     // (out-of-range items: every operand is 0)
        acc[1] += g.x + g.y + g.z + g.w;
        if (hasr) acc[2] += g.x * rv.x + g.y * rv.y + g.z * rv.z + g.w * rv.w;
        cfn_bst128(__builtin_bit_cast(unsigned __attribute__((ext_vector_type(4))), g), rd, off[k], 0);
    }
    block_sum<3>(acc, sh);
    if (threadIdx.x == 0) {
        cfn_add64(&a.gA[nc], (double)acc[0]);
        cfn_add64(&a.gB[nc], (double)acc[1]);
        if (hasr) cfn_add64(&a.gAr[nc], (double)acc[2]);
    }
}

// g = (gout [+ gout2]) * (out > 0);  gy = g*A;  gres = g*Ar;  gA += sum g*y;  gB += sum g;  gAr += sum g*res
// gout2 (optional): the block output feeds two consumers (next conv1 and next residual); their two gradients are
// summed here on the fly instead of by a separate 3-pass add kernel
template <int VEC>
__global__ __launch_bounds__(256) void bn_add_relu_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ gout2,
                                                              const float* __restrict__ out,
                                                              const float* __restrict__ y, const double* __restrict__ A,
                                                              const float* __restrict__ res, const double* __restrict__ Ar,
                                                              float* __restrict__ gy, float* __restrict__ gres,
                                                              double* __restrict__ gA, double* __restrict__ gB,
                                                              double* __restrict__ gAr, long vol) {
    __shared__ float sh[12];
    const long nc = blockIdx.y + (long)blockIdx.z * gridDim.y;      // (y, z) = exact factorisation of N*C (cfn_split_nc)
    const float a = A[nc], ar = Ar ? Ar[nc] : 1.0f;
    const long base = nc * vol;
    float acc[3] = {0.f, 0.f, 0.f};
    long i = ((long)blockIdx.x * 256 * EW_ITEMS + threadIdx.x) * VEC;
#pragma unroll
    for (int k = 0; k < EW_ITEMS; ++k, i += 256 * VEC) {
        if (i >= vol) break;
        if (VEC == 4) {
            f4v go = *reinterpret_cast<const f4v*>(gout + base + i);
            if (gout2) go += *reinterpret_cast<const f4v*>(gout2 + base + i);
            const f4v ov = *reinterpret_cast<const f4v*>(out + base + i);
            const f4v yv = *reinterpret_cast<const f4v*>(y + base + i);
            f4v g;
            g.x = ov.x > 0.f ? go.x : 0.f; g.y = ov.y > 0.f ? go.y : 0.f;
            g.z = ov.z > 0.f ? go.z : 0.f; g.w = ov.w > 0.f ? go.w : 0.f;
            acc[0] += g.x * yv.x + g.y * yv.y + g.z * yv.z + g.w * yv.w;
            acc[1] += g.x + g.y + g.z + g.w;
            if (Ar) {
                const f4v rv = *reinterpret_cast<const f4v*>(res + base + i);
                acc[2] += g.x * rv.x + g.y * rv.y + g.z * rv.z + g.w * rv.w;
            }
            *reinterpret_cast<f4v*>(gy + base + i) = g * a;
            *reinterpret_cast<f4v*>(gres + base + i) = g * ar;
        } else {
            const float g = out[base + i] > 0.f ? gout[base + i] + (gout2 ? gout2[base + i] : 0.f) : 0.f;
            acc[0] = fmaf(g, y[base + i], acc[0]);
            acc[1] += g;
            if (Ar) acc[2] = fmaf(g, res[base + i], acc[2]);
            gy[base + i] = g * a;
            gres[base + i] = g * ar;
        }
    }
    block_sum<3>(acc, sh);
    if (threadIdx.x == 0) {
        cfn_add64(&gA[nc], (double)acc[0]);
        cfn_add64(&gB[nc], (double)acc[1]);
        if (gAr) cfn_add64(&gAr[nc], (double)acc[2]);
    }
}

// ---- out = act(A*x + B)   (+ optional per-(n,c) sum / sumsq of x itself: channel statistics) ----
template <int VEC>
__global__ __launch_bounds__(256) void affine_act_fwd_kernel(const float* __restrict__ x, const double* __restrict__ A,
                                                             const double* __restrict__ B, int act, float* __restrict__ out,
                                                             long vol) {
    const long nc = blockIdx.y + (long)blockIdx.z * gridDim.y;      // (y, z) = exact factorisation of N*C (cfn_split_nc)
    const float a = A[nc], b = B[nc];
    const long base = nc * vol;
    long i = ((long)blockIdx.x * 256 * EW_ITEMS + threadIdx.x) * VEC;
#pragma unroll
    for (int k = 0; k < EW_ITEMS; ++k, i += 256 * VEC) {
        if (i >= vol) break;
        if (VEC == 4) {
            const f4v v = *reinterpret_cast<const f4v*>(x + base + i);
            f4v o;
            o.x = cfn_act_rt(fmaf(v.x, a, b), act); o.y = cfn_act_rt(fmaf(v.y, a, b), act);
            o.z = cfn_act_rt(fmaf(v.z, a, b), act); o.w = cfn_act_rt(fmaf(v.w, a, b), act);
            *reinterpret_cast<f4v*>(out + base + i) = o;
        } else {
            out[base + i] = cfn_act_rt(fmaf(x[base + i], a, b), act);
        }
    }
}

// dz = gout * act'(A*x+B);  gx = dz*A;  gA += sum dz*x;  gB += sum dz
template <int VEC>
__global__ __launch_bounds__(256) void affine_act_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ x,
                                                             const double* __restrict__ A, const double* __restrict__ B, int act,
                                                             float* __restrict__ gx, double* __restrict__ gA,
                                                             double* __restrict__ gB, long vol) {
    __shared__ float sh[8];
    const long nc = blockIdx.y + (long)blockIdx.z * gridDim.y;      // (y, z) = exact factorisation of N*C (cfn_split_nc)
    const float a = A[nc], b = B[nc];
    const long base = nc * vol;
    float acc[2] = {0.f, 0.f};
    long i = ((long)blockIdx.x * 256 * EW_ITEMS + threadIdx.x) * VEC;
#pragma unroll
    for (int k = 0; k < EW_ITEMS; ++k, i += 256 * VEC) {
        if (i >= vol) break;
#pragma unroll
        for (int u = 0; u < VEC; ++u) {
            const float xv = x[base + i + u];
            const float dz = gout[base + i + u] * cfn_act_grad_rt(fmaf(xv, a, b), act);
            acc[0] = fmaf(dz, xv, acc[0]);
            acc[1] += dz;
            gx[base + i + u] = dz * a;
        }
    }
    block_sum<2>(acc, sh);
    if (threadIdx.x == 0) { cfn_add64(&gA[nc], (double)acc[0]); cfn_add64(&gB[nc], (double)acc[1]); }
}

template <int VEC>
__global__ __launch_bounds__(256) void channel_stats_kernel(const float* __restrict__ x, double* __restrict__ sum,
                                                            double* __restrict__ sumsq, long vol) {
    __shared__ float sh[8];
    const long nc = blockIdx.y + (long)blockIdx.z * gridDim.y;      // (y, z) = exact factorisation of N*C (cfn_split_nc)
    const long base = nc * vol;
    float acc[2] = {0.f, 0.f};
    long i = ((long)blockIdx.x * 256 * EW_ITEMS + threadIdx.x) * VEC;
#pragma unroll
    for (int k = 0; k < EW_ITEMS; ++k, i += 256 * VEC) {
        if (i >= vol) break;
        if (VEC == 4) {
            const f4v v = *reinterpret_cast<const f4v*>(x + base + i);
            acc[0] += v.x + v.y + v.z + v.w;
            acc[1] += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        } else {
            const float v = x[base + i];
            acc[0] += v;
            acc[1] = fmaf(v, v, acc[1]);
        }
    }
    block_sum<2>(acc, sh);
    if (threadIdx.x == 0) { cfn_add64(&sum[nc], (double)acc[0]); cfn_add64(&sumsq[nc], (double)acc[1]); }
}

// ---- adaptive spatial average of act(A*x+B): (N,C,T,H,W) -> (N,C,T,OH,OW) -----------------------------
// adaptive_avg_pool3d((None,1,1)) of the head (x3d_fine.py:255,366) and ((None,7,7)) of the feature tower
// (:345-363); window of output cell o along a size-S axis: [floor(o*S/O), ceil((o+1)*S/O)) as in ATen.
__device__ __forceinline__ int ap_start(int o, int O, int S) { return (o * S) / O; }
__device__ __forceinline__ int ap_end(int o, int O, int S) { return ((o + 1) * S + O - 1) / O; }

__global__ __launch_bounds__(256) void pool_hw_fwd_kernel(const float* __restrict__ x, const double* __restrict__ A,
                                                          const double* __restrict__ B, int act, float* __restrict__ out,
                                                          int T, int H, int W, int OH, int OW) {
    const long nc = blockIdx.y + (long)blockIdx.z * gridDim.y;      // (y, z) = exact factorisation of N*C (cfn_split_nc)
    const long ovol = (long)T * OH * OW;
    const long o = (long)blockIdx.x * 256 + threadIdx.x;
    if (o >= ovol) return;
    const float a = A ? A[nc] : 1.0f, b = A ? B[nc] : 0.0f;
    const int ow = (int)(o % OW), oh = (int)((o / OW) % OH), t = (int)(o / ((long)OW * OH));
    const int h0 = ap_start(oh, OH, H), h1 = ap_end(oh, OH, H), w0 = ap_start(ow, OW, W), w1 = ap_end(ow, OW, W);
    const float* p = x + (nc * T + t) * (long)H * W;
    float s = 0.f;
    for (int i = h0; i < h1; ++i)
        for (int j = w0; j < w1; ++j) s += cfn_act_rt(fmaf(p[i * W + j], a, b), act);
    out[nc * ovol + o] = s / (float)((h1 - h0) * (w1 - w0));
}

__global__ __launch_bounds__(256) void pool_hw_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ x,
                                                          const double* __restrict__ A, const double* __restrict__ B, int act,
                                                          float* __restrict__ gx, double* __restrict__ gA,
                                                          double* __restrict__ gB, int T, int H, int W, int OH, int OW) {
    __shared__ float sh[8];
    const long nc = blockIdx.y + (long)blockIdx.z * gridDim.y;      // (y, z) = exact factorisation of N*C (cfn_split_nc)
    const long vol = (long)T * H * W;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    float acc[2] = {0.f, 0.f};
    if (i < vol) {
        const float a = A ? A[nc] : 1.0f, b = A ? B[nc] : 0.0f;
        const int w = (int)(i % W), h = (int)((i / W) % H), t = (int)(i / ((long)W * H));
        const float* gp = gout + (nc * T + t) * (long)OH * OW;
        float g = 0.f;
        // every output window that contains (h, w); adaptive windows may overlap, OH*OW <= 49 here
        for (int oh = 0; oh < OH; ++oh) {
            const int hs = ap_start(oh, OH, H), he = ap_end(oh, OH, H);
            if (h < hs || h >= he) continue;
            for (int ow = 0; ow < OW; ++ow) {
                const int ws = ap_start(ow, OW, W), we = ap_end(ow, OW, W);
                if (w >= ws && w < we) g += gp[oh * OW + ow] / (float)((he - hs) * (we - ws));
            }
        }
        const float xv = x[nc * vol + i];
        const float dz = g * cfn_act_grad_rt(fmaf(xv, a, b), act);
        acc[0] = dz * xv;
        acc[1] = dz;
        gx[nc * vol + i] = dz * a;
    }
    if (gA) {
        block_sum<2>(acc, sh);
        if (threadIdx.x == 0) { cfn_add64(&gA[nc], (double)acc[0]); cfn_add64(&gB[nc], (double)acc[1]); }
    }
}

// Global spatial mean (OH = OW = 1) of small planes (H*W <= 64: the head's 7x7, x3d_fine.py:255,366): one WAVE per frame -- lane =
// plane element, one coalesced load of the frame's contiguous run, prologue, wave reduction -- instead of one thread walking a
// whole plane (49 serial strided loads per thread: 0.69 ms for 173 MB at 8 clips x 256 frames).  Frames of a (n, c) row are dealt
// to the four waves of its workgroup, eight frames in flight per wave.
__global__ __launch_bounds__(256) void pool_hw1_fwd_kernel(const float* __restrict__ x, const double* __restrict__ A,
                                                           const double* __restrict__ B, int act, float* __restrict__ out,
                                                           int T, int HW) {
    const long nc = blockIdx.y + (long)blockIdx.z * gridDim.y;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float a = A ? (float)A[nc] : 1.0f, b = A ? (float)B[nc] : 0.0f;
    const float inv = 1.0f / (float)HW;
    const float* p = x + nc * (long)T * HW;
    for (int t0 = (blockIdx.x * 4 + wv) * 8; t0 < T; t0 += gridDim.x * 32) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (lane < HW && t0 + u < T) ? p[(long)(t0 + u) * HW + lane] : 0.0f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float s = (lane < HW && t0 + u < T) ? cfn_act_rt(fmaf(v[u], a, b), act) : 0.0f;
            s = cfn_wave_sum(s);
            if (lane == 0 && t0 + u < T) out[nc * T + t0 + u] = s * inv;
        }
    }
}

// backward of the same: one workgroup per (n, c) row, element-per-thread (coalesced), ONE block reduction and one pair of fp64
// atomics per row (the general kernel reduces and commits once per 256 elements)
__global__ __launch_bounds__(256) void pool_hw1_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ x,
                                                           const double* __restrict__ A, const double* __restrict__ B, int act,
                                                           float* __restrict__ gx, double* __restrict__ gA,
                                                           double* __restrict__ gB, int T, int HW) {
    __shared__ float sh[8];
    const long nc = blockIdx.y + (long)blockIdx.z * gridDim.y;
    const float a = A ? (float)A[nc] : 1.0f, b = A ? (float)B[nc] : 0.0f;
    const float inv = 1.0f / (float)HW;
    const long vol = (long)T * HW;
    const float* xp = x + nc * vol;
    const float* gp = gout + nc * T;
    float* op = gx + nc * vol;
    float acc[2] = {0.f, 0.f};
    for (long i = threadIdx.x; i < vol; i += 256) {
        const int t = (int)(i / HW);
        const float xv = xp[i];
        const float dz = gp[t] * inv * cfn_act_grad_rt(fmaf(xv, a, b), act);
        acc[0] = fmaf(dz, xv, acc[0]);
        acc[1] += dz;
        op[i] = dz * a;
    }
    if (gA) {
        block_sum<2>(acc, sh);
        if (threadIdx.x == 0) { cfn_add64(&gA[nc], (double)acc[0]); cfn_add64(&gB[nc], (double)acc[1]); }
    }
}

// ---- FiLM with 7x7 block-constant modulation (x3d_coarse.py:663-679 after the fusion branch has been
// evaluated at its native 7x7 resolution): out = x * m[h/f, w/f] + c[h/f, w/f] -------------------------
__global__ __launch_bounds__(256) void film_fwd_kernel(const float* __restrict__ x, const float* __restrict__ m,
                                                       const float* __restrict__ c, float* __restrict__ out, int T, int H,
                                                       int W, int f) {
    const long nc = blockIdx.y + (long)blockIdx.z * gridDim.y;      // (y, z) = exact factorisation of N*C (cfn_split_nc)
    const long vol = (long)T * H * W;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= vol) return;
    const int Hs = H / f, Ws = W / f;
    const int w = (int)(i % W), h = (int)((i / W) % H), t = (int)(i / ((long)W * H));
    const long s = (nc * T + t) * (long)Hs * Ws + (long)(h / f) * Ws + w / f;
    out[nc * vol + i] = fmaf(x[nc * vol + i], m[s], c[s]);
}

// gx = g*m ;  gm[s] = sum_window g*x ; gc[s] = sum_window g   (one thread per 7x7 cell)
__global__ __launch_bounds__(256) void film_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                       const float* __restrict__ m, float* __restrict__ gx,
                                                       float* __restrict__ gm, float* __restrict__ gc, int T, int H, int W,
                                                       int f) {
    const long nc = blockIdx.y + (long)blockIdx.z * gridDim.y;      // (y, z) = exact factorisation of N*C (cfn_split_nc)
    const int Hs = H / f, Ws = W / f;
    const long svol = (long)T * Hs * Ws;
    const long s = (long)blockIdx.x * 256 + threadIdx.x;
    if (s >= svol) return;
    const int ws = (int)(s % Ws), hs = (int)((s / Ws) % Hs), t = (int)(s / ((long)Ws * Hs));
    const long base = (nc * T + t) * (long)H * W + (long)hs * f * W + ws * f;
    const float mv = m[nc * svol + s];
    float a0 = 0.f, a1 = 0.f;
    for (int i = 0; i < f; ++i)
        for (int j = 0; j < f; ++j) {
            const float gv = g[base + i * W + j];
            a0 = fmaf(gv, x[base + i * W + j], a0);
            a1 += gv;
            gx[base + i * W + j] = gv * mv;
        }
    gm[nc * svol + s] = a0;
    gc[nc * svol + s] = a1;
}

// ---------------------------------------------------------------------------------------------
static inline dim3 ew_grid(long vol, long NC, int vec) { unsigned gy, gz; cfn_split_nc(NC, gy, gz); return dim3(cfn_cdiv(vol, 256L * EW_ITEMS * vec), gy, gz); }
static inline dim3 nc_grid(long xblocks, long NC) { unsigned gy, gz; cfn_split_nc(NC, gy, gz); return dim3((unsigned)xblocks, gy, gz); }
static inline bool ew_vec4(long vol, const void* p0, const void* p1 = nullptr, const void* p2 = nullptr,
                           const void* p3 = nullptr, const void* p4 = nullptr, const void* p5 = nullptr) {
    auto al = [](const void* p) { return p == nullptr || ((uintptr_t)p & 15) == 0; };
    return vol % 4 == 0 && al(p0) && al(p1) && al(p2) && al(p3) && al(p4) && al(p5);
}
// flat elementwise kernels: one buffer descriptor per channel (< 2 GB), a 1-D grid
static inline bool ew_flat_ok(long vol, long NC) {
    static const int on = getenv("CFN_EW_FLAT") ? atoi(getenv("CFN_EW_FLAT")) : 1;
    return on && vol * 4 < 0x7ffffff0L && NC * (long)cfn_cdiv(vol, 256L * EW_ITEMS * 4) < 0x7fffffffL && NC < 0x7fffffffL;
}
#define CFN_NC_CHECK(NC) CFN_REQUIRE((NC) > 0 && cfn_split_nc_ok(NC), "N*C = %ld has no grid factorisation", (long)(NC))

// words of the ReLU bit mask cfn_bn_add_relu_fwd can emit for (NC, vol); 0 = no mask for this shape (vol % 4 != 0)
extern "C" long cfn_bn_add_relu_mask_words(long NC, long vol) {
    if (NC <= 0 || vol <= 0 || vol % 4 != 0) return 0;
    return NC * (long)cfn_cdiv(vol, 256L * EW_ITEMS * 4) * 256;
}

extern "C" int cfn_bn_add_relu_fwd(const float* y, const double* A, const double* B, const float* res, const double* Ar,
                                   const double* Br, float* out, int* mask, long NC, long vol, void* stream) {
    CFN_REQUIRE(y && A && B && res && out, "cfn_bn_add_relu_fwd: null tensor");
    CFN_REQUIRE((Ar == nullptr) == (Br == nullptr), "cfn_bn_add_relu_fwd: Ar/Br mismatch");
    CFN_NC_CHECK(NC);
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_ELEMWISE, st, 12.0 * NC * vol);
    const bool v4 = ew_vec4(vol, y, res, out);
    CFN_REQUIRE(mask == nullptr || v4, "cfn_bn_add_relu_fwd: the bit mask needs vol %% 4 == 0 and 16-byte aligned tensors");
    if (v4 && ew_flat_ok(vol, NC)) {
        EwFlatArgs a = {};
        a.y = y; a.A = A; a.B = B; a.res = res; a.Ar = Ar; a.Br = Br; a.out = out; a.mask = (unsigned*)mask; a.vol = vol;
        a.nchunks = (unsigned)cfn_cdiv(vol, 256L * EW_ITEMS * 4);
        hipLaunchKernelGGL(bn_add_relu_fwd_flat_kernel, dim3((unsigned)(NC * a.nchunks)), dim3(256), 0, st, a);
    } else if (v4) hipLaunchKernelGGL(bn_add_relu_fwd_kernel<4>, ew_grid(vol, NC, 4), dim3(256), 0, st, y, A, B, res, Ar, Br, out, (unsigned*)mask, vol);
    else hipLaunchKernelGGL(bn_add_relu_fwd_kernel<1>, ew_grid(vol, NC, 1), dim3(256), 0, st, y, A, B, res, Ar, Br, out, (unsigned*)nullptr, vol);
    return cfn_check_launch("bn_add_relu_fwd");
}

extern "C" int cfn_bn_add_relu_bwd_g(const float* gout, const float* gout2, const float* out, const int* mask, const float* y,
                                     const float* res, float* g, double* gA, double* gB, double* gAr, long NC, long vol,
                                     void* stream) {
    CFN_REQUIRE(gout && y && g && gA && gB, "cfn_bn_add_relu_bwd_g: null tensor");
    CFN_REQUIRE((out != nullptr) != (mask != nullptr), "cfn_bn_add_relu_bwd_g: exactly one of out / mask");
    CFN_REQUIRE(gAr == nullptr || res != nullptr, "cfn_bn_add_relu_bwd_g: gAr needs res");
    CFN_NC_CHECK(NC);
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_ELEMWISE, st, (12.0 + (gout2 ? 4.0 : 0.0) + (out ? 4.0 : 0.125) + (gAr ? 4.0 : 0.0)) * NC * vol);
    const bool v4 = ew_vec4(vol, gout, out, y, gAr ? res : nullptr, g) && (((uintptr_t)gout2) & 15) == 0;
    CFN_REQUIRE(mask == nullptr || v4, "cfn_bn_add_relu_bwd_g: the bit mask needs vol %% 4 == 0 and 16-byte aligned tensors");
    if (v4 && ew_flat_ok(vol, NC)) {
        EwFlatArgs a = {};
        a.gout = gout; a.gout2 = gout2; a.outr = out; a.maskr = (const unsigned*)mask; a.y = y; a.res = res; a.g = g; a.gA = gA; a.gB = gB; a.gAr = gAr;
        a.vol = vol; a.nchunks = (unsigned)cfn_cdiv(vol, 256L * EW_ITEMS * 4);
        hipLaunchKernelGGL(bn_add_relu_bwd_g_flat_kernel, dim3((unsigned)(NC * a.nchunks)), dim3(256), 0, st, a);
    } else if (v4)
        hipLaunchKernelGGL(bn_add_relu_bwd_g_kernel<4>, ew_grid(vol, NC, 4), dim3(256), 0, st, gout, gout2, out, (const unsigned*)mask, y, res, g, gA, gB, gAr, vol);
    else
        hipLaunchKernelGGL(bn_add_relu_bwd_g_kernel<1>, ew_grid(vol, NC, 1), dim3(256), 0, st, gout, gout2, out, (const unsigned*)nullptr, y, res, g, gA, gB, gAr, vol);
    return cfn_check_launch("bn_add_relu_bwd_g");
}

extern "C" int cfn_bn_add_relu_bwd(const float* gout, const float* gout2, const float* out, const float* y, const double* A, const float* res,
                                   const double* Ar, float* gy, float* gres, double* gA, double* gB, double* gAr, long NC,
                                   long vol, void* stream) {
    CFN_REQUIRE(gout && out && y && A && gy && gres && gA && gB, "cfn_bn_add_relu_bwd: null tensor");
    CFN_REQUIRE(Ar == nullptr || (res != nullptr && gAr != nullptr), "cfn_bn_add_relu_bwd: Ar needs res and gAr");
    CFN_NC_CHECK(NC);
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_ELEMWISE, st, ((Ar ? 24.0 : 20.0) + (gout2 ? 4.0 : 0.0)) * NC * vol);
    if (ew_vec4(vol, gout, out, y, res, gy, gres) && (((uintptr_t)gout2) & 15) == 0)
        hipLaunchKernelGGL(bn_add_relu_bwd_kernel<4>, ew_grid(vol, NC, 4), dim3(256), 0, st, gout, gout2, out, y, A, res, Ar, gy, gres, gA, gB, gAr, vol);
    else
        hipLaunchKernelGGL(bn_add_relu_bwd_kernel<1>, ew_grid(vol, NC, 1), dim3(256), 0, st, gout, gout2, out, y, A, res, Ar, gy, gres, gA, gB, gAr, vol);
    return cfn_check_launch("bn_add_relu_bwd");
}

extern "C" int cfn_affine_act_fwd(const float* x, const double* A, const double* B, int act, float* out, long NC, long vol,
                                  void* stream) {
    CFN_REQUIRE(x && A && B && out, "cfn_affine_act_fwd: null tensor");
    CFN_NC_CHECK(NC);
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_ELEMWISE, st, 8.0 * NC * vol);
    if (ew_vec4(vol, x, out)) hipLaunchKernelGGL(affine_act_fwd_kernel<4>, ew_grid(vol, NC, 4), dim3(256), 0, st, x, A, B, act, out, vol);
    else hipLaunchKernelGGL(affine_act_fwd_kernel<1>, ew_grid(vol, NC, 1), dim3(256), 0, st, x, A, B, act, out, vol);
    return cfn_check_launch("affine_act_fwd");
}

extern "C" int cfn_affine_act_bwd(const float* gout, const float* x, const double* A, const double* B, int act, float* gx,
                                  double* gA, double* gB, long NC, long vol, void* stream) {
    CFN_REQUIRE(gout && x && A && B && gx && gA && gB, "cfn_affine_act_bwd: null tensor");
    CFN_NC_CHECK(NC);
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_ELEMWISE, st, 12.0 * NC * vol);
    if (vol % 4 == 0) hipLaunchKernelGGL(affine_act_bwd_kernel<4>, ew_grid(vol, NC, 4), dim3(256), 0, st, gout, x, A, B, act, gx, gA, gB, vol);
    else hipLaunchKernelGGL(affine_act_bwd_kernel<1>, ew_grid(vol, NC, 1), dim3(256), 0, st, gout, x, A, B, act, gx, gA, gB, vol);
    return cfn_check_launch("affine_act_bwd");
}

extern "C" int cfn_channel_stats(const float* x, double* sum, double* sumsq, long NC, long vol, void* stream) {
    CFN_REQUIRE(x && sum && sumsq, "cfn_channel_stats: null tensor");
    CFN_NC_CHECK(NC);
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_ELEMWISE, st, 4.0 * NC * vol);
    if (ew_vec4(vol, x)) hipLaunchKernelGGL(channel_stats_kernel<4>, ew_grid(vol, NC, 4), dim3(256), 0, st, x, sum, sumsq, vol);
    else hipLaunchKernelGGL(channel_stats_kernel<1>, ew_grid(vol, NC, 1), dim3(256), 0, st, x, sum, sumsq, vol);
    return cfn_check_launch("channel_stats");
}

extern "C" int cfn_pool_hw_fwd(const float* x, const double* A, const double* B, int act, float* out, long NC, int T, int H,
                               int W, int OH, int OW, void* stream) {
    CFN_REQUIRE(x && out, "cfn_pool_hw_fwd: null tensor");
    CFN_REQUIRE((A == nullptr) == (B == nullptr), "cfn_pool_hw_fwd: A/B mismatch");
    CFN_REQUIRE(OH > 0 && OW > 0 && OH <= 64 && OW <= 64, "cfn_pool_hw_fwd: bad output size %dx%d", OH, OW);
    CFN_NC_CHECK(NC);
    hipStream_t st = (hipStream_t)stream;
    const long ovol = (long)T * OH * OW;
    if (OH == 1 && OW == 1 && H * W <= 64) {
        hipLaunchKernelGGL(pool_hw1_fwd_kernel, nc_grid(cfn_cdiv(T, 32) < 16 ? cfn_cdiv(T, 32) : 16, NC), dim3(256), 0, st, x, A, B, act, out, T, H * W);
        return cfn_check_launch("pool_hw_fwd(global mean)");
    }
    hipLaunchKernelGGL(pool_hw_fwd_kernel, nc_grid(cfn_cdiv(ovol, 256), NC), dim3(256), 0, st, x, A, B, act, out, T, H, W, OH, OW);
    return cfn_check_launch("pool_hw_fwd");
}

extern "C" int cfn_pool_hw_bwd(const float* gout, const float* x, const double* A, const double* B, int act, float* gx,
                               double* gA, double* gB, long NC, int T, int H, int W, int OH, int OW, void* stream) {
    CFN_REQUIRE(gout && x && gx, "cfn_pool_hw_bwd: null tensor");
    CFN_REQUIRE((A == nullptr) == (B == nullptr), "cfn_pool_hw_bwd: A/B mismatch");
    CFN_REQUIRE(A == nullptr || (gA && gB), "cfn_pool_hw_bwd: prologue needs gA, gB");
    CFN_REQUIRE(OH > 0 && OW > 0 && OH <= 64 && OW <= 64, "cfn_pool_hw_bwd: bad output size %dx%d", OH, OW);
    CFN_NC_CHECK(NC);
    hipStream_t st = (hipStream_t)stream;
    const long vol = (long)T * H * W;
    if (OH == 1 && OW == 1 && H * W <= 64) {
        hipLaunchKernelGGL(pool_hw1_bwd_kernel, nc_grid(1, NC), dim3(256), 0, st, gout, x, A, B, act, gx,
                           A ? gA : nullptr, A ? gB : nullptr, T, H * W);
        return cfn_check_launch("pool_hw_bwd(global mean)");
    }
    hipLaunchKernelGGL(pool_hw_bwd_kernel, nc_grid(cfn_cdiv(vol, 256), NC), dim3(256), 0, st, gout, x, A, B, act, gx,
                       A ? gA : nullptr, A ? gB : nullptr, T, H, W, OH, OW);
    return cfn_check_launch("pool_hw_bwd");
}

extern "C" int cfn_film_fwd(const float* x, const float* m, const float* c, float* out, long NC, int T, int H, int W, int f,
                            void* stream) {
    CFN_REQUIRE(x && m && c && out, "cfn_film_fwd: null tensor");
    CFN_REQUIRE(f > 0 && H % f == 0 && W % f == 0, "cfn_film_fwd: factor %d does not tile %dx%d", f, H, W);
    CFN_NC_CHECK(NC);
    hipStream_t st = (hipStream_t)stream;
    const long vol = (long)T * H * W;
    CfnProfScope prof(CFN_K_FUSION, st, 8.0 * NC * vol);
    hipLaunchKernelGGL(film_fwd_kernel, nc_grid(cfn_cdiv(vol, 256), NC), dim3(256), 0, st, x, m, c, out, T, H, W, f);
    return cfn_check_launch("film_fwd");
}

extern "C" int cfn_film_bwd(const float* g, const float* x, const float* m, float* gx, float* gm, float* gc, long NC, int T,
                            int H, int W, int f, void* stream) {
    CFN_REQUIRE(g && x && m && gx && gm && gc, "cfn_film_bwd: null tensor");
    CFN_REQUIRE(f > 0 && H % f == 0 && W % f == 0, "cfn_film_bwd: factor %d does not tile %dx%d", f, H, W);
    CFN_NC_CHECK(NC);
    hipStream_t st = (hipStream_t)stream;
    const long svol = (long)T * (H / f) * (W / f);
    CfnProfScope prof(CFN_K_FUSION, st, 12.0 * NC * T * H * W);
    hipLaunchKernelGGL(film_bwd_kernel, nc_grid(cfn_cdiv(svol, 256), NC), dim3(256), 0, st, g, x, m, gx, gm, gc, T, H, W, f);
    return cfn_check_launch("film_bwd");
}
