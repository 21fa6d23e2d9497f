// Depthwise 5x1x1 temporal convolution of the X3D stem (conv1_t, x3d_fine.py:216-222): pure streaming.
// A thread owns 4 consecutive positions of one (n,c) plane and marches along t with a 5-frame
// register window, so every input element is loaded exactly once per t-chunk (+4 halo frames) with
// fully coalesced float4 accesses; no LDS.
//   T5_FWD    y = conv(x)                       + per-(n,c) sum / sumsq of y
//   T5_DGRAD  gx = conv_flipped(gy + gs + 2 y gq)
//   T5_WGRAD  gw[c][kt] += sum (gy + gs + 2 y gq)[t] * x[t+kt-2]
#include "cfn_common.h"
#include "h16.h"

typedef float __attribute__((ext_vector_type(4))) f4v;
typedef unsigned __attribute__((ext_vector_type(4))) u4v_t5;
enum { T5_FWD = 0, T5_DGRAD = 1, T5_WGRAD = 2 };

// BF (bf16 activation path): the conv OUTPUT side (y, gy) is bf16, the input side (x, gx) stays fp32 -- the stem conv
// that produces x and consumes gx runs in fp32 (its 3-channel input clip is fp32), so only conv1_t's output is narrowed.
struct T5Args {
    const void* src;     // FWD/WGRAD: x (N,C,T,P) fp32    DGRAD: gy (fp32 | bf16)
    const void* src2;    // DGRAD: y (for gq) or null
    const double* gs; const double* gq;
    const float* w;      // (C,5)
    void* dst;           // FWD: y (fp32 | bf16)   DGRAD: gx fp32
    const void* gy;      // WGRAD (fp32 | bf16)
    const void* yout;    // WGRAD: y (for gq) or null
    double* s1; double* s2;
    int C, T, TT, nchunks, pchunks;
    long plane;
};

// b16: 0 = fp32 elements, H16_BF16 / H16_F16 = 2-byte elements of that kind (h16.h)
__device__ __forceinline__ f4v t5_ld(const void* ptr, long idx, int b16, int vec) {
    f4v v = {0.f, 0.f, 0.f, 0.f};
    if (b16) {
        const unsigned short* p = static_cast<const unsigned short*>(ptr) + idx;
        if (vec == 4) {
            const uint2 u = *reinterpret_cast<const uint2*>(p);
            if (b16 == H16_F16) { v.x = h16k_lo<H16_F16>(u.x); v.y = h16k_hi<H16_F16>(u.x); v.z = h16k_lo<H16_F16>(u.y); v.w = h16k_hi<H16_F16>(u.y); }
            else { v.x = h16k_lo<H16_BF16>(u.x); v.y = h16k_hi<H16_BF16>(u.x); v.z = h16k_lo<H16_BF16>(u.y); v.w = h16k_hi<H16_BF16>(u.y); }
        } else v.x = b16 == H16_F16 ? h16k_lo<H16_F16>((unsigned)p[0]) : h16k_lo<H16_BF16>((unsigned)p[0]);
    } else {
        const float* p = static_cast<const float*>(ptr) + idx;
        if (vec == 4) v = *reinterpret_cast<const f4v*>(p);
        else v.x = p[0];
    }
    return v;
}

template <int MODE, int VEC, int BF>
__global__ __launch_bounds__(256) void dwt5_kernel(const T5Args a) {
    typedef typename h16_types<BF ? BF : H16_BF16>::v4 bf4;
    constexpr int SRC16 = MODE == T5_DGRAD ? BF : 0;      // element type of src / src2
    constexpr int DST16 = MODE == T5_FWD ? BF : 0;        // element type of dst
    __shared__ float sh[20];
    const long nc = blockIdx.y + (long)blockIdx.z * gridDim.y;
    const int c = (int)(nc % a.C);
    const int chunk = blockIdx.x % a.nchunks, pc = blockIdx.x / a.nchunks;
    const long p = ((long)pc * 256 + threadIdx.x) * VEC;
    const bool ok = p < a.plane;
    const int t0 = chunk * a.TT, t1 = min(t0 + a.TT, a.T);
    const long base = nc * a.T * a.plane + p;

    float wk[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) wk[k] = a.w[c * 5 + (MODE == T5_DGRAD ? 4 - k : k)];
    const float gsv = (MODE != T5_FWD && a.gs) ? (float)a.gs[nc] : 0.0f;
    const float gqv = (MODE != T5_FWD && a.gq) ? 2.0f * (float)a.gq[nc] : 0.0f;

    auto ldt = [&](const void* ptr, int t, int b16) -> f4v {
        f4v v = {0.f, 0.f, 0.f, 0.f};
        if (ok && t >= 0 && t < a.T) v = t5_ld(ptr, base + (long)t * a.plane, b16, VEC);
        return v;
    };
    auto ld = [&](const void* ptr, int t) -> f4v { return ldt(ptr, t, SRC16); };
    auto ld_src = [&](int t) -> f4v {   // staged tensor of the window
        f4v v = ld(a.src, t);
        if (MODE == T5_DGRAD && ok && t >= 0 && t < a.T) {
            v += gsv;
            if (a.src2) v += ld(a.src2, t) * gqv;
        }
        return v;
    };

    // 5-frame register window + PF frames of look-ahead: PF independent loads stay in flight per thread, so the
    // stream is not latency bound by one dependent load per output frame.  Look-ahead registers hold RAW loads; the
    // gy + gs + 2*y*gq arithmetic is applied when a frame enters the window, so no wait sits right behind a load.
    constexpr int PF = 4;
    const f4v zero = {0.f, 0.f, 0.f, 0.f};
    const bool two = MODE == T5_DGRAD && a.src2 != nullptr;
    auto fin_src = [&](f4v v, f4v v2, int t) -> f4v {
        if (MODE != T5_DGRAD) return v;
        if (!(ok && t >= 0 && t < a.T)) return zero;
        v += gsv;
        if (two) v += v2 * gqv;
        return v;
    };
    f4v win[5], nxt[PF], nxt2[PF];
#pragma unroll
    for (int k = 0; k < 4; ++k) win[k + 1] = ld_src(t0 - 2 + k);   // frames t0-2 .. t0+1
#pragma unroll
    for (int k = 0; k < PF; ++k) {                                 // t0+2 ..
        const bool need = t0 + 2 + k <= t1 + 1;
        nxt[k] = need ? ld(a.src, t0 + 2 + k) : zero;
        nxt2[k] = (need && two) ? ld(a.src2, t0 + 2 + k) : zero;
    }
    f4v gr4[PF], yr4[PF];   // WGRAD: raw gy / y of frames t .. t+PF-1
    const bool wy = MODE == T5_WGRAD && a.yout != nullptr;
    if (MODE == T5_WGRAD) {
#pragma unroll
        for (int k = 0; k < PF; ++k) { gr4[k] = ldt(a.gy, t0 + k, BF); yr4[k] = wy ? ldt(a.yout, t0 + k, BF) : zero; }
    }
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    float st1 = 0.f, st2 = 0.f;
    for (int tb = t0; tb < t1; tb += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int t = tb + u;
            if (t >= t1) break;
#pragma unroll
            for (int k = 0; k < 4; ++k) win[k] = win[k + 1];
            win[4] = fin_src(nxt[u], nxt2[u], t + 2);
            const bool need = t + 2 + PF <= t1 + 1;
            nxt[u] = need ? ld(a.src, t + 2 + PF) : zero;
            if (MODE == T5_DGRAD) nxt2[u] = (need && two) ? ld(a.src2, t + 2 + PF) : zero;
            if (MODE == T5_WGRAD) {
                f4v g = zero;
                if (ok) { g = gr4[u] + gsv; if (wy) g += yr4[u] * gqv; }    // t < t1 <= T here
                gr4[u] = (t + PF < t1) ? ldt(a.gy, t + PF, BF) : zero;
                if (wy) yr4[u] = (t + PF < t1) ? ldt(a.yout, t + PF, BF) : zero;
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const f4v pr = g * win[k];
                    acc[k] += VEC == 4 ? pr.x + pr.y + pr.z + pr.w : pr.x;
                }
            } else {
                f4v y = win[0] * wk[0] + win[1] * wk[1] + win[2] * wk[2] + win[3] * wk[3] + win[4] * wk[4];
                if (ok) {
                    if (DST16) {      // statistics are taken over the rounded values the consumer will read
                        const bf4 yb = __builtin_convertvector(y, bf4);
                        unsigned short* dp = static_cast<unsigned short*>(a.dst) + base + (long)t * a.plane;
                        if (VEC == 4) *reinterpret_cast<bf4*>(dp) = yb;
                        else dp[0] = __builtin_bit_cast(unsigned short, yb.x);
                        y = __builtin_convertvector(yb, f4v);
                    } else {
                        float* dp = static_cast<float*>(a.dst) + base + (long)t * a.plane;
                        if (VEC == 4) *reinterpret_cast<f4v*>(dp) = y;
                        else dp[0] = y.x;
                    }
                    if (MODE == T5_FWD) {
                        if (VEC == 4) { st1 += y.x + y.y + y.z + y.w; st2 += y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w; }
                        else { st1 += y.x; st2 = fmaf(y.x, y.x, st2); }
                    }
                }
            }
        }
    }
    // block reductions
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (MODE == T5_WGRAD) {
#pragma unroll
        for (int k = 0; k < 5; ++k) { acc[k] = cfn_wave_sum(acc[k]); if (lane == 0) sh[k * 4 + wave] = acc[k]; }
        __syncthreads();
        if (threadIdx.x < 5)
            cfn_add64(&a.s1[c * 5 + threadIdx.x],
                      (double)(sh[threadIdx.x * 4] + sh[threadIdx.x * 4 + 1] + sh[threadIdx.x * 4 + 2] + sh[threadIdx.x * 4 + 3]));
    } else if (MODE == T5_FWD && a.s1) {
        st1 = cfn_wave_sum(st1); st2 = cfn_wave_sum(st2);
        if (lane == 0) { sh[wave] = st1; sh[4 + wave] = st2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            cfn_add64(&a.s1[nc], (double)(sh[0] + sh[1] + sh[2] + sh[3]));
            cfn_add64(&a.s2[nc], (double)(sh[4] + sh[5] + sh[6] + sh[7]));
        }
    }
}

// Forward, float4 rows (plane % 4 == 0): the same march, but every global access of the frame loop is an UNCONDITIONAL
// buffer load / store (an unwanted access gets an out-of-range offset).  With the loads under `if (t < T)` the compiler can
// only wait with vmcnt(0), i.e. for the load it issued a moment ago as well: one HBM round trip per frame and 4.9 TB/s; with
// exact vmcnt(N) waits the PF look-ahead loads really stay in flight (a one-float4-per-thread copy of the same tensor runs at
// 6.35 TB/s on this box, tools/probe/stream_probe.hip).  env CFN_T5_STREAM=0 falls back to dwt5_kernel<T5_FWD>.
template <int BF>
__global__ __launch_bounds__(256) void dwt5_fwd_stream_kernel(const T5Args a) {
    typedef typename h16_types<BF ? BF : H16_BF16>::v4 bf4;
    typedef unsigned __attribute__((ext_vector_type(2))) u2v;
    constexpr int PF = 3, OOB = 0x7ffffff0, OES = BF ? 2 : 4;
    __shared__ float sh[8];
    const long nc = blockIdx.y + (long)blockIdx.z * gridDim.y;
    const int c = (int)(nc % a.C);
    const int chunk = blockIdx.x % a.nchunks, pc = blockIdx.x / a.nchunks;
    const int p = (pc * 256 + (int)threadIdx.x) * 4;
    const bool ok = p < a.plane;
    const int T = a.T, t0 = chunk * a.TT, t1 = min(t0 + a.TT, T);
    const int plane = (int)a.plane;
    __amdgpu_buffer_rsrc_t rx = cfn_rsrc(static_cast<const float*>(a.src) + nc * T * a.plane, (unsigned)((long)T * plane * 4));
    __amdgpu_buffer_rsrc_t ry = cfn_rsrc(static_cast<char*>(a.dst) + nc * T * a.plane * OES, (unsigned)((long)T * plane * OES));
    const int vx = ok ? p * 4 : OOB, vy = ok ? p * OES : OOB;
    float wk[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) wk[k] = cfn_uni(a.w[c * 5 + k]);
    auto ld = [&](int t) -> f4v {
        const bool tv = t >= 0 && t < T && t <= t1 + 1;                   // wave uniform
        return __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(rx, tv ? vx : OOB, tv ? t * plane * 4 : 0, 0));
    };
    // ring of RING = 5 + PF frame registers with STATIC slots (the loop is unrolled RING steps): slot (k % RING) holds frame
    // t0 - 2 + k.  No register moves between steps, so the wait before step j is for the load issued PF steps earlier and the
    // PF - 1 younger loads (and the stores) stay in flight.
    constexpr int RING = 5 + PF;
    f4v R[RING];
#pragma unroll
    for (int k = 0; k < RING - 1; ++k) R[k] = ld(t0 - 2 + k);             // frames t0-2 .. t0+1+PF (slots 0 .. RING-2)
    float st1 = 0.f, st2 = 0.f;
    for (int tb = t0; tb < t1; tb += RING) {
#pragma unroll
        for (int j = 0; j < RING; ++j) {
            const int t = tb + j;
            const bool em = t < t1;                                        // wave uniform: steps beyond the chunk store nothing
            R[(j + RING - 1) % RING] = ld(t + 2 + PF);                     // frame t-3's slot is free
            f4v y = R[j % RING] * wk[0] + R[(j + 1) % RING] * wk[1] + R[(j + 2) % RING] * wk[2] + R[(j + 3) % RING] * wk[3] + R[(j + 4) % RING] * wk[4];
            const int so = em ? t * plane * OES : 0;
            if (BF) {                                                      // statistics over the rounded values the consumer reads
                const bf4 yb = __builtin_convertvector(y, bf4);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2v, yb), ry, em ? vy : OOB, so, 0);
                y = __builtin_convertvector(yb, f4v);
            } else {
                cfn_bst128(__builtin_bit_cast(u4v_t5, y), ry, em ? vy : OOB, so);
            }
            const float m = (em && ok) ? 1.0f : 0.0f;
            const f4v ym = y * m;
            st1 += ym.x + ym.y + ym.z + ym.w;
            st2 += ym.x * y.x + ym.y * y.y + ym.z * y.z + ym.w * y.w;
        }
    }
    if (a.s1) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        st1 = cfn_wave_sum(st1); st2 = cfn_wave_sum(st2);
        if (lane == 0) { sh[wave] = st1; sh[4 + wave] = st2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            cfn_add64(&a.s1[nc], (double)(sh[0] + sh[1] + sh[2] + sh[3]));
            cfn_add64(&a.s2[nc], (double)(sh[4] + sh[5] + sh[6] + sh[7]));
        }
    }
}

// FLAT forward (round 3, tools/probe/t5_probe.hip): one thread = TO consecutive output frames of ONE float4 position, all
// TO + 4 input frames requested up front, blocks in memory order, and each XCD walks one contiguous eighth of the items
// (cfn_xcd_remap) so that the temporal halo re-reads of the neighbouring frame group hit ITS L2.  Measured on conv1_t
// (8 x 24 x 256 x 112 x 112): marching kernel 5.0-5.2 TB/s (deeper look-ahead: +2 %), flat TO = 4 / 8: 5.66 / 5.77 TB/s; the same
// flat kernel without the XCD remap 3.6 TB/s, with the block order scrambled inside each XCD 3.4-4.1 TB/s, TO = 1 (5 x L2 reads)
// 4.0 TB/s: what this kernel responds to is the ORDER in which the chip walks memory and the L2 re-read factor, not the
// look-ahead depth.  Ragged T / planes: surplus frames and threads get out-of-range offsets.
template <int BF, int TO>
__global__ __launch_bounds__(256) void dwt5_fwd_flat_kernel(const T5Args a) {
    typedef typename h16_types<BF ? BF : H16_BF16>::v4 bf4;
    typedef unsigned __attribute__((ext_vector_type(2))) u2v;
    constexpr int OOB = 0x7ffffff0, OES = BF ? 2 : 4;
    __shared__ float sh[8];
    const unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    const unsigned bpn = (unsigned)a.pchunks;                     // blocks per (n, c)
    const long nc = cfn_uni((int)(L / bpn));
    const unsigned item = (L - (unsigned)nc * bpn) * 256u + threadIdx.x;
    const int c = (int)(nc % a.C), T = a.T, plane = (int)a.plane, P4 = plane >> 2;
    const int tg = item / (unsigned)P4, p4 = item - tg * P4;
    const int t0 = tg * TO;
    const bool ok = t0 < T;
    __amdgpu_buffer_rsrc_t rx = cfn_rsrc(static_cast<const float*>(a.src) + nc * T * a.plane, (unsigned)((long)T * plane * 4));
    __amdgpu_buffer_rsrc_t ry = cfn_rsrc(static_cast<char*>(a.dst) + nc * T * a.plane * OES, (unsigned)((long)T * plane * OES));
    float wk[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) wk[k] = cfn_uni(a.w[c * 5 + k]);
    f4v R[TO + 4];
#pragma unroll
    for (int k = 0; k < TO + 4; ++k) {
        const int t = t0 - 2 + k;
        const bool tv = ok && t >= 0 && t < T;
        R[k] = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(rx, tv ? (t * plane + p4 * 4) * 4 : OOB, 0, 0));
    }
    float st1 = 0.f, st2 = 0.f;
#pragma unroll
    for (int j = 0; j < TO; ++j) {
        f4v y = R[j] * wk[0] + R[j + 1] * wk[1] + R[j + 2] * wk[2] + R[j + 3] * wk[3] + R[j + 4] * wk[4];
        const bool em = ok && t0 + j < T;
        const int vo = em ? ((t0 + j) * plane + p4 * 4) * OES : OOB;
        if (BF) {                                                      // statistics over the rounded values the consumer reads
            const bf4 yb = __builtin_convertvector(y, bf4);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2v, yb), ry, vo, 0, 0);
            y = __builtin_convertvector(yb, f4v);
        } else {
            cfn_bst128(__builtin_bit_cast(u4v_t5, y), ry, vo, 0);
        }
        const f4v ym = y * (em ? 1.0f : 0.0f);
        st1 += ym.x + ym.y + ym.z + ym.w;
        st2 += ym.x * y.x + ym.y * y.y + ym.z * y.z + ym.w * y.w;
    }
    if (a.s1) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        st1 = cfn_wave_sum(st1); st2 = cfn_wave_sum(st2);
        if (lane == 0) { sh[wave] = st1; sh[4 + wave] = st2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            cfn_add64(&a.s1[nc], (double)(sh[0] + sh[1] + sh[2] + sh[3]));
            cfn_add64(&a.s2[nc], (double)(sh[4] + sh[5] + sh[6] + sh[7]));
        }
    }
}

// -1 = not handled
template <int BF>
static int t5_fwd_flat(T5Args& a, int N, hipStream_t st) {
    static const int on = getenv("CFN_T5_FLAT") ? atoi(getenv("CFN_T5_FLAT")) : 1;       // 0: marching kernel, 4 / 8: force TO
    if (!on || a.plane % 4 != 0 || (long)a.T * a.plane * 4 >= 0x7ffffff0L) return -1;
    if ((((uintptr_t)a.src | (uintptr_t)a.dst) & 15) != 0) return -1;
    const int TO = on == 4 || on == 8 ? on : (a.T >= 32 ? 8 : 4);
    const long per_nc = (long)cfn_cdiv(a.T, TO) * (a.plane / 4);
    const long bpn = cfn_cdiv(per_nc, 256L), blocks = bpn * N * a.C;
    if (blocks >= 0x7fffffffL || (long)N * a.C >= 0x7fffffffL) return -1;
    a.pchunks = (int)bpn;
    if (TO == 8) hipLaunchKernelGGL((dwt5_fwd_flat_kernel<BF, 8>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((dwt5_fwd_flat_kernel<BF, 4>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    return cfn_check_launch("dwconv_t5 flat forward");
}

// Backward, float4 rows: data gradient AND weight gradient in one march (gy, y, x read once, gx written once: 4 tensor
// passes instead of the 6 of dwt5_kernel<T5_DGRAD> + <T5_WGRAD>), same streaming scheme as dwt5_fwd_stream_kernel: static
// register rings, unconditional buffer accesses.  Ring slot (k % RING) holds frame t0 - 2 + k of g' = gy + gs + 2 y gq and of
// x; a frame's raw gy / y land PF steps before it enters the 5-frame window and are combined in place at that step.
//   gx(t)  = sum_k g'(t - 2 + k) w[4 - k]          gw[k] += sum_t g'(t) x(t - 2 + k)
template <int BF, bool HASY>
__global__ __launch_bounds__(256) void dwt5_bwd_fused_kernel(const T5Args a) {
    typedef unsigned __attribute__((ext_vector_type(2))) u2v;
    constexpr int PF = 3, RING = 5 + PF, OOB = 0x7ffffff0, GES = BF ? 2 : 4;
    __shared__ float sh[20];
    const long nc = blockIdx.y + (long)blockIdx.z * gridDim.y;
    const int c = (int)(nc % a.C);
    const int chunk = blockIdx.x % a.nchunks, pc = blockIdx.x / a.nchunks;
    const int p = (pc * 256 + (int)threadIdx.x) * 4;
    const bool ok = p < a.plane;
    const int T = a.T, t0 = chunk * a.TT, t1 = min(t0 + a.TT, T);
    const int plane = (int)a.plane;
    // a.src = gy, a.src2 = y (output side: fp32 | bf16), a.yout = x (fp32), a.dst = gx (fp32), a.s1 = gw
    __amdgpu_buffer_rsrc_t rg = cfn_rsrc(static_cast<const char*>(a.src) + nc * T * a.plane * GES, (unsigned)((long)T * plane * GES));
    __amdgpu_buffer_rsrc_t ry = cfn_rsrc(static_cast<const char*>(HASY ? a.src2 : a.src) + nc * T * a.plane * GES, (unsigned)((long)T * plane * GES));
    __amdgpu_buffer_rsrc_t rx = cfn_rsrc(static_cast<const float*>(a.yout) + nc * T * a.plane, (unsigned)((long)T * plane * 4));
    __amdgpu_buffer_rsrc_t rd = cfn_rsrc(static_cast<float*>(a.dst) + nc * T * a.plane, (unsigned)((long)T * plane * 4));
    const int vg = ok ? p * GES : OOB, vx = ok ? p * 4 : OOB;
    float wk[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) wk[k] = cfn_uni(a.w[c * 5 + 4 - k]);        // flipped taps
    const float gsv = cfn_uni(a.gs ? (float)a.gs[nc] : 0.0f);
    const float gqv = cfn_uni((HASY && a.gq) ? 2.0f * (float)a.gq[nc] : 0.0f);
    auto wanted = [&](int t) { return t >= 0 && t < T && t <= t1 + 1; };     // wave uniform
    auto ldg = [&](__amdgpu_buffer_rsrc_t r, int t) -> f4v {
        const bool tv = wanted(t);
        const int vo = tv ? vg : OOB, so = tv ? t * plane * GES : 0;
        if (BF) {
            const u2v u = __builtin_bit_cast(u2v, __builtin_amdgcn_raw_buffer_load_b64(r, vo, so, 0));
            return (f4v){h16k_lo<BF ? BF : H16_BF16>(u.x), h16k_hi<BF ? BF : H16_BF16>(u.x), h16k_lo<BF ? BF : H16_BF16>(u.y), h16k_hi<BF ? BF : H16_BF16>(u.y)};
        }
        return __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0));
    };
    auto ldx = [&](int t) -> f4v {
        const bool tv = wanted(t);
        return __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(rx, tv ? vx : OOB, tv ? t * plane * 4 : 0, 0));
    };
    auto fin = [&](f4v g, f4v y, int t) -> f4v {                            // g' of frame t (zero outside the clip)
        const float m = (t >= 0 && t < T && ok) ? 1.0f : 0.0f;
        f4v v = g + gsv;
        if (HASY) v += y * gqv;
        return v * m;
    };
    f4v RG[RING], RY[RING], RX[RING];
#pragma unroll
    for (int k = 0; k < RING - 1; ++k) {                                   // frames t0-2 .. t0+1+PF (slots 0 .. RING-2)
        RG[k] = ldg(rg, t0 - 2 + k);
        if (HASY) RY[k] = ldg(ry, t0 - 2 + k);
        RX[k] = ldx(t0 - 2 + k);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) RG[k] = fin(RG[k], HASY ? RY[k] : RG[k], t0 - 2 + k);   // frames t0-2 .. t0+1 enter the window now
    f4v acc[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) acc[k] = (f4v){0.f, 0.f, 0.f, 0.f};
    for (int tb = t0; tb < t1; tb += RING) {
#pragma unroll
        for (int j = 0; j < RING; ++j) {
            const int t = tb + j;
            const bool em = t < t1;                                        // wave uniform: steps beyond the chunk contribute nothing
            const int sn = (j + RING - 1) % RING;                          // frame t-3's slot is free: frame t + 2 + PF
            RG[sn] = ldg(rg, t + 2 + PF);
            if (HASY) RY[sn] = ldg(ry, t + 2 + PF);
            RX[sn] = ldx(t + 2 + PF);
            const int s4 = (j + 4) % RING;                                 // frame t+2 enters the window
            RG[s4] = fin(RG[s4], HASY ? RY[s4] : RG[s4], t + 2);
            const f4v gx = RG[j % RING] * wk[0] + RG[(j + 1) % RING] * wk[1] + RG[(j + 2) % RING] * wk[2] + RG[(j + 3) % RING] * wk[3] +
                           RG[(j + 4) % RING] * wk[4];
            cfn_bst128(__builtin_bit_cast(u4v_t5, gx), rd, em ? vx : OOB, em ? t * plane * 4 : 0);
            const f4v gc = RG[(j + 2) % RING] * (em ? 1.0f : 0.0f);        // g'(t)
#pragma unroll
            for (int k = 0; k < 5; ++k) acc[k] += gc * RX[(j + k) % RING];
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const float v = cfn_wave_sum(acc[k].x + acc[k].y + acc[k].z + acc[k].w);
        if (lane == 0) sh[k * 4 + wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < 5)
        cfn_add64(&a.s1[c * 5 + threadIdx.x], (double)(sh[threadIdx.x * 4] + sh[threadIdx.x * 4 + 1] + sh[threadIdx.x * 4 + 2] + sh[threadIdx.x * 4 + 3]));
}

// FLAT fused backward (same mapping as dwt5_fwd_flat_kernel): one thread = TO consecutive frames of one float4 position.
//   gx(t) = sum_k g'(t - 2 + k) w[4 - k]                      t in the thread's TO frames (g' over TO + 4 frames)
//   gw[k] += sum_s x(s) g'(s + 2 - k)                          s in the thread's TO frames -- the weight-gradient sum re-indexed by
// the frame of x, so that x needs NO halo (gy and y are read (TO + 4) / TO times out of L2, x once).
template <int BF, bool HASY, int TO>
__global__ __launch_bounds__(256) void dwt5_bwd_flat_kernel(const T5Args a) {
    typedef unsigned __attribute__((ext_vector_type(2))) u2v;
    constexpr int OOB = 0x7ffffff0, GES = BF ? 2 : 4;
    __shared__ float sh[20];
    const unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    const unsigned bpn = (unsigned)a.pchunks;                     // blocks per (n, c)
    const long nc = cfn_uni((int)(L / bpn));
    const unsigned item = (L - (unsigned)nc * bpn) * 256u + threadIdx.x;
    const int c = (int)(nc % a.C), T = a.T, plane = (int)a.plane, P4 = plane >> 2;
    const int tg = item / (unsigned)P4, p4 = item - tg * P4;
    const int t0 = tg * TO;
    const bool ok = t0 < T;
    // a.src = gy, a.src2 = y (output side: fp32 | bf16), a.yout = x (fp32), a.dst = gx (fp32), a.s1 = gw
    __amdgpu_buffer_rsrc_t rg = cfn_rsrc(static_cast<const char*>(a.src) + nc * T * a.plane * GES, (unsigned)((long)T * plane * GES));
    __amdgpu_buffer_rsrc_t ry = cfn_rsrc(static_cast<const char*>(HASY ? a.src2 : a.src) + nc * T * a.plane * GES, (unsigned)((long)T * plane * GES));
    __amdgpu_buffer_rsrc_t rx = cfn_rsrc(static_cast<const float*>(a.yout) + nc * T * a.plane, (unsigned)((long)T * plane * 4));
    __amdgpu_buffer_rsrc_t rd = cfn_rsrc(static_cast<float*>(a.dst) + nc * T * a.plane, (unsigned)((long)T * plane * 4));
    float wk[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) wk[k] = cfn_uni(a.w[c * 5 + 4 - k]);        // flipped taps
    const float gsv = cfn_uni(a.gs ? (float)a.gs[nc] : 0.0f);
    const float gqv = cfn_uni((HASY && a.gq) ? 2.0f * (float)a.gq[nc] : 0.0f);
    auto ldg = [&](__amdgpu_buffer_rsrc_t r, int vo) -> f4v {
        if (BF) {
            const u2v u = __builtin_bit_cast(u2v, __builtin_amdgcn_raw_buffer_load_b64(r, vo, 0, 0));
            return (f4v){h16k_lo<BF ? BF : H16_BF16>(u.x), h16k_hi<BF ? BF : H16_BF16>(u.x), h16k_lo<BF ? BF : H16_BF16>(u.y), h16k_hi<BF ? BF : H16_BF16>(u.y)};
        }
        return __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(r, vo, 0, 0));
    };
    f4v G[TO + 4], Y[HASY ? TO + 4 : 1], X[TO];
#pragma unroll
    for (int k = 0; k < TO + 4; ++k) {
        const int t = t0 - 2 + k;
        const bool tv = ok && t >= 0 && t < T;
        const int vo = tv ? (t * plane + p4 * 4) * GES : OOB;
        G[k] = ldg(rg, vo);
        if (HASY) Y[k] = ldg(ry, vo);
    }
#pragma unroll
    for (int j = 0; j < TO; ++j) {
        const bool tv = ok && t0 + j < T;
        X[j] = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(rx, tv ? ((t0 + j) * plane + p4 * 4) * 4 : OOB, 0, 0));
    }
#pragma unroll
    for (int k = 0; k < TO + 4; ++k) {                                     // g' (zero outside the clip)
        const int t = t0 - 2 + k;
        const float m = (ok && t >= 0 && t < T) ? 1.0f : 0.0f;
        f4v v = G[k] + gsv;
        if (HASY) v += Y[k] * gqv;
        G[k] = v * m;
    }
    f4v acc[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) acc[k] = (f4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < TO; ++j) {
        const f4v gx = G[j] * wk[0] + G[j + 1] * wk[1] + G[j + 2] * wk[2] + G[j + 3] * wk[3] + G[j + 4] * wk[4];
        const bool em = ok && t0 + j < T;
        cfn_bst128(__builtin_bit_cast(u4v_t5, gx), rd, em ? ((t0 + j) * plane + p4 * 4) * 4 : OOB, 0);
        // x(s), s = t0 + j (zero beyond the clip: its load was switched off): g'(s + 2 - k) = G[j + 4 - k]
#pragma unroll
        for (int k = 0; k < 5; ++k) acc[k] += X[j] * G[j + 4 - k];
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const float v = cfn_wave_sum(acc[k].x + acc[k].y + acc[k].z + acc[k].w);
        if (lane == 0) sh[k * 4 + wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < 5)
        cfn_add64(&a.s1[c * 5 + threadIdx.x], (double)(sh[threadIdx.x * 4] + sh[threadIdx.x * 4 + 1] + sh[threadIdx.x * 4 + 2] + sh[threadIdx.x * 4 + 3]));
}

// -1 = not handled (the plane is not a whole number of float4s): the caller runs the two separate kernels
template <int BF>
static int t5_bwd_fused(const void* gy, const void* y, const double* gs, const double* gq, const float* w, const float* x, float* gx,
                        double* gw, int N, int C, int T, long plane, hipStream_t st) {
    static const int on = getenv("CFN_T5_FUSED") ? atoi(getenv("CFN_T5_FUSED")) : 1;
    if (!on || plane % 4 != 0 || (long)T * plane * 4 >= 0x7ffffff0L) return -1;
    if ((((uintptr_t)gy | (uintptr_t)x | (uintptr_t)gx | (uintptr_t)(y ? y : gy)) & 15) != 0) return -1;
    const bool hasy = y != nullptr && gq != nullptr;
    T5Args a = {};
    a.src = gy; a.src2 = hasy ? y : nullptr; a.gs = gs; a.gq = hasy ? gq : nullptr; a.w = w; a.yout = x; a.dst = gx; a.s1 = gw;
    a.C = C; a.T = T; a.plane = plane;
    const long NC = (long)N * C;
    static const int flat = getenv("CFN_T5_FLAT_BWD") ? atoi(getenv("CFN_T5_FLAT_BWD")) : 8;    // 0: marching kernel, 4 / 8: frames per thread
    if (flat == 4 || flat == 8) {
        const long per_nc = (long)cfn_cdiv(T, flat) * (plane / 4);
        const long bpn = cfn_cdiv(per_nc, 256L), blocks = bpn * NC;
        if (blocks < 0x7fffffffL && NC < 0x7fffffffL) {
            a.pchunks = (int)bpn;
#define T5_FB(TOV) do { if (hasy) hipLaunchKernelGGL((dwt5_bwd_flat_kernel<BF, true, TOV>), dim3((unsigned)blocks), dim3(256), 0, st, a); \
                        else hipLaunchKernelGGL((dwt5_bwd_flat_kernel<BF, false, TOV>), dim3((unsigned)blocks), dim3(256), 0, st, a); } while (0)
            if (flat == 8) T5_FB(8); else T5_FB(4);
#undef T5_FB
            return cfn_check_launch("dwconv_t5 flat backward");
        }
    }
    unsigned gy_, gz_;
    CFN_REQUIRE(cfn_split_nc(NC, gy_, gz_), "dwconv_t5: N*C = %ld exceeds grid.y", NC);
    a.pchunks = cfn_cdiv(plane, 1024L);
    int TT = 64;
    while (TT > 16 && NC * a.pchunks * cfn_cdiv(T, TT) < 2048) TT >>= 1;
    if (TT > T) TT = T;
    a.TT = TT;
    a.nchunks = cfn_cdiv(T, TT);
    dim3 grid((unsigned)(a.pchunks * a.nchunks), gy_, gz_);
    if (hasy) hipLaunchKernelGGL((dwt5_bwd_fused_kernel<BF, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((dwt5_bwd_fused_kernel<BF, false>), grid, dim3(256), 0, st, a);
    return cfn_check_launch("dwconv_t5 fused backward");
}

template <int MODE, int BF = 0>
static int t5_launch(T5Args& a, int N, hipStream_t st) {
    if (MODE == T5_FWD) {
        const int rc = t5_fwd_flat<BF>(a, N, st);
        if (rc != -1) return rc;
    }
    const long NC = (long)N * a.C;
    unsigned gy_, gz_;
    CFN_REQUIRE(cfn_split_nc(NC, gy_, gz_), "dwconv_t5: N*C = %ld exceeds grid.y", NC);
    const bool v4 = a.plane % 4 == 0;
    const int vec = v4 ? 4 : 1;
    a.pchunks = cfn_cdiv(a.plane, 256L * vec);
    int TT = 64;
    while (TT > 16 && NC * a.pchunks * cfn_cdiv(a.T, TT) < 2048) TT >>= 1;
    if (TT > a.T) TT = a.T;
    a.TT = TT;
    a.nchunks = cfn_cdiv(a.T, TT);
    dim3 grid((unsigned)(a.pchunks * a.nchunks), gy_, gz_);
    static const int stream_on = getenv("CFN_T5_STREAM") ? atoi(getenv("CFN_T5_STREAM")) : 1;
    if (MODE == T5_FWD && v4 && stream_on && (long)a.T * a.plane * 4 < 0x7ffffff0L) {
        hipLaunchKernelGGL((dwt5_fwd_stream_kernel<BF>), grid, dim3(256), 0, st, a);
        return cfn_check_launch("dwconv_t5");
    }
    if (v4) hipLaunchKernelGGL((dwt5_kernel<MODE, 4, BF>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((dwt5_kernel<MODE, 1, BF>), grid, dim3(256), 0, st, a);
    return cfn_check_launch("dwconv_t5");
}

extern "C" int cfn_dwconv_t5_fwd(const float* x, const float* w, float* y, double* sum, double* sumsq, int N, int C, int T,
                                 long plane, void* stream) {
    CFN_REQUIRE(x && w && y, "cfn_dwconv_t5_fwd: null tensor");
    CFN_REQUIRE((sum == nullptr) == (sumsq == nullptr), "cfn_dwconv_t5_fwd: sum/sumsq mismatch");
    T5Args a = {};
    a.src = x; a.w = w; a.dst = y; a.s1 = sum; a.s2 = sumsq; a.C = C; a.T = T; a.plane = plane;
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_DWCONV_FWD, st, 8.0 * N * C * T * plane);
    return t5_launch<T5_FWD>(a, N, st);
}

extern "C" int cfn_dwconv_t5_bwd_data(const float* gy, const float* y, const double* gsum, const double* gsumsq,
                                      const float* w, float* gx, int N, int C, int T, long plane, void* stream) {
    CFN_REQUIRE(gy && w && gx, "cfn_dwconv_t5_bwd_data: null tensor");
    CFN_REQUIRE(gsumsq == nullptr || y != nullptr, "cfn_dwconv_t5_bwd_data: gsumsq needs y");
    T5Args a = {};
    a.src = gy; a.src2 = gsumsq ? y : nullptr; a.gs = gsum; a.gq = gsumsq; a.w = w; a.dst = gx; a.C = C; a.T = T; a.plane = plane;
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_DWCONV_BWD, st, 4.0 * N * C * T * plane * (a.src2 ? 3 : 2));
    return t5_launch<T5_DGRAD>(a, N, st);
}

extern "C" int cfn_dwconv_t5_bwd_weight(const float* gy, const float* y, const double* gsum, const double* gsumsq,
                                        const float* x, double* gw, int N, int C, int T, long plane, void* stream) {
    CFN_REQUIRE(gy && x && gw, "cfn_dwconv_t5_bwd_weight: null tensor");
    CFN_REQUIRE(gsumsq == nullptr || y != nullptr, "cfn_dwconv_t5_bwd_weight: gsumsq needs y");
    T5Args a = {};
    a.src = x; a.gy = gy; a.yout = gsumsq ? y : nullptr; a.gs = gsum; a.gq = gsumsq; a.s1 = gw; a.C = C; a.T = T; a.plane = plane;
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_DWCONV_BWD, st, 4.0 * N * C * T * plane * (a.yout ? 3 : 2));
    return t5_launch<T5_WGRAD>(a, N, st);
}

// ---- bf16 activation path: y / gy bf16, x / gx fp32 (see T5Args) ----------------------------------------------------
extern "C" int cfn_dwconv_t5_fwd_bf16(const float* x, const float* w, unsigned short* y, double* sum, double* sumsq, int N, int C,
                                      int T, long plane, void* stream) {
    CFN_REQUIRE(x && w && y, "cfn_dwconv_t5_fwd_bf16: null tensor");
    CFN_REQUIRE((sum == nullptr) == (sumsq == nullptr), "cfn_dwconv_t5_fwd_bf16: sum/sumsq mismatch");
    T5Args a = {};
    a.src = x; a.w = w; a.dst = y; a.s1 = sum; a.s2 = sumsq; a.C = C; a.T = T; a.plane = plane;
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_DWCONV_FWD, st, 6.0 * N * C * T * plane);
    return t5_launch<T5_FWD, H16_BF16>(a, N, st);
}

extern "C" int cfn_dwconv_t5_bwd_data_bf16(const unsigned short* gy, const unsigned short* y, const double* gsum,
                                           const double* gsumsq, const float* w, float* gx, int N, int C, int T, long plane,
                                           void* stream) {
    CFN_REQUIRE(gy && w && gx, "cfn_dwconv_t5_bwd_data_bf16: null tensor");
    CFN_REQUIRE(gsumsq == nullptr || y != nullptr, "cfn_dwconv_t5_bwd_data_bf16: gsumsq needs y");
    T5Args a = {};
    a.src = gy; a.src2 = gsumsq ? y : nullptr; a.gs = gsum; a.gq = gsumsq; a.w = w; a.dst = gx; a.C = C; a.T = T; a.plane = plane;
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_DWCONV_BWD, st, N * (double)C * T * plane * (a.src2 ? 8.0 : 6.0));
    return t5_launch<T5_DGRAD, H16_BF16>(a, N, st);
}

extern "C" int cfn_dwconv_t5_bwd_weight_bf16(const unsigned short* gy, const unsigned short* y, const double* gsum,
                                             const double* gsumsq, const float* x, double* gw, int N, int C, int T, long plane,
                                             void* stream) {
    CFN_REQUIRE(gy && x && gw, "cfn_dwconv_t5_bwd_weight_bf16: null tensor");
    CFN_REQUIRE(gsumsq == nullptr || y != nullptr, "cfn_dwconv_t5_bwd_weight_bf16: gsumsq needs y");
    T5Args a = {};
    a.src = x; a.gy = gy; a.yout = gsumsq ? y : nullptr; a.gs = gsum; a.gq = gsumsq; a.s1 = gw; a.C = C; a.T = T; a.plane = plane;
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_DWCONV_BWD, st, N * (double)C * T * plane * (a.yout ? 8.0 : 6.0));
    return t5_launch<T5_WGRAD, H16_BF16>(a, N, st);
}

// data AND weight gradient in one pass (gy, y, x read once); -1 = not handled, call the two entry points above
extern "C" int cfn_dwconv_t5_bwd_fused(const float* gy, const float* y, const double* gsum, const double* gsumsq, const float* w,
                                       const float* x, float* gx, double* gw, int N, int C, int T, long plane, void* stream) {
    CFN_REQUIRE(gy && w && x && gx && gw, "cfn_dwconv_t5_bwd_fused: null tensor");
    CFN_REQUIRE(gsumsq == nullptr || y != nullptr, "cfn_dwconv_t5_bwd_fused: gsumsq needs y");
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_DWCONV_BWD, st, 4.0 * N * C * T * plane * (gsumsq ? 4 : 3));
    return t5_bwd_fused<0>(gy, gsumsq ? y : nullptr, gsum, gsumsq, w, x, gx, gw, N, C, T, plane, st);
}

extern "C" int cfn_dwconv_t5_bwd_fused_bf16(const unsigned short* gy, const unsigned short* y, const double* gsum, const double* gsumsq,
                                            const float* w, const float* x, float* gx, double* gw, int N, int C, int T, long plane,
                                            void* stream) {
    CFN_REQUIRE(gy && w && x && gx && gw, "cfn_dwconv_t5_bwd_fused_bf16: null tensor");
    CFN_REQUIRE(gsumsq == nullptr || y != nullptr, "cfn_dwconv_t5_bwd_fused_bf16: gsumsq needs y");
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_DWCONV_BWD, st, N * (double)C * T * plane * (gsumsq ? 12.0 : 10.0));
    return t5_bwd_fused<H16_BF16>(gy, gsumsq ? y : nullptr, gsum, gsumsq, w, x, gx, gw, N, C, T, plane, st);
}

// ---- fp16 activation path (BASELINE configs[4]): the same kernels with IEEE-half elements on the output side (h16.h) ------------
extern "C" int cfn_dwconv_t5_fwd_f16(const float* x, const float* w, unsigned short* y, double* sum, double* sumsq, int N, int C,
                                      int T, long plane, void* stream) {
    CFN_REQUIRE(x && w && y, "cfn_dwconv_t5_fwd_f16: null tensor");
    CFN_REQUIRE((sum == nullptr) == (sumsq == nullptr), "cfn_dwconv_t5_fwd_f16: sum/sumsq mismatch");
    T5Args a = {};
    a.src = x; a.w = w; a.dst = y; a.s1 = sum; a.s2 = sumsq; a.C = C; a.T = T; a.plane = plane;
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_DWCONV_FWD, st, 6.0 * N * C * T * plane);
    return t5_launch<T5_FWD, H16_F16>(a, N, st);
}

extern "C" int cfn_dwconv_t5_bwd_data_f16(const unsigned short* gy, const unsigned short* y, const double* gsum,
                                           const double* gsumsq, const float* w, float* gx, int N, int C, int T, long plane,
                                           void* stream) {
    CFN_REQUIRE(gy && w && gx, "cfn_dwconv_t5_bwd_data_f16: null tensor");
    CFN_REQUIRE(gsumsq == nullptr || y != nullptr, "cfn_dwconv_t5_bwd_data_f16: gsumsq needs y");
    T5Args a = {};
    a.src = gy; a.src2 = gsumsq ? y : nullptr; a.gs = gsum; a.gq = gsumsq; a.w = w; a.dst = gx; a.C = C; a.T = T; a.plane = plane;
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_DWCONV_BWD, st, N * (double)C * T * plane * (a.src2 ? 8.0 : 6.0));
    return t5_launch<T5_DGRAD, H16_F16>(a, N, st);
}

extern "C" int cfn_dwconv_t5_bwd_weight_f16(const unsigned short* gy, const unsigned short* y, const double* gsum,
                                             const double* gsumsq, const float* x, double* gw, int N, int C, int T, long plane,
                                             void* stream) {
    CFN_REQUIRE(gy && x && gw, "cfn_dwconv_t5_bwd_weight_f16: null tensor");
    CFN_REQUIRE(gsumsq == nullptr || y != nullptr, "cfn_dwconv_t5_bwd_weight_f16: gsumsq needs y");
    T5Args a = {};
    a.src = x; a.gy = gy; a.yout = gsumsq ? y : nullptr; a.gs = gsum; a.gq = gsumsq; a.s1 = gw; a.C = C; a.T = T; a.plane = plane;
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_DWCONV_BWD, st, N * (double)C * T * plane * (a.yout ? 8.0 : 6.0));
    return t5_launch<T5_WGRAD, H16_F16>(a, N, st);
}

extern "C" int cfn_dwconv_t5_bwd_fused_f16(const unsigned short* gy, const unsigned short* y, const double* gsum, const double* gsumsq,
                                            const float* w, const float* x, float* gx, double* gw, int N, int C, int T, long plane,
                                            void* stream) {
    CFN_REQUIRE(gy && w && x && gx && gw, "cfn_dwconv_t5_bwd_fused_f16: null tensor");
    CFN_REQUIRE(gsumsq == nullptr || y != nullptr, "cfn_dwconv_t5_bwd_fused_f16: gsumsq needs y");
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_DWCONV_BWD, st, N * (double)C * T * plane * (gsumsq ? 12.0 : 10.0));
    return t5_bwd_fused<H16_F16>(gy, gsumsq ? y : nullptr, gsum, gsumsq, w, x, gx, gw, N, C, T, plane, st);
}

