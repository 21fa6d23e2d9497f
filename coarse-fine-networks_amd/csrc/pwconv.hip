// Pointwise (1x1x1) channel contractions of X3D (conv1/conv3/downsample/conv5/fc1:
// x3d_fine.py:100-105, :115, :119, :245-250, :256, :286) on the CDNA4 matrix cores.
//
// NCDHW keeps positions contiguous, so per sample the op is  Y[M x Q] = Wm[M x K] * X[K x Q].
// fp32 operands use v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, k-ordered):
//   A operand (weights, tiny, shared by the 4 waves)  -> LDS image As[k][m], conflict-free rows
//   B operand (activations, streamed once from HBM)    -> straight global->VGPR: lane (j=l&31,
//     kk=l>>5) reads X[k0+kk][q0+j], i.e. two fully used 128-byte lines per instruction, gets the
//     load-time prologue in registers and feeds the MFMA without touching LDS.
//   C: lane l holds position q0+(l&31) for 16 rows -> every store instruction writes 2 full lines.
// Modes:
//   PW_FWD    y = W * act(A x + B)            (+ per-(n,m) sum / sum-of-squares of y)
//   PW_DGRAD  da = W^T * (gy + gs + 2 y gq);  gx = da * act'(A x + B) * A  (+ sum(dz x), sum(dz))
//   pw_wgrad_kernel   gW[m][k] += (gy + gs + 2 y gq)[m,:] . act(A x + B)[k,:]   (both through LDS)
// Spatial stride 2 (shortcut conv, x3d_fine.py:284-287) is a position map on the strided side.
#include "cfn_common.h"

typedef float __attribute__((ext_vector_type(16))) f16v;
typedef float __attribute__((ext_vector_type(4))) f4v;

enum { PW_FWD = 0, PW_DGRAD = 1 };
#define PW_KC 32
#define PW_RED_PITCH 33

struct PwArgs {
    const float* src;    // FWD: x raw (N,K,Pin)          DGRAD: gy (N,K,Q)
    const float* src2;   // DGRAD: y raw (N,K,Q) for the 2*y*gq term (may be null)
    const float* pa;     // FWD: prologue A[n,k] (null = identity)
    const float* pb;
    const double* gs;    // DGRAD: d/d sum(y)   [n,k]  (may be null)
    const double* gq;    // DGRAD: d/d sum(y^2) [n,k]  (may be null)
    const float* w;      // (Cout, Cin) row major
    float* dst;          // FWD: y (N,M,Q)                DGRAD: gx (N,M,Pin)
    const float* ex;     // DGRAD: forward input x raw (N,M,Pin) (needed when ea != null)
    const float* ea;     // DGRAD: forward prologue A[n,m] (null = identity => gx = da)
    const float* eb;
    double* s1;          // FWD: sum(y) [n,m]             DGRAD: sum(dz*x) [n,m]
    double* s2;          // FWD: sum(y^2)                 DGRAD: sum(dz)
    int N, M, K, Q, Pin, Hi, Wi, Ho, Wo, stride, act;
    int Cin;             // row pitch of w
    int mtiles, nstrips, tpb, resident, Kpad;
    int stem, Cimg;      // stem != 0: B operand is the im2col view of a (N,Cimg,T,Hi,Wi) clip for a 1x3x3 stride-2 pad-1 conv
};

__device__ __forceinline__ int pw_pmap(int q, int Ho, int Wo, int Hi, int Wi, int stride) {
    if (stride == 1) return q;
    const int hw = Ho * Wo;
    const int t = q / hw, r = q - t * hw;
    const int oh = r / Wo, ow = r - oh * Wo;
    return (t * Hi + oh * stride) * Wi + ow * stride;
}

template <int MT, int MODE, bool STATS>
__global__ __launch_bounds__(256) void pw_gemm_kernel(const PwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BM = 32 * MT;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, col = lane & 31;
    const int K = a.K, M = a.M, Q = a.Q, Kpad = a.Kpad;

    unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    const int mtile = L % a.mtiles; L /= a.mtiles;
    const int strip = L % a.nstrips;
    const int n = L / a.nstrips;
    const int m0 = mtile * BM;

    const int as_rows = a.resident ? Kpad : PW_KC;
    float* As = smem;                         // [as_rows][BM]
    float* sPA = As + as_rows * BM;           // [Kpad]
    float* sPB = sPA + Kpad;                  // [Kpad]
    float* sEA = sPB + Kpad;                  // [BM]
    float* sEB = sEA + BM;                    // [BM]
    float* sSt = sEB + BM;                    // [BM][2]
    float* red = sSt + 2 * BM + wave * (32 * PW_RED_PITCH);   // per wave [32][33]

    for (int k = tid; k < Kpad; k += 256) {
        float va, vb;
        if (MODE == PW_FWD) {
            va = (k < K && a.pa) ? a.pa[(long)n * K + k] : 1.0f;
            vb = (k < K && a.pb) ? a.pb[(long)n * K + k] : 0.0f;
        } else {
            va = (k < K && a.gs) ? (float)a.gs[(long)n * K + k] : 0.0f;
            vb = (k < K && a.gq && a.src2) ? 2.0f * (float)a.gq[(long)n * K + k] : 0.0f;
        }
        sPA[k] = va; sPB[k] = vb;
    }
    for (int m = tid; m < BM; m += 256) {
        const bool ok = (m0 + m) < M && MODE == PW_DGRAD && a.ea;
        sEA[m] = ok ? a.ea[(long)n * M + m0 + m] : 1.0f;
        sEB[m] = ok ? a.eb[(long)n * M + m0 + m] : 0.0f;
        sSt[2 * m] = 0.0f; sSt[2 * m + 1] = 0.0f;
    }
    auto load_w = [&](int kbase, int rows) {   // As[kk][m] = Wm[m0+m][kbase+kk], zero padded
        for (int e = tid; e < rows * BM; e += 256) {
            const int kk = e / BM, m = e - kk * BM;
            const int k = kbase + kk, mm = m0 + m;
            float v = 0.0f;
            if (k < K && mm < M) v = (MODE == PW_FWD) ? a.w[(long)mm * a.Cin + k] : a.w[(long)k * a.Cin + mm];
            As[e] = v;
        }
    };
    if (a.resident) load_w(0, Kpad);
    __syncthreads();

    const bool two_src = MODE == PW_DGRAD && a.src2 != nullptr;
    const long src_n = a.stem ? (long)n * a.Cimg * a.Pin : (long)n * K * (MODE == PW_FWD ? a.Pin : Q);
    const long dst_n = (long)n * M * (MODE == PW_FWD ? Q : a.Pin);
    const int src_pitch = MODE == PW_FWD ? a.Pin : Q;
    const int dst_pitch = MODE == PW_FWD ? Q : a.Pin;
    float sacc[MT], qacc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) { sacc[i] = 0.0f; qacc[i] = 0.0f; }

    for (int tile = 0; tile < a.tpb; ++tile) {
        const int qt = (strip * a.tpb + tile) * 128;
        if (qt >= Q) break;
        const int q = qt + wave * 32 + col;
        const bool valid = q < Q;
        const int qc = valid ? q : Q - 1;
        const int pm = pw_pmap(qc, a.Ho, a.Wo, a.Hi, a.Wi, a.stride);
        const int in_pos = MODE == PW_FWD ? pm : qc;
        const int out_pos = MODE == PW_FWD ? qc : pm;
        const float* bp = a.src + src_n + in_pos;
        const float* bp2 = two_src ? a.src2 + src_n + in_pos : nullptr;

        f16v acc[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

        for (int kc = 0; kc < Kpad; kc += PW_KC) {
            const int kcount = min(PW_KC, Kpad - kc);   // even
            if (!a.resident) {
                __syncthreads();
                load_w(kc, PW_KC);
                __syncthreads();
            }
            float bv[PW_KC / 2], bv2[PW_KC / 2];
            if (MODE == PW_FWD && a.stem) {   // im2col gather: k -> (ci, kh, kw), zero outside the image
                const int hw = a.Ho * a.Wo;
                const int tq = qc / hw, rq = qc - tq * hw;
                const int oh = rq / a.Wo, ow = rq - oh * a.Wo;
                const float* sp = a.src + src_n + (long)tq * a.Hi * a.Wi;
#pragma unroll
                for (int s = 0; s < PW_KC / 2; ++s) {
                    const int k = kc + 2 * s + half;
                    if (2 * s < kcount) {
                        const int ci = k / 9, kr = k - ci * 9, kh = kr / 3, kw = kr - kh * 3;
                        const int ih = oh * 2 + kh - 1, iw = ow * 2 + kw - 1;
                        const bool inb = k < K && ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi;
                        bv[s] = inb ? sp[(long)ci * a.Pin + (long)ih * a.Wi + iw] : 0.0f;
                    }
                }
            } else {
#pragma unroll
                for (int s = 0; s < PW_KC / 2; ++s) {
                    const int k = kc + 2 * s + half;
                    const int kq = k < K ? k : K - 1;
                    if (2 * s < kcount) {
                        bv[s] = bp[(long)kq * src_pitch];
                        if (MODE == PW_DGRAD && two_src) bv2[s] = bp2[(long)kq * src_pitch];
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < PW_KC / 2; ++s) {
                if (2 * s < kcount) {
                    const int k = kc + 2 * s + half;
                    float v = bv[s];
                    const float ca = sPA[k], cb = sPB[k];   // k < Kpad always
                    if (MODE == PW_FWD) v = cfn_act_rt(fmaf(v, ca, cb), a.act);
                    else v = two_src ? fmaf(bv2[s], cb, v + ca) : v + ca;
                    if (!valid || k >= K) v = 0.0f;
                    const float* ar = As + ((a.resident ? kc : 0) + 2 * s + half) * BM + col;
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[i * 32], v, acc[i], 0, 0, 0);
                }
            }
        }

        // ---- epilogue ---------------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float t1[16], t2[16];
            float xe[16];
            if (MODE == PW_DGRAD && a.ea) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const int mm = min(m0 + ml, M - 1);
                    xe[r] = a.ex[dst_n + (long)mm * dst_pitch + out_pos];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int mm = m0 + ml;
                float v = acc[i][r];
                if (MODE == PW_FWD) {
                    t1[r] = v;
                } else if (a.ea) {
                    const float ca = sEA[ml], cb = sEB[ml];
                    const float dz = (valid && mm < M) ? v * cfn_act_grad_rt(fmaf(xe[r], ca, cb), a.act) : 0.0f;
                    t1[r] = dz * xe[r];
                    t2[r] = dz;
                    v = dz * ca;
                }
                if (valid && mm < M) a.dst[dst_n + (long)mm * dst_pitch + out_pos] = v;
            }
            if (STATS) {
                // wave-private transpose through LDS: lane -> (row = lane&31, 16 of the 32 columns)
#pragma unroll
                for (int pass = 0; pass < (MODE == PW_FWD ? 1 : 2); ++pass) {
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        red[((r & 3) + 8 * (r >> 2) + 4 * half) * PW_RED_PITCH + col] = pass == 0 ? t1[r] : t2[r];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    float s = 0.0f, qq = 0.0f;
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const float v = red[col * PW_RED_PITCH + half * 16 + j];
                        s += v;
                        qq = fmaf(v, v, qq);
                    }
                    if (MODE == PW_FWD) { sacc[i] += s; qacc[i] += qq; }
                    else if (pass == 0) sacc[i] += s;
                    else qacc[i] += s;
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            }
        }
    }

    if (STATS) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const float s = sacc[i] + __shfl_xor(sacc[i], 32, 64);
            const float qq = qacc[i] + __shfl_xor(qacc[i], 32, 64);
            if (half == 0) { atomicAdd(&sSt[2 * (i * 32 + col)], s); atomicAdd(&sSt[2 * (i * 32 + col) + 1], qq); }
        }
        __syncthreads();
        for (int m = tid; m < BM; m += 256) {
            if (m0 + m < M) {
                atomicAdd(&a.s1[(long)n * M + m0 + m], (double)sSt[2 * m]);
                atomicAdd(&a.s2[(long)n * M + m0 + m], (double)sSt[2 * m + 1]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// weight gradient: gW[m][k] = sum_{n,q} G[n,m,q] * Aop[n,k,p(q)]
//   block = (row tile of <=96 output channels) x (col tile of <=64 input channels) x one strip
//   of positions of one sample; positions are staged 64 at a time in LDS as [channel][65], the
//   4 waves split each stage's positions (in-block split-K), partials are combined through LDS
//   and leave the block as one fp64 atomic per element.
// ---------------------------------------------------------------------------------------------
#define WG_PT 64
#define WG_PITCH 65

struct WgArgs {
    const float* gy;     // (N,M,Q)
    const float* y;      // (N,M,Q) raw conv output, for the gq term (may be null)
    const double* gs; const double* gq;   // [n,m] (may be null)
    const float* x;      // (N,K,Pin) forward input raw
    const float* pa; const float* pb;     // forward prologue [n,k] (null = identity)
    double* gw;          // (M,K) fp64 accumulators (row pitch Kc)
    int N, M, K, Q, Pin, Hi, Wi, Ho, Wo, stride, act;
    int mtiles, ktiles, nstrips, stages;   // stages = LDS stages (of 64 positions) per block
    int stem, Cimg;      // stem != 0: x rows are the im2col view (k -> ci,kh,kw) of a (N,Cimg,T,Hi,Wi) clip
};

template <int MTW, int NTW>
__global__ __launch_bounds__(256) void pw_wgrad_kernel(const WgArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BM = 32 * MTW, BN = 32 * NTW;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, col = lane & 31;
    unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    const int mt = L % a.mtiles; L /= a.mtiles;
    const int kt = L % a.ktiles; L /= a.ktiles;
    const int strip = L % a.nstrips;
    const int n = L / a.nstrips;
    const int m0 = mt * BM, k0 = kt * BN;
    const int M = a.M, K = a.K, Q = a.Q;

    float* sG = smem;                       // [BM][65]
    float* sX = sG + BM * WG_PITCH;         // [BN][65]
    float* sCg = sX + BN * WG_PITCH;        // [BM][2]  (gs, 2gq)
    float* sCx = sCg + 2 * BM;              // [BN][2]  (A, B)
    for (int m = tid; m < BM; m += 256) {
        const bool ok = m0 + m < M;
        sCg[2 * m] = (ok && a.gs) ? (float)a.gs[(long)n * M + m0 + m] : 0.0f;
        sCg[2 * m + 1] = (ok && a.gq && a.y) ? 2.0f * (float)a.gq[(long)n * M + m0 + m] : 0.0f;
    }
    for (int k = tid; k < BN; k += 256) {
        const bool ok = k0 + k < K && a.pa;
        sCx[2 * k] = ok ? a.pa[(long)n * K + k0 + k] : 1.0f;
        sCx[2 * k + 1] = ok ? a.pb[(long)n * K + k0 + k] : 0.0f;
    }
    f16v acc[MTW][NTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    __syncthreads();

    const bool vec_ok = (Q % 4 == 0) && a.stride == 1;
    const bool xvec_ok = vec_ok && !a.stem;
    for (int st = 0; st < a.stages; ++st) {
        const int q0 = (strip * a.stages + st) * WG_PT;
        if (q0 >= Q) break;
        if (st) __syncthreads();
        // ---- stage G rows (BM x 64) and X rows (BN x 64) with their prologues -----------------
        for (int e = tid; e < (BM + BN) * (WG_PT / 4); e += 256) {
            const int row = e / (WG_PT / 4), c4 = (e - row * (WG_PT / 4)) * 4;
            const bool isg = row < BM;
            const int ch = isg ? m0 + row : k0 + row - BM;
            const bool chok = isg ? ch < M : ch < K;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (chok) {
                if (isg) {
                    const long base = ((long)n * M + ch) * Q + q0 + c4;
                    const float cs = sCg[2 * row], cq = sCg[2 * row + 1];
                    if (vec_ok && q0 + c4 + 3 < Q) {
                        const f4v g = *reinterpret_cast<const f4v*>(a.gy + base);
                        f4v yy = {0.f, 0.f, 0.f, 0.f};
                        if (a.y) yy = *reinterpret_cast<const f4v*>(a.y + base);
                        v[0] = fmaf(yy.x, cq, g.x + cs); v[1] = fmaf(yy.y, cq, g.y + cs);
                        v[2] = fmaf(yy.z, cq, g.z + cs); v[3] = fmaf(yy.w, cq, g.w + cs);
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (q0 + c4 + u < Q) v[u] = fmaf(a.y ? a.y[base + u] : 0.0f, cq, a.gy[base + u] + cs);
                    }
                } else {
                    const int kr = row - BM;
                    const float ca = sCx[2 * kr], cb = sCx[2 * kr + 1];
                    const long base = ((long)n * K + ch) * a.Pin;
                    if (a.stem) {
                        const int ci = ch / 9, kr9 = ch - ci * 9, kh = kr9 / 3, kw = kr9 - kh * 3;
                        const int hw = a.Ho * a.Wo;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int qq = q0 + c4 + u;
                            if (qq < Q) {
                                const int tq = qq / hw, rq = qq - tq * hw;
                                const int oh = rq / a.Wo, ow = rq - oh * a.Wo;
                                const int ih = oh * 2 + kh - 1, iw = ow * 2 + kw - 1;
                                if (ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi)
                                    v[u] = a.x[((long)n * a.Cimg + ci) * a.Pin + ((long)tq * a.Hi + ih) * a.Wi + iw];
                            }
                        }
                    } else if (xvec_ok && q0 + c4 + 3 < Q) {
                        const f4v xx = *reinterpret_cast<const f4v*>(a.x + base + q0 + c4);
                        v[0] = cfn_act_rt(fmaf(xx.x, ca, cb), a.act); v[1] = cfn_act_rt(fmaf(xx.y, ca, cb), a.act);
                        v[2] = cfn_act_rt(fmaf(xx.z, ca, cb), a.act); v[3] = cfn_act_rt(fmaf(xx.w, ca, cb), a.act);
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (q0 + c4 + u < Q)
                                v[u] = cfn_act_rt(fmaf(a.x[base + pw_pmap(q0 + c4 + u, a.Ho, a.Wo, a.Hi, a.Wi, a.stride)], ca, cb), a.act);
                    }
                }
            }
            float* d = (isg ? sG + row * WG_PITCH : sX + (row - BM) * WG_PITCH) + c4;
            d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        }
        __syncthreads();
        // ---- each wave contracts its 16 positions of the stage ---------------------------------
        const int pbase = wave * (WG_PT / 4);
#pragma unroll
        for (int s = 0; s < WG_PT / 8; ++s) {
            const int p = pbase + 2 * s + half;
            float av[MTW], bv[NTW];
#pragma unroll
            for (int i = 0; i < MTW; ++i) av[i] = sG[(i * 32 + col) * WG_PITCH + p];
#pragma unroll
            for (int j = 0; j < NTW; ++j) bv[j] = sX[(j * 32 + col) * WG_PITCH + p];
#pragma unroll
            for (int i = 0; i < MTW; ++i)
#pragma unroll
                for (int j = 0; j < NTW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    }
    // ---- combine the 4 waves through LDS, then fp64 atomics ---------------------------------------
    __syncthreads();
    float* cw = smem;   // [BM][BN+1] reuse (BM*(BN+1) <= (BM+BN)*65 holds for BM<=96, BN<=64)
    for (int e = tid; e < BM * (BN + 1); e += 256) cw[e] = 0.0f;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                atomicAdd(&cw[ml * (BN + 1) + j * 32 + col], acc[i][j][r]);
            }
    __syncthreads();
    for (int e = tid; e < BM * BN; e += 256) {
        const int ml = e / BN, kl = e - ml * BN;
        if (m0 + ml < M && k0 + kl < K) atomicAdd(&a.gw[(long)(m0 + ml) * K + k0 + kl], (double)cw[ml * (BN + 1) + kl]);
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
template <int MODE, bool STATS>
static int pw_launch(const PwArgs& a, int MT, unsigned blocks, size_t lds, hipStream_t st) {
#define CFN_PW_GO(MTV)                                                                                         \
    do {                                                                                                       \
        auto k = pw_gemm_kernel<MTV, MODE, STATS>;                                                             \
        if (lds > 48 * 1024) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, st, a);                                            \
    } while (0)
    switch (MT) {
        case 1: CFN_PW_GO(1); break;
        case 2: CFN_PW_GO(2); break;
        case 3: CFN_PW_GO(3); break;
        default: CFN_PW_GO(4); break;
    }
#undef CFN_PW_GO
    return cfn_check_launch("pwconv");
}

static int pw_plan(PwArgs& a, int& MT, unsigned& blocks, size_t& lds) {
    const int M32 = cfn_cdiv(a.M, 32);
    // fewest row tiles of <=128 rows, then the smallest tile that covers M with that count
    const int ntile = cfn_cdiv(M32, 4);
    MT = cfn_cdiv(M32, ntile);
    a.mtiles = cfn_cdiv(a.M, 32 * MT);
    a.Kpad = (a.K + 1) & ~1;
    const int BM = 32 * MT;
    a.resident = ((size_t)a.Kpad * BM * 4 <= 40 * 1024) ? 1 : 0;
    const long tiles = cfn_cdiv(a.Q, 128);
    int tpb = 1;
    while (tpb < 16 && (long)a.N * cfn_cdiv(tiles, tpb * 2) * a.mtiles >= 2048) tpb *= 2;
    a.tpb = tpb;
    a.nstrips = cfn_cdiv(tiles, tpb);
    blocks = (unsigned)((long)a.N * a.nstrips * a.mtiles);
    lds = ((size_t)(a.resident ? a.Kpad : PW_KC) * BM + 2 * a.Kpad + 4 * BM + 4 * 32 * PW_RED_PITCH) * sizeof(float);
    return CFN_OK;
}

static void pw_geom(PwArgs& a, int T, int Hi, int Wi, int stride) {
    a.Hi = Hi; a.Wi = Wi; a.stride = stride;
    a.Ho = (Hi - 1) / stride + 1;
    a.Wo = (Wi - 1) / stride + 1;
    a.Pin = T * Hi * Wi;
    a.Q = T * a.Ho * a.Wo;
}

extern "C" int cfn_pwconv_fwd(const float* x, const float* A, const float* B, int act, const float* w, float* y,
                              double* sum, double* sumsq, int N, int Cin, int Cout, int T, int Hi, int Wi, int stride,
                              void* stream) {
    CFN_REQUIRE(x && w && y, "cfn_pwconv_fwd: null tensor");
    CFN_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && T > 0 && Hi > 0 && Wi > 0, "cfn_pwconv_fwd: bad shape");
    CFN_REQUIRE(stride == 1 || stride == 2, "cfn_pwconv_fwd: stride must be 1 or 2");
    CFN_REQUIRE((A == nullptr) == (B == nullptr), "cfn_pwconv_fwd: A/B mismatch");
    CFN_REQUIRE((sum == nullptr) == (sumsq == nullptr), "cfn_pwconv_fwd: sum/sumsq mismatch");
    PwArgs a = {};
    a.src = x; a.pa = A; a.pb = B; a.act = act; a.w = w; a.dst = y; a.s1 = sum; a.s2 = sumsq;
    a.N = N; a.M = Cout; a.K = Cin; a.Cin = Cin;
    pw_geom(a, T, Hi, Wi, stride);
    CFN_REQUIRE((long)T * Hi * Wi < (1L << 31), "cfn_pwconv_fwd: per-sample volume too large");
    int MT; unsigned blocks; size_t lds;
    pw_plan(a, MT, blocks, lds);
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_PWCONV_FWD, st, 4.0 * N * ((double)Cin * a.Q + (double)Cout * a.Q) + 4.0 * Cin * Cout);
    return sum ? pw_launch<PW_FWD, true>(a, MT, blocks, lds, st) : pw_launch<PW_FWD, false>(a, MT, blocks, lds, st);
}

// gx must be zero-filled by the caller when stride == 2 (only the strided positions are written)
extern "C" int cfn_pwconv_bwd_data(const float* gy, const float* y, const double* gsum, const double* gsumsq,
                                   const float* w, const float* x, const float* A, const float* B, int act, float* gx,
                                   double* gA, double* gB, int N, int Cin, int Cout, int T, int Hi, int Wi, int stride,
                                   void* stream) {
    CFN_REQUIRE(gy && w && gx, "cfn_pwconv_bwd_data: null tensor");
    CFN_REQUIRE(stride == 1 || stride == 2, "cfn_pwconv_bwd_data: stride must be 1 or 2");
    CFN_REQUIRE((A == nullptr) == (B == nullptr), "cfn_pwconv_bwd_data: A/B mismatch");
    CFN_REQUIRE(A == nullptr || (x && gA && gB), "cfn_pwconv_bwd_data: prologue needs x, gA, gB");
    CFN_REQUIRE(gsumsq == nullptr || y != nullptr, "cfn_pwconv_bwd_data: gsumsq needs y");
    PwArgs a = {};
    a.src = gy; a.src2 = gsumsq ? y : nullptr; a.gs = gsum; a.gq = gsumsq; a.w = w; a.dst = gx;
    a.ex = x; a.ea = A; a.eb = B; a.act = act; a.s1 = gA; a.s2 = gB;
    a.N = N; a.M = Cin; a.K = Cout; a.Cin = Cin;
    pw_geom(a, T, Hi, Wi, stride);
    int MT; unsigned blocks; size_t lds;
    pw_plan(a, MT, blocks, lds);
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_PWCONV_BWD, st, 4.0 * N * ((double)Cout * a.Q * (a.src2 ? 2 : 1) + (double)Cin * a.Q * (A ? 2 : 1)));
    return A ? pw_launch<PW_DGRAD, true>(a, MT, blocks, lds, st) : pw_launch<PW_DGRAD, false>(a, MT, blocks, lds, st);
}

extern "C" int cfn_pwconv_bwd_weight(const float* gy, const float* y, const double* gsum, const double* gsumsq,
                                     const float* x, const float* A, const float* B, int act, double* gw, int N,
                                     int Cin, int Cout, int T, int Hi, int Wi, int stride, void* stream) {
    CFN_REQUIRE(gy && x && gw, "cfn_pwconv_bwd_weight: null tensor");
    CFN_REQUIRE(stride == 1 || stride == 2, "cfn_pwconv_bwd_weight: stride must be 1 or 2");
    CFN_REQUIRE((A == nullptr) == (B == nullptr), "cfn_pwconv_bwd_weight: A/B mismatch");
    CFN_REQUIRE(gsumsq == nullptr || y != nullptr, "cfn_pwconv_bwd_weight: gsumsq needs y");
    WgArgs a = {};
    a.gy = gy; a.y = gsumsq ? y : nullptr; a.gs = gsum; a.gq = gsumsq; a.x = x; a.pa = A; a.pb = B; a.act = act;
    a.gw = gw; a.N = N; a.M = Cout; a.K = Cin;
    a.Hi = Hi; a.Wi = Wi; a.stride = stride;
    a.Ho = (Hi - 1) / stride + 1; a.Wo = (Wi - 1) / stride + 1;
    a.Pin = T * Hi * Wi; a.Q = T * a.Ho * a.Wo;
    // tile shape: rows <= 96, cols <= 64
    const int M32 = cfn_cdiv(Cout, 32), K32 = cfn_cdiv(Cin, 32);
    const int MTW = cfn_cdiv(M32, cfn_cdiv(M32, 3)), NTW = cfn_cdiv(K32, cfn_cdiv(K32, 2));
    a.mtiles = cfn_cdiv(Cout, 32 * MTW);
    a.ktiles = cfn_cdiv(Cin, 32 * NTW);
    const long nst = cfn_cdiv(a.Q, WG_PT);
    int stages = 64;
    while (stages > 4 && (long)N * cfn_cdiv(nst, stages) * a.mtiles * a.ktiles < 1024) stages >>= 1;
    a.stages = stages;
    a.nstrips = cfn_cdiv(nst, stages);
    const unsigned blocks = (unsigned)((long)N * a.nstrips * a.mtiles * a.ktiles);
    const size_t lds = ((size_t)(32 * MTW + 32 * NTW) * WG_PITCH + 2 * (32 * MTW + 32 * NTW)) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_PWCONV_BWD, st, 4.0 * N * ((double)Cout * a.Q * (a.y ? 2 : 1) + (double)Cin * a.Q));
#define CFN_WG_GO(MW, NW)                                                                                       \
    do {                                                                                                        \
        auto k = pw_wgrad_kernel<MW, NW>;                                                                       \
        if (lds > 48 * 1024) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, st, a);                                             \
    } while (0)
    if (MTW == 1 && NTW == 1) CFN_WG_GO(1, 1);
    else if (MTW == 1) CFN_WG_GO(1, 2);
    else if (MTW == 2 && NTW == 1) CFN_WG_GO(2, 1);
    else if (MTW == 2) CFN_WG_GO(2, 2);
    else if (NTW == 1) CFN_WG_GO(3, 1);
    else CFN_WG_GO(3, 2);
#undef CFN_WG_GO
    return cfn_check_launch("pwconv_bwd_weight");
}


// ---------------------------------------------------------------------------------------------
// X3D stem spatial conv (conv1_s: 1x3x3, stride (1,2,2), pad (0,1,1), x3d_fine.py:210-215) as the
// same MFMA contraction over the im2col view of the clip: K = Cimg*9, no im2col buffer in HBM.
// ---------------------------------------------------------------------------------------------
extern "C" int cfn_stem_conv_fwd(const float* x, const float* w, float* y, int N, int Cimg, int Cout, int T, int Hi, int Wi,
                                 void* stream) {
    CFN_REQUIRE(x && w && y, "cfn_stem_conv_fwd: null tensor");
    CFN_REQUIRE(N > 0 && Cimg > 0 && Cout > 0 && T > 0 && Hi > 1 && Wi > 1, "cfn_stem_conv_fwd: bad shape");
    PwArgs a = {};
    a.src = x; a.w = w; a.dst = y; a.act = CFN_ACT_NONE;
    a.N = N; a.M = Cout; a.K = Cimg * 9; a.Cin = Cimg * 9;
    a.stem = 1; a.Cimg = Cimg;
    a.Hi = Hi; a.Wi = Wi; a.stride = 1;
    a.Ho = (Hi + 2 - 3) / 2 + 1; a.Wo = (Wi + 2 - 3) / 2 + 1;
    a.Pin = T * Hi * Wi; a.Q = T * a.Ho * a.Wo;
    int MT; unsigned blocks; size_t lds;
    pw_plan(a, MT, blocks, lds);
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_STEM, st, 4.0 * N * ((double)Cimg * a.Pin + (double)Cout * a.Q));
    return pw_launch<PW_FWD, false>(a, MT, blocks, lds, st);
}

extern "C" int cfn_stem_conv_bwd_weight(const float* gy, const float* x, double* gw, int N, int Cimg, int Cout, int T,
                                        int Hi, int Wi, void* stream) {
    CFN_REQUIRE(gy && x && gw, "cfn_stem_conv_bwd_weight: null tensor");
    WgArgs a = {};
    a.gy = gy; a.x = x; a.act = CFN_ACT_NONE; a.gw = gw; a.N = N; a.M = Cout; a.K = Cimg * 9;
    a.stem = 1; a.Cimg = Cimg;
    a.Hi = Hi; a.Wi = Wi; a.stride = 1;
    a.Ho = (Hi + 2 - 3) / 2 + 1; a.Wo = (Wi + 2 - 3) / 2 + 1;
    a.Pin = T * Hi * Wi; a.Q = T * a.Ho * a.Wo;
    const int M32 = cfn_cdiv(Cout, 32), K32 = cfn_cdiv(a.K, 32);
    CFN_REQUIRE(M32 <= 3 && K32 <= 2, "cfn_stem_conv_bwd_weight: Cout <= 96 and Cimg*9 <= 64 supported (got %d, %d)", Cout, a.K);
    a.mtiles = 1; a.ktiles = 1;
    const long nst = cfn_cdiv(a.Q, WG_PT);
    int stages = 64;
    while (stages > 4 && (long)N * cfn_cdiv(nst, stages) < 1024) stages >>= 1;
    a.stages = stages;
    a.nstrips = cfn_cdiv(nst, stages);
    const unsigned blocks = (unsigned)((long)N * a.nstrips);
    const size_t lds = ((size_t)(32 * M32 + 32 * K32) * WG_PITCH + 2 * (32 * M32 + 32 * K32)) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_STEM, st, 4.0 * N * ((double)Cimg * a.Pin + (double)Cout * a.Q));
#define CFN_WG_GO(MW, NW) hipLaunchKernelGGL((pw_wgrad_kernel<MW, NW>), dim3(blocks), dim3(256), lds, st, a)
    if (M32 == 1 && K32 == 1) CFN_WG_GO(1, 1);
    else if (M32 == 1) CFN_WG_GO(1, 2);
    else if (M32 == 2 && K32 == 1) CFN_WG_GO(2, 1);
    else if (M32 == 2) CFN_WG_GO(2, 2);
    else if (K32 == 1) CFN_WG_GO(3, 1);
    else CFN_WG_GO(3, 2);
#undef CFN_WG_GO
    return cfn_check_launch("stem_conv_bwd_weight");
}
