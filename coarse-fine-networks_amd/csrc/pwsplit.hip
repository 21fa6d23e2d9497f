// Pointwise (1x1x1) channel contractions on FP32 tensors with SPLIT-BF16 arithmetic (x3d_fine.py:100-105 conv1 / conv3 of
// layers 2-4): tensors stay fp32 in HBM, every MFMA operand is split on the fly into NS bf16 terms
//     v = v1 + v2 (+ v3),   v1 = bf16(v), v2 = bf16(v - v1), v3 = bf16(v - v1 - v2)        (each subtraction is exact in fp32)
// and the product a*b is evaluated as the leading terms of (a1 + a2 + a3)(b1 + b2 + b3) on v_mfma_f32_32x32x16_bf16 with
// fp32 accumulation (bf16 x bf16 products are exact in fp32):
//     NS = 2: a1b1 + a1b2 + a2b1                        3 MFMAs, dropped terms <= 3 * 2^-18 |a b|
//     NS = 3: a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1   6 MFMAs, dropped terms <= 3 * 2^-27 |a b|  (below fp32 rounding)
// The bf16 matrix pipe runs 16x the fp32 one (v_mfma_f32_32x32x2_f32 = the fp32 vector rate), so the layers that were bound
// by the 157 TFLOP/s fp32-MFMA ceiling (K >= 48, profiles/r02_microbench_b8.txt) become HBM bound.
//
// Layout (N, C, Q), Q = T*H*W contiguous fp32 positions per (sample, channel) row.  The contraction runs over channels,
// Q*4 bytes apart, while the MFMA wants 8 consecutive k per lane: lane (j = l & 31, kg = l >> 5) loads ONE 8-byte position
// pair (q0 + 2j, q0 + 2j + 1) of channel kb*16 + kg*8 + i, i = 0..7 (32 lanes x 8 B = two whole 128-byte lines per row and
// instruction), applies the load-time prologue, splits, and packs the even positions into the B operands of an EVEN tile
// and the odd positions into those of an ODD tile; the two 32x32 results leave as one 8-byte store per lane (whole lines).
// A wave owns all BM = 32*MT output rows of its 64 positions: every activation is loaded, activated and split once.
// 8 waves share a resident weight slab, split once per workgroup into NS bf16 images in LDS (rows padded to an odd number
// of 16-byte slots: conflict-free ds_read_b128).  Rows beyond the slab = more slabs side by side (re-reads hit the XCD's L2).
#include "pw_common.h"
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

#define PWS_WAVES 8
#define PWS_OOB 0x40000000     // beyond every range used here (< 2^30 bytes per sample block); OOB + row offsets stay positive

// NOTE: __builtin_bit_cast(float, v.y) on an ELEMENT of an ext-vector lvalue reads element 0 (hipcc 7.2 front end: the element
// index is dropped); pws_f takes the element by value first
__device__ __forceinline__ float pws_f(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ float pws_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float pws_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ unsigned pws_pack(float lo, float hi) {
    const bf16x2 b = __builtin_convertvector((f2v){lo, hi}, bf16x2);   // v_cvt_pk_bf16_f32 (round to nearest even)
    return __builtin_bit_cast(unsigned, b);
}
// two fp32 values -> NS packed bf16 pairs, p[s] = (term s of v0 | term s of v1 << 16)
template <int NS>
__device__ __forceinline__ void pws_split(float v0, float v1, unsigned (&p)[NS]) {
    p[0] = pws_pack(v0, v1);
#pragma unroll
    for (int s = 1; s < NS; ++s) {
        v0 -= pws_lo(p[s - 1]);
        v1 -= pws_hi(p[s - 1]);
        p[s] = pws_pack(v0, v1);
    }
}
// the leading terms of the split product, smallest first
template <int NS, class F>
__device__ __forceinline__ void pws_terms(F&& f) {
    if (NS == 3) { f(2, 0); f(0, 2); f(1, 1); }
    if (NS >= 2) { f(1, 0); f(0, 1); }
    f(0, 0);
}

// transpose-reduce over the 32 column lanes (as pwbf16.hip): a lane starts with 16 row values of its column; the lane ends
// up with the 32-lane sum of row (lane & 31) >> 1 of the 16-row set
__device__ __forceinline__ float pws_fold16(float lo_row, float hi_row, int lane) {
    const bool b4 = lane & 16;
    const float send = b4 ? lo_row : hi_row, keep = b4 ? hi_row : lo_row;
    return keep + __shfl_xor(send, 16, 64);
}
__device__ __forceinline__ float pws_rowsum(const float (&a)[8], int lane) {
    float b[4], c[2];
    const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float send = b3 ? a[i] : a[i + 4], keep = b3 ? a[i + 4] : a[i];
        b[i] = keep + __shfl_xor(send, 8, 64);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float send = b2 ? b[i] : b[i + 2], keep = b2 ? b[i + 2] : b[i];
        c[i] = keep + __shfl_xor(send, 4, 64);
    }
    const float send = b1 ? c[0] : c[1], keep = b1 ? c[1] : c[0];
    float d = keep + __shfl_xor(send, 2, 64);
    d += __shfl_xor(d, 1, 64);
    return d;
}

// PwArgs fields re-used by the plan: Kpad = K padded to 32, mtiles = row slabs, nstrips = workgroups per (n, slab),
// kres = LDS bytes per weight row of ONE split image
template <int MT, int MODE, bool STATS, int ACT, bool TWO, int NS>
__global__ __launch_bounds__(64 * PWS_WAVES) void pws_kernel(const PwArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BM = 32 * MT;
    const int tid = threadIdx.x, wave = cfn_uni(tid >> 6), lane = tid & 63, kg = lane >> 5, j = lane & 31;
    const int K = a.K, M = a.M, Q = a.Q, Kp = a.Kpad, rowb = a.kres;

    unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    const int slab = L % a.mtiles; L /= a.mtiles;     // slabs of the same positions run side by side: re-reads hit the XCD's L2
    const int wg = L % a.nstrips;
    const int n = L / a.nstrips;
    const int m0 = slab * BM;

    unsigned char* Ws = smem;                                              // [NS][BM][rowb] bf16 weight images
    const size_t img = (size_t)BM * rowb;
    float4* sP = reinterpret_cast<float4*>(Ws + NS * img);                 // [Kp] prologue coefficients
    float2* sE = reinterpret_cast<float2*>(sP + Kp);                       // [BM] epilogue coefficients (DGRAD)
    float* red = reinterpret_cast<float*>(sE + BM);                        // [PWS_WAVES][BM][2]

    for (int k = tid; k < Kp; k += 64 * PWS_WAVES) {
        float4 c = {1.0f, 0.0f, 1.0f, 0.0f};
        if (MODE == PW_FWD) {
            c.x = (k < K && a.pa) ? (float)a.pa[(long)n * K + k] : 1.0f;
            c.y = (k < K && a.pb) ? (float)a.pb[(long)n * K + k] : 0.0f;
        } else {
            c.x = (k < K && a.gs) ? (float)a.gs[(long)n * K + k] : 0.0f;
            c.y = (k < K && a.gq && a.src2) ? 2.0f * (float)a.gq[(long)n * K + k] : 0.0f;
            c.z = (k < K && a.gsc) ? (float)a.gsc[(long)n * K + k] : 1.0f;
        }
        sP[k] = c;
    }
    for (int m = tid; m < BM; m += 64 * PWS_WAVES) {
        const bool ok = (m0 + m) < M && MODE == PW_DGRAD && a.ea;
        sE[m] = ok ? float2{(float)a.ea[(long)n * M + m0 + m], (float)a.eb[(long)n * M + m0 + m]} : float2{1.0f, 0.0f};
    }
    // weight images: Ws[s][m][k] = term s of W[m0+m][k] (FWD, w is (M,K)) or of W[k][m0+m] (DGRAD, w is (K,M)); zero padded
    for (int e = tid; e < BM * (Kp / 2); e += 64 * PWS_WAVES) {
        int m, k2;
        if (MODE == PW_FWD) { m = e / (Kp / 2); k2 = (e - m * (Kp / 2)) * 2; }     // consecutive threads along k (w rows)
        else { k2 = (e / BM) * 2; m = e - (e / BM) * BM; }                          // consecutive threads along m (w rows)
        float v0 = 0.0f, v1 = 0.0f;
        if (m0 + m < M) {
            if (MODE == PW_FWD) {
                if (k2 < K) v0 = a.w[(long)(m0 + m) * a.Cin + k2];
                if (k2 + 1 < K) v1 = a.w[(long)(m0 + m) * a.Cin + k2 + 1];
            } else {
                if (k2 < K) v0 = a.w[(long)k2 * a.Cin + m0 + m];
                if (k2 + 1 < K) v1 = a.w[(long)(k2 + 1) * a.Cin + m0 + m];
            }
        }
        unsigned p[NS];
        pws_split<NS>(v0, v1, p);
#pragma unroll
        for (int s = 0; s < NS; ++s) *reinterpret_cast<unsigned*>(Ws + s * img + (size_t)m * rowb + k2 * 2) = p[s];
    }
    __syncthreads();

    constexpr bool two_src = MODE == PW_DGRAD && TWO;
    const long src_n = (long)n * K * Q, dst_n = (long)n * M * Q;
    __amdgpu_buffer_rsrc_t rs1 = cfn_rsrc(const_cast<float*>(a.src + src_n), (unsigned)((long)K * Q * 4));
    __amdgpu_buffer_rsrc_t rs2 = cfn_rsrc(const_cast<float*>((two_src ? a.src2 : a.src) + src_n), (unsigned)((long)K * Q * 4));
    // rows m0.. of the output sample block: rows >= M fall outside the range (stores dropped, loads return 0)
    const int mrows = max(min(BM, M - m0), 0);
    __amdgpu_buffer_rsrc_t rd = cfn_rsrc(a.dst + dst_n + (long)m0 * Q, (unsigned)((long)mrows * Q * 4));
    const bool has_ex = MODE == PW_DGRAD && a.ex && a.ea;
    __amdgpu_buffer_rsrc_t rx = cfn_rsrc(const_cast<float*>(has_ex ? a.ex + dst_n + (long)m0 * Q : a.src), has_ex ? (unsigned)((long)mrows * Q * 4) : 0u);
    const bool has_acc = MODE == PW_DGRAD && a.acc;
    const long accP = has_acc ? (long)(Q / ((long)a.Hi * a.Wi)) * a.acc_Ho * a.acc_Wo : 0;   // positions per (n, row) of the compact tensor
    __amdgpu_buffer_rsrc_t rac = cfn_rsrc(const_cast<float*>(has_acc ? a.acc + ((long)n * M + m0) * accP : a.src), has_acc ? (unsigned)((long)mrows * accP * 4) : 0u);

    float ssum[MT], qsum[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) ssum[mt] = qsum[mt] = 0.0f;

    const int ntiles = (Q + 63) >> 6, nkb = Kp >> 4;
    const int lane_voff = kg * 8 * Q * 4 + j * 8;                           // this lane's (channel group, position pair) offset
    const unsigned char* wrow = Ws + (size_t)j * rowb + kg * 16;             // A operand: row j (+32*mt), k = kb*16 + kg*8 ..

    for (int tile = wg * PWS_WAVES + wave; tile < ntiles; tile += a.nstrips * PWS_WAVES) {
        const int q0 = tile << 6;
        const bool cv = q0 + 2 * j < Q;                                      // Q is even: a pair is valid or not as a whole
        f16v acc[MT][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { acc[mt][0] = (f16v)0.0f; acc[mt][1] = (f16v)0.0f; }

        // offsets into the compact lattice tensor for the two positions of the pair (OOB = not on the lattice): with an even
        // width only the even position can be on it
        int ao = PWS_OOB, ao2 = PWS_OOB;
        const bool odd_w = has_acc && (a.Wi & 1);
        if (has_acc && cv) {
            auto lat = [&](int q) {
                const int w_ = q % a.Wi, h_ = (q / a.Wi) % a.Hi, t_ = q / (a.Wi * a.Hi);
                return ((h_ % a.acc_s) == 0 && (w_ % a.acc_s) == 0) ? ((t_ * a.acc_Ho + h_ / a.acc_s) * a.acc_Wo + w_ / a.acc_s) * 4 : PWS_OOB;
            };
            ao = lat(q0 + 2 * j);
            if (odd_w) ao2 = lat(q0 + 2 * j + 1);
        }
        // Row addressing of the epilogue: the lane part (column pair, kg's 4-row offset) sits in the vector offset, the
        // wave-uniform row base in the scalar offset; a row base beyond the slab's valid rows would push the scalar offset
        // past the range (which wraps instead of failing the check), so such rows are switched off through the vector offset.
        const int cvk = cv ? (q0 + 2 * j) * 4 + 4 * kg * Q * 4 : PWS_OOB;
        auto rowbase = [&](int mt, int r) { return mt * 32 + (r & 3) + 8 * (r >> 2); };
        u2v ld[2][8], ld2[2][8];
        // unconditional loads (exact vmcnt waits).  The hardware checks  voffset >= num_records - soffset: the scalar part
        // must never exceed the range (it would wrap), so a k-block that starts beyond K is switched off through the lane
        // offset; channels >= K inside a live block fall out of range by themselves and read as 0
        auto issue = [&](int kb, u2v (&d)[8], u2v (&d2)[8]) {
            const bool live = kb * 16 < K;
            const int so = live ? (kb * 16 * Q + q0) * 4 : 0;
            const int vo = live ? lane_voff : PWS_OOB;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                d[i] = __builtin_amdgcn_raw_buffer_load_b64(rs1, vo + i * Q * 4, so, 0);
                if (two_src) d2[i] = __builtin_amdgcn_raw_buffer_load_b64(rs2, vo + i * Q * 4, so, 0);
            }
        };
        auto compute = [&](int kb, const u2v (&d)[8], const u2v (&d2)[8]) {
            float ve[8], vo[8];
            const float4* cp = sP + kb * 16 + kg * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4 c = cp[i];
                float e = pws_f(d[i].x), o = pws_f(d[i].y);
                if (MODE == PW_FWD) {
                    e = cfn_act<ACT>(fmaf(e, c.x, c.y));
                    o = cfn_act<ACT>(fmaf(o, c.x, c.y));
                } else {
                    e = fmaf(e, c.z, c.x);
                    o = fmaf(o, c.z, c.x);
                    if (two_src) { e = fmaf(pws_f(d2[i].x), c.y, e); o = fmaf(pws_f(d2[i].y), c.y, o); }
                }
                ve[i] = e; vo[i] = o;
            }
            u4v pe[NS], po[NS];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                unsigned se[NS], so_[NS];
                pws_split<NS>(ve[2 * h], ve[2 * h + 1], se);
                pws_split<NS>(vo[2 * h], vo[2 * h + 1], so_);
#pragma unroll
                for (int s = 0; s < NS; ++s) { pe[s][h] = se[s]; po[s][h] = so_[s]; }
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (m0 + mt * 32 < M) {                                      // block-uniform: a ragged last slab skips its empty tiles
                    bf16x8 A[NS];
#pragma unroll
                    for (int s = 0; s < NS; ++s) A[s] = *reinterpret_cast<const bf16x8*>(wrow + s * img + (size_t)mt * 32 * rowb + kb * 32);
#define PWS_MM(SA, SB)                                                                                                     \
                    acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[SA], __builtin_bit_cast(bf16x8, pe[SB]), acc[mt][0], 0, 0, 0); \
                    acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[SA], __builtin_bit_cast(bf16x8, po[SB]), acc[mt][1], 0, 0, 0)
                    if constexpr (NS == 3) { PWS_MM(2, 0); PWS_MM(0, 2); PWS_MM(1, 1); }
                    PWS_MM(1, 0); PWS_MM(0, 1); PWS_MM(0, 0);
#undef PWS_MM
                }
            }
        };
        issue(0, ld[0], ld2[0]);
        for (int kb = 0; kb < nkb; kb += 2) {                               // Kp is a multiple of 32: nkb is even
            issue(kb + 1, ld[1], ld2[1]);
            compute(kb, ld[0], ld2[0]);
            issue(kb + 2, ld[0], ld2[0]);                                    // kb + 2 == nkb: out of range -> zeros, never used
            compute(kb + 1, ld[1], ld2[1]);
        }

        // ---- epilogue: C layout of the 32x32 tile: column = lane & 31 (position pair j), row = (r & 3) + 8 (r >> 2) + 4 kg
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (m0 + mt * 32 >= M) continue;
            // DGRAD epilogue operands (forward input x of the output rows, for act'): the 16 row loads of this 32-row tile go
            // out as ONE batch (next to their consumers they would cost one HBM round trip each)
            u2v xq[16];
            if (MODE == PW_DGRAD && STATS) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool live = rowbase(mt, r) < mrows;
                    xq[r] = __builtin_amdgcn_raw_buffer_load_b64(rx, live ? cvk : PWS_OOB, live ? rowbase(mt, r) * Q * 4 : 0, 0);
                }
            }
            if (has_acc) {      // compact gradient of the strided second consumer, added on its lattice
                float ap[16];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ap[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rac, ao + (rowbase(mt, r) + 4 * kg) * (int)accP * 4, 0, 0));
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][0][r] += ap[r];
                if (odd_w) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        ap[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rac, ao2 + (rowbase(mt, r) + 4 * kg) * (int)accP * 4, 0, 0));
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][1][r] += ap[r];
                }
            }
            float f1[8], f2[8];
#pragma unroll
            for (int rp = 0; rp < 8; ++rp) {
                float t1[2], t2[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int r = rp + 8 * hh;
                    const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                    float e = acc[mt][0][r], o = acc[mt][1][r];
                    t1[hh] = t2[hh] = 0.0f;
                    const bool live = rowbase(mt, r) < mrows;
                    if (MODE == PW_FWD) {
                        if (STATS) {
                            const float em = cv ? e : 0.0f, om = cv ? o : 0.0f;
                            t1[hh] = em + om; t2[hh] = fmaf(em, em, om * om);
                        }
                    } else if (STATS) {                                      // act' epilogue + prologue-coefficient gradients
                        const float2 c = sE[row];
                        const float xe = pws_f(xq[r].x), xo = pws_f(xq[r].y);
                        const float de = cv ? e * cfn_act_grad<ACT>(fmaf(xe, c.x, c.y)) : 0.0f;
                        const float dn = cv ? o * cfn_act_grad<ACT>(fmaf(xo, c.x, c.y)) : 0.0f;
                        t1[hh] = fmaf(de, xe, dn * xo); t2[hh] = de + dn;
                        e = de * c.x; o = dn * c.x;
                    }
                    const u2v st = {__builtin_bit_cast(unsigned, e), __builtin_bit_cast(unsigned, o)};
                    __builtin_amdgcn_raw_buffer_store_b64(st, rd, live ? cvk : PWS_OOB, live ? rowbase(mt, r) * Q * 4 : 0, 0);
                }
                if (STATS) { f1[rp] = pws_fold16(t1[0], t1[1], lane); f2[rp] = pws_fold16(t2[0], t2[1], lane); }
            }
            if (STATS) { ssum[mt] += pws_rowsum(f1, lane); qsum[mt] += pws_rowsum(f2, lane); }
        }
    }

    if (STATS && a.s1) {
        // lane (j, kg) holds row (j >> 1) of the 16-row set, i.e. tile row (r & 3) + 8 (r >> 2) + 4 kg with r = j >> 1
        if ((j & 1) == 0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int r = j >> 1, row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                red[(wave * BM + row) * 2] = ssum[mt];
                red[(wave * BM + row) * 2 + 1] = qsum[mt];
            }
        }
        __syncthreads();
        for (int m = tid; m < BM; m += 64 * PWS_WAVES) {
            if (m0 + m < M) {
                float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
                for (int w = 0; w < PWS_WAVES; ++w) { t1 += red[(w * BM + m) * 2]; t2 += red[(w * BM + m) * 2 + 1]; }
                atomicAdd(&a.s1[(long)n * M + m0 + m], (double)t1);
                atomicAdd(&a.s2[(long)n * M + m0 + m], (double)t2);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
static int g_pws_terms = -1;      // 0 = off (fp32 MFMA kernels), 3 / 6 = MFMAs per k-block
static int pws_terms_now() {
    if (g_pws_terms < 0) {
        const char* e = getenv("CFN_PW_SPLIT");
        g_pws_terms = e ? atoi(e) : 6;
        if (g_pws_terms != 0 && g_pws_terms != 3 && g_pws_terms != 6) g_pws_terms = 6;
    }
    return g_pws_terms;
}
extern "C" int cfn_pw_split_terms(int terms) {
    const int prev = pws_terms_now();
    if (terms == 0 || terms == 3 || terms == 6) g_pws_terms = terms;
    else if (terms != -1) return cfn_fail(CFN_ERR_ARG, "cfn_pw_split_terms: terms must be 0 (fp32 MFMA), 3 or 6 (or -1 to query)"), -2;
    return prev;
}

static size_t pws_lds(int BM, int Kp, int rowb, int NS) {
    return (size_t)NS * BM * rowb + (size_t)Kp * 16 + (size_t)BM * 8 + (size_t)PWS_WAVES * BM * 2 * 4;
}

template <int MODE, bool STATS, int ACT, bool TWO, int NS>
static int pws_go_mt(const PwArgs& a, int MT, unsigned blocks, size_t lds, hipStream_t st) {
#define PWS_GO(MTV)                                                                                                        \
    do {                                                                                                                   \
        auto k = pws_kernel<MTV, MODE, STATS, ACT, TWO, NS>;                                                               \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * PWS_WAVES), lds, st, a);                                             \
    } while (0)
    if constexpr (MODE == PW_DGRAD && TWO) {
        if (MT == 1) PWS_GO(1); else PWS_GO(2);
    } else if constexpr (MODE == PW_DGRAD) {
        switch (MT) { case 1: PWS_GO(1); break; case 2: PWS_GO(2); break; default: PWS_GO(3); break; }
    } else {
        switch (MT) { case 1: PWS_GO(1); break; case 2: PWS_GO(2); break; case 3: PWS_GO(3); break; default: PWS_GO(4); break; }
    }
#undef PWS_GO
    return cfn_check_launch("pwconv(split bf16)");
}

template <int MODE, bool STATS, bool TWO, int NS>
static int pws_go_act(const PwArgs& a, int MT, unsigned blocks, size_t lds, hipStream_t st) {
    if constexpr (MODE == PW_DGRAD && !STATS) {
        return pws_go_mt<MODE, STATS, CFN_ACT_NONE, TWO, NS>(a, MT, blocks, lds, st);
    } else {
        switch (a.act) {
            case CFN_ACT_RELU: return pws_go_mt<MODE, STATS, CFN_ACT_RELU, TWO, NS>(a, MT, blocks, lds, st);
            case CFN_ACT_SWISH: return pws_go_mt<MODE, STATS, CFN_ACT_SWISH, TWO, NS>(a, MT, blocks, lds, st);
            default: return pws_go_mt<MODE, STATS, CFN_ACT_NONE, TWO, NS>(a, MT, blocks, lds, st);
        }
    }
}

template <int NS>
static int pws_go(const PwArgs& a, int mode, bool stats, int MT, unsigned blocks, size_t lds, hipStream_t st) {
    if (mode == PW_FWD) return stats ? pws_go_act<PW_FWD, true, false, NS>(a, MT, blocks, lds, st) : pws_go_act<PW_FWD, false, false, NS>(a, MT, blocks, lds, st);
    if (a.src2) return stats ? pws_go_act<PW_DGRAD, true, true, NS>(a, MT, blocks, lds, st) : pws_go_act<PW_DGRAD, false, true, NS>(a, MT, blocks, lds, st);
    return stats ? pws_go_act<PW_DGRAD, true, false, NS>(a, MT, blocks, lds, st) : pws_go_act<PW_DGRAD, false, false, NS>(a, MT, blocks, lds, st);
}

// returns -1 when the shape is not handled (caller falls through to the fp32-MFMA kernels), otherwise the launch status.
// FWD prologue: pa == nullptr means A = 1, B = 0 (act still applies, as in the fp32-MFMA kernels); DGRAD: stats == (ea != nullptr)
int pws_try_launch(PwArgs& a, int mode, bool stats, hipStream_t st) {
    const int terms = pws_terms_now();
    if (terms == 0) return -1;
    if (a.stem || a.stride != 1 || a.K < 48 || a.M <= 32 || (a.Q & 1)) return -1;
    if (a.act != CFN_ACT_NONE && a.act != CFN_ACT_RELU && a.act != CFN_ACT_SWISH) return -1;
    if ((long)a.K * a.Q * 4 >= 0x3ffffff0L || (long)a.M * a.Q * 4 >= 0x3ffffff0L) return -1;
    if (((uintptr_t)a.src | (uintptr_t)a.dst | (uintptr_t)(a.src2 ? a.src2 : a.src) | (uintptr_t)(a.ex ? a.ex : a.src)) & 7) return -1;
    const int NS = terms == 3 ? 2 : 3;
    PwArgs b = a;
    b.Kpad = (a.K + 31) / 32 * 32;
    b.kres = b.Kpad * 2 + 16;
    if (((b.kres / 16) & 1) == 0) b.kres += 16;                          // odd number of 16-byte slots per row
    if (mode == PW_DGRAD && !stats) b.act = CFN_ACT_NONE;
    // rows per slab: registers allow 128 forward (8 accumulator tiles), 96 backward, 64 backward with two staged tensors
    // (96 spills: measured 80-188 bytes per lane); the NS weight images of a slab must fit LDS
    int mt_max = mode == PW_FWD ? 4 : (a.src2 ? 2 : 3);
    while (mt_max > 0 && pws_lds(32 * mt_max, b.Kpad, b.kres, NS) > 160 * 1024) --mt_max;
    if (mt_max < 1) return -1;
    int slabs = cfn_cdiv(a.M, 32 * mt_max);
    const int per = cfn_cdiv(a.M, slabs);
    const int MT = cfn_cdiv(per, 32);
    slabs = cfn_cdiv(a.M, 32 * MT);
    b.mtiles = slabs;
    const size_t lds = pws_lds(32 * MT, b.Kpad, b.kres, NS);
    const int ntiles = cfn_cdiv(a.Q, 64);
    const long groups = (long)a.N * slabs;
    static const int wg_env = getenv("CFN_PWS_WGS") ? atoi(getenv("CFN_PWS_WGS")) : 0;
    long wgs = cfn_cdiv(wg_env > 0 ? wg_env : 256, groups);              // one 8-wave workgroup per CU (up to 256 VGPRs)
    const long maxw = cfn_cdiv(ntiles, PWS_WAVES);
    if (wgs > maxw) wgs = maxw;
    if (wgs < 1) wgs = 1;
    b.nstrips = (int)wgs;
    const unsigned blocks = (unsigned)(groups * wgs);
    return NS == 2 ? pws_go<2>(b, mode, stats, MT, blocks, lds, st) : pws_go<3>(b, mode, stats, MT, blocks, lds, st);
}
