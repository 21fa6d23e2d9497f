// Pointwise (1x1x1) channel contractions on bf16 activations: v_mfma_f32_32x32x16_bf16, fp32 accumulation, fp32 master
// weights (x3d_fine.py:100-105 conv1 / conv3 / shortcut, conv5 :245-250).  The bf16 storage path of BASELINE configs[1].
//
// Tensors are (N, C, Q) with Q = T*H*W contiguous positions per (sample, channel) row, bf16.  The contraction runs over
// channels, which are Q*2 bytes apart, so the MFMA's "8 consecutive k per lane" operand is built from 8 row loads:
//   lane (j = l & 31, kg = l >> 5) loads ONE dword = the position pair (q0+2j, q0+2j+1) of channel kb*16 + kg*8 + i,
//   i = 0..7 (32 lanes x 4 B = one whole 128-byte line per row per instruction), applies the load-time prologue in
//   fp32, and packs the eight low halves into the B operand of the EVEN-position tile and the eight high halves into the
//   B operand of the ODD-position tile.  Two 32x32 accumulator tiles per 32 output rows cover 64 consecutive positions;
//   their (even, odd) results are re-packed into one dword per lane, so every store instruction again writes whole
//   128-byte lines.
// A wave owns all BM = 32*MT (<= 128) output rows of its 64 positions: every activation is loaded, pushed through the
// prologue and converted once, and feeds 2*MT MFMAs.  8 waves share one resident bf16 weight slab in LDS (rows padded to
// an odd number of 16-byte slots: conflict-free ds_read_b128 of the A operand); position tiles are dealt round robin.
// Per-(n, row) reductions (BN statistics forward; sum dz*x, sum dz backward) use a transpose-reduce butterfly over the
// 32 column lanes (16 shuffles per 16 rows) and accumulate in registers across tiles: one fp64 atomic per row per block.
#include "pw_common.h"
#include <stdint.h>

// element kind of the 2-byte tensors: bf16, or IEEE half when compiled through pwf16.hip (h16.h); "bf16" in the names below = "16-bit"
#include "h16.h"
typedef h16x8 bf16x8;
#define PwbArgs H16N(PwbArgs)
#define PwbWgArgs H16N(PwbWgArgs)
#define subsample_hw_bf16_kernel H16N(subsample_hw_kernel)
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

enum { PWB_FWD = 0, PWB_DGRAD = 1 };
#define PWB_WAVES 8
#define PWB_OOB 0x40000000     // beyond every range used here (< 2^30 bytes per sample block), and OOB + offset stays positive

struct PwbArgs {
    const uint16_t* src;    // FWD: x (N,K,Q)        DGRAD: gy (N,K,Q)          K = contraction channels
    const uint16_t* src2;   // DGRAD: raw y (N,K,Q) for the 2*y*gq term (may be null)
    const double* pa;       // FWD prologue A[n,k], B[n,k] (null = identity)
    const double* pb;
    const double* gs;       // DGRAD: g' = gsc*gy + gs + 2*y*gq, all [n,k], each may be null
    const double* gq;
    const double* gsc;
    const float* w;         // (Cout, Cin) fp32
    uint16_t* dst;          // FWD: y (N,M,Q)        DGRAD: gx (N,M,Q)          M = output rows
    const uint16_t* ex;     // DGRAD: forward input x (N,M,Q) (needed when ea != null)
    const double* ea;       // DGRAD epilogue: forward prologue A[n,m], B[n,m] (null: gx = W^T g')
    const double* eb;
    const uint16_t* acc;    // DGRAD: compact gradient (N,M,T,aHo,aWo) of a strided second consumer, added on its lattice
    int acc_s, H, W, aHo, aWo;
    double* s1;             // FWD: sum(y)   DGRAD: sum(dz*x)   [n,m]
    double* s2;             // FWD: sum(y^2) DGRAD: sum(dz)
    int N, M, K, Q, act, Cin, Cout;
    int Kp, mslabs, wgs, rowb;   // Kp: K padded to 32; wgs: workgroups per (n, slab); rowb: LDS bytes per weight row
};

__device__ __forceinline__ float pwb_lo(unsigned u) { return h16_lo(u); }
__device__ __forceinline__ float pwb_hi(unsigned u) { return h16_hi(u); }
__device__ __forceinline__ unsigned pwb_pack(float lo, float hi) { return h16_pk(lo, hi); }   // v_cvt_pk_{bf16,f16}_f32 (round to nearest even)

// transpose-reduce over the 32 column lanes: a lane starts with 16 row values (its column); level 1 (pwb_fold16) exchanges
// rows i / i+8 with lane ^ 16 as soon as both exist (8 live registers instead of 16), pwb_rowsum finishes: the lane ends up
// with the 32-lane sum of row (lane & 31) >> 1
__device__ __forceinline__ float pwb_fold16(float lo_row, float hi_row, int lane) {
    const bool b4 = lane & 16;
    const float send = b4 ? lo_row : hi_row, keep = b4 ? hi_row : lo_row;
    return keep + __shfl_xor(send, 16, 64);
}
__device__ __forceinline__ float pwb_rowsum(const float (&a)[8], int lane) {
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

// TWO (DGRAD): the 2*y*gq term is present, i.e. a second tensor is streamed.  A template parameter, not a runtime flag: a load
// under a (even uniform) branch makes the compiler wait with vmcnt(0), which would serialise the k-block prefetch.
template <int MT, int MODE, bool STATS, int ACT, bool TWO>
__global__ __launch_bounds__(64 * PWB_WAVES) void pwb_kernel(const PwbArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BM = 32 * MT;
    const int tid = threadIdx.x, wave = cfn_uni(tid >> 6), lane = tid & 63, kg = lane >> 5, j = lane & 31;
    const int K = a.K, M = a.M, Q = a.Q, Kp = a.Kp, rowb = a.rowb;

    unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    const int slab = L % a.mslabs; L /= a.mslabs;     // slabs of the same positions run side by side: re-reads hit the XCD's L2
    const int wg = L % a.wgs;
    const int n = L / a.wgs;
    const int m0 = slab * BM;

    unsigned char* Ws = smem;                                              // [BM][rowb] bf16 weight slab
    float4* sP = reinterpret_cast<float4*>(Ws + (size_t)BM * rowb);        // [Kp] prologue coefficients
    float2* sE = reinterpret_cast<float2*>(sP + Kp);                       // [BM] epilogue coefficients (DGRAD)
    float* red = reinterpret_cast<float*>(sE + BM);                        // [PWB_WAVES][MT][2][16][2]

    for (int k = tid; k < Kp; k += 64 * PWB_WAVES) {
        float4 c = {1.0f, 0.0f, 1.0f, 0.0f};
        if (MODE == PWB_FWD) {
            c.x = (k < K && a.pa) ? (float)a.pa[(long)n * K + k] : 1.0f;
            c.y = (k < K && a.pb) ? (float)a.pb[(long)n * K + k] : 0.0f;
        } else {
            c.x = (k < K && a.gs) ? (float)a.gs[(long)n * K + k] : 0.0f;
            c.y = (k < K && a.gq && a.src2) ? 2.0f * (float)a.gq[(long)n * K + k] : 0.0f;
            c.z = (k < K && a.gsc) ? (float)a.gsc[(long)n * K + k] : 1.0f;
        }
        sP[k] = c;
    }
    for (int m = tid; m < BM; m += 64 * PWB_WAVES) {
        const bool ok = (m0 + m) < M && MODE == PWB_DGRAD && a.ea;
        sE[m] = ok ? float2{(float)a.ea[(long)n * M + m0 + m], (float)a.eb[(long)n * M + m0 + m]} : float2{1.0f, 0.0f};
    }
    // weight slab: Ws[m][k] = bf16(W[m0+m][k]) (FWD, w is (M,K)) or bf16(W[k][m0+m]) (DGRAD, w is (K,M)); zero padded
    for (int e = tid; e < BM * (Kp / 2); e += 64 * PWB_WAVES) {
        int m, k2;
        if (MODE == PWB_FWD) { m = e / (Kp / 2); k2 = (e - m * (Kp / 2)) * 2; }     // consecutive threads along k (w rows)
        else { k2 = (e / BM) * 2; m = e - (e / BM) * BM; }                          // consecutive threads along m (w rows)
        float v0 = 0.0f, v1 = 0.0f;
        if (m0 + m < M) {
            if (MODE == PWB_FWD) {
                if (k2 < K) v0 = a.w[(long)(m0 + m) * a.Cin + k2];
                if (k2 + 1 < K) v1 = a.w[(long)(m0 + m) * a.Cin + k2 + 1];
            } else {
                if (k2 < K) v0 = a.w[(long)k2 * a.Cin + m0 + m];
                if (k2 + 1 < K) v1 = a.w[(long)(k2 + 1) * a.Cin + m0 + m];
            }
        }
        *reinterpret_cast<unsigned*>(Ws + (size_t)m * rowb + k2 * 2) = pwb_pack(v0, v1);
    }
    __syncthreads();

    constexpr bool two_src = MODE == PWB_DGRAD && TWO;
    const long src_n = (long)n * K * Q, dst_n = (long)n * M * Q;
    __amdgpu_buffer_rsrc_t rs1 = cfn_rsrc(const_cast<uint16_t*>(a.src + src_n), (unsigned)((long)K * Q * 2));
    __amdgpu_buffer_rsrc_t rs2 = cfn_rsrc(const_cast<uint16_t*>((two_src ? a.src2 : a.src) + src_n), (unsigned)((long)K * Q * 2));
    // rows m0.. of the output sample block: rows >= M fall outside the range (stores dropped, loads return 0)
    const int mrows = max(min(BM, M - m0), 0);
    __amdgpu_buffer_rsrc_t rd = cfn_rsrc(a.dst + dst_n + (long)m0 * Q, (unsigned)((long)mrows * Q * 2));
    __amdgpu_buffer_rsrc_t rx = cfn_rsrc(const_cast<uint16_t*>((MODE == PWB_DGRAD && a.ex ? a.ex : a.src) + (MODE == PWB_DGRAD && a.ex ? dst_n + (long)m0 * Q : 0)), (MODE == PWB_DGRAD && a.ex) ? (unsigned)((long)mrows * Q * 2) : 0u);
    const long accP = (long)(Q / ((long)a.H * a.W)) * a.aHo * a.aWo;      // positions per (n, row) of the compact tensor
    __amdgpu_buffer_rsrc_t rac = cfn_rsrc(const_cast<uint16_t*>((MODE == PWB_DGRAD && a.acc ? a.acc + ((long)n * M + m0) * accP : a.src)), (MODE == PWB_DGRAD && a.acc) ? (unsigned)((long)mrows * accP * 2) : 0u);

    float ssum[MT], qsum[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) ssum[mt] = qsum[mt] = 0.0f;

    const int ntiles = (Q + 63) >> 6, nkb = Kp >> 4;
    const int lane_voff = kg * 8 * Q * 2 + j * 4;                           // this lane's (channel group, position pair) offset
    const unsigned char* wrow = Ws + (size_t)j * rowb + kg * 16;             // A operand: row j (+32*mt), k = kb*16 + kg*8 ..

    for (int tile = wg * PWB_WAVES + wave; tile < ntiles; tile += a.wgs * PWB_WAVES) {
        const int q0 = tile << 6;
        const bool cv = q0 + 2 * j < Q;                                      // Q is even: a pair is valid or not as a whole
        f16v acc[MT][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { acc[mt][0] = (f16v)0.0f; acc[mt][1] = (f16v)0.0f; }

        const int cvo = cv ? (q0 + 2 * j) * 2 : PWB_OOB;
        // offsets into the compact lattice tensor for the two positions of the pair (OOB = not on the lattice): with an even
        // width only the even position can be on it
        int ao = PWB_OOB, ao2 = PWB_OOB;
        const bool odd_w = MODE == PWB_DGRAD && a.acc && (a.W & 1);
        if (MODE == PWB_DGRAD && a.acc && cv) {
            auto lat = [&](int q) {
                const int w_ = q % a.W, h_ = (q / a.W) % a.H, t_ = q / (a.W * a.H);
                return ((h_ % a.acc_s) == 0 && (w_ % a.acc_s) == 0) ? ((t_ * a.aHo + h_ / a.acc_s) * a.aWo + w_ / a.acc_s) * 2 : PWB_OOB;
            };
            ao = lat(q0 + 2 * j);
            if (odd_w) ao2 = lat(q0 + 2 * j + 1);
        }
        // DGRAD epilogue operands (forward input x of the output rows, for act'): 16 row loads per 32-row tile, issued as ONE
        // batch one phase ahead of their use (tile 0 here, behind the first k-block; tile mt+1 before the epilogue of tile mt)
        // -- issued next to their consumers they would cost one HBM round trip EACH
        // Row addressing: the lane part (column pair, kg's 4-row offset) sits in the vector offset, the wave-uniform row base
        // in the scalar offset; a row base beyond the slab's valid rows would push the scalar offset past the range (which
        // wraps instead of failing the check), so such rows are switched off through the vector offset.
        const int cvk = cv ? (q0 + 2 * j) * 2 + 4 * kg * Q * 2 : PWB_OOB;
        auto rowbase = [&](int mt, int r) { return mt * 32 + (r & 3) + 8 * (r >> 2); };
        unsigned ld[2][8], ld2[2][8];
        // unconditional loads (exact vmcnt waits).  The hardware checks  voffset >= num_records - soffset: the scalar part
        // must never exceed the range (it would wrap), so a k-block that starts beyond K is switched off through the lane
        // offset; channels >= K inside a live block fall out of range by themselves and read as 0
        auto issue = [&](int kb, unsigned (&d)[8], unsigned (&d2)[8]) {
            const bool live = kb * 16 < K;
            const int so = live ? (kb * 16 * Q + q0) * 2 : 0;
            const int vo = live ? lane_voff : PWB_OOB;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                d[i] = __builtin_amdgcn_raw_buffer_load_b32(rs1, vo + i * Q * 2, so, 0);
                if (MODE == PWB_DGRAD && two_src) d2[i] = __builtin_amdgcn_raw_buffer_load_b32(rs2, vo + i * Q * 2, so, 0);
            }
        };
        auto compute = [&](int kb, const unsigned (&d)[8], const unsigned (&d2)[8]) {
            float ve[8], vo[8];
            const float4* cp = sP + kb * 16 + kg * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4 c = cp[i];
                float e = pwb_lo(d[i]), o = pwb_hi(d[i]);
                if (MODE == PWB_FWD) {
                    e = cfn_act<ACT>(fmaf(e, c.x, c.y));
                    o = cfn_act<ACT>(fmaf(o, c.x, c.y));
                } else {
                    e = fmaf(e, c.z, c.x);
                    o = fmaf(o, c.z, c.x);
                    if (two_src) { e = fmaf(pwb_lo(d2[i]), c.y, e); o = fmaf(pwb_hi(d2[i]), c.y, o); }
                }
                ve[i] = e; vo[i] = o;
            }
            u4v pe, po;
            pe.x = pwb_pack(ve[0], ve[1]); pe.y = pwb_pack(ve[2], ve[3]); pe.z = pwb_pack(ve[4], ve[5]); pe.w = pwb_pack(ve[6], ve[7]);
            po.x = pwb_pack(vo[0], vo[1]); po.y = pwb_pack(vo[2], vo[3]); po.z = pwb_pack(vo[4], vo[5]); po.w = pwb_pack(vo[6], vo[7]);
            const bf16x8 Be = __builtin_bit_cast(bf16x8, pe), Bo = __builtin_bit_cast(bf16x8, po);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (m0 + mt * 32 < M) {                                      // block-uniform: a ragged last slab skips its empty tiles
                    const bf16x8 A = *reinterpret_cast<const bf16x8*>(wrow + (size_t)mt * 32 * rowb + kb * 32);
                    acc[mt][0] = H16_MFMA32(A, Be, acc[mt][0], 0, 0, 0);
                    acc[mt][1] = H16_MFMA32(A, Bo, acc[mt][1], 0, 0, 0);
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
            // DGRAD epilogue operands: the 16 row loads of this 32-row tile go out as ONE batch (next to their consumers
            // they would cost one HBM round trip each)
            unsigned xq[16];
            if (MODE == PWB_DGRAD && STATS) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool live = rowbase(mt, r) < mrows;
                    xq[r] = __builtin_amdgcn_raw_buffer_load_b32(rx, live ? cvk : PWB_OOB, live ? rowbase(mt, r) * Q * 2 : 0, 0);
                }
            }
            if (MODE == PWB_DGRAD && a.acc) {      // compact gradient of the strided second consumer, added on its lattice
                unsigned ap[16];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ap[r] = (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rac, ao + (rowbase(mt, r) + 4 * kg) * (int)accP * 2, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][0][r] += pwb_lo(ap[r]);
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
                    if (MODE == PWB_FWD) {
                        const unsigned p = pwb_pack(e, o);
                        __builtin_amdgcn_raw_buffer_store_b32(p, rd, rowbase(mt, r) < mrows ? cvk : PWB_OOB, rowbase(mt, r) < mrows ? rowbase(mt, r) * Q * 2 : 0, 0);
                        if (STATS) {                                         // statistics of what the consumer will read
                            e = cv ? pwb_lo(p) : 0.0f; o = cv ? pwb_hi(p) : 0.0f;
                            t1[hh] = e + o; t2[hh] = fmaf(e, e, o * o);
                        }
                    } else {
                        if (odd_w) o += pwb_lo((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rac, ao2 + row * (int)accP * 2, 0, 0));
                        if (STATS) {                                         // act' epilogue + prologue-coefficient gradients
                            const unsigned xp = xq[r];
                            const float2 c = cfn_settle(sE[row]);                // (even / odd position: one packed FMA behind the LDS read, see cfn_settle)
                            const float xe = pwb_lo(xp), xo = pwb_hi(xp);
                            const float de = cv ? e * cfn_act_grad<ACT>(fmaf(xe, c.x, c.y)) : 0.0f;
                            const float dn = cv ? o * cfn_act_grad<ACT>(fmaf(xo, c.x, c.y)) : 0.0f;
                            t1[hh] = fmaf(de, xe, dn * xo); t2[hh] = de + dn;
                            e = de * c.x; o = dn * c.x;
                        }
                        __builtin_amdgcn_raw_buffer_store_b32(pwb_pack(e, o), rd, rowbase(mt, r) < mrows ? cvk : PWB_OOB, rowbase(mt, r) < mrows ? rowbase(mt, r) * Q * 2 : 0, 0);
                    }
                }
                if (STATS) { f1[rp] = pwb_fold16(t1[0], t1[1], lane); f2[rp] = pwb_fold16(t2[0], t2[1], lane); }
            }
            if (STATS) { ssum[mt] += pwb_rowsum(f1, lane); qsum[mt] += pwb_rowsum(f2, lane); }
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
        for (int m = tid; m < BM; m += 64 * PWB_WAVES) {
            if (m0 + m < M) {
                float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
                for (int w = 0; w < PWB_WAVES; ++w) { t1 += red[(w * BM + m) * 2]; t2 += red[(w * BM + m) * 2 + 1]; }
                cfn_add64(&a.s1[(long)n * M + m0 + m], (double)t1);
                cfn_add64(&a.s2[(long)n * M + m0 + m], (double)t2);
            }
        }
    }
}

// rows per weight slab: 128 forward, 64 backward (two staged tensors and the act' epilogue need the registers)
static int pwb_plan(PwbArgs& a, int& MT, unsigned& blocks, size_t& lds, int max_rows) {
    if (a.Q % 2) return cfn_fail(CFN_ERR_UNSUPPORTED, H16_NAME " pointwise conv: T*H*W = %d must be even (position pairs share a dword)", a.Q);
    a.Kp = (a.K + 31) / 32 * 32;
    // every descriptor spans one sample's (rows x positions) block: the contraction side (K rows, padded k-blocks address up
    // to Kp), the output / epilogue side (M rows)
    if ((long)a.Kp * a.Q * 2 >= 0x3ffffff0L || (long)a.M * a.Q * 2 >= 0x3ffffff0L)
        return cfn_fail(CFN_ERR_UNSUPPORTED, H16_NAME " pointwise conv: a sample's (channels x positions) block exceeds the 1 GiB buffer range");
    a.mslabs = (a.M + max_rows - 1) / max_rows;
    const int per = (a.M + a.mslabs - 1) / a.mslabs;
    MT = (per + 31) / 32;
    a.mslabs = (a.M + MT * 32 - 1) / (MT * 32);
    a.rowb = a.Kp * 2 + 16;
    if (((a.rowb / 16) & 1) == 0) a.rowb += 16;                          // odd number of 16-byte slots per row
    const int BM = 32 * MT;
    lds = (size_t)BM * a.rowb + (size_t)a.Kp * 16 + (size_t)BM * 8 + (size_t)PWB_WAVES * BM * 2 * 4;
    if (lds > 160 * 1024) return cfn_fail(CFN_ERR_UNSUPPORTED, H16_NAME " pointwise conv: K = %d does not fit the LDS weight slab", a.K);
    const int ntiles = (a.Q + 63) / 64;
    const long groups = (long)a.N * a.mslabs;
    long wgs = (512 + groups - 1) / groups;                              // ~2 rounds of the chip
    const long maxw = (ntiles + PWB_WAVES - 1) / PWB_WAVES;
    if (wgs > maxw) wgs = maxw;
    if (wgs < 1) wgs = 1;
    a.wgs = (int)wgs;
    blocks = (unsigned)(groups * wgs);
    return CFN_OK;
}

template <int MODE, bool STATS, int ACT, bool TWO>
static int pwb_launch_mt(const PwbArgs& a, int MT, unsigned blocks, size_t lds, hipStream_t st) {
#define PWB_GO(MTV)                                                                                                        \
    do {                                                                                                                   \
        auto k = pwb_kernel<MTV, MODE, STATS, ACT, TWO>;                                                                        \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * PWB_WAVES), lds, st, a);                                             \
    } while (0)
    if constexpr (MODE == PWB_DGRAD) {
        if (MT == 1) PWB_GO(1); else PWB_GO(2);
    } else {
        switch (MT) {
            case 1: PWB_GO(1); break;
            case 2: PWB_GO(2); break;
            case 3: PWB_GO(3); break;
            default: PWB_GO(4); break;
        }
    }
#undef PWB_GO
    return cfn_check_launch("pwconv " H16_NAME);
}

template <int MODE, bool STATS, bool TWO = false>
static int pwb_launch(const PwbArgs& a, int MT, unsigned blocks, size_t lds, hipStream_t st) {
    switch (a.act) {
        case CFN_ACT_RELU: return pwb_launch_mt<MODE, STATS, CFN_ACT_RELU, TWO>(a, MT, blocks, lds, st);
        case CFN_ACT_SWISH: return pwb_launch_mt<MODE, STATS, CFN_ACT_SWISH, TWO>(a, MT, blocks, lds, st);
        default: return pwb_launch_mt<MODE, STATS, CFN_ACT_NONE, TWO>(a, MT, blocks, lds, st);
    }
}

extern "C" int H16N(cfn_pwconv_fwd)(const uint16_t* x, const double* A, const double* B, int act, const float* w, uint16_t* y,
                                   double* sum, double* sumsq, int N, int Cin, int Cout, long Q, void* stream) {
    CFN_REQUIRE(x && w && y, "cfn_pwconv_fwd_" H16_NAME ": null tensor");
    CFN_REQUIRE((A == nullptr) == (B == nullptr) && (sum == nullptr) == (sumsq == nullptr), "cfn_pwconv_fwd_" H16_NAME ": A/B, sum/sumsq go together");
    CFN_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && Q > 0 && Q < 0x7fffffffL, "cfn_pwconv_fwd_" H16_NAME ": bad shape");
    CFN_REQUIRE(act >= CFN_ACT_NONE && act <= CFN_ACT_SWISH, "cfn_pwconv_fwd_" H16_NAME ": bad activation %d", act);
    PwbArgs a = {};
    a.src = x; a.pa = A; a.pb = B; a.act = A ? act : CFN_ACT_NONE; a.w = w; a.dst = y; a.s1 = sum; a.s2 = sumsq;
    a.N = N; a.M = Cout; a.K = Cin; a.Q = (int)Q; a.Cin = Cin; a.Cout = Cout; a.H = a.W = 1;
    int MT; unsigned blocks; size_t lds;
    int rc = pwb_plan(a, MT, blocks, lds, 128);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_PWCONV_FWD, st, 2.0 * N * ((double)Cin + Cout) * (double)Q + 4.0 * Cin * Cout);
    return sum ? pwb_launch<PWB_FWD, true>(a, MT, blocks, lds, st) : pwb_launch<PWB_FWD, false>(a, MT, blocks, lds, st);
}

extern "C" int H16N(cfn_pwconv_bwd_data)(const uint16_t* gy, const uint16_t* y, const double* gsum, const double* gsumsq,
                                        const float* w, const uint16_t* x, const double* A, const double* B, int act,
                                        uint16_t* gx, double* gA, double* gB, int N, int Cin, int Cout, int T, int H, int W,
                                        const uint16_t* acc, int acc_stride, const double* gscale, void* stream) {
    CFN_REQUIRE(gy && w && gx, "cfn_pwconv_bwd_data_" H16_NAME ": null tensor");
    CFN_REQUIRE((A == nullptr) == (B == nullptr), "cfn_pwconv_bwd_data_" H16_NAME ": A/B mismatch");
    CFN_REQUIRE(A == nullptr || (x && gA && gB), "cfn_pwconv_bwd_data_" H16_NAME ": prologue needs x, gA, gB");
    CFN_REQUIRE(gsumsq == nullptr || y != nullptr, "cfn_pwconv_bwd_data_" H16_NAME ": gsumsq needs y");
    CFN_REQUIRE(acc == nullptr || acc_stride >= 1, "cfn_pwconv_bwd_data_" H16_NAME ": bad acc_stride");
    PwbArgs a = {};
    a.src = gy; a.src2 = gsumsq ? y : nullptr; a.gs = gsum; a.gq = gsumsq; a.gsc = gscale; a.w = w; a.dst = gx;
    a.ex = A ? x : nullptr; a.ea = A; a.eb = B; a.act = A ? act : CFN_ACT_NONE; a.s1 = gA; a.s2 = gB;
    a.acc = acc; a.acc_s = acc ? acc_stride : 1; a.H = H; a.W = W;
    a.aHo = (H - 1) / a.acc_s + 1; a.aWo = (W - 1) / a.acc_s + 1;
    a.N = N; a.M = Cin; a.K = Cout; a.Q = T * H * W; a.Cin = Cin; a.Cout = Cout;
    CFN_REQUIRE((long)T * H * W < 0x7fffffffL, "cfn_pwconv_bwd_data_" H16_NAME ": too many positions");
    int MT; unsigned blocks; size_t lds;
    int rc = pwb_plan(a, MT, blocks, lds, 64);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_PWCONV_BWD, st, 2.0 * N * ((double)Cout * a.Q * (a.src2 ? 2 : 1) + (double)Cin * a.Q * (A ? 2 : 1)));
    if (a.src2) return A ? pwb_launch<PWB_DGRAD, true, true>(a, MT, blocks, lds, st) : pwb_launch<PWB_DGRAD, false, true>(a, MT, blocks, lds, st);
    return A ? pwb_launch<PWB_DGRAD, true, false>(a, MT, blocks, lds, st) : pwb_launch<PWB_DGRAD, false, false>(a, MT, blocks, lds, st);
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient  gw[m,k] += sum_q g'[n,m,q] * act(A x + B)[n,k,q]: the contraction runs over POSITIONS, which are
// contiguous in both operands, so each MFMA operand is one 16-byte load per lane: lane (r = l & 31, kg = l >> 5) takes the 8
// positions p0 + kg*8 .. +7 of row r (gy / y rows for the A operand, x rows for the B operand), applies the prologue in
// fp32 and re-packs.  A wave owns an (32*MTW) x (32*NTW) block of gw and a strip of positions; the 8 waves of a workgroup
// take different strips and are combined through LDS into one fp64 atomic per element per workgroup.
// ---------------------------------------------------------------------------------------------------------------------
struct PwbWgArgs {
    const uint16_t* gy; const uint16_t* y; const double* gs; const double* gq; const double* gsc;
    const uint16_t* x; const double* pa; const double* pb; double* gw;
    int N, M, K, Q, act, mblocks, kblocks, strips;
};

template <int MTW, int NTW, int ACT, bool HASY>
__global__ __launch_bounds__(64 * PWB_WAVES) void pwb_wgrad_kernel(const PwbWgArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* red = reinterpret_cast<float*>(smem);                           // [PWB_WAVES][MTW*NTW][64*16]  (one tile set per wave)
    const int tid = threadIdx.x, wave = cfn_uni(tid >> 6), lane = tid & 63, kg = lane >> 5, r = lane & 31;
    const int M = a.M, K = a.K, Q = a.Q;
    unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    const int strip = L % a.strips; L /= a.strips;
    const int kb = L % a.kblocks; L /= a.kblocks;
    const int mb = L % a.mblocks;
    const int n = L / a.mblocks;
    const int m0 = mb * 32 * MTW, k0 = kb * 32 * NTW;

    // per-lane row coefficients: A-operand rows m0 + 32 i + r (gs, 2 gq, gsc), B-operand rows k0 + 32 i + r (A, B)
    float cgs[MTW], cgq[MTW], cgc[MTW], cpa[NTW], cpb[NTW];
    int offm[MTW], offk[NTW];
    constexpr bool has_y = HASY;
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        const int m = m0 + 32 * i + r;
        const bool ok = m < M;
        cgs[i] = (ok && a.gs) ? (float)a.gs[(long)n * M + m] : 0.0f;
        cgq[i] = (ok && has_y) ? 2.0f * (float)a.gq[(long)n * M + m] : 0.0f;
        cgc[i] = (ok && a.gsc) ? (float)a.gsc[(long)n * M + m] : 1.0f;
        offm[i] = ok ? (m * Q + kg * 8) * 2 : PWB_OOB;
    }
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int k = k0 + 32 * i + r;
        const bool ok = k < K;
        cpa[i] = (ok && a.pa) ? (float)a.pa[(long)n * K + k] : 1.0f;
        cpb[i] = (ok && a.pb) ? (float)a.pb[(long)n * K + k] : 0.0f;
        offk[i] = ok ? (k * Q + kg * 8) * 2 : PWB_OOB;
    }
    __amdgpu_buffer_rsrc_t rg = cfn_rsrc(const_cast<uint16_t*>(a.gy + (long)n * M * Q), (unsigned)((long)M * Q * 2));
    __amdgpu_buffer_rsrc_t ry = cfn_rsrc(const_cast<uint16_t*>((has_y ? a.y : a.gy) + (long)n * M * Q), (unsigned)((long)M * Q * 2));
    __amdgpu_buffer_rsrc_t rx = cfn_rsrc(const_cast<uint16_t*>(a.x + (long)n * K * Q), (unsigned)((long)K * Q * 2));

    f16v acc[MTW][NTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int jn = 0; jn < NTW; ++jn) acc[i][jn] = (f16v)0.0f;

    // positions in steps of 16 (Q % 8 == 0: a lane's 8 positions never straddle the row end; the range check of the row
    // block does not protect rows m < M-1, so the tail step masks by position)
    const int nsteps = (Q + 15) >> 4;
    const int per = (nsteps + a.strips * PWB_WAVES - 1) / (a.strips * PWB_WAVES);
    const int s0 = (strip * PWB_WAVES + wave) * per, s1 = min(s0 + per, nsteps);
    // software pipeline: the operand loads of step s+1 are in flight while step s is converted and multiplied
    // (unconditional buffer loads; a step beyond the strip reads nothing: out-of-range offsets)
    u4v ga[2][MTW], ya[2][MTW], xb[2][NTW];
    auto issue = [&](int s, u4v (&g)[MTW], u4v (&yy)[MTW], u4v (&xx)[NTW]) {
        const int p0 = s << 4;
        const bool pv = s < s1 && p0 + kg * 8 < Q;
        const int so = cfn_uni(s < s1 ? p0 * 2 : 0);                        // wave uniform; the lane's own validity is in the vector offset
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            g[i] = __builtin_bit_cast(u4v, __builtin_amdgcn_raw_buffer_load_b128(rg, pv ? offm[i] : PWB_OOB, so, 0));
            if (has_y) yy[i] = __builtin_bit_cast(u4v, __builtin_amdgcn_raw_buffer_load_b128(ry, pv ? offm[i] : PWB_OOB, so, 0));
        }
#pragma unroll
        for (int i = 0; i < NTW; ++i) xx[i] = __builtin_bit_cast(u4v, __builtin_amdgcn_raw_buffer_load_b128(rx, pv ? offk[i] : PWB_OOB, so, 0));
    };
    auto compute = [&](int s, const u4v (&g)[MTW], const u4v (&yy)[MTW], const u4v (&xx)[NTW]) {
        const bool pv = (s << 4) + kg * 8 < Q;
        bf16x8 Aop[MTW], Bop[NTW];
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            u4v o;
            const bool live = pv && offm[i] != PWB_OOB;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float lo = fmaf(pwb_lo(g[i][e]), cgc[i], cgs[i]), hi = fmaf(pwb_hi(g[i][e]), cgc[i], cgs[i]);
                if (has_y) { lo = fmaf(pwb_lo(yy[i][e]), cgq[i], lo); hi = fmaf(pwb_hi(yy[i][e]), cgq[i], hi); }
                o[e] = live ? pwb_pack(lo, hi) : 0u;                       // masked rows / positions contribute nothing
            }
            Aop[i] = __builtin_bit_cast(bf16x8, o);
        }
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            u4v o;
            const bool live = pv && offk[i] != PWB_OOB;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lo = cfn_act<ACT>(fmaf(pwb_lo(xx[i][e]), cpa[i], cpb[i])), hi = cfn_act<ACT>(fmaf(pwb_hi(xx[i][e]), cpa[i], cpb[i]));
                o[e] = live ? pwb_pack(lo, hi) : 0u;
            }
            Bop[i] = __builtin_bit_cast(bf16x8, o);
        }
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
            for (int jn = 0; jn < NTW; ++jn) acc[i][jn] = H16_MFMA32(Aop[i], Bop[jn], acc[i][jn], 0, 0, 0);
    };
    issue(s0, ga[0], ya[0], xb[0]);
    for (int s = s0; s < s1; s += 2) {
        issue(s + 1, ga[1], ya[1], xb[1]);
        compute(s, ga[0], ya[0], xb[0]);
        issue(s + 2, ga[0], ya[0], xb[0]);
        if (s + 1 < s1) compute(s + 1, ga[1], ya[1], xb[1]);
    }

    // combine the 8 waves (fixed order) and add into gw: tile element (row = (e & 3) + 8 (e >> 2) + 4 kg, col = r)
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int jn = 0; jn < NTW; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) red[((wave * MTW * NTW + i * NTW + jn) * 16 + e) * 64 + lane] = acc[i][jn][e];
    __syncthreads();
    for (int idx = tid; idx < MTW * NTW * 16 * 64; idx += 64 * PWB_WAVES) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < PWB_WAVES; ++w) t += red[w * MTW * NTW * 1024 + idx];
        const int ln = idx & 63, e = (idx >> 6) & 15, tl = idx >> 10;
        const int i = tl / NTW, jn = tl - i * NTW;
        const int m = m0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * (ln >> 5), k = k0 + 32 * jn + (ln & 31);
        if (m < M && k < K) cfn_add64(&a.gw[(long)m * K + k], (double)t);
    }
}

extern "C" int H16N(cfn_pwconv_bwd_weight)(const uint16_t* gy, const uint16_t* y, const double* gsum, const double* gsumsq,
                                          const uint16_t* x, const double* A, const double* B, int act, double* gw, int N,
                                          int Cin, int Cout, long Q, const double* gscale, void* stream) {
    CFN_REQUIRE(gy && x && gw, "cfn_pwconv_bwd_weight_" H16_NAME ": null tensor");
    CFN_REQUIRE((A == nullptr) == (B == nullptr), "cfn_pwconv_bwd_weight_" H16_NAME ": A/B mismatch");
    CFN_REQUIRE(gsumsq == nullptr || y != nullptr, "cfn_pwconv_bwd_weight_" H16_NAME ": gsumsq needs y");
    CFN_REQUIRE(Q > 0 && Q % 8 == 0, "cfn_pwconv_bwd_weight_" H16_NAME ": T*H*W = %ld must be a multiple of 8 (16-byte operand loads)", Q);
    CFN_REQUIRE((long)Cout * Q * 2 < 0x3ffffff0L && (long)Cin * Q * 2 < 0x3ffffff0L, "cfn_pwconv_bwd_weight_" H16_NAME ": sample block exceeds the 1 GiB buffer range");
    PwbWgArgs a = {};
    a.gy = gy; a.y = gsumsq ? y : nullptr; a.gs = gsum; a.gq = gsumsq; a.gsc = gscale; a.x = x; a.pa = A; a.pb = B; a.gw = gw;
    a.N = N; a.M = Cout; a.K = Cin; a.Q = (int)Q; a.act = A ? act : CFN_ACT_NONE;
    // 2x2 tiles of 32x32 per wave (64 accumulator registers): each operand row is re-read by the other dimension's blocks
    const int MTW = Cout <= 32 ? 1 : 2, NTW = Cin <= 32 ? 1 : 2;
    a.mblocks = cfn_cdiv(Cout, 32 * MTW);
    a.kblocks = cfn_cdiv(Cin, 32 * NTW);
    const long groups = (long)N * a.mblocks * a.kblocks;
    const int nsteps = (int)((Q + 15) / 16);
    static const int wgs_env = getenv("CFN_PWB_WG_WGS") ? atoi(getenv("CFN_PWB_WG_WGS")) : 0;
    long strips = ((wgs_env > 0 ? wgs_env : 1024) + groups - 1) / groups;
    const long maxs = cfn_cdiv(nsteps, PWB_WAVES * 4);
    if (strips > maxs) strips = maxs;
    if (strips < 1) strips = 1;
    a.strips = (int)strips;
    const size_t lds = (size_t)PWB_WAVES * MTW * NTW * 1024 * 4;
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_PWCONV_WGRAD, st, 2.0 * N * ((double)Cout * Q * (a.y ? 2 : 1) + (double)Cin * Q));
#ifndef CFN_F16
    {   // LDS-staged kernel (pwsplitw.hip): whole-line loads, operands formed once per element (bf16 tensors; the fp16 build keeps the direct kernel)
        const int rc = pwss_wgrad_try_bf16(gy, gsumsq ? y : nullptr, gsum, gsumsq, gscale, x, A, B, A ? act : CFN_ACT_NONE, gw, N, Cout, Cin, (int)Q,
                                           st);
        if (rc >= 0) return rc;
    }
#endif
    const dim3 grid((unsigned)(groups * strips));
#define PWB_WG_GO(MV, NV, AV)                                                                                              \
    do {                                                                                                                   \
        if (a.y) PWB_WG_GO2(MV, NV, AV, true); else PWB_WG_GO2(MV, NV, AV, false);                                         \
    } while (0)
#define PWB_WG_GO2(MV, NV, AV, HY)                                                                                         \
    do {                                                                                                                   \
        auto k = pwb_wgrad_kernel<MV, NV, AV, HY>;                                                                             \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(k, grid, dim3(64 * PWB_WAVES), lds, st, a);                                                     \
    } while (0)
#define PWB_WG_ACT(MV, NV)                                                                                                 \
    do {                                                                                                                   \
        if (a.act == CFN_ACT_RELU) PWB_WG_GO(MV, NV, CFN_ACT_RELU);                                                        \
        else if (a.act == CFN_ACT_SWISH) PWB_WG_GO(MV, NV, CFN_ACT_SWISH);                                                 \
        else PWB_WG_GO(MV, NV, CFN_ACT_NONE);                                                                              \
    } while (0)
    if (MTW == 1 && NTW == 1) PWB_WG_ACT(1, 1);
    else if (MTW == 1) PWB_WG_ACT(1, 2);
    else if (NTW == 1) PWB_WG_ACT(2, 1);
    else PWB_WG_ACT(2, 2);
#undef PWB_WG_ACT
#undef PWB_WG_GO
#undef PWB_WG_GO2
    return cfn_check_launch("pwconv wgrad " H16_NAME);
}

// ---------------------------------------------------------------------------------------------------------------------
// Spatial subsampling x[..., ::s, ::s] of a bf16 tensor: the strided shortcut conv (x3d_fine.py:284-287) becomes this
// gather (reads the even rows, writes 1/4 of the tensor) followed by the stride-1 contraction above.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void subsample_hw_bf16_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ out, int H, int W,
                                                                 int Ho, int Wo, int s, long total) {
    const long e = ((long)blockIdx.x * 256 + threadIdx.x) * 2;              // two outputs per thread: one dword store
    if (e >= total) return;
    const int wo = (int)(e % Wo), ho = (int)((e / Wo) % Ho);
    const long pl = e / ((long)Wo * Ho);                                    // (n, c, t) plane
    const uint16_t* row = x + (pl * H + (long)ho * s) * W;
    const unsigned lo = row[(long)wo * s];
    unsigned hi = 0;
    if (e + 1 < total) {
        const long e1 = e + 1;
        const int wo1 = (int)(e1 % Wo), ho1 = (int)((e1 / Wo) % Ho);
        const long pl1 = e1 / ((long)Wo * Ho);
        hi = x[(pl1 * H + (long)ho1 * s) * W + (long)wo1 * s];
    }
    if (e + 1 < total) *reinterpret_cast<unsigned*>(out + e) = lo | (hi << 16);
    else out[e] = (uint16_t)lo;
}

extern "C" int H16N(cfn_subsample_hw)(const uint16_t* x, uint16_t* out, long planes, int H, int W, int s, void* stream) {
    CFN_REQUIRE(x && out && planes > 0 && H > 0 && W > 0 && s >= 1, "cfn_subsample_hw_" H16_NAME ": bad arguments");
    const int Ho = (H - 1) / s + 1, Wo = (W - 1) / s + 1;
    const long total = planes * Ho * Wo;
    hipStream_t st = (hipStream_t)stream;
    CfnProfScope prof(CFN_K_ELEMWISE, st, 2.0 * planes * ((double)H * W / s + (double)Ho * Wo));
    hipLaunchKernelGGL(subsample_hw_bf16_kernel, dim3((unsigned)cfn_cdiv(cfn_cdiv(total, 2), 256)), dim3(256), 0, st, x, out, H, W, Ho, Wo, s, total);
    return cfn_check_launch("subsample_hw " H16_NAME);
}
