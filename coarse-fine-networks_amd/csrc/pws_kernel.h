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
#pragma once
#include "pw_common.h"
#include <stdlib.h>
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

#define PWS_WAVES 8
#define PWS_NSET 3
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

// PwArgs fields re-used by the plan: Kpad = K padded to 48 (3 k-blocks), mtiles = row slabs, nstrips = workgroups per (n, slab),
// kres = LDS bytes per weight row of ONE split image.
// NP = positions per lane: 2 (64-position wave tiles, 8-byte accesses) or 1 (32-position tiles, 4-byte accesses: half the
// accumulators per row tile, so a slab of up to 7 row tiles = 224 rows fits the registers and a 216-row layer needs ONE slab --
// a second slab re-reads every activation, and re-reads cost the vector-memory path as much as HBM reads do).
// KCH > 0 (pwstream.hip): WEIGHT STREAMING for contractions too deep for a resident slab (K = 432: three images of 192 rows are 498 KB).
// a.w then is the pre-split workspace `[slab][chunk][term][BM rows][KCH * 32 + 16 bytes]` (pws_presplit_kernel: the byte image of one LDS
// chunk buffer); the kernel keeps TWO chunk buffers of KCH k-blocks, every thread copies its share of chunk g + 1 (16-byte loads at the
// head of each k-block step, written to LDS behind the step's MFMAs) while chunk g is multiplied; ONE barrier per chunk.  The tile loop
// runs the same number of rounds in every wave of a workgroup (a wave without a tile only copies); the weights of a tile round are the
// same as the last one's, so the chunk sequence is periodic and the copy runs straight on into the next round.
template <int MT, int NP, int MODE, bool STATS, int ACT, bool TWO, int NS, int KCH = 0>
__global__ __launch_bounds__(64 * PWS_WAVES) void pws_kernel(const PwArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BM = 32 * MT;
    constexpr int TP = 32 * NP;                                             // positions per wave tile
    const int tid = threadIdx.x, wave = cfn_uni(tid >> 6), lane = tid & 63, kg = lane >> 5, j = lane & 31;
    constexpr bool STREAM = KCH > 0;
    const int K = a.K, M = a.M, Q = a.Q, Kp = a.Kpad, rowb = STREAM ? KCH * 32 + 16 : a.kres;

    unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    const int slab = L % a.mtiles; L /= a.mtiles;     // slabs of the same positions run side by side: re-reads hit the XCD's L2
    const int wg = L % a.nstrips;
    const int n = L / a.nstrips;
    const int m0 = slab * BM;

    unsigned char* Ws = smem;                                              // [NS][BM][rowb] bf16 weight images (STREAM: two chunk buffers of that shape)
    const size_t img = (size_t)BM * rowb;
    float4* sP = reinterpret_cast<float4*>(Ws + (STREAM ? 2 : 1) * NS * img);   // [Kp] prologue coefficients
    float2* sE = reinterpret_cast<float2*>(sP + Kp);                       // [BM] epilogue coefficients (DGRAD)
    float* red = reinterpret_cast<float*>(sE + BM);                        // [PWS_WAVES][32][20] transpose scratch, then [PWS_WAVES][BM][2]

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
    // weight images: Ws[s][m][k] = term s of W[m0+m][k] (FWD, w is (M,K)) or of W[k][m0+m] (DGRAD, w is (K,M)); zero padded.
    // Batches of 8 pairs per thread: all 16 loads in flight, then the splits and LDS writes (a dependent load -> store loop costs one
    // L2 round trip per iteration: 20-50 iterations at the start of EVERY workgroup)
    if constexpr (!STREAM) {
        constexpr int UB = 8, NT = 64 * PWS_WAVES;
        const int total = BM * (Kp / 2);
        for (int e0 = tid; e0 < total; e0 += NT * UB) {
            float v0[UB], v1[UB];
            int off[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int e = e0 + u * NT;
                int m, k2;
                if (MODE == PW_FWD) { m = e / (Kp / 2); k2 = (e - m * (Kp / 2)) * 2; }     // consecutive threads along k (w rows)
                else { k2 = (e / BM) * 2; m = e - (e / BM) * BM; }                          // consecutive threads along m (w rows)
                v0[u] = v1[u] = 0.0f;
                off[u] = e < total ? m * rowb + k2 * 2 : -1;
                if (e < total && m0 + m < M) {
                    if (MODE == PW_FWD) {
                        if (k2 < K) v0[u] = a.w[(long)(m0 + m) * a.Cin + k2];
                        if (k2 + 1 < K) v1[u] = a.w[(long)(m0 + m) * a.Cin + k2 + 1];
                    } else {
                        if (k2 < K) v0[u] = a.w[(long)k2 * a.Cin + m0 + m];
                        if (k2 + 1 < K) v1[u] = a.w[(long)(k2 + 1) * a.Cin + m0 + m];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                if (off[u] >= 0) {
                    unsigned p[NS];
                    pws_split<NS>(v0[u], v1[u], p);
#pragma unroll
                    for (int sp = 0; sp < NS; ++sp) *reinterpret_cast<unsigned*>(Ws + sp * img + off[u]) = p[sp];
                }
            }
        }
    }
    // STREAM: the copy of one chunk = CU 16-byte units, UPT per thread, in three parts (one per k-block step of the chunk being multiplied)
    constexpr int CU = STREAM ? NS * BM * (KCH * 2 + 1) : 1, UPT = (CU + 64 * PWS_WAVES - 1) / (64 * PWS_WAVES), UPP = (UPT + 2) / 3;
    const int nchunks = STREAM ? Kp / (16 * KCH) : 1;
    const u4v* wsrc = reinterpret_cast<const u4v*>(a.w) + (size_t)slab * nchunks * CU;
    u4v cw[STREAM ? UPP : 1];
    auto copy_load = [&](int c, int part) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < UPP; ++u) {
            const int idx = (part * UPP + u) * (64 * PWS_WAVES) + tid;
            if (part * UPP + u < UPT && idx < CU) cw[u] = wsrc[(size_t)c * CU + idx];
        }
    };
    auto copy_store = [&](int buf, int part) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < UPP; ++u) {
            const int idx = (part * UPP + u) * (64 * PWS_WAVES) + tid;
            if (part * UPP + u < UPT && idx < CU) *reinterpret_cast<u4v*>(Ws + (size_t)buf * NS * img + (size_t)idx * 16) = cw[u];
        }
    };
    if constexpr (STREAM) {
#pragma unroll
        for (int part = 0; part < 3; ++part) { copy_load(0, part); copy_store(0, part); }
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
    float ssum[MT], qsum[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) ssum[mt] = qsum[mt] = 0.0f;

    const int ntiles = (Q + TP - 1) / TP, nkb = Kp >> 4;
    const int lane_voff = kg * 8 * Q * 4 + j * 4 * NP;                      // this lane's (channel group, position [pair]) offset
    const unsigned char* wrow0 = Ws + (size_t)j * rowb + kg * 16;            // A operand: row j (+32*mt), k = kb*16 + kg*8 ..
    const unsigned char* wrow = wrow0;                                      // (STREAM: + the chunk buffer being multiplied)

    // element p (< NP) of a loaded / stored position group
    auto ldp = [&](__amdgpu_buffer_rsrc_t r, int vo, int so, float (&out)[NP]) {
        if constexpr (NP == 2) {
            const u2v d = __builtin_amdgcn_raw_buffer_load_b64(r, vo, so, 0);
            out[0] = pws_f(d.x); out[1] = pws_f(d.y);
        } else {
            out[0] = pws_f(__builtin_amdgcn_raw_buffer_load_b32(r, vo, so, 0));
        }
    };

    // Operand ring of PWS_NSET = 3 load sets with STATIC slots (k-block kb of a tile lives in slot kb % 3; Kp is a multiple of 48 so
    // every tile starts at slot 0) and an issue cursor that runs TWO k-blocks ahead of the arithmetic and straight on into the
    // wave's next tile: with one set ahead (round-3 first version) a CU had 16-32 KB in flight -- by Little's law 2-4 TB/s at the
    // loaded HBM latency -- and the pipeline drained at every tile end.  Unconditional loads (exact vmcnt waits).  The hardware
    // checks  voffset >= num_records - soffset: the scalar part must never exceed the range (it would wrap), so a k-block that
    // starts beyond K (or a tile beyond the last) is switched off through the lane offset; channels >= K inside a live block fall
    // out of range by themselves and read as 0.
    float ld[PWS_NSET][8][NP], ld2[PWS_NSET][8][NP];
    const int tstep = a.nstrips * PWS_WAVES;
    int itile = wg * PWS_WAVES + wave, ikb = 0;                             // issue cursor (wave uniform)
    auto issue_next = [&](float (&d)[8][NP], float (&d2)[8][NP]) {
        const bool tlive = itile < ntiles;
        const int vo = tlive ? lane_voff : PWS_OOB;
        const int base = tlive ? itile * TP * 4 : 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // the row part of the address is SCALAR (s_cselect / s_mul: no VALU).  A row at or beyond K must not enter the scalar
            // offset (see above): it is replaced by row 0 -- the lanes then read finite activations of rows 0 / 8, which meet
            // the zero-padded weight columns k >= K
            const int row = ikb * 16 + i;
            const int so = row < K ? row * Q * 4 + base : base;
            ldp(rs1, vo, so, d[i]);
            if (two_src) ldp(rs2, vo, so, d2[i]);
        }
        if (++ikb == nkb) { ikb = 0; itile += tstep; }
    };
    issue_next(ld[0], ld2[0]);
    issue_next(ld[1], ld2[1]);

    const int rounds = STREAM ? (ntiles - wg * PWS_WAVES + tstep - 1) / tstep : 0;   // the same in every wave of the workgroup
    int gc = 0;                                                             // (STREAM) chunks multiplied so far: chunk g sits in buffer g & 1
    for (int tile = wg * PWS_WAVES + wave, rnd = 0; STREAM ? rnd < rounds : tile < ntiles; tile += tstep, ++rnd) {
        const bool live = !STREAM || tile < ntiles;                         // wave uniform; a wave without a tile still copies and meets the barriers
        const int q0 = tile * TP;
        const bool cv = q0 + NP * j < Q;                                     // NP == 2: Q is even, a pair is valid or not as a whole
        f16v acc[MT][NP];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int p = 0; p < NP; ++p) acc[mt][p] = (f16v)0.0f;

        // FORWARD output side (round 3, second pass): the ACTIVATIONS are the MFMA's A operand and the weights its B operand, so the 32x32
        // result arrives transposed -- lane (j, kg) holds output channel mt*32 + j and, in registers 4 g + i, tile rows (= positions)
        // 8 g + 4 kg + i: four consecutive positions of ONE channel per register group.  Per-channel statistics are then in-lane
        // work (no LDS transpose of partial sums, no shuffles: the old
        // epilogue, one position x 16 rows per lane, was ~1,500 of the ~3,500 instructions per tile, with SGPR spills, and these
        // kernels are instruction-issue bound at 2 waves per SIMD).  Memory wants the other layout (a 16-byte access per lane over
        // 32 different rows = 64 separate requests per instruction: measured 10-35 % slower than the old kernel), so tensor data
        // crosses a wave-private LDS scratch in UNITS of 32 channels x 16 positions: 16-byte LDS accesses both ways, and in the
        // memory-side layout lane l owns row (l >> 2) + 16 s, 16 bytes at position 4 (l & 3): four lanes cover 64 contiguous bytes,
        // an instruction 16 rows -- 2 stores per unit instead of 8 four-byte ones, all whole 64-byte segments.
        // The DATA GRADIENT keeps the weights as the A operand and the epilogue below it: the same scheme with the act' epilogue's
        // forward input crossing the scratch the other way was built and produced run-to-run different gradients (one position of
        // 16 channels of a tile's first row tile, 15-50 % of launches; tools/ab_pw.py) that full vmcnt / lgkmcnt drains, wait states
        // around the wide LDS / buffer accesses, the MFMA groups and the transcendentals did not remove -- not in the tree.
        constexpr int UPT = 2 * NP;                                         // units per row tile
        float* scr = red + wave * (32 * 20);                                // [32 channels][16 positions + 4 pad]
        const int mrow = lane >> 2, mcol = 4 * (lane & 3);                  // memory-side role of this lane
        const int lane_mem = mrow * Q * 4 + mcol * 4;                       // + 16 rows for s = 1
        // Row addressing of the epilogue: the lane part (column, kg's 4-row offset) sits in the vector offset, the
        // wave-uniform row base in the scalar offset; a row base beyond the slab's valid rows would push the scalar offset
        // past the range (which wraps instead of failing the check), so such rows are switched off through the vector offset.
        const int cvk = cv ? (q0 + NP * j) * 4 + 4 * kg * Q * 4 : PWS_OOB;
        auto rowbase = [&](int mt, int r) { return mt * 32 + (r & 3) + 8 * (r >> 2); };
        auto compute = [&](int kb, int kbw, const float (&d)[8][NP], const float (&d2)[8][NP]) {   // kbw: the k-block's place in the weight image
            float v[NP][8];
            const float4* cp = sP + kb * 16 + kg * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4 c = cp[i];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    float e = d[i][p];
                    if (MODE == PW_FWD) {
                        e = cfn_act<ACT>(fmaf(e, c.x, c.y));
                    } else {
                        e = fmaf(e, c.z, c.x);
                        if (two_src) e = fmaf(d2[i][p], c.y, e);
                    }
                    v[p][i] = e;
                }
            }
            u4v pb[NP][NS];
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    unsigned sp[NS];
                    pws_split<NS>(v[p][2 * h], v[p][2 * h + 1], sp);
#pragma unroll
                    for (int s = 0; s < NS; ++s) pb[p][s][h] = sp[s];
                }
            // All MT row tiles, branch free (a ragged last slab multiplies its zero-padded weight rows: a block-uniform skip per
            // tile put every tile's MFMAs in a basic block of their own, each opening with its ds_read + s_waitcnt lgkmcnt).
            // The A operand of tile mt+1 is read from LDS before the MFMAs of tile mt issue (two static register sets); the
            // 3 / 6 MFMAs of a term set chain on one accumulator (srcC = vDst of the predecessor: forwarded, no wait states).
            bf16x8 A[2][NS];
            auto lda = [&](bf16x8 (&dst)[NS], int mt) {
#pragma unroll
                for (int s = 0; s < NS; ++s) dst[s] = *reinterpret_cast<const bf16x8*>(wrow + s * img + (size_t)mt * 32 * rowb + kbw * 32);
            };
            lda(A[0], 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (mt + 1 < MT) lda(A[(mt + 1) & 1], mt + 1);
                __builtin_amdgcn_sched_barrier(0);
#define PWS_MM(SA, SB)                                                                                                     \
                _Pragma("unroll") for (int p = 0; p < NP; ++p)                                                             \
                    acc[mt][p] = MODE == PW_FWD ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pb[p][SB]), A[mt & 1][SA], acc[mt][p], 0, 0, 0) \
                                            : __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt & 1][SA], __builtin_bit_cast(bf16x8, pb[p][SB]), acc[mt][p], 0, 0, 0)
                if constexpr (NS == 3) { PWS_MM(2, 0); PWS_MM(0, 2); PWS_MM(1, 1); }
                PWS_MM(1, 0); PWS_MM(0, 1); PWS_MM(0, 0);
#undef PWS_MM
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        for (int kb = 0; kb < nkb; kb += PWS_NSET) {                        // Kp is a multiple of 48: nkb is a multiple of 3
            if constexpr (!STREAM) {
                issue_next(ld[2], ld2[2]);
                compute(kb, kb, ld[0], ld2[0]);
                issue_next(ld[0], ld2[0]);
                compute(kb + 1, kb + 1, ld[1], ld2[1]);
                issue_next(ld[1], ld2[1]);
                compute(kb + 2, kb + 2, ld[2], ld2[2]);
            } else {
                static_assert(!STREAM || KCH == PWS_NSET, "a chunk = one trip of the operand ring");
                __syncthreads();                                            // chunk gc is complete in its buffer; the other buffer is free
                const int cn = kb + PWS_NSET < nkb ? kb / PWS_NSET + 1 : 0;  // the chunk after this one (the next round starts over)
                const bool more = kb + PWS_NSET < nkb || rnd + 1 < rounds;
                const int nb = (gc + 1) & 1;
                const unsigned char* wr = wrow0 + (size_t)(gc & 1) * NS * img;
                auto part = [&](int p, float (&li)[8][NP], float (&li2)[8][NP], const float (&lc)[8][NP], const float (&lc2)[8][NP]) __attribute__((always_inline)) {
                    if (more) copy_load(cn, p);
                    if (live) {
                        issue_next(li, li2);
                        wrow = wr;
                        compute(kb + p, p, lc, lc2);
                    }
                    if (more) copy_store(nb, p);
                };
                part(0, ld[2], ld2[2], ld[0], ld2[0]);
                part(1, ld[0], ld2[0], ld[1], ld2[1]);
                part(2, ld[1], ld2[1], ld[2], ld2[2]);
                ++gc;
            }
        }
        if (!live) continue;

        // ---- forward epilogue (transposed C layout, see above)
        auto wsync = [&]() {                                           // LDS ops of a wave run in order; only the compiler is told
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        };
        // memory-side offsets of unit u of row tile mt: rows mt*32 + mrow (+16), positions q0 + 16 u + mcol .. +3
        auto mem_vo = [&](auto full_tag, int mt, int u, int s) {
            constexpr bool FULL = decltype(full_tag)::value;
            const bool ok = FULL || (mt * 32 + 16 * s + mrow < mrows && q0 + 16 * u + mcol < Q);
            return ok ? lane_mem + s * 16 * Q * 4 : PWS_OOB;
        };
        auto epilogue_fwd = [&](auto full_tag, int mt) {
            constexpr bool FULL = decltype(full_tag)::value;          // all 32 channels and all 32 NP positions of the tile exist
            const bool chv = FULL || mt * 32 + j < mrows;
            const int so = (mt * 32 * Q + q0) * 4;                     // wave uniform
            float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
            for (int u = 0; u < UPT; ++u) {
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    // NP == 1: slot sl = register group g = 2 u + sl (positions 8 g + 4 kg + i);  NP == 2: g = u, the lane's 8 consecutive
                    // positions interleave the even-position tile (p = 0) and the odd one: slot sl = (even, odd) of rows 2 sl, 2 sl + 1
                    const int pos0 = q0 + 16 * u + (NP == 1 ? 8 * sl + 4 * kg : 8 * kg + 4 * sl);
                    const float gmask = (chv && (FULL || pos0 < Q)) ? 1.0f : 0.0f;
                    f4v o;
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        const int r = NP == 1 ? 4 * (2 * u + sl) + e4 : 4 * u + 2 * sl + (e4 >> 1);
                        float e = acc[mt][NP == 1 ? 0 : (e4 & 1)][r];
                        if (STATS) {
                            const float em = FULL ? e : e * gmask;
                            t1 += em;
                            t2 = fmaf(em, em, t2);
                        }
                        o[e4] = e;
                    }
                    *reinterpret_cast<f4v*>(scr + j * 20 + (NP == 1 ? 8 * sl + 4 * kg : 8 * kg + 4 * sl)) = o;
                }
                wsync();
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const f4v v = *reinterpret_cast<const f4v*>(scr + (mrow + 16 * s) * 20 + mcol);
                    cfn_bst128(__builtin_bit_cast(u4v, v), rd, mem_vo(full_tag, mt, u, s) + u * 64, so);
                }
                wsync();
            }
            if (STATS) { ssum[mt] += t1; qsum[mt] += t2; }
        };
        // ---- epilogue: C layout of the 32x32 tile: column = lane & 31 (position [pair] j), row = (r & 3) + 8 (r >> 2) + 4 kg
        // FULL (wave uniform): all 32 rows of the tile exist and all 32 NP columns lie inside the row -- no per-row / per-column
        // selects on the store offsets and the statistics operands (38 v_cndmask per row tile otherwise)
        auto epilogue = [&](auto full_tag, int mt) {
            constexpr bool FULL = decltype(full_tag)::value;
            // DGRAD epilogue operands (forward input x of the output rows, for act'): the 16 row loads of this 32-row tile go
            // out as ONE batch (next to their consumers they would cost one HBM round trip each)
            float xq[16][NP];
            if (MODE == PW_DGRAD && STATS) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool live = FULL || rowbase(mt, r) < mrows;
                    ldp(rx, live ? cvk : PWS_OOB, live ? rowbase(mt, r) * Q * 4 : 0, xq[r]);
                }
            }
            // Stores, and the per-row reductions (statistics forward; sum dz*x, sum dz backward) through a wave-private LDS
            // transpose, 16 rows at a time: 8 ds_write + 8 ds_read + ~16 VALU per half tile and quantity, where the shuffle
            // butterfly of pwbf16.hip costs ~100 VALU (the epilogue was half of this kernel's VALU instructions, and VALU, not the
            // matrix pipe, is what the 6-term product is short of).  Registers r = 8 hf + i hold tile rows 16 hf + (i & 3) +
            // 8 (i >> 2) + 4 kg; lane l then sums row l >> 2 over the 8 columns 8 (l & 3) .. +7 (conflict-free both ways at a
            // 33-float pitch), the quad adds its four partial sums and lane l keeps the total of row 16 (l & 1) + (l >> 2).
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                float t1[8], t2[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int r = 8 * hf + i;
                    const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                    float o[NP];
                    t1[i] = t2[i] = 0.0f;
                    const bool live = FULL || rowbase(mt, r) < mrows;
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        float e = acc[mt][p][r];
                        if (MODE == PW_FWD) {
                            if (STATS) {
                                const float em = (FULL || cv) ? e : 0.0f;
                                t1[i] += em;
                                if (NP == 2) t2[i] = fmaf(em, em, t2[i]);
                            }
                        } else if (STATS) {                                  // act' epilogue + prologue-coefficient gradients
                            // (NP = 2: the two positions' fma(x, A, B) become ONE packed FMA taking B from the high register of the pair just read)
                            const float2 c = NP == 2 ? cfn_settle(sE[row]) : sE[row];
                            const float xe = xq[r][p];
                            const float de = (FULL || cv) ? e * cfn_act_grad<ACT>(fmaf(xe, c.x, c.y)) : 0.0f;
                            t1[i] = fmaf(de, xe, t1[i]); t2[i] += de;
                            e = de * c.x;
                        }
                        o[p] = e;
                    }
                    if constexpr (NP == 2) {
                        const u2v st = {__builtin_bit_cast(unsigned, o[0]), __builtin_bit_cast(unsigned, o[NP - 1])};
                        __builtin_amdgcn_raw_buffer_store_b64(st, rd, live ? cvk : PWS_OOB, live ? rowbase(mt, r) * Q * 4 : 0, 0);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o[0]), rd, live ? cvk : PWS_OOB, live ? rowbase(mt, r) * Q * 4 : 0, 0);
                    }
                }
                if (STATS) {
                    float* scr = red + wave * (16 * 33);
                    auto keep = [&](float sm) {                              // the row's four partial sums: quad permutes;
                        sm += __shfl_xor(sm, 1, 64);                         // lane l keeps row 16 (l & 1) + (l >> 2)
                        sm += __shfl_xor(sm, 2, 64);
                        return (lane & 1) == hf ? sm : 0.0f;
                    };
                    if constexpr (MODE == PW_FWD && NP == 1) {               // one pass: the reader forms sum and sum of squares
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int i = 0; i < 8; ++i) scr[((i & 3) + 8 * (i >> 2) + 4 * kg) * 33 + j] = t1[i];
                        asm volatile("" ::: "memory");
                        float sm = 0.0f, sq = 0.0f;
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const float v = scr[(lane >> 2) * 33 + (lane & 3) * 8 + c];
                            sm += v; sq = fmaf(v, v, sq);
                        }
                        ssum[mt] += keep(sm); qsum[mt] += keep(sq);
                        asm volatile("" ::: "memory");
                    } else {
#pragma unroll
                        for (int pass = 0; pass < 2; ++pass) {
                            asm volatile("" ::: "memory");
#pragma unroll
                            for (int i = 0; i < 8; ++i) scr[((i & 3) + 8 * (i >> 2) + 4 * kg) * 33 + j] = pass == 0 ? t1[i] : t2[i];
                            asm volatile("" ::: "memory");
                            float sm = 0.0f;
#pragma unroll
                            for (int c = 0; c < 8; ++c) sm += scr[(lane >> 2) * 33 + (lane & 3) * 8 + c];
                            if (pass == 0) ssum[mt] += keep(sm); else qsum[mt] += keep(sm);
                            asm volatile("" ::: "memory");
                        }
                    }
                }
            }
        };
        const bool colfull = q0 + TP <= Q;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (m0 + mt * 32 >= M) continue;
            if constexpr (MODE == PW_FWD) {
                if (colfull && mrows - mt * 32 >= 32) epilogue_fwd(std::true_type{}, mt);
                else epilogue_fwd(std::false_type{}, mt);
            } else {
                if (colfull && mrows - mt * 32 >= 32) epilogue(std::true_type{}, mt);
                else epilogue(std::false_type{}, mt);
            }
        }
    }

    if (STATS && a.s1) {
        // `red` doubles as the waves' transpose scratch, hence the barrier before it is re-used for the cross-wave combine
        __syncthreads();
        if constexpr (MODE == PW_FWD) {
            // lane (j, kg) holds the sums of channel mt*32 + j over ITS positions: the two halves are added first
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                ssum[mt] += __shfl_xor(ssum[mt], 32, 64);
                qsum[mt] += __shfl_xor(qsum[mt], 32, 64);
            }
            if (kg == 0) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    red[(wave * BM + mt * 32 + j) * 2] = ssum[mt];
                    red[(wave * BM + mt * 32 + j) * 2 + 1] = qsum[mt];
                }
            }
        } else if ((lane & 2) == 0) {
            // lane l holds the sums of row 16 (l & 1) + (l >> 2) of every row tile (lanes with l & 2 hold copies)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int row = mt * 32 + 16 * (lane & 1) + (lane >> 2);
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
                cfn_add64(&a.s1[(long)n * M + m0 + m], (double)t1);
                cfn_add64(&a.s2[(long)n * M + m0 + m], (double)t2);
            }
        }
    }
}

