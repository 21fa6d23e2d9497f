// Pointwise (1x1x1) channel contractions, MFMA-bound shapes (x3d_fine.py:100-105 conv1/conv3 of layers 2-4).
#include "pw_common.h"
#include <stdlib.h>

// ---------------------------------------------------------------------------------------------
// "Deep" variant for the MFMA-bound contractions (K >= 48: layers 2-4).  A wave owns ALL BM = 32*MT output rows
// (MT <= 7) of its 32 positions, so every activation is loaded from HBM and pushed through the prologue exactly once
// and feeds MT back-to-back MFMAs (the 32-row kernel above re-reads and re-activates x once per 32 output rows).
// 8 waves per workgroup share one resident weight image As[k][BM] (<= 120 KiB of LDS, one workgroup per CU, two
// waves per SIMD); accumulators are MT x 16 registers.  No workgroup barrier inside the tile loop.
// ---------------------------------------------------------------------------------------------

template <int MT, int MODE, bool STATS, int ACT, int PWD_WAVES>
__global__ __launch_bounds__(64 * PWD_WAVES) void pw_deep_kernel(const PwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BM = 32 * MT;
    constexpr int NU = PW_UNIT / 2;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, col = lane & 31;
    const int K = a.K, M = a.M, Q = a.Q, Kpad = a.Kpad;

    unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    const int mtile = L % a.mtiles; L /= a.mtiles;
    const int strip = L % a.nstrips;
    const int n = L / a.nstrips;
    const int m0 = mtile * BM;

    float* As = smem;                                         // [Kpad][BM]
    float4* sP = reinterpret_cast<float4*>(As + Kpad * BM);   // [Kpad + PW_UNIT] prologue coefficients (FWD: A, B; DGRAD: gs, 2gq, gsc)
    float2* sE = reinterpret_cast<float2*>(sP + Kpad + PW_UNIT);   // [BM] epilogue coefficients (DGRAD)
    float* redbase = reinterpret_cast<float*>(sE + BM);       // [PWD_WAVES][32*33] transpose scratch, later wave slots
    float* red = redbase + wave * (32 * PW_RED_PITCH);

    for (int k = tid; k < Kpad + PW_UNIT; k += 64 * PWD_WAVES) {
        float4 c;
        c.z = 1.0f; c.w = 0.0f;
        if (MODE == PW_FWD) {
            c.x = (k < K && a.pa) ? a.pa[(long)n * K + k] : 1.0f;
            c.y = (k < K && a.pb) ? a.pb[(long)n * K + k] : 0.0f;
        } else {
            c.x = (k < K && a.gs) ? (float)a.gs[(long)n * K + k] : 0.0f;
            c.y = (k < K && a.gq && a.src2) ? 2.0f * (float)a.gq[(long)n * K + k] : 0.0f;
            c.z = (k < K && a.gsc) ? (float)a.gsc[(long)n * K + k] : 1.0f;
        }
        sP[k] = c;
    }
    for (int m = tid; m < BM; m += 64 * PWD_WAVES) {
        const bool ok = (m0 + m) < M && MODE == PW_DGRAD && a.ea;
        float2 c;
        c.x = ok ? a.ea[(long)n * M + m0 + m] : 1.0f;
        c.y = ok ? a.eb[(long)n * M + m0 + m] : 0.0f;
        sE[m] = c;
    }
    // As[k][m] = Wm[m0+m][k], zero padded.  The image is loaded in two phases (all global loads in flight, then all
    // LDS writes): a dependent load->store loop costs one L2 round trip per iteration (measured 9 us for 86 KB).
    //   FWD: w is (M,K) row major -> float4 along k (whole lines per row), 4 scalar LDS writes;
    //   DGRAD: w is (K,M)          -> float4 along m on both sides.
    {
        constexpr int NTHR = 64 * PWD_WAVES;
        constexpr int MAXV = (120 * 1024 / 16 + NTHR - 1) / NTHR;           // float4 per thread for the largest image
        const bool v4 = (a.Cin % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.w) & 15) == 0) && (MODE == PW_FWD || m0 % 4 == 0);
        if (v4) {
            const int inner = (MODE == PW_FWD ? Kpad : BM) / 4;                // float4 groups along the contiguous axis
            const int total = (MODE == PW_FWD ? BM : Kpad) * inner;
            f4v wv[MAXV];
#pragma unroll
            for (int it = 0; it < MAXV; ++it) {
                const int e = tid + it * NTHR;
                wv[it] = (f4v){0.f, 0.f, 0.f, 0.f};
                if (e < total) {
                    const int o = e / inner, i4 = (e - o * inner) * 4;
                    if (MODE == PW_FWD) {          // o = m, i4 = k
                        if (m0 + o < M && i4 < K) wv[it] = *reinterpret_cast<const f4v*>(a.w + (long)(m0 + o) * a.Cin + i4);
                        if (i4 + 3 >= K) {         // ragged tail of the row (K % 4 == 0 here, so only the Kpad padding)
                            if (i4 + 0 >= K) wv[it].x = 0.f;
                            if (i4 + 1 >= K) wv[it].y = 0.f;
                            if (i4 + 2 >= K) wv[it].z = 0.f;
                            if (i4 + 3 >= K) wv[it].w = 0.f;
                        }
                    } else {                       // o = k, i4 = m
                        if (o < K && m0 + i4 < M) {
                            wv[it] = *reinterpret_cast<const f4v*>(a.w + (long)o * a.Cin + m0 + i4);
                            if (m0 + i4 + 1 >= M) wv[it].y = 0.f;
                            if (m0 + i4 + 2 >= M) wv[it].z = 0.f;
                            if (m0 + i4 + 3 >= M) wv[it].w = 0.f;
                        }
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < MAXV; ++it) {
                const int e = tid + it * NTHR;
                if (e < total) {
                    const int o = e / inner, i4 = (e - o * inner) * 4;
                    if (MODE == PW_FWD) {
                        As[(i4 + 0) * BM + o] = wv[it].x;
                        As[(i4 + 1) * BM + o] = wv[it].y;
                        As[(i4 + 2) * BM + o] = wv[it].z;
                        As[(i4 + 3) * BM + o] = wv[it].w;
                    } else {
                        *reinterpret_cast<f4v*>(As + o * BM + i4) = wv[it];
                    }
                }
            }
        } else {
            for (int e = tid; e < Kpad * BM; e += NTHR) {
                const int k = e / BM, m = e - k * BM;
                float v = 0.0f;
                if (k < K && m0 + m < M) v = (MODE == PW_FWD) ? a.w[(long)(m0 + m) * a.Cin + k] : a.w[(long)k * a.Cin + m0 + m];
                As[e] = v;
            }
        }
    }
    __syncthreads();

    const bool two_src = MODE == PW_DGRAD && a.src2 != nullptr;
    const int src_pitch = MODE == PW_FWD ? a.Pin : Q;
    const int dst_pitch = MODE == PW_FWD ? Q : a.Pin;
    const long src_n = (long)n * K * src_pitch;
    const long dst_n = (long)n * M * dst_pitch;
    const int row_bytes = src_pitch * 4;
    __amdgpu_buffer_rsrc_t r1 = cfn_rsrc(const_cast<float*>(a.src + src_n), K * row_bytes);
    __amdgpu_buffer_rsrc_t r2 = cfn_rsrc(const_cast<float*>((two_src ? a.src2 : a.src) + src_n), K * row_bytes);
    __amdgpu_buffer_rsrc_t rd = cfn_rsrc(a.dst + dst_n, M * dst_pitch * 4);
    __amdgpu_buffer_rsrc_t rx = cfn_rsrc(const_cast<float*>((a.ex ? a.ex : a.src) + (a.ex ? dst_n : 0)), a.ex ? M * dst_pitch * 4 : 0);
    __amdgpu_buffer_rsrc_t racc = cfn_rsrc(const_cast<float*>(MODE == PW_DGRAD && a.acc ? a.acc + (long)n * M * ((long)(a.Pin / (a.Hi * a.Wi)) * a.acc_Ho * a.acc_Wo) : a.src), MODE == PW_DGRAD && a.acc ? (unsigned)((long)M * (a.Pin / (a.Hi * a.Wi)) * a.acc_Ho * a.acc_Wo * 4) : 0u);
    float sacc[MT], qacc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) { sacc[i] = 0.0f; qacc[i] = 0.0f; }

    // wave tiles (32 positions) are dealt round robin over the nstrips workgroups of this (sample, row tile):
    // neighbouring waves take neighbouring tiles, every wave gets floor or ceil of the average
    for (int tile = 0;; ++tile) {
        const int qt = ((tile * a.nstrips + strip) * PWD_WAVES + wave) * 32;
        if (qt >= Q) break;                                   // wave uniform; no barriers inside the tile loop
        const int q = qt + col;
        const bool valid = q < Q;
        const int qc = valid ? q : Q - 1;
        const int voff = (half * src_pitch + qc) * 4;
        auto bload = [&](int k0, float (&d)[NU], float (&d2)[NU]) {   // rows k0 + 2j + half
#pragma unroll
            for (int j = 0; j < NU; ++j) {
                // a row base at or beyond K would push the scalar offset past the descriptor's range (which wraps instead of
                // failing the check): such rows are switched off through the vector offset and read 0
                const bool live = k0 + 2 * j < K;                          // wave uniform
                const int vo = live ? voff : 0x7fffffff, so = live ? (k0 + 2 * j) * row_bytes : 0;
                d[j] = pw_bload(r1, vo, so);
                if (MODE == PW_DGRAD) d2[j] = two_src ? pw_bload(r2, vo, so) : 0.0f;
            }
        };
        f16v acc[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

        // Software pipeline, all by hand (the compiler otherwise sinks every LDS read right in front of its MFMA and
        // parks the wave on lgkmcnt):
        //   B operand: two register sets ping-pong over units of 8 channels; a set is reloaded right after it is
        //              consumed and is not touched again for a whole unit (28+ MFMAs), no copies, no early waits;
        //   A operand: the MT weights of k-pair j+1 are read from LDS before the MT MFMAs of k-pair j issue.
        float a0[MT], a1[MT];
        auto lda = [&](float (&ar)[MT], int kl) {
            const float* p = As + kl * BM + col;
#pragma unroll
            for (int i = 0; i < MT; ++i) ar[i] = p[i * 32];
        };
        auto mm = [&](const float (&ar)[MT], float v) {
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[i], v, acc[i], 0, 0, 0);
        };
        auto pro = [&](float x, float x2, int kl) -> float {
            if (MODE == PW_FWD) {
                const float2 c = *reinterpret_cast<const float2*>(&sP[kl]);
                return cfn_act<ACT>(fmaf(x, c.x, c.y));
            }
            const float4 c = sP[kl];
            return fmaf(x2, c.y, fmaf(x, c.z, c.x));
        };
        auto unit = [&](float (&r)[NU], float (&r2)[NU], int u) {   // entering: a0 = weights of k-pair (u, 0)
            const int k = u + half;
            lda(a1, k + 2);
            __builtin_amdgcn_sched_barrier(0);
            const float v0 = pro(r[0], r2[0], k);
            mm(a0, v0);
            const float v1 = pro(r[1], r2[1], k + 2);
            __builtin_amdgcn_sched_barrier(0);
            lda(a0, k + 4);
            __builtin_amdgcn_sched_barrier(0);
            mm(a1, v1);
            const float v2 = pro(r[2], r2[2], k + 4);
            __builtin_amdgcn_sched_barrier(0);
            lda(a1, k + 6);
            __builtin_amdgcn_sched_barrier(0);
            mm(a0, v2);
            const float v3 = pro(r[3], r2[3], k + 6);
            __builtin_amdgcn_sched_barrier(0);
            lda(a0, k + 8);           // first k-pair of the next unit (one row past the image at the very end: unused)
            __builtin_amdgcn_sched_barrier(0);
            mm(a1, v3);
        };
        // B operand ring of NS register sets with static slots: the loads of unit u + NS are issued right after unit u was
        // consumed, so they have NS - 1 units of MFMA work (x 2 waves per SIMD) to land.  With the two-set ping-pong of
        // round 1 that was ONE unit = MT x 256 MFMA cycles, less than the loaded HBM latency for MT <= 3 (layers 3-4: 56 % of
        // the MFMA peak); unconditional loads (past the end reads 0: bounds check), units past the end are skipped.
        constexpr int NS = 4;
        float rs[NS][NU], rs2[NS][NU];
#pragma unroll
        for (int s = 0; s < NS; ++s) bload(s * PW_UNIT, rs[s], rs2[s]);
        lda(a0, half);
        for (int u = 0; u < Kpad; u += NS * PW_UNIT) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (u + s * PW_UNIT < Kpad) unit(rs[s], rs2[s], u + s * PW_UNIT);          // wave uniform, no vector memory inside
                bload(u + (s + NS) * PW_UNIT, rs[s], rs2[s]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- epilogue (same scheme as pw_gemm_kernel) -------------------------------------------
        const float vm = valid ? 1.0f : 0.0f;
        const int dvoff = valid ? (4 * half * dst_pitch + qc) * 4 : 0x7fffffff;
        int avoff = 0x7fffffff;                                // compact offset of this lane's position on the acc lattice
        long acc_pitch = 0;
        if (MODE == PW_DGRAD && a.acc) {
            const int hw = a.Hi * a.Wi;
            const int tq = qc / hw, rq = qc - tq * hw;
            const int hq = rq / a.Wi, wq = rq - hq * a.Wi;
            acc_pitch = (long)(a.Pin / hw) * a.acc_Ho * a.acc_Wo;
            if (valid && hq % a.acc_s == 0 && wq % a.acc_s == 0)
                avoff = (int)((4 * half * acc_pitch + ((long)tq * a.acc_Ho + hq / a.acc_s) * a.acc_Wo + wq / a.acc_s) * 4);
        }
        const bool any_acc = MODE == PW_DGRAD && a.acc && __any(avoff != 0x7fffffff);   // tiles inside odd rows skip the loads
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float t1[16], t2[16];
            float xe[16];
            if (MODE == PW_DGRAD && STATS) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    xe[r] = pw_bload(rx, dvoff, (m0 + i * 32 + (r & 3) + 8 * (r >> 2)) * dst_pitch * 4);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                float v = acc[i][r];
                if (any_acc)
                    v += pw_bload(racc, avoff, (int)((m0 + i * 32 + (r & 3) + 8 * (r >> 2)) * acc_pitch * 4));
                if (MODE == PW_FWD) {
                    t1[r] = v * vm;
                } else if (STATS) {
                    const float2 c = sE[ml];
                    const float dz = v * cfn_act_grad<ACT>(fmaf(xe[r], c.x, c.y)) * vm;
                    t1[r] = dz * xe[r];
                    t2[r] = dz;
                    v = dz * c.x;
                }
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rd, dvoff,
                                                      (m0 + i * 32 + (r & 3) + 8 * (r >> 2)) * dst_pitch * 4, 0);
            }
            if (STATS) {
#pragma unroll
                for (int pass = 0; pass < (MODE == PW_FWD ? 1 : 2); ++pass) {
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        red[((r & 3) + 8 * (r >> 2) + 4 * half) * PW_RED_PITCH + col] = pass == 0 ? t1[r] : t2[r];
                    asm volatile("" ::: "memory");
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
                    asm volatile("" ::: "memory");
                }
            }
        }
    }

    if (STATS) {
        __syncthreads();
        float* slot = redbase;                 // [PWD_WAVES][BM][2]  (8*BM*2 <= 8*32*33 floats for BM <= 224)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const float s = sacc[i] + __shfl_xor(sacc[i], 32, 64);
            const float qq = qacc[i] + __shfl_xor(qacc[i], 32, 64);
            if (half == 0) { slot[(wave * BM + i * 32 + col) * 2] = s; slot[(wave * BM + i * 32 + col) * 2 + 1] = qq; }
        }
        __syncthreads();
        for (int m = tid; m < BM; m += 64 * PWD_WAVES) {
            if (m0 + m < M) {
                float s = 0.0f, qq = 0.0f;
#pragma unroll
                for (int w = 0; w < PWD_WAVES; ++w) { s += slot[(w * BM + m) * 2]; qq += slot[(w * BM + m) * 2 + 1]; }
                cfn_add64(&a.s1[(long)n * M + m0 + m], (double)s);
                cfn_add64(&a.s2[(long)n * M + m0 + m], (double)qq);
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
template <int MODE, bool STATS, int ACT>
static int pwd_go(const PwArgs& a, int MT, int NW, unsigned blocks, size_t lds, hipStream_t st) {
#define CFN_PWD_GO(MTV)                                                                                        \
    do {                                                                                                       \
        {                                                                                                      \
            auto k = pw_deep_kernel<MTV, MODE, STATS, ACT, 8>;                                                 \
            if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * 8), lds, st, a);                                     \
        }                                                                                                      \
    } while (0)
    switch (MT) {
        case 2: CFN_PWD_GO(2); break;
        case 3: CFN_PWD_GO(3); break;
        case 4: CFN_PWD_GO(4); break;
        case 5: CFN_PWD_GO(5); break;
        case 6: CFN_PWD_GO(6); break;
        default: CFN_PWD_GO(7); break;
    }
#undef CFN_PWD_GO
    return cfn_check_launch("pwconv(deep)");
}

template <int MODE, bool STATS>
static int pwd_act(const PwArgs& a, int MT, int NW, unsigned blocks, size_t lds, hipStream_t st) {
    switch (a.act) {
        case CFN_ACT_RELU: return pwd_go<MODE, STATS, CFN_ACT_RELU>(a, MT, NW, blocks, lds, st);
        case CFN_ACT_SWISH: return pwd_go<MODE, STATS, CFN_ACT_SWISH>(a, MT, NW, blocks, lds, st);
        default: return pwd_go<MODE, STATS, CFN_ACT_NONE>(a, MT, NW, blocks, lds, st);
    }
}

int pwd_try_launch(PwArgs& a, int mode, bool stats, hipStream_t st) {
    if (a.stem || a.stride != 1 || a.K < 48 || a.M <= 32) return -1;
    { const char* e = getenv("CFN_PWD_OFF"); if (e && atoi(e)) return -1; }
    if (a.act != CFN_ACT_NONE && a.act != CFN_ACT_RELU && a.act != CFN_ACT_SWISH) return -1;
    const int Kpad = (a.K + PW_UNIT - 1) / PW_UNIT * PW_UNIT;
    int mt_max = (120 * 1024 / 4 / Kpad) / 32;            // resident weight image <= 120 KiB
    if (mt_max > 7) mt_max = 7;
    if (mt_max < 2) return -1;
    if ((long)a.K * a.Pin * 4 >= (1L << 31) || (long)a.K * a.Q * 4 >= (1L << 31) || (long)a.M * a.Pin * 4 >= (1L << 31) ||
        (long)a.M * a.Q * 4 >= (1L << 31))
        return -1;
    const int M32 = cfn_cdiv(a.M, 32);
    const int ntile = cfn_cdiv(M32, mt_max);
    const int MT = cfn_cdiv(M32, ntile);
    if (MT < 2) return -1;
    a.Kpad = Kpad;
    a.mtiles = cfn_cdiv(a.M, 32 * MT);
    const int BM = 32 * MT;
    // one workgroup (8 waves, one resident weight image) per CU: the ~256 workgroups are split evenly over the
    // (sample, row tile) groups; inside a group the 32-position wave tiles are dealt round robin
    const int NW = 8;
    const long wtiles = cfn_cdiv(a.Q, 32);
    const long groups = (long)a.N * a.mtiles;
    // two workgroups per CU where both fit (LDS <= 80 KiB, <= 128 VGPRs: 4 row tiles, or 2 with the act' epilogue): the
    // HBM-bound layer-2 shapes gain (48->108 @28 dgrad 0.50 -> 0.42 ms); the big-image layers stay at one per CU
    const size_t lds_probe = ((size_t)Kpad * BM + 4 * (Kpad + PW_UNIT) + 2 * BM + NW * 32 * PW_RED_PITCH) * sizeof(float);
    const bool two = lds_probe <= 80 * 1024 && MT <= ((mode == PW_DGRAD && stats) ? 2 : 4);
    long bpg = (two ? 512L : 256L) / groups;
    if (bpg < 1) bpg = 1;
    if (bpg > cfn_cdiv(wtiles, NW)) bpg = cfn_cdiv(wtiles, NW);
    a.nstrips = (int)bpg;
    a.tpb = 0;
    const unsigned blocks = (unsigned)(groups * bpg);
    const size_t lds = ((size_t)Kpad * BM + 4 * (Kpad + PW_UNIT) + 2 * BM + NW * 32 * PW_RED_PITCH) * sizeof(float);
    if (lds > 160 * 1024) return -1;
    if (mode == PW_FWD) return stats ? pwd_act<PW_FWD, true>(a, MT, NW, blocks, lds, st) : pwd_act<PW_FWD, false>(a, MT, NW, blocks, lds, st);
    if (!stats) { a.act = CFN_ACT_NONE; return pwd_go<PW_DGRAD, false, CFN_ACT_NONE>(a, MT, NW, blocks, lds, st); }
    return pwd_act<PW_DGRAD, true>(a, MT, NW, blocks, lds, st);
}
