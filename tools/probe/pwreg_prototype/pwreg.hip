// Pointwise (1x1x1) forward contraction on fp32 tensors with split-bf16 arithmetic (pwsplit.hip's 6-term product), REGISTER-RESIDENT
// WEIGHTS: the many-row / shallow-contraction shapes (x3d_fine.py:100-105 conv1 of layer 3: 96 -> 216 channels @14x14).
//
// Why: pws_kernel gives a wave ALL output rows of its 32 positions (7 x 16 accumulators, 256 VGPRs, two waves per SIMD) and its
// phases -- load, activate + split, 252 MFMAs, epilogue -- run one after the other (the parts add up;
// neither the matrix pipe, nor the vector ALU, nor HBM is busy).  Here the roles are turned round:
//   * a wave owns ONE 32-row tile of the weight matrix for the whole contraction, split once into 3 bf16 terms and kept in
//     registers (K <= 96: 6 k-blocks x 3 terms x 4 registers = 72 VGPRs) -- no weight image in LDS, no LDS read per MFMA operand;
//   * the workgroup's 8 waves share the ACTIVATIONS of a 32-position tile: wave w loads k-block w (8 coalesced 128-byte row
//     segments per lane group), applies the prologue, splits ONCE and writes the three bf16 B-operand images to LDS (double
//     buffered, one barrier per tile); every wave then reads them back as 16-byte operands for its own row tile: 18 LDS reads and 36
//     MFMAs per tile and wave, 16 accumulators;
//   * <= 128 VGPRs: two workgroups per CU (4 waves per SIMD): one stages / waits for its loads while the other multiplies;
//   * transposed result (activations as the A operand): in-lane statistics, 16-byte stores through the wave's scratch (pwsplit.hip).
// Shapes: forward, stride 1, K <= 96, 128 < M <= 256 (5-8 row tiles = waves), Q % 4 == 0; everything else stays with pws / pw_deep.
#include "pw_common.h"
#include <stdlib.h>
#include <type_traits>

typedef __bf16 bf16x8r __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2r __attribute__((ext_vector_type(2)));
typedef float f2r __attribute__((ext_vector_type(2)));
typedef unsigned u4r __attribute__((ext_vector_type(4)));

#define PWR_WAVES 8
#define PWR_OOB 0x40000000

__device__ __forceinline__ float pwr_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float pwr_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ unsigned pwr_pack(float lo, float hi) {
    const bf16x2r b = __builtin_convertvector((f2r){lo, hi}, bf16x2r);      // v_cvt_pk_bf16_f32 (round to nearest even)
    return __builtin_bit_cast(unsigned, b);
}
// 8 fp32 values -> three 16-byte operands (terms 1..3 of each value, 8 consecutive k)
__device__ __forceinline__ void pwr_split8(const float (&v)[8], u4r (&t)[3]) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        float a = v[2 * h], b = v[2 * h + 1];
        const unsigned p0 = pwr_pack(a, b);
        a -= pwr_lo(p0); b -= pwr_hi(p0);
        const unsigned p1 = pwr_pack(a, b);
        a -= pwr_lo(p1); b -= pwr_hi(p1);
        t[0][h] = p0; t[1][h] = p1; t[2][h] = pwr_pack(a, b);
    }
}

template <int NKB, int ACT, bool STATS>
__global__ __launch_bounds__(64 * PWR_WAVES) void pwr_fwd_kernel(const PwArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KP = 16 * NKB;
    constexpr int PITCH = KP * 2 + 16;                                      // bytes per position row of one image: an odd number of 16-byte slots
    static_assert(((PITCH / 16) & 1) == 1, "conflict-free pitch");
    constexpr int IMG = 32 * PITCH;                                         // one term, 32 positions
    const int tid = threadIdx.x, wave = cfn_uni(tid >> 6), lane = tid & 63, kg = lane >> 5, j = lane & 31;
    const int K = a.K, M = a.M, Q = a.Q;
    const unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    const int wg = L % a.nstrips, n = L / a.nstrips;

    unsigned char* Bs = smem;                                               // [2 buffers][3 terms][32 positions][PITCH]
    float2* sP = reinterpret_cast<float2*>(Bs + 2 * 3 * IMG);              // [KP] prologue coefficients
    float* scr = reinterpret_cast<float*>(sP + KP) + wave * (2 * 32 * 36);  // per wave [2 tiles][32 channels][32 positions + 4 pad]

    // this wave's 32 weight rows, all k-blocks, three terms: the MFMA's B operand (column = channel j, k = kb*16 + kg*8 + i)
    const int mt = wave, row = mt * 32 + j;
    const bool has_rows = mt * 32 < M;                                      // wave uniform
    u4r Wr[NKB][3];
    {
        // the wave's 32 x K block of w through LDS: coalesced 16-byte loads (consecutive lanes along k), then each lane reads its row
        // (a row-per-lane gather straight from memory is 64 separate requests per load instruction: ~16 us of a 0.15 ms launch).
        // The whole dynamic LDS is free at this point: 8 waves x 32 rows x (KP + 4) floats <= 2 x 3 x IMG + scratch.
        float* wtmp = reinterpret_cast<float*>(smem) + wave * (32 * (KP + 4));
        const bool vec = (a.Cin & 3) == 0 && (reinterpret_cast<uintptr_t>(a.w) & 15) == 0;
        for (int e = lane; e < 32 * (KP / 4); e += 64) {
            const int rr = e / (KP / 4), k4 = (e - rr * (KP / 4)) * 4;
            f4v v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (mt * 32 + rr < M) {
                const float* src = a.w + (long)(mt * 32 + rr) * a.Cin + k4;
                if (vec && k4 + 3 < K) v = *reinterpret_cast<const f4v*>(src);
                else { if (k4 < K) v.x = src[0]; if (k4 + 1 < K) v.y = src[1]; if (k4 + 2 < K) v.z = src[2]; if (k4 + 3 < K) v.w = src[3]; }
            }
            *reinterpret_cast<f4v*>(wtmp + rr * (KP + 4) + k4) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            float v[8];
            const float* p = wtmp + j * (KP + 4) + kb * 16 + kg * 8;
            const f4v lo = *reinterpret_cast<const f4v*>(p), hi = *reinterpret_cast<const f4v*>(p + 4);
            v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
            pwr_split8(v, Wr[kb]);
        }
    }
    __syncthreads();                                                        // the staging area is re-used: coefficients, activation buffers
    for (int k = tid; k < KP; k += 64 * PWR_WAVES)
        sP[k] = float2{(k < K && a.pa) ? (float)a.pa[(long)n * K + k] : 1.0f, (k < K && a.pb) ? (float)a.pb[(long)n * K + k] : 0.0f};

    __syncthreads();

    __amdgpu_buffer_rsrc_t rs = cfn_rsrc(const_cast<float*>(a.src + (long)n * K * Q), (unsigned)((long)K * Q * 4));
    const int mrows = max(min(32, M - mt * 32), 0);
    __amdgpu_buffer_rsrc_t rd = cfn_rsrc(a.dst + (long)n * M * Q + (long)mt * 32 * Q, (unsigned)((long)mrows * Q * 4));
    const int ntiles = (Q + 31) / 32, tstep = a.nstrips;
    const bool stager = wave < NKB;                                         // wave w stages k-block w
    const int lane_ld = kg * 8 * Q * 4 + j * 4;
    const int st_off = j * PITCH + (wave * 16 + kg * 8) * 2;                // where this lane's 8 k's of position j live in an image
    const int rd_off = j * PITCH + kg * 16;                                 // + kb * 32: A operand (row = position j, k = kb*16 + kg*8 + i)
    const int mrow = lane >> 3, mcol = 4 * (lane & 7);                      // memory-side role: row mrow + 8 s, 16 bytes at position mcol (8 lanes = one 128-byte line)
    const int lane_mem = mrow * Q * 4 + mcol * 4;
    float ssum = 0.0f, qsum = 0.0f;

    auto issue = [&](int tile, float (&ld)[8]) {                                            // unconditional loads: a dead tile / row reads zeros
        const bool live = stager && tile < ntiles;
        const int vo = (live && tile * 32 + j < Q) ? lane_ld : PWR_OOB;
        const int base = live ? tile * 32 * 4 : 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = wave * 16 + i;                                    // + 8 kg through the lane offset
            const int so = cfn_uni((live && r < K) ? r * Q * 4 + base : base);   // wave uniform (stated: otherwise every load is a waterfall loop); a row base beyond the range must not enter it
            ld[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (live && r + 8 * kg < K) ? vo : PWR_OOB, so, 0));
        }
    };
    auto wsync = [&]() {                                                   // LDS ops of a wave run in order; only the compiler is told
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    // Two tiles per loop trip: the LDS buffers, the scratch halves and the two load sets (a set is re-issued for the tile two ahead
    // right after it was consumed) are static in each half -- no register moves, no dynamic LDS offsets.
    int pq0 = -1, last_buf = 0;                                             // previous tile's first position (-1: none), its scratch buffer
    auto drain_read = [&](int pbuf, int sx) -> f4v { return *reinterpret_cast<const f4v*>(scr + pbuf * (32 * 36) + (mrow + 8 * sx) * 36 + mcol); };
    auto drain_store = [&](f4v v, int sx) {
        const bool ok = 8 * sx + mrow < mrows && pq0 + mcol < Q;            // pq0 < 0 (no previous tile): switched off through the scalar branch below
        cfn_bst128(__builtin_bit_cast(u4r, v), rd, (ok && pq0 >= 0) ? lane_mem + sx * 8 * Q * 4 : PWR_OOB, cfn_uni(pq0 >= 0 ? pq0 * 4 : 0));
    };
    // statistics in-lane + transposed tile into scratch half PARV (it leaves during the next tile's MFMAs)
    f16v acc;
    auto finish = [&](int PARV, int q0) {
        const bool full = q0 + 32 <= Q && mrows == 32;                      // wave uniform
        float* sb = scr + PARV * (32 * 36);
        f4v s1v = {0.0f, 0.0f, 0.0f, 0.0f}, s2v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (full) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f4v o = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                if (STATS) { s1v += o; s2v = __builtin_elementwise_fma(o, o, s2v); }
                *reinterpret_cast<f4v*>(sb + j * 36 + 8 * g + 4 * kg) = o;
            }
        } else {
            const bool chv = j < mrows;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float gmask = (chv && q0 + 8 * g + 4 * kg < Q) ? 1.0f : 0.0f;
                const f4v o = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                const f4v om = o * gmask;
                if (STATS) { s1v += om; s2v = __builtin_elementwise_fma(om, om, s2v); }
                *reinterpret_cast<f4v*>(sb + j * 36 + 8 * g + 4 * kg) = o;
            }
        }
        wsync();
        if (STATS) { ssum += (s1v.x + s1v.y) + (s1v.z + s1v.w); qsum += (s2v.x + s2v.y) + (s2v.z + s2v.w); }
        pq0 = q0; last_buf = PARV;
    };
    const bool late = wave >= 4;                                            // wave uniform
    bool have = false; int hq0 = 0;                                         // (late waves) a tile waits to be finished
    auto step = [&](auto par_tag, float (&lc)[8], int tile) {
        constexpr int PAR = decltype(par_tag)::value;
        unsigned char* buf = Bs + PAR * 3 * IMG;
        if (stager) {                                                       // activate, split once, publish
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float2 c = sP[wave * 16 + kg * 8 + i];
                v[i] = cfn_act<ACT>(fmaf(lc[i], c.x, c.y));
            }
            u4r t[3];
            pwr_split8(v, t);
#pragma unroll
            for (int s = 0; s < 3; ++s) *reinterpret_cast<u4r*>(buf + s * IMG + st_off) = t[s];
        }
        issue(tile + 2 * tstep, lc);                                        // two tiles ahead, into the set just consumed
        __syncthreads();
        if (!has_rows) return;
        constexpr int pbuf = PAR ^ 1;
        if (late && have) finish(pbuf, hq0);
        f4v dv[4];
        bf16x8r AA[2][3];                                                   // the operands of k-block kb + 1 are read before the MFMAs of kb issue
#pragma unroll
        for (int s = 0; s < 3; ++s) AA[0][s] = *reinterpret_cast<const bf16x8r*>(buf + s * IMG + rd_off);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            bf16x8r (&A)[3] = AA[kb & 1];
            if (kb + 1 < NKB) {
#pragma unroll
                for (int s = 0; s < 3; ++s) AA[(kb + 1) & 1][s] = *reinterpret_cast<const bf16x8r*>(buf + s * IMG + rd_off + (kb + 1) * 32);
            }
            __builtin_amdgcn_sched_barrier(0);
#define PWR_MM(SA, SW) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[SA], __builtin_bit_cast(bf16x8r, Wr[kb][SW]), acc, 0, 0, 0)
            if (kb == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], __builtin_bit_cast(bf16x8r, Wr[kb][2]), (f16v)0.0f, 0, 0, 0);
            else PWR_MM(0, 2);
            PWR_MM(2, 0); PWR_MM(1, 1);
            // drain: one 16-byte read / store of the previous tile per half group of MFMAs (NKB >= 3: 2 NKB >= 6 slots for 4 reads + 4 stores)
            if (2 * kb < 4) dv[2 * kb] = drain_read(pbuf, 2 * kb);
            if (2 * kb >= 1 && 2 * kb - 1 < 4) drain_store(dv[2 * kb - 1], 2 * kb - 1);
            __builtin_amdgcn_sched_barrier(0);
            PWR_MM(0, 1); PWR_MM(1, 0); PWR_MM(0, 0);
#undef PWR_MM
            if (2 * kb + 1 < 4) dv[2 * kb + 1] = drain_read(pbuf, 2 * kb + 1);
            if (2 * kb < 4) drain_store(dv[2 * kb], 2 * kb);
            __builtin_amdgcn_sched_barrier(0);
        }
        // The two waves of a SIMD (w and w + 4) would otherwise multiply at the same time and do their vector work at the same time:
        // waves 0-3 finish their tile right behind its MFMAs, waves 4-7 carry it over the next barrier and finish it in front of the
        // next tile's MFMAs -- while one wave of a SIMD feeds the matrix pipe the other does statistics / LDS / stores.
        if (!late) finish(PAR, cfn_uni(tile * 32));
        else { have = true; hq0 = cfn_uni(tile * 32); }
    };
    int tile = cfn_uni(wg), last_par = 0;
    float ldA[8], ldB[8];
    issue(tile, ldA);
    issue(tile + tstep, ldB);
    for (;;) {
        if (tile >= ntiles) break;
        step(std::integral_constant<int, 0>{}, ldA, tile);
        last_par = 0;
        tile += tstep;
        if (tile >= ntiles) break;
        step(std::integral_constant<int, 1>{}, ldB, tile);
        last_par = 1;
        tile += tstep;
    }
    if (has_rows && late && have) finish(last_par, hq0);
    if (has_rows && pq0 >= 0) {                                             // the last tile leaves now
#pragma unroll
        for (int sx = 0; sx < 4; ++sx) drain_store(drain_read(last_buf, sx), sx);
    }
    if (STATS && a.s1 && has_rows) {
        ssum += __shfl_xor(ssum, 32, 64);
        qsum += __shfl_xor(qsum, 32, 64);
        if (kg == 0 && row < M) {
            atomicAdd(&a.s1[(long)n * M + row], (double)ssum);
            atomicAdd(&a.s2[(long)n * M + row], (double)qsum);
        }
    }
}

template <int NKB, bool STATS>
static int pwr_go(const PwArgs& a, unsigned blocks, size_t lds, hipStream_t st) {
#define PWR_GO(ACTV)                                                                                                        \
    do {                                                                                                                    \
        auto k = pwr_fwd_kernel<NKB, ACTV, STATS>;                                                                          \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * PWR_WAVES), lds, st, a);                                              \
    } while (0)
    switch (a.act) {
        case CFN_ACT_RELU: PWR_GO(CFN_ACT_RELU); break;
        case CFN_ACT_SWISH: PWR_GO(CFN_ACT_SWISH); break;
        default: PWR_GO(CFN_ACT_NONE); break;
    }
#undef PWR_GO
    return cfn_check_launch("pwconv(split bf16, register-resident weights)");
}

// returns -1 when the shape is not handled (the caller goes on to pws_try_launch / pwd_try_launch)
int pwr_try_launch(PwArgs& a, int mode, bool stats, hipStream_t st) {
    static const int on = getenv("CFN_PWR") ? atoi(getenv("CFN_PWR")) : 1;
    if (!on || pws_terms_now() != 6 || mode != PW_FWD || a.stem || a.stride != 1 || a.acc) return -1;
    if (a.K < 48 || a.K > 96 || a.M <= 128 || a.M > 32 * PWR_WAVES || (a.Q & 3)) return -1;
    if (a.act != CFN_ACT_NONE && a.act != CFN_ACT_RELU && a.act != CFN_ACT_SWISH) return -1;
    if ((long)a.K * a.Q * 4 >= 0x3ffffff0L || (long)a.M * a.Q * 4 >= 0x3ffffff0L) return -1;
    if (((uintptr_t)a.src | (uintptr_t)a.dst) & 15) return -1;
    const int nkb = cfn_cdiv(a.K, 16);
    const int KP = 16 * nkb;
    const size_t lds = (size_t)2 * 3 * 32 * (KP * 2 + 16) + (size_t)KP * 8 + (size_t)PWR_WAVES * 2 * 32 * 36 * 4;
    PwArgs b = a;
    const int ntiles = cfn_cdiv(a.Q, 32);
    static const int wg_env = getenv("CFN_PWR_WGS") ? atoi(getenv("CFN_PWR_WGS")) : 0;
    long wgs = cfn_cdiv(wg_env > 0 ? wg_env : 256, (long)a.N);              // one workgroup per CU
    if (wgs > ntiles) wgs = ntiles;
    if (wgs < 1) wgs = 1;
    b.nstrips = (int)wgs;
    const unsigned blocks = (unsigned)((long)a.N * wgs);
    switch (nkb) {
        case 3: return stats ? pwr_go<3, true>(b, blocks, lds, st) : pwr_go<3, false>(b, blocks, lds, st);
        case 4: return stats ? pwr_go<4, true>(b, blocks, lds, st) : pwr_go<4, false>(b, blocks, lds, st);
        case 5: return stats ? pwr_go<5, true>(b, blocks, lds, st) : pwr_go<5, false>(b, blocks, lds, st);
        case 6: return stats ? pwr_go<6, true>(b, blocks, lds, st) : pwr_go<6, false>(b, blocks, lds, st);
        default: return -1;
    }
}
