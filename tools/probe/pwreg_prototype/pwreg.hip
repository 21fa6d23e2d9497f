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
__global__ __launch_bounds__(64 * PWR_WAVES, 4) void pwr_fwd_kernel(const PwArgs a) {
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
    float* scr = reinterpret_cast<float*>(sP + KP) + wave * (32 * 20);      // per wave [32 channels][16 positions + 4 pad]

    for (int k = tid; k < KP; k += 64 * PWR_WAVES)
        sP[k] = float2{(k < K && a.pa) ? (float)a.pa[(long)n * K + k] : 1.0f, (k < K && a.pb) ? (float)a.pb[(long)n * K + k] : 0.0f};

    // this wave's 32 weight rows, all k-blocks, three terms: the MFMA's B operand (column = channel j, k = kb*16 + kg*8 + i)
    const int mt = wave, row = mt * 32 + j;
    const bool has_rows = mt * 32 < M;                                      // wave uniform
    u4r Wr[NKB][3];
    const int abl = a.tpb;                                                  // ABLATION (temporary)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = kb * 16 + kg * 8 + i;
            v[i] = (row < M && k < K && !(abl & 8)) ? a.w[(long)row * a.Cin + k] : 0.0f;
        }
        pwr_split8(v, Wr[kb]);
    }
    __syncthreads();

    __amdgpu_buffer_rsrc_t rs = cfn_rsrc(const_cast<float*>(a.src + (long)n * K * Q), (unsigned)((long)K * Q * 4));
    const int mrows = max(min(32, M - mt * 32), 0);
    __amdgpu_buffer_rsrc_t rd = cfn_rsrc(a.dst + (long)n * M * Q + (long)mt * 32 * Q, (unsigned)((long)mrows * Q * 4));
    const int ntiles = (Q + 31) / 32, tstep = a.nstrips;
    const bool stager = wave < NKB;                                         // wave w stages k-block w
    const int lane_ld = kg * 8 * Q * 4 + j * 4;
    const int st_off = j * PITCH + (wave * 16 + kg * 8) * 2;                // where this lane's 8 k's of position j live in an image
    const int rd_off = j * PITCH + kg * 16;                                 // + kb * 32: A operand (row = position j, k = kb*16 + kg*8 + i)
    const int mrow = lane >> 2, mcol = 4 * (lane & 3);                      // memory-side role in the epilogue
    const int lane_mem = mrow * Q * 4 + mcol * 4;
    float ssum = 0.0f, qsum = 0.0f;

    float ld[8];
    auto issue = [&](int tile) {                                            // unconditional loads: a dead tile / row reads zeros
        const bool live = stager && tile < ntiles;
        const int vo = (live && tile * 32 + j < Q && !(abl & 4)) ? lane_ld : PWR_OOB;
        const int base = live ? tile * 32 * 4 : 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = wave * 16 + i;                                    // + 8 kg through the lane offset
            const int so = (live && r < K) ? r * Q * 4 + base : base;       // (a row base beyond the range must not enter the scalar offset)
            ld[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (live && r + 8 * kg < K) ? vo : PWR_OOB, so, 0));
        }
    };
    auto wsync = [&]() {                                                   // LDS ops of a wave run in order; only the compiler is told
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    int tile = wg;
    if (abl & 16) {                                                         // EXPERIMENT: the second workgroup of a CU starts half an iteration late
        if ((blockIdx.x >> 8) & 1) { __builtin_amdgcn_s_sleep(55); }
    }
    if (abl & 32) {
        if ((blockIdx.x >> 8) & 1) { __builtin_amdgcn_s_sleep(110); }
    }
    issue(tile);
    for (int it = 0; tile < ntiles; ++it, tile += tstep) {
        unsigned char* buf = Bs + (it & 1) * 3 * IMG;
        if (stager) {                                                       // activate, split once, publish
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float2 c = sP[wave * 16 + kg * 8 + i];
                v[i] = cfn_act<ACT>(fmaf(ld[i], c.x, c.y));
            }
            u4r t[3];
            pwr_split8(v, t);
#pragma unroll
            for (int s = 0; s < 3; ++s) *reinterpret_cast<u4r*>(buf + s * IMG + st_off) = t[s];
        }
        issue(tile + tstep);                                               // the next tile's rows travel during this tile's MFMAs
        __syncthreads();
        if (!has_rows) continue;
        f16v acc = (f16v)0.0f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            bf16x8r A[3];
#pragma unroll
            for (int s = 0; s < 3; ++s) A[s] = *reinterpret_cast<const bf16x8r*>(buf + s * IMG + rd_off + kb * 32);
#define PWR_MM(SA, SW) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[SA], __builtin_bit_cast(bf16x8r, Wr[kb][SW]), acc, 0, 0, 0)
            if (!(abl & 2)) { PWR_MM(0, 2); PWR_MM(2, 0); PWR_MM(1, 1); PWR_MM(0, 1); PWR_MM(1, 0); PWR_MM(0, 0); }
#undef PWR_MM
        }
        // ---- epilogue: lane (j, kg) holds channel mt*32 + j, positions 8 g + 4 kg + e in registers 4 g + e (see pwsplit.hip)
        const int q0 = tile * 32;
        const bool full = q0 + 32 <= Q && mrows == 32;                      // wave uniform
        const bool chv = j < mrows;
        const int so = q0 * 4;
        float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const float gmask = (full || (chv && q0 + 16 * u + 8 * sl + 4 * kg < Q)) ? 1.0f : 0.0f;
                f4v o;
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    const float e = acc[4 * (2 * u + sl) + e4];
                    if (STATS) {
                        const float em = e * gmask;
                        t1 += em;
                        t2 = fmaf(em, em, t2);
                    }
                    o[e4] = e;
                }
                *reinterpret_cast<f4v*>(scr + j * 20 + 8 * sl + 4 * kg) = o;
            }
            wsync();
#pragma unroll
            for (int sx = 0; sx < 2; ++sx) {
                const f4v v = *reinterpret_cast<const f4v*>(scr + (mrow + 16 * sx) * 20 + mcol);
                const bool ok = (full || (16 * sx + mrow < mrows && q0 + 16 * u + mcol < Q)) && !(abl & 1);
                cfn_bst128(__builtin_bit_cast(u4r, v), rd, (ok ? lane_mem + sx * 16 * Q * 4 : PWR_OOB) + u * 64, so);
            }
            wsync();
        }
        ssum += t1; qsum += t2;
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
    const size_t lds = (size_t)2 * 3 * 32 * (KP * 2 + 16) + (size_t)KP * 8 + (size_t)PWR_WAVES * 32 * 20 * 4;
    PwArgs b = a;
    const int ntiles = cfn_cdiv(a.Q, 32);
    static const int wg_env = getenv("CFN_PWR_WGS") ? atoi(getenv("CFN_PWR_WGS")) : 0;
    long wgs = cfn_cdiv(wg_env > 0 ? wg_env : 512, (long)a.N);              // two workgroups per CU
    if (wgs > ntiles) wgs = ntiles;
    if (wgs < 1) wgs = 1;
    b.nstrips = (int)wgs;
    b.tpb = getenv("CFN_PWR_ABL") ? atoi(getenv("CFN_PWR_ABL")) : 0;
    const unsigned blocks = (unsigned)((long)a.N * wgs);
    switch (nkb) {
        case 3: return stats ? pwr_go<3, true>(b, blocks, lds, st) : pwr_go<3, false>(b, blocks, lds, st);
        case 4: return stats ? pwr_go<4, true>(b, blocks, lds, st) : pwr_go<4, false>(b, blocks, lds, st);
        case 5: return stats ? pwr_go<5, true>(b, blocks, lds, st) : pwr_go<5, false>(b, blocks, lds, st);
        case 6: return stats ? pwr_go<6, true>(b, blocks, lds, st) : pwr_go<6, false>(b, blocks, lds, st);
        default: return -1;
    }
}
