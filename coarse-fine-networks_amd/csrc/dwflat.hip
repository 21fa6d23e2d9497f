// Depthwise 3x3x3 forward onto 56x56 / 28x28 / 14x14 output planes, stride 1 and 2 (conv2 of layers 1-3 of X3D-M; x3d_fine.py:89-97,171-201),
// fp32 tensors --
// FLAT kernels (round 4).
//
// dwcp.hip marches a wave along t: per frame step it asks for ONE frame of its band, two frames ahead.  The temporal 5-tap conv (dwt5.hip, 4g)
// went from 5.1 to 5.8-6.1 TB/s when it was rebuilt with every load of a work item issued UP FRONT and the items dealt in memory order.  Same
// scheme here, with the spatial taps through LDS:
//   * a work item is one (sample, channel, chunk of TO output frames[, band of RB rows]); a lane loads ONE float4 position of the item's input
//     rows for ALL TO + 2 frames (10 x 16 bytes per lane in flight), applies the prologue and stores the values into an LDS image
//     [frame][row][W] (no halo columns: the two edge taps read a valid address and are multiplied by 0; rows / frames outside the tensor are
//     stored as zeros);
//   * after one barrier a lane owns a few adjacent outputs of one position for all TO frames: per input frame and image row a few LDS reads
//     and packed FMAs into the three output frames the input frame feeds (weights are SGPR operands);
//   * per-thread statistics, one fp64 atomic pair per item;
//   * items in memory order (band fastest, then t-chunk, then channel), each XCD one contiguous eighth (cfn_xcd_remap): the 2 halo frames /
//     rows an item re-reads were fetched by its neighbour on the same XCD a moment ago.
// 56x56 / 28x28: a WORKGROUP per item (14-row band / whole plane), a lane = 4 columns of one row.  14x14: a WAVE per item (the 49 float4 of a
// plane = 49 lanes; a lane computes a 2 x 2 output block), 4 independent waves per workgroup, no workgroup barrier.
//
// These kernels are POWER bound (tools/clk_watch.sh: 1400 W, sclk 1.85-2.0 GHz instead of 2.4; conv1_t's flat kernel: 1260 W at 2.4 GHz,
// 6.1 TB/s): steady state on one box, 8 clips x T = 256 (tools/busy.py): 56x56 dwcp 586 us, this kernel 515-523 (5.3-5.4 TB/s); 28x28
// 293 -> 257-261.  What was measured on the way: image rows W + 8 apart with zero columns (3 instead of 4 workgroups per CU)
// 530 / 274; TO = 6 / 4: 527 / 548 and 263 / 273; a persistent producer / consumer version (4 loader waves that never store + 4 worker waves
// that never load, LDS double buffer, one barrier per item) 610 / 295; with 8 of the 9 taps per frame switched off 466 (6.0 TB/s).
#include "cfn_common.h"
#include <stdint.h>
#include <stdlib.h>

struct DwFlatArgs {
    const float* x; const double* A; const double* B; const float* w; float* y; double* s1; double* s2;
    int N, C, T, act, nchunks;
    long total;          // 14x14: number of wave items
};

typedef float __attribute__((ext_vector_type(4))) fl_f4;
typedef float __attribute__((ext_vector_type(2))) fl_p2;
typedef unsigned __attribute__((ext_vector_type(4))) fl_u4;
typedef unsigned __attribute__((ext_vector_type(2))) fl_u2;

template <int W, int RB, int TO>
__global__ __launch_bounds__(256, 4) void dw3d_flat_fwd_kernel(const DwFlatArgs a) {
    constexpr int W4 = W / 4, NB = W / RB, IR = RB + 2, PIT = W, NF = TO + 2, FR = IR * PIT, P = W * W, OOB = 0x7fff0000;
    constexpr int NLOAD = IR * W4, NCOMP = RB * W4;
    static_assert(NLOAD <= 256 && W % RB == 0 && W % 4 == 0, "geometry");
    __shared__ __attribute__((aligned(16))) float img[NF * FR];
    __shared__ float red[8];

    const int tid = threadIdx.x;
    const unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    const int band = cfn_uni((int)(L % NB));
    const unsigned rest = cfn_uni(L / NB);
    const int chunk = cfn_uni((int)(rest % (unsigned)a.nchunks));
    const long nc = cfn_uni((int)(rest / (unsigned)a.nchunks));
    const int c = cfn_uni((int)(nc % a.C));
    const int T = a.T, t0 = chunk * TO;

    const int lr = tid / W4, lc = tid - lr * W4;
    const int grow = band * RB - 1 + lr;
    const bool lvalid = tid < NLOAD && grow >= 0 && grow < W;
    const int lofs = lvalid ? (grow * W + lc * 4) * 4 : OOB;
    __amdgpu_buffer_rsrc_t rx = cfn_rsrc(a.x + nc * (long)T * P, (unsigned)((long)T * P * 4));
    __amdgpu_buffer_rsrc_t ry = cfn_rsrc(a.y + nc * (long)T * P, (unsigned)((long)T * P * 4));

    fl_f4 R[NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) {
        const int t = t0 - 1 + k;
        const bool tv = t >= 0 && t < T;                                 // workgroup uniform
        R[k] = __builtin_bit_cast(fl_f4, __builtin_amdgcn_raw_buffer_load_b128(rx, tv ? lofs : OOB, cfn_uni(tv ? t * P * 4 : 0), 0));
    }
    float wr[27];
#pragma unroll
    for (int j = 0; j < 27; ++j) wr[j] = cfn_uni(a.w[(long)c * 27 + j]);
    const float pa = cfn_uni(a.A ? (float)a.A[nc] : 1.0f);
    const float pb = cfn_uni(a.A ? (float)a.B[nc] : 0.0f);
    const float act_lo = a.act == CFN_ACT_RELU ? 0.0f : -__builtin_inff();      // none / ReLU only (the planner checks)

    if (tid < NLOAD) {
        float* dst = img + lr * PIT + lc * 4;
#pragma unroll
        for (int k = 0; k < NF; ++k) {
            const int t = t0 - 1 + k;
            const float m = (lvalid && t >= 0 && t < T) ? 1.0f : 0.0f;   // zero padding is applied AFTER the prologue
            fl_f4 v = R[k];
            v.x = fmaxf(fmaf(v.x, pa, pb), act_lo) * m; v.y = fmaxf(fmaf(v.y, pa, pb), act_lo) * m;
            v.z = fmaxf(fmaf(v.z, pa, pb), act_lo) * m; v.w = fmaxf(fmaf(v.w, pa, pb), act_lo) * m;
            *reinterpret_cast<fl_f4*>(dst + k * FR) = v;
        }
    }
    __syncthreads();

    float st1 = 0.0f, st2 = 0.0f;
    if (tid < NCOMP) {
        fl_p2 acc[TO][2];
#pragma unroll
        for (int j = 0; j < TO; ++j) acc[j][0] = acc[j][1] = (fl_p2){0.0f, 0.0f};
        const float* base = img + lr * PIT + lc * 4 - 1;                 // output row lr of the band = image rows lr .. lr + 2
        // the columns left of 0 / right of W - 1 are not stored: read a valid address, multiply by 0
        const int e0 = lc == 0 ? 1 : 0, e5 = lc == W4 - 1 ? 4 : 5;
        const float m0 = lc == 0 ? 0.0f : 1.0f, m5 = lc == W4 - 1 ? 0.0f : 1.0f;
#pragma unroll
        for (int k = 0; k < NF; ++k) {
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const float* q = base + k * FR + kh * PIT;
                const float q0 = q[e0] * m0, q5 = q[e5] * m5;
                const fl_f4 m = *reinterpret_cast<const fl_f4*>(q + 1);
                const fl_p2 vA = {q0, m.x}, vB = {m.x, m.y}, vC = {m.y, m.z}, vD = {m.z, m.w}, vE = {m.w, q5};
#pragma unroll
                for (int kt = 0; kt < 3; ++kt) {
                    const int j = k - kt;                                // output frame t0 + j reads input frames j .. j + 2 (k = j + kt)
                    if (j >= 0 && j < TO) {
                        const float w0 = wr[kt * 9 + kh * 3 + 0], w1 = wr[kt * 9 + kh * 3 + 1], w2 = wr[kt * 9 + kh * 3 + 2];
                        acc[j][0] = __builtin_elementwise_fma((fl_p2){w0, w0}, vA, __builtin_elementwise_fma((fl_p2){w1, w1}, vB,
                                    __builtin_elementwise_fma((fl_p2){w2, w2}, vC, acc[j][0])));
                        acc[j][1] = __builtin_elementwise_fma((fl_p2){w0, w0}, vC, __builtin_elementwise_fma((fl_p2){w1, w1}, vD,
                                    __builtin_elementwise_fma((fl_p2){w2, w2}, vE, acc[j][1])));
                    }
                }
            }
            if (k >= 2) {                                                // output frame k - 2 is complete
                const int j = k - 2, t = t0 + j;
                const bool emit = t < T;
                const fl_f4 y = {acc[j][0].x, acc[j][0].y, acc[j][1].x, acc[j][1].y};
                cfn_bst128(__builtin_bit_cast(fl_u4, y), ry, emit ? ((band * RB + lr) * W + lc * 4) * 4 : OOB, cfn_uni(emit ? t * P * 4 : 0));
                const fl_f4 ym = y * (emit ? 1.0f : 0.0f);
                st1 += ym.x + ym.y + ym.z + ym.w;
                st2 += ym.x * y.x + ym.y * y.y + ym.z * y.z + ym.w * y.w;
            }
        }
    }
    if (a.s1) {
        const int wave = tid >> 6, lane = tid & 63;
        st1 = cfn_wave_sum(st1); st2 = cfn_wave_sum(st2);
        if (lane == 0) { red[wave] = st1; red[4 + wave] = st2; }
        __syncthreads();
        if (tid == 0) {
            cfn_add64(&a.s1[nc], (double)(red[0] + red[1] + red[2] + red[3]));
            cfn_add64(&a.s2[nc], (double)(red[4] + red[5] + red[6] + red[7]));
        }
    }
}

// 14x14: one WAVE per (sample, channel, chunk of TO frames).  The plane is 49 float4: lane e < 49 loads float4 e of every frame (a float4
// may straddle two rows: staged as two 8-byte halves) and computes the 2 x 2 output block (rows 2 (e / 7), columns 2 (e % 7)).  The image has a
// zero row above and below the plane; its rows are 14 floats apart.
template <int TO>
__global__ __launch_bounds__(256, 4) void dw3d_flat14_fwd_kernel(const DwFlatArgs a) {
    unsigned long long* const det_keys = cfn_det_keys();
    constexpr int W = 14, P = 196, U = 49, PIT = 14, IR = 16, NF = TO + 2, FR = IR * PIT, OOB = 0x7fff0000;
    __shared__ __attribute__((aligned(16))) float smem[4 * NF * FR];
    const int lane = threadIdx.x & 63, wv = cfn_uni((int)(threadIdx.x >> 6));
    const unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    const long widx = cfn_uni((long)L * 4 + wv);
    if (widx >= a.total) return;                                         // whole waves only: no workgroup barrier below
    const int chunk = cfn_uni((int)(widx % a.nchunks));
    const long nc = cfn_uni((long)(widx / a.nchunks));
    const int c = cfn_uni((int)(nc % a.C));
    const int T = a.T, t0 = chunk * TO;
    float* img = smem + wv * NF * FR;

    const bool on = lane < U;
    const int lofs = on ? lane * 16 : OOB;
    __amdgpu_buffer_rsrc_t rx = cfn_rsrc(a.x + nc * (long)T * P, (unsigned)((long)T * P * 4));
    __amdgpu_buffer_rsrc_t ry = cfn_rsrc(a.y + nc * (long)T * P, (unsigned)((long)T * P * 4));
    fl_f4 R[NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) {
        const int t = t0 - 1 + k;
        const bool tv = t >= 0 && t < T;                                 // wave uniform
        R[k] = __builtin_bit_cast(fl_f4, __builtin_amdgcn_raw_buffer_load_b128(rx, tv ? lofs : OOB, cfn_uni(tv ? t * P * 4 : 0), 0));
    }
    float wr[27];
#pragma unroll
    for (int j = 0; j < 27; ++j) wr[j] = cfn_uni(a.w[(long)c * 27 + j]);
    const float pa = cfn_uni(a.A ? (float)a.A[nc] : 1.0f);
    const float pb = cfn_uni(a.A ? (float)a.B[nc] : 0.0f);
    const float act_lo = a.act == CFN_ACT_RELU ? 0.0f : -__builtin_inff();

    // zero rows 0 and 15 of every frame: NF x 2 x 14 floats = NF x 14 float2
    for (int i = lane; i < NF * 14; i += 64) {
        const int f = i / 14, j = i - f * 14;
        *reinterpret_cast<fl_p2*>(img + f * FR + (j >= 7 ? 15 * PIT + (j - 7) * 2 : j * 2)) = (fl_p2){0.0f, 0.0f};
    }
    if (on) {
        const int e0 = lane * 4, e2 = e0 + 2;
        float* d0 = img + (e0 / W + 1) * PIT + e0 % W;
        float* d1 = img + (e2 / W + 1) * PIT + e2 % W;
#pragma unroll
        for (int k = 0; k < NF; ++k) {
            const int t = t0 - 1 + k;
            const float m = (t >= 0 && t < T) ? 1.0f : 0.0f;
            const fl_f4 v = R[k];
            *reinterpret_cast<fl_p2*>(d0 + k * FR) = (fl_p2){fmaxf(fmaf(v.x, pa, pb), act_lo) * m, fmaxf(fmaf(v.y, pa, pb), act_lo) * m};
            *reinterpret_cast<fl_p2*>(d1 + k * FR) = (fl_p2){fmaxf(fmaf(v.z, pa, pb), act_lo) * m, fmaxf(fmaf(v.w, pa, pb), act_lo) * m};
        }
    }
    // LDS operations of a wave run in order; only the compiler has to be told
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    float st1 = 0.0f, st2 = 0.0f;
    if (on) {
        const int ur = lane / 7, uc = lane - ur * 7;
        fl_p2 acc[TO][2];                                                // [frame][row of the 2 x 2 block]
#pragma unroll
        for (int j = 0; j < TO; ++j) acc[j][0] = acc[j][1] = (fl_p2){0.0f, 0.0f};
        const float* base = img + (2 * ur) * PIT + 2 * uc;               // image row 2 ur = plane row 2 ur - 1
        const int eL = uc == 0 ? 0 : -1, eR = uc == 6 ? 1 : 2;
        const float mL = uc == 0 ? 0.0f : 1.0f, mR = uc == 6 ? 0.0f : 1.0f;
        const int yo = ((2 * ur) * W + 2 * uc) * 4;
#pragma unroll
        for (int k = 0; k < NF; ++k) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {                                // image rows 2 ur + r
                const float* q = base + k * FR + r * PIT;
                const float q0 = q[eL] * mL, q3 = q[eR] * mR;
                const fl_p2 v1 = *reinterpret_cast<const fl_p2*>(q);
                const fl_p2 v0 = {q0, v1.x}, v2 = {v1.y, q3};
#pragma unroll
                for (int i = 0; i < 2; ++i) {                            // output row 2 ur + i reads image rows 2 ur + i + kh
                    const int kh = r - i;
                    if (kh >= 0 && kh < 3) {
#pragma unroll
                        for (int kt = 0; kt < 3; ++kt) {
                            const int j = k - kt;
                            if (j >= 0 && j < TO) {
                                const float w0 = wr[kt * 9 + kh * 3 + 0], w1 = wr[kt * 9 + kh * 3 + 1], w2 = wr[kt * 9 + kh * 3 + 2];
                                acc[j][i] = __builtin_elementwise_fma((fl_p2){w0, w0}, v0, __builtin_elementwise_fma((fl_p2){w1, w1}, v1,
                                            __builtin_elementwise_fma((fl_p2){w2, w2}, v2, acc[j][i])));
                            }
                        }
                    }
                }
            }
            if (k >= 2) {
                const int j = k - 2, t = t0 + j;
                const bool emit = t < T;
                const int so = cfn_uni(emit ? t * P * 4 : 0);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const fl_p2 y = acc[j][i];
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(fl_u2, y), ry, emit ? yo + i * W * 4 : OOB, so, 0);
                    const fl_p2 ym = y * (emit ? 1.0f : 0.0f);
                    st1 += ym.x + ym.y;
                    st2 += ym.x * y.x + ym.y * y.y;
                }
            }
        }
    }
    if (a.s1) {
        st1 = cfn_wave_sum(st1); st2 = cfn_wave_sum(st2);
        if (lane == 0) { cfn_add64(&a.s1[nc], (double)st1, det_keys); cfn_add64(&a.s2[nc], (double)st2, det_keys); }
    }
}

// Stride 2 (112 -> 56, 56 -> 28, 28 -> 14: conv2 of the first block of layers 1-3): a workgroup per (sample, channel, chunk of TO output
// frames, band of RBO output rows).  A band reads 2 RBO + 1 input rows; the LDS image keeps the EVEN and the ODD input columns of a row in
// separate halves (E[i] = column 2 i, O[i] = column 2 i + 1), so that the taps of two adjacent outputs (j, j + 1) are the natural pairs
// (O[j-1], O[j]), (E[j], E[j+1]), (O[j], O[j+1]); a loaded float4 is two 8-byte LDS stores.  A compute lane owns 2 adjacent outputs of one
// row for all TO frames (the compute lanes fill waves 0-1 densely: idle lanes still cost issue slots).
template <int W, int RBO, int TO>                    // W: OUTPUT width (square planes; the input plane is 2W x 2W)
__global__ __launch_bounds__(256, 4) void dw3d_flat_s2_fwd_kernel(const DwFlatArgs a) {
    constexpr int WI = 2 * W, W4 = WI / 4, W2 = W / 2, NB = W / RBO, IR = 2 * RBO + 1, PIT = WI, NF = TO + 2, FR = IR * PIT;
    constexpr int PI = WI * WI, PO = W * W, OOB = 0x7fff0000;
    constexpr int NLOAD = IR * W4, NCOMP = RBO * W2;
    static_assert(NLOAD <= 256 && NCOMP <= 256 && W % RBO == 0 && W % 2 == 0, "geometry");
    __shared__ __attribute__((aligned(16))) float img[NF * FR];
    __shared__ float red[8];

    const int tid = threadIdx.x;
    const unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    const int band = cfn_uni((int)(L % NB));
    const unsigned rest = cfn_uni(L / NB);
    const int chunk = cfn_uni((int)(rest % (unsigned)a.nchunks));
    const long nc = cfn_uni((int)(rest / (unsigned)a.nchunks));
    const int c = cfn_uni((int)(nc % a.C));
    const int T = a.T, t0 = chunk * TO;

    const int lr = tid / W4, lc = tid - lr * W4;
    const int grow = 2 * band * RBO - 1 + lr;                            // input row (never beyond the plane: even input sizes)
    const bool lvalid = tid < NLOAD && grow >= 0;
    const int lofs = lvalid ? (grow * WI + lc * 4) * 4 : OOB;
    __amdgpu_buffer_rsrc_t rx = cfn_rsrc(a.x + nc * (long)T * PI, (unsigned)((long)T * PI * 4));
    __amdgpu_buffer_rsrc_t ry = cfn_rsrc(a.y + nc * (long)T * PO, (unsigned)((long)T * PO * 4));

    fl_f4 R[NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) {
        const int t = t0 - 1 + k;
        const bool tv = t >= 0 && t < T;                                 // workgroup uniform
        R[k] = __builtin_bit_cast(fl_f4, __builtin_amdgcn_raw_buffer_load_b128(rx, tv ? lofs : OOB, cfn_uni(tv ? t * PI * 4 : 0), 0));
    }
    float wr[27];
#pragma unroll
    for (int j = 0; j < 27; ++j) wr[j] = cfn_uni(a.w[(long)c * 27 + j]);
    const float pa = cfn_uni(a.A ? (float)a.A[nc] : 1.0f);
    const float pb = cfn_uni(a.A ? (float)a.B[nc] : 0.0f);
    const float act_lo = a.act == CFN_ACT_RELU ? 0.0f : -__builtin_inff();      // none / ReLU only (the planner checks)

    if (tid < NLOAD) {
        float* dst = img + lr * PIT + 2 * lc;
#pragma unroll
        for (int k = 0; k < NF; ++k) {
            const int t = t0 - 1 + k;
            const float m = (lvalid && t >= 0 && t < T) ? 1.0f : 0.0f;   // zero padding is applied AFTER the prologue
            const fl_f4 v = R[k];
            const float x0 = fmaxf(fmaf(v.x, pa, pb), act_lo) * m, x1 = fmaxf(fmaf(v.y, pa, pb), act_lo) * m;
            const float x2 = fmaxf(fmaf(v.z, pa, pb), act_lo) * m, x3 = fmaxf(fmaf(v.w, pa, pb), act_lo) * m;
            *reinterpret_cast<fl_p2*>(dst + k * FR) = (fl_p2){x0, x2};
            *reinterpret_cast<fl_p2*>(dst + k * FR + W) = (fl_p2){x1, x3};
        }
    }
    __syncthreads();

    float st1 = 0.0f, st2 = 0.0f;
    if (tid < NCOMP) {
        const int orow = tid / W2, j = 2 * (tid - orow * W2);            // outputs (orow, j), (orow, j + 1) of the band
        fl_p2 acc[TO];
#pragma unroll
        for (int f = 0; f < TO; ++f) acc[f] = (fl_p2){0.0f, 0.0f};
        const float* base = img + (2 * orow) * PIT + j;                  // output row orow reads image rows 2 orow .. 2 orow + 2
        const int eL = j == 0 ? W : W - 1;                               // O[j - 1]; the column left of 0 is not stored: valid address x 0
        const float mL = j == 0 ? 0.0f : 1.0f;
        const int yo = ((band * RBO + orow) * W + j) * 4;
#pragma unroll
        for (int k = 0; k < NF; ++k) {
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const float* q = base + k * FR + kh * PIT;
                const float om = q[eL] * mL;
                const fl_p2 ve = *reinterpret_cast<const fl_p2*>(q), vo = *reinterpret_cast<const fl_p2*>(q + W);
                const fl_p2 v0 = {om, vo.x};
#pragma unroll
                for (int kt = 0; kt < 3; ++kt) {
                    const int f = k - kt;                                // output frame t0 + f reads input frames f .. f + 2 (k = f + kt)
                    if (f >= 0 && f < TO) {
                        const float w0 = wr[kt * 9 + kh * 3 + 0], w1 = wr[kt * 9 + kh * 3 + 1], w2 = wr[kt * 9 + kh * 3 + 2];
                        acc[f] = __builtin_elementwise_fma((fl_p2){w0, w0}, v0, __builtin_elementwise_fma((fl_p2){w1, w1}, ve,
                                 __builtin_elementwise_fma((fl_p2){w2, w2}, vo, acc[f])));
                    }
                }
            }
            if (k >= 2) {                                                // output frame k - 2 is complete
                const int f = k - 2, t = t0 + f;
                const bool emit = t < T;
                const fl_p2 y = acc[f];
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(fl_u2, y), ry, emit ? yo : OOB, cfn_uni(emit ? t * PO * 4 : 0), 0);
                const fl_p2 ym = y * (emit ? 1.0f : 0.0f);
                st1 += ym.x + ym.y;
                st2 += ym.x * y.x + ym.y * y.y;
            }
        }
    }
    if (a.s1) {
        const int wave = tid >> 6, lane = tid & 63;
        st1 = cfn_wave_sum(st1); st2 = cfn_wave_sum(st2);
        if (lane == 0) { red[wave] = st1; red[4 + wave] = st2; }
        __syncthreads();
        if (tid == 0) {
            cfn_add64(&a.s1[nc], (double)(red[0] + red[1] + red[2] + red[3]));
            cfn_add64(&a.s2[nc], (double)(red[4] + red[5] + red[6] + red[7]));
        }
    }
}

// 14 -> 7 (stride 2, first block of layer 4): a WAVE per (sample, channel, chunk of TO frames); the loader and the image of
// dw3d_flat14_fwd_kernel, a lane per output position (49 of 64 lanes, scalar FMAs).  Same-box steady state, 8 clips x T = 256: band kernel
// 199 us, this kernel 171.  (The same kernel for the 7x7 stride-1 planes -- the 10 frames of an item are one run of 490 floats, 4-byte loads --
// was measured at 110 us with 8-frame items and 94 us with 16-frame items against 95 us of dw3d_small_fwd_kernel: not kept.)
template <int TO>
__global__ __launch_bounds__(256, 4) void dw3d_flat14to7_fwd_kernel(const DwFlatArgs a) {
    unsigned long long* const det_keys = cfn_det_keys();
    constexpr int WI = 14, PI = 196, PO = 49, IR = 16, FR = IR * WI, NF = TO + 2, OOB = 0x7fff0000;
    __shared__ __attribute__((aligned(16))) float smem[4 * NF * FR];
    const int lane = threadIdx.x & 63, wv = cfn_uni((int)(threadIdx.x >> 6));
    const unsigned L = cfn_xcd_remap(blockIdx.x, gridDim.x);
    const long widx = cfn_uni((long)L * 4 + wv);
    if (widx >= a.total) return;                                         // whole waves only: no workgroup barrier below
    const int chunk = cfn_uni((int)(widx % a.nchunks));
    const long nc = cfn_uni((long)(widx / a.nchunks));
    const int c = cfn_uni((int)(nc % a.C));
    const int T = a.T, t0 = chunk * TO;
    float* img = smem + wv * NF * FR;
    __amdgpu_buffer_rsrc_t rx = cfn_rsrc(a.x + nc * (long)T * PI, (unsigned)((long)T * PI * 4));
    __amdgpu_buffer_rsrc_t ry = cfn_rsrc(a.y + nc * (long)T * PO, (unsigned)((long)T * PO * 4));
    const bool on = lane < 49;

    fl_f4 R[NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) {
        const int t = t0 - 1 + k;
        const bool tv = t >= 0 && t < T && on;
        R[k] = __builtin_bit_cast(fl_f4, __builtin_amdgcn_raw_buffer_load_b128(rx, tv ? lane * 16 : OOB, cfn_uni(tv ? t * PI * 4 : 0), 0));
    }
    float wr[27];
#pragma unroll
    for (int j = 0; j < 27; ++j) wr[j] = cfn_uni(a.w[(long)c * 27 + j]);
    const float pa = cfn_uni(a.A ? (float)a.A[nc] : 1.0f);
    const float pb = cfn_uni(a.A ? (float)a.B[nc] : 0.0f);
    const float act_lo = a.act == CFN_ACT_RELU ? 0.0f : -__builtin_inff();

    // zero row 0 of every frame (input row -1; row 15 = input row 14 is never read at stride 2)
    for (int i = lane; i < NF * 7; i += 64) {
        const int f = i / 7, j = i - f * 7;
        *reinterpret_cast<fl_p2*>(img + f * FR + 2 * j) = (fl_p2){0.0f, 0.0f};
    }
    if (on) {
        const int e0 = lane * 4, e2 = e0 + 2;
        float* d0 = img + (e0 / WI + 1) * WI + e0 % WI;
        float* d1 = img + (e2 / WI + 1) * WI + e2 % WI;
#pragma unroll
        for (int k = 0; k < NF; ++k) {
            const int t = t0 - 1 + k;
            const float mk = (t >= 0 && t < T) ? 1.0f : 0.0f;           // zero padding is applied AFTER the prologue
            const fl_f4 v = R[k];
            *reinterpret_cast<fl_p2*>(d0 + k * FR) = (fl_p2){fmaxf(fmaf(v.x, pa, pb), act_lo) * mk, fmaxf(fmaf(v.y, pa, pb), act_lo) * mk};
            *reinterpret_cast<fl_p2*>(d1 + k * FR) = (fl_p2){fmaxf(fmaf(v.z, pa, pb), act_lo) * mk, fmaxf(fmaf(v.w, pa, pb), act_lo) * mk};
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    float st1 = 0.0f, st2 = 0.0f;
    if (on) {
        const int orow = lane / 7, oc = lane - orow * 7;
        float acc[TO];
#pragma unroll
        for (int j = 0; j < TO; ++j) acc[j] = 0.0f;
        // output (orow, oc) reads image rows 2 orow .. 2 orow + 2 (image row = input row + 1), input columns 2 oc - 1 .. 2 oc + 1
        const float* base = img + (2 * orow) * WI + 2 * oc;
        const int eL = oc == 0 ? 0 : -1;                                 // the column left of 0 is not stored: valid address x 0
        const float mL = oc == 0 ? 0.0f : 1.0f;
#pragma unroll
        for (int k = 0; k < NF; ++k) {
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const float* q = base + k * FR + kh * WI;
                const float q0 = q[eL] * mL, q1 = q[0], q2 = q[1];
#pragma unroll
                for (int kt = 0; kt < 3; ++kt) {
                    const int j = k - kt;
                    if (j >= 0 && j < TO)
                        acc[j] = fmaf(wr[kt * 9 + kh * 3 + 0], q0, fmaf(wr[kt * 9 + kh * 3 + 1], q1, fmaf(wr[kt * 9 + kh * 3 + 2], q2, acc[j])));
                }
            }
            if (k >= 2) {
                const int j = k - 2, t = t0 + j;
                const bool emit = t < T;
                const float y = acc[j];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), ry, emit ? lane * 4 : OOB, cfn_uni(emit ? t * PO * 4 : 0), 0);
                const float ym = emit ? y : 0.0f;
                st1 += ym;
                st2 = fmaf(ym, y, st2);
            }
        }
    }
    if (a.s1) {
        st1 = cfn_wave_sum(st1); st2 = cfn_wave_sum(st2);
        if (lane == 0) { cfn_add64(&a.s1[nc], (double)st1, det_keys); cfn_add64(&a.s2[nc], (double)st2, det_keys); }
    }
}

// returns -1 when the shape is not handled; probe: 0 = handled, nothing launched; otherwise the launch status
int dw_flat_fwd_try(const float* x, const double* A, const double* B, int act, const float* w, float* y, double* sum, double* sumsq,
                    int N, int C, int T, int Hi, int Wi, int stride, hipStream_t st, bool probe) {
    // bit mask of the planes served: stride 1: 1 = 56x56, 2 = 28x28, 4 = 14x14; stride 2: 8 = 112 -> 56, 16 = 56 -> 28, 32 = 28 -> 14, 64 = 14 -> 7
    static const int enabled = getenv("CFN_DW_FLAT") ? atoi(getenv("CFN_DW_FLAT")) : 127;
    static const int to_env = getenv("CFN_DW_FLAT_TO") ? atoi(getenv("CFN_DW_FLAT_TO")) : 0;
    if (Hi != Wi) return -1;
    if (stride == 2 && Hi == 14) {
        if (!(enabled & 64)) return -1;
        if (act != CFN_ACT_NONE && act != CFN_ACT_RELU && A != nullptr) return -1;
        if ((long)T * Hi * Wi * 4 >= 0x7fff0000L || (((uintptr_t)x | (uintptr_t)y) & 15) != 0) return -1;
        const int TO = to_env == 4 || to_env == 8 ? to_env : (T >= 12 ? 8 : 4);
        const long nch = (T + TO - 1) / TO, items = (long)N * C * nch, blocks = (items + 3) / 4;
        if (blocks >= 0x7fffffffL) return -1;
        if (probe) return 0;
        DwFlatArgs a = {x, A, B, w, y, sum, sumsq, N, C, T, act, (int)nch, items};
        if (TO == 8) hipLaunchKernelGGL((dw3d_flat14to7_fwd_kernel<8>), dim3((unsigned)blocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((dw3d_flat14to7_fwd_kernel<4>), dim3((unsigned)blocks), dim3(256), 0, st, a);
        return cfn_check_launch("dwconv3d flat 14->7 forward");
    }
    if (stride == 2) {
        if (Hi != 112 && Hi != 56 && Hi != 28) return -1;
        if (!(enabled & (Hi == 112 ? 8 : Hi == 56 ? 16 : 32))) return -1;
        if (act != CFN_ACT_NONE && act != CFN_ACT_RELU && A != nullptr) return -1;
        if ((long)T * Hi * Wi * 4 >= 0x7fff0000L || (((uintptr_t)x | (uintptr_t)y) & 15) != 0) return -1;
        // same-box steady state, 8 clips x T = 256, dwcp.hip / this kernel at TO = 8 / 6 / 4: 112 -> 56 1373 / 1211 / 1260 / 1214 us, 56 -> 28
        // 686 / 622 / 618 / 623, 28 -> 14 358 / 309 / 302 / 311; 2-row bands on 112 -> 56 (7 workgroups per CU, 1.25 x row re-reads): 1250
        const int TO = to_env == 4 || to_env == 8 ? to_env : (T >= 12 ? 8 : 4);
        const int NB = Hi == 112 ? 14 : Hi == 56 ? 4 : 1;
        const long nch = (T + TO - 1) / TO, blocks = (long)N * C * nch * NB;
        if (blocks >= 0x7fffffffL) return -1;
        if (probe) return 0;
        DwFlatArgs a = {x, A, B, w, y, sum, sumsq, N, C, T, act, (int)nch, blocks};
#define CFN_FLAT_GO(...) hipLaunchKernelGGL((dw3d_flat_s2_fwd_kernel<__VA_ARGS__>), dim3((unsigned)blocks), dim3(256), 0, st, a)
        if (Hi == 112) { if (TO == 8) CFN_FLAT_GO(56, 4, 8); else CFN_FLAT_GO(56, 4, 4); }
        else if (Hi == 56) { if (TO == 8) CFN_FLAT_GO(28, 7, 8); else CFN_FLAT_GO(28, 7, 4); }
        else { if (TO == 8) CFN_FLAT_GO(14, 14, 8); else CFN_FLAT_GO(14, 14, 4); }
#undef CFN_FLAT_GO
        return cfn_check_launch("dwconv3d flat stride-2 forward");
    }
    if (stride != 1 || (Hi != 56 && Hi != 28 && Hi != 14)) return -1;
    if (!(enabled & (Hi == 56 ? 1 : Hi == 28 ? 2 : 4))) return -1;
    if (act != CFN_ACT_NONE && act != CFN_ACT_RELU && A != nullptr) return -1;      // branch-free prologue: none / ReLU (every X3D conv2)
    if ((long)T * Hi * Wi * 4 >= 0x7fff0000L) return -1;
    if ((((uintptr_t)x | (uintptr_t)y) & 15) != 0) return -1;
    // 8 output frames per item (10 input frames: 1.25 x temporal re-reads out of L2; 4 frames: 1.5 x, measured 5 % slower); short clips: 4
    const int TO = to_env == 4 || to_env == 8 ? to_env : (T >= 12 ? 8 : 4);
    const int NB = Hi == 56 ? 4 : 1;
    const long nch = (T + TO - 1) / TO, items = (long)N * C * nch * NB;
    const long blocks = Hi == 14 ? (items + 3) / 4 : items;
    if (blocks >= 0x7fffffffL || (long)N * C * nch >= 0x7fffffffL) return -1;
    if (probe) return 0;
    DwFlatArgs a = {x, A, B, w, y, sum, sumsq, N, C, T, act, (int)nch, items};
#define CFN_FLAT_GO(K, ...) hipLaunchKernelGGL((K<__VA_ARGS__>), dim3((unsigned)blocks), dim3(256), 0, st, a)
    if (Hi == 56) { if (TO == 8) CFN_FLAT_GO(dw3d_flat_fwd_kernel, 56, 14, 8); else CFN_FLAT_GO(dw3d_flat_fwd_kernel, 56, 14, 4); }
    else if (Hi == 28) { if (TO == 8) CFN_FLAT_GO(dw3d_flat_fwd_kernel, 28, 28, 8); else CFN_FLAT_GO(dw3d_flat_fwd_kernel, 28, 28, 4); }
    else { if (TO == 8) CFN_FLAT_GO(dw3d_flat14_fwd_kernel, 8); else CFN_FLAT_GO(dw3d_flat14_fwd_kernel, 4); }
#undef CFN_FLAT_GO
    return cfn_check_launch("dwconv3d flat forward");
}
