// Probe for the data-level signature of the round-3 weight-gradient race: in a wave that converts operands while the other wave of
// its SIMD multiplies, the packed FMA right behind `ds_read_b64 coefficient pair; s_waitcnt lgkmcnt(0)` used the HIGH register of the returned
// pair as zero in lanes 48-63.  Here: 8 waves per workgroup, one workgroup per CU; waves 0-3 stream ds_read_b128 + v_mfma_f32_32x32x16_bf16
// (the multiplying waves), waves 4-7 repeat
//     poison v[lo:hi]; ds_read_b64 v[lo:hi], table[row]; s_waitcnt lgkmcnt(0); v_pk_fma_f32 r, x, v[lo:hi], v[lo:hi] op_sel:[0,0,1] op_sel_hi:[1,0,1]
// and compare r with fmaf(x, A, B) bit for bit.  Prints mismatches by lane quarter and by what the wrong addend looked like (poison / zero / other).
//   hipcc --offload-arch=gfx950 -O3 lds_ret_probe.hip -o lds_ret_probe && ./lds_ret_probe [iters] [mode]
//   mode 0: barrier-synchronised phases as in the kernel; 1: free running
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2v __attribute__((ext_vector_type(2)));

#define POISON 3.0e30f

__global__ __launch_bounds__(512) void probe(const float* __restrict__ xin, unsigned long long* cnt, float* sink, int iters, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned char img[2][96 * 80 * 3];      // operand images (multiplying waves read them)
    __shared__ float2 tab[128];                                                        // coefficient table (A, B) per row
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    for (int i = tid; i < 128; i += 512) tab[i] = float2{1.0f + 0.01f * i, -0.5f + 0.003f * i};
    for (int i = tid; i < (int)sizeof(img) / 4; i += 512) reinterpret_cast<unsigned*>(img)[i] = 0x3f803f80u + i;
    __syncthreads();
    const int row = (tid >> 3) & 127;
    const float A = 1.0f + 0.01f * row, B = -0.5f + 0.003f * row;
    const unsigned taddr = (unsigned)(uintptr_t)&tab[row];
    f16v acc0 = (f16v)0.0f, acc1 = (f16v)0.0f;
    unsigned long long bad_q[4] = {0, 0, 0, 0}, kind[3] = {0, 0, 0};
    float x0 = xin[(blockIdx.x * 512 + tid) * 2], x1 = xin[(blockIdx.x * 512 + tid) * 2 + 1];
    for (int it = 0; it < iters; ++it) {
        if (mode == 0) __syncthreads();
        if (wave < 4) {
            const unsigned char* b = img[it & 1];
            const int r = lane & 31, kg = lane >> 5;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    bf16x8 a[3], c[3];
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
                        a[s] = *reinterpret_cast<const bf16x8*>(b + s * 32 * 80 + ((t * 32 + r) % 32) * 80 + kg * 16 + kb * 32);
                        c[s] = *reinterpret_cast<const bf16x8*>(b + (3 + s) * 32 * 80 + r * 80 + kg * 16 + kb * 32);
                    }
                    f16v& acc = t ? acc1 : acc0;
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], c[0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], c[2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], c[1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], c[0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], c[1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], c[0], acc, 0, 0, 0);
                }
            }
        } else {
            // three conversion blocks per phase, as the kernel's staging-only waves ran them
#pragma unroll
            for (int blk = 0; blk < 3; ++blk) {
                f2v xv = {x0 + blk, x1 - blk}, res;
                f2v co = {POISON, POISON};
                asm volatile(
                    "ds_read_b64 %1, %3\n\t"
                    "s_waitcnt lgkmcnt(0)\n\t"
                    "v_pk_fma_f32 %0, %2, %1, %1 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"
                    : "=&v"(res), "+v"(co)
                    : "v"(xv), "v"(taddr)
                    : "memory");
                const float e0 = __builtin_fmaf(xv.x, A, B), e1 = __builtin_fmaf(xv.y, A, B);
                const float r0 = res.x, r1 = res.y;       // (by value: __builtin_bit_cast on an ext-vector element reads element 0, hipcc 7.2)
                if (r0 != e0 || r1 != e1) {
                    bad_q[lane >> 4]++;
                    if (atomicAdd(&cnt[7], 1ull) == 0) { sink[1] = res.x; sink[2] = e0; sink[3] = xv.x; sink[4] = A; sink[5] = B; sink[6] = co.x; sink[7] = co.y; sink[8] = res.y; sink[9] = e1; sink[10] = xv.y; }
                    const float w = (res.x != e0) ? res.x : res.y, xx = (res.x != e0) ? xv.x : xv.y;
                    if (w == __builtin_fmaf(xx, A, 0.0f) || w == xx * A) kind[1]++;             // addend read as zero
                    else if (!(w < 1e29f)) kind[0]++;                                            // poison (stale register)
                    else kind[2]++;
                }
                // some conversion-like VALU + LDS stores into the other image (what the staging waves do)
                float v = res.x * res.y;
                v = v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.442695f * v));
                *reinterpret_cast<float2*>(img[(it + 1) & 1] + (blk * 32 + (row & 31)) * 80 + (tid & 7) * 8) = float2{v, res.y};
                x0 += 1e-3f * v; x1 -= 1e-3f;
                if (!(x0 < 4.0f && x0 > -4.0f)) x0 = 0.25f;
                if (!(x1 < 4.0f && x1 > -4.0f)) x1 = -0.25f;
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    if (s == 12345.678f) sink[0] = s;
#pragma unroll
    for (int q = 0; q < 4; ++q) if (bad_q[q]) atomicAdd(&cnt[q], bad_q[q]);
#pragma unroll
    for (int q = 0; q < 3; ++q) if (kind[q]) atomicAdd(&cnt[4 + q], kind[q]);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000, mode = argc > 2 ? atoi(argv[2]) : 0, launches = argc > 3 ? atoi(argv[3]) : 20;
    const int blocks = 256;
    std::vector<float> hx(blocks * 512 * 2);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) % 2001) / 1000.0f - 1.0f;
    float *dx, *sink; unsigned long long* cnt;
    hipMalloc(&dx, hx.size() * 4); hipMalloc(&sink, 64); hipMalloc(&cnt, 8 * 8);
    hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemset(cnt, 0, 64);
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 0, 0, dx, cnt, sink, iters, mode);
    hipDeviceSynchronize();
    unsigned long long h[8]; hipMemcpy(h, cnt, 64, hipMemcpyDeviceToHost);
    const double ev = (double)launches * blocks * 256 * 3 * iters;
    printf("mode %d: %.3g packed FMAs behind an LDS coefficient read; mismatches by lane quarter [0-15, 16-31, 32-47, 48-63] = %llu %llu %llu %llu; addend seen as poison (stale register) %llu, as zero %llu, other %llu\n",
           mode, ev, h[0], h[1], h[2], h[3], h[4], h[5], h[6]);
    float hs[11]; hipMemcpy(hs, sink, 44, hipMemcpyDeviceToHost);
    if (h[7]) printf("first mismatch: res.x %.9g expected %.9g (x %.9g A %.9g B %.9g; pair read %.9g %.9g) res.y %.9g expected %.9g (x %.9g)\n", hs[1], hs[2], hs[3], hs[4], hs[5], hs[6], hs[7], hs[8], hs[9], hs[10]);
    return 0;
}
