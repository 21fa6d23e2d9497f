// Issue rate of the small fp32 MFMAs and of the vector FMA forms on gfx950 (cycles per wave instruction at one wave per SIMD
// and at four): decides whether a depthwise conv should run its taps on v_mfma_f32_4x4x1 or on the vector ALU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float __attribute__((ext_vector_type(4))) f4;
typedef float __attribute__((ext_vector_type(2))) f2;
typedef float __attribute__((ext_vector_type(16))) f16;
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    f4 c[8]; f2 p[8]; float s[8]; f16 big[2];
    for (int i = 0; i < 8; ++i) { c[i] = (f4){0, 0, 0, 0}; p[i] = (f2){0, 0}; s[i] = 0; }
    big[0] = big[1] = (f16)(0.0f);
    const float av = a + threadIdx.x, bv = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) c[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, c[i], 0, 0, 0);
            if (KIND == 1) p[i] = __builtin_elementwise_fma((f2){av, av}, (f2){bv, bv}, p[i]);
            if (KIND == 2) s[i] = fmaf(av, bv, s[i]);
            if (KIND == 3) big[i & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, big[i & 1], 0, 0, 0);
            if (KIND == 4) c[i] = __builtin_amdgcn_mfma_f32_4x4x4f16((__fp16 __attribute__((ext_vector_type(4)))){(__fp16)av, 0, 0, 0}, (__fp16 __attribute__((ext_vector_type(4)))){(__fp16)bv, 0, 0, 0}, c[i], 0, 0, 0);
        }
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += c[i].x + c[i].y + p[i].x + p[i].y + s[i];
    r += big[0][0] + big[1][3];
    if (r == 12345.678f) out[0] = r;
}
template <int KIND> void run(const char* name, int wavesPerSimd) {
    float* d; hipMalloc(&d, 4);
    const int iters = 20000, blocks = 256 * wavesPerSimd;
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0f, 2.0f); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f, 2.0f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns_per_instr = ms * 1e6 / ((double)iters * 8 * wavesPerSimd);
    printf("%-34s waves/SIMD %d: %.2f ns per wave-instruction per SIMD (= %.1f cycles at 2.4 GHz)\n", name, wavesPerSimd, ns_per_instr, ns_per_instr * 2.4);
}
int main() {
    for (int w : {1, 4}) {
        run<0>("v_mfma_f32_4x4x1_16b_f32", w);
        run<3>("v_mfma_f32_32x32x2_f32", w);
        run<4>("v_mfma_f32_4x4x4_16b_f16", w);
        run<1>("v_pk_fma_f32", w);
        run<2>("v_fma_f32", w);
    }
    return 0;
}
