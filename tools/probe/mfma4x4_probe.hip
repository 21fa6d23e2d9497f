// Layout probe for v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer products per wave): which lane / register holds D[i][j]?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float __attribute__((ext_vector_type(4))) f4;
__global__ void k(float* out) {
    const int l = threadIdx.x;
    const float a = 1.0f + l;            // A[i] of block b: lane 4b+i  -> value 1 + lane
    const float b = 1000.0f * (1 + l);   // B[j] of block b: lane 4b+j  -> value 1000 (1 + lane)
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
    float* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); float h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    // expectation under "lane 4b+j, register i holds A[4b+i] * B[4b+j]":  (1 + 4b + i) * 1000 (1 + lane)
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        const float e = (1.0f + (l / 4) * 4 + r) * 1000.0f * (1 + l);
        if (h[l * 4 + r] != e) ++bad;
    }
    printf("layout 'lane 4b+j, reg i = A[4b+i]*B[4b+j]': %s (%d mismatches)\n", bad ? "NO" : "YES", bad);
    for (int l : {0, 1, 5, 63}) printf("lane %2d: %g %g %g %g\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    return 0;
}
