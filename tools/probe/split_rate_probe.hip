// Cost of the 3-term bf16 split of a PAIR of fp32 values (the conversion every split-bf16 kernel runs once per operand element) in three
// formulations, and of the single instructions they are made of: cycles per pair / per wave instruction and SIMD at 1, 2 and 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/split_rate_probe.hip -o tools/probe/split_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector((f2){a, b}, bf2)); }
__device__ __forceinline__ float lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// KIND 0: round-to-nearest terms (the kernels' pws_split<3>): cvt_pk, 2 x (shl, and, pk_add / 2 sub, cvt_pk)
// KIND 1: truncated leading terms, rounded last term: 2 x (2 and, pk_add), 2 perm, cvt_pk
// KIND 2: single instructions: v_cvt_pk_bf16_f32      KIND 3: v_and_b32      KIND 4: v_perm_b32      KIND 5: v_pk_add_f32     KIND 6: v_sub_f32
template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, float seed) {
    constexpr int NP = 8;                         // independent pairs in flight per lane
    float a[NP], b[NP];
    unsigned acc = 0;
    for (int i = 0; i < NP; ++i) { a[i] = seed + threadIdx.x * 0.37f + i; b[i] = seed * 1.7f + threadIdx.x * 0.11f - i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            float x = a[i], y = b[i];
            if (KIND == 0) {
                const unsigned p0 = pack(x, y);
                x -= lo(p0); y -= hi(p0);
                const unsigned p1 = pack(x, y);
                x -= lo(p1); y -= hi(p1);
                const unsigned p2 = pack(x, y);
                acc ^= p0 + p1 + p2;
                a[i] += __builtin_bit_cast(float, (p2 & 0x007f0000u) | 0x3f800000u);     // keep the chain alive with one cheap op
            } else if (KIND == 1) {
                const unsigned xa = __builtin_bit_cast(unsigned, x) & 0xffff0000u, ya = __builtin_bit_cast(unsigned, y) & 0xffff0000u;
                const unsigned p0 = __builtin_amdgcn_perm(ya, xa, 0x07060302u);
                x -= __builtin_bit_cast(float, xa); y -= __builtin_bit_cast(float, ya);
                const unsigned xb = __builtin_bit_cast(unsigned, x) & 0xffff0000u, yb = __builtin_bit_cast(unsigned, y) & 0xffff0000u;
                const unsigned p1 = __builtin_amdgcn_perm(yb, xb, 0x07060302u);
                x -= __builtin_bit_cast(float, xb); y -= __builtin_bit_cast(float, yb);
                const unsigned p2 = pack(x, y);
                acc ^= p0 + p1 + p2;
                a[i] += __builtin_bit_cast(float, (p2 & 0x007f0000u) | 0x3f800000u);
            } else if (KIND == 2) {
                const unsigned p = pack(x, y); a[i] = __builtin_bit_cast(float, p | 0x3f000000u);
            } else if (KIND == 3) {
                a[i] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & __builtin_bit_cast(unsigned, y));
            } else if (KIND == 4) {
                a[i] = __builtin_bit_cast(float, __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), 0x07060302u));
            } else if (KIND == 5) {
                const f2 r = (f2){x, y} + (f2){y, x}; a[i] = r.x; b[i] = r.y;
            } else {
                a[i] = x - y;
            }
        }
    }
    unsigned r = acc;
    for (int i = 0; i < NP; ++i) r ^= __builtin_bit_cast(unsigned, a[i]) + __builtin_bit_cast(unsigned, b[i]);
    if (r == 0x12345678u) out[0] = r;
}
template <int KIND> void run(const char* name, int wps, double per) {
    unsigned* d; hipMalloc(&d, 4);
    const int iters = 4000, blocks = 256 * wps;
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0f); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / ((double)iters * 8 * wps);
    printf("%-44s waves/SIMD %d: %7.2f ns per %s and SIMD (= %6.1f cycles at 2.4 GHz)\n", name, wps, ns, per == 1 ? "pair" : "instruction", ns * 2.4);
    hipFree(d);
}
int main() {
    for (int w : {1, 2, 4}) {
        run<0>("split, round-to-nearest terms (kernels)", w, 1);
        run<1>("split, truncated terms + v_perm", w, 1);
        run<2>("v_cvt_pk_bf16_f32 (+ v_or)", w, 0);
        run<3>("v_and_b32", w, 0);
        run<4>("v_perm_b32", w, 0);
        run<5>("v_pk_add_f32", w, 0);
        run<6>("v_sub_f32", w, 0);
    }
    return 0;
}
