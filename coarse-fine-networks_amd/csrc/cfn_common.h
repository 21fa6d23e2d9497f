// Shared device/host helpers for the Coarse-Fine HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#define CFN_OK 0
#define CFN_ERR_ARG 1
#define CFN_ERR_LAUNCH 2
#define CFN_ERR_UNSUPPORTED 3

// activation codes of the load-time prologue  a = act(A[n,c] * x + B[n,c])
enum { CFN_ACT_NONE = 0, CFN_ACT_RELU = 1, CFN_ACT_SWISH = 2, CFN_ACT_SIGMOID = 3 };

extern "C" const char* cfn_last_error(void);
int cfn_fail(int code, const char* fmt, ...);  // records the message, returns code
int cfn_check_launch(const char* what);        // hipGetLastError -> CFN_ERR_LAUNCH

#define CFN_REQUIRE(cond, ...) \
    do { if (!(cond)) return cfn_fail(CFN_ERR_ARG, __VA_ARGS__); } while (0)

// optional per-kernel-family HIP-event timing (bench.py roofline leg); see capi.hip
enum { CFN_K_DWCONV_FWD = 0, CFN_K_DWCONV_BWD = 1, CFN_K_PWCONV_FWD = 2, CFN_K_PWCONV_BWD = 3,
       CFN_K_GRIDPOOL = 4, CFN_K_ELEMWISE = 5, CFN_K_STEM = 6, CFN_K_FUSION = 7, CFN_K_PWCONV_WGRAD = 8,
       CFN_K_DWCONV_WGRAD = 9, CFN_K_DENSE_FWD = 10, CFN_K_GRIDPOOL_BWD = 11, CFN_K_COUNT = 12 };
struct CfnProfScope {
    int fam; hipStream_t s; hipEvent_t e0; bool on;
    CfnProfScope(int family, hipStream_t stream, double bytes);
    ~CfnProfScope();
};

// ---------------------------------------------------------------------------------------------
// Deterministic mode (cfn_deterministic(1); SURVEY 8(b): "deterministic reductions preferred for parity tests (atomics order) -- offer a
// deterministic mode").  Every cross-workgroup fp64 accumulation of the library goes through cfn_add64().  Normally that is one fp64 atomic
// (the order in which workgroups arrive decides the rounding of the running sum -- exact unless an addend is below 2^-29 of it).  In
// deterministic mode the (address, addend) pair is appended to a record buffer instead, and after the entry point's launches capi.hip sorts
// the records by (address, addend bits) and adds every address's addends to it in that order: the result no longer depends on the order of
// arrival, by construction.  The state lives in one __constant__ variable per translation unit (no relocatable device code in this build);
// each unit registers a setter with capi.hip.
struct CfnDetState { unsigned long long* keys; unsigned long long* vals; unsigned long long* count; unsigned long long cap; };
static __constant__ CfnDetState cfn_det_dev;      // constant address space: read with ONE scalar load (a __device__ global is fetched with a vector load, and the wait for it also drains the wave's stores: the stride-2 depthwise forward kernels lost 7 %)
int cfn_det_register(void (*set)(const CfnDetState*));
static void cfn_det_tu_set(const CfnDetState* st) {
    if (hipMemcpyToSymbol(HIP_SYMBOL(cfn_det_dev), st, sizeof(*st)) != hipSuccess) (void)hipGetLastError();   // (a unit without accumulations has no symbol)
}
static const int cfn_det_tu_registered = cfn_det_register(&cfn_det_tu_set);
// (wave-per-item kernels fetch the mode word at their START with cfn_det_keys() and pass it on: the two dependent scalar loads then overlap
// the item's tensor loads instead of sitting in every wave's tail -- 2 % on the 14x14 / 7x7 depthwise forward kernels)
__device__ __forceinline__ unsigned long long* cfn_det_keys() { return cfn_det_dev.keys; }
__device__ __forceinline__ void cfn_add64(double* p, double v, unsigned long long* keys);
__device__ __forceinline__ void cfn_add64(double* p, double v) { cfn_add64(p, v, cfn_det_dev.keys); }
__device__ __forceinline__ void cfn_add64(double* p, double v, unsigned long long* const keys) {
    if (__builtin_expect(keys != nullptr, 0)) {
        const unsigned long long i = atomicAdd(cfn_det_dev.count, 1ull);
        if (i < cfn_det_dev.cap) { keys[i] = (unsigned long long)(uintptr_t)p; cfn_det_dev.vals[i] = __builtin_bit_cast(unsigned long long, v); }
    } else {
        atomicAdd(p, v);
    }
}

static inline int cfn_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// grid.y is limited to 65535: kernels indexed by (n, c) = blockIdx.y + blockIdx.z * gridDim.y get an EXACT factorisation
// gy * gz = N*C with gy, gz <= 65535, so no in-kernel range check is needed (a prime N*C > 65535 has none and is refused)
static inline bool cfn_split_nc(long NC, unsigned& gy, unsigned& gz) {
    gy = (unsigned)NC; gz = 1;
    if (NC <= 65535) return NC > 0;
    for (long d = 65535; d >= 1; --d)
        if (NC % d == 0) { gy = (unsigned)d; gz = (unsigned)(NC / d); return gz <= 65535; }
    return false;
}
static inline bool cfn_split_nc_ok(long NC) { unsigned a, b; return cfn_split_nc(NC, a, b); }

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
// Wave-uniform values the compiler cannot prove uniform (a wave index threadIdx.x >> 6, a value merged out of a divergent
// branch) put buffer descriptors / scalar offsets in VGPRs, and every buffer instruction built on them is wrapped in a
// "waterfall" loop (readfirstlane + compare + saveexec + branch: ~10 extra instructions and a taken branch per access).
// cfn_uni() states the uniformity; tools/waterfall_scan.py lists the kernels whose ISA still contains such loops.
__device__ __forceinline__ int cfn_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned cfn_uni(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ long cfn_uni(long v) {
    const unsigned lo = cfn_uni((unsigned)v), hi = cfn_uni((unsigned)((unsigned long)v >> 32));
    return (long)(((unsigned long)hi << 32) | lo);
}
__device__ __forceinline__ float cfn_uni(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
template <class T> __device__ __forceinline__ T* cfn_uni(T* p) { return (T*)cfn_uni((long)p); }

// raw buffer descriptor over `bytes` bytes at p (both stated wave uniform: see cfn_uni)
template <class T> __device__ __forceinline__ __amdgpu_buffer_rsrc_t cfn_rsrc(T* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)cfn_uni((long)p), 0, cfn_uni(bytes), 0x00020000);
}

// 16-byte buffer store.  Measured on gfx950 with hipcc 7.2 (csrc/dwt5.hip, fused backward): a VALU instruction that overwrites the
// data registers RIGHT behind a 16-byte buffer store with a register soffset can reach the register file before the store has read
// them -- lanes 12-15 / 28-31 of the second dword carried the NEXT frame's value.  LLVM's hazard recognizer inserts wait states
// for this pair only when soffset is not a register.  The asm keeps the data registers alive for two more wait states.
__device__ __forceinline__ void cfn_bst128(unsigned __attribute__((ext_vector_type(4))) data, __amdgpu_buffer_rsrc_t r, int vo, int so) {
    __builtin_amdgcn_raw_buffer_store_b128(data, r, vo, so, 0);
    asm volatile("s_nop 1" : "+v"(data));
}

// v_exp_f32 + v_rcp_f32 (1 ulp each): ~6 VALU slots instead of the ~16 of an IEEE division
__device__ __forceinline__ float cfn_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

template <int ACT>
__device__ __forceinline__ float cfn_act(float z) {
    if (ACT == CFN_ACT_RELU) return fmaxf(z, 0.0f);
    if (ACT == CFN_ACT_SWISH) return z * cfn_sigmoid(z);
    return z;
}
// d act(z) / dz ; Swish derivative as in SwishEfficient.backward (x3d_fine.py:82-86)
template <int ACT>
__device__ __forceinline__ float cfn_act_grad(float z) {
    if (ACT == CFN_ACT_RELU) return z > 0.0f ? 1.0f : 0.0f;
    if (ACT == CFN_ACT_SWISH) { float s = cfn_sigmoid(z); return s * (1.0f + z * (1.0f - s)); }
    return 1.0f;
}
__device__ __forceinline__ float cfn_act_rt(float z, int act) {
    if (act == CFN_ACT_RELU) return fmaxf(z, 0.0f);
    if (act == CFN_ACT_SWISH) return z * cfn_sigmoid(z);
    if (act == CFN_ACT_SIGMOID) return cfn_sigmoid(z);
    return z;
}
__device__ __forceinline__ float cfn_act_grad_rt(float z, int act) {
    if (act == CFN_ACT_RELU) return z > 0.0f ? 1.0f : 0.0f;
    if (act == CFN_ACT_SWISH) { float s = cfn_sigmoid(z); return s * (1.0f + z * (1.0f - s)); }
    if (act == CFN_ACT_SIGMOID) { float s = cfn_sigmoid(z); return s * (1.0f - s); }
    return 1.0f;
}

// wave64 all-lane sum (DPP/bpermute through __shfl_xor)
// DESIGN 4.1 / tools/pkfma_ldsret_scan.py: next to an MFMA-bound wave on its SIMD, the LOW half of a `v_pk_fma_f32` that was the first reader of a
// register pair an LDS read had just returned took the op_sel-ed HIGH register as zero in lanes 48-63 (behind `s_waitcnt lgkmcnt(0)`; the mechanism
// below the ISA is not known).  Values that come out of LDS and feed packed fp32 arithmetic inside an MFMA loop pass through ONE plain v_mov first
// (volatile: the compiler can neither fold it into the packed instruction nor drop it), so the first reader is never a packed instruction.
__device__ __forceinline__ float cfn_settle(float v) {
    float o;
    asm volatile("v_mov_b32 %0, %1" : "=v"(o) : "v"(v));
    return o;
}
__device__ __forceinline__ float2 cfn_settle(float2 v) { return float2{cfn_settle(v.x), cfn_settle(v.y)}; }
__device__ __forceinline__ float4 cfn_settle3(float4 v) { return float4{cfn_settle(v.x), cfn_settle(v.y), cfn_settle(v.z), 0.0f}; }      // (.w unused by the callers)

__device__ __forceinline__ float cfn_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double cfn_wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Bijective XCD-aware remap: hardware places block b on XCD (b % 8); give each XCD one contiguous
// run of logical ids so neighbouring logical blocks (which share halo frames / operand panels)
// hit the same private L2.  Placement only affects speed, never results.
__device__ __forceinline__ unsigned cfn_xcd_remap(unsigned b, unsigned total) {
    const unsigned q = total >> 3, r = total & 7u;
    const unsigned xcd = b & 7u, i = b >> 3;
#ifdef CFN_XCD_REVERSE      // experiment (tools/mall_probe.py): every XCD walks its eighth from the END (what the producer wrote last is read first)
    return total - 1u - (xcd * q + (xcd < r ? xcd : r) + i);
#else
    return xcd * q + (xcd < r ? xcd : r) + i;
#endif
}
