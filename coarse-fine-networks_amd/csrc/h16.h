// 16-bit storage element kinds of the reduced-precision activation paths: bf16 (BASELINE configs[1]) and IEEE half (configs[4]: "fp16 MFMA
// pointwise").  Tensors in HBM hold 2-byte elements; every kernel converts to fp32 on load, does its arithmetic in fp32 (the pointwise
// products on v_mfma_f32_32x32x16_{bf16,f16} with fp32 accumulation) and rounds to nearest even on store.  A source that serves one kind per
// compilation is built twice -- as is (bf16; entry points *_bf16) and through its *_f16.hip wrapper with CFN_F16 defined (entry points *_f16) --
// and uses the h16_* helpers / H16N() below; dwt5.hip serves both kinds in one compilation through the <KIND> templates.
#pragma once
#include "cfn_common.h"

enum { H16_BF16 = 1, H16_F16 = 2 };
typedef float __attribute__((ext_vector_type(2))) h16_f2;
typedef float __attribute__((ext_vector_type(4))) h16_f4;
template <int KIND> struct h16_types;
template <> struct h16_types<H16_BF16> {
    typedef __bf16 __attribute__((ext_vector_type(2))) v2; typedef __bf16 __attribute__((ext_vector_type(4))) v4; typedef __bf16 __attribute__((ext_vector_type(8))) v8;
};
template <> struct h16_types<H16_F16> {
    typedef _Float16 __attribute__((ext_vector_type(2))) v2; typedef _Float16 __attribute__((ext_vector_type(4))) v4; typedef _Float16 __attribute__((ext_vector_type(8))) v8;
};
template <int KIND> __device__ __forceinline__ float h16k_lo(unsigned u) {            // element in the low half of a dword
    if (KIND == H16_BF16) return __builtin_bit_cast(float, u << 16);
    return (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu));
}
template <int KIND> __device__ __forceinline__ float h16k_hi(unsigned u) {
    if (KIND == H16_BF16) return __builtin_bit_cast(float, u & 0xffff0000u);
    return (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16));
}
template <int KIND> __device__ __forceinline__ unsigned h16k_pk(float a, float b) {   // (a -> low half, b -> high half), round to nearest even
    return __builtin_bit_cast(unsigned, __builtin_convertvector((h16_f2){a, b}, typename h16_types<KIND>::v2));
}

#ifdef CFN_F16
#define H16_KIND H16_F16
#define H16N(name) name##_f16
#define H16_MFMA32 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define H16_NAME "fp16"
#else
#define H16_KIND H16_BF16
#define H16N(name) name##_bf16
#define H16_MFMA32 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define H16_NAME "bf16"
#endif
typedef h16_types<H16_KIND>::v8 h16x8;
__device__ __forceinline__ float h16_lo(unsigned u) { return h16k_lo<H16_KIND>(u); }
__device__ __forceinline__ float h16_hi(unsigned u) { return h16k_hi<H16_KIND>(u); }
__device__ __forceinline__ unsigned h16_pk(float a, float b) { return h16k_pk<H16_KIND>(a, b); }
