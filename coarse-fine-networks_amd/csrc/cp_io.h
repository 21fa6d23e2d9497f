// Element access of the column-pair wave kernels (dwcp.hip, dwcpb.hip, dwcpb2.hip).  The sources are compiled twice: as they
// are (fp32 tensors) and through dwcp*_bf16.hip with DW_BF16 defined (bf16 storage, identical fp32 arithmetic, host functions
// and the C entry point suffixed _bf16, argument structs renamed so that the kernel symbols differ) -- the scheme of
// dwconv3d.hip / dwconv3d_bf16.hip.  LDS images, accumulators, statistics and every reduction stay fp32 / fp64.
#pragma once
#include "cfn_common.h"
#include "h16.h"

typedef float __attribute__((ext_vector_type(4))) cp_f4;
typedef float __attribute__((ext_vector_type(2))) cp_f2;
typedef unsigned __attribute__((ext_vector_type(2))) cp_u2;
typedef unsigned __attribute__((ext_vector_type(4))) cp_u4;

#ifdef DW_BF16
typedef unsigned short cpe_t;
#define CP_ES 2
#define CPN(name) H16N(name)
__device__ __forceinline__ float cp_lo(unsigned u) { return h16_lo(u); }
__device__ __forceinline__ float cp_hi(unsigned u) { return h16_hi(u); }
__device__ __forceinline__ unsigned cp_pk(float a, float b) { return h16_pk(a, b); }
// 4 consecutive elements (8 bytes; the offset is 8-byte aligned)
__device__ __forceinline__ cp_f4 cp_ld4(__amdgpu_buffer_rsrc_t r, int vo, int so) {
    const cp_u2 u = __builtin_bit_cast(cp_u2, __builtin_amdgcn_raw_buffer_load_b64(r, vo, so, 0));
    return (cp_f4){cp_lo(u.x), cp_hi(u.x), cp_lo(u.y), cp_hi(u.y)};
}
__device__ __forceinline__ float cp_ld1(__amdgpu_buffer_rsrc_t r, int vo, int so) { return cp_lo((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, vo, so, 0)); }
__device__ __forceinline__ cp_f2 cp_ld2(__amdgpu_buffer_rsrc_t r, int vo, int so) { const unsigned u = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, vo, so, 0); return (cp_f2){cp_lo(u), cp_hi(u)}; }
__device__ __forceinline__ void cp_st1(float v, __amdgpu_buffer_rsrc_t r, int vo, int so) { __builtin_amdgcn_raw_buffer_store_b16((short)(cp_pk(v, 0.0f) & 0xffffu), r, vo, so, 0); }
__device__ __forceinline__ void cp_st2(cp_f2 v, __amdgpu_buffer_rsrc_t r, int vo, int so) { __builtin_amdgcn_raw_buffer_store_b32(cp_pk(v.x, v.y), r, vo, so, 0); }
__device__ __forceinline__ void cp_st4(cp_f4 v, __amdgpu_buffer_rsrc_t r, int vo, int so) {
    __builtin_amdgcn_raw_buffer_store_b64((cp_u2){cp_pk(v.x, v.y), cp_pk(v.z, v.w)}, r, vo, so, 0);
}
// value as the consumer will read it back (forward statistics are taken over the stored values)
__device__ __forceinline__ cp_f2 cp_rt2(cp_f2 v) { const unsigned u = cp_pk(v.x, v.y); return (cp_f2){cp_lo(u), cp_hi(u)}; }
__device__ __forceinline__ float cp_rt1(float v) { return cp_lo(cp_pk(v, 0.0f)); }
#else
typedef float cpe_t;
#define CP_ES 4
#define CPN(name) name
__device__ __forceinline__ cp_f4 cp_ld4(__amdgpu_buffer_rsrc_t r, int vo, int so) { return __builtin_bit_cast(cp_f4, __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0)); }
__device__ __forceinline__ float cp_ld1(__amdgpu_buffer_rsrc_t r, int vo, int so) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, vo, so, 0)); }
__device__ __forceinline__ cp_f2 cp_ld2(__amdgpu_buffer_rsrc_t r, int vo, int so) { return __builtin_bit_cast(cp_f2, __builtin_amdgcn_raw_buffer_load_b64(r, vo, so, 0)); }
__device__ __forceinline__ void cp_st1(float v, __amdgpu_buffer_rsrc_t r, int vo, int so) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, vo, so, 0); }
__device__ __forceinline__ void cp_st2(cp_f2 v, __amdgpu_buffer_rsrc_t r, int vo, int so) { __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(cp_u2, v), r, vo, so, 0); }
__device__ __forceinline__ void cp_st4(cp_f4 v, __amdgpu_buffer_rsrc_t r, int vo, int so) { cfn_bst128(__builtin_bit_cast(cp_u4, v), r, vo, so); }
__device__ __forceinline__ cp_f2 cp_rt2(cp_f2 v) { return v; }
__device__ __forceinline__ float cp_rt1(float v) { return v; }
#endif
