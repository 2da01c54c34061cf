// Shared by the Mamba-2 SSD kernels on the matrix pipe (ssd.hip forward, ssd_bwd.hip backward): 32x32x16 MFMA wrappers and
// 16-bit pack / unpack per I/O dtype.
#pragma once
#include "dm_common.h"

namespace dm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 ssd_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 ssd_f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t ssd_u32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct ssd_ops;
template <> struct ssd_ops<bf16_t> {
    static __device__ __forceinline__ f32x16 mfma(const ssd_u32x4& a, const ssd_u32x4& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ssd_bf16x8, a), __builtin_bit_cast(ssd_bf16x8, b), c, 0, 0, 0);
    }
    // v_cvt_pk_bf16_f32 through the compiler (NOT inline asm): the kernels convert MFMA results directly, and only an instruction
    // the compiler knows gets the wait states an MFMA result needs before a VALU read (seen as 1 % garbage in K6b's selector tiles)
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ uint32_t pack(float lo, float hi) {
        return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){lo, hi}, b2));
    }
    static __device__ __forceinline__ float lo(uint32_t w) { return __uint_as_float(w << 16); }
    static __device__ __forceinline__ float hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
};
template <> struct ssd_ops<f16_t> {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ f32x16 mfma(const ssd_u32x4& a, const ssd_u32x4& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ssd_f16x8, a), __builtin_bit_cast(ssd_f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint32_t pack(float lo, float hi) {
        h2 v;
        v.x = (_Float16)lo;
        v.y = (_Float16)hi;
        return __builtin_bit_cast(uint32_t, v);
    }
    static __device__ __forceinline__ float lo(uint32_t w) { return (float)__builtin_bit_cast(h2, w).x; }
    static __device__ __forceinline__ float hi(uint32_t w) { return (float)__builtin_bit_cast(h2, w).y; }
};

}  // namespace dm
