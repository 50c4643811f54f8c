// Split-bf16 arithmetic shared by the register-resident (ncde_fast.hip) and batch-tiled (ncde_tiled.hip) families.
#pragma once
#include "ncde_common.h"

namespace {

// gfx950 runs fp32-input MFMA at the fp32 VECTOR rate (1/16 of bf16).  Every fp32 value is split EXACTLY
// into three bf16 pieces x = hi + mid + lo (8+8+8 mantissa bits) and a product a*b is evaluated as the six
// partial products whose weight is >= 2^-16 (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi); each bf16 x bf16
// product is exact in the fp32 accumulator and the dropped terms are <= 3 * 2^-24 relative -- fp32 round-off
// class.  6 x v_mfma_f32_16x16x32_bf16 replace 8 x v_mfma_f32_16x16x4_f32 at ~1/2 the cycles each.
// Lane (s, g) supplies k = 32c + 8g + j of a K chunk c, so with output rows permuted as
// tile t, D-row (g, r) <-> unit 32*(t>>1) + 8g + 4*(t&1) + r the D registers of a layer are again exactly the
// (to-be-split) B operand of the next layer.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Split3 {
    u32x4 hi, mid, lo;  // 8 bf16 each: element j in dword j>>1, even j in the low half
};

__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    const unsigned b0 = __float_as_uint(x0), b1 = __float_as_uint(x1);
    const float r0 = x0 - __uint_as_float(b0 & 0xFFFF0000u), r1 = x1 - __uint_as_float(b1 & 0xFFFF0000u);  // exact
    const unsigned c0 = __float_as_uint(r0), c1 = __float_as_uint(r1);
    const float l0 = r0 - __uint_as_float(c0 & 0xFFFF0000u), l1 = r1 - __uint_as_float(c1 & 0xFFFF0000u);  // exact
    hi = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
    mid = __builtin_amdgcn_perm(c1, c0, 0x07060302u);
    lo = __builtin_amdgcn_perm(__float_as_uint(l1), __float_as_uint(l0), 0x07060302u);
}

__device__ __forceinline__ Split3 split8(const float* v) {
    Split3 o;
    // piece by piece (all hi, then all mid, then all lo): the consumers' first MFMAs need only hi
    float r[8], l[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) o.hi[q] = __builtin_amdgcn_perm(__float_as_uint(v[2 * q + 1]), __float_as_uint(v[2 * q]), 0x07060302u);
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = v[j] - __uint_as_float(__float_as_uint(v[j]) & 0xFFFF0000u);  // exact
#pragma unroll
    for (int q = 0; q < 4; ++q) o.mid[q] = __builtin_amdgcn_perm(__float_as_uint(r[2 * q + 1]), __float_as_uint(r[2 * q]), 0x07060302u);
#pragma unroll
    for (int j = 0; j < 8; ++j) l[j] = r[j] - __uint_as_float(__float_as_uint(r[j]) & 0xFFFF0000u);  // exact
#pragma unroll
    for (int q = 0; q < 4; ++q) o.lo[q] = __builtin_amdgcn_perm(__float_as_uint(l[2 * q + 1]), __float_as_uint(l[2 * q]), 0x07060302u);
    return o;
}

__device__ __forceinline__ f32x4 mfma_bf(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16 mfma_bf32(u32x4 a, u32x4 b, f32x16 c) {  // 32x32x16: lane (i=lane&31, kg=lane>>5) holds k = 8kg..8kg+7
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// c += A * B over one K chunk of 32, fp32-equivalent.  Product order: everything that needs only the hi piece of the
// (freshly split) B operand first, then mid, then lo -- on the serial chain the B operand comes straight out of the
// VALU split of the previous layer, so the first three MFMAs can issue while the mid / lo pieces are still being formed.
__device__ __forceinline__ f32x4 mfma_split(const Split3& A, const Split3& B, f32x4 c) {
    c = mfma_bf(A.lo, B.hi, c);
    c = mfma_bf(A.mid, B.hi, c);
    c = mfma_bf(A.hi, B.hi, c);
    c = mfma_bf(A.mid, B.mid, c);
    c = mfma_bf(A.hi, B.mid, c);
    c = mfma_bf(A.hi, B.lo, c);
    return c;
}

}  // namespace
