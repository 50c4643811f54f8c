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

// ---- 2-way split-fp16 ("h2") -------------------------------------------------------------------------------------------------
// x = h1 + 2^-11 h2 with h1 = rne_f16(x) (11 significant bits) and h2 = rne_f16((x - h1) * 2^11) (the next 11, signed): the
// representation error is <= 2^-24 |x| -- one fp32 half-ulp -- as long as h1 is a NORMAL fp16 number, i.e. 6.1e-5 <= |x| < 65520;
// below that range the error is absolute (<= 2^-36), above it the conversion overflows.  A product a*b is the three partial products
// of weight >= 2^-11 (h1*h1; h1*h2 and h2*h1, which share one accumulator scaled by 2^11); the dropped h2*h2 term is <= 2^-24 |ab|,
// of either sign.  3 MFMAs instead of the 3-way split-bf16's 6, ~24 VALU per 8 values instead of 44 (v_cvt_pk_f16_f32, v_pk_mul_f32,
// v_cvt_f32_f16, v_pk_fma_f32), 8 registers per operand fragment instead of 12.
// The fp16 range is the price: kernels that use this split track the largest magnitude they have split (`mx`) and report a RANGE
// FAULT for their sample tile when it exceeds NCDE_H2_LIMIT (or is infinite); the host re-executes faulted tiles with the split-bf16
// kernel (same arithmetic contract, any magnitude).  Cotangent-side operands, whose scale is arbitrary, are brought into range
// by a per-workgroup power-of-two factor first (ncde_adj_fast3).
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define NCDE_H2_SCALE 2048.0f
#define NCDE_H2_INV 4.8828125e-4f

struct Split2h {
    u32x4 hi, lo;  // 8 fp16 each: element j in dword j>>1, even j in the low half; lo holds (x - hi) * 2^11
};

// Range tracking: mx <- max(mx, |v[0..7]|), one v_max3_f32 per pair.  (fmaxf() would canonicalise every input first -- three
// instructions per pair -- hence the inline asm.  Its inputs are never raw MFMA results: a bias/ReLU/mask or an LDS round trip sits in
// between, which matters because the hazard recognizer does not see MFMA -> inline-asm dependences.)  An infinite operand is caught by
// the limit; a NaN operand is not, it propagates through the GEMM instead -- as it does in the fp32 reference.
#define NCDE_H2_LIMIT 60000.0f
__device__ __forceinline__ Split2h split8h(const float* v, float& mx) {
    Split2h o;
    h16x2 h[4];  // (a __builtin_bit_cast applied directly to an element of an ext-vector reads element 0: keep the pairs as scalars)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x2 x = {v[2 * q], v[2 * q + 1]};
        h[q] = __builtin_convertvector(x, h16x2);
        asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(mx) : "v"(v[2 * q]), "v"(v[2 * q + 1]));
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x2 r;
        r[0] = __builtin_fmaf((float)h[q][0], -NCDE_H2_SCALE, v[2 * q] * NCDE_H2_SCALE);  // exact: (x - h1) * 2^11
        r[1] = __builtin_fmaf((float)h[q][1], -NCDE_H2_SCALE, v[2 * q + 1] * NCDE_H2_SCALE);
        const h16x2 l = __builtin_convertvector(r, h16x2);
        const unsigned hq = __builtin_bit_cast(unsigned, h[q]);
        const unsigned lq = __builtin_bit_cast(unsigned, l);
        o.hi[q] = hq;
        o.lo[q] = lq;
    }
    return o;
}
__device__ __forceinline__ bool h2_range_fault(float mx) { return !(mx <= NCDE_H2_LIMIT); }
// two values -> one dword of each piece (even value in the low half), as split8h does for eight
__device__ __forceinline__ void split_pair_h(float x0, float x1, unsigned& hi, unsigned& lo, float& mx) {
    const f32x2 x = {x0, x1};
    const h16x2 h = __builtin_convertvector(x, h16x2);
    asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(mx) : "v"(x0), "v"(x1));
    f32x2 r;
    r[0] = __builtin_fmaf((float)h[0], -NCDE_H2_SCALE, x0 * NCDE_H2_SCALE);  // exact: (x - h1) * 2^11
    r[1] = __builtin_fmaf((float)h[1], -NCDE_H2_SCALE, x1 * NCDE_H2_SCALE);
    const h16x2 l = __builtin_convertvector(r, h16x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}

__device__ __forceinline__ f32x4 mfma_h(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
}

// main += A.hi B.hi;  cross += A.hi B.lo + A.lo B.hi  (cross carries the factor 2^11: result = main + 2^-11 cross).
// The cross products first: they need only... the hi piece of the weights and BOTH pieces of the fresh B operand are formed before
// the first MFMA anyway (24 VALU), and alternating accumulators keeps dependent MFMAs apart.
__device__ __forceinline__ void mfma_split2(const Split2h& A, const Split2h& B, f32x4& main, f32x4& cross) {
    main = mfma_h(A.hi, B.hi, main);
    cross = mfma_h(A.lo, B.hi, cross);
    cross = mfma_h(A.hi, B.lo, cross);
}
__device__ __forceinline__ f32x4 h2_combine(f32x4 main, f32x4 cross) {  // two v_pk_fma_f32
    const f32x2 k = {NCDE_H2_INV, NCDE_H2_INV};
    const f32x2 lo = __builtin_elementwise_fma((f32x2){cross[0], cross[1]}, k, (f32x2){main[0], main[1]});
    const f32x2 hi = __builtin_elementwise_fma((f32x2){cross[2], cross[3]}, k, (f32x2){main[2], main[3]});
    return (f32x4){lo[0], lo[1], hi[0], hi[1]};
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16 mfma_h32(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
}

// Power-of-two factor that brings a cotangent of largest magnitude m into the fp16 sweet spot: keep `cur` while 4 <= m * cur < 64,
// otherwise 2^(4 - floor(log2 m)).  A pure function of (m, cur): every wave of a workgroup evaluates it on the same inputs.
__device__ __forceinline__ float h2_pick_scale(float m, float cur) {
    if (!(m > 0.0f) || !(m < 3.0e38f)) return cur;  // zero / infinite / NaN cotangent: nothing to normalise
    const float v = m * cur;
    if (v >= 4.0f && v < 64.0f) return cur;
    int k = 4 - (int)((__builtin_bit_cast(unsigned, m) >> 23) & 255u) + 127;
    k = k < -96 ? -96 : (k > 96 ? 96 : k);
    return __builtin_bit_cast(float, (unsigned)(k + 127) << 23);
}
__device__ __forceinline__ float h2_inv_scale(float sig) {  // exact reciprocal of a power of two
    return __builtin_bit_cast(float, (254u - ((__builtin_bit_cast(unsigned, sig) >> 23) & 255u)) << 23);
}
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

// One interface over both splits, so a kernel is written once and instantiated per precision scheme:
//   SplitOps<0>: 3-way split-bf16 (any magnitude), SplitOps<1>: 2-way split-fp16 (|x| < 65504; weights and forward activations).
template <int HP>
struct SplitOps;
template <>
struct SplitOps<0> {
    typedef Split3 T;
    static constexpr int NP = 3;
    // LDS image of one operand fragment: [NP pieces][64 lanes][4 dwords]
    static __device__ __forceinline__ void store(unsigned* img, int lane, const T& v) {
        *reinterpret_cast<u32x4*>(img + (0 * 64 + lane) * 4) = v.hi;
        *reinterpret_cast<u32x4*>(img + (1 * 64 + lane) * 4) = v.mid;
        *reinterpret_cast<u32x4*>(img + (2 * 64 + lane) * 4) = v.lo;
    }
    static __device__ __forceinline__ T load(const unsigned* img, int lane) {
        T v;
        v.hi = *reinterpret_cast<const u32x4*>(img + (0 * 64 + lane) * 4);
        v.mid = *reinterpret_cast<const u32x4*>(img + (1 * 64 + lane) * 4);
        v.lo = *reinterpret_cast<const u32x4*>(img + (2 * 64 + lane) * 4);
        return v;
    }
    struct Acc { f32x4 m; };
    static __device__ __forceinline__ T split(const float* v, float&) { return split8(v); }
    static __device__ __forceinline__ Acc init(f32x4 bias) { return Acc{bias}; }
    static __device__ __forceinline__ void mac(const T& A, const T& B, Acc& c) { c.m = mfma_split(A, B, c.m); }
    static __device__ __forceinline__ f32x4 finish(const Acc& c) { return c.m; }
};
template <>
struct SplitOps<1> {
    typedef Split2h T;
    static constexpr int NP = 2;
    static __device__ __forceinline__ void store(unsigned* img, int lane, const T& v) {
        *reinterpret_cast<u32x4*>(img + (0 * 64 + lane) * 4) = v.hi;
        *reinterpret_cast<u32x4*>(img + (1 * 64 + lane) * 4) = v.lo;
    }
    static __device__ __forceinline__ T load(const unsigned* img, int lane) {
        T v;
        v.hi = *reinterpret_cast<const u32x4*>(img + (0 * 64 + lane) * 4);
        v.lo = *reinterpret_cast<const u32x4*>(img + (1 * 64 + lane) * 4);
        return v;
    }
    struct Acc { f32x4 m, x; };
    static __device__ __forceinline__ T split(const float* v, float& mx) { return split8h(v, mx); }
    static __device__ __forceinline__ Acc init(f32x4 bias) { return Acc{bias, (f32x4){0.0f, 0.0f, 0.0f, 0.0f}}; }
    static __device__ __forceinline__ void mac(const T& A, const T& B, Acc& c) { mfma_split2(A, B, c.m, c.x); }
    static __device__ __forceinline__ f32x4 finish(const Acc& c) { return h2_combine(c.m, c.x); }
};


}  // namespace
