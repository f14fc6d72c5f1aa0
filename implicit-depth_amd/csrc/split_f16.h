// Device helpers of the "f16x3" split-precision MLP kernels (csrc/feature_volume.hip, csrc/mlp.hip):
// an fp32 operand, scaled by an exact power of two into f16 range, is expanded into two
// round-to-nearest f16 pieces (x/s = x0 + x1, |err| <= 2^-23 |x/s|); a product needs the three
// MFMAs x0w0 + x0w1 + x1w0 on v_mfma_f32_16x16x32_f16 (fp32 accumulate).  See csrc/conv_split.hip.
#pragma once
#include "idh_common.h"

namespace idh_f16 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int kMinExp = -100;  // lower clamp of the scaling exponents (all-zero vectors)
__device__ __forceinline__ float exp2_int(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }
__device__ __forceinline__ int exponent_of(unsigned bits) {
    const int e = (int)((bits >> 23) & 0xFF) - 127;
    return e < kMinExp ? kMinExp : e;
}
__device__ __forceinline__ unsigned pack_f16(_Float16 lo, _Float16 hi) {
    f16x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, v);
}
// two 16-blocks (4 + 4 values of this lane) -> hi / lo f16 operand of one 32-wide K block
__device__ __forceinline__ void split_block(const f32x4 &x0, const f32x4 &x1, float mul, u32x4 &hi, u32x4 &lo) {
    _Float16 h[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = (e < 4 ? x0[e] : x1[e - 4]) * mul;
        h[e] = (_Float16)v;
        l[e] = (_Float16)(v - (float)h[e]);
    }
    hi = (u32x4){pack_f16(h[0], h[1]), pack_f16(h[2], h[3]), pack_f16(h[4], h[5]), pack_f16(h[6], h[7])};
    lo = (u32x4){pack_f16(l[0], l[1]), pack_f16(l[2], l[3]), pack_f16(l[4], l[5]), pack_f16(l[6], l[7])};
}
// exponent of the voxel's max |x| over the values of its 4 lanes (lanes ln, ln+16, ln+32, ln+48)
template <int NV>
__device__ __forceinline__ int column_exponent(const f32x4 (&x)[NV]) {
    float m = 0.f;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            m = fmaxf(m, fabsf(x[j][e]));
            bad |= (__float_as_uint(x[j][e]) & 0x7F800000u) == 0x7F800000u;
        }
    if (bad) m = __uint_as_float(0x7F800000u);
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    return exponent_of(__float_as_uint(m));
}
__device__ __forceinline__ f32x4 mfma_f16(const u32x4 &A, const u32x4 &B, const f32x4 &C) {
#ifdef IDH_ABL_NOMFMA
    return C + __builtin_bit_cast(f32x4, A) * __builtin_bit_cast(f32x4, B)[0];
#endif
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B), C, 0, 0, 0);
}


}  // namespace idh_f16
