// Device-side pieces shared by the split-precision kernels (xsplit.hip) and the producers of their operands (elementwise.hip):
// the float32 -> (hi, lo) bf16 split and the "split rows" storage format.
//
// SPLIT ROWS ("planes"): a float32 row of C values (C a multiple of 4) stored in the SAME 4 C bytes as, per group of four consecutive values,
//     [hi0 hi1 hi2 hi3 | lo0 lo1 lo2 lo3]   (8 bf16 = 16 bytes),   hi = bf16(x),  lo = bf16(x - hi),
// i.e. exactly the two 8-byte pieces split4() produces.  A consumer that used to load a float4 and split it loads the same 16 bytes at the same
// address and stores the halves to its hi / lo LDS images verbatim: the conversion is paid ONCE by the producer instead of once per consumer
// tile (the 1x1 weight gradient re-stages its activation tile for every 128-channel tile of the gradient: 21 times at dense_e2).
#pragma once
#include <hip/hip_runtime.h>
namespace rdm {
typedef unsigned int xs_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int xs_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 xs_bf16x2 __attribute__((ext_vector_type(2)));

// x -> (hi, lo) for four values: two packed bf16 pairs each (v_cvt_pk_bf16_f32: round to nearest even)
__device__ __forceinline__ void split4(const float v0, const float v1, const float v2, const float v3, xs_u32x2& hi, xs_u32x2& lo) {
  const xs_bf16x2 h01 = {(__bf16)v0, (__bf16)v1}, h23 = {(__bf16)v2, (__bf16)v3};
  const float r0 = v0 - (float)h01[0], r1 = v1 - (float)h01[1], r2 = v2 - (float)h23[0], r3 = v3 - (float)h23[1];
  const xs_bf16x2 l01 = {(__bf16)r0, (__bf16)r1}, l23 = {(__bf16)r2, (__bf16)r3};
  hi = xs_u32x2{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
  lo = xs_u32x2{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
}
// four values -> one 16-byte group of a split row
__device__ __forceinline__ xs_u32x4 split_row4(const float v0, const float v1, const float v2, const float v3) {
  xs_u32x2 hi, lo;
  split4(v0, v1, v2, v3, hi, lo);
  return xs_u32x4{hi[0], hi[1], lo[0], lo[1]};
}
}  // namespace rdm
