// Exact three-limb bf16 split of fp32 values and the bf16 MFMA wrapper shared by the split-precision kernels
// (aon_mlp_bf16.hip, wgrad_bf16x3_kernel in aon_wgrad.h).  x = hi + mid + lo exactly (24 mantissa bits in 3 x 8);
// a product x*y is formed from the six limb products of weight >= 2^-24 relative, accumulated in fp32.
#pragma once
#include "aon_common.h"

namespace aon {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo_elem, float hi_elem) {
  unsigned p;  // round-to-nearest-even pair conversion: low half <- lo_elem, high half <- hi_elem
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p) : "v"(lo_elem), "v"(hi_elem));
  return p;
}
__device__ __forceinline__ float bf16_lo_as_f32(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf16_hi_as_f32(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

__device__ __forceinline__ f32x16 mfma_bf16(const u32x4s a, const u32x4s b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

struct Limb8 {  // eight values as three packed-bf16 limb vectors (one MFMA operand each)
  u32x4s hi, mid, lo;
};

__device__ __forceinline__ void split_pair_into(float x0, float x1, int jp, Limb8& f) {
  const unsigned ph = cvt_pk_bf16(x0, x1);
  const float r0 = x0 - bf16_lo_as_f32(ph), r1 = x1 - bf16_hi_as_f32(ph);
  const unsigned pm = cvt_pk_bf16(r0, r1);
  const float q0 = r0 - bf16_lo_as_f32(pm), q1 = r1 - bf16_hi_as_f32(pm);
  f.hi[jp] = ph; f.mid[jp] = pm; f.lo[jp] = cvt_pk_bf16(q0, q1);
}

__device__ __forceinline__ Limb8 split8(const f32x4& a, const f32x4& b) {
  Limb8 f;
  split_pair_into(a[0], a[1], 0, f); split_pair_into(a[2], a[3], 1, f);
  split_pair_into(b[0], b[1], 2, f); split_pair_into(b[2], b[3], 3, f);
  return f;
}

// acc += A . B over 16 k-values, A and B as limbs: the six products of weight >= 2^-24, smallest terms first
__device__ __forceinline__ f32x16 mfma_bf16x3(const Limb8& a, const Limb8& b, f32x16 acc) {
  acc = mfma_bf16(a.lo, b.hi, acc);
  acc = mfma_bf16(a.hi, b.lo, acc);
  acc = mfma_bf16(a.mid, b.mid, acc);
  acc = mfma_bf16(a.mid, b.hi, acc);
  acc = mfma_bf16(a.hi, b.mid, acc);
  acc = mfma_bf16(a.hi, b.hi, acc);
  return acc;
}

}  // namespace aon
