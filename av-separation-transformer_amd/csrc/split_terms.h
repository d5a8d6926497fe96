// split_terms.h -- the operand split of the split-precision kernels (gemm_split.hip, attention_split.hip): an fp32 value x is cut
// into THREE bf16 terms by truncation, x = hi + mid + lo exactly (8 + 8 + 8 significand bits): hi = the upper 16 bits of x,
// mid = the upper 16 bits of (x - hi), lo = the upper 16 bits of (x - hi - mid); the two subtractions are exact.
// Anonymous namespace: one copy per translation unit.
#pragma once
#include "kernels.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// Two values at a time: the subtractions are v_pk_add_f32 (one issue for both), 4.5 VALU per element.  Each plane's two bf16
// land in one dword, the first value in the low half (the k order of an MFMA operand).
// (The residual as ONE v_dot2c_f32_bf16 per value from the packed plane dword -- d = x, d += hi * -1, exact -- was tried: the
// instruction accumulates in place, hipcc adds a v_mov per use, and the GEMM's chunk has 117 VALU instead of 115.)
// (Bit casts of WHOLE vectors only: hipcc 7.2 reads element 0 for every e when __builtin_bit_cast is applied to an ext-vector
// element expression v[e] -- found by the one-hot probes of tools/gemm_split_debug.py.)
__device__ __forceinline__ void split_pair(const f32x2 x, unsigned& hi, unsigned& mid, unsigned& lo) {
  const u32x2 u = __builtin_bit_cast(u32x2, x);                       // upper 16 bits are taken by the pack below
  const f32x2 r1 = x - __builtin_bit_cast(f32x2, u & 0xFFFF0000u);    // exact
  const u32x2 m = __builtin_bit_cast(u32x2, r1);
  const f32x2 r2 = r1 - __builtin_bit_cast(f32x2, m & 0xFFFF0000u);   // exact
  const u32x2 l = __builtin_bit_cast(u32x2, r2);
  // bytes [3,2] of the second value | bytes [3,2] of the first
  hi = __builtin_amdgcn_perm(u[1], u[0], 0x07060302u);
  mid = __builtin_amdgcn_perm(m[1], m[0], 0x07060302u);
  lo = __builtin_amdgcn_perm(l[1], l[0], 0x07060302u);
}

}  // namespace

// ---- two fp16 terms ("H2", round 5) ---------------------------------------------------------------------------------------------
// x' = x * 2^e (e: a power-of-two scale that brings the tensor into fp16's range, exact), hi = rn_f16(x'), lo = rn_f16(x' - hi):
// x' = hi + lo + r with |r| <= 2^-22 |x'| while lo is a normal fp16 number (|x'| >= 2^-3) and |r| <= 2^-25 absolutely below
// that.  Three fp16 products (hi, lo) (lo, hi) (hi, hi), each exact in fp32, stand for one fp32 product: HALF the matrix work of
// the three-term bf16 split, at an error below the fp32 accumulation's own (gemm_h2.hip).  The subtraction is exact.
namespace {
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair_h2(const f32x2 x, unsigned& hi, unsigned& lo) {
  const f16x2 h = __builtin_convertvector(x, f16x2);                   // round to nearest even
  const f32x2 r = x - __builtin_convertvector(h, f32x2);               // exact
  const f16x2 l = __builtin_convertvector(r, f16x2);
  hi = __builtin_bit_cast(unsigned, h);                                // the first value in the low half (the k order of an MFMA operand)
  lo = __builtin_bit_cast(unsigned, l);
}
}  // namespace
