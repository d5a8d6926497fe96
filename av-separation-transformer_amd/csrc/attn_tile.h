// attn_tile.h -- the short-sequence attention tile shared by attention.hip (attention_short_kernel, attn_proj_kernel) and the
// dependency-driven step kernel (chain.hip): one 16-query tile of one (clip, head) against <= 64 keys held in registers.
// Anonymous namespace: one copy per translation unit.
#pragma once
#include "kernels.h"

namespace {

// exp(x) for x <= 0 with a compensated argument: x*log2(e) is split into its rounded product and the exact
// rounding error (fma) plus the low part of log2(e), so the relative error stays ~1 ulp even for |x| ~ 100
// (plain exp2(x*log2e) loses |x|*6e-8).  One v_exp_f32 + 5 VALU.
// Reductions over the four lanes {c, c+16, c+32, c+48} that hold one query's partial results (the four 16-lane rows of
// a wave).  The shuffle form (xor 16, xor 32) compiles to ds_bpermute_b32 -- a round trip through the LDS crossbar (its
// latency sits in the middle of the QK^T -> softmax -> PV chain of every key tile, and it shares lgkmcnt with the K / V
// fragment reads).  gfx950's v_permlane16_swap / v_permlane32_swap exchange rows inside the VALU: swapping a value with
// itself leaves (row 0|0|2|2, row 1|1|3|3) resp. (low half twice, high half twice), and combining the pair gives every
// lane the reduction -- the same values as the shuffle form (max and the two-term sums are commutative), bit for bit.
__device__ __forceinline__ float rows_max(float v) {
  unsigned u = __float_as_uint(v);
  auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  u = __float_as_uint(v);
  auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float rows_sum(float v) {
  unsigned u = __float_as_uint(v);
  auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  u = __float_as_uint(v);
  auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

__device__ __forceinline__ float exp_neg(float x) {
  const float L2E_HI = 1.44269502162933349609f, L2E_LO = 1.92596299112661746e-08f;
  x = fmaxf(x, -120.0f);          // exp2(-173) is exactly 0 on v_exp_f32; keeps -inf (masked keys, first tile) finite
  const float t = x * L2E_HI;
  float r = fmaf(x, L2E_HI, -t);
  r = fmaf(x, L2E_LO, r);
  const float e = __builtin_amdgcn_exp2f(t);
  return fmaf(e, r * 0.69314718055994530942f, e);
}

// One 16-query tile of one (clip, head) against all (<= 16*NKT) keys: everything the short-sequence kernels share.  On return
// lane (c, g) holds the UNNORMALISED O[query c][4*(4g+r) + blk] in acc[blk][r] and its partial softmax denominator in lrun.
// COH (chain.hip): q / k / v were written by other workgroups of this launch -- every load is an L1-bypassing sc1 load
// (gemm_tile.h, top); same values, same arithmetic.
template <int NB, int NKT, bool COH = false>
__device__ __forceinline__ void attn_short_tile(const float* __restrict__ qb, const float* __restrict__ kb,
                                                const float* __restrict__ vb, int ldq, int ldk, int ldv, int Lq, int Lk,
                                                int qt, int c, int g, f32x4 (&acc)[NB], float& lrun) {
  auto ld16 = [](const float* base, int float_off) -> f32x4 {
    if (COH) return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                 __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000), float_off * 4, 0, 16));
    return *reinterpret_cast<const f32x4*>(base + float_off);
  };
  const int qrow = min(qt * 16 + c, Lq - 1);
  f32x4 qf[NB], kf[NKT][NB], vf[NKT][4];
#pragma unroll
  for (int s = 0; s < NB; ++s) qf[s] = ld16(qb, qrow * ldq + 16 * s + 4 * g);
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    const int krow = min(kt * 16 + c, Lk - 1);
#pragma unroll
    for (int s = 0; s < NB; ++s) kf[kt][s] = ld16(kb, krow * ldk + 16 * s + 4 * g);
  }
  static_assert(NB == 4, "V fragment = one float4 per key row");
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int vrow = min(kt * 16 + 4 * g + r, Lk - 1);
      vf[kt][r] = ld16(vb, vrow * ldv + NB * c);
    }
  // (hipcc sinks most of these 36 loads between the MFMAs that use them -- 55 VGPRs, a dozen counted waits.  Pinning them in
  // front with a scheduling barrier, "one round trip with every load in flight", was measured: 7.3 us instead of 6.7 for the
  // 32-clip launch (profiles/r03_ab_attention_proj.txt).  The compiler's order stays.)

  // All scores at once: NKT independent accumulators, MFMAs interleaved across tiles so the 40-cycle
  // dependent-accumulator latency never stalls the pipe; then ONE exact softmax over the <= 64 keys held in
  // registers (row max, exp, row sum -- the order nn.MultiheadAttention itself uses), then P V.
  f32x4 st[NKT];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < NB; ++s)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
        st[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][s][j], qf[s][j], st[kt], 0, 0, 0);
  float mrow = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = kt * 16 + 4 * g + r;
      st[kt][r] = key < Lk ? st[kt][r] : -INFINITY;
      mrow = fmaxf(mrow, st[kt][r]);
    }
  mrow = rows_max(mrow);
  lrun = 0.0f;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      st[kt][r] = exp_neg(st[kt][r] - mrow);
      lrun += st[kt][r];
    }
#pragma unroll
  for (int i = 0; i < NB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int blk = 0; blk < NB; ++blk)
        acc[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[kt][r][blk], st[kt][r], acc[blk], 0, 0, 0);
}

}  // namespace
