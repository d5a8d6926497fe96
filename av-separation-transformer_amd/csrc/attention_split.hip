// attention_split.hip -- split-precision attention (round 4) for dh = 64 and long sequences (L = 251 / 501 of configs 3-5):
// fp32 q / k / v, fp32 softmax and output, both matrix products on the bf16 matrix pipe.
//
// After the Linear layers moved to the split-precision GEMM (gemm_split.hip) the fp32-MFMA attention_lds_kernel was 12 % (cfg3)
// and 23 % (cfg5) of the forward at 0.60 of the fp32 matrix peak.  The same scheme applies to S = Q K^T and O = P V: every fp32
// operand (q, k, v, and the probabilities p in [0, 1]) is cut into three bf16 terms by truncation, x = hi + mid + lo exactly,
// and a product is the six bf16 MFMA products (hi,lo) (lo,hi) (mid,mid) (mid,hi) (hi,mid) (hi,hi) accumulated in fp32 --
// 6 / 16 of the fp32 matrix time, results at the fp32 kernel's own distance from float64 (tests/test_gpu_parity.py).
//
// One workgroup = 4 waves = 4 x 32 queries of one (clip, head); keys in steps of 32:
//   * S^T = K Q^T per 16-key tile (v_mfma_f32_16x16x32_bf16, A = K rows, B = Q rows, contraction over dh in two halves): lane
//     (c, g) holds the scores of query c for keys 4g .. 4g+3 of the tile -- the fp32 kernels' layout, so the online softmax is
//     theirs.  Q's three planes live in registers for the whole kernel (48 VGPRs).
//   * O^T += V^T P^T over the step's 32 keys in ONE contraction: its index j = 8g + e stands for key 4g + e of the first tile
//     (e < 4) and key 4g + e - 4 of the second (e >= 4), which is exactly what lane (c, g) holds after the softmax: the
//     probabilities go from the score registers into the B operand without leaving the lane (split in registers, 9 VALU per
//     pair).  The matching A operand -- V^T, four consecutive keys per lane -- is read from a row-major [key][dv] image with
//     gfx950's transposing LDS read (ds_read_b64_tr_b16: each 16-lane group fetches 4 rows x 16 columns and gets them column-major).
//   * K and V are split ONCE per workgroup on their way to LDS (three [32][64] bf16 planes each, 24 KB per step, two buffers);
//     16-byte K slots XOR-swizzled with key & 7, 32-byte V chunks with (key >> 1) & 3: both fragment reads are conflict-free
//     under this chip's per-instruction lane groups (MI355X_MICROARCH.md, LDS).
#include "kernels.h"
#include "attn_tile.h"
#include "split_terms.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int AS_KEYS = 32;                  // keys per step
constexpr int AS_PL = AS_KEYS * 128;         // bytes of one plane: 32 rows x 64 bf16
constexpr int AS_BUF = 6 * AS_PL;            // [K hi|mid|lo][V hi|mid|lo] = 24 KB

// exp_neg (attn_tile.h) without its clamp at -120, scalar and on a pair with packed fp32 VALU.  The clamp exists for -inf
// arguments (masked keys, the first tile's running maximum); this kernel masks with MASKED = -1e30 instead, for which every
// intermediate stays finite and v_exp_f32 returns exactly 0, and for finite arguments the two forms have the same bits
// (below -120 both give 0: exp2(-173) is 0 on v_exp_f32) -- 18 v_max per step less in a VALU-bound loop.
constexpr float MASKED = -1e30f;
__device__ __forceinline__ float exp_neg_finite(float x) {
  const float L2E_HI = 1.44269502162933349609f, L2E_LO = 1.92596299112661746e-08f;
  const float t = x * L2E_HI;
  float r = fmaf(x, L2E_HI, -t);
  r = fmaf(x, L2E_LO, r);
  const float e = __builtin_amdgcn_exp2f(t);
  return fmaf(e, r * 0.69314718055994530942f, e);
}
__device__ __forceinline__ f32x2 exp_neg_pair(f32x2 x) {
  const f32x2 HI = {1.44269502162933349609f, 1.44269502162933349609f};
  const f32x2 LO = {1.92596299112661746e-08f, 1.92596299112661746e-08f};
  const f32x2 LN2 = {0.69314718055994530942f, 0.69314718055994530942f};
  const f32x2 t = x * HI;
  f32x2 r = __builtin_elementwise_fma(x, HI, -t);
  r = __builtin_elementwise_fma(x, LO, r);
  f32x2 e;
  e[0] = __builtin_amdgcn_exp2f(t[0]);
  e[1] = __builtin_amdgcn_exp2f(t[1]);
  return __builtin_elementwise_fma(e, r * LN2, e);
}

// 8 fp32 values (k order) -> the three bf16x8 operand planes
struct Planes {
  unsigned hi[4], mid[4], lo[4];
};
__device__ __forceinline__ Planes split8(const f32x4 a, const f32x4 b) {
  Planes w;
  split_pair(f32x2{a[0], a[1]}, w.hi[0], w.mid[0], w.lo[0]);
  split_pair(f32x2{a[2], a[3]}, w.hi[1], w.mid[1], w.lo[1]);
  split_pair(f32x2{b[0], b[1]}, w.hi[2], w.mid[2], w.lo[2]);
  split_pair(f32x2{b[2], b[3]}, w.hi[3], w.mid[3], w.lo[3]);
  return w;
}
__device__ __forceinline__ u32x4 vec4(const unsigned (&a)[4]) { return u32x4{a[0], a[1], a[2], a[3]}; }

__device__ __forceinline__ bf16x8 as_frag(const u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// (Three workgroups per CU -- 168 VGPRs, 4 spilled -- measured: no difference, profiles/r04_ab_split_attention_wgs.txt.)
template <int QT>
__global__ __launch_bounds__(256, 2) void attention_split_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                                 int ldk, const float* __restrict__ v, int ldv,
                                                                 float* __restrict__ o, int ldo, int nhead, int Lq, int Lk,
                                                                 int nqt, float qscale, unsigned short* __restrict__ op,
                                                                 long long o_rows, float h2_scale) {
  constexpr int DH = 64;
  __shared__ __attribute__((aligned(16))) char lds[2 * AS_BUF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int nqw = (nqt + QT - 1) / QT;
  const int wg_per_head = (nqw + 3) >> 2;
  const int bh = blockIdx.x / wg_per_head;
  const int qw = (blockIdx.x - bh * wg_per_head) * 4 + wave;
  const bool active = qw < nqw;                       // no wave leaves early (barriers): idle waves compute on clamped rows
  const int b = bh / nhead, h = bh - b * nhead;

  const float* qb = q + (size_t)b * Lq * ldq + h * DH;
  const float* kb = k + (size_t)b * Lk * ldk + h * DH;
  const float* vb = v + (size_t)b * Lk * ldv + h * DH;
  float* ob = o + (size_t)b * Lq * ldo + h * DH;

  // Q planes: B operand of S^T, lane (c, g) holds Q[query c][32 s + 8 g .. + 7]
  bf16x8 qh[QT][2], qm[QT][2], ql[QT][2];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int qrow = min((qw * QT + t) * 16 + c, Lq - 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const float* src = qb + (size_t)qrow * ldq + 32 * s + 8 * g;
      const f32x4 a = *reinterpret_cast<const f32x4*>(src) * qscale, bq = *reinterpret_cast<const f32x4*>(src + 4) * qscale;
      const Planes w = split8(a, bq);
      qh[t][s] = as_frag(vec4(w.hi)); qm[t][s] = as_frag(vec4(w.mid)); ql[t][s] = as_frag(vec4(w.lo));
    }
  }
  f32x4 acc[QT][4];
  float mrun[QT], lrun[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    mrun[t] = MASKED;
    lrun[t] = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // staging: thread -> (key row = tid / 8, 8-float piece = tid % 8) of K and of V
  const int srow = tid >> 3, spc = tid & 7;
  const int kst = srow * 128 + ((spc ^ (srow & 7)) << 4);                                  // K: one 16-byte slot per plane
  const int vsw = (srow >> 1) & 3;
  const int vst0 = 3 * AS_PL + srow * 128 + ((((2 * spc) >> 2) ^ vsw) << 5) + (((2 * spc) & 3) << 3);       // V: two 8-byte pieces
  const int vst1 = 3 * AS_PL + srow * 128 + ((((2 * spc + 1) >> 2) ^ vsw) << 5) + (((2 * spc + 1) & 3) << 3);
  f32x4 sk0, sk1, sv0, sv1;
  const int nstep = (Lk + AS_KEYS - 1) / AS_KEYS;
  auto load_step = [&](int st) {
    const int key = min(st * AS_KEYS + srow, Lk - 1);
    const float* ks = kb + (size_t)key * ldk + 8 * spc;
    const float* vs = vb + (size_t)key * ldv + 8 * spc;
    sk0 = *reinterpret_cast<const f32x4*>(ks); sk1 = *reinterpret_cast<const f32x4*>(ks + 4);
    sv0 = *reinterpret_cast<const f32x4*>(vs); sv1 = *reinterpret_cast<const f32x4*>(vs + 4);
  };
  auto store_step = [&](int buf) {
    char* base = lds + buf * AS_BUF;
    const Planes wk = split8(sk0, sk1);
    *reinterpret_cast<u32x4*>(base + kst) = vec4(wk.hi);
    *reinterpret_cast<u32x4*>(base + AS_PL + kst) = vec4(wk.mid);
    *reinterpret_cast<u32x4*>(base + 2 * AS_PL + kst) = vec4(wk.lo);
    const Planes wv = split8(sv0, sv1);
    *reinterpret_cast<u32x2*>(base + vst0) = u32x2{wv.hi[0], wv.hi[1]};
    *reinterpret_cast<u32x2*>(base + vst1) = u32x2{wv.hi[2], wv.hi[3]};
    *reinterpret_cast<u32x2*>(base + AS_PL + vst0) = u32x2{wv.mid[0], wv.mid[1]};
    *reinterpret_cast<u32x2*>(base + AS_PL + vst1) = u32x2{wv.mid[2], wv.mid[3]};
    *reinterpret_cast<u32x2*>(base + 2 * AS_PL + vst0) = u32x2{wv.lo[0], wv.lo[1]};
    *reinterpret_cast<u32x2*>(base + 2 * AS_PL + vst1) = u32x2{wv.lo[2], wv.lo[3]};
  };

  // fragment addresses (inside a buffer).  K: lane (c, g) reads the 8 bf16 dh = 32 s + 8 g .. of key row 16 kt + c.
  int kfo[2][2];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int row = 16 * kt + c;
      kfo[kt][s] = row * 128 + (((4 * s + g) ^ (row & 7)) << 4);
    }
  // V (transposing read): lane 4 qq + p of the 16-lane group g supplies row 16 kt + 4 g + qq, columns 16 blk + 4 p .. + 3 and
  // receives keys 16 kt + 4 g .. + 3 of column dv = 16 blk + (lane & 15)
  int vfo[2];
  {
    const int qq = c >> 2, p = c & 3;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const int row = 16 * kt + 4 * g + qq;
      vfo[kt] = 3 * AS_PL + row * 128 + (p << 3) + ((((row >> 1) & 3)) << 5);     // chunk index blk XORed in below: (blk ^ sw) << 5
    }
  }

  load_step(0);
  store_step(0);
  __syncthreads();
  for (int st = 0; st < nstep; ++st) {
    const int buf = st & 1;
    if (st + 1 < nstep) load_step(st + 1);                          // block-uniform; lands under this step's MFMAs
    const char* base = lds + buf * AS_BUF;

    // ---- S^T: two key tiles x QT query tiles, smallest products first
    f32x4 stt[2][QT];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int t = 0; t < QT; ++t) stt[kt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 kh[2], km[2], kl[2];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        kh[kt] = *reinterpret_cast<const bf16x8*>(base + kfo[kt][s]);
        km[kt] = *reinterpret_cast<const bf16x8*>(base + AS_PL + kfo[kt][s]);
        kl[kt] = *reinterpret_cast<const bf16x8*>(base + 2 * AS_PL + kfo[kt][s]);
      }
#define AS_S(KP, QP)                                                                                       \
  _Pragma("unroll") for (int kt = 0; kt < 2; ++kt) _Pragma("unroll") for (int t = 0; t < QT; ++t)         \
      stt[kt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(KP[kt], QP[t][s], stt[kt][t], 0, 0, 0);
      AS_S(kl, qh)
      AS_S(kh, ql)
      AS_S(km, qm)
      AS_S(kh, qm)
      AS_S(km, qh)
      AS_S(kh, qh)
#undef AS_S
    }

    // ---- online softmax over the step's 32 keys; the probabilities become the B operand's three planes in place
    bf16x8 ph[QT], pm[QT], pl[QT];
    const bool ragged = (st + 1) * AS_KEYS > Lk;                     // block-uniform: only the last step holds keys >= Lk
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      if (ragged) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) stt[kt][t][r] = st * AS_KEYS + 16 * kt + 4 * g + r < Lk ? stt[kt][t][r] : MASKED;
      }
      float tmax = fmaxf(fmaxf(fmaxf(stt[0][t][0], stt[0][t][1]), fmaxf(stt[0][t][2], stt[0][t][3])),
                         fmaxf(fmaxf(stt[1][t][0], stt[1][t][1]), fmaxf(stt[1][t][2], stt[1][t][3])));
      tmax = rows_max(tmax);
      const float mnew = fmaxf(mrun[t], tmax);                       // finite: step 0 always holds key 0
      const float alpha = exp_neg_finite(mrun[t] - mnew);            // 0 on the first step (mrun = MASKED)
      const f32x2 m2 = {mnew, mnew};
      const f32x2 p01 = exp_neg_pair(f32x2{stt[0][t][0], stt[0][t][1]} - m2), p23 = exp_neg_pair(f32x2{stt[0][t][2], stt[0][t][3]} - m2);
      const f32x2 p45 = exp_neg_pair(f32x2{stt[1][t][0], stt[1][t][1]} - m2), p67 = exp_neg_pair(f32x2{stt[1][t][2], stt[1][t][3]} - m2);
      const float psum = (((p01[0] + p01[1]) + (p23[0] + p23[1])) + ((p45[0] + p45[1]) + (p67[0] + p67[1])));
      lrun[t] = lrun[t] * alpha + psum;                              // per-lane partial (own 8 keys per step); reduced at the end
      mrun[t] = mnew;
      Planes w;
      split_pair(p01, w.hi[0], w.mid[0], w.lo[0]);
      split_pair(p23, w.hi[1], w.mid[1], w.lo[1]);
      split_pair(p45, w.hi[2], w.mid[2], w.lo[2]);
      split_pair(p67, w.hi[3], w.mid[3], w.lo[3]);
      ph[t] = as_frag(vec4(w.hi)); pm[t] = as_frag(vec4(w.mid)); pl[t] = as_frag(vec4(w.lo));
      const f32x2 a2 = {alpha, alpha};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x2 lo2 = f32x2{acc[t][i][0], acc[t][i][1]} * a2, hi2 = f32x2{acc[t][i][2], acc[t][i][3]} * a2;
        acc[t][i] = f32x4{lo2[0], lo2[1], hi2[0], hi2[1]};
      }
    }

    // ---- O^T += V^T P^T, one 16-column block of V at a time
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      bf16x8 vh, vm, vl;
      {
        typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
        u32x4 w[3];
#pragma unroll
        for (int pln = 0; pln < 3; ++pln) {
          // (blk ^ sw) << 5 == (blk << 5) ^ (sw << 5): the swizzle term is already in vfo
          const s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base + pln * AS_PL + (vfo[0] ^ (blk << 5))));
          const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base + pln * AS_PL + (vfo[1] ^ (blk << 5))));
          const u32x2 d0 = __builtin_bit_cast(u32x2, r0), d1 = __builtin_bit_cast(u32x2, r1);
          w[pln] = u32x4{d0[0], d0[1], d1[0], d1[1]};
        }
        vh = as_frag(w[0]); vm = as_frag(w[1]); vl = as_frag(w[2]);
      }
#define AS_O(VP, PP) \
  _Pragma("unroll") for (int t = 0; t < QT; ++t) acc[t][blk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(VP, PP[t], acc[t][blk], 0, 0, 0);
      AS_O(vl, ph)
      AS_O(vh, pl)
      AS_O(vm, pm)
      AS_O(vh, pm)
      AS_O(vm, ph)
      AS_O(vh, ph)
#undef AS_O
    }
    if (st + 1 < nstep) store_step(buf ^ 1);                         // the other buffer: last read one barrier ago
    __syncthreads();
  }

  if (!active) return;
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    float l = lrun[t];
    l = rows_sum(l);
    const float inv = 1.0f / l;
    const int qo = (qw * QT + t) * 16 + c;
    // lane (c, g) holds O[qo][16 blk + 4 g + r] in acc[t][blk][r]
    if (op) {                                           // block-uniform: the planes of the out-projection's A operand (gemm_planes.hip)
      // blocks 2u, 2u + 1 = one 32-column chunk; v_permlane16_swap leaves lane g with 8 consecutive columns of it (the plane
      // epilogue of gemm_planes.hip): one 16-byte slot per term.  Every lane takes part in the swap; rows >= Lq store nothing.
      const size_t ts = (size_t)o_rows * 64;
      const int slot = ((g & 1) << 1) | (g >> 1);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float lo4[4], hi4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const u32x2 sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[t][2 * u][r] * inv), __float_as_uint(acc[t][2 * u + 1][r] * inv), false, false);
          lo4[r] = __uint_as_float(sw[0]);
          hi4[r] = __uint_as_float(sw[1]);
        }
        if (h2_scale > 0.0f) {                        // block-uniform: two fp16 terms (gemm_h2.hip), scaled by the site's power of two
          unsigned hh[4], ll[4];
          split_pair_h2(f32x2{lo4[0], lo4[1]} * h2_scale, hh[0], ll[0]);
          split_pair_h2(f32x2{lo4[2], lo4[3]} * h2_scale, hh[1], ll[1]);
          split_pair_h2(f32x2{hi4[0], hi4[1]} * h2_scale, hh[2], ll[2]);
          split_pair_h2(f32x2{hi4[2], hi4[3]} * h2_scale, hh[3], ll[3]);
          if (qo < Lq) {
            char* dst = reinterpret_cast<char*>(op) + (((size_t)(2 * h + u) * 2) * o_rows + (size_t)b * Lq + qo) * 64 + slot * 16;
            *reinterpret_cast<u32x4*>(dst) = vec4(hh);
            *reinterpret_cast<u32x4*>(dst + ts) = vec4(ll);
          }
          continue;
        }
        const Planes w = split8(f32x4{lo4[0], lo4[1], lo4[2], lo4[3]}, f32x4{hi4[0], hi4[1], hi4[2], hi4[3]});
        if (qo < Lq) {
          char* dst = reinterpret_cast<char*>(op) + (((size_t)(2 * h + u) * 3) * o_rows + (size_t)b * Lq + qo) * 64 + slot * 16;
          *reinterpret_cast<u32x4*>(dst) = vec4(w.hi);
          *reinterpret_cast<u32x4*>(dst + ts) = vec4(w.mid);
          *reinterpret_cast<u32x4*>(dst + 2 * ts) = vec4(w.lo);
        }
      }
    } else if (qo < Lq) {
#pragma unroll
      for (int blk = 0; blk < 4; ++blk)
        *reinterpret_cast<f32x4*>(ob + (size_t)qo * ldo + 16 * blk + 4 * g) =
            f32x4{acc[t][blk][0] * inv, acc[t][blk][1] * inv, acc[t][blk][2] * inv, acc[t][blk][3] * inv};
    }
  }
}

// ---- two fp16 terms (round 5): the scheme of gemm_h2.hip applied to both products of the attention ---------------------------------
// Where the caller has a STATIC bound on |q|, |k| and |v| (self-attention behind a LayerNorm: the bound of the in-projection,
// avsep_api.hip h2_prepare), every operand is x 2^e = hi + lo with hi = rn16(x 2^e), lo = rn16(x 2^e - hi) (e puts the bound below
// 2^14; 22 significant bits for every |x| within 17 binades of the bound, 2^-39 of the bound below that), the probabilities --
// in [0, 1] by construction -- with e = 14, and a product is the THREE fp16 MFMA products (lo,hi) (hi,lo) (hi,hi): half the matrix
// work of the three-term kernel above and a cheaper split (one v_cvt_pk_f16_f32 pair per two values).  The scores stay in units of
// 2^(eq + ek): the running maximum and the exponentials work on the scaled values with the constants of exp_neg scaled by the same
// power of two (exact), so the descale costs nothing; O is descaled by 2^-(14 + ev) together with the division by the row sum.
// Same tiling, same LDS images (two planes per operand instead of three: 16 KB per step), same fragment reads.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int AH_BUF = 4 * AS_PL;            // [K hi|lo][V hi|lo] = 16 KB

struct Planes2 {
  unsigned hi[4], lo[4];
};
__device__ __forceinline__ Planes2 split8_h2(const f32x4 a, const f32x4 b, float mul) {
  Planes2 w;
  split_pair_h2(f32x2{a[0], a[1]} * mul, w.hi[0], w.lo[0]);
  split_pair_h2(f32x2{a[2], a[3]} * mul, w.hi[1], w.lo[1]);
  split_pair_h2(f32x2{b[0], b[1]} * mul, w.hi[2], w.lo[2]);
  split_pair_h2(f32x2{b[2], b[3]} * mul, w.hi[3], w.lo[3]);
  return w;
}
__device__ __forceinline__ f16x8 as_frag_h(const u32x4 v) { return __builtin_bit_cast(f16x8, v); }
// exp((x) * c) for a power of two c, x <= 0: exp_neg_finite / exp_neg_pair with their three constants multiplied by c (exact)
__device__ __forceinline__ float exp_neg_scaled(float x, float hi, float lo) {
  const float t = x * hi;
  float r = fmaf(x, hi, -t);
  r = fmaf(x, lo, r);
  const float e = __builtin_amdgcn_exp2f(t);
  return fmaf(e, r * 0.69314718055994530942f, e);
}
__device__ __forceinline__ f32x2 exp_neg_pair_scaled(f32x2 x, float hi, float lo) {
  const f32x2 HI = {hi, hi}, LO = {lo, lo};
  const f32x2 LN2 = {0.69314718055994530942f, 0.69314718055994530942f};
  const f32x2 t = x * HI;
  f32x2 r = __builtin_elementwise_fma(x, HI, -t);
  r = __builtin_elementwise_fma(x, LO, r);
  f32x2 e;
  e[0] = __builtin_amdgcn_exp2f(t[0]);
  e[1] = __builtin_amdgcn_exp2f(t[1]);
  return __builtin_elementwise_fma(e, r * LN2, e);
}

template <int QT>
__global__ __launch_bounds__(256, 3) void attention_h2_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                              int ldk, const float* __restrict__ v, int ldv, float* __restrict__ o,
                                                              int ldo, int nhead, int Lq, int Lk, int nqt, float qmul, float kmul,
                                                              float vmul, float sdesc, float odesc, unsigned short* __restrict__ op,
                                                              long long o_rows, float h2_scale, const int* __restrict__ clip_exp,
                                                              int clip_stride, int eq, float* __restrict__ rs_out) {
  constexpr int DH = 64;
  __shared__ __attribute__((aligned(16))) char lds[2 * AH_BUF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int nqw = (nqt + QT - 1) / QT;
  const int wg_per_head = (nqw + 3) >> 2;
  const int bh = blockIdx.x / wg_per_head;
  const int qw = (blockIdx.x - bh * wg_per_head) * 4 + wave;
  const bool active = qw < nqw;                       // no wave leaves early (barriers): idle waves compute on clamped rows
  const int b = bh / nhead, h = bh - b * nhead;
  float rs_clip = 0.0f;
  if (clip_exp) {                                     // block-uniform: this clip's k / v exponents (cross-attention, clip_exp_kernel)
    const int ek = clip_exp[(size_t)b * clip_stride], ev = clip_exp[(size_t)b * clip_stride + 1];
    kmul = ldexpf(1.0f, ek);
    vmul = ldexpf(1.0f, ev);
    sdesc = ldexpf(1.0f, -(eq + ek));
    odesc = ldexpf(1.0f, -(14 + ev));
    h2_scale = vmul;                                  // |o| <= the bound of v: the out-projection's operand carries the clip's 2^ev
    rs_clip = ldexpf(1.0f, -ev);
  }
  const float* qb = q + (size_t)b * Lq * ldq + h * DH;
  const float* kb = k + (size_t)b * Lk * ldk + h * DH;
  const float* vb = v + (size_t)b * Lk * ldv + h * DH;
  float* ob = o + (size_t)b * Lq * ldo + h * DH;
  const float l2e_hi = 1.44269502162933349609f * sdesc, l2e_lo = 1.92596299112661746e-08f * sdesc;   // sdesc = 2^-(eq + ek)
  // a masked score in the scores' units: below every real one (|scaled score| <= 64 * 2^28) and finite times l2e_hi for sdesc <= 2^60
  const float masked = MASKED * fminf(1.0f, 1.0f / sdesc);

  f16x8 qh[QT][2], ql[QT][2];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int qrow = min((qw * QT + t) * 16 + c, Lq - 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const float* src = qb + (size_t)qrow * ldq + 32 * s + 8 * g;
      const Planes2 w = split8_h2(*reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 4), qmul);
      qh[t][s] = as_frag_h(vec4(w.hi)); ql[t][s] = as_frag_h(vec4(w.lo));
    }
  }
  f32x4 acc[QT][4];
  float mrun[QT], lrun[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    mrun[t] = masked;
    lrun[t] = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int srow = tid >> 3, spc = tid & 7;
  const int kst = srow * 128 + ((spc ^ (srow & 7)) << 4);
  const int vsw = (srow >> 1) & 3;
  const int vst0 = 2 * AS_PL + srow * 128 + ((((2 * spc) >> 2) ^ vsw) << 5) + (((2 * spc) & 3) << 3);
  const int vst1 = 2 * AS_PL + srow * 128 + ((((2 * spc + 1) >> 2) ^ vsw) << 5) + (((2 * spc + 1) & 3) << 3);
  f32x4 sk0, sk1, sv0, sv1;
  const int nstep = (Lk + AS_KEYS - 1) / AS_KEYS;
  auto load_step = [&](int st) {
    const int key = min(st * AS_KEYS + srow, Lk - 1);
    const float* ks = kb + (size_t)key * ldk + 8 * spc;
    const float* vs = vb + (size_t)key * ldv + 8 * spc;
    sk0 = *reinterpret_cast<const f32x4*>(ks); sk1 = *reinterpret_cast<const f32x4*>(ks + 4);
    sv0 = *reinterpret_cast<const f32x4*>(vs); sv1 = *reinterpret_cast<const f32x4*>(vs + 4);
  };
  auto store_step = [&](int buf) {
    char* base = lds + buf * AH_BUF;
    const Planes2 wk = split8_h2(sk0, sk1, kmul);
    *reinterpret_cast<u32x4*>(base + kst) = vec4(wk.hi);
    *reinterpret_cast<u32x4*>(base + AS_PL + kst) = vec4(wk.lo);
    const Planes2 wv = split8_h2(sv0, sv1, vmul);
    *reinterpret_cast<u32x2*>(base + vst0) = u32x2{wv.hi[0], wv.hi[1]};
    *reinterpret_cast<u32x2*>(base + vst1) = u32x2{wv.hi[2], wv.hi[3]};
    *reinterpret_cast<u32x2*>(base + AS_PL + vst0) = u32x2{wv.lo[0], wv.lo[1]};
    *reinterpret_cast<u32x2*>(base + AS_PL + vst1) = u32x2{wv.lo[2], wv.lo[3]};
  };
  int kfo[2][2];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int row = 16 * kt + c;
      kfo[kt][s] = row * 128 + (((4 * s + g) ^ (row & 7)) << 4);
    }
  int vfo[2];
  {
    const int qq = c >> 2, p = c & 3;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const int row = 16 * kt + 4 * g + qq;
      vfo[kt] = 2 * AS_PL + row * 128 + (p << 3) + ((((row >> 1) & 3)) << 5);
    }
  }

  load_step(0);
  store_step(0);
  __syncthreads();
  for (int st = 0; st < nstep; ++st) {
    const int buf = st & 1;
    if (st + 1 < nstep) load_step(st + 1);
    const char* base = lds + buf * AH_BUF;

    f32x4 stt[2][QT];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int t = 0; t < QT; ++t) stt[kt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f16x8 kh[2], kl[2];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        kh[kt] = *reinterpret_cast<const f16x8*>(base + kfo[kt][s]);
        kl[kt] = *reinterpret_cast<const f16x8*>(base + AS_PL + kfo[kt][s]);
      }
#define AH_S(KP, QP)                                                                                       \
  _Pragma("unroll") for (int kt = 0; kt < 2; ++kt) _Pragma("unroll") for (int t = 0; t < QT; ++t)         \
      stt[kt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(KP[kt], QP[t][s], stt[kt][t], 0, 0, 0);
      AH_S(kl, qh)
      AH_S(kh, ql)
      AH_S(kh, qh)
#undef AH_S
    }

    f16x8 ph[QT], pl[QT];
    const bool ragged = (st + 1) * AS_KEYS > Lk;
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      if (ragged) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) stt[kt][t][r] = st * AS_KEYS + 16 * kt + 4 * g + r < Lk ? stt[kt][t][r] : masked;
      }
      float tmax = fmaxf(fmaxf(fmaxf(stt[0][t][0], stt[0][t][1]), fmaxf(stt[0][t][2], stt[0][t][3])),
                         fmaxf(fmaxf(stt[1][t][0], stt[1][t][1]), fmaxf(stt[1][t][2], stt[1][t][3])));
      tmax = rows_max(tmax);
      const float mnew = fmaxf(mrun[t], tmax);
      const float alpha = exp_neg_scaled(mrun[t] - mnew, l2e_hi, l2e_lo);
      const f32x2 m2 = {mnew, mnew};
      const f32x2 p01 = exp_neg_pair_scaled(f32x2{stt[0][t][0], stt[0][t][1]} - m2, l2e_hi, l2e_lo);
      const f32x2 p23 = exp_neg_pair_scaled(f32x2{stt[0][t][2], stt[0][t][3]} - m2, l2e_hi, l2e_lo);
      const f32x2 p45 = exp_neg_pair_scaled(f32x2{stt[1][t][0], stt[1][t][1]} - m2, l2e_hi, l2e_lo);
      const f32x2 p67 = exp_neg_pair_scaled(f32x2{stt[1][t][2], stt[1][t][3]} - m2, l2e_hi, l2e_lo);
      const float psum = (((p01[0] + p01[1]) + (p23[0] + p23[1])) + ((p45[0] + p45[1]) + (p67[0] + p67[1])));
      lrun[t] = lrun[t] * alpha + psum;
      mrun[t] = mnew;
      Planes2 w;
      split_pair_h2(p01 * 16384.0f, w.hi[0], w.lo[0]);
      split_pair_h2(p23 * 16384.0f, w.hi[1], w.lo[1]);
      split_pair_h2(p45 * 16384.0f, w.hi[2], w.lo[2]);
      split_pair_h2(p67 * 16384.0f, w.hi[3], w.lo[3]);
      ph[t] = as_frag_h(vec4(w.hi)); pl[t] = as_frag_h(vec4(w.lo));
      const f32x2 a2 = {alpha, alpha};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x2 lo2 = f32x2{acc[t][i][0], acc[t][i][1]} * a2, hi2 = f32x2{acc[t][i][2], acc[t][i][3]} * a2;
        acc[t][i] = f32x4{lo2[0], lo2[1], hi2[0], hi2[1]};
      }
    }

#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      f16x8 vh, vl;
      {
        typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
        u32x4 w[2];
#pragma unroll
        for (int pln = 0; pln < 2; ++pln) {
          const s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base + pln * AS_PL + (vfo[0] ^ (blk << 5))));
          const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base + pln * AS_PL + (vfo[1] ^ (blk << 5))));
          const u32x2 d0 = __builtin_bit_cast(u32x2, r0), d1 = __builtin_bit_cast(u32x2, r1);
          w[pln] = u32x4{d0[0], d0[1], d1[0], d1[1]};
        }
        vh = as_frag_h(w[0]); vl = as_frag_h(w[1]);
      }
#define AH_O(VP, PP) \
  _Pragma("unroll") for (int t = 0; t < QT; ++t) acc[t][blk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(VP, PP[t], acc[t][blk], 0, 0, 0);
      AH_O(vl, ph)
      AH_O(vh, pl)
      AH_O(vh, ph)
#undef AH_O
    }
    if (st + 1 < nstep) store_step(buf ^ 1);
    __syncthreads();
  }

  if (!active) return;
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    float l = lrun[t];
    l = rows_sum(l);
    const float inv = odesc / l;                        // odesc = 2^-(14 + ev)
    const int qo = (qw * QT + t) * 16 + c;
    if (op) {                                           // block-uniform: the two fp16 terms of the out-projection's A operand (gemm_h2.hip)
      const size_t ts = (size_t)o_rows * 64;
      const int slot = ((g & 1) << 1) | (g >> 1);
      if (rs_out && h == 0 && g == 0 && qo < Lq) rs_out[(size_t)b * Lq + qo] = rs_clip;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float lo4[4], hi4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const u32x2 sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[t][2 * u][r] * inv), __float_as_uint(acc[t][2 * u + 1][r] * inv), false, false);
          lo4[r] = __uint_as_float(sw[0]);
          hi4[r] = __uint_as_float(sw[1]);
        }
        unsigned hh[4], ll[4];
        split_pair_h2(f32x2{lo4[0], lo4[1]} * h2_scale, hh[0], ll[0]);
        split_pair_h2(f32x2{lo4[2], lo4[3]} * h2_scale, hh[1], ll[1]);
        split_pair_h2(f32x2{hi4[0], hi4[1]} * h2_scale, hh[2], ll[2]);
        split_pair_h2(f32x2{hi4[2], hi4[3]} * h2_scale, hh[3], ll[3]);
        if (qo < Lq) {
          char* dst = reinterpret_cast<char*>(op) + (((size_t)(2 * h + u) * 2) * o_rows + (size_t)b * Lq + qo) * 64 + slot * 16;
          *reinterpret_cast<u32x4*>(dst) = vec4(hh);
          *reinterpret_cast<u32x4*>(dst + ts) = vec4(ll);
        }
      }
    } else if (qo < Lq) {
#pragma unroll
      for (int blk = 0; blk < 4; ++blk)
        *reinterpret_cast<f32x4*>(ob + (size_t)qo * ldo + 16 * blk + 4 * g) =
            f32x4{acc[t][blk][0] * inv, acc[t][blk][1] * inv, acc[t][blk][2] * inv, acc[t][blk][3] * inv};
    }
  }
}

}  // namespace

bool attention_split_supported(int dh, int Lq, int Lk) { return dh == 64 && Lq > 0 && Lk > 0; }

hipError_t launch_attention_split(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo,
                                  int B, int nhead, int dh, int Lq, int Lk, float qscale, hipStream_t s, unsigned short* op,
                                  long long o_rows, int h2, int h2_exp) {
  if (B <= 0 || nhead <= 0 || !attention_split_supported(dh, Lq, Lk)) return hipErrorInvalidValue;
  if (op && o_rows < (long long)B * Lq) return hipErrorInvalidValue;
  if ((ldq | ldk | ldv | ldo) & 3) return hipErrorInvalidValue;    // float4 row alignment
  constexpr int QT = 2;
  const int nqt = (Lq + 15) / 16, nqw = (nqt + QT - 1) / QT;
  const dim3 grid((unsigned)((long)B * nhead * ((nqw + 3) / 4)));
  hipLaunchKernelGGL((attention_split_kernel<QT>), grid, dim3(256), 0, s, q, ldq, k, ldk, v, ldv, o, ldo, nhead, Lq, Lk, nqt, qscale, op, o_rows,
                     h2 ? ldexpf(1.0f, h2_exp) : 0.0f);
  return hipGetLastError();
}

// eq / ek / ev: the static exponents of q, k, v (|x| 2^e <= 2^14 for every entry the caller can produce: the caller's bound, not a
// measurement).  op: the attention output as the two fp16 planes of the out-projection (scaled by 2^h2_exp) instead of fp32 o.
bool attention_h2_supported(int dh, int Lq, int Lk, int eq, int ek, int ev) {
  return dh == 64 && Lq > 0 && Lk > 0 && abs(eq) <= 60 && abs(ek) <= 60 && abs(ev) <= 60 && abs(eq + ek) <= 60;
}
hipError_t launch_attention_h2(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo, int B,
                               int nhead, int dh, int Lq, int Lk, int eq, int ek, int ev, hipStream_t s, unsigned short* op,
                               long long o_rows, int h2_exp, const int* clip_exp, int clip_stride, float* rs_out) {
  if (clip_exp) { ek = ev = 0; }                      // read per clip inside the kernel (clamped to +-60 by clip_exp_kernel)
  if (B <= 0 || nhead <= 0 || !attention_h2_supported(dh, Lq, Lk, eq, ek, ev)) return hipErrorInvalidValue;
  if (clip_exp && (clip_stride < 2 || eq < 0 || eq > 60 || !op || !rs_out)) return hipErrorInvalidValue;   // eq + ek >= -60 with the clips' ek >= -60
  if (op && o_rows < (long long)B * Lq) return hipErrorInvalidValue;
  if (!op && !o) return hipErrorInvalidValue;
  if ((ldq | ldk | ldv | ldo) & 3) return hipErrorInvalidValue;
  constexpr int QT = 2;
  const int nqt = (Lq + 15) / 16, nqw = (nqt + QT - 1) / QT;
  const dim3 grid((unsigned)((long)B * nhead * ((nqw + 3) / 4)));
  hipLaunchKernelGGL((attention_h2_kernel<QT>), grid, dim3(256), 0, s, q, ldq, k, ldk, v, ldv, o, ldo, nhead, Lq, Lk, nqt, ldexpf(1.0f, eq),
                     ldexpf(1.0f, ek), ldexpf(1.0f, ev), ldexpf(1.0f, -(eq + ek)), ldexpf(1.0f, -(14 + ev)), op, o_rows, ldexpf(1.0f, h2_exp), clip_exp,
                     clip_stride, eq, rs_out);
  return hipGetLastError();
}
