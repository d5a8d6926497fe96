// attention.hip -- softmax(q k^T) v for one (batch, head, 16-query tile) per wavefront, fp32 on the
// matrix cores, online softmax, no score matrix in memory.
//
// Replaces the score/softmax/PV part of nn.MultiheadAttention for both the self-attention of
// nn.TransformerEncoderLayer (model.py:48-52, 97-101) and the audio-queries-visual cross-attention
// (model.py:155,169).  q arrives already multiplied by 1/sqrt(dh): the weight packer folds the scale into
// Wq/bq, which is where torch's fast path applies it too.
//
// Layout trick (CDNA4 16x16x4 fp32 MFMA, C/D: col = lane&15, row = 4*(lane>>4)+reg):
//   S^T = K Q^T  (A = K rows, B = Q^T)  ->  lane (c = query, g) holds S^T[key = 4g+r][c] in reg r
//   O^T = V^T P^T (A = V^T, B = P^T)    ->  the P^T operand of MFMA step r is exactly reg r of S^T:
//                                           no LDS round trip, no cross-lane movement for P
//   the softmax statistics of query c live on the 4 lanes {c, c+16, c+32, c+48}: two shuffles per tile.
// V^T rows are assigned dv = NB*i + blk so each lane's V read is NB contiguous floats and each lane ends
// up owning 4 consecutive dv per accumulator register -> float4 stores.
// K/V are read straight from L2 (a head's K and V are <= 2*Lk*dh*4 B, shared by the 4 waves of the
// workgroup through the CU's L1); fp32 MFMA needs 1 operand dword per lane per 32 cycles, so there is
// nothing for an LDS stage to win here.
#include "kernels.h"
#include "attn_tile.h"
#include <cstdlib>
#include <cstdio>

namespace {

// exp_neg on a pair with packed fp32 VALU (v_pk_mul / v_pk_fma issue two lanes' worth of IEEE operations per instruction
// at the scalar instruction's cost): the same operations in the same order as exp_neg, so the same bits.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 exp_neg2(f32x2 x) {
  const f32x2 HI = {1.44269502162933349609f, 1.44269502162933349609f};
  const f32x2 LO = {1.92596299112661746e-08f, 1.92596299112661746e-08f};
  const f32x2 LN2 = {0.69314718055994530942f, 0.69314718055994530942f};
  x[0] = fmaxf(x[0], -120.0f);
  x[1] = fmaxf(x[1], -120.0f);
  const f32x2 t = x * HI;
  f32x2 r = __builtin_elementwise_fma(x, HI, -t);
  r = __builtin_elementwise_fma(x, LO, r);
  f32x2 e;
  e[0] = __builtin_amdgcn_exp2f(t[0]);
  e[1] = __builtin_amdgcn_exp2f(t[1]);
  return __builtin_elementwise_fma(e, r * LN2, e);
}

// NB = ceil(dh / 16).  REG = (dh == 16*NB): the fast path with unpredicated vector loads; otherwise the head
// is zero-extended to 16*NB (k >= dh contributes 0 to S, rows dv >= dh of O^T are never stored).
// QT = 16-query tiles per wave.  With QT = 2 every K / V fragment fetched from L2 feeds two score tiles and two
// output tiles (half the L1/L2 traffic per flop: at Lk = 501 a wave streams 256 KB of K and V) and the two score
// accumulators interleave, hiding the 40-cycle dependent latency of back-to-back MFMAs on one accumulator.
template <int NB, bool REG, int QT>
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ q, int ldq,
                                                        const float* __restrict__ k, int ldk,
                                                        const float* __restrict__ v, int ldv,
                                                        float* __restrict__ o, int ldo, int nhead, int Lq,
                                                        int Lk, int nqt, int DH, float qscale,
                                                        float* __restrict__ lse, float drop_p,
                                                        unsigned long long drop_seed) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int c = lane & 15;    // query column (as B operand / C column); key row (as A operand of S^T)
  const int g = lane >> 4;    // k-quarter of the MFMA
  const int nqw = (nqt + QT - 1) / QT;          // wave-units (QT tiles each) per head
  const int wg_per_head = (nqw + 3) >> 2;
  const int bh = blockIdx.x / wg_per_head;
  const int qw = (blockIdx.x - bh * wg_per_head) * 4 + wave;
  if (qw >= nqw) return;      // no barriers in this kernel: a wave may leave early
  const int b = bh / nhead, h = bh - b * nhead;

  const float* qb = q + (size_t)b * Lq * ldq + h * DH;
  const float* kb = k + (size_t)b * Lk * ldk + h * DH;
  const float* vb = v + (size_t)b * Lk * ldv + h * DH;
  float* ob = o + (size_t)b * Lq * ldo + h * DH;

  // Q^T fragments: lane (c,g) holds Q[q0+c][16s+4g .. +3], s < NB   (k-permuted, same as K below)
  f32x4 qf[QT][NB];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int qrow = min((qw * QT + t) * 16 + c, Lq - 1);
#pragma unroll
    for (int s = 0; s < NB; ++s)
      qf[t][s] = ((REG || 16 * s + 4 * g < DH) ? *reinterpret_cast<const f32x4*>(qb + (size_t)qrow * ldq + 16 * s + 4 * g)
                                               : f32x4{0.f, 0.f, 0.f, 0.f}) * qscale;
  }

  f32x4 acc[QT][NB];
  float mrun[QT], lrun[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    mrun[t] = -INFINITY;
    lrun[t] = 0.0f;
#pragma unroll
    for (int i = 0; i < NB; ++i) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int nkt = (Lk + 15) >> 4;
  // software prefetch of the next key tile's K and V fragments
  f32x4 kf[NB];
  float vf[4][NB];
  auto load_tile = [&](int kt) {
    const int krow = min(kt * 16 + c, Lk - 1);
#pragma unroll
    for (int s = 0; s < NB; ++s)
      kf[s] = (REG || 16 * s + 4 * g < DH) ? *reinterpret_cast<const f32x4*>(kb + (size_t)krow * ldk + 16 * s + 4 * g)
                                           : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int vrow = min(kt * 16 + 4 * g + r, Lk - 1);
      const float* src = vb + (size_t)vrow * ldv + NB * c;
      if constexpr (!REG) {
#pragma unroll
        for (int e = 0; e < NB; ++e) vf[r][e] = (NB * c + e < DH) ? src[e] : 0.0f;
      } else if constexpr (NB == 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(src);
        vf[r][0] = t[0]; vf[r][1] = t[1]; vf[r][2] = t[2]; vf[r][3] = t[3];
      } else if constexpr (NB == 8) {
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(src);
        const f32x4 t1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { vf[r][e] = t0[e]; vf[r][4 + e] = t1[e]; }
      } else {
#pragma unroll
        for (int e = 0; e < NB; ++e) vf[r][e] = src[e];
      }
    }
  };

  load_tile(0);
  for (int kt = 0; kt < nkt; ++kt) {
    f32x4 kc[NB];
    float vc[4][NB];
#pragma unroll
    for (int s = 0; s < NB; ++s) kc[s] = kf[s];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int e = 0; e < NB; ++e) vc[r][e] = vf[r][e];
    if (kt + 1 < nkt) load_tile(kt + 1);

    // S^T tiles: 16 keys x 16 queries each, contraction over dh; the QT accumulators interleave
    f32x4 st[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) st[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NB; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < QT; ++t) st[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kc[s][j], qf[t][s][j], st[t], 0, 0, 0);

    float pr[QT][4];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      // mask keys beyond Lk, online softmax over this tile's 16 keys of query c
      float tmax = -INFINITY;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + 4 * g + r;
        st[t][r] = key < Lk ? st[t][r] : -INFINITY;
        tmax = fmaxf(tmax, st[t][r]);
      }
      tmax = rows_max(tmax);
      const float mnew = fmaxf(mrun[t], tmax);        // finite: tile 0 always holds key 0
      const float alpha = exp_neg(mrun[t] - mnew);     // 0 on the first tile (mrun = -inf)
      float psum = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pr[t][r] = exp_neg(st[t][r] - mnew);
        psum += pr[t][r];
      }
      lrun[t] = lrun[t] * alpha + psum;                // per-lane partial (own 4 keys per tile); reduced at the end
      mrun[t] = mnew;
      if (drop_p > 0.0f) {   // training: dropout on the attention probabilities (nn.MultiheadAttention dropout=p);
                             // the normaliser keeps the undropped sum, kept entries are scaled by 1/(1-p)
        const float keep_scale = 1.0f / (1.0f - drop_p);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const unsigned long long e =
              ((unsigned long long)bh * Lq + ((qw * QT + t) * 16 + c)) * Lk + (kt * 16 + 4 * g + r);
          pr[t][r] = dropout_keep(drop_seed, e, drop_p) ? pr[t][r] * keep_scale : 0.0f;
        }
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) acc[t][i] *= alpha;
    }

    // O^T += V^T P^T : step r contracts keys {4g'+r}, A row i <-> dv = NB*i + blk
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int t = 0; t < QT; ++t)
          acc[t][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(vc[r][blk], pr[t][r], acc[t][blk], 0, 0, 0);
  }

#pragma unroll
  for (int t = 0; t < QT; ++t) {
    float l = lrun[t];
    l = rows_sum(l);
    const float inv = 1.0f / l;
    const int qo = (qw * QT + t) * 16 + c;
    if (lse && g == 0 && qo < Lq) lse[(size_t)bh * Lq + qo] = mrun[t] + logf(l);   // training: softmax statistics
    if (qo < Lq) {
      // lane (c, g) holds O[qo][dv = NB*(4g+reg) + blk]; for a fixed reg the NB blks are contiguous
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* dst = ob + (size_t)qo * ldo + NB * (4 * g + r);
        if constexpr (!REG) {
#pragma unroll
          for (int blk = 0; blk < NB; ++blk)
            if (NB * (4 * g + r) + blk < DH) dst[blk] = acc[t][blk][r] * inv;
        } else if constexpr (NB == 4) {
          *reinterpret_cast<f32x4*>(dst) =
              f32x4{acc[t][0][r] * inv, acc[t][1][r] * inv, acc[t][2][r] * inv, acc[t][3][r] * inv};
        } else {
#pragma unroll
          for (int blk = 0; blk < NB; ++blk) dst[blk] = acc[t][blk][r] * inv;
        }
      }
    }
  }
}

// Long-sequence variant for dh = 64 (L = 251 / 501 of configs 3-5): the SAME arithmetic in the SAME order as
// attention_kernel<4, true, QT> (bit-identical results), but the four waves of a workgroup -- four different groups of
// queries of one (clip, head) -- take their K / V fragments from an LDS tile that the workgroup loads ONCE, instead
// of each wave streaming all of K and V through the CU's L1 by itself.  A wave needs 8 KB of K / V per 16 keys for
// 64 MFMAs (2048 matrix cycles, QT = 2); four waves per SIMD-set asked the 64 B/clk L1 for ~48 B/clk (PMC: matrix
// pipes 45 % busy).  Through LDS the global traffic is 4x smaller and the fragment reads cost 16 B/clk of the LDS's 128+.
//   * stage = KT key tiles (16*KT keys): K rows [key][64] with the 16-byte slot XOR-swizzled by the key (the K fragment
//     read is 16 lanes x 16 different rows at one column), V rows [key][64] linear (its fragment read is 16 lanes x one
//     row); double-buffered, global -> registers before the stage's MFMAs, registers -> LDS after them, one barrier.
//   * no wave leaves early (barriers): waves without queries compute on clamped rows and store nothing.
template <int QT, int KT, bool DROP>
__global__ __launch_bounds__(256, QT == 4 ? 2 : DROP ? 1 : (QT == 2 ? 3 : 4)) void attention_lds_kernel(const float* __restrict__ q, int ldq,
                                                            const float* __restrict__ k, int ldk,
                                                            const float* __restrict__ v, int ldv,
                                                            float* __restrict__ o, int ldo, int nhead, int Lq, int Lk,
                                                            int nqt, float qscale, float* __restrict__ lse,
                                                            float drop_p, unsigned long long drop_seed) {
  constexpr int NB = 4, DH = 64;
  constexpr int ROWS = 16 * KT;                       // keys per stage
  constexpr int NLD = (2 * ROWS * 16) / 256;          // float4 loads per thread per stage (K and V)
  static_assert((2 * ROWS * 16) % 256 == 0, "stage must be a whole number of float4 per thread");
  __shared__ __attribute__((aligned(16))) float lds[2][2][ROWS * DH];   // [buffer][K|V][key][64]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int c = lane & 15;
  const int g = lane >> 4;
  const int nqw = (nqt + QT - 1) / QT;
  const int wg_per_head = (nqw + 3) >> 2;
  const int bh = blockIdx.x / wg_per_head;
  const int qw = (blockIdx.x - bh * wg_per_head) * 4 + wave;
  const bool active = qw < nqw;
  const int b = bh / nhead, h = bh - b * nhead;

  const float* qb = q + (size_t)b * Lq * ldq + h * DH;
  const float* kb = k + (size_t)b * Lk * ldk + h * DH;
  const float* vb = v + (size_t)b * Lk * ldv + h * DH;
  float* ob = o + (size_t)b * Lq * ldo + h * DH;

  f32x4 qf[QT][NB];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int qrow = min((qw * QT + t) * 16 + c, Lq - 1);
#pragma unroll
    for (int s = 0; s < NB; ++s) qf[t][s] = *reinterpret_cast<const f32x4*>(qb + (size_t)qrow * ldq + 16 * s + 4 * g) * qscale;
  }
  f32x4 acc[QT][NB];
  float mrun[QT], lrun[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    mrun[t] = -INFINITY;
    lrun[t] = 0.0f;
#pragma unroll
    for (int i = 0; i < NB; ++i) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // staging: float4 index f = tid + 256*l of a stage's 2*ROWS*16 float4s; first ROWS*16 are K, the rest V
  const int nkt = (Lk + 15) >> 4;
  const int nstage = (nkt + KT - 1) / KT;
  f32x4 stg[NLD];
  auto load_stage = [&](int st) {
#pragma unroll
    for (int l = 0; l < NLD; ++l) {
      const int f = tid + 256 * l;
      const bool isv = f >= ROWS * 16;
      const int ff = isv ? f - ROWS * 16 : f;
      const int row = ff >> 4, slot = ff & 15;
      const int key = min(st * ROWS + row, Lk - 1);
      const float* src = (isv ? vb + (size_t)key * ldv : kb + (size_t)key * ldk) + 4 * slot;
      stg[l] = *reinterpret_cast<const f32x4*>(src);
    }
  };
  auto store_stage = [&](int buf) {
#pragma unroll
    for (int l = 0; l < NLD; ++l) {
      const int f = tid + 256 * l;
      const bool isv = f >= ROWS * 16;
      const int ff = isv ? f - ROWS * 16 : f;
      const int row = ff >> 4, slot = ff & 15;
      float* dst = &lds[buf][isv ? 1 : 0][row * DH + ((isv ? slot : (slot ^ (row & 15))) << 2)];
      *reinterpret_cast<f32x4*>(dst) = stg[l];
    }
  };

  load_stage(0);
  store_stage(0);
  __syncthreads();
  for (int st = 0; st < nstage; ++st) {
    const int buf = st & 1;
    if (st + 1 < nstage) load_stage(st + 1);
    const float* Ks = lds[buf][0];
    const float* Vs = lds[buf][1];
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) {
      const int kt = st * KT + kk;
      if (kt < nkt) {                                   // block-uniform
        f32x4 kc[NB];
        float vc[4][NB];
#pragma unroll
        for (int s = 0; s < NB; ++s) {
          const int row = 16 * kk + c;
          kc[s] = *reinterpret_cast<const f32x4*>(Ks + row * DH + (((4 * s + g) ^ (row & 15)) << 2));
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const f32x4 t4 = *reinterpret_cast<const f32x4*>(Vs + (16 * kk + 4 * g + r) * DH + NB * c);
          vc[r][0] = t4[0]; vc[r][1] = t4[1]; vc[r][2] = t4[2]; vc[r][3] = t4[3];
        }
        // ---- from here on: attention_kernel<4, true, QT>'s tile body, operation for operation ----
        f32x4 stt[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) stt[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NB; ++s)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < QT; ++t) stt[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kc[s][j], qf[t][s][j], stt[t], 0, 0, 0);
        float pr[QT][4];
        const bool ragged = (kt + 1) * 16 > Lk;         // block-uniform: only the last tile can hold keys >= Lk
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          if (ragged) {
#pragma unroll
            for (int r = 0; r < 4; ++r) stt[t][r] = kt * 16 + 4 * g + r < Lk ? stt[t][r] : -INFINITY;
          }
          float tmax = fmaxf(fmaxf(fmaxf(fmaxf(-INFINITY, stt[t][0]), stt[t][1]), stt[t][2]), stt[t][3]);
          tmax = rows_max(tmax);
          const float mnew = fmaxf(mrun[t], tmax);
          const float alpha = exp_neg(mrun[t] - mnew);
          // the four probabilities as two packed pairs (same operations, same order, same bits as exp_neg per element)
          const f32x2 m2 = {mnew, mnew};
          const f32x2 p01 = exp_neg2(f32x2{stt[t][0], stt[t][1]} - m2);
          const f32x2 p23 = exp_neg2(f32x2{stt[t][2], stt[t][3]} - m2);
          pr[t][0] = p01[0]; pr[t][1] = p01[1]; pr[t][2] = p23[0]; pr[t][3] = p23[1];
          const float psum = (((0.0f + pr[t][0]) + pr[t][1]) + pr[t][2]) + pr[t][3];
          lrun[t] = lrun[t] * alpha + psum;
          mrun[t] = mnew;
          if (DROP && drop_p > 0.0f) {   // DROP = false: the inference instance carries none of this
            const float keep_scale = 1.0f / (1.0f - drop_p);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const unsigned long long e =
                  ((unsigned long long)bh * Lq + ((qw * QT + t) * 16 + c)) * Lk + (kt * 16 + 4 * g + r);
              pr[t][r] = dropout_keep(drop_seed, e, drop_p) ? pr[t][r] * keep_scale : 0.0f;
            }
          }
          const f32x2 a2 = {alpha, alpha};
#pragma unroll
          for (int i = 0; i < NB; ++i) {                // v_pk_mul_f32 on the accumulator's register pairs
            const f32x2 lo = f32x2{acc[t][i][0], acc[t][i][1]} * a2, hi = f32x2{acc[t][i][2], acc[t][i][3]} * a2;
            acc[t][i] = f32x4{lo[0], lo[1], hi[0], hi[1]};
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int blk = 0; blk < NB; ++blk)
#pragma unroll
            for (int t = 0; t < QT; ++t)
              acc[t][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(vc[r][blk], pr[t][r], acc[t][blk], 0, 0, 0);
      }
    }
    if (st + 1 < nstage) store_stage(buf ^ 1);          // the other buffer: last read one barrier ago
    __syncthreads();
  }

  if (!active) return;
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    float l = lrun[t];
    l = rows_sum(l);
    const float inv = 1.0f / l;
    const int qo = (qw * QT + t) * 16 + c;
    if (lse && g == 0 && qo < Lq) lse[(size_t)bh * Lq + qo] = mrun[t] + logf(l);
    if (qo < Lq) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        *reinterpret_cast<f32x4*>(ob + (size_t)qo * ldo + NB * (4 * g + r)) =
            f32x4{acc[t][0][r] * inv, acc[t][1][r] * inv, acc[t][2][r] * inv, acc[t][3][r] * inv};
    }
  }
}

// Short-sequence specialisation (49 <= Lk <= 64, dh = 64: the 1 s @ 8 kHz clips of BASELINE configs 1/2, T = 63
// audio frames / N = 50 lip frames): all NKT = 4 key tiles of K and V are loaded before the first MFMA, so the
// wave pays ONE memory round trip instead of one per tile (the generic kernel's 1-tile prefetch leaves a ~4 us
// latency chain at this size), and the softmax is the plain two-pass one over registers.
// Two problems may ride in one launch (pa: workgroups [0, blocks_a), pb: the rest) -- the self-attention of the audio and
// of the visual encoder layer i, which are independent and each fill only half of the chip's CUs (B * nhead = 128
// workgroups at the 32-clip batch).  A single problem is launched with blocks_a = gridDim.x.
template <int NB, int NKT>
__global__ __launch_bounds__(256) void attention_short_kernel(const AttnProblem pa, const AttnProblem pb, int blocks_a,
                                                              int nhead) {
  constexpr int DH = 16 * NB;
  const bool second = (int)blockIdx.x >= blocks_a;                      // block-uniform
  // field-wise scalar selects (a reference to one of two kernel arguments would send both through scratch memory)
  const float* __restrict__ q = second ? pb.q : pa.q;
  const float* __restrict__ k = second ? pb.k : pa.k;
  const float* __restrict__ v = second ? pb.v : pa.v;
  float* __restrict__ o = second ? pb.o : pa.o;
  const int ldq = second ? pb.ldq : pa.ldq, ldk = second ? pb.ldk : pa.ldk, ldv = second ? pb.ldv : pa.ldv;
  const int ldo = second ? pb.ldo : pa.ldo, Lq = second ? pb.Lq : pa.Lq, Lk = second ? pb.Lk : pa.Lk;
  const int nqt = (Lq + 15) >> 4;
  const int blk = second ? (int)blockIdx.x - blocks_a : (int)blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int c = lane & 15;
  const int g = lane >> 4;
  const int wg_per_head = (nqt + 3) >> 2;
  const int bh = blk / wg_per_head;
  const int qt = (blk - bh * wg_per_head) * 4 + wave;
  if (qt >= nqt) return;
  const int b = bh / nhead, h = bh - b * nhead;
  const float* qb = q + (size_t)b * Lq * ldq + h * DH;
  const float* kb = k + (size_t)b * Lk * ldk + h * DH;
  const float* vb = v + (size_t)b * Lk * ldv + h * DH;
  float* ob = o + (size_t)b * Lq * ldo + h * DH;

  f32x4 acc[NB];
  float lrun;
  attn_short_tile<NB, NKT>(qb, kb, vb, ldq, ldk, ldv, Lq, Lk, qt, c, g, acc, lrun);
  lrun = rows_sum(lrun);
  const float inv = 1.0f / lrun;
  const int qo = qt * 16 + c;
  if (qo < Lq) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *reinterpret_cast<f32x4*>(ob + (size_t)qo * ldo + NB * (4 * g + r)) =
          f32x4{acc[0][r] * inv, acc[1][r] * inv, acc[2][r] * inv, acc[3][r] * inv};
  }
}

#ifdef AVSEP_DEV
// Short-sequence attention AND the output projection behind it in one launch (round 3, developer experiment):
//     x[rows of the tile] += softmax(q k^T) v  W_o^T + b_o
// i.e. the `self_attn` / `cross_attn` call of a pre-norm block with its out_proj and the residual add (model.py:48-52 via
// nn.MultiheadAttention, 168-170).  One workgroup per (clip, 16-query tile); wavefront w computes head w exactly as
// attention_short_kernel does (attn_short_tile), parks its 16 x 64 slice of O in a swizzled LDS tile [16][d], and after one
// barrier the same wavefront computes the 64 output columns [64w, 64w+64) of the projection over K = d: A fragments from the
// LDS tile, W_o fragments straight from L2 through a register ring (every wave needs different rows of W_o, so there is
// nothing for an LDS stage to share), bias + residual in the epilogue, in place on x.
// Saves, per layer and branch, the attention launch's 128 workgroups on half of the chip, the kernel boundary, the round trip
// of O through memory and the 504-workgroup projection launch.  The MFMAs see the operands in the order of
// attention_short_kernel followed by gemm_kernel (k = 16s + 4q' + j: s outer, j, then the lane quarter inside the MFMA), and
// the epilogue adds bias then residual like gemm_kernel's: BIT-IDENTICAL to the two launches
// (tests/test_gpu_parity.py::test_op_attention_proj_equals_two_launches).
struct AttnProjParams {
  const float *q, *k, *v;     // head h at column offset 64 h
  int ldq, ldk, ldv;
  const float* wo;            // [d][d] (nn.Linear layout: row n = output column, K contiguous)
  const float* bo;            // [d] or null
  float* x;                   // [B*Lq][d]: residual in, result out
  int B, Lq, Lk;
};

template <int NH>
__global__ __launch_bounds__(64 * NH) void attn_proj_kernel(const AttnProjParams p) {
  constexpr int NB = 4, DH = 64, D = DH * NH, D16 = D / 16, P = D16 < 6 ? D16 : 6;
  __shared__ __attribute__((aligned(16))) float Os[16 * D];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;       // wave = head, later = 64-column group of the projection
  const int c = lane & 15, g = lane >> 4;
  const int nqt = (p.Lq + 15) >> 4;
  const int b = blockIdx.x / nqt, qt = blockIdx.x - b * nqt;
  {
    const float* qb = p.q + (size_t)b * p.Lq * p.ldq + wave * DH;
    const float* kb = p.k + (size_t)b * p.Lk * p.ldk + wave * DH;
    const float* vb = p.v + (size_t)b * p.Lk * p.ldv + wave * DH;
    f32x4 acc[NB];
    float lrun;
    attn_short_tile<NB, 4>(qb, kb, vb, p.ldq, p.ldk, p.ldv, p.Lq, p.Lk, qt, c, g, acc, lrun);
    lrun = rows_sum(lrun);
    const float inv = 1.0f / lrun;
    // O[query c][64 wave + 4 (4g + r) + blk] = acc[blk][r] * inv: 16-byte slot 16 wave + 4g + r of row c, XOR-swizzled by the row
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *reinterpret_cast<f32x4*>(Os + c * D + (((16 * wave + 4 * g + r) ^ c) << 2)) =
          f32x4{acc[0][r] * inv, acc[1][r] * inv, acc[2][r] * inv, acc[3][r] * inv};
  }
  // ---- output projection: this wave's columns n0 .. n0 + 63, four 16-column MFMA blocks, K = D
  const int fr = c, fq = g;                         // fragment row (query / W_o row inside a block), k quarter
  const int n0 = DH * wave;
  const float* wsrc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) wsrc[j] = p.wo + (size_t)(n0 + 16 * j + fr) * D + 4 * fq;
  f32x4 wr[P][4];
#pragma unroll
  for (int s = 0; s < P; ++s)
#pragma unroll
    for (int j = 0; j < 4; ++j) wr[s][j] = *reinterpret_cast<const f32x4*>(wsrc[j] + 16 * s);
  f32x4 acc2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc2[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();                                  // every head's O slice is in the tile
#pragma unroll
  for (int s = 0; s < D16; ++s) {
    const f32x4 fa = *reinterpret_cast<const f32x4*>(Os + fr * D + (((4 * s + fq) ^ fr) << 2));
    f32x4 fb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j] = wr[s % P][j];
    if (s + P < D16) {
#pragma unroll
      for (int j = 0; j < 4; ++j) wr[s % P][j] = *reinterpret_cast<const f32x4*>(wsrc[j] + 16 * (s + P));
    }
#pragma unroll
    for (int cc = 0; cc < 4; ++cc)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc2[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j][cc], fa[cc], acc2[j], 0, 0, 0);   // D^T like gemm_kernel
  }
  // ---- epilogue (gemm_kernel's fast path with a residual): all loads, then arithmetic, then stores
  const int qo = qt * 16 + fr;
  const int mrow = min(qo, p.Lq - 1);
  float* xr = p.x + ((size_t)b * p.Lq + mrow) * D + n0 + 4 * fq;
  f32x4 bv[4], rv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    bv[j] = p.bo ? *reinterpret_cast<const f32x4*>(p.bo + n0 + 16 * j + 4 * fq) : f32x4{0.f, 0.f, 0.f, 0.f};
    rv[j] = *reinterpret_cast<const f32x4*>(xr + 16 * j);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float t = p.bo ? acc2[j][e] + bv[j][e] : acc2[j][e];
      rv[j][e] = t + rv[j][e];
    }
  if (qo < p.Lq) {
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(xr + 16 * j) = rv[j];
  }
}
#endif  // AVSEP_DEV

}  // namespace

// the one-round-trip kernel: dh = 64 and 49..64 keys
static bool short_ok(int dh, int Lk) {
  static const bool no_short = dev_env("AVSEP_NO_SHORT_ATTN") != nullptr;   // developer A/B switch
  return dh == 64 && Lk > 48 && Lk <= 64 && !no_short;
}

bool attention_pair_merges(int dh, int Lk_a, int Lk_b) { return short_ok(dh, Lk_a) && short_ok(dh, Lk_b); }

// mirrors the dispatch of launch_attention_ex for inference calls (no LSE, no dropout, qscale 1)
const char* attention_instance_name(int dh, int Lq, int Lk, int B, int nhead) {
  static thread_local char buf[64];
  if (short_ok(dh, Lk)) return "attention_short_kernel<4, 4>";
  const int nb = (dh + 15) / 16, nqt = (Lq + 15) / 16;
  const bool reg = dh == 16 * nb;
  const long wgs2 = (long)B * nhead * (((nqt + 1) / 2 + 3) / 4);
  const char* qt_s = dev_env("AVSEP_ATTN_QT");
  const int qt_env = qt_s ? atoi(qt_s) : 0;
  const bool two = qt_env ? qt_env == 2 : (reg && nb == 4 && wgs2 >= 512);
  const bool lds = reg && nb == 4 && Lk >= 128 && !dev_env("AVSEP_ATTN_NO_LDS") && !dev_env("AVSEP_ATTN_NO_LDS_NOW");
  if (lds) snprintf(buf, sizeof buf, "attention_lds_kernel<%d, 2, false>", qt_env == 4 ? 4 : two ? 2 : 1);
  else snprintf(buf, sizeof buf, "attention_kernel<%d, %s, %d>", nb, reg ? "true" : "false", (two && reg && nb == 4) ? 2 : 1);
  return buf;
}

hipError_t launch_attention_pair(const AttnProblem& a, const AttnProblem& b, int nhead, int dh, hipStream_t s) {
  if (a.B <= 0 || b.B <= 0 || nhead <= 0 || a.Lq <= 0 || a.Lk <= 0 || b.Lq <= 0 || b.Lk <= 0) return hipErrorInvalidValue;
  if ((a.ldq | a.ldk | a.ldv | a.ldo | b.ldq | b.ldk | b.ldv | b.ldo) & 3) return hipErrorInvalidValue;
  if (short_ok(dh, a.Lk) && short_ok(dh, b.Lk)) {
    const int ba = a.B * nhead * ((((a.Lq + 15) >> 4) + 3) >> 2), bb = b.B * nhead * ((((b.Lq + 15) >> 4) + 3) >> 2);
    hipLaunchKernelGGL((attention_short_kernel<4, 4>), dim3((unsigned)(ba + bb)), dim3(256), 0, s, a, b, ba, nhead);
    return hipGetLastError();
  }
  hipError_t e = launch_attention(a.q, a.ldq, a.k, a.ldk, a.v, a.ldv, a.o, a.ldo, a.B, nhead, dh, a.Lq, a.Lk, s);
  if (e != hipSuccess) return e;
  return launch_attention(b.q, b.ldq, b.k, b.ldk, b.v, b.ldv, b.o, b.ldo, b.B, nhead, dh, b.Lq, b.Lk, s);
}

hipError_t launch_attention(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o,
                            int ldo, int B, int nhead, int dh, int Lq, int Lk, hipStream_t s) {
  return launch_attention_ex(q, ldq, k, ldk, v, ldv, o, ldo, B, nhead, dh, Lq, Lk, 1.0f, nullptr, 0.0f, 0ull, s);
}

// qscale: multiplies q on load (training path keeps 1/sqrt(dh) out of the weights); lse (B*nhead*Lq floats, may be
// null): log-sum-exp of each query's scores, consumed by the backward kernels.
hipError_t launch_attention_ex(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o,
                               int ldo, int B, int nhead, int dh, int Lq, int Lk, float qscale, float* lse,
                               float drop_p, unsigned long long drop_seed, hipStream_t s) {
  if (B <= 0 || nhead <= 0 || Lq <= 0 || Lk <= 0) return hipErrorInvalidValue;
  if ((ldq | ldk | ldv | ldo) & 3) return hipErrorInvalidValue;   // float4 row alignment
  const int nqt = (Lq + 15) / 16;
  const int wg_per_head = (nqt + 3) / 4;
  const dim3 grid((unsigned)(B * nhead * wg_per_head)), block(256);
  if (dh <= 0 || dh > 128 || (dh & 3)) return hipErrorInvalidValue;
  if (short_ok(dh, Lk) && !lse && qscale == 1.0f && drop_p == 0.0f) {
    const AttnProblem pa{q, k, v, o, ldq, ldk, ldv, ldo, B, Lq, Lk};
    hipLaunchKernelGGL((attention_short_kernel<4, 4>), grid, block, 0, s, pa, pa, (int)grid.x, nhead);
    return hipGetLastError();
  }
  const int nb = (dh + 15) / 16;
  const bool reg = (dh == 16 * nb);
  // two query tiles per wave once the sequence is long enough to keep >= 2 workgroups per CU that way
  const char* qt_s = dev_env("AVSEP_ATTN_QT");                     // developer A/B switch, read per call (bit-identity test)
  const int qt_env = qt_s ? atoi(qt_s) : 0;
  const long wgs2 = (long)B * nhead * (((nqt + 1) / 2 + 3) / 4);
  const bool two = qt_env ? qt_env == 2 : (reg && nb == 4 && wgs2 >= 512);
#define AVSEP_ATT(NB_)                                                                                              \
  if (nb == NB_) {                                                                                                  \
    if (reg) hipLaunchKernelGGL((attention_kernel<NB_, true, 1>), grid, block, 0, s, q, ldq, k, ldk, v, ldv, o, ldo, \
                                nhead, Lq, Lk, nqt, dh, qscale, lse, drop_p, drop_seed);                            \
    else hipLaunchKernelGGL((attention_kernel<NB_, false, 1>), grid, block, 0, s, q, ldq, k, ldk, v, ldv, o, ldo,   \
                            nhead, Lq, Lk, nqt, dh, qscale, lse, drop_p, drop_seed);                                \
  }
  static const bool no_lds = dev_env("AVSEP_ATTN_NO_LDS") != nullptr;      // developer A/B switch (whole process)
  const bool no_lds_now = dev_env("AVSEP_ATTN_NO_LDS_NOW") != nullptr;     // ... per call (the bit-identity test)
  if (reg && nb == 4 && Lk >= 128 && !no_lds && !no_lds_now) {
    // long sequences: K / V through LDS, shared by the workgroup's four waves (bit-identical to the kernels below)
    const bool drop = drop_p > 0.0f;
    // Four query tiles per wave (inference): one workgroup covers 256 queries, so L = 251 is ONE workgroup per (clip, head)
    // and the 512 workgroups of configs 3 / 5 are exactly one round at two workgroups per CU -- with two tiles per wave
    // they are 1024 workgroups on 768 slots, a second round one third full.  (profiles/r02_attention_bench.txt)
#ifdef AVSEP_DEV   // developer instance: measured without gain (profiles/r02_attention_bench.txt)
    const long wgs4 = (long)B * nhead * (((nqt + 3) / 4 + 3) / 4);
    const bool four = qt_env ? qt_env == 4 : false;
    if (four && !drop) {
      hipLaunchKernelGGL((attention_lds_kernel<4, 2, false>), dim3((unsigned)wgs4), block, 0, s, q, ldq, k, ldk, v, ldv, o, ldo,
                         nhead, Lq, Lk, nqt, qscale, lse, drop_p, drop_seed);
      return hipGetLastError();
    }
#endif
    if (two && drop) {
      hipLaunchKernelGGL((attention_lds_kernel<2, 2, true>), dim3((unsigned)wgs2), block, 0, s, q, ldq, k, ldk, v, ldv, o, ldo,
                         nhead, Lq, Lk, nqt, qscale, lse, drop_p, drop_seed);
    } else if (two) {
      hipLaunchKernelGGL((attention_lds_kernel<2, 2, false>), dim3((unsigned)wgs2), block, 0, s, q, ldq, k, ldk, v, ldv, o, ldo,
                         nhead, Lq, Lk, nqt, qscale, lse, drop_p, drop_seed);
    } else if (drop) {
      hipLaunchKernelGGL((attention_lds_kernel<1, 2, true>), grid, block, 0, s, q, ldq, k, ldk, v, ldv, o, ldo, nhead, Lq, Lk, nqt,
                         qscale, lse, drop_p, drop_seed);
    } else {
      hipLaunchKernelGGL((attention_lds_kernel<1, 2, false>), grid, block, 0, s, q, ldq, k, ldk, v, ldv, o, ldo, nhead, Lq, Lk, nqt,
                         qscale, lse, drop_p, drop_seed);
    }
    return hipGetLastError();
  }
  if (two && reg && nb == 4) {
    const dim3 grid2((unsigned)wgs2);
    hipLaunchKernelGGL((attention_kernel<4, true, 2>), grid2, block, 0, s, q, ldq, k, ldk, v, ldv, o, ldo, nhead, Lq, Lk,
                       nqt, dh, qscale, lse, drop_p, drop_seed);
    return hipGetLastError();
  }
  AVSEP_ATT(1) AVSEP_ATT(2) AVSEP_ATT(3) AVSEP_ATT(4) AVSEP_ATT(5) AVSEP_ATT(6) AVSEP_ATT(7) AVSEP_ATT(8)
#undef AVSEP_ATT
  return hipGetLastError();
}

// ---- attention + output projection + residual in one launch (attn_proj_kernel): dh = 64, 49..64 keys, d = 64 nhead <= 512
// Developer experiment (AVSEP_ATTN_PROJ=1): bit-identical to the two launches and SLOWER -- 15.4 us alone against 6.7 + 6.4, the
// step 0.365 vs 0.355 ms with two in flight and 0.463 vs 0.437 one at a time (profiles/r03_ab_attention_proj.txt).  With 16
// rows per workgroup the projection is 512 waves chasing fragment-shaped W_o loads (16 rows x 64 B per instruction) through
// L2, where the GEMM launch puts 2016 waves behind full-row staged tiles.
bool attn_proj_supported(int nhead, int dh, int Lk) {
  return short_ok(dh, Lk) && nhead >= 1 && nhead <= 8 && dev_env("AVSEP_ATTN_PROJ") != nullptr;
}
const char* attn_proj_instance_name(int nhead) {
  static thread_local char buf[32];
  snprintf(buf, sizeof buf, "attn_proj_kernel<%d>", nhead);
  return buf;
}
hipError_t launch_attn_proj(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* wo,
                            const float* bo, float* x, int B, int nhead, int dh, int Lq, int Lk, hipStream_t s) {
#ifndef AVSEP_DEV
  return hipErrorNotSupported;
#else
  if (!(short_ok(dh, Lk) && nhead >= 1 && nhead <= 8)) return hipErrorNotSupported;
  if (B <= 0 || Lq <= 0 || ((ldq | ldk | ldv) & 3)) return hipErrorInvalidValue;
  const AttnProjParams p{q, k, v, ldq, ldk, ldv, wo, bo, x, B, Lq, Lk};
  const dim3 grid((unsigned)(B * ((Lq + 15) / 16)));
  switch (nhead) {
#define AVSEP_AP(NH_) case NH_: hipLaunchKernelGGL((attn_proj_kernel<NH_>), grid, dim3(64 * NH_), 0, s, p); break;
    AVSEP_AP(1) AVSEP_AP(2) AVSEP_AP(3) AVSEP_AP(4) AVSEP_AP(5) AVSEP_AP(6) AVSEP_AP(7) AVSEP_AP(8)
#undef AVSEP_AP
    default: return hipErrorNotSupported;
  }
  return hipGetLastError();
#endif
}
