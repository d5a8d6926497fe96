// attention_bwd.hip -- backward of softmax(q k^T) v (training path), fp32 on the matrix
// cores, scores recomputed from q, k and the forward's log-sum-exp (nothing of size Lq x Lk is stored).
//
//   kernel A (per 16 queries):  D = rowsum(dO * O);  P^T = exp(S^T - lse);  dP^T = V dO^T;
//                               dS^T = P^T (dP^T - D);  dQ = qscale * dS K           (also writes D)
//   kernel B (per 16 keys):     loops over query tiles:  P, dP, dS as above (untransposed tiles),
//                               dV = P^T dO,  dK = dS^T (qscale q)
// Both use the operand trick of attention.hip: the 16x16 score tile's accumulator registers are directly the
// k-step operands of the following product, so nothing moves between lanes or through LDS.
#include "kernels.h"

namespace {

__device__ __forceinline__ float exp_arg(float x) {   // exp(x) with the compensated argument of attention.hip
  const float L2E_HI = 1.44269502162933349609f, L2E_LO = 1.92596299112661746e-08f;
  x = fmaxf(x, -120.0f);
  const float t = x * L2E_HI;
  float r = fmaf(x, L2E_HI, -t);
  r = fmaf(x, L2E_LO, r);
  const float e = __builtin_amdgcn_exp2f(t);
  return fmaf(e, r * 0.69314718055994530942f, e);
}

// rows fragment: lane (c = l&15, g = l>>4) -> float4 of row (row0 + c) at columns 16s + 4g  (A or B operand of a
// product contracting over the head dim)
template <int NB>
__device__ __forceinline__ void load_rows(const float* base, int ld, int row, float scale, f32x4 (&f)[NB], int g) {
#pragma unroll
  for (int s = 0; s < NB; ++s) f[s] = *reinterpret_cast<const f32x4*>(base + (size_t)row * ld + 16 * s + 4 * g) * scale;
}
// k-step fragment: lane (c, g) -> NB contiguous floats of row (row0 + 4g + r) at columns NB*c  (operand of a product
// contracting over the 16 rows of the tile; output rows map to columns NB*i + blk)
template <int NB>
__device__ __forceinline__ void load_ksteps(const float* base, int ld, int row0, int nrows, float scale,
                                            float (&f)[4][NB], int c, int g) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = min(row0 + 4 * g + r, nrows - 1);
#pragma unroll
    for (int e = 0; e < NB; ++e) f[r][e] = base[(size_t)row * ld + NB * c + e] * scale;
  }
}

template <int NB>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                          int ldk, const float* __restrict__ v, int ldv,
                                                          const float* __restrict__ o, int ldo,
                                                          const float* __restrict__ dO, int lddo,
                                                          const float* __restrict__ lse, float* __restrict__ dvec,
                                                          float* __restrict__ dq, int lddq, int nhead, int Lq, int Lk,
                                                          int nqt, float qscale, float drop_p,
                                                          unsigned long long drop_seed) {
  constexpr int DH = 16 * NB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int wg_per_head = (nqt + 3) >> 2;
  const int bh = blockIdx.x / wg_per_head;
  const int qt = (blockIdx.x - bh * wg_per_head) * 4 + wave;
  if (qt >= nqt) return;
  const int b = bh / nhead, h = bh - b * nhead;
  const float* qb = q + (size_t)b * Lq * ldq + h * DH;
  const float* kb = k + (size_t)b * Lk * ldk + h * DH;
  const float* vb = v + (size_t)b * Lk * ldv + h * DH;
  const float* ob = o + (size_t)b * Lq * ldo + h * DH;
  const float* dob = dO + (size_t)b * Lq * lddo + h * DH;
  float* dqb = dq + (size_t)b * Lq * lddq + h * DH;

  const int qrow = min(qt * 16 + c, Lq - 1);
  f32x4 qf[NB], df[NB], of[NB];
  load_rows<NB>(qb, ldq, qrow, qscale, qf, g);
  load_rows<NB>(dob, lddo, qrow, 1.0f, df, g);
  load_rows<NB>(ob, ldo, qrow, 1.0f, of, g);
  float D = 0.f;
#pragma unroll
  for (int s = 0; s < NB; ++s)
#pragma unroll
    for (int e = 0; e < 4; ++e) D = fmaf(df[s][e], of[s][e], D);
  D += __shfl_xor(D, 16);
  D += __shfl_xor(D, 32);
  const float l = lse[(size_t)bh * Lq + qrow];
  if (g == 0 && qt * 16 + c < Lq) dvec[(size_t)bh * Lq + qt * 16 + c] = D;

  f32x4 acc[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nkt = (Lk + 15) >> 4;
  // A wave owns ONE query tile, so the grid is only ~2 waves per SIMD at the training batch and nothing but the wave
  // itself can hide its loads: the operands of key tile kt+1 are fetched (unconditionally, clamped index) into a second
  // register set before the MFMAs of tile kt.  (First version: load, wait a full memory round trip, compute, per tile:
  // 52 TFLOP/s.)  Same arithmetic per tile, same order over tiles.
  struct KTile {
    f32x4 kf[NB], vf[NB];
    float kk[4][NB];
  };
  auto fetch = [&](int kt, KTile& t) {
    const int krow = min(kt * 16 + c, Lk - 1);
    load_rows<NB>(kb, ldk, krow, 1.0f, t.kf, g);
    load_rows<NB>(vb, ldv, krow, 1.0f, t.vf, g);
    load_ksteps<NB>(kb, ldk, kt * 16, Lk, 1.0f, t.kk, c, g);
  };
  auto compute = [&](int kt, const KTile& t) {
    f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NB; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        st = __builtin_amdgcn_mfma_f32_16x16x4f32(t.kf[s][j], qf[s][j], st, 0, 0, 0);    // S^T[key][q]
        dp = __builtin_amdgcn_mfma_f32_16x16x4f32(t.vf[s][j], df[s][j], dp, 0, 0, 0);    // dP^T[key][q]
      }
    float ds[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = kt * 16 + 4 * g + r;
      const float pr = key < Lk ? exp_arg(st[r] - l) : 0.0f;
      float dpe = dp[r];
      if (drop_p > 0.0f) {   // d/dP of the dropped-and-rescaled probabilities: same mask as the forward
        const unsigned long long e = ((unsigned long long)bh * Lq + (qt * 16 + c)) * Lk + key;
        dpe = dropout_keep(drop_seed, e, drop_p) ? dpe * dropout_scale(drop_p) : 0.0f;
      }
      ds[r] = pr * (dpe - D);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int blk = 0; blk < NB; ++blk)
        acc[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(t.kk[r][blk], ds[r], acc[blk], 0, 0, 0);   // dQ^T += K^T dS^T
  };
  KTile t0, t1;
  fetch(0, t0);
  int kt = 0;
  for (; kt + 1 < nkt; kt += 2) {
    fetch(kt + 1, t1);
    compute(kt, t0);
    fetch(min(kt + 2, nkt - 1), t0);
    compute(kt + 1, t1);
  }
  if (kt < nkt) compute(kt, t0);
  const int qo = qt * 16 + c;
  if (qo < Lq) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int blk = 0; blk < NB; ++blk) dqb[(size_t)qo * lddq + NB * (4 * g + r) + blk] = acc[blk][r] * qscale;
  }
}

template <int NB>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                           int ldk, const float* __restrict__ v, int ldv,
                                                           const float* __restrict__ dO, int lddo,
                                                           const float* __restrict__ lse, const float* __restrict__ dvec,
                                                           float* __restrict__ dk, int lddk, float* __restrict__ dv,
                                                           int lddv, int nhead, int Lq, int Lk, int nkt, float qscale,
                                                           float drop_p, unsigned long long drop_seed) {
  constexpr int DH = 16 * NB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int wg_per_head = (nkt + 3) >> 2;
  const int bh = blockIdx.x / wg_per_head;
  const int kt = (blockIdx.x - bh * wg_per_head) * 4 + wave;
  if (kt >= nkt) return;
  const int b = bh / nhead, h = bh - b * nhead;
  const float* qb = q + (size_t)b * Lq * ldq + h * DH;
  const float* kb = k + (size_t)b * Lk * ldk + h * DH;
  const float* vb = v + (size_t)b * Lk * ldv + h * DH;
  const float* dob = dO + (size_t)b * Lq * lddo + h * DH;

  const int krow = min(kt * 16 + c, Lk - 1);
  const bool kvalid = kt * 16 + c < Lk;
  f32x4 kf[NB], vf[NB];                       // B operands: K^T / V^T with the key on the lane
  load_rows<NB>(kb, ldk, krow, 1.0f, kf, g);
  load_rows<NB>(vb, ldv, krow, 1.0f, vf, g);
  f32x4 acc_k[NB], acc_v[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) acc_k[i] = acc_v[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nqt = (Lq + 15) >> 4;
  // one key tile per wave, ~2 waves per SIMD: the operands of query tile qt+1 (and its lse / D values) are fetched into a
  // second register set before the MFMAs of tile qt (see attn_bwd_dq_kernel)
  // (only the row fragments and the per-row scalars are double-buffered: with the k-step fragments as well the kernel
  // needs 336 registers; those are fetched at the top of the tile -- the same rows again, L2-hot -- and are first used
  // after the 32 score MFMAs)
  struct QTile {
    f32x4 qf[NB], df[NB];
    float l[4], D[4];
  };
  auto fetch = [&](int qt, QTile& t) {
    const int qrow = min(qt * 16 + c, Lq - 1);
    load_rows<NB>(qb, ldq, qrow, qscale, t.qf, g);
    load_rows<NB>(dob, lddo, qrow, 1.0f, t.df, g);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qc = min(qt * 16 + 4 * g + r, Lq - 1);
      t.l[r] = lse[(size_t)bh * Lq + qc];
      t.D[r] = dvec[(size_t)bh * Lq + qc];
    }
  };
  auto compute = [&](int qt, const QTile& t) {
    float qk[4][NB], dk_[4][NB];
    load_ksteps<NB>(qb, ldq, qt * 16, Lq, qscale, qk, c, g);
    load_ksteps<NB>(dob, lddo, qt * 16, Lq, 1.0f, dk_, c, g);
    f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NB; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        st = __builtin_amdgcn_mfma_f32_16x16x4f32(t.qf[s][j], kf[s][j], st, 0, 0, 0);    // S[q][key]
        dp = __builtin_amdgcn_mfma_f32_16x16x4f32(t.df[s][j], vf[s][j], dp, 0, 0, 0);    // dP[q][key]
      }
    float pr[4], ds[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qi = qt * 16 + 4 * g + r;
      const bool ok = kvalid && qi < Lq;
      const int qc = min(qi, Lq - 1);
      const float l = t.l[r];
      const float D = t.D[r];
      pr[r] = ok ? exp_arg(st[r] - l) : 0.0f;
      float dpe = dp[r];
      if (drop_p > 0.0f) {
        const unsigned long long e = ((unsigned long long)bh * Lq + qc) * Lk + (kt * 16 + c);
        const bool keep = dropout_keep(drop_seed, e, drop_p);
        const float ks = 1.0f / (1.0f - drop_p);
        dpe = keep ? dpe * ks : 0.0f;
        ds[r] = pr[r] * (dpe - D);
        pr[r] = keep ? pr[r] * ks : 0.0f;        // dV uses the dropped probabilities
      } else {
        ds[r] = pr[r] * (dpe - D);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int blk = 0; blk < NB; ++blk) {
        acc_v[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(dk_[r][blk], pr[r], acc_v[blk], 0, 0, 0);   // dV^T += dO^T P
        acc_k[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(qk[r][blk], ds[r], acc_k[blk], 0, 0, 0);    // dK^T += Q^T dS
      }
  };
  QTile t0, t1;
  fetch(0, t0);
  int qt = 0;
  for (; qt + 1 < nqt; qt += 2) {
    fetch(qt + 1, t1);
    compute(qt, t0);
    fetch(min(qt + 2, nqt - 1), t0);
    compute(qt + 1, t1);
  }
  if (qt < nqt) compute(qt, t0);
  if (kvalid) {
    float* dkb = dk + (size_t)b * Lk * lddk + h * DH;
    float* dvb = dv + (size_t)b * Lk * lddv + h * DH;
    const int ko = kt * 16 + c;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int blk = 0; blk < NB; ++blk) {
        dkb[(size_t)ko * lddk + NB * (4 * g + r) + blk] = acc_k[blk][r];
        dvb[(size_t)ko * lddv + NB * (4 * g + r) + blk] = acc_v[blk][r];
      }
  }
}

}  // namespace

// dvec: B*nhead*Lq floats of scratch (rowsum(dO*O)), written by the dQ kernel and read by the dK/dV kernel.
hipError_t launch_attention_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                const float* o, int ldo, const float* dO, int lddo, const float* lse, float* dvec,
                                float* dq, int lddq, float* dk, int lddk, float* dv, int lddv, int B, int nhead, int dh,
                                int Lq, int Lk, float qscale, float drop_p, unsigned long long drop_seed, hipStream_t s) {
  if (B <= 0 || nhead <= 0 || Lq <= 0 || Lk <= 0) return hipErrorInvalidValue;
  if (dh <= 0 || dh > 128 || (dh & 15)) return hipErrorInvalidValue;   // training path: head dim multiple of 16
  if ((ldq | ldk | ldv | ldo | lddo) & 3) return hipErrorInvalidValue;
  const int nqt = (Lq + 15) / 16, nkt = (Lk + 15) / 16;
  const dim3 gq((unsigned)(B * nhead * ((nqt + 3) / 4))), gk((unsigned)(B * nhead * ((nkt + 3) / 4))), block(256);
#define AVSEP_ATTB(NB_)                                                                                              \
  if (dh == 16 * NB_) {                                                                                              \
    hipLaunchKernelGGL((attn_bwd_dq_kernel<NB_>), gq, block, 0, s, q, ldq, k, ldk, v, ldv, o, ldo, dO, lddo, lse,    \
                       dvec, dq, lddq, nhead, Lq, Lk, nqt, qscale, drop_p, drop_seed);                               \
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<NB_>), gk, block, 0, s, q, ldq, k, ldk, v, ldv, dO, lddo, lse, dvec, dk, \
                       lddk, dv, lddv, nhead, Lq, Lk, nkt, qscale, drop_p, drop_seed);                               \
  }
  AVSEP_ATTB(1) AVSEP_ATTB(2) AVSEP_ATTB(3) AVSEP_ATTB(4) AVSEP_ATTB(5) AVSEP_ATTB(6) AVSEP_ATTB(7) AVSEP_ATTB(8)
#undef AVSEP_ATTB
  return hipGetLastError();
}
