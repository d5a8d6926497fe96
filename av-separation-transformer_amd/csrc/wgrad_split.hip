// wgrad_split.hip -- split-precision weight gradient (round 4, training only):  dW[n][k] = sum_r dY[r][n] X[r][k]  straight from the
// row-major activations, the products on the bf16 matrix pipe (three bf16 terms per fp32 operand, six products per fp32 product,
// fp32 accumulation: gemm_split.hip's scheme; fp32-equivalent results).
//
// The contracted index r is the SLOW index of both operands, which is what made the fp32 kernel (gemm.hip, wgrad_kernel) scatter
// every float4 it loads into four ds_write_b32.  On the bf16 pipe the transposition is free: both operands are split on their way
// to LDS and stored ROW-MAJOR as they sit in memory ([r][64 columns] bf16 planes, one ds_write_b64 per float4 and plane), and the
// MFMA operands -- eight consecutive r per lane for one output row -- come out of gfx950's transposing LDS read
// (ds_read_b64_tr_b16: a 16-lane group fetches 4 rows x 16 columns and receives them column-major), the recipe of
// attention_split.hip's V operand, here for both sides.  The contraction index of a 32-row chunk is walked as
// j = 8g + e <-> row 4g + e (e < 4), row 16 + 4g + e - 4 (e >= 4) on both operands alike.
//
// 64 x 64 tile of dW, 256 threads = 2 x 2 waves of 32 x 32, chunks of 32 rows, two 24 KB LDS buffers, the r range cut into
// gridDim.y slices summed by the caller in a fixed order (avsep_op_wgrad_*: the fp32 kernel's plan).  Optionally the bias
// gradient db[n] = sum_r dY[r][n] rides along exactly as in the fp32 kernel (fp32 sums of the staged dY chunks, same order, same bits).
#include "kernels.h"
#include "gemm_tile.h"
#include "split_terms.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct WgradSplitParams {
  const float* dy;   // [R][ldy]
  const float* x;    // [R][ldx]
  float* dw;         // [N][K] (+ [N] bias gradients) per slice
  int R, N, K, ldy, ldx;
  int rchunk;        // rows per slice (multiple of 32), gridDim.y slices
  int bias;
};

constexpr int WS_PL = 32 * 128;          // bytes of one plane: 32 r-rows x 64 columns bf16
constexpr int WS_BUF = 6 * WS_PL;        // [dY hi|mid|lo][X hi|mid|lo] = 24 KB

__global__ __launch_bounds__(256, 3) void wgrad_split_kernel(const WgradSplitParams p) {
  constexpr int BM = 64, BN = 64, BK = 32;
  __shared__ __attribute__((aligned(16))) char lds[2 * WS_BUF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nbn = (p.K + BN - 1) / BN;
  const int bm = blockIdx.x / nbn, bn = blockIdx.x - bm * nbn;
  const int m0 = bm * BM, n0 = bn * BN;
  const int r_begin = blockIdx.y * p.rchunk;
  const int r_end = min(p.R, r_begin + p.rchunk);
  const int nk = (r_end - r_begin + BK - 1) / BK;
  float* out = p.dw + (size_t)blockIdx.y * ((size_t)p.N * p.K + (p.bias ? p.N : 0));
  const bool do_bias = p.bias && bn == 0;            // block-uniform

  // staging: float4 l of this thread = r-row kr, float4 column c4 of both operand tiles (the fp32 kernel's map)
  int kr[2], a_col[2], b_col[2], st[2];
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    const int idx = tid + 256 * l, c4 = idx & 15;
    kr[l] = idx >> 4;
    a_col[l] = min(m0 + 4 * c4, p.N - 4);            // clamp: duplicated columns are never stored
    b_col[l] = min(n0 + 4 * c4, p.K - 4);
    st[l] = kr[l] * 128 + (((c4 >> 2) ^ ((kr[l] >> 1) & 3)) << 5) + ((c4 & 3) << 3);
  }
  f32x4 ra[2], rb[2], bsum[2];
#pragma unroll
  for (int l = 0; l < 2; ++l) bsum[l] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto load_chunk = [&](int kc) {
    const int r0 = r_begin + min(kc, max(nk - 1, 0)) * BK;          // beyond the range: a chunk nobody multiplies
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      const int r = r0 + kr[l];
      const f32x4 va = *reinterpret_cast<const f32x4*>(p.dy + (size_t)min(r, p.R - 1) * p.ldy + a_col[l]);
      const f32x4 vb = *reinterpret_cast<const f32x4*>(p.x + (size_t)min(r, p.R - 1) * p.ldx + b_col[l]);
      const bool in = r < r_end && kc < nk;
      ra[l] = in ? va : f32x4{0.f, 0.f, 0.f, 0.f};
      rb[l] = in ? vb : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_chunk = [&](int buf) {
    char* base = lds + buf * WS_BUF;
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      if (do_bias) bsum[l] += ra[l];                 // every chunk is stored exactly once; rows >= r_end hold zeros
      unsigned h[2], m[2], lo[2];
      split_pair(f32x2{ra[l][0], ra[l][1]}, h[0], m[0], lo[0]);
      split_pair(f32x2{ra[l][2], ra[l][3]}, h[1], m[1], lo[1]);
      *reinterpret_cast<u32x2*>(base + st[l]) = u32x2{h[0], h[1]};
      *reinterpret_cast<u32x2*>(base + WS_PL + st[l]) = u32x2{m[0], m[1]};
      *reinterpret_cast<u32x2*>(base + 2 * WS_PL + st[l]) = u32x2{lo[0], lo[1]};
      split_pair(f32x2{rb[l][0], rb[l][1]}, h[0], m[0], lo[0]);
      split_pair(f32x2{rb[l][2], rb[l][3]}, h[1], m[1], lo[1]);
      *reinterpret_cast<u32x2*>(base + 3 * WS_PL + st[l]) = u32x2{h[0], h[1]};
      *reinterpret_cast<u32x2*>(base + 4 * WS_PL + st[l]) = u32x2{m[0], m[1]};
      *reinterpret_cast<u32x2*>(base + 5 * WS_PL + st[l]) = u32x2{lo[0], lo[1]};
    }
  };

  // transposing reads: lane 4 q + pp of the 16-lane group g supplies row 16 half + 4 g + q, columns 16 blk + 4 pp .. + 3 and
  // receives rows 16 half + 4 g .. + 3 of column 16 blk + (lane & 15)
  const int fr = lane & 15, fq = lane >> 4;
  int tro[2];
  {
    const int q = fr >> 2, pp = fr & 3;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int row = 16 * half + 4 * fq + q;
      tro[half] = row * 128 + (pp << 3) + (((row >> 1) & 3) << 5);    // the column chunk is XORed in at the read: (blk ^ sw) << 5
    }
  }
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
  auto frag = [&](const char* plane_base, int blk) -> bf16x8 {
    const s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(plane_base + (tro[0] ^ (blk << 5))));
    const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(plane_base + (tro[1] ^ (blk << 5))));
    const u32x2 d0 = __builtin_bit_cast(u32x2, r0), d1 = __builtin_bit_cast(u32x2, r1);
    return __builtin_bit_cast(bf16x8, u32x4{d0[0], d0[1], d1[0], d1[1]});
  };

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nk > 0) {
    load_chunk(0);
    store_chunk(0);
    load_chunk(1);
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
      const char* rbuf = lds + (kc & 1) * WS_BUF;
      // A-side planes (dY: output rows n), B-side planes (X: output columns k); D^T orientation like gemm_kernel's, so that a lane
      // holds four consecutive k of one n: acc[i][j] = mfma(X fragment j, dY fragment i)
      bf16x8 yh[2], ym[2], xh[2], xm[2], tl[2];
#define WS_MMA(XP, YP)                                                                              \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)       \
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(XP[j], YP[i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) yh[i] = frag(rbuf, wm * 2 + i);
#pragma unroll
      for (int j = 0; j < 2; ++j) tl[j] = frag(rbuf + 5 * WS_PL, wn * 2 + j);
      WS_MMA(tl, yh)                                   // (dY hi, X lo)
#pragma unroll
      for (int j = 0; j < 2; ++j) xh[j] = frag(rbuf + 3 * WS_PL, wn * 2 + j);
#pragma unroll
      for (int i = 0; i < 2; ++i) tl[i] = frag(rbuf + 2 * WS_PL, wm * 2 + i);
      WS_MMA(xh, tl)                                   // (dY lo, X hi)
#pragma unroll
      for (int i = 0; i < 2; ++i) ym[i] = frag(rbuf + WS_PL, wm * 2 + i);
#pragma unroll
      for (int j = 0; j < 2; ++j) xm[j] = frag(rbuf + 4 * WS_PL, wn * 2 + j);
      WS_MMA(xm, ym)                                   // (mid, mid)
      WS_MMA(xh, ym)                                   // (dY mid, X hi)
      WS_MMA(xm, yh)                                   // (dY hi, X mid)
      WS_MMA(xh, yh)                                   // (hi, hi)
#undef WS_MMA
      if (kc + 1 < nk) store_chunk((kc & 1) ^ 1);      // block-uniform; the registers hold chunk kc + 1
      load_chunk(kc + 2);
      __syncthreads();
    }
  }
  if (do_bias) {
    // thread (kr, c4) holds the sum over its chunks' rows kr of columns 4 c4 .. + 3: the 32 kr partials of a column are summed
    // through LDS (free after the loop's last barrier) in a fixed order -- the fp32 kernel's reduction, the same bits
    float* red = reinterpret_cast<float*>(lds);        // [BK][BM]
#pragma unroll
    for (int l = 0; l < 2; ++l) *reinterpret_cast<f32x4*>(red + 4 * (tid + 256 * l)) = bsum[l];
    __syncthreads();
    if (tid < BM && m0 + tid < p.N) {                  // N % 4 == 0: an in-range column was never a clamped duplicate
      float t = 0.0f;
#pragma unroll
      for (int r = 0; r < BK; ++r) t += red[r * BM + tid];
      out[(size_t)p.N * p.K + m0 + tid] = t;
    }
  }
  GemmParams q{};   // float4 rows (K % 4 == 0 is a precondition of this kernel)
  q.C = out; q.M = p.N; q.N = p.K; q.ldc = p.K; q.act = ACT_NONE;
  q.W = p.x;   // the epilogue reads (and discards) K floats from W when there is no bias: a readable buffer, never the null page
  gemm_epilogue<2, 2>(q, acc, m0, n0, wm * 32, wn * 32, fr, fq);
}

}  // namespace

// The split-precision form of launch_wgrad for the problems that take the fp32 kernel's 64 x 64 tile (same slices, same output
// layout: `out` holds `slices` partial results of N*K (+ N) floats which the caller sums in slice order).
hipError_t launch_wgrad_split(const float* dy, int ldy, const float* x, int ldx, float* out, int N, int K, int R, int slices,
                              bool with_bias, hipStream_t s) {
  if (N <= 0 || K <= 0 || R <= 0 || (N & 3) || (K & 3) || (ldy & 3) || (ldx & 3) || slices < 1) return hipErrorInvalidValue;
  WgradSplitParams p{dy, x, out, R, N, K, ldy, ldx, 0, with_bias ? 1 : 0};
  p.rchunk = slices > 1 ? (((R + slices - 1) / slices + 31) / 32 * 32) : ((R + 31) / 32 * 32);
  const long tiles = (long)((N + 63) / 64) * ((K + 63) / 64);
  hipLaunchKernelGGL(wgrad_split_kernel, dim3((unsigned)tiles, slices), dim3(256), 0, s, p);
  return hipGetLastError();
}
