// gemm_h2.hip -- split-precision GEMM, fourth generation (round 5): TWO fp16 terms per operand, THREE products per fp32 product.
//
// Where the round-4 / round-5 kernels stand (gemm_split.hip, gemm_planes.hip): six bf16 MFMA products per fp32 product, the matrix
// pipe 82-86 % busy inside the loop -- at the 1.64-1.70 GHz the chip holds under that load (profiles/r05_planes_stamps2.txt: in-kernel
// s_memtime against the 100 MHz wall clock).  The kernel is at the chip's POWER limit, not short of issue slots: what is left is
// fewer matrix instructions per result.
//
// fp16 carries 11 significant bits where bf16 carries 8: TWO fp16 terms hold 22 bits of an fp32 value (hi = rn16(x'), lo = rn16(x' - hi);
// the third bf16 term exists to reach 24), and x w needs THREE products -- (hi, lo), (lo, hi), (hi, hi), each exact in fp32; the dropped
// (lo, lo) and the representation's remainder are < 2^-21 relative -- instead of six: half the matrix time, half the LDS and
// global bytes per operand (4 B per element: as many as the fp32 tensor itself).  What fp16 lacks is RANGE (6e-5 ... 65504 normal),
// so every operand is stored scaled by a power of two (exact) that puts it under 2^14:
//   * a weight row w[n][:] by 2^ew[n], ew[n] from the row's largest magnitude (computed once by avsep_finalize_weights);
//   * an activation tensor by ONE static exponent eA per site, from a rigorous bound that only needs the weights: a LayerNorm output
//     is at most sqrt(d - 1) |gamma_k| + |beta_k| whatever its input; act(x W^T + b) at most ||x||_2 ||w_n||_2 + |b_n| with ||LayerNorm(x)||_2
//     <= sqrt(d) max|gamma| + ||beta||_2 (|relu(z)|, |gelu(z)| <= |z|); a self-attention output (a convex combination of value rows) at
//     most the bound of its value projection.  No runtime reduction, no scale tensors; the bounds overshoot typical values by 2^3-2^6,
//     which costs nothing: an element keeps all 22 bits while its scaled magnitude is >= 2^-3, i.e. down to 2^-17 of the bound, and
//     an absolute 2^-25 of the bound below that.
// The GEMM accumulates sum_k x'_k w'_k in fp32 and its epilogue multiplies by cscale[n] = 2^-(eA + ew[n]) (exact) before bias /
// activation / residual.  Error against float64 on model-like operands: at or below the fp32 MFMA GEMM's own (CPU emulation
// profiles/r05_h2_emulation.txt; GPU: tests/test_gpu_parity.py::test_op_linear_h2), every golden of the d_model = 512 configurations
// inside MASK_TOL.  It is a NORMWISE-accurate scheme: an element 2^-40 of its tensor's bound is lost, where the three-term bf16 split
// (elementwise exact, kept for the operands that have no static bound) would keep it -- a dot product's error is bounded by
// K max|x| max|w| 2^-22 either way.
//
// PLANE FORMAT "H2" of a matrix X[M][K] (K % 32 == 0) in a buffer of `rows` rows: fp16 P[K/32][2][rows][32] -- gemm_planes.hip's layout
// with two terms.  Same LDS image, swizzle and DMA staging as gemm_planes_kernel_b.
#include "kernels.h"
#include "gemm_tile.h"
#include "split_terms.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// ---- fp32 -> H2 planes, row r scaled by 2^ex[r] (ex null: one exponent e for the whole matrix) ----------------------------------------
__global__ __launch_bounds__(256) void split_h2_kernel(const float* __restrict__ x, int ld, unsigned short* __restrict__ P,
                                                       long long rows, int M, int K, const int* __restrict__ ex, int e) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int spr = K >> 3;
  if (idx >= (size_t)M * spr) return;
  const int m = (int)(idx / spr), s = (int)(idx - (size_t)m * spr);
  const float sc = ldexpf(1.0f, ex ? ex[m] : e);
  const float* src = x + (size_t)m * ld + 8 * s;
  const f32x4 v0 = *reinterpret_cast<const f32x4*>(src) * sc, v1 = *reinterpret_cast<const f32x4*>(src + 4) * sc;
  unsigned h[4], l[4];
  split_pair_h2(f32x2{v0[0], v0[1]}, h[0], l[0]);
  split_pair_h2(f32x2{v0[2], v0[3]}, h[1], l[1]);
  split_pair_h2(f32x2{v1[0], v1[1]}, h[2], l[2]);
  split_pair_h2(f32x2{v1[2], v1[3]}, h[3], l[3]);
  char* base = reinterpret_cast<char*>(P) + (((size_t)(s >> 2) * 2) * rows + m) * 64 + (s & 3) * 16;
  *reinterpret_cast<u32x4*>(base) = u32x4{h[0], h[1], h[2], h[3]};
  *reinterpret_cast<u32x4*>(base + (size_t)rows * 64) = u32x4{l[0], l[1], l[2], l[3]};
}

// per weight row n: ew[n] = the exponent that puts max|w[n][:]| into (2^13, 2^14] (0 for a zero row), l2[n] = ||w[n][:]||_2 rounded up
__global__ __launch_bounds__(256) void h2_row_stats_kernel(const float* __restrict__ w, int K, int* __restrict__ ew, float* __restrict__ l2) {
  const int n = blockIdx.x;
  float mx = 0.0f;
  double sq = 0.0;
  for (int k = threadIdx.x; k < K; k += 256) {
    const float v = w[(size_t)n * K + k];
    mx = fmaxf(mx, fabsf(v));
    sq += (double)v * (double)v;
  }
  __shared__ float rm[256];
  __shared__ double rs[256];
  rm[threadIdx.x] = mx;
  rs[threadIdx.x] = sq;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) {
      rm[threadIdx.x] = fmaxf(rm[threadIdx.x], rm[threadIdx.x + off]);
      rs[threadIdx.x] += rs[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    int e = 0;
    if (rm[0] > 0.0f && rm[0] < 3.0e38f) {
      int ex;
      (void)frexpf(rm[0], &ex);                                    // rm = f * 2^ex, f in [0.5, 1): rm * 2^(14 - ex) in [2^13, 2^14)
      e = 14 - ex;
      e = e > 100 ? 100 : e < -100 ? -100 : e;
    }
    ew[n] = e;
    l2[n] = (float)(sqrt(rs[0]) * (1.0 + 1e-6));
  }
}

// ---- plane-output epilogue (H2) -------------------------------------------------------------------------------------------------
// y = act(acc * cscale + bias), written as the two fp16 terms of the NEXT GEMM's A operand, scaled by p.cp_scale (that operand's static
// power of two).  gemm_planes.hip's epilogue: v_permlane16_swap_b32 turns the lane's 4 + 4 columns of two 16-column blocks into 8
// consecutive ones, one 16-byte slot per term and (row block, chunk).
template <int ACT, int WBN>
__device__ __forceinline__ void h2_planes_epilogue_t(const GemmParams& p, const f32x4 (&acc)[4][WBN], int mbase, int nbase, int fr,
                                                     int fq) {
  const bool has_b = p.bias != nullptr;                                     // block-uniform
  const float* bsrc = has_b ? p.bias : p.cscale;
  const int slot = ((fq & 1) << 1) | (fq >> 1);
  const size_t ts = (size_t)p.c_rows * 64;
  const float ps = p.cp_scale;
#pragma unroll
  for (int t = 0; t < WBN / 2; ++t) {
    const int n32 = nbase + 32 * t;                                        // first column of the chunk (block-uniform validity)
    f32x4 b0, b1, s0, s1;
    {
      const int c0 = min(n32 + 4 * fq, p.N - 4), c1 = min(n32 + 16 + 4 * fq, p.N - 4);
      b0 = *reinterpret_cast<const f32x4*>(bsrc + c0); b1 = *reinterpret_cast<const f32x4*>(bsrc + c1);
      s0 = *reinterpret_cast<const f32x4*>(p.cscale + c0); s1 = *reinterpret_cast<const f32x4*>(p.cscale + c1);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = mbase + 16 * i + fr;
      f32x4 x0, x1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        x0[e] = acc[i][2 * t][e] * s0[e];
        x1[e] = acc[i][2 * t + 1][e] * s1[e];
        if (has_b) { x0[e] += b0[e]; x1[e] += b1[e]; }
      }
      if (ACT == ACT_GELU) {
        x0 = act4_outofline<ACT>(x0);
        x1 = act4_outofline<ACT>(x1);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { x0[e] = act_t<ACT>(x0[e]); x1[e] = act_t<ACT>(x1[e]); }
      }
      float lo4[4], hi4[4];                                                // this lane's 8 consecutive columns, scaled for the consumer
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x0[e] * ps), __float_as_uint(x1[e] * ps), false, false);
        lo4[e] = __uint_as_float(r[0]);
        hi4[e] = __uint_as_float(r[1]);
      }
      unsigned h[4], l[4];
      split_pair_h2(f32x2{lo4[0], lo4[1]}, h[0], l[0]);
      split_pair_h2(f32x2{lo4[2], lo4[3]}, h[1], l[1]);
      split_pair_h2(f32x2{hi4[0], hi4[1]}, h[2], l[2]);
      split_pair_h2(f32x2{hi4[2], hi4[3]}, h[3], l[3]);
      if (m < p.M && n32 < p.N) {                                          // N % 32 == 0 (launcher)
        char* dst = reinterpret_cast<char*>(p.Cp) + (((size_t)(n32 >> 5) * 2) * p.c_rows + m) * 64 + slot * 16;
        *reinterpret_cast<u32x4*>(dst) = u32x4{h[0], h[1], h[2], h[3]};
        *reinterpret_cast<u32x4*>(dst + ts) = u32x4{l[0], l[1], l[2], l[3]};
      }
    }
  }
}

template <int WBN>
__device__ __forceinline__ void h2_planes_epilogue(const GemmParams& p, const f32x4 (&acc)[4][WBN], int mbase, int nbase, int fr, int fq) {
  switch (p.act) {                                // block-uniform
    case ACT_RELU: h2_planes_epilogue_t<ACT_RELU, WBN>(p, acc, mbase, nbase, fr, fq); break;
    case ACT_GELU: h2_planes_epilogue_t<ACT_GELU, WBN>(p, acc, mbase, nbase, fr, fq); break;
    default: h2_planes_epilogue_t<ACT_NONE, WBN>(p, acc, mbase, nbase, fr, fq); break;
  }
}

// fp32 output: the accumulators are brought back to true scale (acc * cscale[n], exact) and handed to gemm_tile.h's epilogues
// (bias / activation / residual, or the mask head's two outputs) unchanged
template <int WBM, int WBN>
__device__ __forceinline__ void h2_descale_rows(const GemmParams& p, f32x4 (&acc)[WBM][WBN], int mbase, int fr) {
  if (!p.rscale) return;                                                 // block-uniform
#pragma unroll
  for (int i = 0; i < WBM; ++i) {
    const float rs = p.rscale[min(mbase + 16 * i + fr, p.M - 1)];          // lane (fr, fq) holds row 16 i + fr of its blocks
#pragma unroll
    for (int j = 0; j < WBN; ++j) acc[i][j] *= rs;
  }
}

template <int WBN>
__device__ __forceinline__ void h2_descale(const GemmParams& p, f32x4 (&acc)[4][WBN], int nbase, int fq) {
#pragma unroll
  for (int j = 0; j < WBN; ++j) {
    const int n = nbase + 16 * j + 4 * fq;
    f32x4 s;
    if (!(p.N & 3)) {
      s = *reinterpret_cast<const f32x4*>(p.cscale + min(n, p.N - 4));
    } else {                                                             // the mask head: N even, not a multiple of 4
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] = p.cscale[min(n + e, p.N - 1)];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i][j] *= s;
  }
}

// ---- the kernel: 256 x BN x 32 tile (BN = 128: 4 x 2 waves of 64 x 64; BN = 256: 4 x 2 waves of 64 x 128), persistent over tiles ----------
// gemm_planes_kernel_b's structure with two terms: per chunk 16 (24) fragment reads and 48 (96) MFMAs per wave, 6 (8) DMA
// instructions; the chunk's barrier sits behind its FIRST product -- every fragment has been requested by then --, the DMA of the
// chunk after next is issued behind the MFMAs of the other two.
struct H2Tile {
  unsigned a0, a1, w0, w1;   // this lane's byte offsets inside a (chunk, term) slab: row * 64 + logical slot * 16
};

// V (developer ablations, timing only, AVSEP_H2_ABL; profiles/r05_gemm_h2_ablations.txt): 1 = the lo fragments alias the hi ones (half
// the LDS reads), 2 = no DMA inside the loop, 4 = no MFMA, 8 = no epilogue
template <int BN, int V = 0>
__global__ __launch_bounds__(512, 1) void gemm_h2_kernel(const GemmParams pin) {
  GemmParams p = pin;
  constexpr int BM = 256, WBN = BN / 32;
  constexpr int APL = BM * 64, WPL = BN * 64;                              // bytes of one term of one stage
  constexpr int BUF = 2 * APL + 2 * WPL;                                   // 48 KB (BN = 128) / 64 KB (BN = 256): [A hi|lo][W hi|lo]
  extern __shared__ __attribute__((aligned(1024))) char ldsh[];           // 2 x BUF
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nbn = (p.N + BN - 1) / BN;
  const int ntiles = ((p.M + BM - 1) / BM) * nbn;
  int tile, tile_end, tile_step;
  {
    const int G = gridDim.x, b = blockIdx.x;
    if (p.no_xcd_remap) {
      tile = b; tile_end = ntiles; tile_step = G;
    } else {                                                               // an XCD's workgroups share a contiguous tile range
      const int xcd = b & 7, q = ntiles >> 3, r = ntiles & 7;
      const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
      tile = start + (b >> 3);
      tile_end = start + q + (xcd < r ? 1 : 0);
      tile_step = (G >> 3) + (xcd < (G & 7) ? 1 : 0);
    }
  }
  if (tile >= tile_end) return;                                            // block-uniform

  const int lr = lane >> 2;
  const int lslot = ((lane & 3) ^ ((0 - (lane >> 4)) & 3)) << 4;
  auto tile_rows = [&](int t) {
    const int bm = t / nbn, bn = t - bm * nbn;
    H2Tile r;
    r.a0 = (unsigned)min(bm * BM + 16 * wave + lr, p.M - 1) * 64u + lslot;
    r.a1 = (unsigned)min(bm * BM + 16 * (wave + 8) + lr, p.M - 1) * 64u + lslot;
    r.w0 = (unsigned)min(bn * BN + 16 * wave + lr, p.N - 1) * 64u + lslot;
    r.w1 = (unsigned)min(bn * BN + 16 * (wave + 8) + lr, p.N - 1) * 64u + lslot;   // BN = 256 only
    return r;
  };
  const size_t a_ts = (size_t)p.a_rows * 64, w_ts = (size_t)p.w_rows * 64;
  // piece q of chunk kc of tile rows T -> buffer buf.  q = 0 .. 3: A (term q / 2, row block wave + 8 (q % 2)); then W: term (q - 4) / WQ,
  // row block wave + 8 ((q - 4) % WQ) with WQ = BN / 128 row blocks per wave and term
#define H2_PIECE(kc, T, buf, q)                                                                                                  \
  {                                                                                                                              \
    char* lb_ = ldsh + (buf) * BUF;                                                                                              \
    if ((q) < 4) {                                                                                                               \
      const int t_ = (q) >> 1, k_ = (q) & 1;                                                                                     \
      const char* as_ = reinterpret_cast<const char*>(p.Ap) + ((size_t)(kc) * 2 + t_) * a_ts;                                    \
      __builtin_amdgcn_global_load_lds((gptr_t)(as_ + (k_ ? T.a1 : T.a0)), (lptr_t)(lb_ + t_ * APL + (wave + 8 * k_) * 1024), 16, 0, 0); \
    } else {                                                                                                                     \
      constexpr int WQ_ = BN / 128;                                                                                              \
      const int t_ = ((q) - 4) / WQ_, k_ = ((q) - 4) % WQ_;                                                                      \
      const char* ws_ = reinterpret_cast<const char*>(p.Wp) + ((size_t)(kc) * 2 + t_) * w_ts;                                    \
      __builtin_amdgcn_global_load_lds((gptr_t)(ws_ + (k_ ? T.w1 : T.w0)), (lptr_t)(lb_ + 2 * APL + t_ * WPL + (wave + 8 * k_) * 1024), 16, 0, 0); \
    }                                                                                                                            \
  }
  constexpr int NPIECE = 4 + 2 * (BN / 128);                               // 6 / 8 per wave and chunk

  const int fr = lane & 15, fq = lane >> 4;
  int a_fo[4], w_fo[WBN];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wm * 64 + 16 * i + fr;
    a_fo[i] = r * 64 + ((fq ^ ((0 - (r >> 2)) & 3)) << 4);
  }
#pragma unroll
  for (int j = 0; j < WBN; ++j) {
    const int c = wn * (BN / 2) + 16 * j + fr;
    w_fo[j] = 2 * APL + c * 64 + ((fq ^ ((0 - (c >> 2)) & 3)) << 4);
  }
  const int nk = p.K >> 5;

  // the fetch cursor runs two chunks ahead of the products, in one flat chunk sequence over this workgroup's tiles; past the last
  // chunk it keeps fetching the last tile's chunks (valid addresses, into buffers nobody reads any more)
  int f_tile = tile, f_kc = 0;
  H2Tile f_rows = tile_rows(tile);
#define H2_ADVANCE                                   \
  if (++f_kc == nk) {                                \
    f_kc = 0;                                        \
    if (f_tile + tile_step < tile_end) {             \
      f_tile += tile_step;                           \
      f_rows = tile_rows(f_tile);                    \
    }                                                \
  }
#define H2_ALL_PIECES(buf) \
  _Pragma("unroll") for (int q = 0; q < NPIECE; ++q) H2_PIECE(f_kc, f_rows, buf, q)
  H2_ALL_PIECES(0)
  H2_ADVANCE
  H2_ALL_PIECES(1)
  H2_ADVANCE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

#define H2_FRAG_A(term, f) \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) f[i] = *reinterpret_cast<const f16x8*>(rb + (term) * APL + a_fo[i]);
#define H2_FRAG_W(term, f) \
  _Pragma("unroll") for (int j = 0; j < WBN; ++j) f[j] = *reinterpret_cast<const f16x8*>(rb + (term) * WPL + w_fo[j]);
#define H2_FENCE __builtin_amdgcn_sched_barrier(0);
#define H2_MMA(fwp, fap)                                                                                      \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < WBN; ++j)               \
      if constexpr (!(V & 4)) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fwp[j], fap[i], acc[i][j], 0, 0, 0); \
      else acc[i][j][0] += __builtin_bit_cast(f32x4, fwp[j])[0] * __builtin_bit_cast(f32x4, fap[i])[1];
  // a product with DMA pieces q0 .. q0 + NQ - 1 of this step's fetch (-> buffer par), one behind every (4 WBN / NQ)-th MFMA
#define H2_MMA_D(fwp, fap, q0, NQ)                                                                            \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < WBN; ++j) {             \
    if constexpr (!(V & 4)) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fwp[j], fap[i], acc[i][j], 0, 0, 0); \
    else acc[i][j][0] += __builtin_bit_cast(f32x4, fwp[j])[0] * __builtin_bit_cast(f32x4, fap[i])[1];         \
    constexpr int every_ = 4 * WBN / (NQ);                                                                    \
    const int idx_ = i * WBN + j;                                                                             \
    if (!(V & 2) && idx_ % every_ == every_ - 1) { H2_FENCE H2_PIECE(l_kc, l_rows, par, (q0) + idx_ / every_) H2_FENCE }  \
  }

  int par = 0, kc = 0;
  f32x4 acc[4][WBN];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < WBN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (;;) {
    f16x8 a_hi[4], a_lo[4], w_hi[WBN], w_lo[WBN];
    {
      const char* rb = ldsh + par * BUF;
      if constexpr (V & 1) {
        H2_FRAG_A(0, a_hi) H2_FRAG_W(0, w_hi)
#pragma unroll
        for (int i = 0; i < 4; ++i) a_lo[i] = a_hi[i];
#pragma unroll
        for (int j = 0; j < WBN; ++j) w_lo[j] = w_hi[j];
      } else {
        H2_FRAG_A(0, a_hi) H2_FRAG_W(1, w_lo) H2_FRAG_A(1, a_lo) H2_FRAG_W(0, w_hi)
      }
      H2_FENCE
      H2_MMA(w_lo, a_hi) /* (hi, lo) */
      H2_FENCE
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int l_kc = f_kc;                                                 // this step's fetch: the chunk after next -> the buffer just read
    const H2Tile l_rows = f_rows;
    H2_ADVANCE
    H2_MMA_D(w_hi, a_lo, 0, NPIECE / 2) /* (lo, hi) */
    H2_MMA_D(w_hi, a_hi, NPIECE / 2, NPIECE / 2) /* (hi, hi) */
    par ^= 1;
    if (++kc == nk) {                                                      // block-uniform
      const int bm = tile / nbn, bn = tile - bm * nbn;
      if constexpr (V & 8) {                                               // ablation: a never-taken store keeps the accumulators live
        float t_ = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < WBN; ++j) t_ += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (t_ == 12345.678f) p.C[tile] = t_;
      } else {
        if (p.Cp) h2_planes_epilogue<WBN>(p, acc, bm * BM + wm * 64, bn * BN + wn * (BN / 2), fr, fq);
        if (p.C) {
          h2_descale<WBN>(p, acc, bn * BN + wn * (BN / 2), fq);
          h2_descale_rows<4, WBN>(p, acc, bm * BM + wm * 64, fr);
          gemm_epilogue<4, WBN>(p, acc, bm * BM, bn * BN, wm * 64, wn * (BN / 2), fr, fq);
        }
      }
      tile += tile_step;
      if (tile >= tile_end) break;
      kc = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < WBN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // no DMA may outlive the workgroup's LDS
#undef H2_PIECE
#undef H2_ADVANCE
#undef H2_ALL_PIECES
#undef H2_FRAG_A
#undef H2_FRAG_W
#undef H2_FENCE
#undef H2_MMA
#undef H2_MMA_D
}

// ---- the same tile with the fragments one chunk ahead ----------------------------------------------------------------------------
// With half the MFMAs per chunk the barrier, the first fragments' LDS latency behind it and the DMA issue weigh twice as much
// (gemm_h2_kernel: ~57 % of the matrix pipe).  Here a chunk's fragments are ALL in registers before its first MFMA -- requested
// right behind the previous chunk's barrier, into a second register set (2 x 64 fragment + 64 accumulator registers) --, so the
// step is: barrier (everyone's fragments of this chunk have arrived: its buffer is free; everyone's DMA of the next chunk has
// landed), request the next chunk's fragments, then 48 MFMAs from registers with the DMA of the chunk after next between them.
// Reads and DMA both get a whole chunk period.  The two register sets alternate in a loop unrolled by two: K % 64 == 0.
// The last chunk of a tile requests nothing (the epilogue wants the registers); the next tile's first fragments follow the epilogue.
__global__ __launch_bounds__(512, 1) void gemm_h2_kernel2(const GemmParams pin) {
  GemmParams p = pin;
  constexpr int BM = 256, BN = 128, WBN = 4;
  constexpr int APL = BM * 64, WPL = BN * 64;
  constexpr int BUF = 2 * APL + 2 * WPL;                                   // 48 KB
  extern __shared__ __attribute__((aligned(1024))) char ldsh[];           // 2 x BUF
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nbn = (p.N + BN - 1) / BN;
  const int ntiles = ((p.M + BM - 1) / BM) * nbn;
  int tile, tile_end, tile_step;
  {
    const int G = gridDim.x, b = blockIdx.x;
    if (p.no_xcd_remap) {
      tile = b; tile_end = ntiles; tile_step = G;
    } else {
      const int xcd = b & 7, q = ntiles >> 3, r = ntiles & 7;
      const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
      tile = start + (b >> 3);
      tile_end = start + q + (xcd < r ? 1 : 0);
      tile_step = (G >> 3) + (xcd < (G & 7) ? 1 : 0);
    }
  }
  if (tile >= tile_end) return;                                            // block-uniform

  const int lr = lane >> 2;
  const int lslot = ((lane & 3) ^ ((0 - (lane >> 4)) & 3)) << 4;
  auto tile_rows = [&](int t) {
    const int bm = t / nbn, bn = t - bm * nbn;
    H2Tile r;
    r.a0 = (unsigned)min(bm * BM + 16 * wave + lr, p.M - 1) * 64u + lslot;
    r.a1 = (unsigned)min(bm * BM + 16 * (wave + 8) + lr, p.M - 1) * 64u + lslot;
    r.w0 = (unsigned)min(bn * BN + 16 * wave + lr, p.N - 1) * 64u + lslot;
    r.w1 = r.w0;
    return r;
  };
  const size_t a_ts = (size_t)p.a_rows * 64, w_ts = (size_t)p.w_rows * 64;
  // piece q = 0 .. 3: A (term q / 2, row block wave + 8 (q % 2)); 4, 5: W (term q - 4, row block wave)
#define K2_PIECE(kc, T, buf, q)                                                                                                  \
  {                                                                                                                              \
    char* lb_ = ldsh + (buf) * BUF;                                                                                              \
    if ((q) < 4) {                                                                                                               \
      const int t_ = (q) >> 1, k_ = (q) & 1;                                                                                     \
      const char* as_ = reinterpret_cast<const char*>(p.Ap) + ((size_t)(kc) * 2 + t_) * a_ts;                                    \
      __builtin_amdgcn_global_load_lds((gptr_t)(as_ + (k_ ? T.a1 : T.a0)), (lptr_t)(lb_ + t_ * APL + (wave + 8 * k_) * 1024), 16, 0, 0); \
    } else {                                                                                                                     \
      const int t_ = (q) - 4;                                                                                                    \
      const char* ws_ = reinterpret_cast<const char*>(p.Wp) + ((size_t)(kc) * 2 + t_) * w_ts;                                    \
      __builtin_amdgcn_global_load_lds((gptr_t)(ws_ + T.w0), (lptr_t)(lb_ + 2 * APL + t_ * WPL + wave * 1024), 16, 0, 0);        \
    }                                                                                                                            \
  }
  const int fr = lane & 15, fq = lane >> 4;
  int a_fo[4], w_fo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wm * 64 + 16 * i + fr;
    a_fo[i] = r * 64 + ((fq ^ ((0 - (r >> 2)) & 3)) << 4);
    const int c = wn * 64 + 16 * i + fr;
    w_fo[i] = 2 * APL + c * 64 + ((fq ^ ((0 - (c >> 2)) & 3)) << 4);
  }
  const int nk = p.K >> 5;                                                 // even (launcher)

  int f_tile = tile, f_kc = 0;                                             // fetch cursor: two chunks ahead of the products
  H2Tile f_rows = tile_rows(tile);
#define K2_ADVANCE                                   \
  if (++f_kc == nk) {                                \
    f_kc = 0;                                        \
    if (f_tile + tile_step < tile_end) {             \
      f_tile += tile_step;                           \
      f_rows = tile_rows(f_tile);                    \
    }                                                \
  }
#define K2_ALL_PIECES(buf) _Pragma("unroll") for (int q = 0; q < 6; ++q) K2_PIECE(f_kc, f_rows, buf, q)
  K2_ALL_PIECES(0)
  K2_ADVANCE
  K2_ALL_PIECES(1)
  K2_ADVANCE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  struct Frags { f16x8 a_hi[4], a_lo[4], w_hi[4], w_lo[4]; };
#define K2_READ(F, buf)                                                                        \
  {                                                                                            \
    const char* rb_ = ldsh + (buf) * BUF;                                                      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                            \
      F.a_hi[i] = *reinterpret_cast<const f16x8*>(rb_ + a_fo[i]);                              \
      F.w_lo[i] = *reinterpret_cast<const f16x8*>(rb_ + WPL + w_fo[i]);                        \
    }                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                            \
      F.a_lo[i] = *reinterpret_cast<const f16x8*>(rb_ + APL + a_fo[i]);                        \
      F.w_hi[i] = *reinterpret_cast<const f16x8*>(rb_ + w_fo[i]);                              \
    }                                                                                          \
  }
#define K2_FENCE __builtin_amdgcn_sched_barrier(0);
#define K2_MMA(fwp, fap)                                                                                      \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j)                 \
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fwp[j], fap[i], acc[i][j], 0, 0, 0);
  // a product with the DMA pieces q0, q0 + 1, q0 + 2 of this step's fetch (-> buffer buf) behind its MFMAs 5, 10 and 15
#define K2_MMA_D(fwp, fap, buf, q0)                                                                           \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) {               \
    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fwp[j], fap[i], acc[i][j], 0, 0, 0);                   \
    const int idx_ = i * 4 + j;                                                                               \
    if (idx_ % 5 == 4) { K2_FENCE K2_PIECE(l_kc, l_rows, buf, (q0) + idx_ / 5) K2_FENCE }                     \
  }
  // one chunk: its fragments are in CUR (buffer B); NXT receives the next chunk's (buffer B ^ 1) unless `last`
#define K2_STEP(CUR, NXT, B, last)                                                             \
  {                                                                                            \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                           \
    __syncthreads();                                                                           \
    if (!(last)) K2_READ(NXT, (B) ^ 1)                                                         \
    const int l_kc = f_kc;                                                                     \
    const H2Tile l_rows = f_rows;                                                              \
    K2_ADVANCE                                                                                 \
    K2_FENCE                                                                                   \
    K2_MMA(CUR.w_lo, CUR.a_hi)        /* (hi, lo) */                                           \
    K2_MMA_D(CUR.w_hi, CUR.a_lo, B, 0) /* (lo, hi) */                                          \
    K2_MMA_D(CUR.w_hi, CUR.a_hi, B, 3) /* (hi, hi) */                                          \
  }

  Frags f0, f1;
  f32x4 acc[4][4];
  K2_READ(f0, 0)
  for (;;) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kc = 2; kc < nk; kc += 2) {
      K2_STEP(f0, f1, 0, false)
      K2_STEP(f1, f0, 1, false)
    }
    K2_STEP(f0, f1, 0, false)                                             // the tile's last two chunks, peeled: the very last requests nothing
    K2_STEP(f1, f0, 1, true)
    {
      const int bm = tile / nbn, bn = tile - bm * nbn;
      if (p.Cp) h2_planes_epilogue<WBN>(p, acc, bm * BM + wm * 64, bn * BN + wn * 64, fr, fq);
      if (p.C) {
        h2_descale<WBN>(p, acc, bn * BN + wn * 64, fq);
        h2_descale_rows<4, WBN>(p, acc, bm * BM + wm * 64, fr);
        gemm_epilogue<4, WBN>(p, acc, bm * BM, bn * BN, wm * 64, wn * 64, fr, fq);
      }
    }
    tile += tile_step;
    if (tile >= tile_end) break;
    K2_READ(f0, 0)                                                         // certified by the last step's barrier
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // no DMA may outlive the workgroup's LDS
#undef K2_PIECE
#undef K2_ADVANCE
#undef K2_ALL_PIECES
#undef K2_READ
#undef K2_FENCE
#undef K2_MMA
#undef K2_MMA_D
#undef K2_STEP
}

// ---- 128 x 128 tile, 256 threads, TWO workgroups per CU --------------------------------------------------------------------------
// With three products per chunk a 512-thread workgroup alone on its CU spends as long at its barrier, its first fragments and its DMA
// instructions as in its MFMAs; two independent 4-wave workgroups per CU (2 x 2 waves of 64 x 64, 2 x 32 KB of LDS each) cover each
// other's -- the same eight waves per CU, no shared barrier.  One tile per workgroup (the CU's other workgroup covers prologue and
// epilogue), the next chunk's DMA issued at the top of the chunk.  Same three products in the same order per element: same bits.
__global__ __launch_bounds__(256, 2) void gemm_h2_mid_kernel(const GemmParams pin) {
  GemmParams p = pin;
  constexpr int B = 128, PL = B * 64, BUF = 4 * PL;                        // 32 KB per stage: [A hi|lo][W hi|lo]
  __shared__ __attribute__((aligned(1024))) char lds[2 * BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nbn = (p.N + B - 1) / B;
  const int tile = xcd_tile(p);
  const int bm = tile / nbn, bn = tile - bm * nbn;
  const int m0 = bm * B, n0 = bn * B;
  // DMA: 32 pieces of 1 KiB per chunk, eight per wave: row blocks wave and wave + 4 of both terms of both operands
  const int lr = lane >> 2;
  const int lslot = ((lane & 3) ^ ((0 - (lane >> 4)) & 3)) << 4;
  const unsigned a_off0 = (unsigned)min(m0 + 16 * wave + lr, p.M - 1) * 64u + lslot;
  const unsigned a_off1 = (unsigned)min(m0 + 16 * (wave + 4) + lr, p.M - 1) * 64u + lslot;
  const unsigned w_off0 = (unsigned)min(n0 + 16 * wave + lr, p.N - 1) * 64u + lslot;
  const unsigned w_off1 = (unsigned)min(n0 + 16 * (wave + 4) + lr, p.N - 1) * 64u + lslot;
  const size_t a_ts = (size_t)p.a_rows * 64, w_ts = (size_t)p.w_rows * 64;
  // piece q (0 .. 7) of chunk kc -> buffer buf: operand q / 4, term (q / 2) % 2, row block wave + 4 (q % 2)
#define HM_PIECE(kc, buf, q)                                                                                             \
  {                                                                                                                      \
    char* lb_ = lds + (buf) * BUF + (((q) >> 1) & 1) * PL + ((q) >> 2) * 2 * PL + (wave + 4 * ((q) & 1)) * 1024;          \
    if ((q) < 4) {                                                                                                       \
      const char* as_ = reinterpret_cast<const char*>(p.Ap) + ((size_t)(kc) * 2 + (((q) >> 1) & 1)) * a_ts;               \
      __builtin_amdgcn_global_load_lds((gptr_t)(as_ + (((q) & 1) ? a_off1 : a_off0)), (lptr_t)lb_, 16, 0, 0);            \
    } else {                                                                                                             \
      const char* ws_ = reinterpret_cast<const char*>(p.Wp) + ((size_t)(kc) * 2 + (((q) >> 1) & 1)) * w_ts;               \
      __builtin_amdgcn_global_load_lds((gptr_t)(ws_ + (((q) & 1) ? w_off1 : w_off0)), (lptr_t)lb_, 16, 0, 0);            \
    }                                                                                                                    \
  }
  const int fr = lane & 15, fq = lane >> 4;
  int a_fo[4], w_fo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wm * 64 + 16 * i + fr;
    a_fo[i] = r * 64 + ((fq ^ ((0 - (r >> 2)) & 3)) << 4);
    const int c = wn * 64 + 16 * i + fr;
    w_fo[i] = 2 * PL + c * 64 + ((fq ^ ((0 - (c >> 2)) & 3)) << 4);
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nk = p.K >> 5;
#pragma unroll
  for (int q = 0; q < 8; ++q) HM_PIECE(0, 0, q)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#define HM_FENCE __builtin_amdgcn_sched_barrier(0);
#define HM_MMA(fwp, fap)                                                                      \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) \
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fwp[j], fap[i], acc[i][j], 0, 0, 0);
  // a product with DMA pieces q0 .. q0 + 3 of the next chunk behind its MFMAs 4, 8, 12 and 16 (block-uniform `more`)
#define HM_MMA_D(fwp, fap, q0)                                                                            \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) {           \
    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fwp[j], fap[i], acc[i][j], 0, 0, 0);               \
    if (j == 3 && more) { HM_FENCE HM_PIECE(kc + 1, (kc + 1) & 1, (q0) + i) HM_FENCE }                    \
  }
  for (int kc = 0; kc < nk; ++kc) {
    const bool more = kc + 1 < nk;                                         // the barrier that ended chunk kc - 1 freed the other buffer
    const char* rb = lds + (kc & 1) * BUF;
    f16x8 a_hi[4], a_lo[4], w_hi[4], w_lo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a_hi[i] = *reinterpret_cast<const f16x8*>(rb + a_fo[i]);
      w_lo[i] = *reinterpret_cast<const f16x8*>(rb + PL + w_fo[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a_lo[i] = *reinterpret_cast<const f16x8*>(rb + PL + a_fo[i]);
      w_hi[i] = *reinterpret_cast<const f16x8*>(rb + w_fo[i]);
    }
    HM_FENCE
    HM_MMA_D(w_lo, a_hi, 0) /* (hi, lo) */
    HM_MMA_D(w_hi, a_lo, 4) /* (lo, hi) */
    HM_MMA(w_hi, a_hi)      /* (hi, hi) */
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
#undef HM_PIECE
#undef HM_FENCE
#undef HM_MMA
#undef HM_MMA_D
  if (p.Cp) h2_planes_epilogue<4>(p, acc, m0 + wm * 64, n0 + wn * 64, fr, fq);
  if (p.C) {
    h2_descale<4>(p, acc, n0 + wn * 64, fq);
    h2_descale_rows<4, 4>(p, acc, m0 + wm * 64, fr);
    gemm_epilogue<4, 4>(p, acc, m0, n0, wm * 64, wn * 64, fr, fq);
  }
}

// ---- small problems: 64 x 64 tile, one tile per workgroup, 2 x 2 waves of 32 x 32 (three workgroups per CU) -------------------------
// The same three products in the same order per element: a row has the same bits here as in the 256-row kernel, so the choice
// between them may look at the row count (and the forward computes the same bits at every batch size).
// A three-slot ring, the DMA two chunks ahead (three workgroups per CU still fit): a tile alone on its CU (one clip: 4 x 24 tiles for the QKV projection) used to pay one
// L2 round trip PER CHUNK (issue -> wait -> barrier with 12 MFMAs in between: 9 us for 251 x 1536 x 512); now three chunks' fetches
// are in flight behind the one being multiplied (a four-slot ring -- two workgroups per CU -- was slower on the mid-size problems).  Counted s_waitcnt vmcnt (four DMA instructions per wave and chunk) and a RAW
// s_barrier: __syncthreads() would drain the DMA queue (cdna_hip_programming.md, Pipelining across barriers).
// KC = 2: two K chunks per ring slot and barrier (same products in the same order: the same bits) -- for problems of so few tiles that a
// workgroup's life is its chunk loop (one clip of cfg3: the FFN's second layer is 32 workgroups x 64 chunks); 96 KB of LDS, one
// workgroup per CU, which such a grid does not fill anyway
template <int KC>
__global__ __launch_bounds__(256, KC == 1 ? 3 : 1) void gemm_h2_small_kernel(const GemmParams pin) {
  GemmParams p = pin;
  constexpr int B = 64, PL = B * 64, BUF1 = 4 * PL, BUF = KC * BUF1;       // 16 KB per chunk: [A hi|lo][W hi|lo]
  extern __shared__ __attribute__((aligned(1024))) char lds[];            // a ring of three slots: 3 * BUF
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nbn = (p.N + B - 1) / B;
  const int tile = xcd_tile(p);
  const int bm = tile / nbn, bn = tile - bm * nbn;
  const int m0 = bm * B, n0 = bn * B;
  // DMA: 16 pieces of 1 KiB (16 rows of one term of one operand) per chunk, four per wave: wave -> row block, all four (operand, term)
  const int lr = lane >> 2;
  const int lslot = ((lane & 3) ^ ((0 - (lane >> 4)) & 3)) << 4;
  const unsigned a_off = (unsigned)min(m0 + 16 * wave + lr, p.M - 1) * 64u + lslot;
  const unsigned w_off = (unsigned)min(n0 + 16 * wave + lr, p.N - 1) * 64u + lslot;
  const size_t a_ts = (size_t)p.a_rows * 64, w_ts = (size_t)p.w_rows * 64;
  auto issue = [&](int ks, int buf) {                                       // step ks = chunks KC ks .. KC ks + KC - 1
#pragma unroll
    for (int u = 0; u < KC; ++u) {
      const int kc = KC * ks + u;
      char* lb = lds + buf * BUF + u * BUF1 + wave * 1024;
      const char* as = reinterpret_cast<const char*>(p.Ap) + (size_t)kc * 2 * a_ts + a_off;
      const char* ws = reinterpret_cast<const char*>(p.Wp) + (size_t)kc * 2 * w_ts + w_off;
      __builtin_amdgcn_global_load_lds((gptr_t)as, (lptr_t)lb, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(as + a_ts), (lptr_t)(lb + PL), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)ws, (lptr_t)(lb + 2 * PL), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(ws + w_ts), (lptr_t)(lb + 3 * PL), 16, 0, 0);
    }
  };
  const int fr = lane & 15, fq = lane >> 4;
  int a_fo[2], w_fo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = wm * 32 + 16 * i + fr;
    a_fo[i] = r * 64 + ((fq ^ ((0 - (r >> 2)) & 3)) << 4);
    const int c = wn * 32 + 16 * i + fr;
    w_fo[i] = 2 * PL + c * 64 + ((fq ^ ((0 - (c >> 2)) & 3)) << 4);
  }
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nk = (p.K >> 5) / KC;                                          // steps (the launcher: K % (32 KC) == 0)
  issue(0, 0);
  if (nk > 1) issue(1, 1);
  int slot = 0;                                                            // ks % 3
  for (int kc = 0; kc < nk; ++kc) {
    // step kc has landed once at most the DMA of the step behind it is outstanding (block-uniform choice of the literal)
    if (kc + 1 < nk) {
      if constexpr (KC == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     // this wave's fragment reads of step kc - 1 are home
    __builtin_amdgcn_s_barrier();                                         // every wave's share of step kc is in LDS; slot (kc - 1) % 3 is free
    if (kc + 2 < nk) issue(kc + 2, slot == 0 ? 2 : slot - 1);
    const char* rb0 = lds + slot * BUF;
    slot = slot == 2 ? 0 : slot + 1;
#define H2S_MMA(fwp, fap)                                                                     \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fwp[j], fap[i], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int u = 0; u < KC; ++u) {
      const char* rb = rb0 + u * BUF1;
      f16x8 a_hi[2], a_lo[2], w_hi[2], w_lo[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a_hi[i] = *reinterpret_cast<const f16x8*>(rb + a_fo[i]);
        a_lo[i] = *reinterpret_cast<const f16x8*>(rb + PL + a_fo[i]);
        w_hi[i] = *reinterpret_cast<const f16x8*>(rb + w_fo[i]);
        w_lo[i] = *reinterpret_cast<const f16x8*>(rb + PL + w_fo[i]);
      }
      H2S_MMA(w_lo, a_hi) /* (hi, lo) */
      H2S_MMA(w_hi, a_lo) /* (lo, hi) */
      H2S_MMA(w_hi, a_hi) /* (hi, hi) */
    }
#undef H2S_MMA
  }
  // epilogue on the 32 x 32 wave tile: a 4 x 2 accumulator view padded with the two row blocks this wave does not own is not worth
  // a second code path -- the 2 x 2 forms of the same functions
  if (p.Cp) {
    // plane output, 2 row blocks x 1 chunk: the body of h2_planes_epilogue_t for a 32-column wave tile
    const bool has_b = p.bias != nullptr;
    const float* bsrc = has_b ? p.bias : p.cscale;
    const int slot = ((fq & 1) << 1) | (fq >> 1);
    const size_t ts = (size_t)p.c_rows * 64;
    const int n32 = n0 + wn * 32;
    const int c0 = min(n32 + 4 * fq, p.N - 4), c1 = min(n32 + 16 + 4 * fq, p.N - 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bsrc + c0), b1 = *reinterpret_cast<const f32x4*>(bsrc + c1);
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(p.cscale + c0), s1 = *reinterpret_cast<const f32x4*>(p.cscale + c1);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + wm * 32 + 16 * i + fr;
      f32x4 x0, x1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        x0[e] = acc[i][0][e] * s0[e];
        x1[e] = acc[i][1][e] * s1[e];
        if (has_b) { x0[e] += b0[e]; x1[e] += b1[e]; }
        x0[e] = apply_act(x0[e], p.act);
        x1[e] = apply_act(x1[e], p.act);
      }
      float lo4[4], hi4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x0[e] * p.cp_scale), __float_as_uint(x1[e] * p.cp_scale), false, false);
        lo4[e] = __uint_as_float(r[0]);
        hi4[e] = __uint_as_float(r[1]);
      }
      unsigned h[4], l[4];
      split_pair_h2(f32x2{lo4[0], lo4[1]}, h[0], l[0]);
      split_pair_h2(f32x2{lo4[2], lo4[3]}, h[1], l[1]);
      split_pair_h2(f32x2{hi4[0], hi4[1]}, h[2], l[2]);
      split_pair_h2(f32x2{hi4[2], hi4[3]}, h[3], l[3]);
      if (m < p.M && n32 < p.N) {
        char* dst = reinterpret_cast<char*>(p.Cp) + (((size_t)(n32 >> 5) * 2) * p.c_rows + m) * 64 + slot * 16;
        *reinterpret_cast<u32x4*>(dst) = u32x4{h[0], h[1], h[2], h[3]};
        *reinterpret_cast<u32x4*>(dst + ts) = u32x4{l[0], l[1], l[2], l[3]};
      }
    }
  }
  if (p.C) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 32 + 16 * j + 4 * fq;
      f32x4 s;
      if (!(p.N & 3)) {
        s = *reinterpret_cast<const f32x4*>(p.cscale + min(n, p.N - 4));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] = p.cscale[min(n + e, p.N - 1)];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][j] *= s;
    }
    h2_descale_rows<2, 2>(p, acc, m0 + wm * 32, fr);
    gemm_epilogue<2, 2>(p, acc, m0, n0, wm * 32, wn * 32, fr, fq);
  }
}

}  // namespace

hipError_t launch_split_h2(const float* x, int ld, unsigned short* planes, long long rows, int M, int K, const int* row_exp, int e,
                           hipStream_t s) {
  if (!x || !planes || M <= 0 || K <= 0 || (K & 31) || (ld & 3) || rows < M) return hipErrorInvalidValue;
  const size_t n = (size_t)M * (K >> 3);
  hipLaunchKernelGGL(split_h2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, ld, planes, rows, M, K, row_exp, e);
  return hipGetLastError();
}

hipError_t launch_h2_row_stats(const float* w, int N, int K, int* ew, float* l2, hipStream_t s) {
  if (!w || !ew || !l2 || N <= 0 || K <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(h2_row_stats_kernel, dim3(N), dim3(256), 0, s, w, K, ew, l2);
  return hipGetLastError();
}

bool gemm_h2_supported(const GemmParams& p) {
  const bool fast = !p.C2 && !(p.N & 3) && !(p.ldc & 3) && (!p.R || !(p.ldr & 3));
  const bool mask = p.C2 && p.X && !(p.N & 1) && !(p.ldc & 1) && !p.R;
  const bool c_ok = !p.C || fast || mask;
  const bool cp_ok = !p.Cp || (!(p.N & 31) && p.c_rows >= p.M && !p.R && !p.C2 && p.act != ACT_SIGMOID && p.cp_scale > 0.0f && !p.rscale);
  return p.Ap && p.Wp && p.cscale && (p.C || p.Cp) && c_ok && cp_ok && p.amode == AMODE_PLAIN && p.a_rows >= p.M && p.w_rows >= p.N &&
         !p.lnx_c1 && !p.ln_gamma && !p.ln_stats && p.ksplit <= 1 && p.mag_F == 0 && p.drop_p <= 0.0f && !(p.K & 31) &&
         p.alt.M <= 0 && !p.epi_general;
}

// Which kernel: the 256 x 128 tile from half a round of its tiles on (the rule of the bf16 kernels, split_t2_min); below, the
// 64 x 64 kernel.  Both compute the same bits.  (A 256 x 256 tile -- half the operand traffic per flop -- does not fit eight waves:
// 128 accumulator + 96 fragment registers per lane, hipcc spills the accumulators inside the loop; the template keeps the
// parameter.)
static int h2_variant(const GemmParams& p) {
  const long t128 = (long)((p.M + 255) / 256) * ((p.N + 127) / 128);
  int v = t128 >= (p.split_t2_min > 0 ? p.split_t2_min : 128) ? 128 : 64;
#ifdef AVSEP_DEV
  if (const char* e = getenv("AVSEP_H2_T2MIN")) v = t128 >= atoi(e) ? 128 : 64;                      // developer A/B
  if (const char* e = getenv("AVSEP_H2_TILE")) v = atoi(e) == 64 || atoi(e) == 128 ? atoi(e) : v;   // developer A/B
#endif
  return v;
}

const char* gemm_h2_instance_name(const GemmParams& p) {
  if (h2_variant(p) == 128) return "gemm_h2_kernel<128, 0>";
  const long tiles = (long)((p.M + 63) / 64) * ((p.N + 63) / 64);
  return (tiles <= 256 && !(p.K & 63) && p.K >= 256) ? "gemm_h2_small_kernel<2>" : "gemm_h2_small_kernel<1>";
}

template <int BN, int V = 0>
static hipError_t launch_h2_big(const GemmParams& p, hipStream_t s) {
  constexpr int BUF = 2 * 256 * 64 + 2 * BN * 64;
  auto kern = gemm_h2_kernel<BN, V>;
  static bool raised[64] = {};
  static int cus[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (!raised[dev]) {                                                      // the dynamic-LDS ceiling, once per device (conv_stack.hip)
    const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF);
    if (attr != hipSuccess) return attr;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return hipErrorInvalidDevice;
    cus[dev] = n;
    raised[dev] = true;
  }
  const long tiles = (long)((p.M + 255) / 256) * ((p.N + BN - 1) / BN);
  const long grid = tiles < cus[dev] ? tiles : cus[dev];                 // one resident workgroup per CU walks tiles / grid tiles
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), 2 * BUF, s, p);
  return hipGetLastError();
}

static hipError_t launch_h2_big2(const GemmParams& p, hipStream_t s) {
  constexpr int BUF = 2 * 256 * 64 + 2 * 128 * 64;
  static bool raised[64] = {};
  static int cus[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (!raised[dev]) {
    const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h2_kernel2), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF);
    if (attr != hipSuccess) return attr;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return hipErrorInvalidDevice;
    cus[dev] = n;
    raised[dev] = true;
  }
  const long tiles = (long)((p.M + 255) / 256) * ((p.N + 127) / 128);
  const long grid = tiles < cus[dev] ? tiles : cus[dev];
  hipLaunchKernelGGL(gemm_h2_kernel2, dim3((unsigned)grid), dim3(512), 2 * BUF, s, p);
  return hipGetLastError();
}

hipError_t launch_gemm_h2(GemmParams p, hipStream_t s) {
  if (!gemm_h2_supported(p) || p.M <= 0 || p.N <= 0 || p.K <= 0) return hipErrorInvalidValue;
  p.nbn_magic = 0;
  if (!p.W) p.W = reinterpret_cast<const float*>(p.Wp);   // gemm_tile.h's epilogue reads N floats from W when there is no bias (discarded)
  if (h2_variant(p) == 128) {
#ifdef AVSEP_DEV
    if (getenv("AVSEP_H2_KERNEL2") && !(p.K & 63)) return launch_h2_big2(p, s);   // developer A/B
    if (const char* e = getenv("AVSEP_H2_ABL")) {                                  // developer ablations (wrong results, timing only)
      switch (atoi(e)) {
        case 1: return launch_h2_big<128, 1>(p, s);
        case 2: return launch_h2_big<128, 2>(p, s);
        case 4: return launch_h2_big<128, 4>(p, s);
        case 8: return launch_h2_big<128, 8>(p, s);
        case 10: return launch_h2_big<128, 10>(p, s);
        case 12: return launch_h2_big<128, 12>(p, s);
        default: break;
      }
    }
    if (getenv("AVSEP_H2_MID")) {
      const long tiles = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
      hipLaunchKernelGGL(gemm_h2_mid_kernel, dim3((unsigned)tiles), dim3(256), 0, s, p);
      return hipGetLastError();
    }
#endif
    return launch_h2_big<128>(p, s);
  }
  const long tiles = (long)((p.M + 63) / 64) * ((p.N + 63) / 64);
  // two chunks per barrier where the grid cannot fill the chip anyway (<= one 64 x 64 tile per CU) and the chunk loop is long
  // enough to matter: the same bits, so the choice may look at the row count
  bool kc2 = tiles <= 256 && !(p.K & 63) && p.K >= 256;
#ifdef AVSEP_DEV
  if (const char* e = getenv("AVSEP_H2_SMALL_KC")) kc2 = atoi(e) == 2 && !(p.K & 63);             // developer A/B
#endif
  if (kc2) {
    static bool raised[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!raised[dev]) {                                                    // the dynamic-LDS ceiling, once per device
      const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h2_small_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 2 * 16384);
      if (attr != hipSuccess) return attr;
      raised[dev] = true;
    }
    hipLaunchKernelGGL(gemm_h2_small_kernel<2>, dim3((unsigned)tiles), dim3(256), 3 * 2 * 16384, s, p);
  } else {
    hipLaunchKernelGGL(gemm_h2_small_kernel<1>, dim3((unsigned)tiles), dim3(256), 3 * 16384, s, p);
  }
  return hipGetLastError();
}
