// gemm_planes.hip -- split-precision GEMM, third generation (round 5): the operands arrive PRE-SPLIT.
//
// gemm_split.hip cuts every fp32 operand into its three bf16 terms on the way to LDS, in every column tile again (a row of A is
// split 4-16 times per GEMM), through the VGPR -> LDS store path (72 KB of ds_write_b128 per chunk and CU) -- the part of its
// chunk that neither the matrix pipe nor the LDS reads account for.  Here the three terms of an operand
// are computed ONCE, by whoever produces it (the weight packer, LayerNorm, a GEMM's or the attention's epilogue, the resize), and
// stored as bf16 "planes"; the GEMM's staging is then a copy that the LDS-DMA path (global_load_lds_dwordx4: memory -> LDS, no
// register, no VALU) does while the matrix pipe works.
//
// PLANE FORMAT ("P32") of a matrix X[M][K], K % 32 == 0, in a buffer of `rows` >= M rows:
//     bf16 P[K/32][3][rows][32]        element (m, k) of term t (0 = hi, 1 = mid, 2 = lo) at ((k/32 * 3 + t) * rows + m) * 32 + k%32
// i.e. K-chunk-major: the 32-element chunk of a row is one 64-byte line, consecutive ROWS of a (chunk, term) slab are contiguous.
// A tile's stage (256 rows x 32 k of one term) is ONE contiguous 16 KB region, each DMA instruction (64 lanes x 16 B) moves one
// whole KiB, and every cache line that crosses the fabric is used in full by the instruction that fetched it (with row-major
// bf16 planes a 32-deep chunk would use half of each 128-byte line, and the other half is gone from the 32 KB L1 by the time
// the next chunk wants it).  A row-range view of such a buffer (the half batches of the forward's tail) is the same pointer
// advanced by 32 * row0 elements with the same `rows`.
//
// The terms are those of split_terms.h (truncation, x = hi + mid + lo exactly to 24 bits) and the kernel multiplies the same six
// products in the same order as gemm_split.hip's three kernels, per element: the result has the SAME BITS as theirs on the fp32
// operands the planes were cut from (tests/test_gpu_parity.py::test_op_linear_planes).
//
// Non-finite inputs: a term of +-inf is inf - inf = NaN (split_terms.h) -- non-finite in, non-finite out, but an inf becomes NaN.
#include "kernels.h"
#include "gemm_tile.h"
#include "split_terms.h"
#include <algorithm>
#include <cstdio>
#include <vector>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int PBM = 256, PBN = 128;
constexpr int P_APL = PBM * 64, P_WPL = PBN * 64;                 // bytes of one term of one stage (rows x 32 bf16)
constexpr int P_BUF = 3 * P_APL + 3 * P_WPL;                      // 72 KB: [A hi|mid|lo][W hi|mid|lo]

// ---- fp32 -> planes (weights at finalize time; activations whose producer has no plane epilogue) ------------------------------
// one thread per 8 consecutive elements (one 16-byte slot of each term)
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, int ld, unsigned short* __restrict__ P,
                                                           long long rows, int M, int K) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int spr = K >> 3;                                          // slots per row
  if (idx >= (size_t)M * spr) return;
  const int m = (int)(idx / spr), s = (int)(idx - (size_t)m * spr);
  const float* src = x + (size_t)m * ld + 8 * s;
  const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
  unsigned h[4], mi[4], lo[4];
  split_pair(f32x2{v0[0], v0[1]}, h[0], mi[0], lo[0]);
  split_pair(f32x2{v0[2], v0[3]}, h[1], mi[1], lo[1]);
  split_pair(f32x2{v1[0], v1[1]}, h[2], mi[2], lo[2]);
  split_pair(f32x2{v1[2], v1[3]}, h[3], mi[3], lo[3]);
  const int kc = s >> 2, slot = s & 3;
  char* base = reinterpret_cast<char*>(P) + (((size_t)kc * 3) * rows + m) * 64 + slot * 16;
  const size_t ts = (size_t)rows * 64;
  *reinterpret_cast<u32x4*>(base) = u32x4{h[0], h[1], h[2], h[3]};
  *reinterpret_cast<u32x4*>(base + ts) = u32x4{mi[0], mi[1], mi[2], mi[3]};
  *reinterpret_cast<u32x4*>(base + 2 * ts) = u32x4{lo[0], lo[1], lo[2], lo[3]};
}

// ---- plane-output epilogue ---------------------------------------------------------------------------------------------------
// y = act(acc + bias), written as the three terms of the NEXT GEMM's A operand.  Lane (fr, fq) of a 16 x 16 block holds 4
// consecutive columns 4 fq .. 4 fq + 3 of row fr (gemm_tile.h, D^T); a 16-byte slot of a plane is 8 consecutive columns.
// v_permlane16_swap_b32 (gfx950) exchanges the odd 16-lane rows of one register with the even rows of another: applied to the
// blocks j = 2t and 2t + 1 of a 32-column chunk it leaves lane fq with 8 consecutive columns -- fq 0: block 2t, columns 0-7;
// fq 1: block 2t + 1, columns 0-7 (chunk columns 16-23); fq 2: block 2t, 8-15; fq 3: block 2t + 1, 8-15 (24-31) -- so the 64 lanes
// store one whole contiguous KiB (16 rows x 64 B) per (row block, chunk, term).
template <int ACT>
__device__ __forceinline__ void planes_epilogue_t(const GemmParams& p, const f32x4 (&acc)[4][4], int mbase, int nbase, int fr,
                                                  int fq) {
  const bool has_b = p.bias != nullptr;                                     // block-uniform
  const float* bsrc = has_b ? p.bias : reinterpret_cast<const float*>(p.Wp);   // at least N * K * 6 bytes: always readable
  f32x4 bv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const f32x4*>(bsrc + min(nbase + 16 * j + 4 * fq, p.N - 4));
  const int slot = ((fq & 1) << 1) | (fq >> 1);
  const size_t ts = (size_t)p.c_rows * 64;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = mbase + 16 * i + fr;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x4 x0, x1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        x0[e] = has_b ? acc[i][2 * t][e] + bv[2 * t][e] : acc[i][2 * t][e];
        x1[e] = has_b ? acc[i][2 * t + 1][e] + bv[2 * t + 1][e] : acc[i][2 * t + 1][e];
      }
      if (ACT == ACT_GELU || ACT == ACT_SIGMOID) {
        x0 = act4_outofline<ACT>(x0);
        x1 = act4_outofline<ACT>(x1);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { x0[e] = act_t<ACT>(x0[e]); x1[e] = act_t<ACT>(x1[e]); }
      }
      float lo4[4], hi4[4];                                                // this lane's 8 consecutive columns
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x0[e]), __float_as_uint(x1[e]), false, false);
        lo4[e] = __uint_as_float(r[0]);
        hi4[e] = __uint_as_float(r[1]);
      }
      unsigned h[4], mi[4], lo[4];
      split_pair(f32x2{lo4[0], lo4[1]}, h[0], mi[0], lo[0]);
      split_pair(f32x2{lo4[2], lo4[3]}, h[1], mi[1], lo[1]);
      split_pair(f32x2{hi4[0], hi4[1]}, h[2], mi[2], lo[2]);
      split_pair(f32x2{hi4[2], hi4[3]}, h[3], mi[3], lo[3]);
      const int n32 = nbase + 32 * t;                                      // first column of the chunk
      if (m < p.M && n32 < p.N) {                                          // N % 32 == 0 (launcher)
        char* dst = reinterpret_cast<char*>(p.Cp) + (((size_t)(n32 >> 5) * 3) * p.c_rows + m) * 64 + slot * 16;
        *reinterpret_cast<u32x4*>(dst) = u32x4{h[0], h[1], h[2], h[3]};
        *reinterpret_cast<u32x4*>(dst + ts) = u32x4{mi[0], mi[1], mi[2], mi[3]};
        *reinterpret_cast<u32x4*>(dst + 2 * ts) = u32x4{lo[0], lo[1], lo[2], lo[3]};
      }
    }
  }
}

__device__ __forceinline__ void planes_epilogue(const GemmParams& p, const f32x4 (&acc)[4][4], int mbase, int nbase, int fr, int fq) {
  switch (p.act) {                                // block-uniform
    case ACT_RELU: planes_epilogue_t<ACT_RELU>(p, acc, mbase, nbase, fr, fq); break;
    case ACT_GELU: planes_epilogue_t<ACT_GELU>(p, acc, mbase, nbase, fr, fq); break;
    default: planes_epilogue_t<ACT_NONE>(p, acc, mbase, nbase, fr, fq); break;
  }
}

// ---- the 256 x 128 kernel ------------------------------------------------------------------------------------------------------
// gemm_split_kernel2's tile, wave layout (4 x 2 waves of 64 x 64), LDS image (rows of 64 B, 16-byte slots XOR-swizzled with
// -(row >> 2) & 3: fragment reads conflict-free under this chip's per-instruction lane groups), product order and persistence
// over tiles.  What changed is the staging: per chunk 72 DMA instructions (A: 3 terms x 16 row blocks, W: 3 x 8), 9 per wave,
// each moving the KiB of 16 tile rows of one term into the OTHER buffer; a DMA writes lane l at base + 16 l, i.e. row l / 4,
// physical slot l % 4, so the lane fetches logical slot (l % 4) ^ swizzle(row) -- the swizzle is applied to the SOURCE address.
// The nine are issued three at a time behind the first three products of the chunk (an issue holds its wave ~60-180 cycles:
// spread out, the SIMD's other wave keeps the matrix pipe fed meanwhile) and have the rest of the chunk to land; one
// s_waitcnt vmcnt(0) + one barrier per chunk.  No staging registers, no VALU in the loop but addresses.
struct PTile {
  unsigned a0, a1, w;     // this lane's byte offsets inside a (chunk, term) slab: row * 64 + logical slot * 16
};

// V (developer A/B; the product ships one instance): bit 0 = every fragment of a chunk is requested in its first third, the lo
// terms in registers of their own (the in-flight kernels pass both lo terms through ONE register set, which makes the second
// wait for the first product's MFMAs to issue: four exposed LDS round trips per chunk); bit 1 = s_setprio 1 around the MFMA
// groups; bit 2 = STAGGER: waves 4-7 (the SIMD partners of waves 0-3) run half a chunk behind -- the last three products of a
// chunk need no LDS (every fragment is in registers after the third), so they carry them across the barrier and issue them at
// the top of the next step, while their partners wait for that step's first fragments; the partners' MFMAs in turn cover their
// own fragment reads in the middle of the step (MI355X_MICROARCH.md, Two waves per SIMD, item 9).  Same products, same order,
// same accumulators per element in every variant: same bits.
template <int V>
__global__ __launch_bounds__(512, 1) void gemm_planes_kernel(const GemmParams pin) {
  GemmParams p = pin;
  extern __shared__ __attribute__((aligned(1024))) char ldsp[];           // 2 x P_BUF
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nbn = (p.N + PBN - 1) / PBN;
  const int ntiles = ((p.M + PBM - 1) / PBM) * nbn;
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

  const int lr = lane >> 2;                                                // row inside a DMA instruction's 16-row block
  const int lslot = ((lane & 3) ^ ((0 - (lane >> 4)) & 3)) << 4;           // logical slot (bytes) this lane fetches
  auto tile_rows = [&](int t) {
    const int bm = t / nbn, bn = t - bm * nbn;
    PTile r;
    r.a0 = (unsigned)min(bm * PBM + 16 * wave + lr, p.M - 1) * 64u + lslot;
    r.a1 = (unsigned)min(bm * PBM + 16 * (wave + 8) + lr, p.M - 1) * 64u + lslot;
    r.w = (unsigned)min(bn * PBN + 16 * wave + lr, p.N - 1) * 64u + lslot;
    return r;
  };
  const size_t a_ts = (size_t)p.a_rows * 64, w_ts = (size_t)p.w_rows * 64;   // bytes of one (chunk, term) slab
  // term `t` of chunk kc of tile rows T -> buffer `buf`: three DMA instructions of this wave
#define P_DMA(kc, T, buf, t)                                                                                             \
  {                                                                                                                      \
    const char* as_ = reinterpret_cast<const char*>(p.Ap) + ((size_t)(kc) * 3 + (t)) * a_ts;                              \
    const char* ws_ = reinterpret_cast<const char*>(p.Wp) + ((size_t)(kc) * 3 + (t)) * w_ts;                              \
    char* lb_ = ldsp + (buf) * P_BUF;                                                                                    \
    __builtin_amdgcn_global_load_lds((gptr_t)(as_ + T.a0), (lptr_t)(lb_ + (t) * P_APL + wave * 1024), 16, 0, 0);         \
    __builtin_amdgcn_global_load_lds((gptr_t)(as_ + T.a1), (lptr_t)(lb_ + (t) * P_APL + (wave + 8) * 1024), 16, 0, 0);   \
    __builtin_amdgcn_global_load_lds((gptr_t)(ws_ + T.w), (lptr_t)(lb_ + 3 * P_APL + (t) * P_WPL + wave * 1024), 16, 0, 0); \
  }

  const int fr = lane & 15, fq = lane >> 4;
  int a_fo[4], w_fo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wm * 64 + 16 * i + fr;
    a_fo[i] = r * 64 + ((fq ^ ((0 - (r >> 2)) & 3)) << 4);
    const int c = wn * 64 + 16 * i + fr;
    w_fo[i] = 3 * P_APL + c * 64 + ((fq ^ ((0 - (c >> 2)) & 3)) << 4);
  }

  PTile cur = tile_rows(tile);
  P_DMA(0, cur, 0, 0)
  P_DMA(0, cur, 0, 1)
  P_DMA(0, cur, 0, 2)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

#define P_FRAG_A(term, f) \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) f[i] = *reinterpret_cast<const bf16x8*>(rb + (term) * P_APL + a_fo[i]);
#define P_FRAG_W(term, f) \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) f[i] = *reinterpret_cast<const bf16x8*>(rb + (term) * P_WPL + w_fo[i]);
#define P_MMA(fwp, fap)                                                                                       \
  {                                                                                                           \
    if (V & 2) __builtin_amdgcn_s_setprio(1);                                                                 \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j)               \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fwp[j], fap[i], acc[i][j], 0, 0, 0);              \
    if (V & 2) __builtin_amdgcn_s_setprio(0);                                                                 \
  }
#define P_FENCE __builtin_amdgcn_sched_barrier(0);
#define P_END_STEP                                                                           \
  par ^= 1;                                                                                  \
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                           \
  __syncthreads();
  // the first three products of a chunk (smallest first: gemm_split.hip's order) over buffer `par`: every LDS read of the chunk
#define P_HEAD(DMA0, DMA1, DMA2)                                                             \
  {                                                                                          \
    const char* rb = ldsp + par * P_BUF;                                                     \
    if (V & 1) {                                                                             \
      P_FRAG_A(0, a_hi) P_FRAG_W(2, w_lo) P_FRAG_W(0, w_hi) P_FRAG_A(2, a_lo)                \
      P_FENCE                                                                                \
      P_MMA(w_lo, a_hi) /* (hi, lo) */                                                       \
      P_FRAG_A(1, a_mid) P_FRAG_W(1, w_mid)                                                  \
      DMA0 P_FENCE                                                                           \
      P_MMA(w_hi, a_lo) /* (lo, hi) */                                                       \
      DMA1 P_FENCE                                                                           \
      P_MMA(w_mid, a_mid) /* (mid, mid) */                                                   \
      DMA2 P_FENCE                                                                           \
    } else {                                                                                 \
      P_FRAG_A(0, a_hi) P_FRAG_W(2, w_lo)                                                    \
      P_MMA(w_lo, a_hi)                                                                      \
      DMA0 P_FENCE                                                                           \
      P_FRAG_W(0, w_hi) P_FRAG_A(2, w_lo)                                                    \
      P_MMA(w_hi, w_lo)                                                                      \
      DMA1 P_FENCE                                                                           \
      P_FRAG_A(1, a_mid) P_FRAG_W(1, w_mid)                                                  \
      P_MMA(w_mid, a_mid)                                                                    \
      DMA2 P_FENCE                                                                           \
    }                                                                                        \
  }
  // the last three: registers only
#define P_TAIL(DMA0, DMA1, DMA2)                                                             \
  P_MMA(w_hi, a_mid)  /* (mid, hi) */                                                        \
  DMA0 P_FENCE                                                                               \
  P_MMA(w_mid, a_hi)  /* (hi, mid) */                                                        \
  DMA1 P_FENCE                                                                               \
  P_MMA(w_hi, a_hi)   /* (hi, hi) */                                                         \
  DMA2 P_FENCE
#define P_NONE

  const int nk = p.K >> 5;
  int par = 0;
  bf16x8 a_hi[4], w_hi[4], a_mid[4], w_mid[4], w_lo[4], a_lo[4];
  const bool late = (V & 4) && wave >= 4;                                  // wave-uniform
  for (;;) {
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // a tile's last step fetches the first chunk of the next tile (without one: its own again, into the buffer nobody reads)
    const int tile_next = tile + tile_step;
    const bool more = tile_next < tile_end;
    const PTile nxt = tile_rows(more ? tile_next : tile);
    if (!late) {
      for (int kc = 1; kc < nk; ++kc) {
        P_HEAD(P_DMA(kc, cur, par ^ 1, 0), P_DMA(kc, cur, par ^ 1, 1), P_DMA(kc, cur, par ^ 1, 2))
        P_TAIL(P_NONE, P_NONE, P_NONE)
        P_END_STEP
      }
      P_HEAD(P_DMA(0, nxt, par ^ 1, 0), P_DMA(0, nxt, par ^ 1, 1), P_DMA(0, nxt, par ^ 1, 2))
      P_TAIL(P_NONE, P_NONE, P_NONE)
      P_END_STEP
    } else {
      // staggered waves: step 0 has no carried products; steps 1 .. nk - 1 open with the previous chunk's last three (and issue
      // the DMA behind them, early in the step); the tile's last three follow its last barrier
      if (nk > 1) {
        P_DMA(1, cur, par ^ 1, 0) P_DMA(1, cur, par ^ 1, 1) P_DMA(1, cur, par ^ 1, 2)
      } else {
        P_DMA(0, nxt, par ^ 1, 0) P_DMA(0, nxt, par ^ 1, 1) P_DMA(0, nxt, par ^ 1, 2)
      }
      P_HEAD(P_NONE, P_NONE, P_NONE)
      P_END_STEP
      for (int kc = 2; kc < nk; ++kc) {
        P_TAIL(P_DMA(kc, cur, par ^ 1, 0), P_DMA(kc, cur, par ^ 1, 1), P_DMA(kc, cur, par ^ 1, 2))
        P_HEAD(P_NONE, P_NONE, P_NONE)
        P_END_STEP
      }
      if (nk > 1) {
        P_TAIL(P_DMA(0, nxt, par ^ 1, 0), P_DMA(0, nxt, par ^ 1, 1), P_DMA(0, nxt, par ^ 1, 2))
        P_HEAD(P_NONE, P_NONE, P_NONE)
        P_END_STEP
      }
      P_TAIL(P_NONE, P_NONE, P_NONE)
    }
    {
      const int bm = tile / nbn, bn = tile - bm * nbn;
      if (p.Cp) planes_epilogue(p, acc, bm * PBM + wm * 64, bn * PBN + wn * 64, fr, fq);
      if (p.C) gemm_epilogue<4, 4>(p, acc, bm * PBM, bn * PBN, wm * 64, wn * 64, fr, fq);
    }
    if (!more) break;
    tile = tile_next;
    cur = nxt;
  }
#undef P_HEAD
#undef P_TAIL
#undef P_NONE
#undef P_FENCE
#undef P_END_STEP
#undef P_DMA
#undef P_FRAG_A
#undef P_FRAG_W
#undef P_MMA
}

// ---- the same tile with the barrier in the MIDDLE of the chunk -----------------------------------------------------------------
// A chunk's LDS reads are over after its third product (the last three run from registers).  The chunk's one barrier therefore
// sits THERE: behind it every wave knows that (a) all waves hold the chunk's fragments -- its buffer is free, and the DMA of the
// chunk after next goes into it, issued behind the products that follow --, and (b) all waves' DMA of the NEXT chunk has landed
// (each waited for its own in front of the barrier), so the step from one chunk's last product to the next chunk's first
// fragment reads crosses no barrier: no wave waits at a barrier with an empty matrix pipe and then for its first LDS reads.
// A DMA has a whole chunk period to land (issued behind products 4-6, needed behind the next chunk's product 3).
// The nine DMA instructions of a wave follow products 4-6 at one per four MFMAs (an issue holds its wave for 60-180 cycles: one at
// a time, with MFMAs of both waves of the SIMD queued around it).
// V (developer A/B): bit 0 = waves 4-7 issue theirs two MFMAs later than waves 0-3 (SIMD partners then never hold at a DMA
// instruction at the same time); bit 1 / bit 2 = ablations for timing only (no DMA in the loop / no MFMA: wrong results);
// bit 3 = all nine right behind the barrier.
template <int V>
__global__ __launch_bounds__(512, 1) void gemm_planes_kernel_b(const GemmParams pin) {
  GemmParams p = pin;
  extern __shared__ __attribute__((aligned(1024))) char ldsp[];           // 2 x P_BUF
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nbn = (p.N + PBN - 1) / PBN;
  const int ntiles = ((p.M + PBM - 1) / PBM) * nbn;
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
    PTile r;
    r.a0 = (unsigned)min(bm * PBM + 16 * wave + lr, p.M - 1) * 64u + lslot;
    r.a1 = (unsigned)min(bm * PBM + 16 * (wave + 8) + lr, p.M - 1) * 64u + lslot;
    r.w = (unsigned)min(bn * PBN + 16 * wave + lr, p.N - 1) * 64u + lslot;
    return r;
  };
  const size_t a_ts = (size_t)p.a_rows * 64, w_ts = (size_t)p.w_rows * 64;
  // piece q = 3 * term + kind (kind 0 / 1: this wave's two A row blocks, 2: its W row block) of chunk kc of tile rows T -> buffer buf
#define PB_PIECE(kc, T, buf, q)                                                                                                  \
  {                                                                                                                              \
    const int t_ = (q) / 3, k_ = (q) % 3;                                                                                        \
    char* lb_ = ldsp + (buf) * P_BUF;                                                                                            \
    if (k_ == 2) {                                                                                                               \
      const char* ws_ = reinterpret_cast<const char*>(p.Wp) + ((size_t)(kc) * 3 + t_) * w_ts;                                    \
      __builtin_amdgcn_global_load_lds((gptr_t)(ws_ + T.w), (lptr_t)(lb_ + 3 * P_APL + t_ * P_WPL + wave * 1024), 16, 0, 0);     \
    } else {                                                                                                                     \
      const char* as_ = reinterpret_cast<const char*>(p.Ap) + ((size_t)(kc) * 3 + t_) * a_ts;                                    \
      __builtin_amdgcn_global_load_lds((gptr_t)(as_ + (k_ ? T.a1 : T.a0)), (lptr_t)(lb_ + t_ * P_APL + (wave + 8 * k_) * 1024), 16, 0, 0); \
    }                                                                                                                            \
  }
#define PB_PIECES3(kc, T, buf, q0) PB_PIECE(kc, T, buf, q0) PB_PIECE(kc, T, buf, q0 + 1) PB_PIECE(kc, T, buf, q0 + 2)

  const int fr = lane & 15, fq = lane >> 4;
  int a_fo[4], w_fo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wm * 64 + 16 * i + fr;
    a_fo[i] = r * 64 + ((fq ^ ((0 - (r >> 2)) & 3)) << 4);
    const int c = wn * 64 + 16 * i + fr;
    w_fo[i] = 3 * P_APL + c * 64 + ((fq ^ ((0 - (c >> 2)) & 3)) << 4);
  }
  const int nk = p.K >> 5;

  // the fetch cursor runs two chunks ahead of the products, in one flat chunk sequence over this workgroup's tiles; past the
  // last chunk it keeps fetching the last tile's chunks (valid addresses, into buffers nobody reads any more)
  int f_tile = tile, f_kc = 0;
  PTile f_rows = tile_rows(tile);
#define PB_ADVANCE                                   \
  if (++f_kc == nk) {                                \
    f_kc = 0;                                        \
    if (f_tile + tile_step < tile_end) {             \
      f_tile += tile_step;                           \
      f_rows = tile_rows(f_tile);                    \
    }                                                \
  }
  // V & 16 (developer diagnostics, p.dbg): wall-clock stamps (10 ns ticks) of thread 0 -- [0] entry, [1] first chunks landed, then per
  // tile [2 + 2t] last product issued, [3 + 2t] epilogue issued
#define PB_STAMP(idx) \
  if ((V & 16) && p.dbg && tid == 0 && (idx) < 32) p.dbg[(size_t)blockIdx.x * 32 + (idx)] = __builtin_amdgcn_s_memrealtime();
  int stamp_i = 2;
  PB_STAMP(0)
  if ((V & 16) && p.dbg && tid == 0) p.dbg[(size_t)blockIdx.x * 32 + 30] = __builtin_amdgcn_s_memtime();      // shader-clock cycles
  PB_PIECES3(f_kc, f_rows, 0, 0) PB_PIECES3(f_kc, f_rows, 0, 3) PB_PIECES3(f_kc, f_rows, 0, 6)
  PB_ADVANCE
  PB_PIECES3(f_kc, f_rows, 1, 0) PB_PIECES3(f_kc, f_rows, 1, 3) PB_PIECES3(f_kc, f_rows, 1, 6)
  PB_ADVANCE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  PB_STAMP(1)

#define P_FRAG_A(term, f) \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) f[i] = *reinterpret_cast<const bf16x8*>(rb + (term) * P_APL + a_fo[i]);
#define P_FRAG_W(term, f) \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) f[i] = *reinterpret_cast<const bf16x8*>(rb + (term) * P_WPL + w_fo[i]);
#define P_MMA(fwp, fap)                                                                                       \
  {                                                                                                           \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) {             \
      if (V & 4) asm volatile("" ::"v"(fwp[j]), "v"(fap[i]));                                                 \
      else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fwp[j], fap[i], acc[i][j], 0, 0, 0);           \
    }                                                                                                         \
  }
#define P_FENCE __builtin_amdgcn_sched_barrier(0);
  // a product with one DMA piece (q0, q0 + 1, q0 + 2 of this step's fetch -> buffer par) behind its MFMAs 4, 8 and 12 (6, 10, 14 for
  // the late waves)
#define P_MMA_D(fwp, fap, q0)                                                                                 \
  {                                                                                                           \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) {             \
      if (V & 4) asm volatile("" ::"v"(fwp[j]), "v"(fap[i]));                                                 \
      else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fwp[j], fap[i], acc[i][j], 0, 0, 0);           \
      if (!(V & 2) && !(V & 8)) {                                                                             \
        if (j == 3 && i < 3) { if (!late) { P_FENCE PB_PIECE(l_kc, l_rows, par, q0 + i) P_FENCE } }           \
        if (j == 1 && i >= 1) { if (late) { P_FENCE PB_PIECE(l_kc, l_rows, par, q0 + i - 1) P_FENCE } }       \
      }                                                                                                       \
    }                                                                                                         \
  }
#define PB_ADVANCE_SAVE \
  l_kc = f_kc;          \
  l_rows = f_rows;      \
  PB_ADVANCE

  int par = 0, kc = 0;
  bf16x8 a_hi[4], w_hi[4], a_mid[4], w_mid[4], w_lo[4], a_lo[4];
  const bool late = (V & 1) && wave >= 4;                                  // wave-uniform
  int l_kc = 0;                                                            // this step's fetch
  PTile l_rows = f_rows;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (;;) {
    if (V & 32) {   // TIMING PROBE of a two-term, three-product scheme (wrong results): terms 0 and 1 only
      const char* rb = ldsp + par * P_BUF;
      P_FRAG_A(0, a_hi) P_FRAG_W(1, w_mid) P_FRAG_W(0, w_hi) P_FRAG_A(1, a_mid)
      P_FENCE
      P_MMA(w_mid, a_hi)
      P_FENCE
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      PB_ADVANCE_SAVE
      P_MMA_D(w_hi, a_mid, 0)
      P_MMA_D(w_hi, a_hi, 3)
    } else {
    {
      const char* rb = ldsp + par * P_BUF;
      P_FRAG_A(0, a_hi) P_FRAG_W(2, w_lo) P_FRAG_W(0, w_hi) P_FRAG_A(2, a_lo)
      P_FENCE
      P_MMA(w_lo, a_hi) /* (hi, lo) */
      P_FRAG_A(1, a_mid) P_FRAG_W(1, w_mid)
      P_FENCE
      P_MMA(w_hi, a_lo) /* (lo, hi) */
      P_FENCE
      P_MMA(w_mid, a_mid) /* (mid, mid) */
      P_FENCE
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    PB_ADVANCE_SAVE
    if (!(V & 2) && (V & 8)) { PB_PIECES3(l_kc, l_rows, par, 0) PB_PIECES3(l_kc, l_rows, par, 3) PB_PIECES3(l_kc, l_rows, par, 6) P_FENCE }
    P_MMA_D(w_hi, a_mid, 0)  /* (mid, hi) */
    P_MMA_D(w_mid, a_hi, 3)  /* (hi, mid) */
    P_MMA_D(w_hi, a_hi, 6)   /* (hi, hi) */
    }
    par ^= 1;
    if (++kc == nk) {                                                      // block-uniform
      const int bm = tile / nbn, bn = tile - bm * nbn;
      PB_STAMP(stamp_i)
      if (p.Cp) planes_epilogue(p, acc, bm * PBM + wm * 64, bn * PBN + wn * 64, fr, fq);
      if (p.C) gemm_epilogue<4, 4>(p, acc, bm * PBM, bn * PBN, wm * 64, wn * 64, fr, fq);
      PB_STAMP(stamp_i + 1)
      if ((V & 16) && p.dbg && tid == 0) p.dbg[(size_t)blockIdx.x * 32 + 31] = __builtin_amdgcn_s_memtime();
      stamp_i += 2;
      tile += tile_step;
      if (tile >= tile_end) break;
      kc = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // no DMA may outlive the workgroup's LDS
#undef PB_PIECE
#undef PB_PIECES3
#undef PB_ADVANCE
#undef PB_ADVANCE_SAVE
#undef PB_STAMP
#undef P_MMA_D
#undef P_FENCE
#undef P_FRAG_A
#undef P_FRAG_W
#undef P_MMA
}

// ---- the same tile as a software pipeline over the chunk boundary -------------------------------------------------------------
// What bounds the two kernels above (profiles/r05_planes_probe_v3.txt: without any DMA in the loop they run at 231 instead of
// 212 TFLOP/s -- and still only 65 % of the matrix pipe): every fragment of a chunk is read in its first half (the six products
// need all six operand sets by the third), 192 KB of LDS reads per CU crowd into the time of 48 of the 96 MFMAs, and the second
// half reads nothing.  Here each product's period carries the LDS reads of ONE operand set (4 ds_read_b128 per wave) for a product
// that is at least one period away:
//     product (chunk g)     operands                 reads issued with it (for chunk g + 1 unless noted)
//     1 (hi, lo)            a_hi[g&1], w_lo          w_hi of chunk g  (free since product 6 of g - 1; needed by product 2)
//     2 (lo, hi)            a_lo, w_hi               --      then: wait for own DMA, BARRIER
//     3 (mid, mid)          a_mid, w_mid             w_lo    (free since product 1)            + 3 DMA pieces of chunk g + 2
//     4 (mid, hi)           a_mid, w_hi              a_lo    (free since product 2)            + 3
//     5 (hi, mid)           a_hi[g&1], w_mid         a_mid   (free since product 4)            + 3
//     6 (hi, hi)            a_hi[g&1], w_hi          w_mid (free since product 5), a_hi[(g+1)&1] (a second register set)
// The barrier behind product 2 says (a) every wave has read the last fragment of chunk g's buffer (w_hi, with product 1): the DMA
// of chunk g + 2 may overwrite it, and has until the next barrier to land; (b) every wave's DMA of chunk g + 1 has landed: its
// buffer may be read from product 3 on.  Seven fragment sets (112 registers) + 64 accumulators.  The last chunk of a tile skips the
// prefetch (the epilogue needs the registers) and re-reads after it.  Same products, same order: same bits.
template <int V>
__global__ __launch_bounds__(512, 1) void gemm_planes_kernel_c(const GemmParams pin) {
  GemmParams p = pin;
  extern __shared__ __attribute__((aligned(1024))) char ldsp[];           // 2 x P_BUF
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nbn = (p.N + PBN - 1) / PBN;
  const int ntiles = ((p.M + PBM - 1) / PBM) * nbn;
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
    PTile r;
    r.a0 = (unsigned)min(bm * PBM + 16 * wave + lr, p.M - 1) * 64u + lslot;
    r.a1 = (unsigned)min(bm * PBM + 16 * (wave + 8) + lr, p.M - 1) * 64u + lslot;
    r.w = (unsigned)min(bn * PBN + 16 * wave + lr, p.N - 1) * 64u + lslot;
    return r;
  };
  const size_t a_ts = (size_t)p.a_rows * 64, w_ts = (size_t)p.w_rows * 64;
#define PB_PIECE(kc, T, buf, q)                                                                                                  \
  {                                                                                                                              \
    const int t_ = (q) / 3, k_ = (q) % 3;                                                                                        \
    char* lb_ = ldsp + (buf) * P_BUF;                                                                                            \
    if (k_ == 2) {                                                                                                               \
      const char* ws_ = reinterpret_cast<const char*>(p.Wp) + ((size_t)(kc) * 3 + t_) * w_ts;                                    \
      __builtin_amdgcn_global_load_lds((gptr_t)(ws_ + T.w), (lptr_t)(lb_ + 3 * P_APL + t_ * P_WPL + wave * 1024), 16, 0, 0);     \
    } else {                                                                                                                     \
      const char* as_ = reinterpret_cast<const char*>(p.Ap) + ((size_t)(kc) * 3 + t_) * a_ts;                                    \
      __builtin_amdgcn_global_load_lds((gptr_t)(as_ + (k_ ? T.a1 : T.a0)), (lptr_t)(lb_ + t_ * P_APL + (wave + 8 * k_) * 1024), 16, 0, 0); \
    }                                                                                                                            \
  }
#define PB_PIECES3(kc, T, buf, q0) PB_PIECE(kc, T, buf, q0) PB_PIECE(kc, T, buf, q0 + 1) PB_PIECE(kc, T, buf, q0 + 2)

  const int fr = lane & 15, fq = lane >> 4;
  int a_fo[4], w_fo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wm * 64 + 16 * i + fr;
    a_fo[i] = r * 64 + ((fq ^ ((0 - (r >> 2)) & 3)) << 4);
    const int c = wn * 64 + 16 * i + fr;
    w_fo[i] = 3 * P_APL + c * 64 + ((fq ^ ((0 - (c >> 2)) & 3)) << 4);
  }
  const int nk = p.K >> 5;

  int f_tile = tile, f_kc = 0;                                             // fetch cursor: two chunks ahead of the products
  PTile f_rows = tile_rows(tile);
#define PB_ADVANCE                                   \
  if (++f_kc == nk) {                                \
    f_kc = 0;                                        \
    if (f_tile + tile_step < tile_end) {             \
      f_tile += tile_step;                           \
      f_rows = tile_rows(f_tile);                    \
    }                                                \
  }
  PB_PIECES3(f_kc, f_rows, 0, 0) PB_PIECES3(f_kc, f_rows, 0, 3) PB_PIECES3(f_kc, f_rows, 0, 6)
  PB_ADVANCE
  PB_PIECES3(f_kc, f_rows, 1, 0) PB_PIECES3(f_kc, f_rows, 1, 3) PB_PIECES3(f_kc, f_rows, 1, 6)
  PB_ADVANCE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

#define C_READ_A(term, f, base) \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) f[i] = *reinterpret_cast<const bf16x8*>((base) + (term) * P_APL + a_fo[i]);
#define C_READ_W(term, f, base) \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) f[i] = *reinterpret_cast<const bf16x8*>((base) + (term) * P_WPL + w_fo[i]);
#define P_FENCE __builtin_amdgcn_sched_barrier(0);
#define C_MMA(fwp, fap)                                                                                       \
  {                                                                                                           \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j)               \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fwp[j], fap[i], acc[i][j], 0, 0, 0);              \
  }
  // a product with the DMA pieces q0, q0 + 1, q0 + 2 of this step's fetch (-> buffer buf) behind its MFMAs 4, 8 and 12
#define C_MMA_D(fwp, fap, buf, q0)                                                                            \
  {                                                                                                           \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) {             \
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fwp[j], fap[i], acc[i][j], 0, 0, 0);                \
      if (!(V & 2) && j == 3 && i < 3) { P_FENCE PB_PIECE(l_kc, l_rows, buf, q0 + i) P_FENCE }                \
    }                                                                                                         \
  }
  // every operand set of the chunk in buffer B but w_hi (read with product 1)
#define C_LOAD_ALL(B, AH)                            \
  {                                                  \
    const char* nb_ = ldsp + (B) * P_BUF;            \
    C_READ_A(0, AH, nb_) C_READ_W(2, w_lo, nb_) C_READ_A(2, a_lo, nb_) C_READ_A(1, a_mid, nb_) C_READ_W(1, w_mid, nb_) \
  }
  int kc = 0, par = 0;
  bf16x8 a_hi[4], a_hn[4], w_hi[4], a_mid[4], w_mid[4], w_lo[4], a_lo[4];
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  C_LOAD_ALL(0, a_hi)
  for (;;) {                                                               // one chunk per iteration, in buffer par
    const char* rb = ldsp + par * P_BUF;
    const char* nb = ldsp + (par ^ 1) * P_BUF;
    C_READ_W(0, w_hi, rb)
    P_FENCE
    C_MMA(w_lo, a_hi) /* (hi, lo) */
    P_FENCE
    C_MMA(w_hi, a_lo) /* (lo, hi) */
    P_FENCE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int l_kc = f_kc;                                                 // this step's fetch: the chunk after next -> buffer par
    const PTile l_rows = f_rows;
    PB_ADVANCE
    const bool last = kc + 1 == nk;                                        // block-uniform
    if (!last) { C_READ_W(2, w_lo, nb) }
    P_FENCE
    C_MMA_D(w_mid, a_mid, par, 0) /* (mid, mid) */
    if (!last) { C_READ_A(2, a_lo, nb) }
    P_FENCE
    C_MMA_D(w_hi, a_mid, par, 3) /* (mid, hi) */
    if (!last) { C_READ_A(1, a_mid, nb) }
    P_FENCE
    C_MMA_D(w_mid, a_hi, par, 6) /* (hi, mid) */
    if (!last) { C_READ_W(1, w_mid, nb) C_READ_A(0, a_hn, nb) }
    P_FENCE
    C_MMA(w_hi, a_hi) /* (hi, hi) */
    P_FENCE
    par ^= 1;
    if (last) {
      const int bm = tile / nbn, bn = tile - bm * nbn;
      if (p.Cp) planes_epilogue(p, acc, bm * PBM + wm * 64, bn * PBN + wn * 64, fr, fq);
      if (p.C) gemm_epilogue<4, 4>(p, acc, bm * PBM, bn * PBN, wm * 64, wn * 64, fr, fq);
      tile += tile_step;
      if (tile >= tile_end) break;
      kc = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      C_LOAD_ALL(par, a_hi)
    } else {
      ++kc;
#pragma unroll
      for (int i = 0; i < 4; ++i) a_hi[i] = a_hn[i];                       // 16 v_mov per chunk buy one code path for both buffers
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // no DMA may outlive the workgroup's LDS
#undef PB_PIECE
#undef PB_PIECES3
#undef PB_ADVANCE
#undef P_FENCE
#undef C_READ_A
#undef C_READ_W
#undef C_MMA
#undef C_MMA_D
#undef C_LOAD_ALL
}

}  // namespace

hipError_t launch_split_planes(const float* x, int ld, unsigned short* planes, long long rows, int M, int K, hipStream_t s) {
  if (!x || !planes || M <= 0 || K <= 0 || (K & 31) || (ld & 3) || rows < M) return hipErrorInvalidValue;
  const size_t n = (size_t)M * (K >> 3);
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, ld, planes, rows, M, K);
  return hipGetLastError();
}

bool gemm_planes_supported(const GemmParams& p) {
  // epilogues: fp32 output through gemm_tile.h's straight-line one (bias / activation / residual) or the mask head's; and / or
  // the plane output (bias / activation only)
  const bool fast = !p.C2 && !(p.N & 3) && !(p.ldc & 3) && (!p.R || !(p.ldr & 3));
  const bool mask = p.C2 && p.X && !(p.N & 1) && !(p.ldc & 1) && !p.R;
  const bool c_ok = !p.C || fast || mask;
  const bool cp_ok = !p.Cp || (!(p.N & 31) && p.c_rows >= p.M && !p.R && !p.C2 && p.act != ACT_SIGMOID);
  return p.Ap && p.Wp && (p.C || p.Cp) && c_ok && cp_ok && p.amode == AMODE_PLAIN && p.a_rows >= p.M && p.w_rows >= p.N &&
         !p.lnx_c1 && !p.ln_gamma && !p.ln_stats && p.ksplit <= 1 && p.mag_F == 0 && p.drop_p <= 0.0f && !(p.K & 31) &&
         p.alt.M <= 0 && !p.epi_general;
}

[[maybe_unused]] constexpr int PLANES_V = 0;                                               // the shipped variant of gemm_planes_kernel_b
const char* gemm_planes_instance_name() { return "gemm_planes_kernel_b<0>"; }
#ifdef AVSEP_DEV
static const void* const planes_kernels[] = {
    reinterpret_cast<const void*>(gemm_planes_kernel<0>), reinterpret_cast<const void*>(gemm_planes_kernel<1>),
    reinterpret_cast<const void*>(gemm_planes_kernel<4>),
    reinterpret_cast<const void*>(gemm_planes_kernel_b<0>), reinterpret_cast<const void*>(gemm_planes_kernel_b<1>),
    reinterpret_cast<const void*>(gemm_planes_kernel_b<2>), reinterpret_cast<const void*>(gemm_planes_kernel_b<4>),
    reinterpret_cast<const void*>(gemm_planes_kernel_b<8>), reinterpret_cast<const void*>(gemm_planes_kernel_b<9>),
    reinterpret_cast<const void*>(gemm_planes_kernel_c<0>), reinterpret_cast<const void*>(gemm_planes_kernel_c<2>),
    reinterpret_cast<const void*>(gemm_planes_kernel_b<16>), reinterpret_cast<const void*>(gemm_planes_kernel_b<32>)};
#else
static const void* const planes_kernels[] = {reinterpret_cast<const void*>(gemm_planes_kernel_b<PLANES_V>)};
#endif

hipError_t launch_gemm_planes(GemmParams p, hipStream_t s) {
  if (!gemm_planes_supported(p) || p.M <= 0 || p.N <= 0 || p.K <= 0) return hipErrorInvalidValue;
  p.nbn_magic = 0;
  if (!p.W) p.W = reinterpret_cast<const float*>(p.Wp);   // gemm_tile.h's epilogue reads N floats from W when there is no bias (discarded)
  // the dynamic-LDS ceiling of the kernel is raised once per device (see conv_stack.hip); the CU count is read with it
  static bool raised[64] = {};
  static int cus[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (!raised[dev]) {
    for (const void* k : planes_kernels) {
      const hipError_t attr = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P_BUF);
      if (attr != hipSuccess) return attr;
    }
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return hipErrorInvalidDevice;
    cus[dev] = n;
    raised[dev] = true;
  }
  const long tiles = (long)((p.M + PBM - 1) / PBM) * ((p.N + PBN - 1) / PBN);
  const long grid = tiles < cus[dev] ? tiles : cus[dev];                 // one resident workgroup per CU walks tiles / grid tiles
  int v = -1;                                                              // the shipped kernel
#ifdef AVSEP_DEV
  if (getenv("AVSEP_PLANES_DBG")) {   // developer diagnostics: where a workgroup's life goes (wall-clock stamps, 10 ns ticks)
    unsigned long long* buf = nullptr;
    if (hipMalloc(&buf, (size_t)grid * 32 * 8) != hipSuccess) return hipErrorOutOfMemory;
    (void)hipMemsetAsync(buf, 0, (size_t)grid * 32 * 8, s);
    p.dbg = buf;
    hipLaunchKernelGGL(gemm_planes_kernel_b<16>, dim3((unsigned)grid), dim3(512), 2 * P_BUF, s, p);
    (void)hipStreamSynchronize(s);
    std::vector<unsigned long long> h((size_t)grid * 32);
    (void)hipMemcpy(h.data(), buf, h.size() * 8, hipMemcpyDeviceToHost);
    (void)hipFree(buf);
    unsigned long long t0 = ~0ull, t1 = 0, e_max = 0;
    for (long b = 0; b < grid; ++b) { t0 = std::min(t0, h[b * 32]); e_max = std::max(e_max, h[b * 32]); }
    double pro = 0, loop = 0, epi = 0, clk = 0; long nt = 0; double life = 0;
    unsigned long long end_min = ~0ull;
    for (long b = 0; b < grid; ++b) {
      const unsigned long long* r = &h[b * 32];
      pro += (double)(r[1] - r[0]);
      unsigned long long prev = r[1], last = r[1];
      for (int t = 0; t < 14 && r[2 + 2 * t]; ++t) {
        loop += (double)(r[2 + 2 * t] - prev); epi += (double)(r[3 + 2 * t] - r[2 + 2 * t]); prev = r[3 + 2 * t]; last = prev; ++nt;
      }
      t1 = std::max(t1, last); end_min = std::min(end_min, last); life += (double)(last - r[0]);
      clk += (double)(r[31] - r[30]) / ((double)(last - r[0]) * 10.0);     // cycles per ns = GHz
    }
    fprintf(stderr, "[planes dbg] M=%d N=%d K=%d grid %ld: span %.2f us (first entry -> last epilogue issued); entry skew %.2f us, end skew %.2f us; mean workgroup life %.2f us; "
            "mean prologue %.2f us; per tile: loop %.2f us (%.3f us per chunk), epilogue %.2f us; %ld tiles; in-kernel shader clock %.3f GHz\n",
            p.M, p.N, p.K, grid, (t1 - t0) / 100.0, (e_max - t0) / 100.0, (t1 - end_min) / 100.0, life / grid / 100.0, pro / grid / 100.0,
            loop / nt / 100.0, loop / nt / 100.0 / (p.K / 32), epi / nt / 100.0, nt, clk / grid);
    return hipGetLastError();
  }
  if (const char* e = getenv("AVSEP_PLANES_V")) v = atoi(e);               // developer A/B: 0, 1, 4 = first kernel; 8 + V = gemm_planes_kernel_b<V>
#define PL_(K) hipLaunchKernelGGL(K, dim3((unsigned)grid), dim3(512), 2 * P_BUF, s, p); break;
  switch (v) {
    case 1: PL_(gemm_planes_kernel<1>)
    case 4: PL_(gemm_planes_kernel<4>)
    case 8: PL_(gemm_planes_kernel_b<0>)
    case 9: PL_(gemm_planes_kernel_b<1>)
    case 10: PL_(gemm_planes_kernel_b<2>)
    case 12: PL_(gemm_planes_kernel_b<4>)
    case 16: PL_(gemm_planes_kernel_b<8>)
    case 17: PL_(gemm_planes_kernel_b<9>)
    case 40: PL_(gemm_planes_kernel_b<32>)
    case 32: PL_(gemm_planes_kernel_c<0>)
    case 34: PL_(gemm_planes_kernel_c<2>)
    case 0: PL_(gemm_planes_kernel<0>)
    default: PL_(gemm_planes_kernel_b<0>)
  }
#undef PL_
#else
  (void)v;
  hipLaunchKernelGGL(gemm_planes_kernel_b<PLANES_V>, dim3((unsigned)grid), dim3(512), 2 * P_BUF, s, p);
#endif
  return hipGetLastError();
}
