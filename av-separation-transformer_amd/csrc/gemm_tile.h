// gemm_tile.h -- the tile-level device code of the fp32 matrix-core GEMM family (see gemm.hip for the design notes): fused
// epilogues, the MFMA chunk, and the whole tile body `gemm_tile` that `gemm_kernel` (gemm.hip) wraps one-to-one and that the
// dependency-driven step kernel (chain.hip) calls for the tiles of its work list.  Everything sits in an anonymous namespace:
// each translation unit that includes this file gets its own copy.
#pragma once
#include "kernels.h"

namespace {

// "Coherent" accesses for tiles that exchange data with other workgroups INSIDE one launch (chain.hip): write-through stores
// and L1-bypassing loads (buffer_* ... sc1, aux = 16).  A CU's vector L1 is never refreshed by another CU's stores and the XCDs'
// L2s are not coherent with each other (MI355X_MICROARCH.md, inter-workgroup visibility): a handed-off byte is stored sc1,
// drained (s_waitcnt vmcnt(0)) before its producer signals, and EVERY load of it is an sc1 load -- then no release / acquire
// fence is needed (cdna_hip_programming.md Guideline 16, R1).  COH = 1: both.  COH = 2: sc1 loads, PLAIN stores -- for producer and
// consumer workgroups that are known to sit on the SAME XCD (chain.hip's XCD-local queues): a plain store keeps its line in that
// XCD's L2, the one coherence point of its CUs, where the consumer's L1-bypassing load finds it (an sc1 store DROPS the line from
// L2, and every later load of it pays the trip to memory).  COH = 0 everywhere else: plain loads and stores.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t coh_rsrc(const void* base) {          // base must be wave-uniform
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ f32x4 coh_load16(__amdgpu_buffer_rsrc_t r, int byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16));
}
__device__ __forceinline__ void coh_store16(__amdgpu_buffer_rsrc_t r, int byte_off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, 16);
}

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case ACT_RELU: return fmaxf(v, 0.0f);
    case ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));   // nn.GELU() exact form
    case ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    default: return v;
  }
}

// Workgroups are dealt round-robin over the 8 XCDs, each with a private L2, so in launch order every XCD
// would touch every A row block and pull its own copy through the fabric.  Remap (bijective for any grid
// size) so XCD x owns a contiguous range of tiles = a contiguous range of A rows: A crosses the fabric once
// instead of 8 times; only W is shared by all XCDs.  Speed only, never correctness.
__device__ __forceinline__ int xcd_tile(const GemmParams& p) {
  int tile = blockIdx.x;
  if (!p.no_xcd_remap) {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = tile & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (tile >> 3);
  }
  return tile;
}

// Pair launch (GemmParams::alt): virtual tiles past g_tiles0 belong to the second problem -- same N, K, leading dimensions
// and epilogue, other operands and row count.  Block-uniform: a handful of scalar selects at kernel entry.
__device__ __forceinline__ void select_pair(GemmParams& p, int& tile) {
#ifdef AVSEP_DEV
  if (p.g_tiles0 > 0 && tile >= p.g_tiles0) {
    tile -= p.g_tiles0;
    p.A = p.alt.A; p.W = p.alt.W; p.bias = p.alt.bias; p.R = p.alt.R; p.rperiod = p.alt.rperiod;
    p.ln_gamma = p.alt.ln_gamma; p.ln_beta = p.alt.ln_beta;
    p.C = p.alt.C; p.M = p.alt.M;
  }
#endif
}

// Fused epilogue shared by the GEMM kernels.  The MFMAs are issued with the operands swapped (A operand = W rows,
// B operand = activation rows, see mfma_chunk), so the 16x16 C/D layout -- col = lane&15, row = 4*(lane>>4)+reg --
// holds, for output row m = lane&15, FOUR CONSECUTIVE COLUMNS n = 4*(lane>>4)+reg per accumulator: one float4 store
// per MFMA block and lane, bias / residual read as float4 too.  The straightforward orientation (4 consecutive rows
// per lane) costs one 4-byte store per element with 64-byte row pieces and was store-ISSUE bound: ~1/5 of a 64x64
// tile's life (in-kernel clocks), 3-5 % of every step.  Same products in the same order: results are bit-identical.
// `mw`, `nw`: the wave's offset inside the tile.
template <int ACT>
__device__ __forceinline__ float act_t(float v) {
  if (ACT == ACT_RELU) return fmaxf(v, 0.0f);
  if (ACT == ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  if (ACT == ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
  return v;
}

// erf / exp expand to dozens of instructions per element.  Inlined into the unrolled epilogue, hipcc interleaves the
// chains of all WBM*WBN blocks and the 128x64 tile needs 213 registers -- two workgroups per CU instead of three
// (hipOccupancyMaxActiveBlocksPerMultiprocessor and the resident-workgroup census both showed 2) -- for code that runs
// once per tile.  Out of line, one float4 at a time, the kernel stays at 146.
template <int ACT>
__device__ __attribute__((noinline)) f32x4 act4_outofline(f32x4 v) {
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = act_t<ACT>(v[e]);
  return v;
}

// Training-only epilogue step (GemmParams::drop_p), out of line for the same reason: the 64-bit hash of the mask must not
// cost the inference instances registers.
__device__ __attribute__((noinline)) f32x4 drop4_outofline(f32x4 v, unsigned long long seed, unsigned long long e0, float p) {
  const float ks = dropout_scale(p);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = dropout_keep(seed, e0 + e, p) ? v[e] * ks : 0.0f;
  return v;
}

// Straight-line epilogue for N % 4 == 0 (see the note on gemm_epilogue): every load of the tile is UNCONDITIONAL
// (out-of-range rows / columns are clamped to the tile's last valid row / float4, a missing bias or residual reads the
// same addresses of whatever pointer is valid and is discarded by a select), then the arithmetic, then the stores.
// (Issuing these loads EARLIER was measured three ways, all losses: before the K loop, behind the first chunks, -15 % on
// the 32-clip step (they sit in the same in-order queue as the loop's staging loads, which then wait for cold residual
// rows); conditionally before the last chunk inside the tail loop, -4...-9 %; before a peeled last chunk in straight-line
// code, -3.5 % cfg2 / -8 % cfg3 and cfg5 -- profiles/r02_ab_epilogue_early_loads.txt.)
template <int WBM, int WBN>
struct EpiOperands {
  f32x4 bv[WBN], rv[WBM][WBN];
};

__device__ __forceinline__ bool epilogue_is_fast(const GemmParams& p) {      // block-uniform
  return !(p.N & 3) && !(p.ldc & 3) && (!p.R || !(p.ldr & 3)) && !p.C2 && !p.epi_general && p.mag_F == 0;
}

// HAS_R = false: no residual / positional rows -- no second operand is fetched at all.  (The first straight-line version
// fetched the C tile itself as a stand-in to keep one code path: 32 KB per 128x64 tile of never-used, HBM-cold reads
// that the stores then waited behind -- the QKV / FFN-1 / decoder GEMMs, 70 % of the large configs' flops, have no
// residual.  profiles/r02_ab_epilogue_no_residual_fetch.txt)
template <int WBM, int WBN, bool HAS_R, int COH = 0>
__device__ __forceinline__ void epilogue_load(const GemmParams& p, EpiOperands<WBM, WBN>& o, int mbase, int nbase, int fr,
                                              int fq) {
  const bool has_b = p.bias != nullptr;                                     // block-uniform
  const float* bsrc = has_b ? p.bias : p.W;                               // W: at least N*K >= N floats, always readable
#pragma unroll
  for (int j = 0; j < WBN; ++j) o.bv[j] = *reinterpret_cast<const f32x4*>(bsrc + min(nbase + 16 * j + 4 * fq, p.N - 4));
  if (HAS_R) {
#pragma unroll
    for (int i = 0; i < WBM; ++i) {
      const int mc = min(mbase + 16 * i + fr, p.M - 1);
      const int rr = p.rperiod > 0 ? (mc % p.rperiod) : mc;
#pragma unroll
      for (int j = 0; j < WBN; ++j) {
        if (COH) o.rv[i][j] = coh_load16(coh_rsrc(p.R), (rr * p.ldr + min(nbase + 16 * j + 4 * fq, p.N - 4)) * 4);
        else o.rv[i][j] = *reinterpret_cast<const f32x4*>(p.R + (size_t)rr * p.ldr + min(nbase + 16 * j + 4 * fq, p.N - 4));
      }
    }
  }
}

template <int WBM, int WBN, int ACT, bool HAS_R, int COH = 0>
__device__ __forceinline__ void epilogue_finish(const GemmParams& p, const f32x4 (&acc)[WBM][WBN], EpiOperands<WBM, WBN>& o,
                                                int mbase, int nbase, int fr, int fq) {
  const bool has_b = p.bias != nullptr;
#pragma unroll
  for (int i = 0; i < WBM; ++i)
#pragma unroll
    for (int j = 0; j < WBN; ++j) {
      f32x4 x;
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = has_b ? acc[i][j][e] + o.bv[j][e] : acc[i][j][e];
      if (ACT == ACT_GELU || ACT == ACT_SIGMOID) {
        x = act4_outofline<ACT>(x);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = act_t<ACT>(x[e]);
      }
      if (p.drop_p > 0.0f)                          // block-uniform; training only (see GemmParams::drop_p)
        x = drop4_outofline(x, p.drop_seed, (unsigned long long)(mbase + 16 * i + fr) * p.N + (nbase + 16 * j + 4 * fq), p.drop_p);
#pragma unroll
      for (int e = 0; e < 4; ++e) o.rv[i][j][e] = HAS_R ? x[e] + o.rv[i][j][e] : x[e];
    }
#pragma unroll
  for (int i = 0; i < WBM; ++i)
#pragma unroll
    for (int j = 0; j < WBN; ++j) {
      const int m = mbase + 16 * i + fr, n = nbase + 16 * j + 4 * fq;
      if (m < p.M && n < p.N) {
        if (COH == 1) coh_store16(coh_rsrc(p.C), (m * p.ldc + n) * 4, o.rv[i][j]);
        else *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + n) = o.rv[i][j];
      }
    }
}

template <int WBM, int WBN, bool HAS_R, int COH = 0>
__device__ __forceinline__ void epilogue_finish_act(const GemmParams& p, const f32x4 (&acc)[WBM][WBN],
                                                    EpiOperands<WBM, WBN>& o, int mbase, int nbase, int fr, int fq) {
  switch (p.act) {                                // block-uniform: one straight-line expansion per activation
    case ACT_RELU: epilogue_finish<WBM, WBN, ACT_RELU, HAS_R, COH>(p, acc, o, mbase, nbase, fr, fq); break;
    case ACT_GELU: epilogue_finish<WBM, WBN, ACT_GELU, HAS_R, COH>(p, acc, o, mbase, nbase, fr, fq); break;
    case ACT_SIGMOID: epilogue_finish<WBM, WBN, ACT_SIGMOID, HAS_R, COH>(p, acc, o, mbase, nbase, fr, fq); break;
    default: epilogue_finish<WBM, WBN, ACT_NONE, HAS_R, COH>(p, acc, o, mbase, nbase, fr, fq); break;
  }
}

// Straight-line epilogue of the mask head (decoder.decoder.3 + Sigmoid + SeparationDecoder.separate, model.py:195-220):
// N = S * F is even but not a multiple of 4 (F = 257), two outputs -- C = act(acc + bias), C2 = C * X[m][n % F] -- in 8-byte
// pairs.  Same arithmetic as the block-by-block general path below (tested bit for bit against it), in the order of the
// fast path: every load of the tile (bias pairs, the mixture's magnitudes), the arithmetic, then nothing but stores.
template <int WBM, int WBN>
__device__ __forceinline__ void mask_epilogue(const GemmParams& p, const f32x4 (&acc)[WBM][WBN], int mbase, int nbase, int fr,
                                              int fq) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const bool has_b = p.bias != nullptr;                                     // block-uniform
  const float* bsrc = has_b ? p.bias : p.W;
  f32x2 b0[WBN], b1[WBN];
  float xv[WBM][WBN][4];
#pragma unroll
  for (int j = 0; j < WBN; ++j) {
    const int n = nbase + 16 * j + 4 * fq;
    b0[j] = *reinterpret_cast<const f32x2*>(bsrc + min(n, p.N - 2));
    b1[j] = *reinterpret_cast<const f32x2*>(bsrc + min(n + 2, p.N - 2));
  }
#pragma unroll
  for (int i = 0; i < WBM; ++i) {
    const float* xrow = p.X + (size_t)min(mbase + 16 * i + fr, p.M - 1) * p.ldx;
#pragma unroll
    for (int j = 0; j < WBN; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) xv[i][j][e] = xrow[min(nbase + 16 * j + 4 * fq + e, p.N - 1) % p.F];
  }
  f32x4 c[WBM][WBN], c2[WBM][WBN];
#pragma unroll
  for (int i = 0; i < WBM; ++i)
#pragma unroll
    for (int j = 0; j < WBN; ++j) {
      f32x4 x = acc[i][j];
      if (has_b) { x[0] += b0[j][0]; x[1] += b0[j][1]; x[2] += b1[j][0]; x[3] += b1[j][1]; }
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = apply_act(x[e], p.act);
      c[i][j] = x;
#pragma unroll
      for (int e = 0; e < 4; ++e) c2[i][j][e] = x[e] * xv[i][j][e];
    }
#pragma unroll
  for (int i = 0; i < WBM; ++i)
#pragma unroll
    for (int j = 0; j < WBN; ++j) {
      const int m = mbase + 16 * i + fr, n = nbase + 16 * j + 4 * fq;
      if (m < p.M && n < p.N) {
        float* d = p.C + (size_t)m * p.ldc + n;
        float* d2 = p.C2 + (size_t)m * p.ldc + n;
        *reinterpret_cast<f32x2*>(d) = f32x2{c[i][j][0], c[i][j][1]};
        *reinterpret_cast<f32x2*>(d2) = f32x2{c2[i][j][0], c2[i][j][1]};
        if (n + 2 < p.N) {
          *reinterpret_cast<f32x2*>(d + 2) = f32x2{c[i][j][2], c[i][j][3]};
          *reinterpret_cast<f32x2*>(d2 + 2) = f32x2{c2[i][j][2], c2[i][j][3]};
        }
      }
    }
}

// ORDER MATTERS: on CDNA loads and stores share one in-order counter (vmcnt), so a load issued after a store cannot be
// waited for without also waiting for that store to be acknowledged by memory (1-2 us under load).  The first version
// of this epilogue went block by block -- load bias / residual, compute, store -- and every block's loads waited for
// the previous block's stores: 17.8 us of a 128x64 tile's 58.6 us life (in-kernel stamps, AVSEP_GEMM_DBG;
// profiles/r02_gemm_phase_stamps.txt) -- and with loads on some control-flow path behind the stores, hipcc also drains
// the counter (s_waitcnt vmcnt(0)) after every pair of stores.  The fast path below is straight-line code: ALL loads
// of the tile (bias, residual / positional rows), then the arithmetic, then nothing but stores.
template <int WBM, int WBN, int COH = 0>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, const f32x4 (&acc)[WBM][WBN], int m0, int n0, int mw,
                                              int nw, int fr, int fq) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  if (p.mag_F > 0) {                            // block-uniform: STFT magnitude, (re, im) column pairs -> |.| at (b, f, t)
#pragma unroll
    for (int i = 0; i < WBM; ++i) {
      const int m = m0 + mw + 16 * i + fr;
      if (m >= p.M) continue;
      const int b = m / p.T, t = m - b * p.T;
#pragma unroll
      for (int j = 0; j < WBN; ++j) {
        const int n = n0 + nw + 16 * j + 4 * fq;
        if (n >= p.N) continue;
        const f32x4 v = acc[i][j];
        const int f = n >> 1;
        float* dst = p.C + ((size_t)b * p.mag_F + f) * p.T + t;
        dst[0] = sqrtf(v[0] * v[0] + v[1] * v[1]);
        if (f + 1 < p.mag_F) dst[p.T] = sqrtf(v[2] * v[2] + v[3] * v[3]);
      }
    }
    return;
  }
  const bool v4 = !(p.N & 3) && !(p.ldc & 3) && (!p.R || !(p.ldr & 3));   // block-uniform
  const bool v2 = !(p.N & 1) && !(p.ldc & 1);
  if (epilogue_is_fast(p)) {                      // every GEMM of the model but the mask head
    EpiOperands<WBM, WBN> o;
    if (p.R) {                                    // block-uniform: two straight-line expansions
      epilogue_load<WBM, WBN, true, COH>(p, o, m0 + mw, n0 + nw, fr, fq);
      epilogue_finish_act<WBM, WBN, true, COH>(p, acc, o, m0 + mw, n0 + nw, fr, fq);
    } else {
      epilogue_load<WBM, WBN, false, COH>(p, o, m0 + mw, n0 + nw, fr, fq);
      epilogue_finish_act<WBM, WBN, false, COH>(p, acc, o, m0 + mw, n0 + nw, fr, fq);
    }
    return;
  }
  if (COH) return;   // coherent tiles exist for the fast epilogue only (chain.hip checks its ops on the host)
  if (p.C2 && v2 && !p.R && p.drop_p <= 0.0f && !p.epi_general) {   // block-uniform: the mask head
    mask_epilogue<WBM, WBN>(p, acc, m0 + mw, n0 + nw, fr, fq);
    return;
  }
  // ---- general path: N not a multiple of 4 and / or a residual beside the second output, block by block
#pragma unroll
  for (int i = 0; i < WBM; ++i) {
    const int m = m0 + mw + 16 * i + fr;
    if (m >= p.M) continue;
    const int rr = p.rperiod > 0 ? (m % p.rperiod) : m;
#pragma unroll
    for (int j = 0; j < WBN; ++j) {
      const int n = n0 + nw + 16 * j + 4 * fq;
      if (n >= p.N) continue;
      f32x4 v = acc[i][j];
      float w[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool in = n + e < p.N;
        float x = (p.bias && in) ? v[e] + p.bias[n + e] : v[e];
        x = apply_act(x, p.act);
        if (p.drop_p > 0.0f)
          x = dropout_keep(p.drop_seed, (unsigned long long)m * p.N + (n + e), p.drop_p) ? x * dropout_scale(p.drop_p) : 0.0f;
        if (p.R && in) x += p.R[(size_t)rr * p.ldr + n + e];
        v[e] = x;
        w[e] = (p.C2 && in) ? x * p.X[(size_t)m * p.ldx + (n + e) % p.F] : 0.0f;
      }
      float* c = p.C + (size_t)m * p.ldc + n;
      float* c2 = p.C2 ? p.C2 + (size_t)m * p.ldc + n : nullptr;
      if (v4) {
        *reinterpret_cast<f32x4*>(c) = v;
        if (c2) *reinterpret_cast<f32x4*>(c2) = f32x4{w[0], w[1], w[2], w[3]};
      } else if (v2) {                            // N even: the pairs (n, n+1), (n+2, n+3) are inside or outside whole
        *reinterpret_cast<f32x2*>(c) = f32x2{v[0], v[1]};
        if (c2) *reinterpret_cast<f32x2*>(c2) = f32x2{w[0], w[1]};
        if (n + 2 < p.N) {
          *reinterpret_cast<f32x2*>(c + 2) = f32x2{v[2], v[3]};
          if (c2) *reinterpret_cast<f32x2*>(c2 + 2) = f32x2{w[2], w[3]};
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < p.N) {
            c[e] = v[e];
            if (c2) c2[e] = w[e];
          }
      }
    }
  }
}

// Epilogue of the LayerNorm-in-the-epilogue GEMM (GemmParams::lnx_c1): y = act( rstd (acc - mean c1) + c2 ) with the tile's row
// statistics in LDS (st[2r] = mean, st[2r+1] = rstd of tile row r).  N % 4 == 0 (checked by the launcher); straight line:
// loads, arithmetic, stores.
template <int WBM, int WBN, int COH = 0>
__device__ __forceinline__ void lnx_epilogue(const GemmParams& p, const f32x4 (&acc)[WBM][WBN], const float* st, int m0, int n0,
                                             int mw, int nw, int fr, int fq) {
  f32x4 c1v[WBN], c2v[WBN];
#pragma unroll
  for (int j = 0; j < WBN; ++j) {
    const int n = min(n0 + nw + 16 * j + 4 * fq, p.N - 4);
    c1v[j] = *reinterpret_cast<const f32x4*>(p.lnx_c1 + n);
    c2v[j] = *reinterpret_cast<const f32x4*>(p.lnx_c2 + n);
  }
  float mu[WBM], rs[WBM];
#pragma unroll
  for (int i = 0; i < WBM; ++i) {
    mu[i] = st[2 * (mw + 16 * i + fr)];
    rs[i] = st[2 * (mw + 16 * i + fr) + 1];
  }
  f32x4 out[WBM][WBN];
#pragma unroll
  for (int i = 0; i < WBM; ++i)
#pragma unroll
    for (int j = 0; j < WBN; ++j) {
      f32x4 x;
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = rs[i] * (acc[i][j][e] - mu[i] * c1v[j][e]) + c2v[j][e];
      switch (p.act) {                              // block-uniform
        case ACT_RELU:
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = fmaxf(x[e], 0.0f);
          break;
        case ACT_GELU: x = act4_outofline<ACT_GELU>(x); break;
        case ACT_SIGMOID: x = act4_outofline<ACT_SIGMOID>(x); break;
        default: break;
      }
      out[i][j] = x;
    }
#pragma unroll
  for (int i = 0; i < WBM; ++i)
#pragma unroll
    for (int j = 0; j < WBN; ++j) {
      const int m = m0 + mw + 16 * i + fr, n = n0 + nw + 16 * j + 4 * fq;
      if (m < p.M && n < p.N) {
        if (COH == 1) coh_store16(coh_rsrc(p.C), (m * p.ldc + n) * 4, out[i][j]);
        else *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + n) = out[i][j];
      }
    }
}

// Sum over the aligned group of 8 lanes a lane belongs to (every lane gets the total; fixed order)
__device__ __forceinline__ float lane8_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));   // row_half_mirror
  return v;
}

// One K-chunk of MFMAs from the LDS image (shared by the GEMM kernels).
// PF = false: fragments of a k-step are read right before its MFMAs (what the compiler schedules best for occupancy).
// PF = true : all fragment reads of step s+1 are issued BEFORE the MFMAs of step s (second register set, pinned with a
//             scheduling barrier), so the matrix pipe does not drain during the LDS round trip of every step.  Costs
//             16 VGPRs on the 64x64 tile (4 -> 3 waves per SIMD): measured a win only for long contractions
//             (16064x512x2048: 292 -> 273 us, 123 TFLOP/s), a loss for K = 512 and for the 128x64 tile.
template <int BK, int WBM, int WBN, bool PF = false>
__device__ __forceinline__ void mfma_chunk(const float* a, const float* b, const int (&a_off)[WBM],
                                           const int (&a_swz)[WBM], const int (&b_off)[WBN], const int (&b_swz)[WBN],
                                           int fq, f32x4 (&acc)[WBM][WBN]) {
  constexpr int S = BK / 16;
  f32x4 fa[PF ? 2 : 1][WBM], fb[PF ? 2 : 1][WBN];
  auto fetch = [&](int s, int set) {
#pragma unroll
    for (int i = 0; i < WBM; ++i)
      fa[set][i] = *reinterpret_cast<const f32x4*>(a + a_off[i] + (((4 * s + fq) ^ a_swz[i]) << 2));
#pragma unroll
    for (int jn = 0; jn < WBN; ++jn)
      fb[set][jn] = *reinterpret_cast<const f32x4*>(b + b_off[jn] + (((4 * s + fq) ^ b_swz[jn]) << 2));
  };
  if (PF) fetch(0, 0);
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int cur = PF ? (s & 1) : 0;
    if (PF) {
      if (s + 1 < S) {
        fetch(s + 1, (s + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      fetch(s, 0);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int i = 0; i < WBM; ++i)
#pragma unroll
        for (int jn = 0; jn < WBN; ++jn)
          acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[cur][jn][c], fa[cur][i][c], acc[i][jn], 0, 0, 0);   // D^T: see gemm_epilogue
  }
}

// 16-byte-slot XOR swizzle: conflict-free ds_write_b128 (8 consecutive lanes = 8 slots of one row) and
// ds_read_b128 (16-lane groups = 16 rows x one logical slot) for 128-, 256- and 512-byte rows.
template <int SLOTS>
__device__ __forceinline__ int swz(int row) {
  // 64-byte rows (SLOTS == 4): 16 rows x one slot span four 16-bank groups four times over -> rows r, r+4, r+8, r+12 take
  // four different slots
  return SLOTS == 4 ? ((row >> 2) & 3) : SLOTS == 8 ? ((row >> 1) & 7) : (row & 15);
}

// Developer diagnostics (AVSEP_GEMM_DBG=1): one lane per workgroup stamps the 100 MHz wall clock at entry, after the
// prologue (first chunk in LDS), after the K loop, after the epilogue's stores are issued and after they have
// drained; launch_gemm() prints the per-phase statistics.  p.dbg is null in every normal launch.
__device__ __forceinline__ void dbg_stamp(const GemmParams& p, int slot) {
  if (p.dbg && threadIdx.x == 0) p.dbg[(size_t)blockIdx.x * 8 + slot] = __builtin_amdgcn_s_memrealtime();
}

// One BM x BN output tile at (m0, n0): the whole body of gemm_kernel, which wraps it one-to-one.  p.M bounds the rows (rows at
// or beyond it are clamped for loads and skipped by the stores).  COH (1 / 2): coherent A / residual loads and C stores (top of this file;
// PLAIN / LNX operand modes, fast and LayerNorm epilogues only).
template <int BM, int BN, int BK, int AMODE, bool PF = false, int RING = 0, int COH = 0>
__device__ __forceinline__ void gemm_tile(GemmParams& p, const int m0, const int n0, float* lds) {
  static_assert(!COH || AMODE == AMODE_PLAIN || AMODE == AMODE_LNX, "coherent tiles: plain or LayerNorm-epilogue A operand");
  constexpr int SLOTS = BK / 4;            // 16-byte slots per LDS row
  constexpr int RPP = 256 / SLOTS;         // rows staged per pass (256 threads x float4)
  constexpr int APASS = BM / RPP;
  constexpr int BPASS = BN / RPP;
  constexpr int WBM = BM / 32;             // 16-row MFMA blocks per wave
  constexpr int WBN = BN / 32;             // 16-col MFMA blocks per wave
  // Register prefetch ring (even depth).  Measured on MI355X (profiles/r01b_*): neither a deeper ring nor a
  // larger BK moves the M ~ 2k shapes -- small tiles sit at the L2 -> CU feed limit (8-11 flop per byte),
  // big tiles lack workgroups -- so the ring stays shallow and cheap in registers.
  constexpr int D = RING ? RING : (APASS + BPASS <= 4) ? 4 : 2;
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the staging pass");
  // lds: 2 * (BM + BN) * BK floats (+ 2 * LN_KMAX for AMODE_LN: gamma | beta of the fused LayerNorm behind the two tile buffers)
  float* As = lds;
  float* Bs = lds + 2 * BM * BK;
  const float* lng = lds + 2 * (BM + BN) * BK;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- staging coordinates: thread -> (row srow + RPP*pass, 16-byte slot sslot) -------------------
  const int srow = tid / SLOTS;
  const int sslot = tid % SLOTS;
  float ln_mu[APASS], ln_rs[APASS];   // AMODE_LN: statistics of this thread's rows (launch_layernorm_stats)
  if (AMODE == AMODE_LN) {
    float* lw = lds + 2 * (BM + BN) * BK;
    for (int i = tid; i < p.K / 4; i += 256) {
      *reinterpret_cast<f32x4*>(lw + 4 * i) = *reinterpret_cast<const f32x4*>(p.ln_gamma + 4 * i);
      *reinterpret_cast<f32x4*>(lw + LN_KMAX + 4 * i) = *reinterpret_cast<const f32x4*>(p.ln_beta + 4 * i);
    }
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      int m = m0 + srow + RPP * i;
      m = m < p.M ? m : p.M - 1;
      ln_mu[i] = p.ln_stats[2 * (size_t)m];
      ln_rs[i] = p.ln_stats[2 * (size_t)m + 1];
    }
  }
  const float* a_src[APASS];
  const float *a_prev[APASS], *a_next[APASS], *a_cur[APASS];   // TAPS3: rows of tap 0 / tap 2 / the tap being loaded
  int a_aux0[APASS], a_aux1[APASS];   // CONV2D: iy0, ix0;  FRAMES: first sample
  int a_st[APASS], b_st[BPASS];       // swizzled LDS float offsets of this thread's staging slots
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    const int r = srow + RPP * i;
    a_st[i] = r * BK + ((sslot ^ swz<SLOTS>(r)) << 2);
    int m = m0 + r;
    m = m < p.M ? m : p.M - 1;
    if (AMODE == AMODE_PLAIN || AMODE == AMODE_LN || AMODE == AMODE_LNX) {
      a_src[i] = p.A + (size_t)m * p.lda + 4 * sslot;
      a_aux0[i] = (m * p.lda + 4 * sslot) * 4;     // COH: byte offset of this thread's float4 from p.A
      a_aux1[i] = 0;
    } else if (AMODE == AMODE_TAPS3) {
      a_src[i] = p.A + (size_t)m * p.lda + 4 * sslot;   // the centre tap's row
      const int t = m % p.T;
      const float* zrow = p.zeros + 4 * sslot;
      a_prev[i] = t > 0 ? a_src[i] - p.lda : zrow;       // tap 0: row t - 1, or zeros in front of the sequence
      a_next[i] = t + 1 < p.T ? a_src[i] + p.lda : zrow; // tap 2: row t + 1, or zeros behind it
      a_cur[i] = a_prev[i];
      a_aux0[i] = a_aux1[i] = 0;
    } else if (AMODE == AMODE_FRAMES) {
      const int b = m / p.T, t = m - b * p.T;
      a_src[i] = p.A + (size_t)b * p.frame_len;        // the clip; the frame offset is kept separately for the bound
      a_aux0[i] = t * p.frame_hop + 4 * sslot;          // sample index of this thread's first float4
      a_aux1[i] = 0;
    } else {
      const int hw = p.Hout * p.Wout;
      const int img = m / hw;
      const int rem = m - img * hw;
      const int y = rem / p.Wout;
      const int x = rem - y * p.Wout;
      a_src[i] = p.A + (size_t)img * p.Hin * p.Win * p.Kt + 4 * sslot;
      a_aux0[i] = 2 * y - 1;
      a_aux1[i] = 2 * x - 1;
    }
  }
  const float* b_src[BPASS];
#pragma unroll
  for (int i = 0; i < BPASS; ++i) {
    const int r = srow + RPP * i;
    b_st[i] = r * BK + ((sslot ^ swz<SLOTS>(r)) << 2);
    int n = n0 + r;
    n = n < p.N ? n : p.N - 1;
    b_src[i] = p.W + (size_t)n * p.ldw + 4 * sslot;
  }

  const int nk = p.K / BK;
  const int cpt = (AMODE == AMODE_PLAIN || AMODE == AMODE_LN || AMODE == AMODE_LNX || AMODE == AMODE_FRAMES) ? nk : (p.Kt / BK);   // chunks per tap
  int tap = 0, sub = 0;                                        // (tap, chunk-in-tap) of the NEXT chunk to load
  int kload = 0;                                               // index of the next chunk to load

  // the ring is statically indexed after unrolling; loads past the last chunk are skipped (block-uniform)
  f32x4 ra[D][APASS], rb[D][BPASS];
  bool rok[D][APASS];   // tap modes: is this row's tap inside the sequence / image?
  auto load_chunk = [&](int slot) {
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      if (AMODE == AMODE_PLAIN || AMODE == AMODE_LN || AMODE == AMODE_LNX) {
        if (COH) ra[slot][i] = coh_load16(coh_rsrc(p.A), a_aux0[i] + kload * (BK * 4));
        else ra[slot][i] = *reinterpret_cast<const f32x4*>(a_src[i] + kload * BK);
      } else if (AMODE == AMODE_TAPS3) {
        // the tap's row, or the zero row (GemmParams::zeros) for a tap outside the sequence: no predicate, no select
        ra[slot][i] = *reinterpret_cast<const f32x4*>(a_cur[i] + sub * BK);
      } else if (AMODE == AMODE_FRAMES) {
        const int smp = a_aux0[i] + kload * BK;          // multiples of 4 throughout: a float4 is inside or outside whole
        const bool ok = smp < p.frame_len;
        ra[slot][i] = *reinterpret_cast<const f32x4*>(a_src[i] + (ok ? smp : 0));
        rok[slot][i] = ok;
      } else {
        const int ky = tap / 3, kx = tap - 3 * ky;
        const int iy = a_aux0[i] + ky, ix = a_aux1[i] + kx;
        const bool ok = (iy >= 0) && (iy < p.Hin) && (ix >= 0) && (ix < p.Win);
        const float* src = a_src[i] + ((size_t)(ok ? iy : 0) * p.Win + (ok ? ix : 0)) * p.Kt + sub * BK;
        ra[slot][i] = *reinterpret_cast<const f32x4*>(src);
        rok[slot][i] = ok;
      }
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) rb[slot][i] = *reinterpret_cast<const f32x4*>(b_src[i] + kload * BK);
    ++kload;
    if (AMODE == AMODE_TAPS3 || AMODE == AMODE_CONV2D) {
      if (++sub == cpt) {                                 // block-uniform: next tap
        sub = 0; ++tap;
        if (AMODE == AMODE_TAPS3) {
#pragma unroll
          for (int i = 0; i < APASS; ++i) a_cur[i] = tap == 1 ? a_src[i] : a_next[i];
        }
      }
    }
  };
  // kc = index of the chunk being written (AMODE_LN picks its gamma / beta columns by it)
  // AMODE_LNX: one-pass sums of this thread's rows, taken of x - pilot, where the row's pilot lx_c is the mean of its FIRST
  // CHUNK (32 floats = the 8 lanes that stage the row; set in front of the first store_chunk).  The shifted values -- not the
  // raw ones -- are what goes to LDS (round 4): LayerNorm(x) = LayerNorm(x - pilot) exactly, and by Cauchy-Schwarz
  // |mean(x) - pilot| <= sqrt(K / 32) std(x) for EVERY row, so the cancellation rstd (acc - mean c1) of the epilogue works on
  // a mean of at most 2.83 standard deviations at K = 256, whatever offset the residual stream carries (round 3 staged the raw
  // rows: its error grew with |mean| / std without bound).  Same VALU count: the subtraction was already made for the sums.
  float lx_c[APASS], lx_s1[APASS], lx_s2[APASS];
#pragma unroll
  for (int i = 0; i < APASS; ++i) lx_s1[i] = lx_s2[i] = lx_c[i] = 0.0f;
  auto store_chunk = [&](int slot, int buf, int kc) {
    float* a = As + buf * BM * BK;
    float* b = Bs + buf * BN * BK;
    f32x4 g4, b4;
    if (AMODE == AMODE_LN) {
      g4 = *reinterpret_cast<const f32x4*>(lng + kc * BK + 4 * sslot);
      b4 = *reinterpret_cast<const f32x4*>(lng + LN_KMAX + kc * BK + 4 * sslot);
    }
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      f32x4 v = ra[slot][i];
      if (AMODE == AMODE_LN) {   // nn.LayerNorm on the way to LDS: the formula (and rounding) of layernorm_kernel
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (v[e] - ln_mu[i]) * ln_rs[i] * g4[e] + b4[e];
      } else if (AMODE == AMODE_LNX) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dlt = v[e] - lx_c[i];
          lx_s1[i] += dlt;
          lx_s2[i] = fmaf(dlt, dlt, lx_s2[i]);
          v[e] = dlt;
        }
      } else if (AMODE != AMODE_PLAIN && AMODE != AMODE_TAPS3) {
        v = rok[slot][i] ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      *reinterpret_cast<f32x4*>(a + a_st[i]) = v;
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) *reinterpret_cast<f32x4*>(b + b_st[i]) = rb[slot][i];
  };

  // ---- fragment read coordinates ----------------------------------------------------------------
  const int fr = lane & 15;   // row inside a 16-row block (A: m, W: n)
  const int fq = lane >> 4;   // k quarter
  int a_off[WBM], a_swz[WBM], b_off[WBN], b_swz[WBN];
#pragma unroll
  for (int i = 0; i < WBM; ++i) {
    const int r = wm * (BM / 2) + 16 * i + fr;
    a_off[i] = r * BK;
    a_swz[i] = swz<SLOTS>(r);
  }
#pragma unroll
  for (int j = 0; j < WBN; ++j) {
    const int r = wn * (BN / 2) + 16 * j + fr;
    b_off[j] = r * BK;
    b_swz[j] = swz<SLOTS>(r);
  }

  f32x4 acc[WBM][WBN];
#pragma unroll
  for (int i = 0; i < WBM; ++i)
#pragma unroll
    for (int j = 0; j < WBN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int j = 0; j < D; ++j)
    if (j < nk) load_chunk(j);
  if (AMODE == AMODE_LN) __syncthreads();   // gamma / beta are in LDS
  if (AMODE == AMODE_LNX) {                 // the rows' pilots: mean of the first chunk, summed in a fixed order (tile-independent)
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      const f32x4 v = ra[0][i];
      lx_c[i] = lane8_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / 32.0f);
    }
  }
  store_chunk(0, 0, 0);
  __syncthreads();
  dbg_stamp(p, 1);

  // Steady state: groups of D chunks in which EVERY refill exists, so the loads are unconditional.  With a
  // conditional load in the loop body the compiler cannot know how many loads are outstanding at the ds_write that
  // consumes the OLDER ring slot and waits for all of them (s_waitcnt vmcnt(3..0) instead of vmcnt(4+)), which
  // collapses the prefetch distance (tools/gemm_anatomy.hip: 105 vs 137 TFLOP/s on the bare loop).  The last
  // groups run the guarded copy of the body.
  int kc0 = 0;
  for (; kc0 + 2 * D <= nk; kc0 += D) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      load_chunk(j);                       // slot j (chunk kc0+j) went to LDS one step ago: refill with chunk kc0+j+D
      mfma_chunk<BK, WBM, WBN, PF>(As + (j & 1) * BM * BK, Bs + (j & 1) * BN * BK, a_off, a_swz, b_off, b_swz, fq, acc);
      store_chunk((j + 1) % D, (j + 1) & 1, kc0 + j + 1);
      __syncthreads();
    }
  }
  for (; kc0 < nk; kc0 += D) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const int kc = kc0 + j;
      if (kc < nk) {                       // block-uniform
        if (kc + D < nk) load_chunk(j);
        // D is even: (kc & 1) == (j & 1)
        mfma_chunk<BK, WBM, WBN, PF>(As + (j & 1) * BM * BK, Bs + (j & 1) * BN * BK, a_off, a_swz, b_off, b_swz, fq, acc);
        // chunk kc+1 (ring slot j+1, loaded D-1 iterations ago) -> the other LDS buffer
        if (kc + 1 < nk) store_chunk((j + 1) % D, (j + 1) & 1, kc + 1);
        __syncthreads();
      }
    }
  }

  dbg_stamp(p, 2);
  if (AMODE == AMODE_LNX) {
    // row statistics: combine the SLOTS lanes that staged a row, park (mean, rstd) per tile row in LDS (free after the loop's
    // last barrier) for the lanes that hold the row's accumulators
    static_assert(AMODE != AMODE_LNX || SLOTS == 8, "the LayerNorm-in-the-epilogue instances have BK = 32 (8 lanes per row)");
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      // the 8 lanes of a row are one half of a DPP row: lane ^ 1, lane ^ 2 (quad permutes), then 7 - lane (half mirror);
      // three v_add_f32 with a DPP operand per sum instead of three ds_bpermute with their address arithmetic
      const float s1 = lane8_sum(lx_s1[i]), s2 = lane8_sum(lx_s2[i]);
      const float inv = p.ln_inv_k;
      const float m1 = s1 * inv;
      const float var = fmaxf(s2 * inv - m1 * m1, 0.0f);
      if (sslot == 0) {
        lds[2 * (srow + RPP * i)] = m1;             // mean of the SHIFTED row, the rows the MFMAs saw
        lds[2 * (srow + RPP * i) + 1] = __builtin_amdgcn_rsqf(var + p.ln_eps);   // v_rsq_f32, 1 ulp (1 / sqrtf: ~35 VALU instructions)
      }
    }
    __syncthreads();
    lnx_epilogue<WBM, WBN, COH>(p, acc, lds, m0, n0, wm * (BM / 2), wn * (BN / 2), fr, fq);
  } else {
    gemm_epilogue<WBM, WBN, COH>(p, acc, m0, n0, wm * (BM / 2), wn * (BN / 2), fr, fq);
  }
  if (p.dbg) {                                                   // block-uniform, diagnostics only
    dbg_stamp(p, 3);
    __builtin_amdgcn_s_waitcnt(0);                               // vmcnt(0): the stores have left
    __syncthreads();
    dbg_stamp(p, 4);
    if (threadIdx.x == 0) p.dbg[(size_t)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
  }
}

// (launch bound for the fragment-prefetch instance: 3 waves per SIMD = 168 registers; it sits at 170 otherwise.  The
// staged-LayerNorm instances whose LDS image admits three workgroups per CU get the same bound: 128x64 sits at 186
// without it, and with it spills 24 registers around -- not inside -- the K loop.)
template <int BM, int BN, int BK, int AMODE, bool PF = false, int RING = 0>
__global__ __launch_bounds__(256, RING ? 4 : AMODE == AMODE_LNX ? ((BM + BN) * BK <= 4096 ? 4 : 3) : (PF || (AMODE == AMODE_LN && (BM + BN) * BK <= 6144)) ? 3 : 1) void gemm_kernel(const GemmParams pin) {
  GemmParams p = pin;
  dbg_stamp(p, 0);
  if (p.ksplit > 1) {                      // block-uniform: slice blockIdx.y of the contraction
    const int z = blockIdx.y;
    p.A += (size_t)z * p.kchunk;
    p.W += (size_t)z * p.kchunk;
    p.C += (size_t)z * p.cstride;
    p.K = min(p.kchunk, p.K - z * p.kchunk);
  }
  __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * BK + (AMODE == AMODE_LN ? 2 * LN_KMAX : 0)];
  const int nbn = (p.N + BN - 1) / BN;
  int tile = xcd_tile(p);
  select_pair(p, tile);
  const int bm = p.nbn_magic ? (int)__umulhi((unsigned)tile, p.nbn_magic) : tile / nbn;   // uniform: s_mul_hi_u32
  const int bn = tile - bm * nbn;
  gemm_tile<BM, BN, BK, AMODE, PF, RING, 0>(p, bm * BM, bn * BN, lds);
}


}  // namespace
