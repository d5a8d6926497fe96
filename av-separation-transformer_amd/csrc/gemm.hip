// gemm.hip -- fp32 matrix-core GEMM family for gfx950 (MI355X).
//
// Every dense contraction of the forward path (>99 % of its FLOPs) runs here:
//   nn.Linear           model.py:93,155-161,195,198 and the in/out projections of MultiheadAttention
//   nn.Conv1d k3 p1     model.py:38,40   as ONE GEMM with K = 3*Cin over a 3-tap shifted A view
//   nn.Conv2d k3 s2 p1  model.py:85,88   as an implicit GEMM (im2col gather done by the staging loads)
//   nn.LayerNorm        fused as an A-operand prologue of the Linear it feeds (gemm_ln_kernel)
//
// Design (CDNA4):
//   * v_mfma_f32_16x16x4_f32: exact fp32 FMA chain at 256 FLOP/clk/CU -- the only matrix path that keeps
//     masks within 1e-4 of the fp32 reference (bf16/fp16 miss it, SURVEY.md §7 "Precision").
//   * 256 threads = 4 wavefronts in a 2x2 grid; wave tile (BM/2)x(BN/2) made of 16x16 MFMA blocks.
//   * K is walked in chunks of BK floats.  A and W chunks are staged global -> registers -> LDS with
//     full-row coalesced float4 loads (a small register ring issued ahead of the MFMAs) and a
//     double-buffered LDS image, one barrier per chunk.
//   * LDS image [row][BK] with the 16-byte slot index XOR-swizzled by the row, which makes both the
//     ds_write_b128 of the staging pass and the ds_read_b128 fragment reads bank-conflict free
//     (SQ_LDS_BANK_CONFLICT = 0 measured).
//   * k-permutation trick: lane (r=l&15, q=l>>4) reads ONE float4 = k {16s+4q .. 16s+4q+3} of its row and
//     feeds component j to MFMA j; A and W use the same permutation so the contraction is unchanged and
//     every fragment read is a single ds_read_b128.
//   * XCD-aware tile order (each of the 8 XCDs has a private L2): XCD x owns a contiguous range of A rows.
//   * Epilogue fused: bias, ReLU / erf-GELU / sigmoid, residual or positional-encoding add, and the
//     sigmoid-mask * mixed product of SeparationDecoder (model.py:207,220) with both outputs written
//     in the reference's (B,T,S,F) memory order.
#include "kernels.h"
#include <algorithm>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "gemm_tile.h"

namespace {

// ---------------------------------------------------------------------------------------------------------
// LDS-DMA form of the plain GEMM (global_load_lds_dwordx4: memory -> LDS with no register in between).
// gemm_kernel stages A / W through a register ring (128x64x32: 48 of its 148 VGPRs) and pays one ds_write_b128 per
// float4; here every wave issues (BM + BN) / (4 * RPI) DMA instructions per chunk, each of which moves RPI = 64 / SLOTS
// whole tile rows (1 KB) straight into the other LDS buffer while the MFMAs of the current chunk run.
//   * the DMA writes lane l at (wave-uniform base) + 16 l bytes, i.e. LINEARLY: row = l / SLOTS, physical 16-byte slot
//     = l % SLOTS.  The LDS image keeps gemm_kernel's XOR swizzle (fragment reads unchanged, conflict-free) by swizzling
//     the SOURCE instead: the lane that fills physical slot p of row r fetches logical slot p ^ swz(r) of that row.
//   * one chunk ahead, two LDS buffers: issue chunk k+1 -> MFMAs of chunk k -> s_waitcnt vmcnt(0) -> barrier.  The barrier
//     that ended iteration k-1 is what makes buffer (k+1)&1 free to overwrite.
//   * same fragment order, same MFMA order, same epilogue as gemm_kernel: bit-identical results.
// PLAIN A operand only (the conv modes zero out-of-range taps on the way to LDS, which a DMA cannot).
template <int BM, int BN, int BK>
__global__ __launch_bounds__(256, (BM + BN) * BK * 8 <= 40 * 1024 ? 4 : 3) void gemm_dma_kernel(const GemmParams pin) {
  GemmParams p = pin;
  constexpr int SLOTS = BK / 4;
  constexpr int RPI = 64 / SLOTS;                  // tile rows per DMA instruction (one wave, 64 lanes x 16 B)
  constexpr int GA = BM / RPI, GB = BN / RPI;      // DMA instructions per chunk for A / for W
  constexpr int JA = GA / 4, JB = GB / 4;          // ... per wave
  constexpr int WBM = BM / 32, WBN = BN / 32;
  static_assert(GA % 4 == 0 && GB % 4 == 0, "row groups must divide over the four waves");
  __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * BK];
  float* As = lds;
  float* Bs = lds + 2 * BM * BK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nbn = (p.N + BN - 1) / BN;
  int tile = xcd_tile(p);
  select_pair(p, tile);
  const int bm = tile / nbn, bn = tile - bm * nbn;
  const int m0 = bm * BM, n0 = bn * BN;

  const int r8 = lane / SLOTS, pslot = lane % SLOTS;
  const float* a_src[JA];
  const float* b_src[JB];
#pragma unroll
  for (int j = 0; j < JA; ++j) {
    const int row = RPI * (wave + 4 * j) + r8;
    const int m = min(m0 + row, p.M - 1);
    a_src[j] = p.A + (size_t)m * p.lda + 4 * (pslot ^ swz<SLOTS>(row));
  }
#pragma unroll
  for (int j = 0; j < JB; ++j) {
    const int row = RPI * (wave + 4 * j) + r8;
    const int n = min(n0 + row, p.N - 1);
    b_src[j] = p.W + (size_t)n * p.ldw + 4 * (pslot ^ swz<SLOTS>(row));
  }
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  auto issue = [&](int kc, int buf) {
#pragma unroll
    for (int j = 0; j < JA; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(a_src[j] + (size_t)kc * BK),
                                       (lptr_t)(As + buf * BM * BK + RPI * (wave + 4 * j) * BK), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < JB; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(b_src[j] + (size_t)kc * BK),
                                       (lptr_t)(Bs + buf * BN * BK + RPI * (wave + 4 * j) * BK), 16, 0, 0);
  };

  const int fr = lane & 15, fq = lane >> 4;
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

  const int nk = p.K / BK;
  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kc = 0; kc < nk; ++kc) {
    if (kc + 1 < nk) issue(kc + 1, (kc + 1) & 1);            // block-uniform
    mfma_chunk<BK, WBM, WBN>(As + (kc & 1) * BM * BK, Bs + (kc & 1) * BN * BK, a_off, a_swz, b_off, b_swz, fq, acc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  gemm_epilogue<WBM, WBN>(p, acc, m0, n0, wm * (BM / 2), wn * (BN / 2), fr, fq);
}

template <int BM, int BN, int BK>
hipError_t launch_dma_t(GemmParams p, hipStream_t s) {
  const int nbn = (p.N + BN - 1) / BN;
  p.g_tiles0 = p.alt.M > 0 ? ((p.M + BM - 1) / BM) * nbn : 0;
  const long rt = (long)(p.M + BM - 1) / BM + (p.alt.M > 0 ? (long)(p.alt.M + BM - 1) / BM : 0);
  hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, BK>), dim3((unsigned)(rt * nbn)), dim3(256), 0, s, p);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Weight gradient without transposes:  dW[n][k] = sum_r dY[r][n] * X[r][k]  (nn.Linear / conv-as-GEMM backward).
// Both operands are "k-major" for this contraction (the contracted index r is the slow one), so a tile's chunk of
// 32 r-rows is fetched as float4s ALONG m / n (coalesced rows of dY and X as they sit in memory) and scattered into
// the same swizzled [row][k] LDS image the other kernels use (4 ds_write_b32 per float4 instead of one b128); the
// MFMA side is mfma_chunk unchanged.  gridDim.y slices the r range (split-K, partial tiles summed by the caller in a
// fixed order).  Rows r >= R contribute zeros; N and K must be multiples of 4 (float4 columns).
struct WgradParams {
  const float* dy;   // [R][ldy]
  const float* x;    // [R][ldx]
  float* dw;         // [N][K] (+ z * N*K per slice)
  int R, N, K, ldy, ldx;
  int rchunk;        // rows per slice (multiple of 32), gridDim.y slices
  int bias;          // 1: also the bias gradient db[n] = sum_r dY[r][n], written behind each slice's N*K weight gradients
                     //    (slice stride N*K + N) by the workgroups of the first K tile -- the column sums of the dY
                     //    chunks they stage anyway, instead of two more launches per layer (colreduce)
  // In-launch merge of the r-slices (gridDim.y > 1): `dw` then holds the slices' partial results, and the workgroup that
  // arrives LAST at a tile's ticket counter sums the slices of that tile in slice order (the order of the former
  // sum_slices launch: bit-identical) into `merged`.  cnt: one zero-initialised counter per tile, left at zero again.
  float* merged;     // [N][K] (+ [N] bias gradients behind it), or null: no merge, the caller sums the slices
  unsigned* cnt;
};

template <int BM, int BN>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
  constexpr int BK = 32, SLOTS = BK / 4;
  constexpr int AV = BM / 4, BV = BN / 4;          // float4 columns per r-row of each operand tile
  constexpr int AL = (BK * AV) / 256, BL = (BK * BV) / 256;   // float4 loads per thread per chunk
  constexpr int WBM = BM / 32, WBN = BN / 32;
  static_assert(AL >= 1 && BL >= 1, "tile too small for 256 threads");
  __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * BK];
  float* As = lds;
  float* Bs = lds + 2 * BM * BK;
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
  f32x4 bsum[AL];
#pragma unroll
  for (int l = 0; l < AL; ++l) bsum[l] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staging: load l of this thread covers r-row kr and float4 column cv of the tile
  int a_kr[AL], a_col[AL], b_kr[BL], b_col[BL];
#pragma unroll
  for (int l = 0; l < AL; ++l) {
    const int idx = tid + 256 * l;
    a_kr[l] = idx / AV;
    a_col[l] = min(m0 + 4 * (idx % AV), p.N - 4);      // clamp: duplicated columns are never stored
  }
#pragma unroll
  for (int l = 0; l < BL; ++l) {
    const int idx = tid + 256 * l;
    b_kr[l] = idx / BV;
    b_col[l] = min(n0 + 4 * (idx % BV), p.K - 4);
  }
  f32x4 ra[2][AL], rb[2][BL];
  auto load_chunk = [&](int kc, int slot) {
    const int r0 = r_begin + kc * BK;
#pragma unroll
    for (int l = 0; l < AL; ++l) {
      const int r = r0 + a_kr[l];
      const f32x4 v = *reinterpret_cast<const f32x4*>(p.dy + (size_t)min(r, p.R - 1) * p.ldy + a_col[l]);
      ra[slot][l] = r < r_end ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int l = 0; l < BL; ++l) {
      const int r = r0 + b_kr[l];
      const f32x4 v = *reinterpret_cast<const f32x4*>(p.x + (size_t)min(r, p.R - 1) * p.ldx + b_col[l]);
      rb[slot][l] = r < r_end ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_chunk = [&](int slot, int buf) {
    float* a = As + buf * BM * BK;
    float* b = Bs + buf * BN * BK;
#pragma unroll
    for (int l = 0; l < AL; ++l) {
      const int idx = tid + 256 * l, kr = idx / AV, mrow = 4 * (idx % AV);
      if (do_bias) bsum[l] += ra[slot][l];            // every chunk is stored exactly once; rows >= r_end hold zeros
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = mrow + e;
        a[row * BK + ((((kr >> 2) ^ swz<SLOTS>(row)) << 2) | (kr & 3))] = ra[slot][l][e];
      }
    }
#pragma unroll
    for (int l = 0; l < BL; ++l) {
      const int idx = tid + 256 * l, kr = idx / BV, nrow = 4 * (idx % BV);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = nrow + e;
        b[row * BK + ((((kr >> 2) ^ swz<SLOTS>(row)) << 2) | (kr & 3))] = rb[slot][l][e];
      }
    }
  };

  const int fr = lane & 15, fq = lane >> 4;
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

  if (nk > 0) {
    load_chunk(0, 0);
    store_chunk(0, 0);
    if (nk > 1) load_chunk(1, 1);
    __syncthreads();
    // Steady state (as in gemm_kernel): pairs of chunks in which every refill exists, so the loads are unconditional and
    // the compiler can count them -- with the guarded form alone every ds_write of the older ring slot waited for ALL
    // outstanding loads (vmcnt(0)), i.e. one memory round trip per chunk: PMC had the matrix pipes 34-50 % busy with
    // LDS and its bank conflicts NOT what the waves waited for (profiles/r02_wgrad_staging_experiment.txt).
    int kc = 0;
    for (; kc + 4 <= nk; kc += 2) {
      load_chunk(kc + 2, 0);
      mfma_chunk<BK, WBM, WBN>(As, Bs, a_off, a_swz, b_off, b_swz, fq, acc);
      store_chunk(1, 1);
      __syncthreads();
      load_chunk(kc + 3, 1);
      mfma_chunk<BK, WBM, WBN>(As + BM * BK, Bs + BN * BK, a_off, a_swz, b_off, b_swz, fq, acc);
      store_chunk(0, 0);
      __syncthreads();
    }
    for (; kc < nk; kc += 2) {
      // even chunk: LDS buffer 0, ring slot 0 free -> prefetch chunk kc+2
      if (kc + 2 < nk) load_chunk(kc + 2, 0);
      mfma_chunk<BK, WBM, WBN>(As, Bs, a_off, a_swz, b_off, b_swz, fq, acc);
      if (kc + 1 < nk) store_chunk(1, 1);
      __syncthreads();
      if (kc + 1 < nk) {
        if (kc + 3 < nk) load_chunk(kc + 3, 1);
        mfma_chunk<BK, WBM, WBN>(As + BM * BK, Bs + BN * BK, a_off, a_swz, b_off, b_swz, fq, acc);
        if (kc + 2 < nk) store_chunk(0, 0);
        __syncthreads();
      }
    }
  }
  if (do_bias) {
    // thread (kr, cv) holds the sum over its chunks' rows kr of columns 4cv..4cv+3: the 32 kr partials of a column are
    // summed through LDS (free after the K loop's last barrier) in a fixed order
    float* red = lds;                                  // [BK][BM]
#pragma unroll
    for (int l = 0; l < AL; ++l) *reinterpret_cast<f32x4*>(red + 4 * (tid + 256 * l)) = bsum[l];
    __syncthreads();
    if (tid < BM && m0 + tid < p.N) {                  // N % 4 == 0: an in-range column was never a clamped duplicate
      float t = 0.0f;
#pragma unroll
      for (int kr = 0; kr < BK; ++kr) t += red[kr * BM + tid];
      out[(size_t)p.N * p.K + m0 + tid] = t;
    }
  }
  GemmParams q{};   // float4 rows (K % 4 == 0 is a precondition of this kernel)
  q.C = out; q.M = p.N; q.N = p.K; q.ldc = p.K; q.act = ACT_NONE;
  q.W = p.x;   // the epilogue reads (and discards) K floats from W when there is no bias: a readable buffer, never the null page
  gemm_epilogue<WBM, WBN>(q, acc, m0, n0, wm * (BM / 2), wn * (BN / 2), fr, fq);
  if (p.merged == nullptr) return;                     // block-uniform
  // ---- in-launch slice merge: the split-K hand-off of cdna_hip_programming.md (section 5, "Projection GEMM at M = 256",
  //      item 2), in its plain-store form.  Producer side, every workgroup: all its stores have left (each wave's
  //      vmcnt(0), then the barrier), ONE lane publishes with an agent-scope release and draws the tile's ticket.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                     // also: the bias reduction above is done with `lds`
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // keep: the compiler may drop the fence's own wait
    const unsigned t = __hip_atomic_fetch_add(p.cnt + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = t == gridDim.y - 1;
    if (last) {                                        // consumer side: ONE agent-scope acquire, then the barrier below
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(p.cnt + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
    reinterpret_cast<volatile int*>(lds)[0] = last;
  }
  __syncthreads();
  if (reinterpret_cast<volatile int*>(lds)[0] == 0) return;
  const int S = gridDim.y;
  const size_t stride = (size_t)p.N * p.K + (p.bias ? p.N : 0);
  constexpr int C4 = BN / 4;
  for (int idx = tid; idx < BM * C4; idx += 256) {
    const int m = m0 + idx / C4, n = n0 + 4 * (idx % C4);
    if (m >= p.N || n >= p.K) continue;
    const size_t off = (size_t)m * p.K + n;
    f32x4 a = *reinterpret_cast<const f32x4*>(p.dw + off);
    for (int z = 1; z < S; ++z) a += *reinterpret_cast<const f32x4*>(p.dw + (size_t)z * stride + off);
    *reinterpret_cast<f32x4*>(p.merged + off) = a;
  }
  if (do_bias) {
    for (int i = tid; i < BM; i += 256) {
      const int m = m0 + i;
      if (m >= p.N) continue;
      const size_t off = (size_t)p.N * p.K + m;
      float a = p.dw[off];
      for (int z = 1; z < S; ++z) a += p.dw[(size_t)z * stride + off];
      p.merged[off] = a;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// LayerNorm-fused GEMM:  C = epilogue( LN(A) W^T )  for K = normalised width <= NKMAX*BK.
// Every LayerNorm of the model feeds exactly one Linear (norm1 -> in_proj / q-proj, norm2 -> linear1 / ff.0,
// fusion.norm -> decoder.0; model.py:149,168-172 and nn.TransformerEncoderLayer norm_first), so the
// normalisation is applied to the A operand on its way to LDS and the stand-alone LayerNorm launches (13 per
// forward, ~4.5 us each at their launch floor) disappear.  K is short (d_model), so the whole A row-slice and W
// slice of the tile are loaded into registers up front (all loads in flight at once = one memory round trip),
// the row statistics are computed from those registers (no second pass over A), and the chunks are then fed
// through the same swizzled 2-buffer LDS image / MFMA loop as gemm_kernel.
// Row statistics: shifted one-pass sums (relative to the row's first element), combined across the SLOTS
// lanes that own the row with in-wave shuffles; biased variance, eps inside the sqrt like nn.LayerNorm.
// NK = K / BK exactly (compile time): a runtime "if (kc < nk)" around the register arrays makes hipcc copy them
// through v_mov / v_accvgpr webs (measured 4.5 TFLOP/s on the 8-chunk variant).
// WM x WN wavefronts (default 2 x 2 = 256 threads).  The large-workgroup instances (8 or 16 wavefronts, ONE workgroup per CU)
// exist because of what the resident-workgroup census of the 256-thread form shows on the model's shapes
// (profiles/r03_ln_gemm_phases.txt): its three workgroups per CU each pull their own A and W slabs through the CU's one
// texture-address unit (288 KB per CU and launch at 32x64) and then leave the CU one after the other, a staircase of three
// ~3.4 us steps behind a ~5 us prologue.  One workgroup of 8-16 waves covering the same outputs fetches every A row and W row
// of the CU once (160-192 KB) and keeps 2-4 waves per SIMD on the matrix pipes in one K loop.
template <int BM, int BN, int BK, int NKMAX, int WM = 2, int WN = 2>
__global__ __launch_bounds__(64 * WM * WN) void gemm_ln_kernel(const GemmParams pin) {
  GemmParams p = pin;
  dbg_stamp(p, 0);
  constexpr int NT = 64 * WM * WN;
  constexpr int SLOTS = BK / 4;
  constexpr int RPP = NT / SLOTS;
  constexpr int APASS = BM / RPP;
  constexpr int BPASS = BN / RPP;
  constexpr int WBM = BM / (16 * WM);
  constexpr int WBN = BN / (16 * WN);
  static_assert(BM % RPP == 0 && BN % RPP == 0 && BM % (16 * WM) == 0 && BN % (16 * WN) == 0, "tile / workgroup shape mismatch");
  __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * BK];
  float* As = lds;
  float* Bs = lds + 2 * BM * BK;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int nbn = (p.N + BN - 1) / BN;
  int tile = xcd_tile(p);
  select_pair(p, tile);
  const int bm = tile / nbn;
  const int bn = tile - bm * nbn;
  const int m0 = bm * BM, n0 = bn * BN;
  constexpr int nk = NKMAX;

  const int srow = tid / SLOTS;
  const int sslot = tid % SLOTS;
  f32x4 ra[NKMAX][APASS], rb[NKMAX][BPASS], rg[NKMAX], rbe[NKMAX];
  int a_st[APASS], b_st[BPASS];
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    const int r = srow + RPP * i;
    a_st[i] = r * BK + ((sslot ^ swz<SLOTS>(r)) << 2);
    int m = m0 + r;
    m = m < p.M ? m : p.M - 1;
    const float* src = p.A + (size_t)m * p.lda + 4 * sslot;
#pragma unroll
    for (int kc = 0; kc < NKMAX; ++kc)
      if (kc < nk) ra[kc][i] = *reinterpret_cast<const f32x4*>(src + kc * BK);
  }
#pragma unroll
  for (int kc = 0; kc < NKMAX; ++kc)
    if (kc < nk) {
      rg[kc] = *reinterpret_cast<const f32x4*>(p.ln_gamma + kc * BK + 4 * sslot);
      rbe[kc] = *reinterpret_cast<const f32x4*>(p.ln_beta + kc * BK + 4 * sslot);
    }
#pragma unroll
  for (int i = 0; i < BPASS; ++i) {
    const int r = srow + RPP * i;
    b_st[i] = r * BK + ((sslot ^ swz<SLOTS>(r)) << 2);
    int n = n0 + r;
    n = n < p.N ? n : p.N - 1;
    const float* src = p.W + (size_t)n * p.ldw + 4 * sslot;
#pragma unroll
    for (int kc = 0; kc < NKMAX; ++kc)
      if (kc < nk) rb[kc][i] = *reinterpret_cast<const f32x4*>(src + kc * BK);
  }

  // ---- row statistics from the registers, then normalise in place -----------------------------------
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    const float c = __shfl(ra[0][i][0], lane & ~(SLOTS - 1));   // the row's first element (its sslot-0 lane)
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int kc = 0; kc < NKMAX; ++kc)
      if (kc < nk) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dlt = ra[kc][i][e] - c;
          s1 += dlt;
          s2 = fmaf(dlt, dlt, s2);
        }
      }
#pragma unroll
    for (int off = SLOTS / 2; off >= 1; off >>= 1) {
      s1 += __shfl_xor(s1, off);
      s2 += __shfl_xor(s2, off);
    }
    const float inv = 1.0f / (float)p.K;
    const float m1 = s1 * inv;
    const float var = fmaxf(s2 * inv - m1 * m1, 0.0f);
    const float mu = c + m1;
    const float rs = 1.0f / sqrtf(var + p.ln_eps);
#pragma unroll
    for (int kc = 0; kc < NKMAX; ++kc)
      if (kc < nk) {
#pragma unroll
        for (int e = 0; e < 4; ++e) ra[kc][i][e] = (ra[kc][i][e] - mu) * rs * rg[kc][e] + rbe[kc][e];
      }
  }

  auto store_chunk = [&](int kc, int buf) {
    float* a = As + buf * BM * BK;
    float* b = Bs + buf * BN * BK;
#pragma unroll
    for (int i = 0; i < APASS; ++i) *reinterpret_cast<f32x4*>(a + a_st[i]) = ra[kc][i];
#pragma unroll
    for (int i = 0; i < BPASS; ++i) *reinterpret_cast<f32x4*>(b + b_st[i]) = rb[kc][i];
  };

  const int fr = lane & 15;
  const int fq = lane >> 4;
  int a_off[WBM], a_swz[WBM], b_off[WBN], b_swz[WBN];
#pragma unroll
  for (int i = 0; i < WBM; ++i) {
    const int r = wm * (BM / WM) + 16 * i + fr;
    a_off[i] = r * BK;
    a_swz[i] = swz<SLOTS>(r);
  }
#pragma unroll
  for (int j = 0; j < WBN; ++j) {
    const int r = wn * (BN / WN) + 16 * j + fr;
    b_off[j] = r * BK;
    b_swz[j] = swz<SLOTS>(r);
  }
  f32x4 acc[WBM][WBN];
#pragma unroll
  for (int i = 0; i < WBM; ++i)
#pragma unroll
    for (int j = 0; j < WBN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  store_chunk(0, 0);
  __syncthreads();
  dbg_stamp(p, 1);
#pragma unroll
  for (int kc = 0; kc < NKMAX; ++kc) {
    if (kc < nk) {   // block-uniform
      mfma_chunk<BK, WBM, WBN>(As + (kc & 1) * BM * BK, Bs + (kc & 1) * BN * BK, a_off, a_swz, b_off, b_swz, fq, acc);
      if (kc + 1 < NKMAX) {
        if (kc + 1 < nk) store_chunk(kc + 1, (kc + 1) & 1);
      }
      __syncthreads();
    }
  }

  dbg_stamp(p, 2);
  gemm_epilogue<WBM, WBN>(p, acc, m0, n0, wm * (BM / WM), wn * (BN / WN), fr, fq);
  if (p.dbg) {                                                   // block-uniform, diagnostics only
    dbg_stamp(p, 3);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    dbg_stamp(p, 4);
  }
}

#ifdef AVSEP_DEV   // developer instances: measured slower, kept bit-identical for tests and sweeps (make dev)
// ---------------------------------------------------------------------------------------------------------
// Large-tile instance on v_mfma_f32_32x32x2_f32 (BM x BN = 128 x 128 or 256 x 128, 4 wavefronts as 2 x 2, each wave a
// (BM/2) x (BN/2) tile made of 32 x 32 MFMA blocks).  For the M ~ 16 k shapes of configs 3-5 and the training step.
//   * Half the matrix instructions of the 16x16x4 form for the same tile (one 64-cycle MFMA per 32x32x2 product): the
//     issue slots between MFMAs are what the fragment reads, the staging loads and the LDS writes of the next chunk
//     live in, and a 64x64 wave tile needs only 8 ds_read_b128 per 32 MFMAs (2048 matrix cycles).
//   * A chunk of BK = 32 is 4096 matrix cycles per wave, longer than an L2 / Infinity-Cache round trip, so the register
//     ring is ONE chunk deep: loads of chunk c+1 are issued before the MFMAs of chunk c and written to the other LDS
//     buffer after them (issue-early / write-late); 64 KB of LDS -> two workgroups per CU, whose barriers, prologues
//     and epilogues overlap each other's MFMAs.
//   * Same LDS image, same swizzle, same k order as gemm_kernel: lane (r = l&31, h = l>>5) reads the two float4 slots
//     {4s+h, 4s+2+h} of k-step s and feeds component c of the first then of the second to consecutive MFMAs, i.e. the
//     products enter every accumulator in the order k = 16s+c, +4, +8, +12 (c = 0..3) -- exactly the order of the
//     16x16x4 kernels (their k index is the lane quarter), and both instructions are plain k-ordered fma chains, so the
//     results are BIT-IDENTICAL across all tile shapes (tests/test_gpu_parity.py::test_op_linear_tiles_bit_identical).
//   * Operands swapped like gemm_epilogue's: D[n][m], lane = output row m, registers 4g..4g+3 = columns
//     n = 8g + 4h + (0..3): float4 stores, bias / residual read as float4.
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int TM, int TN>
__device__ __forceinline__ void gemm32_epilogue(const GemmParams& p, const f32x16 (&acc)[TM][TN], int m0, int n0, int mw,
                                                int nw, int lr, int lh) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const bool v4 = !(p.N & 3) && !(p.ldc & 3) && (!p.R || !(p.ldr & 3));
  const bool v2 = !(p.N & 1) && !(p.ldc & 1);
  constexpr int NP = 4 * TN;                    // float4 pieces per output row and lane
  f32x4 bv[NP];
  int ncol[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const int n = n0 + nw + 32 * (q >> 2) + 8 * (q & 3) + 4 * lh;
    ncol[q] = n;
    bv[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias && n < p.N) {
      if (v4) bv[q] = *reinterpret_cast<const f32x4*>(p.bias + n);
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[q][e] = (n + e < p.N) ? p.bias[n + e] : 0.0f;
      }
    }
  }
  // one 32-row block at a time: all its loads, then all its stores (see gemm_epilogue: loads behind stores wait for
  // the stores on CDNA's shared in-order memory counter)
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + mw + 32 * i + lr;
    if (m >= p.M) continue;
    const int rr = p.rperiod > 0 ? (m % p.rperiod) : m;
    f32x4 rv[NP], xv[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int n = ncol[q];
      rv[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      xv[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (n >= p.N) continue;
      if (p.R) {
        if (v4) rv[q] = *reinterpret_cast<const f32x4*>(p.R + (size_t)rr * p.ldr + n);
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e) rv[q][e] = (n + e < p.N) ? p.R[(size_t)rr * p.ldr + n + e] : 0.0f;
        }
      }
      if (p.C2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[q][e] = (n + e < p.N) ? p.X[(size_t)m * p.ldx + (n + e) % p.F] : 0.0f;
      }
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int n = ncol[q];
      if (n >= p.N) continue;
      const int j = q >> 2, g = q & 3;
      f32x4 v, w;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = acc[i][j][4 * g + e];
        x = p.bias ? x + bv[q][e] : x;
        x = apply_act(x, p.act);
        if (p.R) x += rv[q][e];
        v[e] = x;
        w[e] = x * xv[q][e];
      }
      float* c = p.C + (size_t)m * p.ldc + n;
      float* c2 = p.C2 ? p.C2 + (size_t)m * p.ldc + n : nullptr;
      if (v4) {
        *reinterpret_cast<f32x4*>(c) = v;
        if (c2) *reinterpret_cast<f32x4*>(c2) = w;
      } else if (v2) {
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

// One BK = 32 chunk (two k-steps of 16) of 32x32x2 MFMAs from the swizzled LDS image.  PF: the fragments of the next
// k-step are fetched before the MFMAs of the current one (second register set).
template <int TM, int TN, bool PF>
__device__ __forceinline__ void mfma32_chunk(const float* a, const float* b, const int (&a_off)[TM], const int (&a_swz)[TM],
                                             const int (&b_off)[TN], const int (&b_swz)[TN], int lh,
                                             f32x16 (&acc)[TM][TN]) {
  constexpr int S = 2;   // BK / 16
  f32x4 fa[PF ? 2 : 1][TM][2], fb[PF ? 2 : 1][TN][2];
  auto fetch = [&](int s, int set) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int q = 0; q < 2; ++q)
        fa[set][i][q] = *reinterpret_cast<const f32x4*>(a + a_off[i] + (((4 * s + 2 * q + lh) ^ a_swz[i]) << 2));
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 2; ++q)
        fb[set][j][q] = *reinterpret_cast<const f32x4*>(b + b_off[j] + (((4 * s + 2 * q + lh) ^ b_swz[j]) << 2));
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
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[cur][j][q][c], fa[cur][i][q][c], acc[i][j], 0, 0, 0);
  }
}

template <int BM, int BN, int AMODE, bool PF>
__global__ __launch_bounds__(256, BM * BN <= 128 * 128 ? 2 : 1) void gemm32_kernel(const GemmParams pin) {
  GemmParams p = pin;
  constexpr int BK = 32, SLOTS = 8, RPP = 32;
  constexpr int APASS = BM / RPP, BPASS = BN / RPP;
  constexpr int TM = BM / 64, TN = BN / 64;            // 32x32 blocks per wave (2 x 2 waves)
  __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * BK];
  float* As = lds;
  float* Bs = lds + 2 * BM * BK;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nbn = (p.N + BN - 1) / BN;
  const int tile = xcd_tile(p);
  const int bm = tile / nbn;
  const int bn = tile - bm * nbn;
  const int m0 = bm * BM, n0 = bn * BN;

  const int srow = tid >> 3;
  const int sslot = tid & 7;
  const float* a_src[APASS];
  int a_t[APASS];                    // TAPS3: time index of the row
  int a_st[APASS], b_st[BPASS];
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    const int r = srow + RPP * i;
    a_st[i] = r * BK + ((sslot ^ swz<SLOTS>(r)) << 2);
    int m = m0 + r;
    m = m < p.M ? m : p.M - 1;
    a_src[i] = p.A + (size_t)m * p.lda + 4 * sslot;
    a_t[i] = AMODE == AMODE_TAPS3 ? m % p.T : 0;
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
  const int cpt = (AMODE == AMODE_PLAIN) ? nk : (p.Kt / BK);   // chunks per tap
  int tap = 0, sub = 0, kload = 0;

  // register ring, two chunks deep: the LDS write of chunk c+1 (wherever the compiler schedules it inside the
  // iteration) waits only for loads issued a whole iteration earlier, never for the ones just issued
  f32x4 ra[2][APASS], rb[2][BPASS];
  bool rok[2][APASS];
  auto load_chunk = [&](int slot) {
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      if (AMODE == AMODE_PLAIN) {
        ra[slot][i] = *reinterpret_cast<const f32x4*>(a_src[i] + kload * BK);
      } else {   // TAPS3: unconditional load from an always-valid address, zeroed at the LDS write when out of range
        const int t = a_t[i] + tap - 1;
        const bool ok = (t >= 0) && (t < p.T);
        ra[slot][i] = *reinterpret_cast<const f32x4*>(a_src[i] + (ptrdiff_t)(ok ? tap - 1 : 0) * p.lda + sub * BK);
        rok[slot][i] = ok;
      }
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) rb[slot][i] = *reinterpret_cast<const f32x4*>(b_src[i] + kload * BK);
    ++kload;
    if (AMODE != AMODE_PLAIN) {
      if (++sub == cpt) { sub = 0; ++tap; }
    }
  };
  auto store_chunk = [&](int slot, int buf) {
    float* a = As + buf * BM * BK;
    float* b = Bs + buf * BN * BK;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      f32x4 v = ra[slot][i];
      if (AMODE != AMODE_PLAIN) v = rok[slot][i] ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(a + a_st[i]) = v;
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) *reinterpret_cast<f32x4*>(b + b_st[i]) = rb[slot][i];
  };

  const int lr = lane & 31, lh = lane >> 5;
  int a_off[TM], a_swz[TM], b_off[TN], b_swz[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int r = wm * (BM / 2) + 32 * i + lr;
    a_off[i] = r * BK;
    a_swz[i] = swz<SLOTS>(r);
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int r = wn * (BN / 2) + 32 * j + lr;
    b_off[j] = r * BK;
    b_swz[j] = swz<SLOTS>(r);
  }
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // chunk c lives in ring slot c & 1 and LDS buffer c & 1
  load_chunk(0);
  if (nk > 1) load_chunk(1);
  store_chunk(0, 0);
  __syncthreads();
  auto step = [&](int j, bool refill, bool has_next) {       // j = parity of the current chunk (compile-time after unroll)
    if (refill) load_chunk(j);                               // chunk kc+2 into the slot chunk kc left one iteration ago
    mfma32_chunk<TM, TN, PF>(As + j * BM * BK, Bs + j * BN * BK, a_off, a_swz, b_off, b_swz, lh, acc);
    if (has_next) store_chunk(j ^ 1, j ^ 1);                 // chunk kc+1: loaded one iteration ago
    __syncthreads();
  };
  int kc = 0;
  for (; kc + 4 <= nk; kc += 2) {     // steady state: every refill exists -> unconditional loads, counted vmcnt waits
    step(0, true, true);
    step(1, true, true);
  }
  for (; kc < nk; kc += 2) {          // tail (block-uniform conditions)
    step(0, kc + 2 < nk, kc + 1 < nk);
    if (kc + 1 < nk) step(1, kc + 3 < nk, kc + 2 < nk);
  }

  gemm32_epilogue<TM, TN>(p, acc, m0, n0, wm * (BM / 2), wn * (BN / 2), lr, lh);
}

template <int BM, int BN, int AMODE, bool PF>
hipError_t launch32_t(const GemmParams& p, hipStream_t s) {
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  hipLaunchKernelGGL((gemm32_kernel<BM, BN, AMODE, PF>), dim3(nbm * nbn), dim3(256), 0, s, p);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Persistent form of gemm_kernel for the large plain GEMMs (M ~ 16 k rows: configs 3-5, the training step).
// In-kernel stamps of the one-tile-per-workgroup kernel on 16064x2048x512 (profiles/r02_gemm_one_cu_timeline.txt): a
// 128x64 tile lives ~38 us = 2.5 prologue (cold first loads, no MFMA) + 30 K loop + 4.4 epilogue + 0.5 store drain, and
// the next workgroup starts ~1.1 us after a slot frees -- 8.5 of every 39 us per residency slot feed the matrix pipes
// nothing, which is the distance between 120 and 136 TFLOP/s (K = 512 vs K = 2048 on the same tile).
// Here a workgroup stays resident and walks its tiles (blockIdx.x, + gridDim.x, ...): the K loop is ONE steady state
// over all chunks of a tile -- the refills of the last D iterations are the first D chunks of the NEXT tile (pointer
// selects, unconditional loads), and the last iteration writes the next tile's chunk 0 to LDS -- so a tile has no
// prologue and no dispatch gap.  The epilogue then runs with the next tile's first chunks already in flight: its loads
// are younger than those on the in-order memory counter, its stores younger still, so neither delays them.
// Same staging, LDS image, fragment order, MFMA order and epilogue as gemm_kernel: results are bit-identical.
// Requires nk % D == 0 (ring slot and LDS buffer of chunk 0 are then the same for every tile); the host falls back to
// gemm_kernel otherwise, and for problems with fewer than two tiles per resident workgroup.
template <int BM, int BN, int BK, bool PF>
__global__ __launch_bounds__(256, 3) void gemm_persist_kernel(const GemmParams p, int ntiles, int stagger) {
  constexpr int SLOTS = BK / 4, RPP = 256 / SLOTS, APASS = BM / RPP, BPASS = BN / RPP;
  constexpr int WBM = BM / 32, WBN = BN / 32;
  constexpr int D = (APASS + BPASS <= 4) ? 4 : 2;
  __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * BK];
  float* As = lds;
  float* Bs = lds + 2 * BM * BK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nbn = (p.N + BN - 1) / BN;
  const int srow = tid / SLOTS, sslot = tid % SLOTS;
  const int nk = p.K / BK;

  // XCD-aware order over ALL tiles (see xcd_tile): virtual tile v of this workgroup -> tile index
  auto tile_of = [&](int v) {
    if (p.no_xcd_remap) return v;
    const int q = ntiles >> 3, r = ntiles & 7, xcd = v & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
  };
  int a_st[APASS], b_st[BPASS];
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    const int r = srow + RPP * i;
    a_st[i] = r * BK + ((sslot ^ swz<SLOTS>(r)) << 2);
  }
#pragma unroll
  for (int i = 0; i < BPASS; ++i) {
    const int r = srow + RPP * i;
    b_st[i] = r * BK + ((sslot ^ swz<SLOTS>(r)) << 2);
  }
  // Source addressing keeps the per-tile part in scalars (a base pointer and the last valid row of the tile, both
  // block-uniform) and recomputes the per-thread offset at the load: two VALU ops per 16-byte load instead of 24
  // address registers for (this tile, next tile), which is what keeps the 128x64 instance at three workgroups per CU.
  struct Src {
    const float* a;
    const float* b;
    int alim, blim;                                     // last row of the tile that exists (rows past it re-read it)
  };
  auto src_of = [&](int tile) {
    const int bm = tile / nbn, bn = tile - bm * nbn;
    Src r;
    r.a = p.A + (size_t)bm * BM * p.lda;
    r.b = p.W + (size_t)bn * BN * p.ldw;
    r.alim = min(BM, p.M - bm * BM) - 1;
    r.blim = min(BN, p.N - bn * BN) - 1;
    return r;
  };
  f32x4 ra[D][APASS], rb[D][BPASS];
  auto store_chunk = [&](int slot, int buf) {
    float* a = As + buf * BM * BK;
    float* b = Bs + buf * BN * BK;
#pragma unroll
    for (int i = 0; i < APASS; ++i) *reinterpret_cast<f32x4*>(a + a_st[i]) = ra[slot][i];
#pragma unroll
    for (int i = 0; i < BPASS; ++i) *reinterpret_cast<f32x4*>(b + b_st[i]) = rb[slot][i];
  };

  const int fr = lane & 15, fq = lane >> 4;
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

  auto load_chunk = [&](int slot, const Src& t, int ko) {
    const float* a = t.a + ko;
    const float* b = t.b + ko;
#pragma unroll
    for (int i = 0; i < APASS; ++i)
      ra[slot][i] = *reinterpret_cast<const f32x4*>(a + (unsigned)(min(srow + RPP * i, t.alim) * p.lda + 4 * sslot));
#pragma unroll
    for (int i = 0; i < BPASS; ++i)
      rb[slot][i] = *reinterpret_cast<const f32x4*>(b + (unsigned)(min(srow + RPP * i, t.blim) * p.ldw + 4 * sslot));
  };
  if (stagger > 0) {   // developer experiment: de-phase the workgroups that share a CU (100 MHz ticks per residency class)
    const unsigned long long t0 = wall_clock64(), wait = (unsigned long long)stagger * (blockIdx.x / 256u);
    while (wall_clock64() - t0 < wait) __builtin_amdgcn_s_sleep(8);
  }
  int v = blockIdx.x;                                   // virtual tile; the grid never exceeds ntiles
  int tile = tile_of(v);
  Src cur = src_of(tile);
  // the only prologue of the launch: chunks 0 .. D-1 of the first tile
#pragma unroll
  for (int j = 0; j < D; ++j) load_chunk(j, cur, j * BK);
  store_chunk(0, 0);
  __syncthreads();

  while (true) {
    const int vn = v + (int)gridDim.x;
    const bool has_next = vn < ntiles;                  // block-uniform
    // the last tile of a workgroup refills from its own first chunks: valid addresses, data never used
    const int tile_n = has_next ? tile_of(vn) : tile;
    const Src nxt = src_of(tile_n);
    for (int kc0 = 0; kc0 < nk; kc0 += D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        // slot j (chunk kc0+j, in LDS since the last step) is refilled with chunk kc0+j+D of this tile, or -- in the
        // last D steps -- with chunk j of the next tile: the address is selected, the load is unconditional
        const int kl = kc0 + j + D;
        const bool own = kl < nk;                       // block-uniform
        load_chunk(j, own ? cur : nxt, (own ? kl : kl - nk) * BK);
        mfma_chunk<BK, WBM, WBN, PF>(As + (j & 1) * BM * BK, Bs + (j & 1) * BN * BK, a_off, a_swz, b_off, b_swz, fq, acc);
        store_chunk((j + 1) % D, (j + 1) & 1);          // chunk kc0+j+1 -- at the very end: chunk 0 of the next tile
        __syncthreads();
      }
    }
    {
      const int bm = tile / nbn, bn = tile - bm * nbn;
      gemm_epilogue<WBM, WBN>(p, acc, bm * BM, bn * BN, wm * (BM / 2), wn * (BN / 2), fr, fq);
    }
    if (!has_next) break;
#pragma unroll
    for (int i = 0; i < WBM; ++i)
#pragma unroll
      for (int j = 0; j < WBN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    v = vn;
    tile = tile_n;
    cur = nxt;
  }
}

// Launch the persistent form when it applies: plain A, no split-K, no diagnostics, nk % D == 0, and at least two tiles
// per resident workgroup (the grid is the resident capacity the runtime reports for the instance).
template <int BM, int BN, int BK, bool PF>
long persist_grid(const GemmParams& p) {   // workgroups of the persistent launch, 0 = use gemm_kernel
  constexpr int SLOTS = BK / 4, RPP = 256 / SLOTS, APASS = BM / RPP, BPASS = BN / RPP;
  constexpr int D = (APASS + BPASS <= 4) ? 4 : 2;
  if (p.amode != AMODE_PLAIN || p.ksplit > 1 || p.dbg || p.mag_F > 0 || p.ln_gamma) return 0;
  const int nk = p.K / BK;
  if (nk < D || nk % D) return 0;
  const long ntiles = (long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  if (ntiles < 256 || ntiles > 0x7fffffffL) return 0;
  static int per_cu = 0, cus = 0;
  if (!per_cu) {
    int nb = 0, dev = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(gemm_persist_kernel<BM, BN, BK, PF>), 256, 0) != hipSuccess || nb < 1) nb = 1;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    per_cu = nb;
  }
  // Measured NEGATIVE (profiles/r02_ab_persistent_gemm.txt: -2..3 % on every cfg3-5 shape and on cfg3 end to end, with
  // or without de-phasing the workgroups of a CU): with three workgroups resident per CU the other two already cover a
  // tile's prologue and the dispatch gap, so the form stays a developer instance behind AVSEP_PERSIST.  The switches
  // are read per launch so one process can compare the two forms (tests/test_gpu_parity.py).
  if (!getenv("AVSEP_PERSIST")) return 0;
  const char* e = getenv("AVSEP_PERSIST_ROUNDS");
  const double min_rounds = e ? atof(e) : 2.0;
  e = getenv("AVSEP_PERSIST_WGS");
  const long grid = (long)cus * (e && atoi(e) > 0 && atoi(e) < per_cu ? atoi(e) : per_cu);
  if ((double)ntiles < min_rounds * (double)grid) return 0;
  return grid;
}

template <int BM, int BN, int BK, bool PF>
bool try_launch_persist(const GemmParams& p, hipStream_t s, hipError_t* err) {
  const long grid = persist_grid<BM, BN, BK, PF>(p);
  if (!grid) return false;
  const long ntiles = (long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  const char* st = getenv("AVSEP_PERSIST_STAGGER");
  hipLaunchKernelGGL((gemm_persist_kernel<BM, BN, BK, PF>), dim3((unsigned)grid), dim3(256), 0, s, p, (int)ntiles, st ? atoi(st) : 0);
  *err = hipGetLastError();
  return true;
}

#endif  // AVSEP_DEV

// rows -> 16-row-block tiles of BOTH problems of a pair launch (alt.M > 0), else of the one problem
inline long row_tiles(const GemmParams& p, int bm) {
  return (long)(p.M + bm - 1) / bm + (p.alt.M > 0 ? (long)(p.alt.M + bm - 1) / bm : 0);
}

// GemmParams::nbn_magic: exact for every tile < ntiles when ntiles * nbn < 2^32 (the quotient's error term is tile * e / (nbn 2^32)
// with e <= nbn)
unsigned tile_row_magic(long ntiles, int nbn) {
  if (nbn < 2 || ntiles * (long)nbn >= (1L << 32)) return 0u;
  return (unsigned)((1ULL << 32) / (unsigned)nbn) + 1u;
}

template <int BM, int BN, int BK, int AMODE, bool PF = false, int RING = 0>
hipError_t launch_t(GemmParams p, hipStream_t s) {
  const int nbn = (p.N + BN - 1) / BN;
  p.g_tiles0 = p.alt.M > 0 ? ((p.M + BM - 1) / BM) * nbn : 0;
  p.nbn_magic = tile_row_magic(row_tiles(p, BM) * nbn, nbn);
  p.ln_inv_k = 1.0f / (float)p.K;
#ifdef AVSEP_DEV
  if (p.dbg) {   // diagnostics: what the runtime says about residency of this instance
    int nb = 0;
    hipFuncAttributes fa{};
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(gemm_kernel<BM, BN, BK, AMODE, PF, RING>), 256, 0);
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(gemm_kernel<BM, BN, BK, AMODE, PF, RING>));
    fprintf(stderr, "[gemm dbg] instance <%d,%d,%d,%d,%d>: occupancy API %d workgroups / CU, numRegs %d, static LDS %zu B, scratch %zu B\n",
            BM, BN, BK, AMODE, (int)PF, nb, fa.numRegs, fa.sharedSizeBytes, fa.localSizeBytes);
  }
#endif
  hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, AMODE, PF, RING>), dim3((unsigned)(row_tiles(p, BM) * nbn), p.ksplit > 1 ? p.ksplit : 1),
                     dim3(256), 0, s, p);
  return hipGetLastError();
}

struct Tile { int bm, bn, bk; };   // bm >= 128 && bn >= 128: the 32x32x2 large-tile developer kernel (gemm32_kernel)
inline bool is_g32(const Tile& t) { return t.bm >= 128 && t.bn >= 128; }

// Block tile and K-chunk for one launch.
//  * (BM,BN): measured on MI355X (profiles/r01a_gemm_tile_sweep.txt, r02_gemm_tile_sweep.txt): what decides the time at
//    M ~ 2k is how many workgroups are resident, so small problems take small tiles; 128x64 from 1000 tiles (the mask
//    head -- N % 4 != 0, two outputs, block-by-block epilogue -- from 2048), 64x64 (register ring of depth 2, four
//    workgroups per CU) from 448, 64x32 from 512, else 32x32.
//  * BK: 64 where it divides K (and the per-tap K of the conv modes) and the tile is small, else 32.
// The developer build can override every threshold (AVSEP_GEMM_TILE=BMxBNxBK, AVSEP_CONV_TILE, AVSEP_T64, AVSEP_T6432,
// AVSEP_T12864, AVSEP_T128, AVSEP_T256) for hardware sweeps and for the tile bit-identity tests.
Tile pick_tile(const GemmParams& p) {
  if (const char* e = dev_env("AVSEP_GEMM_TILE")) {
    Tile t{0, 0, 32};
    if (sscanf(e, "%dx%dx%d", &t.bm, &t.bn, &t.bk) >= 2) return t;
  }
  if (p.lnx_c1) {
    // LayerNorm in the epilogue: BK = 32 on every tile, so that the side sums of the row statistics run over the same 8
    // lanes per row in the same order whatever the tile -- the results do not depend on the tile (nor, through it, on the
    // batch size).  Inside the cfg2 step the 32x64 tile (24 KB of LDS, 4 workgroups per SIMD) measured best at every
    // LayerNorm -> Linear site (profiles/r03_ab_ln_epilogue.txt); large problems take the tiles of the plain GEMM.
    if (const char* e = dev_env("AVSEP_LNX_TILE")) {
      Tile t{0, 0, 32};
      if (sscanf(e, "%dx%dx%d", &t.bm, &t.bn, &t.bk) >= 2) return t;
    }
    auto blocks = [&](int bm, int bn) { return row_tiles(p, bm) * ((p.N + bn - 1) / bn); };
    if (blocks(128, 64) >= 1000) return Tile{128, 64, 32};
    if (blocks(64, 64) >= 1024) return Tile{64, 64, 32};
    return p.N >= 64 ? Tile{32, 64, 32} : Tile{32, 32, 32};
  }
  if (p.amode == AMODE_TAPS3) {
    if (const char* e = dev_env("AVSEP_TAPS_TILE")) {
      Tile t{0, 0, 32};
      if (sscanf(e, "%dx%dx%d", &t.bm, &t.bn, &t.bk) >= 2) return t;
    }
  }
  if (p.amode == AMODE_CONV2D) {
    if (const char* e = dev_env("AVSEP_CONV_TILE")) {
      Tile t{0, 0, 32};
      if (sscanf(e, "%dx%dx%d", &t.bm, &t.bn, &t.bk) >= 2) return t;
    }
  }
  const long slices = p.ksplit > 1 ? p.ksplit : 1;
  auto blocks = [&](int bm, int bn) { return slices * row_tiles(p, bm) * ((p.N + bn - 1) / bn); };
  static const long t64 = dev_env("AVSEP_T64") ? atol(dev_env("AVSEP_T64")) : 448;
  static const long t6432 = dev_env("AVSEP_T6432") ? atol(dev_env("AVSEP_T6432")) : 512;
  static const long t12864 = dev_env("AVSEP_T12864") ? atol(dev_env("AVSEP_T12864")) : 1000;
  static const long t128 = dev_env("AVSEP_T128") ? atol(dev_env("AVSEP_T128")) : 1L << 40;
  static const long t256 = dev_env("AVSEP_T256") ? atol(dev_env("AVSEP_T256")) : 1L << 40;
  Tile pick{32, 32, 32};
  const bool g32_ok = (p.amode == AMODE_PLAIN || p.amode == AMODE_TAPS3) && slices == 1 && p.mag_F == 0 && p.alt.M <= 0;
  if (g32_ok && blocks(256, 128) >= t256) pick = Tile{256, 128, 32};
  else if (g32_ok && blocks(128, 128) >= t128) pick = Tile{128, 128, 32};
  else if (blocks(128, 64) >= t12864 && !(p.N & 3) && !p.C2) pick = Tile{128, 64, 32};
  else if (blocks(128, 64) >= 2048) pick = Tile{128, 64, 32};
  else if (blocks(64, 64) >= t64) pick = Tile{64, 64, 32};
  else if (blocks(64, 32) >= t6432) pick = Tile{64, 32, 32};
  // the two audio convolutions of a small batch run beside conv_stack, the most contended phase of the step: 32x64x32 (24 KB,
  // half the staging instructions per MFMA of 32x32) is 1 % faster one step at a time, equal with two in flight
  // (profiles/r03_ab_ln_epilogue.txt, item 7c)
  if (p.amode == AMODE_TAPS3 && pick.bm == 32 && pick.bn == 32 && p.N >= 64 && !(p.N & 3) && !p.C2) return Tile{32, 64, 32};
  if (p.amode == AMODE_PLAIN && pick.bm == 32 && pick.bn == 32 && slices == 1) {
    if (const char* e = dev_env("AVSEP_SMALL_TILE")) {       // developer A/B: the tile of the small plain GEMMs
      Tile t{0, 0, 32};
      if (sscanf(e, "%dx%dx%d", &t.bm, &t.bn, &t.bk) >= 2 && p.K % t.bk == 0) return t;
    }
  }
  const int kunit = (p.amode == AMODE_TAPS3 || p.amode == AMODE_CONV2D) ? p.Kt : p.K;   // a chunk must not straddle a tap
  const char* bk32 = dev_env("AVSEP_BK32");   // developer A/B: "plain" / "all" keep BK = 32 on the small tiles
  const bool keep32 = bk32 && (!strcmp(bk32, "all") || (!strcmp(bk32, "plain") && p.amode == AMODE_PLAIN));
  if (pick.bm + pick.bn <= 96 && kunit % 64 == 0 && !keep32) pick.bk = 64;
  return pick;
}

// LN-fused launch.  BK = 64 (or 32 when d % 64 != 0); the whole K = NK chunks lives in registers, NK in
// {1,2,4} (d_model <= 256).
int ln_bk(int K) { return (K % 64 == 0) ? 64 : 32; }
bool ln_fusable(const GemmParams& p) {
  if (!p.ln_gamma || !p.ln_beta || p.amode != AMODE_PLAIN) return false;
  // NK = 8 (d_model = 512) needs ~230 VGPRs and measured 11 TFLOP/s: not worth fusing, the stand-alone
  // LayerNorm + plain GEMM is used there
  const int nk = p.K / ln_bk(p.K);
  return nk == 1 || nk == 2 || nk == 4;
}

// LN-fused tiles: (bm, bn, bk) + wavefronts of the workgroup (4 = the 256-thread form)
struct LnTile { int bm, bn, bk, waves; };

LnTile pick_ln_tile(const GemmParams& p) {
  const int bk = ln_bk(p.K);
  if (const char* e = dev_env("AVSEP_LN_TILE")) {   // developer override: AVSEP_LN_TILE=BMxBN[xWAVES]
    LnTile t{32, 32, bk, 4};
    if (sscanf(e, "%dx%dx%d", &t.bm, &t.bn, &t.waves) >= 2) return t;
  }
  LnTile pick{32, 32, bk, 4};
  // Developer experiment (AVSEP_LN_BIG=1): one workgroup of 8 / 16 wavefronts per CU where its tile grid fills the chip in
  // ONE round (161..256 workgroups) -- the CU fetches each of its A and W rows once instead of once per 256-thread workgroup
  // and the prologue is one memory round trip instead of three staggered ones.  ALONE on the chip these instances are 8-12 %
  // faster (profiles/r03_ln_gemm_sweep.txt: QKV 2016x768 14.4 -> 13.4 us, FFN-1 2016x1024 18.1 -> 16.0, 1008x1024 10.4 -> 9.6)
  // and bit-identical; INSIDE the two-stream step they are 3 % SLOWER with two steps in flight and 4.5 % slower one step at a
  // time (profiles/r03_ab_ln_big_workgroups.txt): a workgroup that holds 80-96 KB of LDS and 8-16 wave slots of its CU keeps
  // the other branch's kernels off that CU, and the step lives on exactly that overlap.  Not in the product library.
  if (bk == 64 && (p.K == 256 || p.K == 128) && p.alt.M <= 0 && dev_env("AVSEP_LN_BIG")) {
    auto fits = [&](int bm, int bn) {
      const long t = row_tiles(p, bm) * ((p.N + bn - 1) / bn);
      return t > 160 && t <= 256;
    };
    if (p.N % 96 == 0 && fits(64, 96)) return LnTile{64, 96, bk, 8};
    if (p.N % 128 == 0 && fits(64, 128)) return LnTile{64, 128, bk, 16};
    if (p.N % 128 == 0 && fits(32, 128)) return LnTile{32, 128, bk, 8};
  }
  if (bk == 64 && p.K / bk <= 4) {
    static const LnTile cands[] = {{64, 64, 64, 4}, {64, 32, 64, 4}};
    auto blocks = [&](const LnTile& t) { return row_tiles(p, t.bm) * ((p.N + t.bn - 1) / t.bn); };
    // measured in the full cfg2 step: 32-row LN tiles beat 64x32 / 64x64 even where the bigger tiles win in isolation
    static const long tln = dev_env("AVSEP_TLN") ? atol(dev_env("AVSEP_TLN")) : 2048;
    bool found = false;
    for (const LnTile& t : cands)
      if (!found && blocks(t) >= tln) { pick = t; found = true; }
    // 32 rows x 64 columns: every A row block is fetched and normalised for half as many column tiles (the LN-fused
    // prologue is an L2 burst of the workgroup's whole A and W slabs: 97 -> 73 MB per 2016x768x256 launch) while the row
    // parallelism stays; only where that still leaves >= 2 workgroups per CU
    static const long tln64 = dev_env("AVSEP_TLN64") ? atol(dev_env("AVSEP_TLN64")) : 512;
    const LnTile wide{32, 64, 64, 4};
    if (!found && blocks(wide) >= tln64) pick = wide;
  }
  return pick;
}

template <int BM, int BN, int BK, int NK, int WM = 2, int WN = 2>
hipError_t launch_ln_t(GemmParams p, hipStream_t s) {
  const int nbn = (p.N + BN - 1) / BN;
  p.g_tiles0 = p.alt.M > 0 ? ((p.M + BM - 1) / BM) * nbn : 0;
  hipLaunchKernelGGL((gemm_ln_kernel<BM, BN, BK, NK, WM, WN>), dim3((unsigned)(row_tiles(p, BM) * nbn)), dim3(64 * WM * WN), 0, s, p);
  return hipGetLastError();
}

template <int BM, int BN, int BK>
hipError_t launch_ln_nk(const GemmParams& p, int nk, hipStream_t s) {
  switch (nk) {
    case 1: return launch_ln_t<BM, BN, BK, 1>(p, s);
    case 2: return launch_ln_t<BM, BN, BK, 2>(p, s);
    case 4: return launch_ln_t<BM, BN, BK, 4>(p, s);
    default: return hipErrorInvalidValue;
  }
}

template <int BM, int BN, int WM, int WN>
hipError_t launch_ln_big(const GemmParams& p, int nk, hipStream_t s) {     // the 8- / 16-wave instances: K = 128 or 256
  if (nk == 4) return launch_ln_t<BM, BN, 64, 4, WM, WN>(p, s);
  if (nk == 2) return launch_ln_t<BM, BN, 64, 2, WM, WN>(p, s);
  return hipErrorInvalidValue;
}

hipError_t launch_gemm_ln(const GemmParams& p, hipStream_t s) {
  const LnTile t = pick_ln_tile(p);
  const int nk = p.K / t.bk;
#ifdef AVSEP_DEV
  if (t.bk == 64 && t.waves != 4) {
    if (t.bm == 128 && t.bn == 64 && t.waves == 16) return launch_ln_big<128, 64, 8, 2>(p, nk, s);
    if (t.bm == 64 && t.bn == 128 && t.waves == 16) return launch_ln_big<64, 128, 4, 4>(p, nk, s);
    if (t.bm == 64 && t.bn == 96 && t.waves == 8) return launch_ln_big<64, 96, 4, 2>(p, nk, s);
    if (t.bm == 64 && t.bn == 64 && t.waves == 8) return launch_ln_big<64, 64, 4, 2>(p, nk, s);
    if (t.bm == 128 && t.bn == 32 && t.waves == 8) return launch_ln_big<128, 32, 8, 1>(p, nk, s);
    if (t.bm == 32 && t.bn == 128 && t.waves == 8) return launch_ln_big<32, 128, 2, 4>(p, nk, s);
    return hipErrorInvalidValue;
  }
#endif
  if (t.bk == 64) {
    if (t.bm == 64 && t.bn == 64 && nk <= 4) return launch_ln_nk<64, 64, 64>(p, nk, s);
    if (t.bm == 64 && t.bn == 32 && nk <= 4) return launch_ln_nk<64, 32, 64>(p, nk, s);
    if (t.bm == 32 && t.bn == 64 && nk <= 4) return launch_ln_nk<32, 64, 64>(p, nk, s);
    return launch_ln_nk<32, 32, 64>(p, nk, s);
  }
  return launch_ln_nk<32, 32, 32>(p, nk, s);
}

}  // namespace

bool gemm_ln_staged_supported(int K) {
#ifdef AVSEP_DEV
  return K > 0 && K % 32 == 0 && K <= LN_KMAX;
#else
  (void)K;
  return false;   // the staged form (LayerNorm applied while A is staged, AMODE_LN) is a developer instance
#endif
}

bool gemm_ln_supported(int K) {
  static const float dummy = 0.f;
  GemmParams p{};
  p.K = K;
  p.amode = AMODE_PLAIN;
  p.ln_gamma = p.ln_beta = &dummy;
  return ln_fusable(p);
}

#ifdef AVSEP_DEV
// long contractions on the 64x64 tile: see mfma_chunk.  Off by default since the ring-2 instance (4 workgroups per CU)
// beats it at every K measured; AVSEP_PF_KMIN=1024 restores the round-1 choice
bool fragment_prefetch(const Tile& t, const GemmParams& p) {
  static const int kmin = dev_env("AVSEP_PF_KMIN") ? atoi(dev_env("AVSEP_PF_KMIN")) : 1 << 30;
  return t.bm == 64 && t.bn == 64 && t.bk == 32 && p.amode == AMODE_PLAIN && p.K >= kmin;
}
bool g32_prefetch() {
  static const bool pf = dev_env("AVSEP_G32_PF") ? atoi(dev_env("AVSEP_G32_PF")) != 0 : true;
  return pf;
}
bool ring4_6464() { return dev_env("AVSEP_6464_RING4") != nullptr; }   // read per launch (bit-identity test)
#endif

// The name rocprofv3 prints for the instance launch_gemm() will pick (without the "void (anonymous namespace)::" prefix and
// the argument list), so the live profiler's table joins profiles/*_kernel_stats.csv and pmc_hbm_traffic.json by equality.
const char* gemm_instance_name(const GemmParams& p) {
  static thread_local char buf[64];
  if (p.ln_gamma && !p.ln_stats) {
    const LnTile t = pick_ln_tile(p);
    if (t.waves == 4) snprintf(buf, sizeof buf, "gemm_ln_kernel<%d, %d, %d, %d, 2, 2>", t.bm, t.bn, t.bk, p.K / t.bk);
    else {
      const int wn = (t.bm == 128 && t.bn == 32) ? 1 : (t.bn == 128) ? 4 : 2;
      snprintf(buf, sizeof buf, "gemm_ln_kernel<%d, %d, %d, %d, %d, %d>", t.bm, t.bn, t.bk, p.K / t.bk, t.waves / wn, wn);
    }
    return buf;
  }
  const Tile t = pick_tile(p);
#ifdef AVSEP_DEV
  if (p.ln_stats) {
    snprintf(buf, sizeof buf, "gemm_kernel<%d, %d, %d, %d, false, 0>", t.bm, t.bn, t.bk, (int)AMODE_LN);
    return buf;
  }
  if (is_g32(t)) { snprintf(buf, sizeof buf, "gemm32_kernel<%d, %d, %d, %s>", t.bm, t.bn, p.amode, g32_prefetch() ? "true" : "false"); return buf; }
  if (fragment_prefetch(t, p) && persist_grid<64, 64, 32, true>(p)) { snprintf(buf, sizeof buf, "gemm_persist_kernel<64, 64, 32, true>"); return buf; }
  if (!fragment_prefetch(t, p) && t.bm == 128 && t.bn == 64 && t.bk == 32 && persist_grid<128, 64, 32, false>(p)) { snprintf(buf, sizeof buf, "gemm_persist_kernel<128, 64, 32, false>"); return buf; }
  if (!fragment_prefetch(t, p) && t.bm == 64 && t.bn == 64 && t.bk == 32 && persist_grid<64, 64, 32, false>(p)) { snprintf(buf, sizeof buf, "gemm_persist_kernel<64, 64, 32, false>"); return buf; }
  if (fragment_prefetch(t, p)) { snprintf(buf, sizeof buf, "gemm_kernel<64, 64, 32, 0, true, 0>"); return buf; }
  if (t.bm == 128 && t.bn == 64 && t.bk == 16) { snprintf(buf, sizeof buf, "gemm_kernel<128, 64, 16, 0, false, 2>"); return buf; }
  const bool ring4 = ring4_6464();
#else
  const bool ring4 = false;
#endif
  const int am = p.lnx_c1 ? (int)AMODE_LNX : p.amode;
  if (t.bm == 64 && t.bn == 64 && t.bk == 32 && am == AMODE_LNX)
    snprintf(buf, sizeof buf, "gemm_kernel<64, 64, 32, %d, false, 2>", am);
  else if (p.lnx_c1)
    snprintf(buf, sizeof buf, "gemm_kernel<%d, %d, %d, %d, false, 0>", t.bm, t.bn, t.bk, am);
  else if (t.bm == 64 && t.bn == 64 && t.bk == 32 && p.amode == AMODE_PLAIN && !ring4)
    snprintf(buf, sizeof buf, "gemm_kernel<64, 64, 32, 0, false, 2>");
  else snprintf(buf, sizeof buf, "gemm_kernel<%d, %d, %d, %d, false, 0>", t.bm, t.bn, t.bk, p.amode);
  return buf;
}

hipError_t launch_gemm_impl(GemmParams p, hipStream_t s);

#ifdef AVSEP_DEV
// AVSEP_GEMM_DBG: run the launch with the stamp buffer, wait, print where a workgroup's life goes (10 ns ticks)
hipError_t launch_gemm_dbg(const GemmParams& p_in, hipStream_t s) {
  static unsigned long long* buf = nullptr;
  const size_t cap = (size_t)1 << 20;   // workgroups
  if (!buf && hipMalloc(reinterpret_cast<void**>(&buf), cap * 8 * sizeof(unsigned long long)) != hipSuccess) return hipErrorOutOfMemory;
  GemmParams p = p_in;
  p.dbg = buf;
  (void)hipMemsetAsync(buf, 0, cap * 8 * sizeof(unsigned long long), s);
  hipError_t e = launch_gemm_impl(p, s);
  if (e != hipSuccess) return e;
  (void)hipStreamSynchronize(s);
  static int shown = 0;
  static const bool all = getenv("AVSEP_GEMM_DBG") && (!strcmp(getenv("AVSEP_GEMM_DBG"), "all") || !strcmp(getenv("AVSEP_GEMM_DBG"), "cu"));
  ++shown;
  if (all ? shown > 400 : shown % 16 != 9) return hipSuccess;   // "all": every launch (capped); else one report per 16
  std::vector<unsigned long long> h(cap * 8);
  (void)hipMemcpy(h.data(), buf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  unsigned long long t0 = ~0ull, t1 = 0;
  size_t n = 0;
  double pro = 0, loop = 0, epi = 0, drain = 0;
  std::vector<std::pair<unsigned long long, unsigned long long>> spans;
  for (size_t b = 0; b < cap; ++b) {
    const unsigned long long* r = &h[b * 8];
    if (!r[0] || !r[4]) continue;
    ++n;
    t0 = std::min(t0, r[0]); t1 = std::max(t1, r[4]);
    pro += r[1] - r[0]; loop += r[2] - r[1]; epi += r[3] - r[2]; drain += r[4] - r[3];
    spans.emplace_back(r[0], r[4]);
  }
  if (!n) return hipSuccess;
  // resident workgroups over time, sampled: how long is the chip under-filled at the start / end of the launch?
  const double span = (double)(t1 - t0);
  fprintf(stderr, "[gemm dbg] %s M=%d N=%d K=%d: %zu workgroups, launch span %.2f us; mean per workgroup: prologue %.2f  loop %.2f  "
                  "epilogue(issue) %.2f  store drain %.2f us (life %.2f us)\n", gemm_instance_name(p_in), p.M, p.N, p.K, n, span / 100,
          pro / n / 100, loop / n / 100, epi / n / 100, drain / n / 100, (pro + loop + epi + drain) / n / 100);
  const int NB = 24;
  fprintf(stderr, "[gemm dbg] resident workgroups at %d sample points: ", NB);
  for (int i = 0; i < NB; ++i) {
    const unsigned long long t = t0 + (unsigned long long)(span * (i + 0.5) / NB);
    size_t c = 0;
    for (auto& sp : spans) c += (sp.first <= t && t < sp.second);
    fprintf(stderr, "%zu ", c);
  }
  fprintf(stderr, "\n");
  if (getenv("AVSEP_GEMM_DBG") && !strcmp(getenv("AVSEP_GEMM_DBG"), "cu")) {
    // every workgroup that ran on the CU of workgroup 0 (HW_ID cu / sh / se fields + XCC id), in start order
    const unsigned long long key0 = (h[5] & 0xff00ull) | (h[5] >> 32 << 16);
    std::vector<size_t> ids;
    for (size_t b = 0; b < cap; ++b) {
      const unsigned long long* r = &h[b * 8];
      if (r[0] && r[4] && (((r[5] & 0xff00ull) | (r[5] >> 32 << 16)) == key0)) ids.push_back(b);
    }
    std::sort(ids.begin(), ids.end(), [&](size_t a, size_t b) { return h[a * 8] < h[b * 8]; });
    fprintf(stderr, "[gemm dbg] one CU, %zu workgroups (us from launch start): id  start | prologue end | loop end | stores issued | drained\n", ids.size());
    for (size_t b : ids) {
      const unsigned long long* r = &h[b * 8];
      fprintf(stderr, "[gemm dbg]   %6zu  %8.2f %8.2f %8.2f %8.2f %8.2f\n", b, (r[0] - t0) / 100.0, (r[1] - t0) / 100.0, (r[2] - t0) / 100.0,
              (r[3] - t0) / 100.0, (r[4] - t0) / 100.0);
    }
  }
  return hipSuccess;
}
#endif  // AVSEP_DEV

hipError_t launch_gemm(const GemmParams& p_in, hipStream_t s) {
  GemmParams p = p_in;
#ifdef AVSEP_DEV
  static const bool epi_general = getenv("AVSEP_EPI_GENERAL") != nullptr;   // A/B: block-by-block epilogue
  p.epi_general = (epi_general || p_in.epi_general || (p_in.C2 && getenv("AVSEP_MASK_GENERAL"))) ? 1 : 0;   // (mask head only: A/B)
  static const bool dbg = getenv("AVSEP_GEMM_DBG") != nullptr;
  if (dbg && !p.dbg) return launch_gemm_dbg(p, s);
  static const bool no_remap = getenv("AVSEP_NO_XCD_REMAP") != nullptr;     // A/B: launch-order tiles
  p.no_xcd_remap = no_remap ? 1 : 0;
#endif
  return launch_gemm_impl(p, s);
}

#ifdef AVSEP_DEV
// Two problems that differ only in their operands and row count, as ONE launch (GemmParams::alt).
hipError_t launch_gemm_pair(const GemmParams& p0, const GemmParams& p1, hipStream_t s) {
  const bool same = p0.N == p1.N && p0.K == p1.K && p0.lda == p1.lda && p0.ldw == p1.ldw && p0.ldc == p1.ldc &&
                    p0.ldr == p1.ldr && p0.amode == AMODE_PLAIN && p1.amode == AMODE_PLAIN && p0.act == p1.act &&
                    (p0.bias == nullptr) == (p1.bias == nullptr) && (p0.R == nullptr) == (p1.R == nullptr) &&
                    (p0.ln_gamma == nullptr) == (p1.ln_gamma == nullptr) && p0.ln_eps == p1.ln_eps && !p0.C2 && !p1.C2 &&
                    !p0.ln_stats && !p1.ln_stats && p0.ksplit <= 1 && p1.ksplit <= 1 && p0.mag_F == 0 && p1.mag_F == 0 &&
                    p0.alt.M <= 0 && p1.alt.M <= 0;
  if (!same || p1.M <= 0) return hipErrorInvalidValue;
  GemmParams p = p0;
  p.alt.A = p1.A; p.alt.W = p1.W; p.alt.bias = p1.bias; p.alt.R = p1.R; p.alt.rperiod = p1.rperiod;
  p.alt.ln_gamma = p1.ln_gamma; p.alt.ln_beta = p1.ln_beta; p.alt.C = p1.C; p.alt.M = p1.M;
  return launch_gemm(p, s);
}
#endif  // AVSEP_DEV

hipError_t launch_gemm_impl(GemmParams p, hipStream_t s) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0 || (p.K & 31)) return hipErrorInvalidValue;
#ifdef AVSEP_DEV
  // developer A/B (AVSEP_GEMM_SPLIT=<min 128x128 tiles>): avsep_op_linear & co. on the split-precision GEMM (the forward's own
  // rule is in avsep_api.hip run_gemm)
  if (const char* e = getenv("AVSEP_GEMM_SPLIT")) {
    const long tiles = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    if (gemm_split_supported(p) && tiles >= atol(e)) return launch_gemm_split(p, s);
  }
#endif
  if ((p.amode == AMODE_TAPS3 || p.amode == AMODE_CONV2D) && (p.Kt <= 0 || (p.Kt & 31))) return hipErrorInvalidValue;
  if (p.amode == AMODE_TAPS3 && !p.zeros) return hipErrorInvalidValue;
  if (p.amode == AMODE_FRAMES && (p.T <= 0 || p.frame_hop <= 0 || (p.frame_hop & 3) || p.frame_len <= 0 || (p.frame_len & 3)))
    return hipErrorInvalidValue;
  if (p.mag_F > 0 && ((p.N & 1) || p.N != 2 * p.mag_F || p.T <= 0 || p.bias || p.R || p.C2 || p.act != ACT_NONE))
    return hipErrorInvalidValue;
  if (p.ksplit > 1) {
    if (p.amode != AMODE_PLAIN || p.bias || p.R || p.C2 || p.ln_gamma || p.act != ACT_NONE || p.alt.M > 0) return hipErrorInvalidValue;
    if (p.kchunk <= 0 || (p.kchunk & 63) || (long long)(p.ksplit - 1) * p.kchunk >= p.K) return hipErrorInvalidValue;
  }
#ifdef AVSEP_DEV
  if (p.alt.M <= 0) p.alt.M = 0;
#endif
  if (p.drop_p < 0.0f || p.drop_p >= 1.0f || (p.drop_p > 0.0f && (p.ldc != p.N || p.C2 || p.mag_F > 0 || p.ksplit > 1)))
    return hipErrorInvalidValue;
  if (p.lnx_c1) {                                      // LayerNorm in the epilogue (GemmParams::lnx_c1)
    if (!p.lnx_c2 || p.amode != AMODE_PLAIN || p.bias || p.R || p.C2 || p.ln_gamma || p.ln_stats || p.ksplit > 1 || p.mag_F > 0 ||
        p.drop_p > 0.0f || p.alt.M > 0 || (p.N & 3) || (p.ldc & 3))
      return hipErrorInvalidValue;
    p.amode = AMODE_LNX;
    if (pick_tile(p).bk != 32) return hipErrorInvalidValue;   // (a developer override: the statistics' summation order is BK's)
  }
  if (p.ln_gamma && p.ln_stats) {                      // LayerNorm applied while staging A, statistics given
#ifdef AVSEP_DEV
    if (!p.ln_beta || p.amode != AMODE_PLAIN || !gemm_ln_staged_supported(p.K) || p.ksplit > 1 || p.alt.M > 0) return hipErrorInvalidValue;
    p.amode = AMODE_LN;
#else
    return hipErrorNotSupported;                       // developer instance (make dev)
#endif
  } else if (p.ln_gamma) {
    if (!ln_fusable(p)) return hipErrorInvalidValue;   // callers check gemm_ln_supported() first
    return launch_gemm_ln(p, s);
  }
  const Tile t = pick_tile(p);
  if (p.K % t.bk) return hipErrorInvalidValue;
#ifdef AVSEP_DEV
  if (is_g32(t)) {
    if ((p.amode != AMODE_PLAIN && p.amode != AMODE_TAPS3) || p.mag_F > 0 || p.ksplit > 1 || t.bk != 32 || t.bn != 128 || p.alt.M > 0)
      return hipErrorInvalidValue;
    const bool pf = g32_prefetch();
#define AVSEP_G32(BM_, AM_)                                                              \
    if (t.bm == BM_ && p.amode == AM_)                                                   \
      return pf ? launch32_t<BM_, 128, AM_, true>(p, s) : launch32_t<BM_, 128, AM_, false>(p, s);
    AVSEP_G32(128, AMODE_PLAIN) AVSEP_G32(128, AMODE_TAPS3) AVSEP_G32(256, AMODE_PLAIN) AVSEP_G32(256, AMODE_TAPS3)
#undef AVSEP_G32
    return hipErrorInvalidValue;
  }
  if (p.alt.M <= 0) {
    hipError_t pe = hipSuccess;
    if (fragment_prefetch(t, p)) {
      if (try_launch_persist<64, 64, 32, true>(p, s, &pe)) return pe;
    } else if (t.bm == 128 && t.bn == 64 && t.bk == 32) {
      if (try_launch_persist<128, 64, 32, false>(p, s, &pe)) return pe;
    } else if (t.bm == 64 && t.bn == 64 && t.bk == 32) {
      if (try_launch_persist<64, 64, 32, false>(p, s, &pe)) return pe;
    }
  }
  if (dev_env("AVSEP_GEMM_DMA") && p.amode == AMODE_PLAIN && p.ksplit <= 1) {       // LDS-DMA staging (A/B measurement)
    if (t.bm == 128 && t.bn == 64 && t.bk == 32) return launch_dma_t<128, 64, 32>(p, s);
    if (t.bm == 128 && t.bn == 64 && t.bk == 16) return launch_dma_t<128, 64, 16>(p, s);
    if (t.bm == 64 && t.bn == 64 && t.bk == 32) return launch_dma_t<64, 64, 32>(p, s);
  }
  if (t.bm == 128 && t.bn == 64 && t.bk == 16 && p.amode == AMODE_PLAIN)   // four workgroups per CU, measured equal
    return launch_t<128, 64, 16, AMODE_PLAIN, false, 2>(p, s);
  if (fragment_prefetch(t, p)) return launch_t<64, 64, 32, AMODE_PLAIN, true>(p, s);
  const bool ring4 = ring4_6464();
#else
  const bool ring4 = false;
#endif
  // 64x64x32, plain A: register ring of depth 2 instead of 4 -> 104 registers, 4 workgroups per CU instead of 3
  // (profiles/r02_ab_ring2_64x64.txt: +1..3 % on the N = 512 shapes)
  if (t.bm == 64 && t.bn == 64 && t.bk == 32 && p.amode == AMODE_PLAIN && !ring4)
    return launch_t<64, 64, 32, AMODE_PLAIN, false, 2>(p, s);
  if (t.bm == 64 && t.bn == 64 && t.bk == 32 && p.amode == AMODE_LNX) return launch_t<64, 64, 32, AMODE_LNX, false, 2>(p, s);
#define AVSEP_CASE(BM_, BN_, BK_, AM_) \
  if (t.bm == BM_ && t.bn == BN_ && t.bk == BK_ && p.amode == AM_) return launch_t<BM_, BN_, BK_, AM_>(p, s);
#ifdef AVSEP_DEV
#define AVSEP_MODES(BM_, BN_, BK_) \
  AVSEP_CASE(BM_, BN_, BK_, AMODE_PLAIN) AVSEP_CASE(BM_, BN_, BK_, AMODE_TAPS3) AVSEP_CASE(BM_, BN_, BK_, AMODE_CONV2D) \
  AVSEP_CASE(BM_, BN_, BK_, AMODE_LN)
#else
#define AVSEP_MODES(BM_, BN_, BK_) \
  AVSEP_CASE(BM_, BN_, BK_, AMODE_PLAIN) AVSEP_CASE(BM_, BN_, BK_, AMODE_TAPS3) AVSEP_CASE(BM_, BN_, BK_, AMODE_CONV2D)
#endif
  AVSEP_MODES(128, 64, 32)
  AVSEP_MODES(64, 64, 32)
  AVSEP_MODES(64, 32, 32)
  AVSEP_MODES(64, 32, 64)
  AVSEP_MODES(32, 32, 32)
  AVSEP_MODES(32, 32, 64)
  AVSEP_CASE(32, 32, 64, AMODE_FRAMES) AVSEP_CASE(64, 32, 64, AMODE_FRAMES) AVSEP_CASE(64, 64, 32, AMODE_FRAMES)
  AVSEP_CASE(32, 32, 32, AMODE_FRAMES) AVSEP_CASE(64, 32, 32, AMODE_FRAMES) AVSEP_CASE(128, 64, 32, AMODE_FRAMES)
  AVSEP_CASE(32, 64, 32, AMODE_LNX) AVSEP_CASE(32, 32, 32, AMODE_LNX) AVSEP_CASE(64, 32, 32, AMODE_LNX)
  AVSEP_CASE(128, 64, 32, AMODE_LNX)
  AVSEP_CASE(32, 64, 32, AMODE_TAPS3)
#ifdef AVSEP_DEV   // reachable only through the tile overrides
  AVSEP_CASE(32, 64, 32, AMODE_PLAIN) AVSEP_CASE(32, 64, 64, AMODE_PLAIN)
  AVSEP_MODES(64, 64, 64)
  AVSEP_CASE(32, 32, 128, AMODE_PLAIN)
  AVSEP_CASE(32, 32, 128, AMODE_TAPS3)
#endif
#undef AVSEP_MODES
#undef AVSEP_CASE
  return hipErrorInvalidValue;
}

// dW = dY^T X straight from the row-major activations (no transposes).  Returns the number of r-slices used through
// *slices (1 = dw written directly; > 1 = `partial` holds the slices, the caller sums them).
// Tile and r-slices of the weight-gradient launch (tools/wgrad_sweep.py, profiles/r02_wgrad_sweep.txt).  One slice
// walks all R rows serially (160 us at R = 4016 whatever the tile count), so the contraction is cut until ~1024
// workgroups exist; 64x64 tiles from 128 tiles up (or 64 with a long contraction), 32x32 below.
// (AVSEP_WGRAD_TILE / AVSEP_WGRAD_SLICES: developer sweeps)
static int wgrad_tile(int N, int K, int R) {
  if (const char* e = dev_env("AVSEP_WGRAD_TILE")) return atoi(e) == 64 ? 64 : 32;
  const long tiles64 = (long)((N + 63) / 64) * ((K + 63) / 64);
  return (tiles64 >= 128 || (tiles64 >= 64 && R >= 2048)) ? 64 : 32;
}
int wgrad_slices(int N, int K, int R) {
  const int bt = wgrad_tile(N, K, R);
  const long tiles = (long)((N + bt - 1) / bt) * ((K + bt - 1) / bt);
  long want;
  if (const char* e = dev_env("AVSEP_WGRAD_SLICES")) {
    want = atol(e);
  } else {
    if (tiles >= 1024 || R < 1024) return 1;
    want = std::min<long>(1024 / tiles, R / 256);
  }
  if (want < 2) return 1;
  const int rchunk = (int)(((R + want - 1) / want + 31) / 32 * 32);
  return (R + rchunk - 1) / rchunk;
}

int wgrad_tiles(int N, int K, int R) {
  const int bt = wgrad_tile(N, K, R);
  return ((N + bt - 1) / bt) * ((K + bt - 1) / bt);
}

hipError_t launch_wgrad(const float* dy, int ldy, const float* x, int ldx, float* out, int N, int K, int R, int slices,
                        bool with_bias, hipStream_t s, float* merged, unsigned* counters) {
  if (N <= 0 || K <= 0 || R <= 0 || (N & 3) || (K & 3) || (ldy & 3) || (ldx & 3) || slices < 1) return hipErrorInvalidValue;
  if (merged && (slices < 2 || !counters)) return hipErrorInvalidValue;
  WgradParams p{dy, x, out, R, N, K, ldy, ldx, 0, with_bias ? 1 : 0, merged, counters};
  p.rchunk = slices > 1 ? (((R + slices - 1) / slices + 31) / 32 * 32) : ((R + 31) / 32 * 32);
  const long tiles64 = (long)((N + 63) / 64) * ((K + 63) / 64);
  if (wgrad_tile(N, K, R) == 64) {
    hipLaunchKernelGGL((wgrad_kernel<64, 64>), dim3((unsigned)tiles64, slices), dim3(256), 0, s, p);
  } else {
    const long tiles = (long)((N + 31) / 32) * ((K + 31) / 32);
    hipLaunchKernelGGL((wgrad_kernel<32, 32>), dim3((unsigned)tiles, slices), dim3(256), 0, s, p);
  }
  return hipGetLastError();
}
