// gemm_split.hip -- split-precision GEMM (round 4): fp32 operands, fp32 result, the products on the bf16 matrix pipe.
//
// The fp32 MFMA (v_mfma_f32_16x16x4_f32: 256 FLOP/clk/CU, 157 TFLOP/s) is the ceiling every GEMM of this path sits under, and three
// rounds of work on the 128x64 tile stand at 0.78 of it.  SURVEY.md §7 names the one other road that keeps the 1e-4 mask contract:
// "a split-precision (hi+lo) scheme validated against the golden vectors".  This file builds it:
//
//   * every fp32 operand x is cut into THREE bf16 terms by truncation, x = hi + mid + lo (+ < 2^-24 |x|): hi = the upper 16 bits of
//     x, mid = the upper 16 bits of (x - hi), lo = the upper 16 bits of (x - hi - mid); the two subtractions are exact.
//   * x w = SIX bf16 products -- (hi,hi) (hi,mid) (mid,hi) (mid,mid) (hi,lo) (lo,hi) -- each exact in fp32 (8 x 8 mantissa bits),
//     accumulated in fp32 by v_mfma_f32_16x16x32_bf16, which runs at 16x the fp32 MFMA's rate: 6 / 16 of the fp32 matrix time.
//     The dropped terms (mid,lo) (lo,mid) (lo,lo) are < 2^-23 relative: CPU emulation on model-like operands (K = 256 ... 2048)
//     puts the split's own error at 6e-8 of max|y|, below the 5e-7 of an fp32 GEMM's accumulation, and the config-1 model's masks
//     at 4.1e-7 from float64 with every Linear computed this way (fp32 path: 5.3e-7) -- profiles/r04_split_precision_accuracy.txt.
//   * 128 x 128 x 32 tile, 256 threads = 2 x 2 waves of 64 x 64; operands are split on their way to LDS (5.5 VALU per element) into
//     three [rows][32] bf16 planes per operand, 16-byte slots XOR-swizzled for this chip's per-instruction LDS lane groups; a fragment is one ds_read_b128 (8 bf16
//     along k); 96 MFMAs per wave and chunk.  The MFMA's C / D layout is the fp32 16x16 one, operands swapped like gemm_kernel's
//     (D^T), so the fused epilogues of gemm_tile.h are used unchanged.
//
// PLAIN or 3-tap (Conv1d) A operand; the straight-line bias / activation / residual epilogue or the mask head's.  Not bit-identical to the fp32 kernels (another rounding of the same
// products): parity is by tolerance against float64 (tests/test_gpu_parity.py::test_op_linear_split_precision).
#include "kernels.h"
#include "gemm_tile.h"
#include "split_terms.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// three bf16 planes of 4 fp32 values: plane p of elements (e0,e1) and (e2,e3) as two dwords each (low half = the even element)
__device__ __forceinline__ void split4(const f32x4 v, unsigned (&hi)[2], unsigned (&mid)[2], unsigned (&lo)[2]) {
  split_pair(f32x2{v[0], v[1]}, hi[0], mid[0], lo[0]);
  split_pair(f32x2{v[2], v[3]}, hi[1], mid[1], lo[1]);
}

constexpr int SBM = 128, SBN = 128, SBK = 32;
constexpr int PLANE = SBM * SBK;          // bf16 elements of one plane of one operand tile (rows x 32)

// (W pre-packed into its three planes by the weight packer -- 6 B per weight from L2, no VALU for W -- was built and measured 4.5 %
// SLOWER in the cfg3 / cfg5 forward than splitting W on its way to LDS like A: profiles/r04_ab_split_gemm_w_planes.txt.  The
// kernel waits for operand bytes, not for the VALU.)
__global__ __launch_bounds__(256, 2) void gemm_split_kernel(const GemmParams pin) {
  GemmParams p = pin;
  // [operand A|W][plane hi|mid|lo][row][32 bf16]: 6 x 8 KB = 48 KB; ONE buffer, two barriers per chunk (the next chunk's global
  // loads are in flight under this chunk's MFMAs; the split + LDS writes follow them)
  __shared__ __attribute__((aligned(16))) unsigned short lds[6 * PLANE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nbn = (p.N + SBN - 1) / SBN;
  int tile = xcd_tile(p);
  const int bm = tile / nbn, bn = tile - bm * nbn;
  const int m0 = bm * SBM, n0 = bn * SBN;

  // staging: thread -> (row, 16-float half = tid % 2) of both operand tiles.  LDS banking on gfx950 is per instruction
  // (MI355X_MICROARCH.md, LDS): ds_read_b128 is served in the lane groups {0-3,12-15,20-27} {4-11,16-19,28-31} (+32), banks
  // (a/4) mod 64; ds_write_b128 in groups of 8 consecutive lanes, banks (a/4) mod 32.  With 64-byte rows that asks for
  //   * the slot swizzle sw(row) = -(row >> 2) & 3 (gemm.hip's (row >> 2) & 3 is 2-way conflicted under these read groups:
  //     SQ_LDS_BANK_CONFLICT was 50 % of SQ_LDS_IDX_ACTIVE in the first version, profiles/r04_pmc_gemm_split.txt), and
  //   * 8 consecutive lanes writing rows {b, b+1, b+4, b+5}, not {b .. b+3} (rows b and b+2 share their 128-byte bank window and
  //     their slot pair): bits 1 and 2 of the row index are swapped in the thread -> row map.
  const int su = tid >> 1, shalf = tid & 1;
  const int srow = (su & ~6) | ((su & 2) << 1) | ((su & 4) >> 1);
  const int am = min(m0 + srow, p.M - 1), wnr = min(n0 + srow, p.N - 1);
  const float* a_src = p.A + (size_t)am * p.lda + 16 * shalf;
  // TAPS3 (nn.Conv1d k = 3, pad 1 as ONE GEMM over K = 3 Kt, like gemm_kernel's AMODE_TAPS3): the rows of tap 0 / 2 are the
  // sequence's previous / next row, or the zero row (GemmParams::zeros) outside the sequence; block-uniform tap per chunk
  const bool taps = p.amode == AMODE_TAPS3;
  const float *a_prev = a_src, *a_next = a_src;
  if (taps) {
    const int t = am % p.T;
    const float* zrow = p.zeros + 16 * shalf;
    a_prev = t > 0 ? a_src - p.lda : zrow;
    a_next = t + 1 < p.T ? a_src + p.lda : zrow;
  }
  const int cpt = taps ? p.Kt / SBK : 1 << 30;                          // chunks per tap
  const float* w_src = p.W + (size_t)wnr * p.ldw + 16 * shalf;
  const int sw = (0 - (srow >> 2)) & 3;
  // byte offsets (inside a plane) of this thread's two 16-byte slots: logical slots 2*shalf, 2*shalf + 1
  const int st0 = srow * 64 + (((2 * shalf) ^ sw) << 4), st1 = srow * 64 + (((2 * shalf + 1) ^ sw) << 4);

  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  f32x4 ra[4], rw[4];
  auto load_chunk = [&](int kc) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (taps) {
        const int tap = kc / cpt, sub = kc - tap * cpt;                  // uniform
        const float* row = tap == 0 ? a_prev : tap == 1 ? a_src : a_next;
        ra[j] = *reinterpret_cast<const f32x4*>(row + sub * SBK + 4 * j);
      } else {
        ra[j] = *reinterpret_cast<const f32x4*>(a_src + kc * SBK + 4 * j);
      }
      rw[j] = *reinterpret_cast<const f32x4*>(w_src + kc * SBK + 4 * j);
    }
  };
  auto store_chunk = [&]() {
    char* base = reinterpret_cast<char*>(lds);
#pragma unroll
    for (int opnd = 0; opnd < 2; ++opnd) {
      unsigned hi[8], mid[8], lo[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned h2[2], m2[2], l2[2];
        split4(opnd ? rw[j] : ra[j], h2, m2, l2);
        hi[2 * j] = h2[0]; hi[2 * j + 1] = h2[1];
        mid[2 * j] = m2[0]; mid[2 * j + 1] = m2[1];
        lo[2 * j] = l2[0]; lo[2 * j + 1] = l2[1];
      }
      char* ob = base + opnd * 3 * PLANE * 2;
      *reinterpret_cast<u4*>(ob + 0 * PLANE * 2 + st0) = u4{hi[0], hi[1], hi[2], hi[3]};
      *reinterpret_cast<u4*>(ob + 0 * PLANE * 2 + st1) = u4{hi[4], hi[5], hi[6], hi[7]};
      *reinterpret_cast<u4*>(ob + 1 * PLANE * 2 + st0) = u4{mid[0], mid[1], mid[2], mid[3]};
      *reinterpret_cast<u4*>(ob + 1 * PLANE * 2 + st1) = u4{mid[4], mid[5], mid[6], mid[7]};
      *reinterpret_cast<u4*>(ob + 2 * PLANE * 2 + st0) = u4{lo[0], lo[1], lo[2], lo[3]};
      *reinterpret_cast<u4*>(ob + 2 * PLANE * 2 + st1) = u4{lo[4], lo[5], lo[6], lo[7]};
    }
  };

  // fragment coordinates: lane (r = lane & 15, q = lane >> 4) reads the 8 bf16 k = 8q .. 8q + 7 of row 16 blk + r: logical slot q
  const int fr = lane & 15, fq = lane >> 4;
  int a_fo[4], w_fo[4];                                                 // byte offsets inside a plane
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wm * 64 + 16 * i + fr;
    a_fo[i] = r * 64 + ((fq ^ ((0 - (r >> 2)) & 3)) << 4);
    const int c = wn * 64 + 16 * i + fr;
    w_fo[i] = c * 64 + ((fq ^ ((0 - (c >> 2)) & 3)) << 4);
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / SBK;
  load_chunk(0);
  store_chunk();
  __syncthreads();
  for (int kc = 0; kc < nk; ++kc) {
    if (kc + 1 < nk) load_chunk(kc + 1);                                // block-uniform; in flight under the MFMAs below
    const char* ab = reinterpret_cast<const char*>(lds);
    const char* wb = ab + 3 * PLANE * 2;
    // smallest products first: (hi,lo) (lo,hi) (mid,mid) (mid,hi) (hi,mid) (hi,hi).  Every plane's fragments are read from LDS
    // ONCE per chunk (24 ds_read_b128 per wave; re-reading them per product doubled that and put 8 waves per CU above the LDS's
    // 128 B/clk): hi and mid of both operands stay in registers across the products that use them, lo passes through.
    auto frag = [&](const char* base, int plane, const int (&off)[4], bf16x8 (&f)[4]) {
#pragma unroll
      for (int i = 0; i < 4; ++i) f[i] = *reinterpret_cast<const bf16x8*>(base + plane * PLANE * 2 + off[i]);
    };
    auto mma = [&](const bf16x8 (&fwp)[4], const bf16x8 (&fap)[4]) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fwp[j], fap[i], acc[i][j], 0, 0, 0);     // D^T: see gemm_epilogue
    };
    bf16x8 a_hi[4], w_hi[4], a_mid[4], w_mid[4], t_lo[4];
    frag(ab, 0, a_fo, a_hi);
    frag(wb, 2, w_fo, t_lo);
    mma(t_lo, a_hi);                                  // (hi, lo)
    frag(wb, 0, w_fo, w_hi);
    frag(ab, 2, a_fo, t_lo);
    mma(w_hi, t_lo);                                  // (lo, hi)
    frag(ab, 1, a_fo, a_mid);
    frag(wb, 1, w_fo, w_mid);
    mma(w_mid, a_mid);                                // (mid, mid)
    mma(w_hi, a_mid);                                 // (mid, hi)
    mma(w_mid, a_hi);                                 // (hi, mid)
    mma(w_hi, a_hi);                                  // (hi, hi)
    __syncthreads();                                                    // every wave has read this chunk's planes
    if (kc + 1 < nk) {
      store_chunk();
      __syncthreads();
    }
  }
  gemm_epilogue<4, 4>(p, acc, m0, n0, wm * 64, wn * 64, fr, fq);
}

// ---- the 256 x 128 tile (large M) -------------------------------------------------------------------------------------------
// gemm_split_kernel alternates two phases per chunk -- 96 MFMAs, then the split of the next chunk (VALU + LDS writes) between two
// barriers -- and leans on the CU's other workgroup to fill the matrix pipe during the second: SQ_VALU_MFMA_BUSY_CYCLES says
// 49 %.  On this chip VALU instructions of ANOTHER wave hardly overlap with a wave's MFMAs, but one or two VALU instructions
// BEHIND each MFMA of the same wave hide about half their cycles (profiles/r04_mfma_bf16_valu_share.txt).  This kernel therefore
// has no phases: 512 threads = 4 x 2 waves of 64 x 64, LDS double-buffered (2 x 72 KB), ONE barrier per chunk, and inside a chunk
// every product's 16 MFMAs are followed in program order by one sixth of the next chunk's split (22 VALU), its LDS writes into the
// other buffer, and the global loads of the chunk after that into the registers just consumed -- so a load has one whole chunk
// (~0.8 us) to land.  Per thread and chunk 24 operand elements instead of 32 (the 256-row tile shares W among more rows).
// The loop body is branch-free: the last chunks split / reload clamped data into the buffer nobody reads.
constexpr int S2BM = 256, S2BN = 128;
constexpr int S2_APL = S2BM * SBK * 2, S2_WPL = S2BN * SBK * 2;   // bytes of one plane
constexpr int S2_BUF = 3 * S2_APL + 3 * S2_WPL;                  // 72 KB: [A hi|mid|lo][W hi|mid|lo]

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// 8 fp32 values -> one 16-byte slot of each of the three planes
__device__ __forceinline__ void split8_store(const f32x4 v0, const f32x4 v1, char* hi_p, char* mid_p, char* lo_p) {
  unsigned h0[2], m0[2], l0[2], h1[2], m1[2], l1[2];
  split4(v0, h0, m0, l0);
  split4(v1, h1, m1, l1);
  *reinterpret_cast<u32x4*>(hi_p) = u32x4{h0[0], h0[1], h1[0], h1[1]};
  *reinterpret_cast<u32x4*>(mid_p) = u32x4{m0[0], m0[1], m1[0], m1[1]};
  *reinterpret_cast<u32x4*>(lo_p) = u32x4{l0[0], l0[1], l1[0], l1[1]};
}

// per-thread operand rows of one 256 x 128 tile
struct S2Tile {
  const float *a, *a_prev, *a_next, *w;
};

template <bool TAPS>
__device__ __forceinline__ S2Tile s2_tile_rows(const GemmParams& p, int tile, int nbn, int srow, int shalf, int wrow, int wq) {
  const int bm = tile / nbn, bn = tile - bm * nbn;
  const int am = min(bm * S2BM + srow, p.M - 1), wnr = min(bn * S2BN + wrow, p.N - 1);
  S2Tile t;
  t.a = p.A + (size_t)am * p.lda + 16 * shalf;
  t.a_prev = t.a_next = t.a;
  if (TAPS) {
    const int ts = am % p.T;
    const float* zrow = p.zeros + 16 * shalf;
    t.a_prev = ts > 0 ? t.a - p.lda : zrow;
    t.a_next = ts + 1 < p.T ? t.a + p.lda : zrow;
  }
  t.w = p.W + (size_t)wnr * p.ldw + 8 * wq;
  return t;
}

// Persistent over tiles: with ONE workgroup per CU nothing covers a tile's first loads (~2 us) and its epilogue -- measured as
// 3.3 chunk times per tile, 17 % of a K = 512 tile (K = 512 against K = 2048 at equal flops, profiles/r04_gemm_split_probe_256x128.txt).
// A workgroup therefore walks its tiles in one flat chunk sequence: the last two chunks of a tile load, split and store the
// first two of the NEXT tile, so only the epilogue itself stands between two tiles' MFMAs.  Tiles are dealt so that an XCD's
// workgroups share a contiguous tile range (xcd_tile's rule; blockIdx % 8 = XCD).
template <bool TAPS>
__global__ __launch_bounds__(512, 1) void gemm_split_kernel2(const GemmParams pin) {
  GemmParams p = pin;
  extern __shared__ __attribute__((aligned(16))) char lds2[];           // 2 x S2_BUF
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nbn = (p.N + S2BN - 1) / S2BN;
  const int ntiles = ((p.M + S2BM - 1) / S2BM) * nbn;
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
  if (tile >= tile_end) return;                                          // block-uniform (never with grid <= tiles)

  // A staging as in gemm_split_kernel (thread -> row, 16-float half; bits 1 and 2 of the row swapped); W staging: thread ->
  // (row = tid / 4, 8-float quarter): 8 consecutive lanes write two whole 64-byte rows = every bank once
  const int su = tid >> 1, shalf = tid & 1;
  const int srow = (su & ~6) | ((su & 2) << 1) | ((su & 4) >> 1);
  const int wrow = tid >> 2, wq = tid & 3;
  const int cpt = TAPS ? p.Kt / SBK : 1 << 30;
  const int sw = (0 - (srow >> 2)) & 3;
  const int ast0 = srow * 64 + (((2 * shalf) ^ sw) << 4), ast1 = srow * 64 + (((2 * shalf + 1) ^ sw) << 4);
  const int wst = 3 * S2_APL + wrow * 64 + ((wq ^ ((0 - (wrow >> 2)) & 3)) << 4);

  const int nk = p.K / SBK;                                             // >= 2 (launch_gemm_split)
  f32x4 ra0, ra1, ra2, ra3, rw0, rw1;
  // chunk kc of tile rows T: A pointer (TAPS3: block-uniform tap per chunk)
#define S2_A_ROW(ptr, T, kc)                                                                       \
  const float* ptr;                                                                                \
  if (TAPS) {                                                                                      \
    const int tap_ = (kc) / cpt, sub_ = (kc) - tap_ * cpt;                                         \
    ptr = (tap_ == 0 ? T.a_prev : tap_ == 1 ? T.a : T.a_next) + sub_ * SBK;                        \
  } else {                                                                                         \
    ptr = T.a + (kc) * SBK;                                                                        \
  }
#define S2_LD4(ptr, j) (*reinterpret_cast<const f32x4*>((ptr) + 4 * (j)))

  const int fr = lane & 15, fq = lane >> 4;
  int a_fo[4], w_fo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wm * 64 + 16 * i + fr;
    a_fo[i] = r * 64 + ((fq ^ ((0 - (r >> 2)) & 3)) << 4);
    const int c = wn * 64 + 16 * i + fr;
    w_fo[i] = 3 * S2_APL + c * 64 + ((fq ^ ((0 - (c >> 2)) & 3)) << 4);
  }

  S2Tile cur = s2_tile_rows<TAPS>(p, tile, nbn, srow, shalf, wrow, wq);
  // prologue (first tile only): chunk 0 into buffer 0, chunk 1 into the registers
  {
    S2_A_ROW(ap, cur, 0)
    ra0 = S2_LD4(ap, 0); ra1 = S2_LD4(ap, 1); ra2 = S2_LD4(ap, 2); ra3 = S2_LD4(ap, 3);
    rw0 = S2_LD4(cur.w, 0); rw1 = S2_LD4(cur.w, 1);
  }
  split8_store(ra0, ra1, lds2 + ast0, lds2 + S2_APL + ast0, lds2 + 2 * S2_APL + ast0);
  split8_store(ra2, ra3, lds2 + ast1, lds2 + S2_APL + ast1, lds2 + 2 * S2_APL + ast1);
  split8_store(rw0, rw1, lds2 + wst, lds2 + S2_WPL + wst, lds2 + 2 * S2_WPL + wst);
  {
    S2_A_ROW(ap, cur, 1)
    ra0 = S2_LD4(ap, 0); ra1 = S2_LD4(ap, 1); ra2 = S2_LD4(ap, 2); ra3 = S2_LD4(ap, 3);
    rw0 = S2_LD4(cur.w + SBK, 0); rw1 = S2_LD4(cur.w + SBK, 1);
  }
  __syncthreads();

#define S2_FRAG_A(plane, f)                                                                                   \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) f[i] = *reinterpret_cast<const bf16x8*>(rb + (plane) * S2_APL + a_fo[i]);
#define S2_FRAG_W(plane, f)                                                                                   \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) f[i] = *reinterpret_cast<const bf16x8*>(rb + (plane) * S2_WPL + w_fo[i]);
#define S2_MMA(fwp, fap)                                                                                      \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j)                 \
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fwp[j], fap[i], acc[i][j], 0, 0, 0);
  // One chunk: the six products over buffer `par`; the registers (the chunk after it) are split into the other buffer and
  // reloaded from ap / wp (two chunks ahead).  Four scheduling groups (the compiler interleaves inside a group, never across
  // the fences): without them it hoists the three splits to the top of the chunk, where their loads are a third of a chunk old.
#define S2_CHUNK(ap, wp)                                                                     \
  {                                                                                          \
    const char* rb = lds2 + par * S2_BUF;                                                    \
    char* sb = lds2 + (par ^ 1) * S2_BUF;                                                    \
    par ^= 1;                                                                                \
    bf16x8 a_hi[4], w_hi[4], a_mid[4], w_mid[4], t_lo[4];                                    \
    S2_FRAG_A(0, a_hi)                                                                       \
    S2_FRAG_W(2, t_lo)                                                                       \
    S2_MMA(t_lo, a_hi) /* (hi, lo) */                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    S2_FRAG_W(0, w_hi)                                                                       \
    S2_FRAG_A(2, t_lo)                                                                       \
    S2_MMA(w_hi, t_lo) /* (lo, hi) */                                                        \
    split8_store(ra0, ra1, sb + ast0, sb + S2_APL + ast0, sb + 2 * S2_APL + ast0);           \
    ra0 = S2_LD4(ap, 0);                                                                     \
    ra1 = S2_LD4(ap, 1);                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    S2_FRAG_A(1, a_mid)                                                                      \
    S2_FRAG_W(1, w_mid)                                                                      \
    S2_MMA(w_mid, a_mid) /* (mid, mid) */                                                    \
    S2_MMA(w_hi, a_mid)  /* (mid, hi) */                                                     \
    split8_store(ra2, ra3, sb + ast1, sb + S2_APL + ast1, sb + 2 * S2_APL + ast1);           \
    ra2 = S2_LD4(ap, 2);                                                                     \
    ra3 = S2_LD4(ap, 3);                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    S2_MMA(w_mid, a_hi) /* (hi, mid) */                                                      \
    S2_MMA(w_hi, a_hi)  /* (hi, hi) */                                                       \
    split8_store(rw0, rw1, sb + wst, sb + S2_WPL + wst, sb + 2 * S2_WPL + wst);              \
    rw0 = S2_LD4(wp, 0);                                                                     \
    rw1 = S2_LD4(wp, 1);                                                                     \
    __syncthreads();                                                                         \
  }

  int par = 0;                                                          // buffer of the chunk being multiplied
  for (;;) {
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kc = 2; kc < nk; ++kc) {                                   // chunks 0 .. nk - 3 load chunks 2 .. nk - 1 of this tile
      S2_A_ROW(ap, cur, kc)
      const float* wp = cur.w + kc * SBK;
      S2_CHUNK(ap, wp)
    }
    // the last two chunks load (and the very last splits) the first two of the next tile; without one they re-read this
    // tile's and split them into the buffer nobody reads.  Only one tile's row pointers are live at any time.
    const int tile_next = tile + tile_step;
    const bool more = tile_next < tile_end;
    cur = s2_tile_rows<TAPS>(p, more ? tile_next : tile, nbn, srow, shalf, wrow, wq);
    for (int kc = 0; kc < 2; ++kc) {
      S2_A_ROW(ap, cur, kc)
      const float* wp = cur.w + kc * SBK;
      S2_CHUNK(ap, wp)
    }
    {
      const int bm = tile / nbn, bn = tile - bm * nbn;
      gemm_epilogue<4, 4>(p, acc, bm * S2BM, bn * S2BN, wm * 64, wn * 64, fr, fq);
    }
    if (!more) break;
    tile = tile_next;
  }
#undef S2_CHUNK
#undef S2_A_ROW
#undef S2_LD4
#undef S2_FRAG_A
#undef S2_FRAG_W
#undef S2_MMA
}


// ---- the 64 x 64 tile (small problems) ----------------------------------------------------------------------------------------
// The same phase-free chunk as gemm_split_kernel2 for problems that give the 256 x 128 kernel less than half a round of tiles
// (the visual branch: M = 3200 rows; small batches): 256 threads = 2 x 2 waves of 32 x 32, 24 MFMAs per wave and chunk,
// 2 x 24 KB of LDS, three workgroups per CU (they cover each other's first loads and epilogues: one tile per workgroup).
// Per element the same six products in the same order as the two kernels above: a row has the same bits in all three.
constexpr int S3B = 64;
constexpr int S3_PL = S3B * SBK * 2;                                    // bytes of one plane (4 KB)
constexpr int S3_BUF = 6 * S3_PL;                                       // [A hi|mid|lo][W hi|mid|lo]

__global__ __launch_bounds__(256, 3) void gemm_split_kernel3(const GemmParams pin) {
  GemmParams p = pin;
  __shared__ __attribute__((aligned(16))) char lds3[2 * S3_BUF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nbn = (p.N + S3B - 1) / S3B;
  const int tile = xcd_tile(p);
  const int bm = tile / nbn, bn = tile - bm * nbn;
  const int m0 = bm * S3B, n0 = bn * S3B;

  // staging: thread -> (row = tid / 4, 8-float quarter) of both operands: 8 consecutive lanes write two whole 64-byte rows
  const int srow = tid >> 2, sq = tid & 3;
  const float* a_src = p.A + (size_t)min(m0 + srow, p.M - 1) * p.lda + 8 * sq;
  const float* w_src = p.W + (size_t)min(n0 + srow, p.N - 1) * p.ldw + 8 * sq;
  const int ast = srow * 64 + ((sq ^ ((0 - (srow >> 2)) & 3)) << 4), wst = 3 * S3_PL + ast;
  const int nk = p.K / SBK;
  f32x4 ra0, ra1, rw0, rw1;
#define S3_LD4(ptr, j) (*reinterpret_cast<const f32x4*>((ptr) + 4 * (j)))

  const int fr = lane & 15, fq = lane >> 4;
  int a_fo[2], w_fo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = wm * 32 + 16 * i + fr;
    a_fo[i] = r * 64 + ((fq ^ ((0 - (r >> 2)) & 3)) << 4);
    const int c = wn * 32 + 16 * i + fr;
    w_fo[i] = 3 * S3_PL + c * 64 + ((fq ^ ((0 - (c >> 2)) & 3)) << 4);
  }
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  ra0 = S3_LD4(a_src, 0); ra1 = S3_LD4(a_src, 1);
  rw0 = S3_LD4(w_src, 0); rw1 = S3_LD4(w_src, 1);
  split8_store(ra0, ra1, lds3 + ast, lds3 + S3_PL + ast, lds3 + 2 * S3_PL + ast);
  split8_store(rw0, rw1, lds3 + wst, lds3 + S3_PL + wst, lds3 + 2 * S3_PL + wst);
  {
    const int k1 = min(1, nk - 1);
    ra0 = S3_LD4(a_src + k1 * SBK, 0); ra1 = S3_LD4(a_src + k1 * SBK, 1);
    rw0 = S3_LD4(w_src + k1 * SBK, 0); rw1 = S3_LD4(w_src + k1 * SBK, 1);
  }
  __syncthreads();

#define S3_FRAG(plane, off, f) \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) f[i] = *reinterpret_cast<const bf16x8*>(rb + (plane) * S3_PL + off[i]);
#define S3_MMA(fwp, fap)                                                                                      \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                 \
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fwp[j], fap[i], acc[i][j], 0, 0, 0);
  for (int kc = 0; kc < nk; ++kc) {
    const char* rb = lds3 + (kc & 1) * S3_BUF;
    char* sb = lds3 + ((kc & 1) ^ 1) * S3_BUF;
    const int k2 = min(kc + 2, nk - 1);                                 // the last chunks split / reload clamped data nobody reads
    const float* ap = a_src + k2 * SBK;
    const float* wp = w_src + k2 * SBK;
    bf16x8 a_hi[2], w_hi[2], a_mid[2], w_mid[2], t_lo[2];
    S3_FRAG(0, a_fo, a_hi)
    S3_FRAG(2, w_fo, t_lo)
    S3_MMA(t_lo, a_hi)                                // (hi, lo)
    S3_FRAG(0, w_fo, w_hi)
    S3_FRAG(2, a_fo, t_lo)
    S3_MMA(w_hi, t_lo)                                // (lo, hi)
    __builtin_amdgcn_sched_barrier(0);
    S3_FRAG(1, a_fo, a_mid)
    S3_FRAG(1, w_fo, w_mid)
    S3_MMA(w_mid, a_mid)                              // (mid, mid)
    S3_MMA(w_hi, a_mid)                               // (mid, hi)
    split8_store(ra0, ra1, sb + ast, sb + S3_PL + ast, sb + 2 * S3_PL + ast);
    ra0 = S3_LD4(ap, 0);
    ra1 = S3_LD4(ap, 1);
    __builtin_amdgcn_sched_barrier(0);
    S3_MMA(w_mid, a_hi)                               // (hi, mid)
    S3_MMA(w_hi, a_hi)                                // (hi, hi)
    split8_store(rw0, rw1, sb + wst, sb + S3_PL + wst, sb + 2 * S3_PL + wst);
    rw0 = S3_LD4(wp, 0);
    rw1 = S3_LD4(wp, 1);
    __syncthreads();
  }
#undef S3_LD4
#undef S3_FRAG
#undef S3_MMA
  gemm_epilogue<2, 2>(p, acc, m0, n0, wm * 32, wn * 32, fr, fq);
}

}  // namespace

bool gemm_split_supported(const GemmParams& p) {
  const bool a_ok = p.amode == AMODE_PLAIN || (p.amode == AMODE_TAPS3 && p.Kt > 0 && !(p.Kt & 31) && p.K == 3 * p.Kt && p.T > 0 && p.zeros);
  // epilogues: the straight-line bias / activation / residual one (N % 4 == 0), or the mask head's (two outputs, N even)
  const bool fast = !p.C2 && !(p.N & 3) && !(p.ldc & 3) && (!p.R || !(p.ldr & 3));
  const bool mask = p.C2 && p.X && !(p.N & 1) && !(p.ldc & 1) && !p.R;
  return a_ok && !p.lnx_c1 && !p.ln_gamma && !p.ln_stats && p.ksplit <= 1 && p.mag_F == 0 && (fast || mask) &&
         (p.drop_p <= 0.0f || (fast && p.drop_p < 1.0f && p.ldc == p.N)) &&      // training: dropout in the straight-line epilogue
         !(p.K & 31) && !(p.lda & 3) && !(p.ldw & 3) && p.alt.M <= 0 && !p.epi_general;
}

// Which kernel.  The 256 x 128 one is the efficient one per flop (190-210 TFLOP/s against 135-150 of the 64 x 64 one and 160-185 of
// the 128 x 128 one); the 64 x 64 one fills the chip from small problems on (M = 3200, N = 512, K = 2048 alone: 65 us against 137).
// A problem running ALONE on the chip wants the 256 x 128 kernel from half a round of its tiles on (128: it wins from 156 tiles
// down in profiles/r04_gemm_split_probe_3kernels.txt) -- the default.  With several forwards in flight the chip is full anyway
// and the per-flop figure decides: the visual branch's M = 3200 GEMMs (52 tiles of 256 x 128) are 1.3 % of the cfg3 step faster
// on the big tile (profiles/r04_ab_split_gemm_64x64.txt), so the inference forward asks for it from 48 tiles on
// (GemmParams::split_t2_min).  Below the threshold: the 64 x 64 kernel (plain A operand) or the 128 x 128 one (3-tap A operand).
// All three compute the same bits.
static int split_variant(const GemmParams& p) {
  const long t2 = (long)((p.M + S2BM - 1) / S2BM) * ((p.N + S2BN - 1) / S2BN);
  int v = t2 >= (p.split_t2_min > 0 ? p.split_t2_min : 128) ? 2 : 3;
#ifdef AVSEP_DEV
  if (const char* e = getenv("AVSEP_SPLIT_VARIANT")) v = atoi(e) >= 1 && atoi(e) <= 3 ? atoi(e) : v;   // developer A/B
#endif
  if (v == 2 && p.K < 2 * SBK) v = 3;
  if (v == 3 && p.amode != AMODE_PLAIN) v = 1;
  return v;
}

const char* gemm_split_instance_name(const GemmParams& p) {
  const int v = split_variant(p);
  return v == 2 ? (p.amode == AMODE_TAPS3 ? "gemm_split_kernel2<true>" : "gemm_split_kernel2<false>") : v == 3 ? "gemm_split_kernel3" : "gemm_split_kernel";
}

template <bool TAPS>
static hipError_t launch_split2(const GemmParams& p, hipStream_t s) {
  auto kern = gemm_split_kernel2<TAPS>;
  // the dynamic-LDS ceiling of the instance is raised once per device (see conv_stack.hip); the CU count is read with it
  static bool raised[64] = {};
  static int cus[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (!raised[dev]) {
    const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * S2_BUF);
    if (attr != hipSuccess) return attr;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return hipErrorInvalidDevice;
    cus[dev] = n;
    raised[dev] = true;
  }
  const long tiles = (long)((p.M + S2BM - 1) / S2BM) * ((p.N + S2BN - 1) / S2BN);
  long grid = tiles < cus[dev] ? tiles : cus[dev];                       // one resident workgroup per CU walks tiles / grid tiles
#ifdef AVSEP_DEV
  // developer A/B: 0 = one tile each; otherwise >= 8 workgroups (every XCD's tile range needs one)
  if (const char* e = getenv("AVSEP_SPLIT_GRID")) grid = atol(e) >= 8 && atol(e) < tiles ? atol(e) : tiles;
#endif
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), 2 * S2_BUF, s, p);
  return hipGetLastError();
}

hipError_t launch_gemm_split(GemmParams p, hipStream_t s) {
  if (!gemm_split_supported(p) || p.M <= 0 || p.N <= 0 || p.K <= 0) return hipErrorInvalidValue;
  p.nbn_magic = 0;
  const int v = split_variant(p);
  if (v == 2) return p.amode == AMODE_TAPS3 ? launch_split2<true>(p, s) : launch_split2<false>(p, s);
  if (v == 3) {
    const long tiles = (long)((p.M + S3B - 1) / S3B) * ((p.N + S3B - 1) / S3B);
    hipLaunchKernelGGL(gemm_split_kernel3, dim3((unsigned)tiles), dim3(256), 0, s, p);
    return hipGetLastError();
  }
  const long tiles = (long)((p.M + SBM - 1) / SBM) * ((p.N + SBN - 1) / SBN);
  hipLaunchKernelGGL(gemm_split_kernel, dim3((unsigned)tiles), dim3(256), 0, s, p);
  return hipGetLastError();
}
