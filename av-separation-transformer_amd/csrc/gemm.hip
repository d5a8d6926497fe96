// gemm.hip -- fp32 matrix-core GEMM family for gfx950 (MI355X).
//
// Every dense contraction of the forward path (>99 % of its FLOPs) runs here:
//   nn.Linear           model.py:93,155-161,195,198 and the in/out projections of MultiheadAttention
//   nn.Conv1d k3 p1     model.py:38,40   as ONE GEMM with K = 3*Cin over a 3-tap shifted A view
//   nn.Conv2d k3 s2 p1  model.py:85,88   as an implicit GEMM (im2col gather done by the staging loads)
//
// Design (CDNA4):
//   * v_mfma_f32_16x16x4_f32: exact fp32 FMA chain at 256 FLOP/clk/CU -- the only matrix path that keeps
//     masks within 1e-4 of the fp32 reference (bf16/fp16 miss it, SURVEY.md §7 "Precision").
//   * 256 threads = 4 wavefronts in a 2x2 grid; wave tile (BM/2)x(BN/2) made of 16x16 MFMA blocks.
//   * K is walked in chunks of 32 floats.  A and W chunks are staged global -> registers -> LDS with
//     128-byte-row coalesced float4 loads (issued one chunk ahead of the MFMAs that hide them) and a
//     double-buffered LDS image, one barrier per chunk.
//   * LDS image [row][32] with the 16-byte slot index XOR-swizzled by (row>>1)&7, which makes both the
//     ds_write_b128 of the staging pass and the ds_read_b128 fragment reads bank-conflict free.
//   * k-permutation trick: lane (r=l&15, q=l>>4) reads ONE float4 = k {16s+4q .. 16s+4q+3} of its row and
//     feeds component j to MFMA j; A and W use the same permutation so the contraction is unchanged and
//     every fragment read is a single ds_read_b128.
//   * Epilogue fused: bias, ReLU / erf-GELU / sigmoid, residual or positional-encoding add, and the
//     sigmoid-mask * mixed product of SeparationDecoder (model.py:207,220) with both outputs written
//     in the reference's (B,T,S,F) memory order.
#include "kernels.h"
#include <cstdio>
#include <cstdlib>

namespace {

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case ACT_RELU: return fmaxf(v, 0.0f);
    case ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));   // nn.GELU() exact form
    case ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    default: return v;
  }
}

template <int BM, int BN, int AMODE>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmParams p) {
  constexpr int D = (BM + BN >= 256) ? 2 : 4;   // prefetch ring depth (even); the biggest tile keeps VGPRs < 256
  constexpr int WBM = BM / 32;   // 16-row MFMA blocks per wave
  constexpr int WBN = BN / 32;   // 16-col MFMA blocks per wave
  constexpr int APASS = BM / 32; // staging passes of 32 rows (256 threads x float4 = 32 rows x 128 B)
  constexpr int BPASS = BN / 32;
  __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * 32];
  float* As = lds;
  float* Bs = lds + 2 * BM * 32;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nbn = (p.N + BN - 1) / BN;
  const int bm = blockIdx.x / nbn;
  const int bn = blockIdx.x - bm * nbn;
  const int m0 = bm * BM, n0 = bn * BN;

  // ---- staging coordinates: thread -> (row srow + 32*pass, 16-byte slot schunk) -------------------
  const int srow = tid >> 3;
  const int schunk = tid & 7;
  const float* a_src[APASS];
  int a_aux0[APASS], a_aux1[APASS];   // TAPS3: t index;  CONV2D: iy0, ix0
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    int m = m0 + srow + 32 * i;
    m = m < p.M ? m : p.M - 1;
    if (AMODE == AMODE_PLAIN) {
      a_src[i] = p.A + (size_t)m * p.lda + 4 * schunk;
      a_aux0[i] = a_aux1[i] = 0;
    } else if (AMODE == AMODE_TAPS3) {
      a_src[i] = p.A + (size_t)m * p.lda + 4 * schunk;
      a_aux0[i] = m % p.T;
      a_aux1[i] = 0;
    } else {
      const int hw = p.Hout * p.Wout;
      const int img = m / hw;
      const int rem = m - img * hw;
      const int y = rem / p.Wout;
      const int x = rem - y * p.Wout;
      a_src[i] = p.A + (size_t)img * p.Hin * p.Win * p.Kt + 4 * schunk;
      a_aux0[i] = 2 * y - 1;
      a_aux1[i] = 2 * x - 1;
    }
  }
  const float* b_src[BPASS];
#pragma unroll
  for (int i = 0; i < BPASS; ++i) {
    int n = n0 + srow + 32 * i;
    n = n < p.N ? n : p.N - 1;
    b_src[i] = p.W + (size_t)n * p.ldw + 4 * schunk;
  }
  // swizzled LDS float offset of this thread's staging slot (same for every pass up to +32 rows: the
  // swizzle uses (row>>1)&7 and 32 rows keep it unchanged)
  const int st_off = srow * 32 + ((schunk ^ ((srow >> 1) & 7)) << 2);

  const int nk = p.K >> 5;
  const int cpt = (AMODE == AMODE_PLAIN) ? nk : (p.Kt >> 5);   // chunks per tap
  int tap = 0, sub = 0;                                        // (tap, chunk-in-tap) of the NEXT chunk to load
  int kload = 0;                                               // index of the next chunk to load (saturates at nk-1)

  // D-deep register ring: the global loads of chunk kc+D are issued while chunk kc is being multiplied, so
  // D-1 chunks of MFMA time cover one L2 / Infinity-Cache / HBM round trip (measured ~2k cycles per chunk
  // exposed with a 1-deep prefetch at one workgroup per CU: profiles/r01_*).  fp32 MFMA leaves the VGPR file
  // nearly empty, so the ring is free.  Loads are unconditional (the chunk index saturates) to keep the
  // compiler's counted vmcnt waits exact; the ring is statically indexed after unrolling.
  f32x4 ra[D][APASS], rb[D][BPASS];
  auto load_chunk = [&](int slot) {
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      if (AMODE == AMODE_PLAIN) {
        ra[slot][i] = *reinterpret_cast<const f32x4*>(a_src[i] + (kload << 5));
      } else if (AMODE == AMODE_TAPS3) {
        const int t = a_aux0[i] + tap - 1;
        const bool ok = (t >= 0) && (t < p.T);
        const float* src = a_src[i] + (ptrdiff_t)(tap - 1) * p.lda + (sub << 5);
        ra[slot][i] = ok ? *reinterpret_cast<const f32x4*>(src) : f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
        const int ky = tap / 3, kx = tap - 3 * ky;
        const int iy = a_aux0[i] + ky, ix = a_aux1[i] + kx;
        const bool ok = (iy >= 0) && (iy < p.Hin) && (ix >= 0) && (ix < p.Win);
        const float* src = a_src[i] + ((size_t)(ok ? iy : 0) * p.Win + (ok ? ix : 0)) * p.Kt + (sub << 5);
        ra[slot][i] = ok ? *reinterpret_cast<const f32x4*>(src) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) rb[slot][i] = *reinterpret_cast<const f32x4*>(b_src[i] + (kload << 5));
    if (kload < nk - 1) {
      ++kload;
      if (AMODE != AMODE_PLAIN) {
        if (++sub == cpt) { sub = 0; ++tap; }
      }
    }
  };
  auto store_chunk = [&](int slot, int buf) {
    float* a = As + buf * BM * 32 + st_off;
    float* b = Bs + buf * BN * 32 + st_off;
#pragma unroll
    for (int i = 0; i < APASS; ++i) *reinterpret_cast<f32x4*>(a + i * 32 * 32) = ra[slot][i];
#pragma unroll
    for (int i = 0; i < BPASS; ++i) *reinterpret_cast<f32x4*>(b + i * 32 * 32) = rb[slot][i];
  };

  // ---- fragment read coordinates ----------------------------------------------------------------
  const int fr = lane & 15;   // row inside a 16-row block (A: m, W: n)
  const int fq = lane >> 4;   // k quarter
  int a_off[WBM], a_swz[WBM], b_off[WBN], b_swz[WBN];
#pragma unroll
  for (int i = 0; i < WBM; ++i) {
    const int r = wm * (BM / 2) + 16 * i + fr;
    a_off[i] = r * 32;
    a_swz[i] = (r >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < WBN; ++j) {
    const int r = wn * (BN / 2) + 16 * j + fr;
    b_off[j] = r * 32;
    b_swz[j] = (r >> 1) & 7;
  }

  f32x4 acc[WBM][WBN];
#pragma unroll
  for (int i = 0; i < WBM; ++i)
#pragma unroll
    for (int j = 0; j < WBN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int j = 0; j < D; ++j) load_chunk(j);
  store_chunk(0, 0);
  __syncthreads();

  for (int kc0 = 0; kc0 < nk; kc0 += D) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const int kc = kc0 + j;
      if (kc < nk) {                       // block-uniform
        load_chunk(j);                     // slot j (chunk kc) went to LDS one iteration ago: refill with chunk kc+D
        const float* a = As + (j & 1) * BM * 32;   // D is even: (kc & 1) == (j & 1)
        const float* b = Bs + (j & 1) * BN * 32;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          f32x4 fa[WBM], fb[WBN];
#pragma unroll
          for (int i = 0; i < WBM; ++i)
            fa[i] = *reinterpret_cast<const f32x4*>(a + a_off[i] + (((4 * s + fq) ^ a_swz[i]) << 2));
#pragma unroll
          for (int jn = 0; jn < WBN; ++jn)
            fb[jn] = *reinterpret_cast<const f32x4*>(b + b_off[jn] + (((4 * s + fq) ^ b_swz[jn]) << 2));
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < WBM; ++i)
#pragma unroll
              for (int jn = 0; jn < WBN; ++jn)
                acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][c], fb[jn][c], acc[i][jn], 0, 0, 0);
        }
        // chunk kc+1 (ring slot j+1, loaded D-1 iterations ago) -> the other LDS buffer
        if (kc + 1 < nk) store_chunk((j + 1) % D, (j + 1) & 1);
        __syncthreads();
      }
    }
  }

  // ---- epilogue: C/D layout of 16x16 MFMA: col = lane&15, row = 4*(lane>>4) + reg -----------------
#pragma unroll
  for (int j = 0; j < WBN; ++j) {
    const int n = n0 + wn * (BN / 2) + 16 * j + fr;
    if (n >= p.N) continue;
    const float bias = p.bias ? p.bias[n] : 0.0f;
    int nf = 0;
    if (p.C2) nf = n % p.F;
#pragma unroll
    for (int i = 0; i < WBM; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * (BM / 2) + 16 * i + 4 * fq + r;
        if (m >= p.M) continue;
        float v = acc[i][j][r] + bias;
        v = apply_act(v, p.act);
        if (p.R) {
          const int rr = p.rperiod > 0 ? (m % p.rperiod) : m;
          v += p.R[(size_t)rr * p.ldr + n];
        }
        p.C[(size_t)m * p.ldc + n] = v;
        if (p.C2) p.C2[(size_t)m * p.ldc + n] = v * p.X[(size_t)m * p.ldx + nf];
      }
    }
  }
}

template <int BM, int BN, int AMODE>
hipError_t launch_t(const GemmParams& p, hipStream_t s) {
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  hipLaunchKernelGGL((gemm_kernel<BM, BN, AMODE>), dim3(nbm * nbn), dim3(256), 0, s, p);
  return hipGetLastError();
}

struct Tile { int bm, bn; };

// Pick the block tile that minimises (rounds over 256 CUs) x (tile work) x (operand-reuse penalty).
// At the small M of batch-32 inference (M = B*T = 2016) the big tiles leave most CUs idle, so the
// choice matters more than the inner loop (DESIGN.md "tile selection").
Tile pick_tile(int M, int N, int amode) {
  // developer override for hardware sweeps (tools/gemm_sweep.py): AVSEP_GEMM_TILE=BMxBN
  if (const char* e = getenv("AVSEP_GEMM_TILE")) {
    Tile t{0, 0};
    if (sscanf(e, "%dx%d", &t.bm, &t.bn) == 2) return t;
  }
  // Measured on MI355X (tools/gemm_sweep.py, profiles/r01_gemm_sweep.txt): with fp32 MFMA the inner loop is
  // cheap to feed, so what decides the time at M ~ 2k is how many workgroups are resident to hide the
  // staging latency.  Take the largest tile that still gives >= 4 workgroups per CU, else >= 2, else the
  // smallest tile.
  static const Tile cands[] = {{128, 64}, {64, 64}, {64, 32}, {32, 32}};
  auto blocks = [&](const Tile& t) { return (long)((M + t.bm - 1) / t.bm) * ((N + t.bn - 1) / t.bn); };
  for (long need : {1024L, 512L})
    for (const Tile& t : cands)
      if (blocks(t) >= need) return t;
  return Tile{32, 32};
}

}  // namespace

const char* gemm_instance_name(const GemmParams& p) {
  static thread_local char buf[64];
  const Tile t = pick_tile(p.M, p.N, p.amode);
  snprintf(buf, sizeof buf, "gemm_kernel<%d, %d, %d>", t.bm, t.bn, p.amode);
  return buf;
}

hipError_t launch_gemm(const GemmParams& p, hipStream_t s) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0 || (p.K & 31)) return hipErrorInvalidValue;
  const Tile t = pick_tile(p.M, p.N, p.amode);
#define AVSEP_CASE(BM_, BN_, AM_) \
  if (t.bm == BM_ && t.bn == BN_ && p.amode == AM_) return launch_t<BM_, BN_, AM_>(p, s);
  AVSEP_CASE(128, 128, AMODE_PLAIN)
  AVSEP_CASE(128, 64, AMODE_PLAIN)
  AVSEP_CASE(64, 128, AMODE_PLAIN)
  AVSEP_CASE(64, 64, AMODE_PLAIN)
  AVSEP_CASE(64, 32, AMODE_PLAIN)
  AVSEP_CASE(32, 64, AMODE_PLAIN)
  AVSEP_CASE(32, 32, AMODE_PLAIN)
  AVSEP_CASE(128, 128, AMODE_TAPS3)
  AVSEP_CASE(128, 64, AMODE_TAPS3)
  AVSEP_CASE(64, 128, AMODE_TAPS3)
  AVSEP_CASE(64, 64, AMODE_TAPS3)
  AVSEP_CASE(64, 32, AMODE_TAPS3)
  AVSEP_CASE(32, 32, AMODE_TAPS3)
  AVSEP_CASE(128, 128, AMODE_CONV2D)
  AVSEP_CASE(128, 64, AMODE_CONV2D)
  AVSEP_CASE(64, 128, AMODE_CONV2D)
  AVSEP_CASE(64, 64, AMODE_CONV2D)
  AVSEP_CASE(64, 32, AMODE_CONV2D)
  AVSEP_CASE(32, 32, AMODE_CONV2D)
#undef AVSEP_CASE
  return hipErrorInvalidValue;
}
