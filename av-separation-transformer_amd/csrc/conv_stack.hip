// conv_stack.hip -- the visual front-end fused into one kernel:
//   3 x [Conv2d(k3,s2,p1) -> BatchNorm2d(eval) -> ReLU]  (1->32->64->128 channels)  -> AdaptiveAvgPool2d(1)
// (VisualEncoder.conv, model.py:81-92) for G lip frames per workgroup pass, with every intermediate
// activation kept in LDS: the only HBM traffic is the frames in (H*W*4 B each) and 128 floats out per frame,
// instead of writing and re-reading act1 (52 MB) / act2 (26 MB) / act3 (13 MB) per 32-clip batch.
//
//   phase 1  conv1 (Cin = 1: 9 taps, zero-extended to K = 12) as three 16x16x4 MFMA steps per 16 pixels x 16 channels,
//            the 3x3 window gathered from the raw frame in LDS; channels-last into a zero-haloed LDS image
//   phase 2  conv2 as an implicit GEMM on the fp32 matrix cores: rows = output positions of the G frames,
//            K = 9 taps x 32 ch; A fragments are ds_read_b128 gathers from the haloed image (no predication),
//            W fragments stream from L2 (each wave owns 16 of the 64 output channels, so a workgroup reads
//            the 73 KB of conv2 weights once per pass); bias+ReLU -> second haloed LDS image
//   phase 3  conv3 the same way (K = 9 x 64, each wave owns 32 of the 128 channels), then the epilogue
//            averages the valid positions of each frame in registers/shuffles and stores 128 floats.
//
// BatchNorm is folded into the conv weights/bias by the packer (avsep_api.hip).  Pixel stride in LDS is
// C+4 floats so neighbouring positions fall on different banks for the b128 gathers.
#include "kernels.h"
#include "split_terms.h"
#include <cstdlib>
#include <cstdio>
#include <vector>

namespace {

constexpr int C1 = 32, C2 = 64, C3 = 128;
constexpr int C1P = C1 + 4, C2P = C2 + 4;   // padded pixel strides (floats)

struct ConvStackParams {
  const float* frames;   // (Mv, H, W)
  const float* w1;       // [9][32]
  const float* b1;       // [32]
  const float* w2;       // [64][9][32]
  const float* b2;       // [64]
  const float* w3;       // [128][9][64]
  const float* b3;       // [128]
  float* pooled;         // (Mv, 128)
  int Mv, H, W, H1, W1, H2, W2, H3, W3;
  int a1_frame, a2_frame;   // floats per frame image incl. halo
  // floor(2^32 / d) + 1 for d = P1, W1, P2, W2, P3, W3 (0 where d == 1): pixel -> (frame, row, column) by v_mul_hi_u32
  // instead of integer divisions of ~25 VALU instructions each (fp32 MFMA and VALU do not overlap on a SIMD)
  unsigned mg_p1, mg_w1, mg_p2, mg_w2, mg_p3, mg_w3;
  unsigned long long* dbg;  // developer diagnostics (AVSEP_CONV_DBG): per-workgroup phase clock sums, null otherwise
  // Two fp16 terms, three products (conv_stack_h2_kernel; round 5): conv2 / conv3 weights as H2 planes [Co][9][2][Ci] (row n scaled by
  // 2^ew[n]), sc2 / sc3 [Co] = 2^-ew[n], and the constants the per-pass activation exponents are derived from
  const unsigned short* w2h;
  const unsigned short* w3h;
  const float* sc2;
  const float* sc3;
  float s1max, b1max;       // max_c sum_k |w1[k][c]|, max_c |b1[c]|: |conv1 out| <= max|frame| s1max + b1max
  float l2max2, b2max;      // max_n ||w2[n]||_2, max_n |b2[n]|: |conv2 out| <= sqrt(288) max|a1| l2max2 + b2max
};

// m / d for 0 <= m < 2^16, 1 <= d < 2^16 with mg = floor(2^32 / d) + 1 (exact: m * d < 2^32); mg == 0 <=> d == 1
__device__ __forceinline__ int qdiv(int m, unsigned mg) { return mg ? (int)__umulhi((unsigned)m, mg) : m; }

// AVSEP_CONV_DBG: thread 0 of every workgroup accumulates the 100 MHz wall-clock time of each phase over its passes
// into dbg[blockIdx * 8 + phase]; launch_conv_stack prints the means.  One scalar clock read per phase otherwise unused.
__device__ __forceinline__ unsigned long long cs_tick(const ConvStackParams& p) {
  return p.dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
}

// RB2 / RB3: 16-row MFMA blocks of the conv2 / conv3 output rows, COMPILE-TIME so the accumulator arrays stay in
// registers with no per-block predication (a runtime "if (i < rb)" around each MFMA made hipcc shuffle the whole
// accumulator file through v_accvgpr moves: 300k VALU instructions per wave, 10x slower).  Rows past the valid
// range read row 0's window and are simply not stored.
// NW = wavefronts per workgroup (4 or 8).  With 8, two waves share each SIMD so one wave's LDS gathers / weight
// loads / barriers hide behind the other's MFMAs: conv2 rows are split in two halves (waves 0-3 / 4-7, each wave
// still owning 16 output channels), conv3 gives every wave 16 of the 128 channels.
// The sum of one frame's post-ReLU conv3 outputs of one channel, in an order that depends on the position inside the frame only,
// never on where the frame's rows start in the pass (first0 = g * P3): a frame must pool to the same bits as the first and as the
// second frame of a pass, or a clip's output would depend on the parity of its first frame's index in the batch (odd frame counts).
// Lane-quad q, slot r holds sum class k = (4q + r - first0) mod 16 = the positions k, k+16, k+32 ... of the frame, added in that
// order by the caller; the 16 classes are then added in the order k = 0..15 (empty classes are +0, post-ReLU values are >= 0).
__device__ __forceinline__ float pool_classes(const float (&u)[4], int first0, int lane) {
  float tot = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int m16 = (k + first0) & 15, rk = m16 & 3;                    // wave-uniform
    const float src = rk == 0 ? u[0] : rk == 1 ? u[1] : rk == 2 ? u[2] : u[3];
    tot += __shfl(src, (lane & 15) + 16 * (m16 >> 2));
  }
  return tot;
}

template <int G, int RB2, int RB3MAX, int NW>
__global__ __launch_bounds__(64 * NW) void conv_stack_kernel(const ConvStackParams p) {
  constexpr int NT = 64 * NW;
  constexpr int RSPLIT = NW / 4;                       // conv2 row halves
  constexpr int RB2MAX = (RB2 + RSPLIT - 1) / RSPLIT;  // conv2 row blocks per wave
  constexpr int CB3 = 8 / NW;                          // conv3 16-channel blocks per wave
  constexpr int RAWN = 5;                              // raw-frame prefetch registers per thread (G*H*W <= RAWN*NT)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* a1 = lds;                         // G x (2*H2+1) x (2*W2+1) x C1P
  float* a2 = lds + G * p.a1_frame;        // G x (2*H3+1) x (2*W3+1) x C2P
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int c = lane & 15;
  const int q = lane >> 4;
  const int P1 = p.H1 * p.W1, P2 = p.H2 * p.W2, P3 = p.H3 * p.W3;
  // row strides (pixels) of the haloed images: a stride-2 3x3 window over Ho outputs touches halo rows 0..2*Ho,
  // so 2*Ho+1 rows/cols are enough (one halo line less than H+2 for even sizes: leaves LDS for the other stream)
  const int s1w = 2 * p.W2 + 1, s2w = 2 * p.W3 + 1;

  // zero both images once: the halo is never written again
  for (int i = tid * 4; i < G * (p.a1_frame + p.a2_frame); i += 4 * NT)
    *reinterpret_cast<f32x4*>(lds + i) = f32x4{0.f, 0.f, 0.f, 0.f};

  const int ngroups = (p.Mv + G - 1) / G;

  // per-lane LDS base offsets of the 3x3 window origin of each output row (row = 16*rb + c)
  const int cb2 = wave & 3;                 // conv2 column block of this wave
  const int rb2_0 = (wave >> 2) * RB2MAX;   // first conv2 row block of this wave
  // (and, for conv2, where the lane's output position lands in the a2 image: pass-invariant, so computed once here and
  // not in every pass's epilogue; -1 = no such position)
  int base2[RB2MAX], dst2[RB2MAX], base3[RB3MAX];
#pragma unroll
  for (int i = 0; i < RB2MAX; ++i) {
    const int mu = 16 * (rb2_0 + i) + c;
    const int m = mu < G * P2 ? mu : 0;
    const int g = qdiv(m, p.mg_p2), pos = m - g * P2;
    const int y = qdiv(pos, p.mg_w2), x = pos - y * p.W2;
    base2[i] = g * p.a1_frame + ((2 * y) * s1w + 2 * x) * C1P + 4 * q;
    dst2[i] = mu < G * P2 ? g * p.a2_frame + ((y + 1) * s2w + (x + 1)) * C2P + 16 * cb2 + 4 * q : -1;
  }
#pragma unroll
  for (int i = 0; i < RB3MAX; ++i) {
    int m = 16 * i + c;
    m = m < G * P3 ? m : 0;
    const int g = qdiv(m, p.mg_p3), pos = m - g * P3;
    const int y = qdiv(pos, p.mg_w3), x = pos - y * p.W3;
    base3[i] = g * p.a2_frame + ((2 * y) * s2w + 2 * x) * C2P + 4 * q;
  }
  // weights of this wave's output channels: conv2 col-block cb2, conv3 col-blocks CB3*wave .. CB3*wave+CB3-1
  const float* w2l = p.w2 + (size_t)(16 * cb2 + c) * 9 * C1 + 4 * q;
  const int ch3 = 16 * CB3 * wave + c;      // first conv3 channel of this lane
  const float* w3l0 = p.w3 + (size_t)ch3 * 9 * C2 + 4 * q;
  const float* w3l1 = w3l0 + (size_t)(CB3 > 1 ? 16 : 0) * 9 * C2;
  const f32x4 bias2v = *reinterpret_cast<const f32x4*>(p.b2 + 16 * cb2 + 4 * q);
  const float bias3_0 = p.b3[ch3], bias3_1 = p.b3[ch3 + (CB3 > 1 ? 16 : 0)];

  // conv1 as MFMA operands: weights of channel 16cb + c for the taps k = 4s + q (zero beyond the 9th), bias of the four
  // channels 16cb + 4q .. +3 this lane's accumulator holds
  float w1a[2][3];
  f32x4 b1c[2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
      const int k = 4 * s3 + q;
      w1a[cb][s3] = k < 9 ? p.w1[k * C1 + 16 * cb + c] : 0.0f;
    }
    b1c[cb] = *reinterpret_cast<const f32x4*>(p.b1 + 16 * cb + 4 * q);
  }

  // Raw frames are staged through LDS: the next pass's G*H*W pixels are fetched into registers (RAWN coalesced
  // loads per thread) while this pass runs on the matrix cores, so conv1 never waits on HBM.
  float* raw = lds + G * (p.a1_frame + p.a2_frame);
  const int HW = p.H * p.W, nraw = G * HW;
  float rawv[RAWN];
  auto fetch_raw = [&](int grp) {
    const float* src = p.frames + (size_t)grp * G * HW;
    const int valid = min(G, p.Mv - grp * G) * HW;
#pragma unroll
    for (int k = 0; k < RAWN; ++k) {
      const int i = tid + k * NT;
      rawv[k] = i < valid ? src[i] : 0.0f;
    }
  };
  if ((int)blockIdx.x < ngroups) fetch_raw(blockIdx.x);

  // conv1's gather is pass-invariant too: which raw pixels feed this lane's three MFMA steps of each of its 16-pixel blocks,
  // and where the block's output lands in a1.  When one sweep of U blocks per wave covers the pass (G * P1 <= 16 U NW pixels:
  // the 32 x 32 lips of configs 1-4) the offsets are computed ONCE; larger frames recompute them per pass as before.
  constexpr int U1 = 4;                                  // blocks in flight per wave
  const int npb1 = (G * P1 + 15) >> 4;                   // 16-pixel blocks of a pass
  const bool c1_once = npb1 <= U1 * NW;                  // workgroup-uniform
  int c1_src[U1][3], c1_dst[U1], c1_g[U1];              // raw offset per step (-1 = outside the frame / beyond tap 8)
  auto conv1_coords = [&](int pb, int (&src)[3], int& dst, int& g) {
    const int px = 16 * pb + c;
    const bool in = px < G * P1;                         // also false for blocks beyond npb1
    const int pxc = in ? px : 0;
    g = qdiv(pxc, p.mg_p1);
    const int pos = pxc - g * P1;
    const int y = qdiv(pos, p.mg_w1), x = pos - y * p.W1;
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
      const int k = 4 * s3 + q;                          // tap index of this lane in MFMA step s3
      const int ky = k / 3, kx = k - 3 * ky;
      const int iy = 2 * y - 1 + ky, ix = 2 * x - 1 + kx;
      const bool ok = k < 9 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      src[s3] = ok ? g * (p.H * p.W) + iy * p.W + ix : -1;
    }
    dst = in ? g * p.a1_frame + ((y + 1) * s1w + (x + 1)) * C1P + 4 * q : -1;
  };
  if (c1_once) {
#pragma unroll
    for (int u = 0; u < U1; ++u) conv1_coords(wave + u * NW, c1_src[u], c1_dst[u], c1_g[u]);
  }

  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int f0 = grp * G;
    const unsigned long long t0 = cs_tick(p);
    __syncthreads();   // previous pass done with a1/a2/raw (and the zero fill on the first pass)
#pragma unroll
    for (int k = 0; k < RAWN; ++k) {
      const int i = tid + k * NT;
      if (i < nraw) raw[i] = rawv[k];
    }
    __syncthreads();
    if (grp + (int)gridDim.x < ngroups) fetch_raw(grp + gridDim.x);
    const unsigned long long t1 = cs_tick(p);

    // ---------------- phase 1: conv1 + BN + ReLU on the matrix cores -> a1 interior ---------------------
    // Cin = 1: K = 9 taps, zero-extended to 12 = three 16x16x4 MFMA steps.  Operands swapped like the GEMM epilogue's:
    // A = weights (lane: channel 16cb + (lane&15), k = 4s + (lane>>4)), B = the 3x3 window gathered from the raw
    // frame (lane: output pixel lane&15 of a 16-pixel block, same k), C = bias, so lane (pixel, q) ends up with the
    // four consecutive channels 16cb + 4q .. +3 of its pixel: one ds_write_b128 per block into the channels-last image.
    // The MFMA is a k-ordered fma chain starting from C, i.e. bias, tap 0, tap 1, ... -- bit-identical to the VALU
    // loop it replaces (which took 7.8 of a pass's 28 us with the matrix pipes idle: in-kernel phase clocks).
    {
      constexpr int U = U1;                               // blocks in flight per wave: independent gather -> MFMA -> store chains
      for (int pb0 = wave; pb0 < npb1; pb0 += U * NW) {   // wave-uniform trip count
        float xv[U][3];
        int dst[U];
        bool pin[U], live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          int src[3], g;
          if (c1_once) {
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) src[s3] = c1_src[u][s3];
            dst[u] = c1_dst[u]; g = c1_g[u];
          } else {
            conv1_coords(pb0 + u * NW, src, dst[u], g);
          }
          pin[u] = dst[u] >= 0;
#pragma unroll
          for (int s3 = 0; s3 < 3; ++s3) xv[u][s3] = src[s3] >= 0 ? raw[src[s3] >= 0 ? src[s3] : 0] : 0.0f;
          live[u] = pin[u] && (f0 + g < p.Mv);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            f32x4 acc = b1c[cb];
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1a[cb][s3], xv[u][s3], acc, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = live[u] ? fmaxf(acc[e], 0.0f) : 0.0f;
            if (pin[u]) *reinterpret_cast<f32x4*>(a1 + dst[u] + 16 * cb) = acc;
          }
      }
    }
    __syncthreads();
    const unsigned long long t2 = cs_tick(p);

    // ---------------- phase 2: conv2 implicit GEMM, wave owns output channels [16*wave, 16*wave+16) -----
    {
      f32x4 acc[RB2MAX];
#pragma unroll
      for (int i = 0; i < RB2MAX; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      // weights are fetched one whole tap ahead (2 x b128 per lane): a tap is ~0.5 us of MFMA work, more than an L2
      // round trip, where a one-step look-ahead (0.13-0.25 us) left the matrix cores waiting on every step
      f32x4 wc[2], wn[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) wc[s] = *reinterpret_cast<const f32x4*>(w2l + s * 16);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - 3 * ky;
        const int toff = (ky * s1w + kx) * C1P;
        if (tap + 1 < 9) {
#pragma unroll
          for (int s = 0; s < 2; ++s) wn[s] = *reinterpret_cast<const f32x4*>(w2l + (tap + 1) * C1 + s * 16);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const f32x4 b = wc[s];
          f32x4 fa[RB2MAX];
#pragma unroll
          for (int i = 0; i < RB2MAX; ++i)
            fa[i] = *reinterpret_cast<const f32x4*>(a1 + base2[i] + toff + 16 * s);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < RB2MAX; ++i)
              acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[e], fa[i][e], acc[i], 0, 0, 0);   // D[channel][position]
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) wc[s] = wn[s];
      }
      // bias + ReLU -> a2 interior.  Operands are swapped (weights as the A operand), so lane (position c, q) holds the
      // four consecutive channels 16*cb2 + 4q .. +3 of its position: one coordinate computation and one ds_write_b128
      // per block instead of sixteen of each (the same products in the same order: bit-identical).
#pragma unroll
      for (int i = 0; i < RB2MAX; ++i) {
        if (dst2[i] >= 0) {
          f32x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(acc[i][r] + bias2v[r], 0.0f);
          *reinterpret_cast<f32x4*>(a2 + dst2[i]) = v;
        }
      }
    }
    __syncthreads();
    const unsigned long long t3 = cs_tick(p);

    // ---------------- phase 3: conv3 implicit GEMM + average pool, wave owns 32 output channels -------
    {
      f32x4 acc0[RB3MAX], acc1[RB3MAX];
#pragma unroll
      for (int i = 0; i < RB3MAX; ++i) acc0[i] = acc1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 wc0[4], wc1[4], wn0[4], wn1[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        wc0[s] = *reinterpret_cast<const f32x4*>(w3l0 + s * 16);
        wc1[s] = *reinterpret_cast<const f32x4*>(w3l1 + s * 16);
      }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - 3 * ky;
        const int toff = (ky * s2w + kx) * C2P;
        if (tap + 1 < 9) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            wn0[s] = *reinterpret_cast<const f32x4*>(w3l0 + (tap + 1) * C2 + s * 16);
            if constexpr (CB3 > 1) wn1[s] = *reinterpret_cast<const f32x4*>(w3l1 + (tap + 1) * C2 + s * 16);
          }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const f32x4 b0 = wc0[s], b1 = wc1[s];
          f32x4 fa[RB3MAX];
#pragma unroll
          for (int i = 0; i < RB3MAX; ++i)
            fa[i] = *reinterpret_cast<const f32x4*>(a2 + base3[i] + toff + 16 * s);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < RB3MAX; ++i) {
              acc0[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][e], b0[e], acc0[i], 0, 0, 0);
              if constexpr (CB3 > 1) acc1[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][e], b1[e], acc1[i], 0, 0, 0);
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          wc0[s] = wn0[s];
          if constexpr (CB3 > 1) wc1[s] = wn1[s];
        }
      }
      // bias + ReLU, then the mean over each frame's P3 positions (rows g*P3 .. (g+1)*P3-1)
      const float invp = 1.0f / (float)P3;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        float u0[4] = {0.f, 0.f, 0.f, 0.f}, u1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < RB3MAX; ++i) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = 16 * i + 4 * q + r;
            const bool in = (m >= g * P3) && (m < (g + 1) * P3);
            u0[r] += in ? fmaxf(acc0[i][r] + bias3_0, 0.0f) : 0.0f;
            u1[r] += in ? fmaxf(acc1[i][r] + bias3_1, 0.0f) : 0.0f;
          }
        }
        const float s0 = pool_classes(u0, g * P3, lane);
        const float s1 = CB3 > 1 ? pool_classes(u1, g * P3, lane) : 0.f;
        if (q == 0 && f0 + g < p.Mv) {
          p.pooled[(size_t)(f0 + g) * C3 + ch3] = s0 * invp;
          if constexpr (CB3 > 1) p.pooled[(size_t)(f0 + g) * C3 + ch3 + 16] = s1 * invp;
        }
      }
    }
    if (p.dbg) {
      const unsigned long long t4 = cs_tick(p);
      ph[0] += t1 - t0; ph[1] += t2 - t1; ph[2] += t3 - t2; ph[3] += t4 - t3; ph[4] += 1;
    }
  }
  if (p.dbg && tid == 0) {
#pragma unroll
    for (int i = 0; i < 5; ++i) p.dbg[(size_t)blockIdx.x * 8 + i] = ph[i];
  }
}

// RB2 / RB3: 16-row MFMA blocks of the conv2 / conv3 output rows, COMPILE-TIME so the accumulator arrays stay in
// registers with no per-block predication (a runtime "if (i < rb)" around each MFMA made hipcc shuffle the whole
// accumulator file through v_accvgpr moves: 300k VALU instructions per wave, 10x slower).  Rows past the valid
// range read row 0's window and are simply not stored.
// NW = wavefronts per workgroup (4 or 8).  With 8, two waves share each SIMD so one wave's LDS gathers / weight
// loads / barriers hide behind the other's MFMAs: conv2 rows are split in two halves (waves 0-3 / 4-7, each wave
// still owning 16 output channels), conv3 gives every wave 16 of the 128 channels.
// ---- conv2 / conv3 on the 16-bit matrix pipe (round 5) ---------------------------------------------------------------------------
// The same kernel with the two implicit GEMMs as TWO-TERM fp16 products (gemm_h2.hip's scheme: three v_mfma_f32_16x16x32_f16 per fp32
// product block instead of eight v_mfma_f32_16x16x4_f32 -- 48 matrix cycles per tap and row block instead of 256): the only
// matrix-bound kernel of the 32-clip step.  The LDS images hold hi | lo fp16 planes per pixel where they held
// fp32 -- the SAME bytes per pixel (32 channels: 64 + 64 B, 64 channels: 128 + 128 B, + 16 B pad), so frame strides, halo and the
// fragment addresses are the fp32 kernel's; a fragment is 8 channels of one term.  Scaling: the weights carry a power of two per
// output channel (packer); the activations ONE power of two per pass and image, from bounds that need one reduction only -- the
// largest |pixel| of the pass's raw frames (taken while they are staged): |conv1 out| <= max|frame| s1max + b1max,
// |conv2 out| <= sqrt(288) bound1 l2max2 + b2max (Cauchy-Schwarz over the 3 x 3 x 32 window).
template <int G, int RB2, int RB3MAX, int NW>
__global__ __launch_bounds__(64 * NW) void conv_stack_h2_kernel(const ConvStackParams p) {
  constexpr int NT = 64 * NW;
  constexpr int RSPLIT = NW / 4;                       // conv2 row halves
  constexpr int RB2MAX = (RB2 + RSPLIT - 1) / RSPLIT;  // conv2 row blocks per wave
  constexpr int CB3 = 8 / NW;                          // conv3 16-channel blocks per wave
  constexpr int RAWN = 5;                              // raw-frame prefetch registers per thread (G*H*W <= RAWN*NT)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* a1 = lds;                         // G x (2*H2+1) x (2*W2+1) x C1P
  float* a2 = lds + G * p.a1_frame;        // G x (2*H3+1) x (2*W3+1) x C2P
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int c = lane & 15;
  const int q = lane >> 4;
  const int P1 = p.H1 * p.W1, P2 = p.H2 * p.W2, P3 = p.H3 * p.W3;
  // row strides (pixels) of the haloed images: a stride-2 3x3 window over Ho outputs touches halo rows 0..2*Ho,
  // so 2*Ho+1 rows/cols are enough (one halo line less than H+2 for even sizes: leaves LDS for the other stream)
  const int s1w = 2 * p.W2 + 1, s2w = 2 * p.W3 + 1;

  // zero both images once: the halo is never written again
  for (int i = tid * 4; i < G * (p.a1_frame + p.a2_frame); i += 4 * NT)
    *reinterpret_cast<f32x4*>(lds + i) = f32x4{0.f, 0.f, 0.f, 0.f};

  const int ngroups = (p.Mv + G - 1) / G;

  // per-lane LDS base offsets of the 3x3 window origin of each output row (row = 16*rb + c)
  const int cb2 = wave & 3;                 // conv2 column block of this wave
  const int rb2_0 = (wave >> 2) * RB2MAX;   // first conv2 row block of this wave
  // (and, for conv2, where the lane's output position lands in the a2 image: pass-invariant, so computed once here and
  // not in every pass's epilogue; -1 = no such position)
  int base2[RB2MAX], dst2[RB2MAX], base3[RB3MAX];
  bool g2[RB2MAX];                                     // this lane's conv2 position belongs to the pass's second frame
#pragma unroll
  for (int i = 0; i < RB2MAX; ++i) {
    const int mu = 16 * (rb2_0 + i) + c;
    const int m = mu < G * P2 ? mu : 0;
    const int g = qdiv(m, p.mg_p2), pos = m - g * P2;
    const int y = qdiv(pos, p.mg_w2), x = pos - y * p.W2;
    base2[i] = g * p.a1_frame + ((2 * y) * s1w + 2 * x) * C1P + 4 * q;
    dst2[i] = mu < G * P2 ? g * p.a2_frame + ((y + 1) * s2w + (x + 1)) * C2P + 16 * cb2 + 4 * q : -1;
    g2[i] = g > 0;
  }
#pragma unroll
  for (int i = 0; i < RB3MAX; ++i) {
    int m = 16 * i + c;
    m = m < G * P3 ? m : 0;
    const int g = qdiv(m, p.mg_p3), pos = m - g * P3;
    const int y = qdiv(pos, p.mg_w3), x = pos - y * p.W3;
    base3[i] = g * p.a2_frame + ((2 * y) * s2w + 2 * x) * C2P + 4 * q;
  }
  // weights of this wave's output channels: conv2 col-block cb2, conv3 col-blocks CB3*wave .. CB3*wave+CB3-1
  const int ch3 = 16 * CB3 * wave + c;      // first conv3 channel of this lane
  const f32x4 bias2v = *reinterpret_cast<const f32x4*>(p.b2 + 16 * cb2 + 4 * q);
  const float bias3_0 = p.b3[ch3], bias3_1 = p.b3[ch3 + (CB3 > 1 ? 16 : 0)];

  // conv1 as MFMA operands: weights of channel 16cb + c for the taps k = 4s + q (zero beyond the 9th), bias of the four
  // channels 16cb + 4q .. +3 this lane's accumulator holds
  float w1a[2][3];
  f32x4 b1c[2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
      const int k = 4 * s3 + q;
      w1a[cb][s3] = k < 9 ? p.w1[k * C1 + 16 * cb + c] : 0.0f;
    }
    b1c[cb] = *reinterpret_cast<const f32x4*>(p.b1 + 16 * cb + 4 * q);
  }

  // Raw frames are staged through LDS: the next pass's G*H*W pixels are fetched into registers (RAWN coalesced
  // loads per thread) while this pass runs on the matrix cores, so conv1 never waits on HBM.
  float* raw = lds + G * (p.a1_frame + p.a2_frame);
  const int HW = p.H * p.W, nraw = G * HW;
  float rawv[RAWN];
  auto fetch_raw = [&](int grp) {
    const float* src = p.frames + (size_t)grp * G * HW;
    const int valid = min(G, p.Mv - grp * G) * HW;
#pragma unroll
    for (int k = 0; k < RAWN; ++k) {
      const int i = tid + k * NT;
      rawv[k] = i < valid ? src[i] : 0.0f;
    }
  };
  if ((int)blockIdx.x < ngroups) fetch_raw(blockIdx.x);

  // conv1's gather is pass-invariant too: which raw pixels feed this lane's three MFMA steps of each of its 16-pixel blocks,
  // and where the block's output lands in a1.  When one sweep of U blocks per wave covers the pass (G * P1 <= 16 U NW pixels:
  // the 32 x 32 lips of configs 1-4) the offsets are computed ONCE; larger frames recompute them per pass as before.
  constexpr int U1 = 4;                                  // blocks in flight per wave
  const int npb1 = (G * P1 + 15) >> 4;                   // 16-pixel blocks of a pass
  const bool c1_once = npb1 <= U1 * NW;                  // workgroup-uniform
  int c1_src[U1][3], c1_dst[U1], c1_g[U1];              // raw offset per step (-1 = outside the frame / beyond tap 8)
  auto conv1_coords = [&](int pb, int (&src)[3], int& dst, int& g) {
    const int px = 16 * pb + c;
    const bool in = px < G * P1;                         // also false for blocks beyond npb1
    const int pxc = in ? px : 0;
    g = qdiv(pxc, p.mg_p1);
    const int pos = pxc - g * P1;
    const int y = qdiv(pos, p.mg_w1), x = pos - y * p.W1;
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
      const int k = 4 * s3 + q;                          // tap index of this lane in MFMA step s3
      const int ky = k / 3, kx = k - 3 * ky;
      const int iy = 2 * y - 1 + ky, ix = 2 * x - 1 + kx;
      const bool ok = k < 9 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      src[s3] = ok ? g * (p.H * p.W) + iy * p.W + ix : -1;
    }
    dst = in ? g * p.a1_frame + ((y + 1) * s1w + (x + 1)) * C1P + 4 * q : -1;
  };
  if (c1_once) {
#pragma unroll
    for (int u = 0; u < U1; ++u) conv1_coords(wave + u * NW, c1_src[u], c1_dst[u], c1_g[u]);
  }

  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int f0 = grp * G;
    const unsigned long long t0 = cs_tick(p);
    __syncthreads();   // previous pass done with a1/a2/raw (and the zero fill on the first pass)
    float fmx[G];                                        // per FRAME: a frame's bits must not depend on the frame it shares a pass with
#pragma unroll
    for (int g = 0; g < G; ++g) fmx[g] = 0.0f;
#pragma unroll
    for (int k = 0; k < RAWN; ++k) {
      const int i = tid + k * NT;
      if (i < nraw) raw[i] = rawv[k];
#pragma unroll
      for (int g = 0; g < G; ++g)
        if (i >= g * HW && i < (g + 1) * HW) fmx[g] = fmaxf(fmx[g], fabsf(rawv[k]));
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) fmx[g] = fmaxf(fmx[g], __shfl_xor(fmx[g], off));
      if (lane == 0) raw[nraw + G * wave + g] = fmx[g];  // G * NW floats behind the staged pixels
    }
    __syncthreads();
    // per-frame exponents of the two images (wave-uniform arithmetic on each frame's largest |pixel|)
    float e1s[G], e2s[G], d1s[G], d2s[G];                // 2^e1, 2^e2 and their inverses
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float m = 0.0f;
#pragma unroll
      for (int w_ = 0; w_ < NW; ++w_) m = fmaxf(m, raw[nraw + G * w_ + g]);
      const float bound1 = m * p.s1max + p.b1max;
      const float bound2 = 16.9706f * bound1 * p.l2max2 + p.b2max;            // sqrt(288) rounded up
      auto expo = [](float bnd) {                                              // e with bnd 2^e <= 2^14 (0 for a zero bound)
        const int ex = (int)((__float_as_uint(bnd * 1.0001f) >> 23) & 255u) - 126;   // bnd < 2^ex
        const int e = bnd > 0.0f ? 14 - ex : 0;
        return e > 100 ? 100 : e < -100 ? -100 : e;
      };
      const int e1 = expo(bound1), e2 = expo(bound2);
      e1s[g] = ldexpf(1.0f, e1); d1s[g] = ldexpf(1.0f, -e1);
      e2s[g] = ldexpf(1.0f, e2); d2s[g] = ldexpf(1.0f, -e2);
    }
    if (grp + (int)gridDim.x < ngroups) fetch_raw(grp + gridDim.x);
    const unsigned long long t1 = cs_tick(p);

    // ---------------- phase 1: conv1 + BN + ReLU on the matrix cores -> a1 interior ---------------------
    // Cin = 1: K = 9 taps, zero-extended to 12 = three 16x16x4 MFMA steps.  Operands swapped like the GEMM epilogue's:
    // A = weights (lane: channel 16cb + (lane&15), k = 4s + (lane>>4)), B = the 3x3 window gathered from the raw
    // frame (lane: output pixel lane&15 of a 16-pixel block, same k), C = bias, so lane (pixel, q) ends up with the
    // four consecutive channels 16cb + 4q .. +3 of its pixel: one ds_write_b128 per block into the channels-last image.
    // The MFMA is a k-ordered fma chain starting from C, i.e. bias, tap 0, tap 1, ... -- bit-identical to the VALU
    // loop it replaces (which took 7.8 of a pass's 28 us with the matrix pipes idle: in-kernel phase clocks).
    {
      constexpr int U = U1;                               // blocks in flight per wave: independent gather -> MFMA -> store chains
      for (int pb0 = wave; pb0 < npb1; pb0 += U * NW) {   // wave-uniform trip count
        float xv[U][3], sc1[U];
        int dst[U];
        bool pin[U], live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          int src[3], g;
          if (c1_once) {
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) src[s3] = c1_src[u][s3];
            dst[u] = c1_dst[u]; g = c1_g[u];
          } else {
            conv1_coords(pb0 + u * NW, src, dst[u], g);
          }
          pin[u] = dst[u] >= 0;
#pragma unroll
          for (int s3 = 0; s3 < 3; ++s3) xv[u][s3] = src[s3] >= 0 ? raw[src[s3] >= 0 ? src[s3] : 0] : 0.0f;
          live[u] = pin[u] && (f0 + g < p.Mv);
          sc1[u] = G > 1 && g ? e1s[G - 1] : e1s[0];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            f32x4 acc = b1c[cb];
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1a[cb][s3], xv[u][s3], acc, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = live[u] ? fmaxf(acc[e], 0.0f) * sc1[u] : 0.0f;
            if (pin[u]) {                                // channels 16cb + 4q .. +3: 8 bytes of the hi plane, 8 of the lo plane
              unsigned h[2], l[2];
              split_pair_h2(f32x2{acc[0], acc[1]}, h[0], l[0]);
              split_pair_h2(f32x2{acc[2], acc[3]}, h[1], l[1]);
              char* px = reinterpret_cast<char*>(a1) + (size_t)(dst[u] - 4 * q) * 4 + (16 * cb + 4 * q) * 2;
              *reinterpret_cast<u32x2*>(px) = u32x2{h[0], h[1]};
              *reinterpret_cast<u32x2*>(px + 64) = u32x2{l[0], l[1]};
            }
          }
      }
    }
    __syncthreads();
    const unsigned long long t2 = cs_tick(p);

    // ---------------- phase 2: conv2 implicit GEMM on two fp16 terms, wave owns output channels [16*cb2, +16) -----
    {
      typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
      f32x4 acc[RB2MAX];
#pragma unroll
      for (int i = 0; i < RB2MAX; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      // weights one whole tap ahead: the hi and lo fragments of this lane's channel (16 bytes each)
      // fragment-order weights: (channel block cb2, tap, term) = one contiguous KiB, this lane's 16 bytes at lane * 16
      const char* w2b = reinterpret_cast<const char*>(p.w2h) + (size_t)cb2 * 9 * 2048 + lane * 16;
      f16x8 wch = *reinterpret_cast<const f16x8*>(w2b), wcl = *reinterpret_cast<const f16x8*>(w2b + 1024), wnh = wch, wnl = wcl;
      const char* a1b = reinterpret_cast<const char*>(a1);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - 3 * ky;
        const int toff = (ky * s1w + kx) * C1P * 4;
        if (tap + 1 < 9) {
          wnh = *reinterpret_cast<const f16x8*>(w2b + (tap + 1) * 2048);
          wnl = *reinterpret_cast<const f16x8*>(w2b + (tap + 1) * 2048 + 1024);
        }
        f16x8 fh[RB2MAX], fl[RB2MAX];
#pragma unroll
        for (int i = 0; i < RB2MAX; ++i) {                // the fp32 kernel's two gathers: bytes [16q, +16) of the hi and of the lo plane
          fh[i] = *reinterpret_cast<const f16x8*>(a1b + (size_t)(base2[i] - 4 * q) * 4 + toff + 16 * q);
          fl[i] = *reinterpret_cast<const f16x8*>(a1b + (size_t)(base2[i] - 4 * q) * 4 + toff + 64 + 16 * q);
        }
#pragma unroll
        for (int i = 0; i < RB2MAX; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wcl, fh[i], acc[i], 0, 0, 0);   // D[channel][position]
#pragma unroll
        for (int i = 0; i < RB2MAX; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wch, fl[i], acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < RB2MAX; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wch, fh[i], acc[i], 0, 0, 0);
        wch = wnh; wcl = wnl;
      }
      // true scale, bias + ReLU, then the a2 image's scale and the two fp16 terms: lane (position c, q) holds channels 16 cb2 + 4q .. +3
      const f32x4 s2v = *reinterpret_cast<const f32x4*>(p.sc2 + 16 * cb2 + 4 * q);
#pragma unroll
      for (int i = 0; i < RB2MAX; ++i) {
        if (dst2[i] >= 0) {
          f32x4 v;
          const float din = G > 1 && g2[i] ? d1s[G - 1] : d1s[0], eout = G > 1 && g2[i] ? e2s[G - 1] : e2s[0];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(acc[i][r] * din * s2v[r] + bias2v[r], 0.0f) * eout;
          unsigned h[2], l[2];
          split_pair_h2(f32x2{v[0], v[1]}, h[0], l[0]);
          split_pair_h2(f32x2{v[2], v[3]}, h[1], l[1]);
          char* px = reinterpret_cast<char*>(a2) + (size_t)(dst2[i] - 16 * cb2 - 4 * q) * 4 + (16 * cb2 + 4 * q) * 2;
          *reinterpret_cast<u32x2*>(px) = u32x2{h[0], h[1]};
          *reinterpret_cast<u32x2*>(px + 128) = u32x2{l[0], l[1]};
        }
      }
    }
    __syncthreads();
    const unsigned long long t3 = cs_tick(p);

    // ---------------- phase 3: conv3 implicit GEMM on two fp16 terms + average pool ------------------------------
    {
      typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
      f32x4 acc0[RB3MAX], acc1[RB3MAX];
#pragma unroll
      for (int i = 0; i < RB3MAX; ++i) acc0[i] = acc1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      // fragment-order weights: (channel block, tap) = 4 KiB = [term][k-step][64 lanes][16 bytes]
      const char* w3b0 = reinterpret_cast<const char*>(p.w3h) + (size_t)(CB3 * wave) * 9 * 4096 + lane * 16;
      const char* w3b1 = w3b0 + (size_t)(CB3 > 1 ? 1 : 0) * 9 * 4096;
      f16x8 wc0[4], wc1[4], wn0[4], wn1[4];                // [k-step 0 hi, k-step 1 hi, k-step 0 lo, k-step 1 lo]
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        wc0[s] = *reinterpret_cast<const f16x8*>(w3b0 + (s & 1) * 1024 + (s >> 1) * 2048);
        wc1[s] = *reinterpret_cast<const f16x8*>(w3b1 + (s & 1) * 1024 + (s >> 1) * 2048);
        wn0[s] = wc0[s]; wn1[s] = wc1[s];
      }
      const char* a2b = reinterpret_cast<const char*>(a2);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - 3 * ky;
        const int toff = (ky * s2w + kx) * C2P * 4;
        if (tap + 1 < 9) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            wn0[s] = *reinterpret_cast<const f16x8*>(w3b0 + (tap + 1) * 4096 + (s & 1) * 1024 + (s >> 1) * 2048);
            if constexpr (CB3 > 1) wn1[s] = *reinterpret_cast<const f16x8*>(w3b1 + (tap + 1) * 4096 + (s & 1) * 1024 + (s >> 1) * 2048);
          }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          f16x8 fh[RB3MAX], fl[RB3MAX];
#pragma unroll
          for (int i = 0; i < RB3MAX; ++i) {
            fh[i] = *reinterpret_cast<const f16x8*>(a2b + (size_t)(base3[i] - 4 * q) * 4 + toff + 64 * ks + 16 * q);
            fl[i] = *reinterpret_cast<const f16x8*>(a2b + (size_t)(base3[i] - 4 * q) * 4 + toff + 128 + 64 * ks + 16 * q);
          }
#pragma unroll
          for (int i = 0; i < RB3MAX; ++i) {                // D[position][channel]: (hi, lo), (lo, hi), (hi, hi)
            acc0[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[i], wc0[2 + ks], acc0[i], 0, 0, 0);
            if constexpr (CB3 > 1) acc1[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[i], wc1[2 + ks], acc1[i], 0, 0, 0);
          }
#pragma unroll
          for (int i = 0; i < RB3MAX; ++i) {
            acc0[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[i], wc0[ks], acc0[i], 0, 0, 0);
            if constexpr (CB3 > 1) acc1[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[i], wc1[ks], acc1[i], 0, 0, 0);
          }
#pragma unroll
          for (int i = 0; i < RB3MAX; ++i) {
            acc0[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[i], wc0[ks], acc0[i], 0, 0, 0);
            if constexpr (CB3 > 1) acc1[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[i], wc1[ks], acc1[i], 0, 0, 0);
          }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          wc0[s] = wn0[s];
          if constexpr (CB3 > 1) wc1[s] = wn1[s];
        }
      }
      const float sc3_0 = p.sc3[ch3], sc3_1 = p.sc3[ch3 + (CB3 > 1 ? 16 : 0)];   // the channel's 2^-ew; the frame's 2^-e2 below
      // bias + ReLU, then the mean over each frame's P3 positions (rows g*P3 .. (g+1)*P3-1)
      const float invp = 1.0f / (float)P3;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        float u0[4] = {0.f, 0.f, 0.f, 0.f}, u1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < RB3MAX; ++i) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = 16 * i + 4 * q + r;
            const bool in = (m >= g * P3) && (m < (g + 1) * P3);
            u0[r] += in ? fmaxf(acc0[i][r] * d2s[g] * sc3_0 + bias3_0, 0.0f) : 0.0f;
            u1[r] += in ? fmaxf(acc1[i][r] * d2s[g] * sc3_1 + bias3_1, 0.0f) : 0.0f;
          }
        }
        const float s0 = pool_classes(u0, g * P3, lane);
        const float s1 = CB3 > 1 ? pool_classes(u1, g * P3, lane) : 0.f;
        if (q == 0 && f0 + g < p.Mv) {
          p.pooled[(size_t)(f0 + g) * C3 + ch3] = s0 * invp;
          if constexpr (CB3 > 1) p.pooled[(size_t)(f0 + g) * C3 + ch3 + 16] = s1 * invp;
        }
      }
    }
    if (p.dbg) {
      const unsigned long long t4 = cs_tick(p);
      ph[0] += t1 - t0; ph[1] += t2 - t1; ph[2] += t3 - t2; ph[3] += t4 - t3; ph[4] += 1;
    }
  }
  if (p.dbg && tid == 0) {
#pragma unroll
    for (int i = 0; i < 5; ++i) p.dbg[(size_t)blockIdx.x * 8 + i] = ph[i];
  }
}

inline int conv_out(int x) { return (x - 1) / 2 + 1; }

template <int G, int RB2, int RB3, int NW, bool H2 = false>
hipError_t launch_cs_nw(const ConvStackParams& p, size_t lds_bytes, hipStream_t s) {
  if (!H2 && p.w2h) return launch_cs_nw<G, RB2, RB3, NW, true>(p, lds_bytes + 64, s);   // + the per-wave, per-frame |pixel| maxima behind the raw frames (G * NW floats)
  auto kern = H2 ? conv_stack_h2_kernel<G, RB2, RB3, NW> : conv_stack_kernel<G, RB2, RB3, NW>;
  // the instance's dynamic-LDS ceiling is raised (to the hardware's 160 KiB) once PER DEVICE, not on every launch: the
  // library keeps contexts on several devices in one process (avsep_ctx::device), and a failure is not latched
  static bool raised[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
  if (dev < 0 || !raised[dev]) {
    const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr != hipSuccess) return attr;
    if (dev >= 0) raised[dev] = true;
  }
  const int ngroups = (p.Mv + G - 1) / G;
  // One workgroup per CU: the audio branch runs concurrently on the other stream and its GEMMs need LDS too
  // (measured: with the CUs' LDS full of conv images the two branches serialise, profiles/r01c_step_timeline.txt).
  int per_cu = 1;
  if (const char* e = dev_env("AVSEP_CONV_WGPC")) per_cu = atoi(e) > 0 ? atoi(e) : 1;   // developer A/B switch
  int grid = 256 * per_cu;
  if (const char* e = dev_env("AVSEP_CONV_GRID")) grid = atoi(e) > 0 ? atoi(e) : grid;   // developer A/B switch
  if (grid > ngroups) grid = ngroups;
  static const bool dbg = dev_env("AVSEP_CONV_DBG") != nullptr;
  if (!dbg) {
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds_bytes, s, p);
    return hipGetLastError();
  }
  static unsigned long long* buf = nullptr;
  if (!buf && hipMalloc(reinterpret_cast<void**>(&buf), 8192 * 8 * sizeof(unsigned long long)) != hipSuccess) return hipErrorOutOfMemory;
  ConvStackParams q = p;
  q.dbg = buf;
  (void)hipMemsetAsync(buf, 0, 8192 * 8 * sizeof(unsigned long long), s);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds_bytes, s, q);
  (void)hipStreamSynchronize(s);
  static int shown = 0;
  if (shown++ < 3) {
    std::vector<unsigned long long> h((size_t)grid * 8);
    (void)hipMemcpy(h.data(), buf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double sum[4] = {0, 0, 0, 0}, passes = 0;
    for (int b = 0; b < grid; ++b) { for (int i = 0; i < 4; ++i) sum[i] += (double)h[b * 8 + i]; passes += (double)h[b * 8 + 4]; }
    fprintf(stderr, "[conv dbg] G=%d NW=%d grid %d, %.0f passes: mean per pass: stage raw %.2f  conv1 %.2f  conv2 %.2f  conv3+pool %.2f us\n",
            G, NW, grid, passes, sum[0] / passes / 100, sum[1] / passes / 100, sum[2] / passes / 100, sum[3] / passes / 100);
  }
  return hipGetLastError();
}

template <int G, int RB2, int RB3>
hipError_t launch_cs(const ConvStackParams& p, size_t lds_bytes, hipStream_t s) {
  static const bool four = dev_env("AVSEP_CONV_NW4") != nullptr;   // developer A/B switch
  if (four) return launch_cs_nw<G, RB2, RB3, 4>(p, lds_bytes, s);
  return launch_cs_nw<G, RB2, RB3, 8>(p, lds_bytes, s);
}

}  // namespace

// conv weights [Co][9][Ci] fp32 (BN folded, tap-major) -> H2 terms in FRAGMENT order, row n scaled by 2^ew[n]; sc[n] = 2^-ew[n].
// A wave's weight fragment (16 channels x 32 input channels of one tap and term: lane c + 16 q holds channel c's inputs 8q .. 8q+7) is
// one contiguous KiB -- [Co/16][9][2 terms][Ci/32 k-steps][64 lanes][8] fp16 -- so every load instruction of the kernel reads whole
// cache lines (with [Co][9][2][Ci] a lane group fetched 64 bytes per channel row: half of every line, 33 GB/s per CU where the L2
// serves 66-73).
namespace {
__global__ __launch_bounds__(256) void pack_conv_h2_kernel(const float* __restrict__ w, const int* __restrict__ ew,
                                                           unsigned short* __restrict__ wh, float* __restrict__ sc, int Co, int Ci) {
  const int idx = blockIdx.x * 256 + threadIdx.x;                      // one thread per (channel, tap, pair of input channels)
  const int half = Ci >> 1;
  if (idx >= Co * 9 * half) return;
  const int k2 = idx % half, tap = (idx / half) % 9, n = idx / (9 * half);
  const float scl = ldexpf(1.0f, ew[n]);
  const float* src = w + ((size_t)n * 9 + tap) * Ci + 2 * k2;
  unsigned h, l;
  split_pair_h2(f32x2{src[0] * scl, src[1] * scl}, h, l);
  const int k = 2 * k2, ks = k >> 5, q = (k & 31) >> 3, e2 = (k & 7) >> 1;        // k-step, lane group, dword inside the lane's 16 bytes
  const int nks = Ci >> 5;
  // dword index: ((((n / 16) * 9 + tap) * 2 + term) * nks + ks) * 64 lanes * 4 dwords + (16 q + n % 16) * 4 + e2
  const size_t f0 = ((((size_t)(n >> 4) * 9 + tap) * 2 + 0) * nks + ks) * 256 + ((16 * q + (n & 15)) << 2) + e2;
  unsigned* dst = reinterpret_cast<unsigned*>(wh);
  dst[f0] = h;
  dst[f0 + (size_t)nks * 256] = l;
  if (tap == 0 && k2 == 0) sc[n] = ldexpf(1.0f, -ew[n]);
}
}  // namespace

hipError_t launch_pack_conv_h2(const float* w, const int* ew, unsigned short* wh, float* sc, int Co, int Ci, hipStream_t s) {
  if (!w || !ew || !wh || !sc || Co <= 0 || Ci <= 0 || (Ci & 1)) return hipErrorInvalidValue;
  const int n = Co * 9 * (Ci >> 1);
  hipLaunchKernelGGL(pack_conv_h2_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w, ew, wh, sc, Co, Ci);
  return hipGetLastError();
}

// The instance launch_conv_stack() picks, spelled as rocprofv3 prints it ("conv_stack_kernel" when the frame does not fit).
const char* conv_stack_instance_name(int Mv, int H, int W, bool h2) {
  const bool g_conv_h2_name = h2 && dev_env("AVSEP_CONV_FP32") == nullptr;
  static thread_local char buf[48];
  const int H1 = conv_out(H), W1 = conv_out(W), H2 = conv_out(H1), W2 = conv_out(W1), H3 = conv_out(H2), W3 = conv_out(W2);
  const int P2 = H2 * W2, P3 = H3 * W3;
  const size_t frame_bytes = (size_t)((2 * H2 + 1) * (2 * W2 + 1) * C1P + (2 * H3 + 1) * (2 * W3 + 1) * C2P + H * W) * sizeof(float);
  const bool four = dev_env("AVSEP_CONV_NW4") != nullptr;
  const size_t raw_cap = (size_t)5 * (four ? 256 : 512), LDS_MAX = 160 * 1024;
  const int r2g2 = (2 * P2 + 15) / 16, r3g2 = (2 * P3 + 15) / 16, r2 = (P2 + 15) / 16, r3 = (P3 + 15) / 16;
  int g = 0, a = 0, b = 0;
  if (!dev_env("AVSEP_CONV_G1") && 2 * frame_bytes <= LDS_MAX && Mv > 1 && (size_t)2 * H * W <= raw_cap) {
    if (r2g2 <= 1 && r3g2 <= 1) { g = 2; a = 1; b = 1; }
    else if (r2g2 <= 2 && r3g2 <= 1) { g = 2; a = 2; b = 1; }
    else if (r2g2 <= 4 && r3g2 <= 1) { g = 2; a = 4; b = 1; }
    else if (r2g2 <= 8 && r3g2 <= 2) { g = 2; a = 8; b = 2; }
  }
  if (!g && frame_bytes <= LDS_MAX && (size_t)H * W <= raw_cap) {
    if (r2 <= 1 && r3 <= 1) { g = 1; a = 1; b = 1; }
    else if (r2 <= 4 && r3 <= 1) { g = 1; a = 4; b = 1; }
    else if (r2 <= 9 && r3 <= 3) { g = 1; a = 9; b = 3; }
    else if (r2 <= 12 && r3 <= 4) { g = 1; a = 12; b = 4; }
  }
  if (!g) return "conv_stack_kernel";
  snprintf(buf, sizeof buf, "conv_stack_%skernel<%d, %d, %d, %d>", g_conv_h2_name ? "h2_" : "", g, a, b, four ? 4 : 8);
  return buf;
}

// Returns hipErrorNotSupported when the frame size does not fit the fused kernel's LDS / register tiling
// (the caller then takes the unfused conv1 + implicit-GEMM path).
hipError_t launch_conv_stack(const float* frames, const float* w1, const float* b1, const float* w2,
                             const float* b2, const float* w3, const float* b3, float* pooled, int Mv, int H,
                             int W, hipStream_t s, const ConvH2* h2) {
  if (Mv <= 0 || H <= 0 || W <= 0) return hipErrorInvalidValue;
  ConvStackParams p{};
  p.frames = frames; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.w3 = w3; p.b3 = b3; p.pooled = pooled;
  static const bool no_h2 = dev_env("AVSEP_CONV_FP32") != nullptr;   // developer A/B
  if (h2 && h2->w2h && !no_h2) {
    p.w2h = h2->w2h; p.w3h = h2->w3h; p.sc2 = h2->sc2; p.sc3 = h2->sc3;
    p.s1max = h2->s1max; p.b1max = h2->b1max; p.l2max2 = h2->l2max2; p.b2max = h2->b2max;
  }
  p.Mv = Mv; p.H = H; p.W = W;
  p.H1 = conv_out(H); p.W1 = conv_out(W);
  p.H2 = conv_out(p.H1); p.W2 = conv_out(p.W1);
  p.H3 = conv_out(p.H2); p.W3 = conv_out(p.W2);
  p.a1_frame = (2 * p.H2 + 1) * (2 * p.W2 + 1) * C1P;
  p.a2_frame = (2 * p.H3 + 1) * (2 * p.W3 + 1) * C2P;
  const int P2 = p.H2 * p.W2, P3 = p.H3 * p.W3;
  auto magic = [](int d) { return d > 1 ? (unsigned)((1ULL << 32) / (unsigned)d) + 1u : 0u; };
  if (H * W >= 65536) return hipErrorNotSupported;   // (qdiv's range; far beyond what fits LDS anyway)
  p.mg_p1 = magic(p.H1 * p.W1); p.mg_w1 = magic(p.W1); p.mg_p2 = magic(P2); p.mg_w2 = magic(p.W2);
  p.mg_p3 = magic(P3); p.mg_w3 = magic(p.W3);
  // per frame: the two haloed images + the raw pixels staged for conv1
  const size_t frame_bytes = (size_t)(p.a1_frame + p.a2_frame + H * W) * sizeof(float);
  static const bool four_waves = dev_env("AVSEP_CONV_NW4") != nullptr;
  const size_t raw_cap = (size_t)5 * (four_waves ? 256 : 512);            // RAWN * NT prefetch registers per pass
  const size_t LDS_MAX = 160 * 1024;
  static const bool force_g1 = dev_env("AVSEP_CONV_G1") != nullptr;   // developer A/B switch
  // Instantiations (G frames per pass, conv2 row blocks, conv3 row blocks); the smallest one that covers the
  // frame size is used -- surplus row blocks recompute row 0 and are discarded.  Two frames per pass halve the
  // weight traffic per frame.
  const int r2g2 = (2 * P2 + 15) / 16, r3g2 = (2 * P3 + 15) / 16, r2 = (P2 + 15) / 16, r3 = (P3 + 15) / 16;
  if (!force_g1 && 2 * frame_bytes <= LDS_MAX && Mv > 1 && (size_t)2 * H * W <= raw_cap) {
    if (r2g2 <= 1 && r3g2 <= 1) return launch_cs<2, 1, 1>(p, 2 * frame_bytes, s);
    if (r2g2 <= 2 && r3g2 <= 1) return launch_cs<2, 2, 1>(p, 2 * frame_bytes, s);
    if (r2g2 <= 4 && r3g2 <= 1) return launch_cs<2, 4, 1>(p, 2 * frame_bytes, s);
    if (r2g2 <= 8 && r3g2 <= 2) return launch_cs<2, 8, 2>(p, 2 * frame_bytes, s);
  }
  if (frame_bytes > LDS_MAX || (size_t)H * W > raw_cap) return hipErrorNotSupported;
  if (r2 <= 1 && r3 <= 1) return launch_cs<1, 1, 1>(p, frame_bytes, s);
  if (r2 <= 4 && r3 <= 1) return launch_cs<1, 4, 1>(p, frame_bytes, s);
  if (r2 <= 9 && r3 <= 3) return launch_cs<1, 9, 3>(p, frame_bytes, s);
  if (r2 <= 12 && r3 <= 4) return launch_cs<1, 12, 4>(p, frame_bytes, s);
  return hipErrorNotSupported;
}
