// rowops.hip -- the HBM-bound kernels of the forward path (wavefront reductions and layout passes) and
// the one-off weight packers.  None of these has arithmetic intensity worth the matrix cores; they are
// written for coalesced 16-byte accesses along the contiguous axis and 64-lane wavefront reductions.
#include "kernels.h"
#include "split_terms.h"
#include <cstdio>

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// nn.LayerNorm (model.py:143,162-163 and the norm1/norm2 of TransformerEncoderLayer): one wavefront per
// row, the row held in registers (VEC float4 per lane), two-pass mean / biased variance like ATen.
// PLANES (round 5): the normalised row is written as the three bf16 terms of the consuming GEMM's A operand (plane format of
// gemm_planes.hip, `rows` = the buffer's row count) instead of fp32 -- the same values, cut by split_terms.h's split: 6 bytes per
// element instead of 4, and the GEMM no longer splits them in every column tile.  d % 32 == 0.
// PLANES = 2: as the two fp16 terms of gemm_h2.hip, scaled by `pscale` = 2^e, the static exponent of this LayerNorm site
// (sqrt(d - 1) |gamma| + |beta| <= 2^(14 - e): no overflow whatever the input).
template <int VEC, bool STATS_ONLY = false, int PLANES = 0>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gam,
                                                        const float* __restrict__ bet, float* __restrict__ y,
                                                        int M, int d, float eps, long long rows = 0, float pscale = 1.0f) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (size_t)row * d;
  f32x4 v[VEC];
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const int col = (lane + 64 * i) * 4;
    v[i] = col < d ? *reinterpret_cast<const f32x4*>(xr + col) : f32x4{0.f, 0.f, 0.f, 0.f};
    s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
  const float mean = wave_sum(s) / (float)d;
  float sq = 0.0f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const int col = (lane + 64 * i) * 4;
    if (col < d) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float c = v[i][e] - mean;
        sq += c * c;
      }
    }
  }
  const float var = wave_sum(sq) / (float)d;
  const float rstd = 1.0f / sqrtf(var + eps);
  if (STATS_ONLY) {   // y = (M, 2): the LayerNorm is applied by the GEMM that consumes x (gemm.hip AMODE_LN)
    if (lane == 0) *reinterpret_cast<float2*>(y + 2 * (size_t)row) = make_float2(mean, rstd);
    return;
  }
  float* yr = y + (size_t)row * d;
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const int col = (lane + 64 * i) * 4;
    if (col < d) {
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(gam + col);
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(bet + col);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g4[e] + b4[e];
      if (PLANES == 2) {
        unsigned h[2], l[2];
        split_pair_h2(f32x2{o[0], o[1]} * pscale, h[0], l[0]);
        split_pair_h2(f32x2{o[2], o[3]} * pscale, h[1], l[1]);
        char* dst = reinterpret_cast<char*>(y) + (((size_t)(col >> 5) * 2) * rows + row) * 64 + (col & 31) * 2;
        *reinterpret_cast<u32x2*>(dst) = u32x2{h[0], h[1]};
        *reinterpret_cast<u32x2*>(dst + (size_t)rows * 64) = u32x2{l[0], l[1]};
      } else if (PLANES == 1) {                    // 4 consecutive columns = 8 bytes of each term; 8 lanes fill a chunk's 64-byte line
        unsigned h[2], m[2], l[2];
        split_pair(f32x2{o[0], o[1]}, h[0], m[0], l[0]);
        split_pair(f32x2{o[2], o[3]}, h[1], m[1], l[1]);
        char* dst = reinterpret_cast<char*>(y) + (((size_t)(col >> 5) * 3) * rows + row) * 64 + (col & 31) * 2;
        const size_t ts = (size_t)rows * 64;
        *reinterpret_cast<u32x2*>(dst) = u32x2{h[0], h[1]};
        *reinterpret_cast<u32x2*>(dst + ts) = u32x2{m[0], m[1]};
        *reinterpret_cast<u32x2*>(dst + 2 * ts) = u32x2{l[0], l[1]};
      } else {
        *reinterpret_cast<f32x4*>(yr + col) = o;
      }
    }
  }
}

// mixed_spec (B,F,T) -> (B,T,Fp): makes the frequency axis contiguous so Conv1d becomes a K-contiguous
// GEMM and the mask epilogue reads mixed along n; 32x32 LDS tile transpose, both sides coalesced.
__global__ __launch_bounds__(256) void transpose_pad_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            int F, int T, int Fp) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int f0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const float* xb = x + (size_t)b * F * T;
  float* yb = y + (size_t)b * T * Fp;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = f0 + ty + 8 * i, t = t0 + tx;
    tile[ty + 8 * i][tx] = (f < F && t < T) ? xb[(size_t)f * T + t] : 0.0f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = t0 + ty + 8 * i, f = f0 + tx;
    if (t < T && f < Fp) yb[(size_t)t * Fp + f] = tile[tx][ty + 8 * i];
  }
}

// First visual conv: Conv2d(1->32,k3,s2,p1)+BatchNorm2d(eval)+ReLU (model.py:82-84) with BN folded into
// w/b by the packer.  Cin = 1 means 9 MACs per output: VALU work, output written channels-last
// (M,Ho,Wo,32) so the next conv's implicit-GEMM gather reads 128-byte channel rows.
__global__ __launch_bounds__(256) void conv1_c1_kernel(const float* __restrict__ frames, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ out,
                                                       int total_px, int H, int W, int Ho, int Wo) {
  const int cq = threadIdx.x & 7;             // channel quad: channels 4cq..4cq+3
  const int px = blockIdx.x * 32 + (threadIdx.x >> 3);
  if (px >= total_px) return;
  const int hw = Ho * Wo;
  const int img = px / hw;
  const int rem = px - img * hw;
  const int y = rem / Wo, x = rem - y * Wo;
  const float* f = frames + (size_t)img * H * W;
  f32x4 a = *reinterpret_cast<const f32x4*>(bias + 4 * cq);
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = 2 * y - 1 + ky;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = 2 * x - 1 + kx;
      const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
      const float p = ok ? f[iy * W + ix] : 0.0f;
      const f32x4 w4 = *reinterpret_cast<const f32x4*>(w + (ky * 3 + kx) * 32 + 4 * cq);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] = fmaf(p, w4[e], a[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) a[e] = fmaxf(a[e], 0.0f);
  *reinterpret_cast<f32x4*>(out + (size_t)px * 32 + 4 * cq) = a;
}

// AdaptiveAvgPool2d(1) (model.py:91): x (M,P,C) channels-last -> y (M,C)
__global__ __launch_bounds__(256) void avgpool_kernel(const float* __restrict__ x, float* __restrict__ y, int M,
                                                      int P, int C) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int cq = C >> 2;
  if (idx >= M * cq) return;
  const int m = idx / cq, c4 = idx - m * cq;
  const float* src = x + (size_t)m * P * C + 4 * c4;
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int p = 0; p < P; ++p) s += *reinterpret_cast<const f32x4*>(src + (size_t)p * C);
  const float inv = 1.0f / (float)P;
  *reinterpret_cast<f32x4*>(y + (size_t)m * C + 4 * c4) = s * inv;
}

// F.interpolate(mode='linear', align_corners=False) along time (model.py:114-116); index arithmetic in
// fp32 exactly as ATen's area_pixel_compute_source_index.
template <bool PLANES>
__global__ __launch_bounds__(256) void interp_linear_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            int B, int N, int T, int d, float scale, long long rows) {
  const int dq = d >> 2;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)B * T * dq) return;
  const int c4 = (int)(idx % dq);
  const size_t bt = idx / dq;
  const int t = (int)(bt % T);
  const int b = (int)(bt / T);
  // scale = (float)N / (float)T comes from the host (one IEEE rounding).  ATen's CPU kernel evaluates
  // scale*(i+0.5)-0.5 as ONE fused multiply-add (measured: tests/test_oracle.py::test_interp_index_is_fma),
  // and a 1-ulp difference in src moves the lerp weight by ~4e-6, so the fma is part of the contract.
  float src = fmaf(scale, (float)t + 0.5f, -0.5f);
  src = src < 0.0f ? 0.0f : src;
  int i0 = (int)src;
  i0 = i0 < N - 1 ? i0 : N - 1;
  const int i1 = i0 + 1 < N ? i0 + 1 : N - 1;
  const float w1 = src - (float)i0;
  const float w0 = 1.0f - w1;
  const f32x4 a = *reinterpret_cast<const f32x4*>(x + ((size_t)b * N + i0) * d + 4 * c4);
  const f32x4 c = *reinterpret_cast<const f32x4*>(x + ((size_t)b * N + i1) * d + 4 * c4);
  const f32x4 o = w0 * a + w1 * c;
  if (PLANES) {                                    // see layernorm_kernel
    unsigned h[2], m[2], l[2];
    split_pair(f32x2{o[0], o[1]}, h[0], m[0], l[0]);
    split_pair(f32x2{o[2], o[3]}, h[1], m[1], l[1]);
    const int col = 4 * c4;
    char* dst = reinterpret_cast<char*>(y) + (((size_t)(col >> 5) * 3) * rows + bt) * 64 + (col & 31) * 2;
    const size_t ts = (size_t)rows * 64;
    *reinterpret_cast<u32x2*>(dst) = u32x2{h[0], h[1]};
    *reinterpret_cast<u32x2*>(dst + ts) = u32x2{m[0], m[1]};
    *reinterpret_cast<u32x2*>(dst + 2 * ts) = u32x2{l[0], l[1]};
  } else {
    *reinterpret_cast<f32x4*>(y + bt * d + 4 * c4) = o;
  }
}

// The same resize as the two fp16 terms of the fusion K/V projection's operand (gemm_h2.hip), one WAVE per output row: the visual
// stream has no static bound, so each row carries its own power of two -- from the row's largest magnitude, taken in registers
// (a row of a clip alone has the exponent it has inside any batch) -- and rscale[row] = 2^-e brings the GEMM's accumulators back.
template <int VEC>
__global__ __launch_bounds__(256) void interp_linear_h2_kernel(const float* __restrict__ x, unsigned short* __restrict__ yp,
                                                               float* __restrict__ rscale, long long rows, int B, int N, int T, int d,
                                                               float scale) {
  const int lane = threadIdx.x & 63;
  const long long bt = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bt >= (long long)B * T) return;
  const int t = (int)(bt % T), b = (int)(bt / T);
  float src = fmaf(scale, (float)t + 0.5f, -0.5f);     // interp_linear_kernel's index arithmetic
  src = src < 0.0f ? 0.0f : src;
  int i0 = (int)src;
  i0 = i0 < N - 1 ? i0 : N - 1;
  const int i1 = i0 + 1 < N ? i0 + 1 : N - 1;
  const float w1 = src - (float)i0;
  const float w0 = 1.0f - w1;
  const float* r0 = x + ((size_t)b * N + i0) * d;
  const float* r1 = x + ((size_t)b * N + i1) * d;
  f32x4 o[VEC][2];
  float mx = 0.0f;
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    const int col = (lane + 64 * v) * 8;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (col < d) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(r0 + col + 4 * h), c = *reinterpret_cast<const f32x4*>(r1 + col + 4 * h);
        o[v][h] = w0 * a + w1 * c;
      } else {
        o[v][h] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) mx = fmaxf(mx, fabsf(o[v][h][e]));
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  // e with mx 2^e in [2^13, 2^14) (0 for a zero row; clamped): the weight rows' rule (gemm_h2.hip h2_row_stats_kernel)
  int e = mx == 0.0f ? 100 : 0;                                            // a zero row: the smallest bound (it never sets a clip's bound, below)
  if (mx > 0.0f && mx < 3.0e38f) {
    const int ex = (int)((__float_as_uint(mx) >> 23) & 255u) - 126;        // mx < 2^ex
    e = 14 - ex;
    e = e > 100 ? 100 : e < -100 ? -100 : e;
  }
  const float sc = ldexpf(1.0f, e);
  if (lane == 0) rscale[bt] = ldexpf(1.0f, -e);
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    const int col = (lane + 64 * v) * 8;
    if (col < d) {
      unsigned h[4], l[4];
      split_pair_h2(f32x2{o[v][0][0], o[v][0][1]} * sc, h[0], l[0]);
      split_pair_h2(f32x2{o[v][0][2], o[v][0][3]} * sc, h[1], l[1]);
      split_pair_h2(f32x2{o[v][1][0], o[v][1][1]} * sc, h[2], l[2]);
      split_pair_h2(f32x2{o[v][1][2], o[v][1][3]} * sc, h[3], l[3]);
      char* dst = reinterpret_cast<char*>(yp) + (((size_t)(col >> 5) * 2) * rows + bt) * 64 + (col & 31) * 2;
      *reinterpret_cast<u32x4*>(dst) = u32x4{h[0], h[1], h[2], h[3]};
      *reinterpret_cast<u32x4*>(dst + (size_t)rows * 64) = u32x4{l[0], l[1], l[2], l[3]};
    }
  }
}

// Per clip and fusion layer: the exponents of the cross-attention's K and V operands (attention_h2_kernel).  Row t of clip b of the
// resized visual stream satisfies |x| < 2^14 rscale[b T + t] (interp_linear_h2_kernel), so U = 2^14 max_t rscale bounds the clip, and
// |K| <= U ck + bk, |V| <= U cv + bv with the layer's constants kc[l] = {sqrt(d) max ||w_n||_2, max |b_n|} of its K rows and V rows
// (avsep_api.hip h2_prepare).  out[(b Lf + l) 2 + {0, 1}] = e with bound 2^e <= 2^14.  A clip's exponents depend on that clip alone.
__global__ __launch_bounds__(64) void clip_exp_kernel(const float* __restrict__ rscale, const float* __restrict__ kc, int* __restrict__ out,
                                                      int T, int Lf) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float mx = 0.0f;
  for (int t = lane; t < T; t += 64) mx = fmaxf(mx, rscale[(size_t)b * T + t]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  const float U = 16384.0f * mx;
  for (int i = lane; i < 2 * Lf; i += 64) {
    const float bound = fmaf(U, kc[2 * i], kc[2 * i + 1]) * 1.00001f;
    int e = 0;
    if (bound > 0.0f && bound < 3.0e38f) {
      const int ex = (int)((__float_as_uint(bound) >> 23) & 255u) - 126;   // bound < 2^ex
      e = 14 - ex;
      e = e > 60 ? 60 : e < -60 ? -60 : e;
    }
    out[(size_t)b * 2 * Lf + i] = e;
  }
}

// Profiling aid: keeps the stream busy for `us` microseconds so the host can queue a whole forward behind
// it; the per-kernel HIP events of avsep_profile_* then see back-to-back kernels instead of launch gaps.
__global__ void delay_kernel(unsigned us) {
  const unsigned long long t0 = wall_clock64();          // 100 MHz constant clock
  while (wall_clock64() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(32);
}

// Timeline aid: one lane writes the 100 MHz wall clock into slot `idx` (avsep_read_stamps).
__global__ void stamp_kernel(unsigned long long* buf, int idx) {
  if (threadIdx.x == 0) buf[idx] = wall_clock64();
}

// ------------------------------------------------------------------------------------ weight packers
__global__ void pack_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int K, int Kp,
                                 float scale, int scale_rows) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)rows * Kp) return;
  const int r = (int)(idx / Kp), k = (int)(idx - (size_t)r * Kp);
  float v = k < K ? src[(size_t)r * K + k] : 0.0f;
  if (r < scale_rows) v *= scale;
  dst[idx] = v;
}

__global__ void pack_conv1d_kernel(const float* __restrict__ w, float* __restrict__ dst, int Co, int Ci, int Cip) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)Co * 3 * Cip) return;
  const int ci = (int)(idx % Cip);
  const int tap = (int)((idx / Cip) % 3);
  const int co = (int)(idx / ((size_t)3 * Cip));
  dst[idx] = ci < Ci ? w[((size_t)co * Ci + ci) * 3 + tap] : 0.0f;
}

// BN(eval) folding: y = (conv(x)+b - mean) * gamma / sqrt(var+eps) + beta  (model.py:82-90)
__global__ void pack_conv2d_bn_kernel(const float* __restrict__ w, const float* __restrict__ b,
                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ mean, const float* __restrict__ var,
                                      float* __restrict__ wp, float* __restrict__ bp, int Co, int Ci, float eps) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= Co * 9 * Ci) return;
  const int ci = idx % Ci;
  const int tap = (idx / Ci) % 9;
  const int co = idx / (9 * Ci);
  const float sc = gamma[co] / sqrtf(var[co] + eps);
  const float v = w[((size_t)co * Ci + ci) * 9 + tap] * sc;
  if (Ci == 1) wp[tap * Co + co] = v;            // [9][Co] for the VALU kernel
  else wp[((size_t)co * 9 + tap) * Ci + ci] = v;  // [Co][9][Ci] for the implicit GEMM
  if (ci == 0 && tap == 0) bp[co] = (b[co] - mean[co]) * sc + beta[co];
}

__global__ void scale_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, float scale,
                                  int scale_n) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx < n) dst[idx] = idx < scale_n ? src[idx] * scale : src[idx];
}

}  // namespace

hipError_t launch_stamp(unsigned long long* buf, int idx, hipStream_t s) {
  hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s, buf, idx);
  return hipGetLastError();
}

hipError_t launch_delay(unsigned us, hipStream_t s) {
  hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, s, us);
  return hipGetLastError();
}

hipError_t launch_layernorm(const float* x, const float* g, const float* b, float* y, int M, int d, float eps,
                            hipStream_t s) {
  if (M <= 0 || d <= 0 || (d & 3) || d > 2048) return hipErrorInvalidValue;
  const dim3 grid((M + 3) / 4), block(256);
  const int vec = (d + 255) / 256;
  if (vec <= 1) hipLaunchKernelGGL((layernorm_kernel<1>), grid, block, 0, s, x, g, b, y, M, d, eps);
  else if (vec <= 2) hipLaunchKernelGGL((layernorm_kernel<2>), grid, block, 0, s, x, g, b, y, M, d, eps);
  else if (vec <= 4) hipLaunchKernelGGL((layernorm_kernel<4>), grid, block, 0, s, x, g, b, y, M, d, eps);
  else hipLaunchKernelGGL((layernorm_kernel<8>), grid, block, 0, s, x, g, b, y, M, d, eps);
  return hipGetLastError();
}

hipError_t launch_layernorm_planes(const float* x, const float* g, const float* b, unsigned short* yp, long long rows, int M, int d,
                                   float eps, hipStream_t s) {
  if (M <= 0 || d <= 0 || (d & 31) || d > 2048 || rows < M) return hipErrorInvalidValue;
  const dim3 grid((M + 3) / 4), block(256);
  const int vec = (d + 255) / 256;
  float* y = reinterpret_cast<float*>(yp);
  if (vec <= 1) hipLaunchKernelGGL((layernorm_kernel<1, false, 1>), grid, block, 0, s, x, g, b, y, M, d, eps, rows, 1.0f);
  else if (vec <= 2) hipLaunchKernelGGL((layernorm_kernel<2, false, 1>), grid, block, 0, s, x, g, b, y, M, d, eps, rows, 1.0f);
  else if (vec <= 4) hipLaunchKernelGGL((layernorm_kernel<4, false, 1>), grid, block, 0, s, x, g, b, y, M, d, eps, rows, 1.0f);
  else hipLaunchKernelGGL((layernorm_kernel<8, false, 1>), grid, block, 0, s, x, g, b, y, M, d, eps, rows, 1.0f);
  return hipGetLastError();
}

hipError_t launch_layernorm_h2(const float* x, const float* g, const float* b, unsigned short* yp, long long rows, int M, int d,
                               float eps, int e, hipStream_t s) {
  if (M <= 0 || d <= 0 || (d & 31) || d > 2048 || rows < M || e < -120 || e > 120) return hipErrorInvalidValue;
  const dim3 grid((M + 3) / 4), block(256);
  const int vec = (d + 255) / 256;
  float* y = reinterpret_cast<float*>(yp);
  const float sc = ldexpf(1.0f, e);
  if (vec <= 1) hipLaunchKernelGGL((layernorm_kernel<1, false, 2>), grid, block, 0, s, x, g, b, y, M, d, eps, rows, sc);
  else if (vec <= 2) hipLaunchKernelGGL((layernorm_kernel<2, false, 2>), grid, block, 0, s, x, g, b, y, M, d, eps, rows, sc);
  else if (vec <= 4) hipLaunchKernelGGL((layernorm_kernel<4, false, 2>), grid, block, 0, s, x, g, b, y, M, d, eps, rows, sc);
  else hipLaunchKernelGGL((layernorm_kernel<8, false, 2>), grid, block, 0, s, x, g, b, y, M, d, eps, rows, sc);
  return hipGetLastError();
}

const char* layernorm_instance_name(int d, bool stats_only) {
  static thread_local char buf[48];
  const int vec = (d + 255) / 256;
  snprintf(buf, sizeof buf, "layernorm_kernel<%d, %s, 0>", vec <= 1 ? 1 : vec <= 2 ? 2 : vec <= 4 ? 4 : 8, stats_only ? "true" : "false");
  return buf;
}

const char* layernorm_planes_instance_name(int d, int kind) {
  static thread_local char buf[48];
  const int vec = (d + 255) / 256;
  snprintf(buf, sizeof buf, "layernorm_kernel<%d, false, %d>", vec <= 1 ? 1 : vec <= 2 ? 2 : vec <= 4 ? 4 : 8, kind);
  return buf;
}

hipError_t launch_layernorm_stats(const float* x, float* stats, int M, int d, float eps, hipStream_t s) {
  if (M <= 0 || d <= 0 || (d & 3) || d > 2048) return hipErrorInvalidValue;
  const dim3 grid((M + 3) / 4), block(256);
  const int vec = (d + 255) / 256;
  const float* none = nullptr;
  if (vec <= 1) hipLaunchKernelGGL((layernorm_kernel<1, true>), grid, block, 0, s, x, none, none, stats, M, d, eps);
  else if (vec <= 2) hipLaunchKernelGGL((layernorm_kernel<2, true>), grid, block, 0, s, x, none, none, stats, M, d, eps);
  else if (vec <= 4) hipLaunchKernelGGL((layernorm_kernel<4, true>), grid, block, 0, s, x, none, none, stats, M, d, eps);
  else hipLaunchKernelGGL((layernorm_kernel<8, true>), grid, block, 0, s, x, none, none, stats, M, d, eps);
  return hipGetLastError();
}

hipError_t launch_transpose_pad(const float* x, float* y, int B, int F, int T, int Fp, hipStream_t s) {
  const dim3 grid((T + 31) / 32, (Fp + 31) / 32, B), block(256);
  hipLaunchKernelGGL(transpose_pad_kernel, grid, block, 0, s, x, y, F, T, Fp);
  return hipGetLastError();
}

hipError_t launch_conv1_c1(const float* frames, const float* w9x32, const float* bias32, float* out, int M, int H,
                           int W, int Ho, int Wo, hipStream_t s) {
  const long total = (long)M * Ho * Wo;
  if (total <= 0 || total > 0x7fffffffL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(conv1_c1_kernel, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, s, frames, w9x32, bias32,
                     out, (int)total, H, W, Ho, Wo);
  return hipGetLastError();
}

hipError_t launch_avgpool(const float* x, float* y, int M, int P, int C, hipStream_t s) {
  if (C & 3) return hipErrorInvalidValue;
  const int n = M * (C / 4);
  hipLaunchKernelGGL(avgpool_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, y, M, P, C);
  return hipGetLastError();
}

hipError_t launch_interp_linear(const float* x, float* y, int B, int N, int T, int d, hipStream_t s) {
  if (d & 3) return hipErrorInvalidValue;
  const size_t n = (size_t)B * T * (d / 4);
  hipLaunchKernelGGL(interp_linear_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, y, B, N, T, d,
                     (float)N / (float)T, 0LL);
  return hipGetLastError();
}

hipError_t launch_interp_linear_h2(const float* x, unsigned short* yp, float* rscale, long long rows, int B, int N, int T, int d,
                                   hipStream_t s) {
  if ((d & 31) || d > 2048 || rows < (long long)B * T || !rscale) return hipErrorInvalidValue;
  const long long nrow = (long long)B * T;
  const dim3 grid((unsigned)((nrow + 3) / 4)), block(256);
  const float sc = (float)N / (float)T;
  if (d <= 512) hipLaunchKernelGGL(interp_linear_h2_kernel<1>, grid, block, 0, s, x, yp, rscale, rows, B, N, T, d, sc);
  else if (d <= 1024) hipLaunchKernelGGL(interp_linear_h2_kernel<2>, grid, block, 0, s, x, yp, rscale, rows, B, N, T, d, sc);
  else hipLaunchKernelGGL(interp_linear_h2_kernel<4>, grid, block, 0, s, x, yp, rscale, rows, B, N, T, d, sc);
  return hipGetLastError();
}

hipError_t launch_clip_exp(const float* rscale, const float* kc, int* out, int B, int T, int Lf, hipStream_t s) {
  if (B <= 0 || T <= 0 || Lf <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(clip_exp_kernel, dim3((unsigned)B), dim3(64), 0, s, rscale, kc, out, T, Lf);
  return hipGetLastError();
}

hipError_t launch_interp_linear_planes(const float* x, unsigned short* yp, long long rows, int B, int N, int T, int d, hipStream_t s) {
  if ((d & 31) || rows < (long long)B * T) return hipErrorInvalidValue;
  const size_t n = (size_t)B * T * (d / 4);
  hipLaunchKernelGGL(interp_linear_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, reinterpret_cast<float*>(yp), B,
                     N, T, d, (float)N / (float)T, rows);
  return hipGetLastError();
}

hipError_t launch_pack_rows(const float* src, float* dst, int rows, int K, int Kp, float scale, int scale_rows,
                            hipStream_t s) {
  const size_t n = (size_t)rows * Kp;
  hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, rows, K, Kp,
                     scale, scale_rows);
  return hipGetLastError();
}

hipError_t launch_pack_conv1d(const float* w, float* dst, int Co, int Ci, int Cip, hipStream_t s) {
  const size_t n = (size_t)Co * 3 * Cip;
  hipLaunchKernelGGL(pack_conv1d_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, dst, Co, Ci, Cip);
  return hipGetLastError();
}

hipError_t launch_pack_conv2d_bn(const float* w, const float* b, const float* gamma, const float* beta,
                                 const float* mean, const float* var, float* wp, float* bp, int Co, int Ci,
                                 float eps, hipStream_t s) {
  const int n = Co * 9 * Ci;
  hipLaunchKernelGGL(pack_conv2d_bn_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w, b, gamma, beta, mean, var,
                     wp, bp, Co, Ci, eps);
  return hipGetLastError();
}

namespace {
// one workgroup per output feature n: wp[n][:] = w[n][:] o gamma, c1[n] = sum_k wp[n][k], c2[n] = sum_k w[n][k] beta[k] + bias[n]
__global__ __launch_bounds__(256) void pack_lnx_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ wp, float* __restrict__ c1, float* __restrict__ c2,
                                                       int K) {
  const int n = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int k = threadIdx.x; k < K; k += 256) {
    const float wv = w[(size_t)n * K + k];
    const float v = wv * gamma[k];
    wp[(size_t)n * K + k] = v;
    s1 += (double)v;                       // the sum of the ROUNDED products: what the GEMM multiplies by
    s2 += (double)wv * (double)beta[k];
  }
  __shared__ double r1[256], r2[256];
  r1[threadIdx.x] = s1;
  r2[threadIdx.x] = s2;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) {
      r1[threadIdx.x] += r1[threadIdx.x + off];
      r2[threadIdx.x] += r2[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    c1[n] = (float)r1[0];
    c2[n] = (float)(r2[0] + (bias ? (double)bias[n] : 0.0));
  }
}
}  // namespace

hipError_t launch_pack_lnx(const float* w, const float* bias, const float* gamma, const float* beta, float* wp, float* c1,
                           float* c2, int N, int K, hipStream_t s) {
  if (!w || !gamma || !beta || !wp || !c1 || !c2 || N <= 0 || K <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(pack_lnx_kernel, dim3(N), dim3(256), 0, s, w, bias, gamma, beta, wp, c1, c2, K);
  return hipGetLastError();
}

hipError_t launch_scale_copy(const float* src, float* dst, int n, float scale, int scale_n, hipStream_t s) {
  hipLaunchKernelGGL(scale_copy_kernel, dim3((n + 255) / 256), dim3(256), 0, s, src, dst, n, scale, scale_n);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// Hann-windowed real-DFT basis for the STFT front-end (dataset.py:122-135: np.hanning(n_fft) window, np.fft.rfft):
// basis[2f][k] = w[k] cos(2 pi f k / n), basis[2f+1][k] = -w[k] sin(2 pi f k / n), w[k] = 0.5 - 0.5 cos(2 pi k / (n-1)).
// Angles are reduced exactly in integers (f*k mod n) and evaluated in double precision; one rounding to fp32.
namespace {
__global__ __launch_bounds__(256) void stft_basis_kernel(float* __restrict__ basis, int n) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int F = n / 2 + 1;
  if (idx >= (size_t)2 * F * n) return;
  const int row = (int)(idx / n), k = (int)(idx - (size_t)row * n);
  const int f = row >> 1;
  const double two_pi = 6.283185307179586476925286766559;
  const double w = n > 1 ? 0.5 - 0.5 * cos(two_pi * (double)k / (double)(n - 1)) : 1.0;
  const double ang = two_pi * (double)(((long long)f * k) % n) / (double)n;
  basis[idx] = (float)((row & 1) ? -w * sin(ang) : w * cos(ang));
}
}  // namespace

hipError_t launch_stft_basis(float* basis, int n_fft, hipStream_t s) {
  if (!basis || n_fft <= 0) return hipErrorInvalidValue;
  const size_t n = (size_t)2 * (n_fft / 2 + 1) * n_fft;
  hipLaunchKernelGGL(stft_basis_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, basis, n_fft);
  return hipGetLastError();
}
