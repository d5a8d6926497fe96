// train_ops.hip -- kernels of the TRAINING path: the memory-bound forward/backward
// pieces that the inference path fuses away or never needs.  Round-1 goal is a CORRECT fwd+bwd with gradient
// parity against the reference (tests/golden/train_*.npz); each op is its own launch, driven from Python
// autograd wrappers (av_separation/_train.py).  All dense contractions, forward and backward, still run on
// the fp32-MFMA kernels of gemm.hip (dX = dY W on gemm_kernel; dW = dY^T X on wgrad_kernel, which reads both operands
// as they lie in memory -- the transposing path below is only used when N is not a multiple of 4).
//
// Layout convention: every activation is a row tensor [rows][C] (channels last), so conv layers become
// im2col + GEMM and their backward col2im + GEMM; BatchNorm / bias / LayerNorm-affine gradients are
// column reductions done in two deterministic stages (no float atomics: bit-reproducible gradients).
#include "kernels.h"

#include <algorithm>
#include <cstdint>

namespace {

// ------------------------------------------------------------------------------------------ transposes / im2col
// y[c][r] = x[r][c] for r < R, zero for R <= r < Rp  (x: [R][C], y: [C][Rp]); used for dW = dY^T X operands.
__global__ __launch_bounds__(256) void transpose2d_kernel(const float* __restrict__ x, float* __restrict__ y, int R,
                                                          int C, int Rp) {
  __shared__ float tile[32][33];
  const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (r < R && c < C) ? x[(size_t)r * C + c] : 0.0f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (c < C && r < Rp) y[(size_t)c * Rp + r] = tile[tx][ty + 8 * i];
  }
}

// The same transposition for a whole table of matrices in ONE launch (blockIdx.z = table entry): the W^T operands of
// every Linear layer's activation-gradient GEMM are made once per training step instead of one ~5 us launch in front of
// each of them (76 per cfg4 step).
__global__ __launch_bounds__(256) void transpose_many_kernel(const TransposeDesc* __restrict__ table) {
  __shared__ float tile[32][33];
  const TransposeDesc d = table[blockIdx.z];
  const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  if (r0 >= d.Rp || c0 >= d.C) return;                 // block-uniform: the grid is sized for the largest entry
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (r < d.R && c < d.C) ? d.src[(size_t)r * d.C + c] : 0.0f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (c < d.C && r < d.Rp) d.dst[(size_t)c * d.Rp + r] = tile[tx][ty + 8 * i];
  }
}

// Conv1d(k3,p1) im2col on sequence rows: x [B*T][C] -> col [B*T][3*C], col[m][tap*C + c] = x[m+tap-1][c] inside
// the same sequence, else 0 (model.py:38,40).
__global__ __launch_bounds__(256) void im2col1d_kernel(const float* __restrict__ x, float* __restrict__ col, int M,
                                                       int T, int C) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int cq = C >> 2;
  if (idx >= (size_t)M * 3 * cq) return;
  const int c4 = (int)(idx % cq);
  const int tap = (int)((idx / cq) % 3);
  const int m = (int)(idx / ((size_t)3 * cq));
  const int t = m % T + tap - 1;
  f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
  if (t >= 0 && t < T) v = *reinterpret_cast<const f32x4*>(x + (size_t)(m + tap - 1) * C + 4 * c4);
  *reinterpret_cast<f32x4*>(col + (size_t)m * 3 * C + tap * C + 4 * c4) = v;
}

// adjoint of im2col1d: dx[m][c] = sum_tap dcol[m-tap+1][tap*C + c] (rows of the same sequence only)
__global__ __launch_bounds__(256) void col2im1d_kernel(const float* __restrict__ dcol, float* __restrict__ dx, int M,
                                                       int T, int C) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int cq = C >> 2;
  if (idx >= (size_t)M * cq) return;
  const int c4 = (int)(idx % cq);
  const int m = (int)(idx / cq);
  const int t = m % T;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int tap = 0; tap < 3; ++tap) {
    const int to = t - tap + 1;   // output row whose tap `tap` read input row t
    if (to >= 0 && to < T) acc += *reinterpret_cast<const f32x4*>(dcol + (size_t)(m - tap + 1) * 3 * C + tap * C + 4 * c4);
  }
  *reinterpret_cast<f32x4*>(dx + (size_t)m * C + 4 * c4) = acc;
}

// Conv2d(k3,s2,p1) im2col on channels-last images: x [I][H][W][C] -> col [I*Ho*Wo][Kp], column tap*C + c
// (tap = ky*3+kx), zero outside the image and in the K padding (Kp >= 9*C, multiple of 32).
__global__ __launch_bounds__(256) void im2col2d_kernel(const float* __restrict__ x, float* __restrict__ col, int I,
                                                       int H, int W, int C, int Ho, int Wo, int Kp) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t rows = (size_t)I * Ho * Wo;
  if (idx >= rows * Kp) return;
  const int k = (int)(idx % Kp);
  const size_t row = idx / Kp;
  float v = 0.0f;
  if (k < 9 * C) {
    const int tap = k / C, c = k - tap * C;
    const int ky = tap / 3, kx = tap - 3 * ky;
    const int img = (int)(row / (Ho * Wo));
    const int rem = (int)(row - (size_t)img * Ho * Wo);
    const int y = rem / Wo, xo = rem - y * Wo;
    const int iy = 2 * y - 1 + ky, ix = 2 * xo - 1 + kx;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[(((size_t)img * H + iy) * W + ix) * C + c];
  }
  col[idx] = v;
}

// adjoint of im2col2d: dx[img][iy][ix][c] = sum over the (<= 4) windows that read this pixel
__global__ __launch_bounds__(256) void col2im2d_kernel(const float* __restrict__ dcol, float* __restrict__ dx, int I,
                                                       int H, int W, int C, int Ho, int Wo, int Kp) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)I * H * W * C) return;
  const int c = (int)(idx % C);
  const size_t px = idx / C;
  const int ix = (int)(px % W);
  const int iy = (int)((px / W) % H);
  const int img = (int)(px / ((size_t)W * H));
  float acc = 0.0f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int ty = iy + 1 - ky;          // 2*y = iy + 1 - ky
    if (ty < 0 || (ty & 1)) continue;
    const int y = ty >> 1;
    if (y >= Ho) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int txx = ix + 1 - kx;
      if (txx < 0 || (txx & 1)) continue;
      const int xo = txx >> 1;
      if (xo >= Wo) continue;
      acc += dcol[(((size_t)img * Ho + y) * Wo + xo) * Kp + (ky * 3 + kx) * C + c];
    }
  }
  dx[idx] = acc;
}

// ------------------------------------------------------------------------------------------ column reductions
// Deterministic two-stage column sums of a row-major [M][C] matrix.  A 256-thread block covers CW columns (power of
// two, 32..256) x RL = 256/CW row lanes, so narrow matrices (C = 32 conv channels x 300k rows) still fill their
// wavefronts; the row lanes are folded through LDS in a fixed order.
//   stage 1: part[blk][j][c] = sum over the block's rows of f_j(row, c)     grid (nb, ceil(C/CW))
//   stage 2: out_j[c] = scale * sum_blk part[blk][j][c]                       same kernel shape, 32 columns x 8 lanes
// MODE 0: f0 = a                      (bias gradients, BN mean)
// MODE 1: f0 = a, f1 = a*b            (BN/LN affine gradients: a = dy, b = xhat)
// MODE 2: f0 = (a - b[c])^2           (BN variance, b = mean)
struct RedPlan { int cwl, rows, nb; };
inline RedPlan red_plan(int M, int C) {
  RedPlan r;
  r.cwl = 5;
  while (r.cwl < 8 && (1 << r.cwl) < C) ++r.cwl;
  const int cw = 1 << r.cwl, rl = 256 / cw;
  const int nby = (C + cw - 1) / cw;
  int nb = (1024 + nby - 1) / nby;                       // ~1024 workgroups in stage 1
  const int maxnb = (M + 4 * rl - 1) / (4 * rl);         // at least 4 rows per lane
  nb = std::max(1, std::min(std::min(nb, maxnb), 512));
  r.rows = (M + nb - 1) / nb;
  r.rows = (r.rows + rl - 1) / rl * rl;
  r.nb = (M + r.rows - 1) / r.rows;
  return r;
}

template <int MODE>
__global__ __launch_bounds__(256) void colreduce1_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ part, int M, int C, int rows, int cwl) {
  __shared__ float red[2][256];
  const int cw = 1 << cwl, rl = 256 >> cwl;
  const int cl = threadIdx.x & (cw - 1), lane = threadIdx.x >> cwl;
  const int c = blockIdx.y * cw + cl;
  const int r0 = blockIdx.x * rows, r1 = min(r0 + rows, M);
  float s0 = 0.f, s1 = 0.f;
  if (c < C) {
    const float mu = MODE == 2 ? b[c] : 0.f;
    int r = r0 + lane;
    // 4 independent loads in flight per lane
    for (; r + 3 * rl < r1; r += 4 * rl) {
      float v[4], w[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        v[u] = a[(size_t)(r + u * rl) * C + c];
        if (MODE == 1) w[u] = b[(size_t)(r + u * rl) * C + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (MODE == 2) { const float d = v[u] - mu; s0 = fmaf(d, d, s0); }
        else s0 += v[u];
        if (MODE == 1) s1 = fmaf(v[u], w[u], s1);
      }
    }
    for (; r < r1; r += rl) {
      const float v = a[(size_t)r * C + c];
      if (MODE == 2) { const float d = v - mu; s0 = fmaf(d, d, s0); }
      else s0 += v;
      if (MODE == 1) s1 = fmaf(v, b[(size_t)r * C + c], s1);
    }
  }
  red[0][threadIdx.x] = s0;
  if (MODE == 1) red[1][threadIdx.x] = s1;
  __syncthreads();
  if (lane == 0 && c < C) {
    for (int l = 1; l < rl; ++l) {
      s0 += red[0][l * cw + cl];
      if (MODE == 1) s1 += red[1][l * cw + cl];
    }
    const int nj = MODE == 1 ? 2 : 1;
    part[((size_t)blockIdx.x * nj + 0) * C + c] = s0;
    if (MODE == 1) part[((size_t)blockIdx.x * nj + 1) * C + c] = s1;
  }
}

// stage 2 over the nblk partial rows of width W = nj*C: 32 columns x 8 row lanes per block
__global__ __launch_bounds__(256) void colreduce2_kernel(const float* __restrict__ part, float* __restrict__ out0,
                                                         float* __restrict__ out1, int nblk, int nj, int C, float scale) {
  __shared__ float red[256];
  const int cl = threadIdx.x & 31, lane = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl, Wd = nj * C;
  float s = 0.f;
  if (c < Wd) {
    // the partial rows were just written by other XCDs: issue 16 loads together instead of paying their latency serially
    for (int base = lane; base < nblk; base += 128) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = (base + 8 * u < nblk) ? part[(size_t)(base + 8 * u) * Wd + c] : 0.0f;
#pragma unroll
      for (int u = 0; u < 16; ++u) s += v[u];
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (lane == 0 && c < Wd) {
    for (int l = 1; l < 8; ++l) s += red[l * 32 + cl];
    if (c < C) out0[c] = s * scale;
    else out1[c - C] = s * scale;
  }
}

// out[i] = sum_z part[z][i]  (split-K weight gradients; slices summed in a fixed order)
__global__ __launch_bounds__(256) void sum_slices_kernel(const float* __restrict__ part, float* __restrict__ out, int S,
                                                         size_t n4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4* p4 = reinterpret_cast<const float4*>(part);
  float4 acc = p4[i];
  for (int z = 1; z < S; ++z) {
    const float4 v = p4[(size_t)z * n4 + i];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  reinterpret_cast<float4*>(out)[i] = acc;
}

// ------------------------------------------------------------------------------------------ BatchNorm (train)
// y = relu?( (x - mean) * rstd * gamma + beta ),  xhat saved for the backward
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ var, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ xhat,
                                                       float* __restrict__ y, size_t n, int C, int relu, float eps) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  const int c = (int)(idx % C);
  const float xh = (x[idx] - mean[c]) * (1.0f / sqrtf(var[c] + eps));
  xhat[idx] = xh;
  const float v = xh * gamma[c] + beta[c];
  y[idx] = relu ? fmaxf(v, 0.0f) : v;
}

// dx = gamma*rstd * ( dyr - sum(dyr)/M - xhat * sum(dyr*xhat)/M ),  dyr = dy masked by the ReLU (y > 0)
__global__ __launch_bounds__(256) void bn_bwd_kernel(const float* __restrict__ dyr, const float* __restrict__ xhat,
                                                     const float* __restrict__ gamma, const float* __restrict__ var,
                                                     const float* __restrict__ sum_dy, const float* __restrict__ sum_dyx,
                                                     float* __restrict__ dx, size_t n, int C, float invM, float eps) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  const int c = (int)(idx % C);
  const float rstd = 1.0f / sqrtf(var[c] + eps);
  dx[idx] = gamma[c] * rstd * (dyr[idx] - sum_dy[c] * invM - xhat[idx] * sum_dyx[c] * invM);
}

// running_mean/var <- (1-mom)*running + mom*batch (unbiased variance M/(M-1)), nn.BatchNorm2d training update
__global__ void bn_running_kernel(float* __restrict__ rmean, float* __restrict__ rvar, const float* __restrict__ mean,
                                  const float* __restrict__ var, int C, float momentum, float unbias) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  rmean[c] = (1.0f - momentum) * rmean[c] + momentum * mean[c];
  rvar[c] = (1.0f - momentum) * rvar[c] + momentum * var[c] * unbias;
}

// ------------------------------------------------------------------------------------------ elementwise
// act: 1 relu (aux = y), 2 gelu (aux = pre-activation), 3 sigmoid (aux = y)
// The elementwise kernels move V = 4 floats per thread (one 16-byte access each way) when the element count allows:
// with 4-byte accesses they were bound by the number of memory instructions, not by HBM.
template <int V> struct VecT { typedef float type; };
template <> struct VecT<4> { typedef f32x4 type; };
template <int V> __device__ __forceinline__ float& lane_of(typename VecT<V>::type& v, int e);
template <> __device__ __forceinline__ float& lane_of<1>(float& v, int) { return v; }
template <> __device__ __forceinline__ float& lane_of<4>(f32x4& v, int e) { return reinterpret_cast<float*>(&v)[e]; }

// ---- im2col2d / col2im2d, round 3: the kernels above pay five runtime integer divisions (two of them 64-bit) PER SCALAR ELEMENT,
// ~150 VALU instructions per float moved: 56 us per launch where the bytes take 20 (PMC: the three im2col2d launches of a cfg4
// step were the third-largest VALU consumer of the whole step).  Here one thread moves V = 4 channels with 16-byte accesses
// and every quotient is a multiply-high by a host-computed magic number (floor(2^32 / d) + 1, exact while n * d < 2^32 --
// the launcher checks the ranges and otherwise takes the kernels above).  Same values, element for element.
__device__ __forceinline__ unsigned mdiv(unsigned n, unsigned mg) { return mg ? __umulhi(n, mg) : n; }   // mg == 0 <=> d == 1

template <int V>
__global__ __launch_bounds__(256) void im2col2d_fast_kernel(const float* __restrict__ x, float* __restrict__ col, unsigned total,
                                                            int H, int W, int C, int Ho, int Wo, int Kp, unsigned mg_kpv,
                                                            unsigned mg_c, unsigned mg_hw, unsigned mg_wo) {
  typedef typename VecT<V>::type T;
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;
  if (idx >= total) return;
  const unsigned kpv = (unsigned)Kp / V;
  const unsigned row = mdiv(idx, mg_kpv);
  const int k = (int)(idx - row * kpv) * V;
  T v;
#pragma unroll
  for (int e = 0; e < V; ++e) lane_of<V>(v, e) = 0.0f;
  if (k < 9 * C) {
    const int tap = (int)mdiv((unsigned)k, mg_c), c = k - tap * C;
    const int ky = tap / 3, kx = tap - 3 * ky;
    const unsigned img = mdiv(row, mg_hw);
    const int rem = (int)(row - img * (unsigned)(Ho * Wo));
    const int y = (int)mdiv((unsigned)rem, mg_wo), xo = rem - y * Wo;
    const int iy = 2 * y - 1 + ky, ix = 2 * xo - 1 + kx;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const T*>(x + (((size_t)img * H + iy) * W + ix) * C + c);
  }
  *reinterpret_cast<T*>(col + (size_t)idx * V) = v;
}

template <int V>
__global__ __launch_bounds__(256) void col2im2d_fast_kernel(const float* __restrict__ dcol, float* __restrict__ dx, unsigned total,
                                                            int H, int W, int C, int Ho, int Wo, int Kp, unsigned mg_cv,
                                                            unsigned mg_w, unsigned mg_h) {
  typedef typename VecT<V>::type T;
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;
  if (idx >= total) return;
  const unsigned cv = (unsigned)C / V;
  const unsigned px = mdiv(idx, mg_cv);
  const int c = (int)(idx - px * cv) * V;
  const unsigned rowp = mdiv(px, mg_w);
  const int ix = (int)(px - rowp * (unsigned)W);
  const unsigned img = mdiv(rowp, mg_h);
  const int iy = (int)(rowp - img * (unsigned)H);
  T acc;
#pragma unroll
  for (int e = 0; e < V; ++e) lane_of<V>(acc, e) = 0.0f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int ty = iy + 1 - ky;          // 2*y = iy + 1 - ky
    if (ty < 0 || (ty & 1)) continue;
    const int y = ty >> 1;
    if (y >= Ho) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int txx = ix + 1 - kx;
      if (txx < 0 || (txx & 1)) continue;
      const int xo = txx >> 1;
      if (xo >= Wo) continue;
      T w = *reinterpret_cast<const T*>(dcol + (((size_t)img * Ho + y) * Wo + xo) * Kp + (ky * 3 + kx) * C + c);
#pragma unroll
      for (int e = 0; e < V; ++e) lane_of<V>(acc, e) += lane_of<V>(w, e);
    }
  }
  *reinterpret_cast<T*>(dx + (size_t)idx * V) = acc;
}

__device__ __forceinline__ float act_value(float v, int act) {
  if (act == ACT_RELU) return fmaxf(v, 0.0f);
  if (act == ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  if (act == ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
  return v;
}
__device__ __forceinline__ float act_grad(float g, float a, int act) {
  if (act == ACT_RELU) return a > 0.0f ? g : 0.0f;
  if (act == ACT_GELU) {
    // d/dx [ x Phi(x) ] = Phi(x) + x phi(x)
    const float cdf = 0.5f * (1.0f + erff(a * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * a * a);
    return g * (cdf + a * pdf);
  }
  if (act == ACT_SIGMOID) return g * a * (1.0f - a);
  return g;
}

template <int V>
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t nv,
                                                      int act) {
  typedef typename VecT<V>::type T;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= nv) return;
  T v = reinterpret_cast<const T*>(x)[idx];
#pragma unroll
  for (int e = 0; e < V; ++e) lane_of<V>(v, e) = act_value(lane_of<V>(v, e), act);
  reinterpret_cast<T*>(y)[idx] = v;
}

template <int V>
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ aux,
                                                      float* __restrict__ dx, size_t nv, int act) {
  typedef typename VecT<V>::type T;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= nv) return;
  T g = reinterpret_cast<const T*>(dy)[idx], a = reinterpret_cast<const T*>(aux)[idx];
#pragma unroll
  for (int e = 0; e < V; ++e) lane_of<V>(g, e) = act_grad(lane_of<V>(g, e), lane_of<V>(a, e), act);
  reinterpret_cast<T*>(dx)[idx] = g;
}

// backward of y = dropout(relu(z)) (the FFN's hidden activation, fused into the GEMM epilogue) from y alone
template <int V>
__global__ __launch_bounds__(256) void relu_dropout_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                               float* __restrict__ dx, size_t nv, float p) {
  typedef typename VecT<V>::type T;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= nv) return;
  T g = reinterpret_cast<const T*>(dy)[idx], a = reinterpret_cast<const T*>(y)[idx];
#pragma unroll
  for (int e = 0; e < V; ++e) lane_of<V>(g, e) = lane_of<V>(a, e) > 0.0f ? lane_of<V>(g, e) * dropout_scale(p) : 0.0f;
  reinterpret_cast<T*>(dx)[idx] = g;
}

// out[m][s*F + f] = a[m][s*F + f] * xt[m][f]   (SeparationDecoder.separate and its adjoint w.r.t. the masks)
__global__ __launch_bounds__(256) void mul_mixed_kernel(const float* __restrict__ a, const float* __restrict__ xt,
                                                        float* __restrict__ out, size_t M, int S, int F, int ldx) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * S * F) return;
  const int f = (int)(idx % F);
  const size_t m = idx / ((size_t)S * F);
  out[idx] = a[idx] * xt[m * ldx + f];
}

// inverted dropout: y = keep ? x / (1-p) : 0 with the stateless mask of kernels.h (the backward applies the same op
// to the gradient with the same seed); the mask is a function of the ELEMENT index, whatever the vector width
template <int V>
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, size_t nv,
                                                      float p, unsigned long long seed) {
  typedef typename VecT<V>::type T;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= nv) return;
  T v = reinterpret_cast<const T*>(x)[idx];
#pragma unroll
  for (int e = 0; e < V; ++e) lane_of<V>(v, e) = dropout_keep(seed, idx * V + e, p) ? lane_of<V>(v, e) * dropout_scale(p) : 0.0f;
  reinterpret_cast<T*>(y)[idx] = v;
}

// y = r + dropout(x): the residual add behind dropout1 / dropout2 of a transformer block in the dropout's launch (same
// two roundings as the separate kernels: the rescaled value, then the sum)
template <int V>
__global__ __launch_bounds__(256) void dropout_add_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                          float* __restrict__ y, size_t nv, float p,
                                                          unsigned long long seed) {
  typedef typename VecT<V>::type T;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= nv) return;
  T v = reinterpret_cast<const T*>(x)[idx];
  T w = reinterpret_cast<const T*>(r)[idx];
#pragma unroll
  for (int e = 0; e < V; ++e) {
    const float d = dropout_keep(seed, idx * V + e, p) ? lane_of<V>(v, e) * dropout_scale(p) : 0.0f;
    lane_of<V>(v, e) = lane_of<V>(w, e) + d;
  }
  reinterpret_cast<T*>(y)[idx] = v;
}

// y[m][c] = x[m][c] + r[m % period][c]   (x + pe[:, :L], model.py:300, when it cannot ride a GEMM epilogue); C % V == 0
template <int V>
__global__ __launch_bounds__(256) void add_rows_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                       float* __restrict__ y, size_t nv, int Cv, int period) {
  typedef typename VecT<V>::type T;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= nv) return;
  const size_t m = idx / Cv;
  T v = reinterpret_cast<const T*>(x)[idx];
  const T w = reinterpret_cast<const T*>(r)[(m % period) * Cv + idx % Cv];
#pragma unroll
  for (int e = 0; e < V; ++e) lane_of<V>(v, e) += lane_of<V>(const_cast<T&>(w), e);
  reinterpret_cast<T*>(y)[idx] = v;
}

// AdaptiveAvgPool2d(1) adjoint: dx[m][p][c] = dy[m][c] / P
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int M,
                                                          int P, int C) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)M * P * C) return;
  const int c = (int)(idx % C);
  const int m = (int)(idx / ((size_t)P * C));
  dx[idx] = dy[(size_t)m * C + c] / (float)P;
}

// linear-interpolation adjoint (model.py:115) as a GATHER: one thread per (batch, source row n, 4 channels) sums the
// contributions of the few output rows t whose two taps include n, in ascending t and tap order -- the order the
// straightforward scatter loop adds them in, so the result is deterministic (no atomics) and bit-identical to it.
// (The first version walked all T outputs serially in B*d/4 threads: 131 us per call at config 4.)
__global__ __launch_bounds__(256) void interp_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B,
                                                         int N, int T, int d, float scale) {
  const int dq = d >> 2;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)B * N * dq) return;
  const int c4 = (int)(idx % dq);
  const int n = (int)((idx / dq) % N);
  const int b = (int)(idx / ((size_t)dq * N));
  // rows t with floor(src(t)) in {n-1, n}: src(t) = scale*(t+0.5) - 0.5 in [n-1, n+1) -> a window around n/scale; two
  // extra rows each side cover the fp rounding of the bounds (every candidate is re-derived exactly like the forward)
  const float inv = (float)T / (float)N;
  int t_lo = (int)floorf(((float)n - 0.5f) * inv - 0.5f) - 2;
  int t_hi = (int)ceilf(((float)n + 1.5f) * inv - 0.5f) + 2;
  t_lo = t_lo < 0 ? 0 : t_lo;
  t_hi = t_hi > T - 1 ? T - 1 : t_hi;
  if (n == N - 1) t_hi = T - 1;                  // both taps clamp to the last row for every t beyond it
  if (n == 0) t_lo = 0;                          // src clamps to 0 for the first rows
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int t = t_lo; t <= t_hi; ++t) {
    float src = fmaf(scale, (float)t + 0.5f, -0.5f);
    src = src < 0.0f ? 0.0f : src;
    int i0 = (int)src;
    i0 = i0 < N - 1 ? i0 : N - 1;
    const int i1 = i0 + 1 < N ? i0 + 1 : N - 1;
    if (i0 != n && i1 != n) continue;
    const float w1 = src - (float)i0, w0 = 1.0f - w1;
    const f32x4 g = *reinterpret_cast<const f32x4*>(dy + ((size_t)b * T + t) * d + 4 * c4);
    if (i0 == n) acc += w0 * g;
    if (i1 == n) acc += w1 * g;
  }
  *reinterpret_cast<f32x4*>(dx + ((size_t)b * N + n) * d + 4 * c4) = acc;
}

// ------------------------------------------------------------------------------------------ LayerNorm backward
// one wavefront per row: dx = rstd * ( g - mean(g) - xhat * mean(g*xhat) ), g = dy*gamma; also writes xhat so the
// affine gradients (dgamma = sum dy*xhat, dbeta = sum dy) can use the generic column reduction.
template <int VEC>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ gam, const float* __restrict__ dres,
                                                            float* __restrict__ dx, float* __restrict__ xhat, int M, int d,
                                                            float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (size_t)row * d;
  const float* gr = dy + (size_t)row * d;
  f32x4 v[VEC], g[VEC];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const int col = (lane + 64 * i) * 4;
    const bool ok = col < d;
    v[i] = ok ? *reinterpret_cast<const f32x4*>(xr + col) : f32x4{0.f, 0.f, 0.f, 0.f};
    g[i] = ok ? *reinterpret_cast<const f32x4*>(gr + col) * *reinterpret_cast<const f32x4*>(gam + col)
              : f32x4{0.f, 0.f, 0.f, 0.f};
    s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
  const float mean = s / (float)d;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < VEC; ++i)
    if ((lane + 64 * i) * 4 < d) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float c = v[i][e] - mean;
        sq += c * c;
      }
    }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off);
  const float rstd = 1.0f / sqrtf(sq / (float)d + eps);
  float sg = 0.f, sgx = 0.f;
#pragma unroll
  for (int i = 0; i < VEC; ++i)
    if ((lane + 64 * i) * 4 < d) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[i][e] = (v[i][e] - mean) * rstd;   // xhat
        sg += g[i][e];
        sgx = fmaf(g[i][e], v[i][e], sgx);
      }
    }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    sg += __shfl_xor(sg, off);
    sgx += __shfl_xor(sgx, off);
  }
  const float mg = sg / (float)d, mgx = sgx / (float)d;
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const int col = (lane + 64 * i) * 4;
    if (col < d) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = rstd * (g[i][e] - mg - v[i][e] * mgx);
      if (dres) o += *reinterpret_cast<const f32x4*>(dres + (size_t)row * d + col);   // the residual path's gradient
      *reinterpret_cast<f32x4*>(dx + (size_t)row * d + col) = o;
      *reinterpret_cast<f32x4*>(xhat + (size_t)row * d + col) = v[i];
    }
  }
}

// LayerNorm backward with the affine gradients' first reduction stage inside (round 3): a workgroup takes 4 * RPW consecutive
// rows (wave w the rows RPW w .. RPW w + RPW - 1 of the strip), every lane keeps the column sums of dy and dy * xhat of its
// wave's rows in registers, the four waves' sums are added in wave order through LDS, and the strip's two partial rows go to
// part[strip][2][d] -- the layout colreduce2_kernel reads.  Against layernorm_bwd_kernel + colreduce1_kernel: xhat is never
// written (M d floats) and dy / xhat are not read a second time (2 M d floats), one launch less per LayerNorm.  The row
// arithmetic is layernorm_bwd_kernel's, statement for statement (dx bit-identical); the column sums are added in another
// (fixed) order than colreduce1's.
template <int VEC>
__global__ __launch_bounds__(256) void layernorm_bwd_fused_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                  const float* __restrict__ gam, const float* __restrict__ dres,
                                                                  float* __restrict__ dx, float* __restrict__ part, int M, int d,
                                                                  int rpw, float eps) {
  extern __shared__ __attribute__((aligned(16))) float red[];      // [4 waves][2][d]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = (blockIdx.x * 4 + wave) * rpw;
  f32x4 cs[VEC], csx[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) cs[i] = csx[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int rr = 0; rr < rpw; ++rr) {
    const int row = row0 + rr;
    if (row >= M) break;                                            // wave-uniform
    const float* xr = x + (size_t)row * d;
    const float* gr = dy + (size_t)row * d;
    f32x4 v[VEC], g[VEC], dyv[VEC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const int col = (lane + 64 * i) * 4;
      const bool ok = col < d;
      v[i] = ok ? *reinterpret_cast<const f32x4*>(xr + col) : f32x4{0.f, 0.f, 0.f, 0.f};
      dyv[i] = ok ? *reinterpret_cast<const f32x4*>(gr + col) : f32x4{0.f, 0.f, 0.f, 0.f};
      g[i] = ok ? dyv[i] * *reinterpret_cast<const f32x4*>(gam + col) : f32x4{0.f, 0.f, 0.f, 0.f};
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s / (float)d;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i)
      if ((lane + 64 * i) * 4 < d) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float c = v[i][e] - mean;
          sq += c * c;
        }
      }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off);
    const float rstd = 1.0f / sqrtf(sq / (float)d + eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i)
      if ((lane + 64 * i) * 4 < d) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[i][e] = (v[i][e] - mean) * rstd;   // xhat
          sg += g[i][e];
          sgx = fmaf(g[i][e], v[i][e], sgx);
        }
      }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      sg += __shfl_xor(sg, off);
      sgx += __shfl_xor(sgx, off);
    }
    const float mg = sg / (float)d, mgx = sgx / (float)d;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const int col = (lane + 64 * i) * 4;
      if (col < d) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rstd * (g[i][e] - mg - v[i][e] * mgx);
        if (dres) o += *reinterpret_cast<const f32x4*>(dres + (size_t)row * d + col);   // the residual path's gradient
        *reinterpret_cast<f32x4*>(dx + (size_t)row * d + col) = o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {                       // dbeta += dy, dgamma += dy * xhat (a product, then a sum: colreduce1's)
          cs[i][e] += dyv[i][e];
          csx[i][e] += dyv[i][e] * v[i][e];
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const int col = (lane + 64 * i) * 4;
    if (col < d) {
      *reinterpret_cast<f32x4*>(red + (size_t)(wave * 2 + 0) * d + col) = cs[i];
      *reinterpret_cast<f32x4*>(red + (size_t)(wave * 2 + 1) * d + col) = csx[i];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * d; c += 256) {           // c = j * d + column: exactly the partial row's layout
    float t = red[c];
#pragma unroll
    for (int w = 1; w < 4; ++w) t += red[(size_t)w * 2 * d + c];
    part[(size_t)blockIdx.x * 2 * d + c] = t;
  }
}

inline unsigned nblk(size_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

// ------------------------------------------------------------------------------------------ launchers
hipError_t launch_transpose2d(const float* x, float* y, int R, int C, int Rp, hipStream_t s) {
  hipLaunchKernelGGL(transpose2d_kernel, dim3((Rp + 31) / 32, (C + 31) / 32), dim3(256), 0, s, x, y, R, C, Rp);
  return hipGetLastError();
}
hipError_t launch_transpose_many(const TransposeDesc* table_dev, int n, int max_rp, int max_c, hipStream_t s) {
  if (n <= 0 || n > 65535 || max_rp <= 0 || max_c <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(transpose_many_kernel, dim3((max_rp + 31) / 32, (max_c + 31) / 32, n), dim3(256), 0, s, table_dev);
  return hipGetLastError();
}
hipError_t launch_im2col1d(const float* x, float* col, int M, int T, int C, hipStream_t s) {
  if (C & 3) return hipErrorInvalidValue;
  hipLaunchKernelGGL(im2col1d_kernel, dim3(nblk((size_t)M * 3 * (C / 4))), dim3(256), 0, s, x, col, M, T, C);
  return hipGetLastError();
}
hipError_t launch_col2im1d(const float* dcol, float* dx, int M, int T, int C, hipStream_t s) {
  if (C & 3) return hipErrorInvalidValue;
  hipLaunchKernelGGL(col2im1d_kernel, dim3(nblk((size_t)M * (C / 4))), dim3(256), 0, s, dcol, dx, M, T, C);
  return hipGetLastError();
}
namespace {
unsigned magic32(unsigned d) { return d > 1 ? (unsigned)((1ULL << 32) / d) + 1u : 0u; }
// mdiv(n, magic32(d)) is exact for every n < nmax when nmax * d < 2^32
bool magic_ok(unsigned long long nmax, unsigned long long d) { return d >= 1 && nmax * d < (1ULL << 32); }
}  // namespace

hipError_t launch_im2col2d(const float* x, float* col, int I, int H, int W, int C, int Ho, int Wo, int Kp,
                           hipStream_t s) {
  if (I <= 0 || H <= 0 || W <= 0 || C <= 0 || Ho <= 0 || Wo <= 0 || Kp < 9 * C) return hipErrorInvalidValue;
  {
    const int V = (!(C & 3) && !(Kp & 3) && !((uintptr_t)x & 15) && !((uintptr_t)col & 15)) ? 4 : 1;
    const unsigned long long rows = (unsigned long long)I * Ho * Wo, total = rows * (Kp / V);
    if (magic_ok(total, Kp / V) && magic_ok(Kp, C) && magic_ok(rows, (unsigned long long)Ho * Wo) && magic_ok((unsigned long long)Ho * Wo, Wo)) {
      const unsigned mk = magic32(Kp / V), mc = magic32(C), mhw = magic32(Ho * Wo), mwo = magic32(Wo);
      if (V == 4) hipLaunchKernelGGL((im2col2d_fast_kernel<4>), dim3(nblk(total)), dim3(256), 0, s, x, col, (unsigned)total, H, W, C, Ho, Wo, Kp, mk, mc, mhw, mwo);
      else hipLaunchKernelGGL((im2col2d_fast_kernel<1>), dim3(nblk(total)), dim3(256), 0, s, x, col, (unsigned)total, H, W, C, Ho, Wo, Kp, mk, mc, mhw, mwo);
      return hipGetLastError();
    }
  }
  hipLaunchKernelGGL(im2col2d_kernel, dim3(nblk((size_t)I * Ho * Wo * Kp)), dim3(256), 0, s, x, col, I, H, W, C, Ho, Wo, Kp);
  return hipGetLastError();
}
hipError_t launch_col2im2d(const float* dcol, float* dx, int I, int H, int W, int C, int Ho, int Wo, int Kp,
                           hipStream_t s) {
  if (I <= 0 || H <= 0 || W <= 0 || C <= 0 || Ho <= 0 || Wo <= 0 || Kp < 9 * C) return hipErrorInvalidValue;
  {
    const int V = (!(C & 3) && !(Kp & 3) && !((uintptr_t)dcol & 15) && !((uintptr_t)dx & 15)) ? 4 : 1;
    const unsigned long long pixels = (unsigned long long)I * H * W, total = pixels * (C / V);
    if (magic_ok(total, C / V) && magic_ok(pixels, W) && magic_ok((unsigned long long)I * H, H)) {
      const unsigned mcv = magic32(C / V), mw = magic32(W), mh = magic32(H);
      if (V == 4) hipLaunchKernelGGL((col2im2d_fast_kernel<4>), dim3(nblk(total)), dim3(256), 0, s, dcol, dx, (unsigned)total, H, W, C, Ho, Wo, Kp, mcv, mw, mh);
      else hipLaunchKernelGGL((col2im2d_fast_kernel<1>), dim3(nblk(total)), dim3(256), 0, s, dcol, dx, (unsigned)total, H, W, C, Ho, Wo, Kp, mcv, mw, mh);
      return hipGetLastError();
    }
  }
  hipLaunchKernelGGL(col2im2d_kernel, dim3(nblk((size_t)I * H * W * C)), dim3(256), 0, s, dcol, dx, I, H, W, C, Ho, Wo, Kp);
  return hipGetLastError();
}
// out0[c] = sum_r a[r][c];  if b: out1[c] = sum_r a[r][c]*b[r][c].  `part` needs colreduce_part_floats(M, C) floats.
hipError_t launch_colreduce(const float* a, const float* b, float* part, float* out0, float* out1, int M, int C,
                            float scale, hipStream_t s) {
  const RedPlan pl = red_plan(M, C);
  const int cw = 1 << pl.cwl;
  const dim3 grid(pl.nb, (C + cw - 1) / cw);
  if (b) hipLaunchKernelGGL((colreduce1_kernel<1>), grid, dim3(256), 0, s, a, b, part, M, C, pl.rows, pl.cwl);
  else hipLaunchKernelGGL((colreduce1_kernel<0>), grid, dim3(256), 0, s, a, b, part, M, C, pl.rows, pl.cwl);
  const int nj = b ? 2 : 1;
  hipLaunchKernelGGL(colreduce2_kernel, dim3((nj * C + 31) / 32), dim3(256), 0, s, part, out0, out1, pl.nb, nj, C, scale);
  return hipGetLastError();
}
// biased variance: out[c] = sum_r (x - mean)^2 / M
hipError_t launch_bn_var(const float* x, const float* mean, float* part, float* out, int M, int C, hipStream_t s) {
  const RedPlan pl = red_plan(M, C);
  const int cw = 1 << pl.cwl;
  hipLaunchKernelGGL((colreduce1_kernel<2>), dim3(pl.nb, (C + cw - 1) / cw), dim3(256), 0, s, x, mean, part, M, C,
                     pl.rows, pl.cwl);
  hipLaunchKernelGGL(colreduce2_kernel, dim3((C + 31) / 32), dim3(256), 0, s, part, out, (float*)nullptr, pl.nb, 1, C,
                     1.0f / (float)M);
  return hipGetLastError();
}
int colreduce_part_floats(int M, int C) { return red_plan(M, C).nb * 2 * C; }

// dx (+ dres) and dgamma / dbeta in two launches: the fused backward above, then the second reduction stage.  `part` holds
// colreduce_part_floats(M, d) floats (the strips are never more than the generic plan's workgroups).
hipError_t launch_layernorm_bwd_affine(const float* dy, const float* x, const float* gamma, const float* dres, float* dx,
                                       float* dgamma, float* dbeta, float* part, int M, int d, float eps, hipStream_t s) {
  if (M <= 0 || d <= 0 || (d & 3) || d > 2048) return hipErrorInvalidValue;
  const RedPlan pl = red_plan(M, d);
  const int rpw = (pl.rows + 3) / 4;                         // rows per wave: 4 rpw >= pl.rows, so strips <= pl.nb
  const int strips = (M + 4 * rpw - 1) / (4 * rpw);
  const size_t lds = (size_t)8 * d * sizeof(float);
  const int vec = (d + 255) / 256;
  const dim3 grid(strips), block(256);
  if (vec <= 1) hipLaunchKernelGGL((layernorm_bwd_fused_kernel<1>), grid, block, lds, s, dy, x, gamma, dres, dx, part, M, d, rpw, eps);
  else if (vec <= 2) hipLaunchKernelGGL((layernorm_bwd_fused_kernel<2>), grid, block, lds, s, dy, x, gamma, dres, dx, part, M, d, rpw, eps);
  else if (vec <= 4) hipLaunchKernelGGL((layernorm_bwd_fused_kernel<4>), grid, block, lds, s, dy, x, gamma, dres, dx, part, M, d, rpw, eps);
  else hipLaunchKernelGGL((layernorm_bwd_fused_kernel<8>), grid, block, lds, s, dy, x, gamma, dres, dx, part, M, d, rpw, eps);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(colreduce2_kernel, dim3((2 * d + 31) / 32), dim3(256), 0, s, part, dbeta, dgamma, strips, 2, d, 1.0f);
  return hipGetLastError();
}
hipError_t launch_sum_slices(const float* part, float* out, int S, size_t n, hipStream_t s) {
  if (n & 3) return hipErrorInvalidValue;
  hipLaunchKernelGGL(sum_slices_kernel, dim3(nblk(n / 4)), dim3(256), 0, s, part, out, S, n / 4);
  return hipGetLastError();
}
hipError_t launch_bn_apply(const float* x, const float* mean, const float* var, const float* gamma, const float* beta,
                           float* xhat, float* y, size_t n, int C, int relu, float eps, hipStream_t s) {
  hipLaunchKernelGGL(bn_apply_kernel, dim3(nblk(n)), dim3(256), 0, s, x, mean, var, gamma, beta, xhat, y, n, C, relu, eps);
  return hipGetLastError();
}
hipError_t launch_bn_bwd(const float* dyr, const float* xhat, const float* gamma, const float* var,
                         const float* sum_dy, const float* sum_dyx, float* dx, size_t n, int C, float invM, float eps,
                         hipStream_t s) {
  hipLaunchKernelGGL(bn_bwd_kernel, dim3(nblk(n)), dim3(256), 0, s, dyr, xhat, gamma, var, sum_dy, sum_dyx, dx, n, C, invM, eps);
  return hipGetLastError();
}
hipError_t launch_bn_running(float* rmean, float* rvar, const float* mean, const float* var, int C, float momentum,
                             int M, hipStream_t s) {
  const float unbias = M > 1 ? (float)M / (float)(M - 1) : 1.0f;
  hipLaunchKernelGGL(bn_running_kernel, dim3((C + 255) / 256), dim3(256), 0, s, rmean, rvar, mean, var, C, momentum, unbias);
  return hipGetLastError();
}
namespace {
inline bool vec4_ok(size_t n, const void* a, const void* b = nullptr, const void* c = nullptr) {
  auto al = [](const void* q) { return q == nullptr || (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  return (n & 3) == 0 && al(a) && al(b) && al(c);
}
}  // namespace
hipError_t launch_act_fwd(const float* x, float* y, size_t n, int act, hipStream_t s) {
  if (vec4_ok(n, x, y)) hipLaunchKernelGGL((act_fwd_kernel<4>), dim3(nblk(n / 4)), dim3(256), 0, s, x, y, n / 4, act);
  else hipLaunchKernelGGL((act_fwd_kernel<1>), dim3(nblk(n)), dim3(256), 0, s, x, y, n, act);
  return hipGetLastError();
}
hipError_t launch_act_bwd(const float* dy, const float* aux, float* dx, size_t n, int act, hipStream_t s) {
  if (vec4_ok(n, dy, aux, dx))
    hipLaunchKernelGGL((act_bwd_kernel<4>), dim3(nblk(n / 4)), dim3(256), 0, s, dy, aux, dx, n / 4, act);
  else hipLaunchKernelGGL((act_bwd_kernel<1>), dim3(nblk(n)), dim3(256), 0, s, dy, aux, dx, n, act);
  return hipGetLastError();
}
hipError_t launch_relu_dropout_bwd(const float* dy, const float* y, float* dx, size_t n, float p, hipStream_t s) {
  if (vec4_ok(n, dy, y, dx))
    hipLaunchKernelGGL((relu_dropout_bwd_kernel<4>), dim3(nblk(n / 4)), dim3(256), 0, s, dy, y, dx, n / 4, p);
  else hipLaunchKernelGGL((relu_dropout_bwd_kernel<1>), dim3(nblk(n)), dim3(256), 0, s, dy, y, dx, n, p);
  return hipGetLastError();
}
hipError_t launch_mul_mixed(const float* a, const float* xt, float* out, size_t M, int S, int F, int ldx,
                            hipStream_t s) {
  hipLaunchKernelGGL(mul_mixed_kernel, dim3(nblk(M * S * F)), dim3(256), 0, s, a, xt, out, M, S, F, ldx);
  return hipGetLastError();
}
hipError_t launch_dropout(const float* x, float* y, size_t n, float p, unsigned long long seed, hipStream_t s) {
  if (vec4_ok(n, x, y)) hipLaunchKernelGGL((dropout_kernel<4>), dim3(nblk(n / 4)), dim3(256), 0, s, x, y, n / 4, p, seed);
  else hipLaunchKernelGGL((dropout_kernel<1>), dim3(nblk(n)), dim3(256), 0, s, x, y, n, p, seed);
  return hipGetLastError();
}
hipError_t launch_dropout_add(const float* x, const float* r, float* y, size_t n, float p, unsigned long long seed,
                              hipStream_t s) {
  if (vec4_ok(n, x, y, r)) hipLaunchKernelGGL((dropout_add_kernel<4>), dim3(nblk(n / 4)), dim3(256), 0, s, x, r, y, n / 4, p, seed);
  else hipLaunchKernelGGL((dropout_add_kernel<1>), dim3(nblk(n)), dim3(256), 0, s, x, r, y, n, p, seed);
  return hipGetLastError();
}
hipError_t launch_add_rows(const float* x, const float* r, float* y, size_t M, int C, int period, hipStream_t s) {
  if ((C & 3) == 0 && vec4_ok(4, x, r, y))
    hipLaunchKernelGGL((add_rows_kernel<4>), dim3(nblk(M * C / 4)), dim3(256), 0, s, x, r, y, M * C / 4, C / 4, period);
  else hipLaunchKernelGGL((add_rows_kernel<1>), dim3(nblk(M * C)), dim3(256), 0, s, x, r, y, M * C, C, period);
  return hipGetLastError();
}
hipError_t launch_avgpool_bwd(const float* dy, float* dx, int M, int P, int C, hipStream_t s) {
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(nblk((size_t)M * P * C)), dim3(256), 0, s, dy, dx, M, P, C);
  return hipGetLastError();
}
hipError_t launch_interp_bwd(const float* dy, float* dx, int B, int N, int T, int d, hipStream_t s) {
  if (d & 3) return hipErrorInvalidValue;
  hipLaunchKernelGGL(interp_bwd_kernel, dim3(nblk((size_t)B * N * (d / 4))), dim3(256), 0, s, dy, dx, B, N, T, d,
                     (float)N / (float)T);
  return hipGetLastError();
}
hipError_t launch_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* dres, float* dx,
                                float* xhat, int M, int d, float eps, hipStream_t s) {
  if (M <= 0 || d <= 0 || (d & 3) || d > 2048) return hipErrorInvalidValue;
  const dim3 grid((M + 3) / 4), block(256);
  const int vec = (d + 255) / 256;
  if (vec <= 1) hipLaunchKernelGGL((layernorm_bwd_kernel<1>), grid, block, 0, s, dy, x, gamma, dres, dx, xhat, M, d, eps);
  else if (vec <= 2) hipLaunchKernelGGL((layernorm_bwd_kernel<2>), grid, block, 0, s, dy, x, gamma, dres, dx, xhat, M, d, eps);
  else if (vec <= 4) hipLaunchKernelGGL((layernorm_bwd_kernel<4>), grid, block, 0, s, dy, x, gamma, dres, dx, xhat, M, d, eps);
  else hipLaunchKernelGGL((layernorm_bwd_kernel<8>), grid, block, 0, s, dy, x, gamma, dres, dx, xhat, M, d, eps);
  return hipGetLastError();
}
