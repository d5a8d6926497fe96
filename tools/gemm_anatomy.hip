// Developer microbenchmark (GPU box): which ingredient of the GEMM main loop costs what?  The loop of the 64x64x32
// 4-wave tile (per chunk and wave: 32 MFMAs, 8 ds_read_b128, 4 ds_write_b128, 4 global float4 loads, 1 barrier) is
// rebuilt ingredient by ingredient on dummy data; everything is launched with 4 workgroups per CU like the real one.
//   hipcc --offload-arch=gfx950 -O3 tools/gemm_anatomy.hip -o /tmp/gemm_anatomy && /tmp/gemm_anatomy
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool LDSREAD, bool BARRIER, bool LDSWRITE, bool GLOAD>
__global__ __launch_bounds__(256) void loop(const float* __restrict__ src, float* out, int chunks, size_t stride) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 128 * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * 128 * 32; i += 256) lds[i] = 1.0f + (i & 7) * 1e-3f;
  __syncthreads();
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fq = lane >> 4;
  const int arow = (wave >> 1) * 32 + fr, brow = 64 + (wave & 1) * 32 + fr;
  f32x4 fa[2], fb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { fa[i] = f32x4{1.f, 1.1f, 1.2f, 1.3f}; fb[i] = f32x4{0.9f, 0.8f, 0.7f, 0.6f}; }
  const float* g = src + (size_t)(blockIdx.x & 15) * stride + tid * 4;   // 16 x 256 KB: stays L2-resident like shared A/W tiles
  f32x4 r[2][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) r[0][q] = r[1][q] = f32x4{1.f, 1.f, 1.f, 1.f};
  for (int kc = 0; kc < chunks; ++kc) {
    const int buf = kc & 1;
    if (GLOAD) {
#pragma unroll
      for (int q = 0; q < 4; ++q) r[buf][q] = *reinterpret_cast<const f32x4*>(g + (size_t)(kc & 15) * 4096 + q * 1024);
    }
    const float* a = lds + buf * 128 * 32;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (LDSREAD) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          fa[i] = *reinterpret_cast<const f32x4*>(a + (arow + 16 * i) * 32 + (((4 * s + fq) ^ ((arow >> 1) & 7)) << 2));
          fb[i] = *reinterpret_cast<const f32x4*>(a + (brow + 16 * i) * 32 + (((4 * s + fq) ^ ((brow >> 1) & 7)) << 2));
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][c], fb[j][c], acc[i][j], 0, 0, 0);
    }
    if (LDSWRITE) {
      float* w = lds + (buf ^ 1) * 128 * 32;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = (tid >> 3) + 32 * q;
        *reinterpret_cast<f32x4*>(w + row * 32 + (((tid & 7) ^ ((row >> 1) & 7)) << 2)) = r[buf ^ 1][q];
      }
    }
    if (BARRIER) __syncthreads();
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][3];
  if (s == 12345.678f) out[0] = s;
}

// Same full loop, but unrolled by two so both register sets and LDS buffers are static and every load is
// unconditional: the compiler can then wait for "all but the 4 newest" loads (vmcnt(4)) before the ds_writes.
__global__ __launch_bounds__(256) void loop_static(const float* __restrict__ src, float* out, int chunks, size_t stride) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 128 * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * 128 * 32; i += 256) lds[i] = 1.0f + (i & 7) * 1e-3f;
  __syncthreads();
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fq = lane >> 4;
  const int arow = (wave >> 1) * 32 + fr, brow = 64 + (wave & 1) * 32 + fr;
  const float* g = src + (size_t)(blockIdx.x & 15) * stride + tid * 4;
  f32x4 r0[4], r1[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) r1[q] = *reinterpret_cast<const f32x4*>(g + q * 1024);
  auto compute = [&](const float* a) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f32x4 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[i] = *reinterpret_cast<const f32x4*>(a + (arow + 16 * i) * 32 + (((4 * s + fq) ^ ((arow >> 1) & 7)) << 2));
        fb[i] = *reinterpret_cast<const f32x4*>(a + (brow + 16 * i) * 32 + (((4 * s + fq) ^ ((brow >> 1) & 7)) << 2));
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][c], fb[j][c], acc[i][j], 0, 0, 0);
    }
  };
  auto stage = [&](float* w, const f32x4 (&r)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = (tid >> 3) + 32 * q;
      *reinterpret_cast<f32x4*>(w + row * 32 + (((tid & 7) ^ ((row >> 1) & 7)) << 2)) = r[q];
    }
  };
  for (int kc = 0; kc < chunks; kc += 2) {
#pragma unroll
    for (int q = 0; q < 4; ++q) r0[q] = *reinterpret_cast<const f32x4*>(g + (size_t)(kc & 15) * 4096 + q * 1024);
    compute(lds);
    stage(lds + 128 * 32, r1);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) r1[q] = *reinterpret_cast<const f32x4*>(g + (size_t)((kc + 1) & 15) * 4096 + q * 1024);
    compute(lds + 128 * 32);
    stage(lds, r0);
    __syncthreads();
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][3];
  if (s == 12345.678f) out[0] = s;
}

template <bool A, bool B, bool C, bool D>
void run(const char* what, const float* src, float* out) {
  const int chunks = 2048, grid = 256 * 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  loop<A, B, C, D><<<grid, 256>>>(src, out, 64, 64 * 4096);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  loop<A, B, C, D><<<grid, 256>>>(src, out, chunks, 64 * 4096);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 * chunks * 32.0 * 2048.0;
  printf("%-58s %7.3f ms  %6.1f TFLOP/s\n", what, ms, flops / ms / 1e9);
}

int main() {
  float *src, *out;
  const size_t n = (size_t)1024 * 64 * 4096;
  hipMalloc(&src, n * 4); hipMemset(src, 0, n * 4); hipMalloc(&out, 4);
  run<false, false, false, false>("MFMA only (2x2 accumulators, 4 waves/SIMD)", src, out);
  run<true, false, false, false>("+ 8 ds_read_b128 per 32 MFMAs", src, out);
  run<true, true, false, false>("+ barrier per chunk", src, out);
  run<true, true, true, false>("+ 4 ds_write_b128 per chunk", src, out);
  run<true, true, true, true>("+ 4 global float4 loads per chunk (L2-resident stream)", src, out);
  run<false, true, false, false>("MFMA + barrier only", src, out);
  {
    const int chunks = 2048, grid = 256 * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    loop_static<<<grid, 256>>>(src, out, 64, 64 * 4096);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    loop_static<<<grid, 256>>>(src, out, chunks, 64 * 4096);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s %7.3f ms  %6.1f TFLOP/s\n", "full loop, static double buffers, unconditional loads", ms,
           (double)grid * 4 * chunks * 32.0 * 2048.0 / ms / 1e9);
  }
  return 0;
}
