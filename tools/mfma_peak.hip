// Developer microbenchmark (GPU box): sustained fp32 MFMA rate of the chip with nothing else going on --
// `hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak`
// NACC independent accumulators per wave, WAVES waves per SIMD (via blocks per CU), no memory traffic in the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;
}

template <int NACC>
void run(int blocks_per_cu) {
  float* out; hipMalloc(&out, 4);
  const int iters = 4000, grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  mfma_loop<NACC><<<grid, 256>>>(out, 100, 1.0f, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  mfma_loop<NACC><<<grid, 256>>>(out, iters, 1.0f, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 /*waves*/ * iters * 8.0 * NACC * 2048.0;
  printf("accumulators/wave %2d, waves/SIMD %d: %7.3f ms  %6.1f TFLOP/s  (=> %.2f GHz if the pipes never idle)\n", NACC,
         blocks_per_cu, ms, flops / ms / 1e9, flops / ms / 1e9 / (256 * 4 * 64.0) * 1e3 / 1e3);
  hipFree(out);
}

int main() {
  run<1>(1); run<2>(1); run<4>(1); run<4>(2); run<4>(4); run<8>(2); run<16>(1);
  return 0;
}
