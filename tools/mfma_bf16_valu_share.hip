// Developer microbenchmark (GPU box): do bf16 MFMAs (v_mfma_f32_16x16x32_bf16) and plain VALU instructions overlap on a SIMD of gfx950 --
// (a) issued by ANOTHER wave of the SIMD, (b) interleaved in the SAME wave?  (Round 3 measured 0 % for the fp32 MFMA.)
// `hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16_valu_share.hip -o /tmp/mbv && /tmp/mbv`
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: waves 0-3 MFMA loop, waves 4-7 VALU loop (iteration counts 0 = skip).  MODE 1: every wave runs MFMAs with VPM VALU
// instructions interleaved behind each MFMA in program order.
template <int MODE, int VPM>
__global__ __launch_bounds__(512) void both(float* out, int mfma_iters, int valu_iters, float a0, float k) {
  const int wave = threadIdx.x >> 6;
  u32x4 au = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  const bf16x8 a = __builtin_bit_cast(bf16x8, au), b = a;
  if (MODE == 1 || wave < 4) {
    f32x4 acc[8];
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; x[i] = a0 + i + threadIdx.x * 1e-3f; }
    if (MODE == 1 && wave >= 4) mfma_iters = 0;
    for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
          if (MODE == 1) {
#pragma unroll
            for (int v = 0; v < VPM; ++v) x[(i + v) & 7] = __builtin_fmaf(x[(i + v) & 7], k, 1.0f);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + x[i];
    if (s == 12345.678f) out[0] = s;
  } else {
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = a0 + i + threadIdx.x * 1e-3f;
    for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], k, 1.0f);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 12345.678f) out[1] = s;
  }
}

template <int MODE, int VPM>
float time_launch(float* out, int mi, int vi) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  both<MODE, VPM><<<256, 512>>>(out, mi ? 10 : 0, vi ? 10 : 0, 1.0f, 0.999f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  both<MODE, VPM><<<256, 512>>>(out, mi, vi, 1.0f, 0.999f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* out; hipMalloc(&out, 8);
  const int mi = 40000;                     // 32 MFMAs per iteration, 16 cycles each = 512 matrix cycles
  const float tm = time_launch<0, 0>(out, mi, 0);
  int vi = 20000;
  float tv = time_launch<0, 0>(out, 0, vi);
  vi = (int)(vi * (0.5f * tm / tv));
  tv = time_launch<0, 0>(out, 0, vi);
  const float tb = time_launch<0, 0>(out, mi, vi);
  printf("(a) other wave of the SIMD: bf16 MFMA only %7.3f ms (%6.1f TFLOP/s) | v_fma_f32 only %7.3f ms (%4.1f cycles per instruction) | both %7.3f ms -> overlap %4.0f %%\n",
         tm, 256.0 * 4 * mi * 32 * 16384.0 / tm / 1e9, tv, tv * 2.4e6 / ((double)vi * 32), tb, 100.0 * (tm + tv - tb) / tv);
  const float t0 = time_launch<1, 0>(out, mi, 0);
  const float t1 = time_launch<1, 1>(out, mi, 0), t2 = time_launch<1, 2>(out, mi, 0), t3 = time_launch<1, 3>(out, mi, 0), t4 = time_launch<1, 4>(out, mi, 0);
  printf("(b) same wave, VALU behind each MFMA in program order: 0 / 1 / 2 / 3 / 4 v_fma_f32 per MFMA: %7.3f / %7.3f / %7.3f / %7.3f / %7.3f ms"
         "  (16 matrix cycles per MFMA, 4 per VALU instruction: fully hidden up to 3-4)\n", t0, t1, t2, t3, t4);
  return 0;
}
