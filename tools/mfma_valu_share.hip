// Developer microbenchmark (GPU box): do fp32 MFMAs and plain VALU instructions of ANOTHER wave on the same SIMD overlap?
// `hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_share.hip -o /tmp/mvs && /tmp/mvs`
// One workgroup of 512 threads per CU: waves 0-3 (one per SIMD) run a loop of independent v_mfma_f32_16x16x4_f32, waves 4-7
// (one per SIMD, beside them) run a loop of independent VALU instructions of one kind, or nothing.  Reported: the time of the
// launch against the MFMA-only and the VALU-only launch.  If the two kinds of work overlap, t(both) ~ max(t_mfma, t_valu);
// if they share an execution resource, t(both) ~ t_mfma + t_valu.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// KIND: 0 v_fma_f32, 1 v_pk_mul_f32, 2 v_exp_f32, 3 v_max_f32 (DPP row_shr), 4 v_cndmask / integer (v_add_u32), 5 v_mov (dpp)
template <int KIND>
__device__ __forceinline__ void valu_body(float (&x)[8], float k) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (KIND == 0) x[i] = __builtin_fmaf(x[i], k, 1.0f);
    else if (KIND == 1) { }
    else if (KIND == 2) x[i] = __builtin_amdgcn_exp2f(x[i]);
    else if (KIND == 3) x[i] = fmaxf(x[i], __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x[i]), 0x111, 0xf, 0xf, false)));
    else if (KIND == 4) x[i] = __builtin_bit_cast(float, __builtin_bit_cast(int, x[i]) + 12345);
  }
  if (KIND == 1) {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
      f32x2 v = {x[i], x[i + 1]};
      v = v * f32x2{k, k};
      x[i] = v[0]; x[i + 1] = v[1];
    }
  }
}

template <int KIND>
__global__ __launch_bounds__(512) void both(float* out, int mfma_iters, int valu_iters, float a0, float k) {
  const int wave = threadIdx.x >> 6;
  if (wave < 4) {
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x * 1e-6f, b = a0;
    for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
  } else {
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = a0 + i + threadIdx.x * 1e-3f;
    for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r) valu_body<KIND>(x, k);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 12345.678f) out[1] = s;
  }
}

template <int KIND>
float time_launch(float* out, int mi, int vi) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  both<KIND><<<256, 512>>>(out, mi ? 10 : 0, vi ? 10 : 0, 1.0f, 0.999f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  both<KIND><<<256, 512>>>(out, mi, vi, 1.0f, 0.999f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

template <int KIND>
void run(const char* name, int per_iter_instrs) {
  float* out; hipMalloc(&out, 8);
  const int mi = 20000;                     // 32 MFMAs per iteration = 1024 matrix cycles
  const float tm = time_launch<KIND>(out, mi, 0);
  // VALU iterations chosen so that the VALU-only launch takes about half the MFMA-only launch
  int vi = 20000;
  float tv = time_launch<KIND>(out, 0, vi);
  vi = (int)(vi * (0.5f * tm / tv));
  tv = time_launch<KIND>(out, 0, vi);
  const float tb = time_launch<KIND>(out, mi, vi);
  printf("%-28s MFMA only %7.3f ms (%5.1f TFLOP/s) | VALU only %7.3f ms (%4.1f cycles per instruction at 2.4 GHz) | both %7.3f ms"
         "  -> overlap %4.0f %%\n", name, tm, 256.0 * 4 * mi * 32 * 2048.0 / tm / 1e9, tv,
         tv * 2.4e6 / ((double)vi * per_iter_instrs), tb, 100.0 * (tm + tv - tb) / tv);
  hipFree(out);
}

int main() {
  run<0>("v_fma_f32", 32);
  run<1>("v_pk_mul_f32", 16);
  run<2>("v_exp_f32", 32);
  run<3>("v_max_f32 + dpp row_shr", 32);
  run<4>("v_add_u32", 32);
  return 0;
}
