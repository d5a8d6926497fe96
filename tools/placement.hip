// Developer tool (GPU box): where does the dispatcher put the workgroups of a 3-per-CU grid?
//   hipcc --offload-arch=gfx950 -O2 tools/placement.hip -o /tmp/placement && /tmp/placement
// Each workgroup (256 threads, 48 KB of LDS so that three fit a CU) records HW_REG_HW_ID, XCC_ID and its start tick.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>
__global__ __launch_bounds__(256) void census(unsigned* out, int spin) {
  __shared__ float pad[12288];
  pad[threadIdx.x] = threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0) {
    out[4 * blockIdx.x + 0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID
    out[4 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
    out[4 * blockIdx.x + 2] = (unsigned)__builtin_amdgcn_s_memrealtime();
  }
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
  if (pad[(threadIdx.x + 1) & 255] < 0) out[0] = 1;
}
int main() {
  const int n = 2048;
  unsigned* d; hipMalloc(&d, n * 16);
  census<<<n, 256>>>(d, 2000);   // 20 us per workgroup
  hipDeviceSynchronize();
  std::vector<unsigned> h(n * 4); hipMemcpy(h.data(), d, n * 16, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> cu;   // (xcc, se, sh, cu) -> block ids in arrival order
  unsigned tmin = ~0u; for (int b = 0; b < n; ++b) tmin = std::min(tmin, h[4 * b + 2]);
  for (int b = 0; b < n; ++b) {
    unsigned hw = h[4 * b], xcc = h[4 * b + 1] & 0xf;
    unsigned key = (xcc << 16) | (hw & 0xff00);        // cu_id[11:8], sh_id[12], se_id[15:13]
    cu[key].push_back(b);
  }
  printf("%zu distinct (xcc, se, sh, cu) keys for %d workgroups\n", cu.size(), n);
  int shown = 0;
  for (auto& kv : cu) {
    if (shown++ >= 12) break;
    printf("xcc %u se %u cu %2u :", kv.first >> 16, (kv.first >> 13) & 7, (kv.first >> 8) & 0xf);
    for (int b : kv.second) printf(" %4d(t+%u)", b, h[4 * b + 2] - tmin);
    printf("\n");
  }
  return 0;
}
