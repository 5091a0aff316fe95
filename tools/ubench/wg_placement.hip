// Where do the 2048 one-wave workgroups of the step kernel land?  Records (XCC, SE, CU, SIMD) per block for the same launch
// shape (64 threads, 20016 B of dynamic LDS => 8 blocks per CU).  hipcc --offload-arch=gfx950 -O2 -o wg_placement wg_placement.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ void probe(unsigned* out, int spin) {
  extern __shared__ float lds[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  float a = threadIdx.x;
  for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;     // stay resident long enough for all blocks to be placed
  lds[threadIdx.x] = a;
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc | (lds[1] > 1e30f ? 1u << 31 : 0u); }
}
int main() {
  const int N = 2048;
  unsigned* d; hipMalloc(&d, N * 8);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 20016);
  hipLaunchKernelGGL(probe, dim3(N), dim3(64), 20016, 0, d, 200000);
  std::vector<unsigned> h(2 * N); hipMemcpy(h.data(), d, N * 8, hipMemcpyDeviceToHost);
  std::map<unsigned long long, std::vector<int>> simd;
  for (int b = 0; b < N; ++b) {
    const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xF;
    const unsigned wave = hw & 15, sd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    if (b < 24 || (b % 256) == 0) printf("block %4d: xcc %u se %u sh %u cu %2u simd %u wave %u\n", b, xcc, se, sh, cu, sd, wave);
    simd[((unsigned long long)xcc << 32) | (se << 16) | (sh << 12) | (cu << 4) | sd].push_back(b);
  }
  printf("distinct SIMDs used: %zu\n", simd.size());
  std::map<int, int> diffhist; std::map<size_t, int> occ;
  for (auto& kv : simd) { occ[kv.second.size()]++; if (kv.second.size() == 2) diffhist[kv.second[1] - kv.second[0]]++; }
  for (auto& kv : occ) printf("SIMDs with %zu blocks: %d\n", kv.first, kv.second);
  int shown = 0;
  for (auto& kv : diffhist) if (shown++ < 12) printf("partner distance %d: %d pairs\n", kv.first, kv.second);
  return 0;
}
