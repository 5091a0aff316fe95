// Microbenchmark: throughput of INDEPENDENT LDS float atomics vs plain read-modify-write, one wave per block.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long* out, int iters) {
  __shared__ float s[4096];
  const int lane = threadIdx.x;
  for (int i = lane; i < 4096; i += 64) s[i] = 1.0f;
  __syncthreads();
  unsigned long long t0, t1;
  // (a) 16 independent atomics per iteration, distinct addresses, one wait at the end of the batch
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) atomicAdd(&s[(lane + 64 * u + i) & 4095], 1e-9f);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[0] = (t1 - t0) / (iters * 16);
  // (b) 16 independent plain RMW (read all, add, write all)
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = s[(lane + 64 * u + i) & 4095];
#pragma unroll
    for (int u = 0; u < 16; ++u) s[(lane + 64 * u + i) & 4095] = v[u] + 1e-9f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[1] = (t1 - t0) / (iters * 16);
  // (c) 16 independent atomics, 8 lanes share each address
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) atomicAdd(&s[((lane >> 3) + 64 * u + i) & 4095], 1e-9f);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[2] = (t1 - t0) / (iters * 16);
}
int main() {
  unsigned long long* d; (void)hipMalloc(&d, 64);
  for (int blocks : {1, 1024, 2048}) {
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, 500);
    unsigned long long r[8]; (void)hipMemcpy(r, d, 64, hipMemcpyDeviceToHost);
    printf("blocks %4d | independent atomic %llu cyc/instr | plain RMW %llu cyc/(rd+wr) | atomic 8-way shared %llu cyc/instr\n", blocks, r[0], r[1], r[2]);
  }
  return 0;
}
