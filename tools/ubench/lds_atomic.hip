// Microbenchmark: latency of dependent LDS operation chains on one wavefront (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long* out, int iters, int conflict) {
  __shared__ float s[1024];
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) s[i] = 1.0f;
  __syncthreads();
  unsigned long long t0, t1;
  float acc = 0.f;
  // (a) read -> fma -> write -> wait, chain through address dependency
  int idx = lane;
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    float v = s[idx];
    s[(idx + 64) & 1023] = v * 1.0001f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    idx = (idx + 64) & 1023;
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[0] = (t1 - t0) / iters;
  // (b) read, read -> mul -> ds_add_f32 (no return) -> wait  (distinct addresses per lane)
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    float a = s[idx], b = s[(idx + 128) & 1023];
    atomicAdd(&s[(idx + 256) & 1023], a * b * 1e-9f);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    idx = (idx + 64) & 1023;
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[1] = (t1 - t0) / iters;
  // (c) same with `conflict` lanes hitting one address
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    float a = s[idx], b = s[(idx + 128) & 1023];
    const int dst = lane < conflict ? 900 : ((idx + 256) & 1023);
    atomicAdd(&s[dst], a * b * 1e-9f);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    idx = (idx + 64) & 1023;
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[2] = (t1 - t0) / iters;
  // (d) plain read only chain
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    float v = s[idx];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    idx = (idx + 64 + (v > 2.f)) & 1023;
    acc += v;
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) { out[3] = (t1 - t0) / iters; out[7] = (unsigned long long)acc; }
  // (e) global load (L2-resident small table) latency chain
}
__global__ void g(unsigned long long* out, const int* tab, int iters) {
  int idx = threadIdx.x;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) idx = tab[idx];
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[4] = (t1 - t0) / iters; out[6] = idx; }
}
int main() {
  unsigned long long* d; hipMalloc(&d, 64);
  int* tab; hipMalloc(&tab, 4096 * 4);
  int h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (i + 64) % 4096;
  hipMemcpy(tab, h, sizeof(h), hipMemcpyHostToDevice);
  for (int blocks : {1, 2048}) for (int conflict : {1, 8, 32}) {
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, 2000, conflict);
    hipLaunchKernelGGL(g, dim3(blocks), dim3(64), 0, 0, d, tab, 2000);
    unsigned long long r[8]; hipMemcpy(r, d, 64, hipMemcpyDeviceToHost);
    printf("blocks %4d conflict %2d | rd->wr %llu | rd,rd->atomic %llu | atomic w/ conflict %llu | rd %llu | global(L2) %llu cycles\n", blocks, conflict, r[0], r[1], r[2], r[3], r[4]);
  }
  return 0;
}
