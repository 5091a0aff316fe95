// What does v_mfma_f32_32x32x2_f32 sustain on this part?  Register-only loop (no memory): W waves per SIMD, each with A accumulators,
// back-to-back MFMAs; short (50 us) and long (20 ms) launches -- the long one shows the clock the matrix pipes hold under sustained load.
// The learner's kernels are priced against the 157 TF data-sheet figure; this is the ceiling a kernel can actually reach.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f32_peak mfma_f32_peak.hip && ./mfma_f32_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
template <int A>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a0, float b0) {
  v16f acc[A];
#pragma unroll
  for (int i = 0; i < A; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.0f;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < A; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < A; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 123.456f) out[0] = s;
}
template <int A>
static void run(int wg_per_cu, int iters, const char* tag) {
  float* d; hipMalloc(&d, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * wg_per_cu;
  hipLaunchKernelGGL(k<A>, dim3(grid), dim3(256), 0, 0, d, iters / 10 + 1, 1.0f, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<A>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0f, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)grid * 4 /*waves*/ * iters * A * 2.0 * 32 * 32 * 2;
  printf("%-28s accumulators %d, %d workgroups of 4 waves per CU, %8d iterations: %8.3f ms  %7.1f TFLOP/s\n", tag, A, wg_per_cu, iters, ms, flop / ms * 1e-9);
  hipFree(d);
}
int main() {
  run<4>(1, 2000, "short, 1 wave/SIMD");
  run<4>(2, 2000, "short, 2 waves/SIMD");
  run<2>(3, 2000, "short, 3 waves/SIMD (2 acc)");
  run<4>(1, 800000, "sustained, 1 wave/SIMD");
  run<4>(2, 400000, "sustained, 2 waves/SIMD");
  run<2>(3, 400000, "sustained, 3 waves/SIMD (2 acc)");
  run<1>(3, 800000, "sustained, 3 waves, 1 acc");
  return 0;
}
