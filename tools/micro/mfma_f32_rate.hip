// Microbenchmark: sustained v_mfma_f32_32x32x2_f32 rate, 1 wave/SIMD (512-thread... 256-thread WG per CU),
// for the two accumulator orders used by nerf_mlp_f32 (chain of 4 on one accumulator vs round-robin over 8).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k(float* out, int iters, float seed) {
  f32x16 acc[8];
  for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = seed * (j + r);
  float a[4], b[4];
  for (int q = 0; q < 4; ++q) { a[q] = seed + threadIdx.x * 0.001f + q; b[q] = seed - q * 0.5f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[q], acc[j], 0, 0, 0);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[q], acc[j], 0, 0, 0);
      }
    }
  }
  float s = 0;
  for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 256 * 256 * 4 * 8);
  const int iters = 20000;
  for (int mode = 0; mode < 2; ++mode) for (int rep = 0; rep < 3; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, iters, 1e-3f);
    else hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, iters, 1e-3f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = 256.0 * 4 * iters * 128.0 * 4096.0;
    printf("mode %d (%s): %.3f ms  %.1f TFLOP/s  (%.2f cycles/MFMA @2.4GHz)\n", mode, mode ? "round-robin 8 acc" : "chain of 4",
           ms, flop / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 128.0));
  }
  return 0;
}
