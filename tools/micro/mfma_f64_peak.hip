// Microbenchmark: sustained v_mfma_f64_16x16x4_f64 rate on gfx950 (registers only, independent accumulators),
// the roof the fp64 invert / eigensolver kernels are priced against ("measure it").
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_f64_peak.hip -o tools/micro/mfma_f64_peak && ./tools/micro/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256) k64(double* out, int iters) {
  f64x4 c[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) c[i] = f64x4{0, 0, 0, 0};
  double a = threadIdx.x * 0.5, b = 1.0 + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k32(float* out, int iters) {
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  float a = threadIdx.x * 0.5f, b = 1.0f + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  void* out;
  hipMalloc(&out, 8 * 256 * 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  for (int wg = 1; wg <= 4; wg *= 2) {
    const int grid = 256 * wg, iters = 40000;
    hipLaunchKernelGGL(k64<8>, dim3(grid), dim3(256), 0, 0, (double*)out, 100);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k64<8>, dim3(grid), dim3(256), 0, 0, (double*)out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      // one 16x16x4 MFMA = 2 * 16 * 16 * 4 = 2048 flops per wave
      const double flops = (double)grid * 4 * iters * 8 * 2048.0;
      printf("f64 16x16x4, %d wave(s)/SIMD, 8 independent accumulators: %.3f ms  %.1f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n",
             wg, ms, flops / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)wg * iters * 8));
    }
  }
  for (int wg = 1; wg <= 2; ++wg) {
    const int grid = 256 * wg, iters = 40000;
    hipLaunchKernelGGL(k32, dim3(grid), dim3(256), 0, 0, (float*)out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k32, dim3(grid), dim3(256), 0, 0, (float*)out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 4 * 4096.0;
    printf("f32 32x32x2, %d wave(s)/SIMD: %.3f ms  %.1f TFLOP/s\n", wg, ms, flops / ms / 1e9);
  }
  return 0;
}
