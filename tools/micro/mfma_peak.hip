// Microbenchmark: sustained v_mfma_f32_32x32x2_f32 rate under the conditions of the SYRK inner loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void __launch_bounds__(256, 2) k(float* out, int iters, int stride) {
  __shared__ float lds[16384];
  const int tid = threadIdx.x;
  for (int i = tid; i < 16384; i += 256) lds[i] = (float)(i & 7) * 0.001f;
  __syncthreads();
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  float a0 = tid * 0.5f, a1 = tid * 0.25f, b0 = 1.0f + tid, b1 = 2.0f - tid;
  int addr = (tid * 37) & 8191;
  for (int it = 0; it < iters; ++it) {
    float x0 = a0, x1 = a1, y0 = b0, y1 = b1, x2 = a0, x3 = a1, y2 = b0, y3 = b1;
    if (MODE >= 1) {
      x0 = lds[addr]; x1 = lds[addr + 1]; y0 = lds[addr + 2]; y1 = lds[addr + 3];
      x2 = lds[addr + 64]; x3 = lds[addr + 65]; y2 = lds[addr + 66]; y3 = lds[addr + 67];
      addr = (addr + stride) & 8191;
    }
    if (MODE >= 2) {
#pragma unroll
      for (int v = 0; v < 24; ++v) addr = (addr * 3 + v) & 8191;
    }
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y0, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y1, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y0, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y1, c3, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x2, y2, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x2, y3, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x3, y2, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x3, y3, c3, 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void run(const char* name, int wg_per_cu, int stride) {
  float* out;
  hipMalloc(&out, 4 * 256 * 1024 * 4);
  const int iters = 20000, grid = 256 * wg_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<grid, 256>>>(out, 100, stride);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<grid, 256>>>(out, iters, stride);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)grid * 4 * iters * 8 * 4096.0;
  printf("%-40s wg/cu=%d: %.3f ms  %.1f TFLOP/s\n", name, wg_per_cu, ms, flops / ms / 1e9);
  hipFree(out);
}

int main() {
  run<0>("regs only", 1, 0);
  run<0>("regs only", 2, 0);
  run<1>("8 ds_read_b32 / 8 mfma (no conflicts)", 1, 128);
  run<1>("8 ds_read_b32 / 8 mfma (no conflicts)", 2, 128);
  run<2>("+24 dependent VALU / 8 mfma", 1, 128);
  run<2>("+24 dependent VALU / 8 mfma", 2, 128);
  return 0;
}
