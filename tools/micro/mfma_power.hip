// Does the fp32 MFMA rate depend on the DATA?  v_mfma_f32_32x32x2_f32 back to back from registers, 2 waves per SIMD, with
// (a) all-zero operands, (b) constant operands, (c) random operands that change every instruction (a rotating register
// set) - for a short run (~5 ms) and a long one (~100 ms).  If the board regulates power, (c) runs slower than (a) and the
// long run slower than the short one: the sustainable fp32 MFMA rate on real data is then below the 157.3 TFLOP/s datasheet
// figure that bench.py's roofline divides by.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_power mfma_power.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256) loop(float* out, const float* in, int iters, int mode) {
  const int tid = threadIdx.x + blockIdx.x * blockDim.x;
  float a[8], b[8];
  for (int k = 0; k < 8; ++k) {
    a[k] = mode == 0 ? 0.0f : mode == 1 ? 1.5f : in[(tid * 16 + k) & 0xffff];
    b[k] = mode == 0 ? 0.0f : mode == 1 ? 0.75f : in[(tid * 16 + 8 + k) & 0xffff];
  }
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; k += 4) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[k], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k + 1], b[k + 1], c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k + 2], b[k + 2], c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k + 3], b[k + 3], c3, 0, 0, 0);
    }
  }
  float s = 0.0f;
  for (int k = 0; k < 16; ++k) s += c0[k] + c1[k] + c2[k] + c3[k];
  out[tid] = s;
}

int main() {
  const int grid = 256 * 2, threads = 256;       // 2 workgroups of 4 waves per CU = 2 waves per SIMD
  float *out, *in;
  hipMalloc(&out, (size_t)grid * threads * 4);
  hipMalloc(&in, 65536 * 4);
  float* h = new float[65536];
  unsigned s = 12345u;
  for (int i = 0; i < 65536; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
  hipMemcpy(in, h, 65536 * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[3] = {"zero operands", "constant operands", "random operands"};
  for (int iters : {20000, 400000}) {
    for (int mode = 0; mode < 3; ++mode) {
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(loop, dim3(grid), dim3(threads), 0, 0, out, in, iters, mode);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      }
      const double flops = (double)grid * 4 * iters * 8 * 4096.0;
      printf("%-18s %7d iterations: %8.3f ms  %6.1f TFLOP/s\n", names[mode], iters, ms, flops / ms / 1e9);
    }
  }
  return 0;
}
