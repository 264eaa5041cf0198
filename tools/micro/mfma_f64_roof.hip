// Settles the fp64 MFMA roof of the MI355X (round-3 review, item 2a): is 105 "cycles at 2.4 GHz" per
// v_mfma_f64_16x16x4_f64 a property of the instruction or of the clock under fp64 load?
//   * 16 independent accumulators, s_setprio 3, registers only, 1 / 2 / 4 waves per SIMD on every CU, and ONE wave on
//     ONE CU (no power limit in sight): wall time per MFMA from HIP events
//   * the same for v_mfma_f64_4x4x4_4b_f64 (4 blocks of 4x4x4: 512 flops)
//   * in-kernel: s_memtime (constant 100 MHz on gfx9) around a 1024-MFMA stretch of wave 0
//   * run under `rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace`: GUI_ACTIVE / kernel duration = the shader clock
//     during each launch  ->  cycles per MFMA = wall time per MFMA x that clock
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_f64_roof.hip -o tools/micro/mfma_f64_roof
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ void __launch_bounds__(256) roof(double* out, long long* stamps, int iters) {
  f64x4 c[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) c[i] = f64x4{0, 0, 0, 0};
  double a = threadIdx.x * 0.5, b = 1.0 + threadIdx.x;
  __builtin_amdgcn_s_setprio(3);
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (KIND == 0) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
      else if (KIND == 1) { double r = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[i][0], 0, 0, 0); c[i][0] = r; }
      else {       // one 16x16x4 product as four 4x4x4_4b with A broadcast (cbsz = 2, abid = accumulator register)
        c[i][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[i][0], 2, 0, 0);
        c[i][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[i][1], 2, 1, 0);
        c[i][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[i][2], 2, 2, 0);
        c[i][3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[i][3], 2, 3, 0);
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) stamps[blockIdx.x] = t1 - t0;
}

int main() {
  double* out; long long* st;
  hipMalloc(&out, 8 * 256 * 4096); hipMalloc(&st, 8 * 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  struct { const char* name; int grid; int threads; } cfg[] = {
      {"1 wave on 1 CU", 1, 64}, {"4 waves on 1 CU", 1, 256}, {"1 wave/SIMD, all CUs", 256, 256},
      {"2 waves/SIMD, all CUs", 512, 256}, {"4 waves/SIMD, all CUs", 1024, 256}};
  for (int kind = 0; kind < 3; ++kind) {
    const double flops_per = kind == 1 ? 512.0 : 2048.0;
    for (auto& c : cfg) {
      const int iters = 20000;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(roof<0>, dim3(c.grid), dim3(c.threads), 0, 0, out, st, iters);
        else if (kind == 1) hipLaunchKernelGGL(roof<1>, dim3(c.grid), dim3(c.threads), 0, 0, out, st, iters);
        else hipLaunchKernelGGL(roof<2>, dim3(c.grid), dim3(c.threads), 0, 0, out, st, iters);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      }
      long long ticks = 0; hipMemcpy(&ticks, st, 8, hipMemcpyDeviceToHost);
      const double waves = (double)c.grid * c.threads / 64.0;
      const double n_mfma = (double)iters * 16;
      const double wps = c.grid >= 256 ? c.grid / 256.0 * (c.threads / 256.0) : c.threads / 256.0;   // waves per SIMD (one CU: 1 wave = 0.25)
      printf("%s %-24s: %.3f ms  %.2f TFLOP/s  wall ns per MFMA per wave %.2f  s_memtime ticks per MFMA (wave 0) %.4f\n",
             kind == 0 ? "f64 16x16x4   " : kind == 1 ? "f64 4x4x4_4b  " : "16x16x4 as 4x4x4x4 bcast", c.name, ms, waves * n_mfma * flops_per / ms / 1e9,
             ms * 1e6 / n_mfma, (double)ticks / n_mfma);
      (void)wps;
    }
  }
  return 0;
}
