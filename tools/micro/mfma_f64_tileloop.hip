// K loop of a 64x64 fp64 tile product from LDS operands (the inner loop of invert.hip's outer update), two ways:
//   native : per k-step of 4 and wave (32x32 quadrant) 4 ds_read_b64 + 4 v_mfma_f64_16x16x4_f64
//   blocks : 8 ds_read_b64 (A rows rotated by 0 / 1 row blocks, B columns by 0 / 2 column blocks) + 16 v_mfma_f64_4x4x4_4b_f64
// registers / LDS only (operands constant), 3 workgroups of 256 threads per CU as in outer_update_kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int OKS = 32, OPA = OKS + 1, NB = 64;

template <int KIND>
__global__ void __launch_bounds__(256, 3) loop(double* out, int iters) {
  __shared__ double As[NB * OPA], Bs[NB * OPA];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  for (int e = tid; e < NB * OPA; e += 256) { As[e] = e * 0.001; Bs[e] = 1.0 - e * 0.002; }
  __syncthreads();
  const int r16 = lane & 15, kq = lane >> 4;
  f64x4 acc[2][2] = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll 4
    for (int ks = 0; ks < OKS / 4; ++ks) {
      const int k = 4 * ks + kq;
      if (KIND == 0) {
        double a[2], b[2];
        for (int m = 0; m < 2; ++m) a[m] = As[(32 * wm + 16 * m + r16) * OPA + k];
        for (int n = 0; n < 2; ++n) b[n] = Bs[(32 * wn + 16 * n + r16) * OPA + k];
        for (int m = 0; m < 2; ++m)
          for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
      } else {
        double a[2][2], b[2][2];
        for (int m = 0; m < 2; ++m)
          for (int r = 0; r < 2; ++r) a[m][r] = As[(32 * wm + 16 * m + ((r16 + 4 * r) & 15)) * OPA + k];
        for (int n = 0; n < 2; ++n)
          for (int r = 0; r < 2; ++r) b[n][r] = Bs[(32 * wn + 16 * n + ((r16 + 8 * r) & 15)) * OPA + k];
        for (int m = 0; m < 2; ++m)
          for (int n = 0; n < 2; ++n)
            for (int q = 0; q < 4; ++q)
              acc[m][n][q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[m][q >> 1], b[n][q & 1], acc[m][n][q], 0, 0, 0);
      }
    }
  }
  double s = 0;
  for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
  out[blockIdx.x * 256 + tid] = s;
}

int main() {
  double* out; hipMalloc(&out, 8 * 256 * 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int kind = 0; kind < 2; ++kind)
    for (int wg = 1; wg <= 3; ++wg) {
      const int grid = 256 * wg, iters = 4000; float ms;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(loop<0>, dim3(grid), dim3(256), 0, 0, out, iters);
        else hipLaunchKernelGGL(loop<1>, dim3(grid), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      }
      const double flops = (double)grid * iters * (OKS / 4) * 4 /*waves*/ * 4 /*16x16x4 per wave-step*/ * 2048.0;
      printf("%s, %d workgroup(s)/CU: %.3f ms  %.1f TFLOP/s\n", kind == 0 ? "native 16x16x4" : "4x4x4_4b blocks", wg, ms, flops / ms / 1e9);
    }
  return 0;
}
