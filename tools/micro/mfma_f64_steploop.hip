// What costs the far update of invert.hip its MFMA rate?  The K loop of outer_update_kernel rebuilt piece by piece, five
// workgroups of 256 threads per CU (82 VGPRs, 17.4 KB of LDS as in the kernel), K steps of 16:
//   V0  LDS operand reads + 16 MFMAs per wave and step, nothing else
//   V1  + the two barriers of a step
//   V2  + the LDS stores of a step (8 doubles per thread from registers)
//   V3  + the global loads of the next step (8 per thread, one step ahead), operands in L2 (small, re-read buffer)
//   V4  as V3 with operands streamed from HBM (every workgroup its own 64 x K panel pair)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int OKS = 16, OPA = OKS + 1, NB = 64;

template <int V>
__global__ void __launch_bounds__(256, 5) loop(double* out, const double* __restrict__ src, int steps, long long wg_stride, int ld) {
  __shared__ double As[NB * OPA], Bs[NB * OPA];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  for (int e = tid; e < NB * OPA; e += 256) { As[e] = e * 0.001; Bs[e] = 1.0 - e * 0.002; }
  __syncthreads();
  const int r16 = lane & 15, kq = lane >> 4;
  const double* a0 = src + (long long)blockIdx.x * wg_stride;
  const double* b0 = a0 + 64ll * ld;
  const long long voff = (long long)(tid / OKS) * ld + (tid % OKS);
  double ra[4], rb[4];
  for (int u = 0; u < 4; ++u) { ra[u] = 0.5 + u; rb[u] = 0.25 * u; }
  auto fetch = [&](int ke) {
    for (int u = 0; u < 4; ++u) { ra[u] = a0[voff + (long long)u * 16 * ld + ke]; rb[u] = b0[voff + (long long)u * 16 * ld + ke]; }
  };
  f64x4 acc[2][2] = {};
  if (V >= 3) fetch(0);
  for (int st = 0; st < steps; ++st) {
    if (V >= 2) {
      for (int u = 0; u < 4; ++u) {
        const int e = tid + u * 256;
        As[(e / OKS) * OPA + (e % OKS)] = ra[u];
        Bs[(e / OKS) * OPA + (e % OKS)] = rb[u];
      }
    }
    if (V >= 1) __syncthreads();
    if (V >= 3) fetch(((st + 1) * OKS) % (ld - OKS));
#pragma unroll 4
    for (int ks = 0; ks < OKS / 4; ++ks) {
      const int k = 4 * ks + kq;
      double a[2], b[2];
      for (int m = 0; m < 2; ++m) a[m] = As[(32 * wm + 16 * m + r16) * OPA + k];
      for (int n = 0; n < 2; ++n) b[n] = Bs[(32 * wn + 16 * n + r16) * OPA + k];
      for (int m = 0; m < 2; ++m)
        for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
    }
    if (V >= 1) __syncthreads();
  }
  double s = ra[0] + rb[3];
  for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
  out[blockIdx.x * 256 + tid] = s;
}

int main() {
  const int grid = 256 * 5, steps = 4096, ld = 4096;            // K = 65536 per workgroup in steps of 16
  double *out, *small, *big;
  hipMalloc(&out, (size_t)grid * 256 * 8);
  hipMalloc(&small, 128ull * ld * 8);                            // 4 MB: stays in L2
  hipMalloc(&big, (size_t)grid * 128 * ld * 8);                  // 5.4 GB: streamed from HBM
  hipMemset(small, 0, 128ull * ld * 8); hipMemset(big, 0, (size_t)grid * 128 * ld * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[5] = {"V0 reads + MFMA", "V1 + 2 barriers / step", "V2 + LDS stores", "V3 + global loads (L2)", "V4 + global loads (HBM)"};
  for (int v = 0; v < 5; ++v) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      const double* src = v == 4 ? big : small;
      const long long stride = v == 4 ? 128ll * ld : 0;
      switch (v) {
        case 0: hipLaunchKernelGGL(loop<0>, dim3(grid), dim3(256), 0, 0, out, src, steps, stride, ld); break;
        case 1: hipLaunchKernelGGL(loop<1>, dim3(grid), dim3(256), 0, 0, out, src, steps, stride, ld); break;
        case 2: hipLaunchKernelGGL(loop<2>, dim3(grid), dim3(256), 0, 0, out, src, steps, stride, ld); break;
        case 3: hipLaunchKernelGGL(loop<3>, dim3(grid), dim3(256), 0, 0, out, src, steps, stride, ld); break;
        default: hipLaunchKernelGGL(loop<3>, dim3(grid), dim3(256), 0, 0, out, src, steps, stride, ld); break;
      }
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    const double flops = (double)grid * steps * (OKS / 4) * 4 * 4 * 2048.0;
    printf("%-28s: %.3f ms  %.1f TFLOP/s\n", names[v], ms, flops / ms / 1e9);
  }
  return 0;
}
