// The same question for a 128x128 macro tile (four waves of 64x64 = 4x4 MFMA tiles, K steps of 16, PF steps of loads in
// flight, 2 workgroups per CU): operands streamed from HBM (every workgroup its own 128 x K panel pair, 16 flops per byte)
// or from L2.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int OKS = 16, OPA = OKS + 1;

template <int PF>
__global__ void __launch_bounds__(256, 2) loop(double* out, const double* __restrict__ src, int steps, long long wg_stride, int ld) {
  __shared__ double As[128 * OPA], Bs[128 * OPA];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  for (int e = tid; e < 128 * OPA; e += 256) { As[e] = e * 0.001; Bs[e] = 1.0 - e * 0.002; }
  __syncthreads();
  const int r16 = lane & 15, kq = lane >> 4;
  const double* a0 = src + (long long)blockIdx.x * wg_stride;
  const double* b0 = a0 + 128ll * ld;
  const long long voff = (long long)(tid / OKS) * ld + (tid % OKS);
  double ra[PF][8], rb[PF][8];
  auto fetch = [&](int slot, int ke) {
    for (int u = 0; u < 8; ++u) { ra[slot][u] = a0[voff + (long long)u * 16 * ld + ke]; rb[slot][u] = b0[voff + (long long)u * 16 * ld + ke]; }
  };
  f64x4 acc[4][4] = {};
#pragma unroll
  for (int p = 0; p < PF; ++p) fetch(p, p * OKS);
  for (int st = 0; st < steps; st += PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      for (int u = 0; u < 8; ++u) {
        const int e = tid + u * 256;
        As[(e / OKS) * OPA + (e % OKS)] = ra[p][u];
        Bs[(e / OKS) * OPA + (e % OKS)] = rb[p][u];
      }
      __syncthreads();
      fetch(p, ((st + p + PF) * OKS) % (ld - OKS));
#pragma unroll
      for (int ks = 0; ks < OKS / 4; ++ks) {
        const int k = 4 * ks + kq;
        double a[4], b[4];
        for (int m = 0; m < 4; ++m) a[m] = As[(64 * wm + 16 * m + r16) * OPA + k];
        for (int n = 0; n < 4; ++n) b[n] = Bs[(64 * wn + 16 * n + r16) * OPA + k];
        for (int m = 0; m < 4; ++m)
          for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
      }
      __syncthreads();
    }
  }
  double s = ra[0][0] + rb[0][3];
  for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
  out[blockIdx.x * 256 + tid] = s;
}

int main() {
  const int grid = 256 * 2, steps = 2048, ld = 4096;
  double *out, *small, *big;
  hipMalloc(&out, (size_t)grid * 256 * 8);
  hipMalloc(&small, 256ull * ld * 8);
  hipMalloc(&big, (size_t)grid * 256 * ld * 8);                  // 4.3 GB
  hipMemset(small, 0, 256ull * ld * 8); hipMemset(big, 0, (size_t)grid * 256 * ld * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int v = 0; v < 4; ++v) {
    float ms = 0;
    const bool hbm = v >= 2;
    const double* src = hbm ? big : small;
    const long long stride = hbm ? 256ll * ld : 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (v % 2 == 0) hipLaunchKernelGGL(loop<1>, dim3(grid), dim3(256), 0, 0, out, src, steps, stride, ld);
      else hipLaunchKernelGGL(loop<2>, dim3(grid), dim3(256), 0, 0, out, src, steps, stride, ld);
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    const double flops = (double)grid * steps * (OKS / 4) * 4 * 16 * 2048.0;
    printf("macro tile, %d step(s) in flight, operands from %s: %.3f ms  %.1f TFLOP/s\n", v % 2 + 1, hbm ? "HBM" : "L2 ", ms, flops / ms / 1e9);
  }
  return 0;
}
