// Which tile form suits the FAR update of the blocked Cholesky sweep (csrc/invert.hip, outer_update_kernel)?
//   C[i][j] -= sum_{k < 256} A[i][k] A[j][k]      for all tiles row0 <= j <= i of three 4608-wide fp64 matrices
// i.e. one outer panel's trailing update of the chain-bound group of ResNet-50 (K = 256 per pass: short tiles, a
// read-modify-write epilogue, a launch of a few rounds).  Forms: the shipped 64x64 tile on 4 waves, and 128x128 tiles on
// 4 / 8 / 16 waves.  Prints TFLOP/s of the executed tiles (diagonal tiles counted in full).
//   hipcc -O3 --offload-arch=gfx950 -o far_update_forms far_update_forms.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) double gdouble;
constexpr int OKS = 16, OPA = OKS + 1;

template <int TILE, int WM, int WN, int MINWG>
__global__ void __launch_bounds__(64 * WM * WN, MINWG)
far(double* __restrict__ W0, long long mat_stride, int np, int row0_blk, int n_rows, int tiles_per_mat, int Kel) {
  constexpr int THREADS = 64 * WM * WN, TM = TILE / WM / 16, TN = TILE / WN / 16, LPT = TILE * OKS / THREADS;
  __shared__ double As[TILE * OPA], Bs[TILE * OPA];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WN, wn = wave % WN;
  const int r16 = lane & 15, kq = lane >> 4;
  const int mat = blockIdx.x / tiles_per_mat;
  int t = blockIdx.x - mat * tiles_per_mat;
  int a = 0;
  while (t > a) { t -= a + 1; ++a; }              // lower-triangular enumeration: row a, column t
  (void)n_rows;
  double* W = W0 + mat * mat_stride;
  const long long i0 = (long long)row0_blk * 64 + (long long)a * TILE, j0 = (long long)row0_blk * 64 + (long long)t * TILE;
  const bool same = a == t;
  const gdouble* Ag = (const gdouble*)(W + i0 * np);
  const gdouble* Bg = (const gdouble*)(W + j0 * np);
  const long long voff = (long long)(tid / OKS) * np + (tid % OKS), vstep = (long long)(THREADS / OKS) * np;
  double ra[LPT], rb[LPT];
  auto fetch = [&](int ke) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < LPT; ++u) { ra[u] = Ag[voff + u * vstep + ke]; rb[u] = same ? 0.0 : Bg[voff + u * vstep + ke]; }
  };
  f64x4 acc[TM][TN] = {};
  fetch(0);
  for (int ke = 0; ke < Kel; ke += OKS) {
#pragma unroll
    for (int u = 0; u < LPT; ++u) {
      const int e = tid + u * THREADS;
      As[(e / OKS) * OPA + (e % OKS)] = ra[u];
      if (!same) Bs[(e / OKS) * OPA + (e % OKS)] = rb[u];
    }
    __syncthreads();
    if (ke + OKS < Kel) fetch(ke + OKS);
    const double* Bt = same ? As : Bs;
#pragma unroll
    for (int ks = 0; ks < OKS / 4; ++ks) {
      const int k = 4 * ks + kq;
      double av[TM], bv[TN];
#pragma unroll
      for (int m = 0; m < TM; ++m) av[m] = As[(16 * TM * wm + 16 * m + r16) * OPA + k];
#pragma unroll
      for (int n = 0; n < TN; ++n) bv[n] = Bt[(16 * TN * wn + 16 * n + r16) * OPA + k];
#pragma unroll
      for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m], bv[n], acc[m][n], 0, 0, 0);
    }
    __syncthreads();
  }
  // C -= acc: rows 16 TM wm + 16 m + kq + 4 q, column 16 TN wn + 16 n + r16
  gdouble* C = (gdouble*)(W + i0 * np + j0);
#pragma unroll
  for (int m = 0; m < TM; ++m) {
    double old[TN][4];
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) old[n][q] = C[(long long)(16 * TM * wm + 16 * m + kq + 4 * q) * np + 16 * TN * wn + 16 * n + r16];
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) C[(long long)(16 * TM * wm + 16 * m + kq + 4 * q) * np + 16 * TN * wn + 16 * n + r16] = old[n][q] - acc[m][n][q];
  }
}

int main() {
  const int np = 4608, P = np / 64, mats = 3, Kel = 256, row0 = 8;
  double* W;
  const size_t bytes = (size_t)mats * np * np * 8;
  hipMalloc(&W, bytes);
  hipMemset(W, 0, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, int tile, auto launch) {
    const int rows = (P - row0) * 64 / tile, tiles = rows * (rows + 1) / 2;
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      hipEventRecord(e0);
      launch(tiles * mats, tiles, rows);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best) best = ms;
    }
    const double flops = (double)tiles * mats * 2.0 * tile * tile * Kel;
    printf("%-44s %5d workgroups  %.3f ms  %.1f TFLOP/s\n", name, tiles * mats, best, flops / best / 1e9);
  };
  run("64x64 tile, 4 waves (2x2 MFMA tiles each), 3/CU", 64, [&](int g, int t, int r) { hipLaunchKernelGGL((far<64, 2, 2, 3>), dim3(g), dim3(256), 0, 0, W, (long long)np * np, np, row0, r, t, Kel); });
  run("128x128 tile, 4 waves (4x4 each), 2/CU", 128, [&](int g, int t, int r) { hipLaunchKernelGGL((far<128, 2, 2, 2>), dim3(g), dim3(256), 0, 0, W, (long long)np * np, np, row0, r, t, Kel); });
  run("128x128 tile, 8 waves (4x2 each), 2/CU", 128, [&](int g, int t, int r) { hipLaunchKernelGGL((far<128, 2, 4, 2>), dim3(g), dim3(512), 0, 0, W, (long long)np * np, np, row0, r, t, Kel); });
  run("128x128 tile, 16 waves (2x2 each), 1/CU", 128, [&](int g, int t, int r) { hipLaunchKernelGGL((far<128, 4, 4, 1>), dim3(g), dim3(1024), 0, 0, W, (long long)np * np, np, row0, r, t, Kel); });
  run("128x128 tile, 16 waves (2x2 each), 2/CU", 128, [&](int g, int t, int r) { hipLaunchKernelGGL((far<128, 4, 4, 2>), dim3(g), dim3(1024), 0, 0, W, (long long)np * np, np, row0, r, t, Kel); });
  run("128x64 tile ... (as 128 tile with 8 waves 2x4)", 128, [&](int g, int t, int r) { hipLaunchKernelGGL((far<128, 4, 2, 2>), dim3(g), dim3(512), 0, 0, W, (long long)np * np, np, row0, r, t, Kel); });
  // a long-K control: the same forms with K = 2048 (steady state)
  return 0;
}
