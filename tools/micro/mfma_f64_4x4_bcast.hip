// What do CBSZ / ABID do to v_mfma_f64_4x4x4_4b_f64 on gfx950?  A = unit vector e_t (one launch block per t), B = distinct
// primes: D[l] then names the B lane that A lane t is multiplied with for output lane l.  Printed as: for every output
// lane the four (A lane, B lane) pairs it sums.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CBSZ, int ABID>
__global__ void probe(const double* b, double* d) {
  const int l = threadIdx.x, t = blockIdx.x;
  d[t * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(l == t ? 1.0 : 0.0, b[l], 0.0, CBSZ, ABID, 0);
}
int main() {
  double hb[64], *db, *dd; static double hd[64 * 64];
  for (int l = 0; l < 64; ++l) hb[l] = 1000 + l;
  hipMalloc(&db, 512); hipMalloc(&dd, 64 * 512);
  hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
  auto show = [&](const char* name) {
    hipDeviceSynchronize();
    hipMemcpy(hd, dd, 64 * 512, hipMemcpyDeviceToHost);
    printf("%s\n", name);
    for (int l = 0; l < 64; ++l) {
      printf("  out %2d:", l);
      for (int t = 0; t < 64; ++t) { const double v = hd[t * 64 + l]; if (v != 0.0) printf(" (A%d,B%d)", t, (int)(v - 1000)); }
      printf("\n");
    }
  };
  hipLaunchKernelGGL((probe<0, 0>), dim3(64), dim3(64), 0, 0, db, dd); show("cbsz 0");
  hipLaunchKernelGGL((probe<2, 1>), dim3(64), dim3(64), 0, 0, db, dd); show("cbsz 2 abid 1");
  hipLaunchKernelGGL((probe<1, 1>), dim3(64), dim3(64), 0, 0, db, dd); show("cbsz 1 abid 1");
  hipLaunchKernelGGL((probe<1, 0>), dim3(64), dim3(64), 0, 0, db, dd); show("cbsz 1 abid 0");
  return 0;
}
