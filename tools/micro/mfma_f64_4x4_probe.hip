// Operand layout and A-broadcast (CBSZ / ABID) of v_mfma_f64_4x4x4_4b_f64 on gfx950, checked against a host
// reference.  Layout (found by brute force over the digit permutations of a first dump, confirmed here): block b,
//   A[i][k] at lane (i + 4 b) + 16 k,   B[k][j] at lane (j + 4 b) + 16 k,   D[i][j] at lane (j + 4 b) + 16 i
// - i.e. the operand registers of v_mfma_f64_16x16x4_f64 (A[row = lane & 15][k = lane >> 4], B[k][col = lane & 15]),
// of which the four blocks are the 4x4 diagonal blocks.  With cbsz = 2, abid = q every block multiplies block q's A:
// D[i][col] = sum_k A[4 q + i][k] B[k][col] for all 16 columns = register q of the 16x16x4 accumulator
// (row = (lane >> 4) + 4 q, col = lane & 15).  Four of them replace one v_mfma_f64_16x16x4_f64.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_f64_4x4_probe.hip -o tools/micro/mfma_f64_4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>

template <int CBSZ, int ABID>
__global__ void probe(const double* a, const double* b, double* d) {
  const int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, CBSZ, ABID, 0);
}

int main() {
  double ha[64], hb[64], hd[64], *da, *db, *dd;
  srand(1);
  for (int l = 0; l < 64; ++l) { ha[l] = (rand() % 17) - 8; hb[l] = (rand() % 13) - 6; }
  hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 512);
  hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
  auto check = [&](const char* name, int cbsz, int abid) {
    hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
    printf("%s D:", name); for (int l = 0; l < 64; ++l) printf(" %g", hd[l]); printf("\n");
    // candidate layouts for D: (a) lane = 16 b + 4 i + j, (b) lane = 16 b + 4 j + i
    for (int cand = 0; cand < 2; ++cand) {
      int bad = 0;
      for (int blk = 0; blk < 4; ++blk) {
        const int ablk = cbsz == 0 ? blk : (cbsz == 2 ? abid : (blk & ~((1 << cbsz) - 1)) + abid);
        for (int i = 0; i < 4; ++i)
          for (int j = 0; j < 4; ++j) {
            double ref = 0;
            for (int k = 0; k < 4; ++k) ref += ha[i + 4 * ablk + 16 * k] * hb[j + 4 * blk + 16 * k];
            const int lane = cand == 0 ? j + 4 * blk + 16 * i : i + 4 * blk + 16 * j;
            if (fabs(hd[lane] - ref) > 1e-9) ++bad;
          }
      }
      printf("%s: D layout candidate %s: %d mismatches of 64\n", name, cand == 0 ? "lane = j + 4 b + 16 i" : "lane = i + 4 b + 16 j", bad);
    }
  };
  hipLaunchKernelGGL((probe<0, 0>), dim3(1), dim3(64), 0, 0, da, db, dd); hipDeviceSynchronize(); check("cbsz 0        ", 0, 0);
  hipLaunchKernelGGL((probe<2, 0>), dim3(1), dim3(64), 0, 0, da, db, dd); hipDeviceSynchronize(); check("cbsz 2 abid 0 ", 2, 0);
  hipLaunchKernelGGL((probe<2, 1>), dim3(1), dim3(64), 0, 0, da, db, dd); hipDeviceSynchronize(); check("cbsz 2 abid 1 ", 2, 1);
  hipLaunchKernelGGL((probe<2, 3>), dim3(1), dim3(64), 0, 0, da, db, dd); hipDeviceSynchronize(); check("cbsz 2 abid 3 ", 2, 3);
  hipLaunchKernelGGL((probe<1, 1>), dim3(1), dim3(64), 0, 0, da, db, dd); hipDeviceSynchronize(); check("cbsz 1 abid 1 ", 1, 1);
  // raw dump for a manual look if nothing matched
  hipLaunchKernelGGL((probe<0, 0>), dim3(1), dim3(64), 0, 0, da, db, dd); hipDeviceSynchronize();
  hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
  printf("A:"); for (int l = 0; l < 64; ++l) printf(" %g", ha[l]); printf("\nB:"); for (int l = 0; l < 64; ++l) printf(" %g", hb[l]);
  printf("\nD:"); for (int l = 0; l < 64; ++l) printf(" %g", hd[l]); printf("\n");
  return 0;
}
