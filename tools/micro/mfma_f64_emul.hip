// One v_mfma_f64_16x16x4_f64 as four v_mfma_f64_4x4x4_4b_f64 (16 cycles each instead of 105-138 for the one): the
// 4-block instruction multiplies the 4x4 DIAGONAL blocks of the same operand registers (A[row = lane & 15][k = lane >> 4],
// B[k][col = lane & 15]; CBSZ / ABID are ignored for f64), so product r uses B with its column blocks rotated by r
// (DPP row rotation of the 16-lane rows) and yields the blocks (b, (b + r) % 4).  Checks the result against the 16x16x4
// instruction and times the sequence (registers only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// rotated[l] = v[(l & 48) | ((l + 4 r) & 15)]
__device__ __forceinline__ void emul(double a, double b, f64x4& acc) {
  const double b1 = dpp_f64<0x120 + 12>(b), b2 = dpp_f64<0x120 + 8>(b), b3 = dpp_f64<0x120 + 4>(b);
  acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[0], 0, 0, 0);
  acc[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b1, acc[1], 0, 0, 0);
  acc[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b2, acc[2], 0, 0, 0);
  acc[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b3, acc[3], 0, 0, 0);
}

__global__ void check(const double* a, const double* b, double* d_ref, double* d_em, double* rot) {
  const int l = threadIdx.x;
  f64x4 c = {0, 0, 0, 0}, e = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], c, 0, 0, 0);
  emul(a[l], b[l], e);
  for (int r = 0; r < 4; ++r) { d_ref[4 * l + r] = c[r]; d_em[4 * l + r] = e[r]; }
  rot[l] = dpp_f64<0x120 + 12>((double)l);
}

__global__ void __launch_bounds__(256) rate(double* out, int iters) {
  f64x4 c[8];
  for (int i = 0; i < 8; ++i) c[i] = f64x4{0, 0, 0, 0};
  double a = threadIdx.x * 0.5, b = 1.0 + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { emul(a, b, c[i]); b += 1.0; }
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  double ha[64], hb[64], href[256], hem[256], hrot[64], *da, *db, *dr, *de, *drot, *out;
  for (int l = 0; l < 64; ++l) { ha[l] = (l * 7 % 13) - 6; hb[l] = (l * 5 % 11) - 5; }
  hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dr, 2048); hipMalloc(&de, 2048); hipMalloc(&drot, 512);
  hipMalloc(&out, 8 * 256 * 1024);
  hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, da, db, dr, de, drot);
  hipDeviceSynchronize();
  hipMemcpy(href, dr, 2048, hipMemcpyDeviceToHost); hipMemcpy(hem, de, 2048, hipMemcpyDeviceToHost);
  hipMemcpy(hrot, drot, 512, hipMemcpyDeviceToHost);
  printf("row_ror:12 of lane ids (first 16):"); for (int l = 0; l < 16; ++l) printf(" %g", hrot[l]); printf("\n");
  // 16x16x4 result: D[row = (l >> 4) + 4 reg][col = l & 15].  Emulated register r of lane l (c = l & 15, i = l >> 4):
  // row = 4 (c >> 2) + i, col = 4 (((c >> 2) + r) & 3) + (c & 3)
  double D[16][16];
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) D[(l >> 4) + 4 * r][l & 15] = href[4 * l + r];
  for (int dir = 0; dir < 2; ++dir) {
    int bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) {
        const int c = l & 15, i = l >> 4, rb = c >> 2;
        const int cb = dir == 0 ? (rb + r) & 3 : (rb - r) & 3;
        if (fabs(hem[4 * l + r] - D[4 * rb + i][4 * cb + (c & 3)]) > 1e-9) ++bad;
      }
    printf("emulated vs 16x16x4, column block (rb %c r) %% 4: %d mismatches of 256\n", dir == 0 ? '+' : '-', bad);
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wg = 1; wg <= 4; wg *= 2) {
    const int grid = 256 * wg, iters = 20000; float ms;
    hipLaunchKernelGGL(rate, dim3(grid), dim3(256), 0, 0, out, 10); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(rate, dim3(grid), dim3(256), 0, 0, out, iters); hipEventRecord(e1);
    hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("emulated 16x16x4 (4 MFMA + 6 DPP), %d wave(s)/SIMD: %.3f ms  %.1f TFLOP/s  %.1f cycles per 16x16x4 per SIMD at 2.4 GHz\n", wg, ms,
           (double)grid * 4 * iters * 8 * 2048.0 / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)wg * iters * 8));
  }
  return 0;
}
