// 64x64 fp64 tile building blocks on v_mfma_f64_16x16x4_f64, shared by the blocked Cholesky / triangular
// inverse (invert.hip) and the block-Jacobi eigensolver (eigh.hip).
#pragma once
#include "common.h"

namespace curv {

constexpr int NB = 64;                 // block edge
constexpr int MMA_THREADS = 256;       // 4 waves, each owning a 32x32 quadrant (2x2 MFMA tiles of 16x16)
constexpr int LDA = NB + 1;            // LDS row pitch (doubles) of an operand tile: bank spread

typedef __attribute__((address_space(1))) double gdouble;

// acc(2x2 of 16x16 per wave, 32x32 wave quadrant) += Atile(64 x 64: [row][k]) * Btile
//   BT = true : B given as [col][k]  (C += A * B^T, both K-contiguous)
//   BT = false: B given as [k][col]
// Both tiles live in LDS with pitch LDA doubles.  Operand maps of v_mfma_f64_16x16x4_f64:
// A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15], C/D col = lane & 15, row = (lane >> 4) + 4 reg.
template <bool BT>
__device__ __forceinline__ void mma_64(const double* __restrict__ As, const double* __restrict__ Bs, int wm,
                                       int wn, int lane, f64x4 (&acc)[2][2]) {
  const int r16 = lane & 15, kq = lane >> 4;
#pragma unroll 4
  for (int ks = 0; ks < NB / 4; ++ks) {
    const int k = 4 * ks + kq;
    double a[2], b[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) a[m] = As[(32 * wm + 16 * m + r16) * LDA + k];
#pragma unroll
    for (int n = 0; n < 2; ++n)
      b[n] = BT ? Bs[(32 * wn + 16 * n + r16) * LDA + k] : Bs[k * LDA + 32 * wn + 16 * n + r16];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
  }
}

// 64x64 block copy global (pitch ld) -> LDS (pitch LDA).  The block base is wave-uniform and every lane's
// offset inside the block is a constant: all 16 loads are issued before the first LDS store, as SGPR base +
// 32-bit VGPR offset, without per-element 64-bit index arithmetic (these copies sit on the serial chain of
// the blocked Cholesky, where their issue time is latency).
typedef __attribute__((address_space(1))) char gbyte;
__device__ __forceinline__ void load_block(const gdouble* __restrict__ g, int ld, double* __restrict__ s) {
  const int tid = threadIdx.x, r0 = tid >> 6, c = tid & 63;
  const unsigned voff = (unsigned)(((long long)r0 * ld + c) * 8);
  const long long step = 4ll * ld * 8;                       // four rows per 256 lanes
  const gbyte* base = (const gbyte*)g;
  double v[NB * NB / MMA_THREADS];
#pragma unroll
  for (int u = 0; u < NB * NB / MMA_THREADS; ++u) v[u] = *(const gdouble*)(base + u * step + voff);
#pragma unroll
  for (int u = 0; u < NB * NB / MMA_THREADS; ++u) s[(r0 + 4 * u) * LDA + c] = v[u];
}

// 64x64 block copy LDS (pitch LDA) -> global (pitch ld), same addressing as load_block
__device__ __forceinline__ void store_block(gdouble* __restrict__ g, int ld, const double* __restrict__ s) {
  const int tid = threadIdx.x, r0 = tid >> 6, c = tid & 63;
  const unsigned voff = (unsigned)(((long long)r0 * ld + c) * 8);
  const long long step = 4ll * ld * 8;
  gbyte* base = (gbyte*)g;
#pragma unroll
  for (int u = 0; u < NB * NB / MMA_THREADS; ++u) *(gdouble*)(base + u * step + voff) = s[(r0 + 4 * u) * LDA + c];
}

// wave quadrant accumulators (2x2 MFMA tiles of the 32x32 quadrant (wm, wn)) <-> a 64x64 global tile of pitch
// ld: element (m, n, q) of a lane sits at a wave-uniform offset from the lane's first element.
//   mode 0: C -= acc   1: C = acc   2: C += acc   3: C = -acc
__device__ __forceinline__ void store_acc(gdouble* __restrict__ C, int ld, const f64x4 (&acc)[2][2], int wm, int wn,
                                          int lane, int mode) {
  const int c16 = lane & 15, rq = lane >> 4;
  const unsigned voff = (unsigned)(((long long)(32 * wm + rq) * ld + 32 * wn + c16) * 8);
  gbyte* base = (gbyte*)C;
  double old[2][2][4];
  if (mode == 0 || mode == 2) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          old[m][n][q] = *(const gdouble*)(base + ((long long)(16 * m + 4 * q) * ld + 16 * n) * 8 + voff);
  }
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double v = acc[m][n][q];
        const double out = mode == 0 ? old[m][n][q] - v : mode == 1 ? v : mode == 2 ? old[m][n][q] + v : -v;
        *(gdouble*)(base + ((long long)(16 * m + 4 * q) * ld + 16 * n) * 8 + voff) = out;
      }
}

// wave quadrant accumulators -> LDS tile [row][col]
__device__ __forceinline__ void acc_to_lds(const f64x4 (&acc)[2][2], int wm, int wn, int lane, double* s) {
  const int c16 = lane & 15, rq = lane >> 4;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) s[(32 * wm + 16 * m + rq + 4 * r) * LDA + 32 * wn + 16 * n + c16] = acc[m][n][r];
}

}  // namespace curv
