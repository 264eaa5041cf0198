// KFAC.invert on gfx950: L = chol_lower( (sqrt(s) F + sqrt(n) I)^-1 ) for every Kronecker factor of a
// model in one batched sweep (curvature/curvatures.py:354-385).
//
// The reference computes inverse() then cholesky() (getrf + getri + potrf, fp32 LAPACK).  Here:
//   chol(M^-1) = J * chol(J M J)^-T * J        (J = index reversal)
// so one Cholesky factorisation C C^T = J M J and one triangular inverse X = C^-1 suffice:
//   L[i][j] = X[n-1-j][n-1-i].
// The damped matrix is formed in fp32 exactly as the reference forms it (:368-375) and everything after
// that runs in fp64 on v_mfma_f64_16x16x4_f64, so the result is the correctly rounded answer for the
// matrix the reference hands to LAPACK (the reference's own fp32 LAPACK noise is 5e-5 .. 6e-4 here).
//
// Batched right-looking blocked algorithm, block 64, all factors of the model advance together; every
// launch is a flat list of independent 64x64 tile operations, so the critical path per step is one block
// operation whatever the matrix size.  Inside an outer panel (4 or 6 block columns):
//   step k:  (1) one workgroup per factor factorises the 64x64 diagonal block in LDS and stores
//                X_kk = L_kk^-1; then, in one launch with (3), A[i][k] <- A[i][k] X_kk^T
//            (2) trailing update  A[i][j] -= A[i][k] A[j][k]^T           for k < j <= i
//            (3) X[k][j] = -X_kk S[k][j]                                 for j < k   (row k of C^-1 final)
//            (4) S[i][j] (+)= C[i][k] X[k][j]                            for i > k, j <= k
//                (S accumulates sum_{k'} C[i][k'] X[k'][j] in the storage of X[i][j])
// which leaves the panel's block square factorised and inverted (X_sq); everything beyond the panel is then updated
// once per panel: the rows below through a triangular product with X_sq, the trailing matrix with K = panel width (near
// strip on the chain's stream, the rest beside the next chain).  For KFAC.invert the inverse OUTSIDE the squares - (3) and
// (4) for rows below / columns left of the panel - is accumulated in fp32 on a second stream (supd32_kernel,
// xrows32_kernel); curv_chol_factor_inverse (INF's fp64 chain) keeps it in fp64 inside the sweep.
// Matrices are padded to a multiple of 64 with an identity tail, so no kernel needs bounds checks.
// "Not positive definite" is reported through a per-factor device info word (0 = ok).
#include <type_traits>
#include "common.h"
#include "mma64.h"
#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <mutex>
#include <vector>

namespace curv {

constexpr int INV_THREADS = MMA_THREADS;
// flag words per factor of chol_square_kernel (zeroed by inv_prepare_kernel at the start of every sweep)
constexpr int SQ_FLAGS = 32;
constexpr int SQ_FD = 0, SQ_FLC = 4, SQ_FLROW = 8, SQ_FT = 12;
constexpr int SQ_SPIN_LIMIT = 1 << 21;
#ifdef CURV_SQ_TRACE
// diagnostics build (tools/sq_trace.py): 100 MHz time stamps of one panel of factor 0
__device__ long long g_sq_trace[256];
#define KT_BEGIN(slot) { if (threadIdx.x == 0 && k0 == 4 * (CURV_SQ_TRACE - 1)) atomicMax((unsigned long long*)&g_sq_trace[slot], (1ull << 62) - (unsigned long long)wall_clock64()); }
#define KT_END(slot) { if (threadIdx.x == 0 && k0 == 4 * (CURV_SQ_TRACE - 1)) atomicMax((unsigned long long*)&g_sq_trace[slot], (unsigned long long)wall_clock64()); }
#define SQT(slot) { if (threadIdx.x == 0 && f == 0 && stamp == CURV_SQ_TRACE) g_sq_trace[slot] = wall_clock64(); }
#define FIT(k) { if (threadIdx.x == 0 && trace_base >= 0) g_sq_trace[trace_base + (k)] = wall_clock64(); }
#else
#define SQT(slot)
#define FIT(k)
#define KT_BEGIN(slot) {}
#define KT_END(slot) {}
#endif
#ifdef CURV_SQ_TRACE
#define PQT(slot) { if (threadIdx.x == 0 && blockIdx.x == 0 && k0 == 4 * (CURV_SQ_TRACE - 1)) g_sq_trace[slot] = wall_clock64(); }
#else
#define PQT(slot)
#endif

struct InvDev {
  const float* F;       // (n x n) fp32 factor
  float* L;             // (n x n) fp32 output
  double* W;            // (np x np) fp64 work matrix: reversed damped factor -> Cholesky factor C
  double* X;            // (np x np) fp64: C^-1
  int* info;            // device status word of this factor
  int n, np, P;
  float sqrt_s, sqrt_n;
  int reverse;          // 1: factor J M J and emit L = J X^T J (KFAC.invert); 0: plain M, emit X = chol(M)^-1
  int f64_in;           // reverse == 0 only: F points to an fp64 matrix (INF's V_s^T V_s), damping added in fp64
  double* Xout;         // reverse == 0: (n x n) fp64 output, lower triangular
  // reverse == 0, right-hand side mode: emit Z = chol(M)^-1 R for a lower-triangular (n x n) fp64 R instead of the inverse
  // itself - INF.pre_sampler's B_c^-1 A_c^-1 without B_c^-1 and without the product (curvatures.py:566-570).  The forward
  // substitution is the sweep's own inverse accumulation with R in the place of the identity: Zm (np x np, workspace) starts
  // as R, every outer panel subtracts C[i][panel] Z[panel][j] from the rows below it and multiplies its own rows by the
  // inverse X_sq of its block square - which is all of C^-1 that is formed (X holds the squares, nothing between them).
  const double* R;
  double* Zm;
  int r_minus, pad_;    // emit R - Z instead of Z
  // reverse == 1 (KFAC.invert): the triangular inverse outside the block squares is accumulated in fp32, off the chain
  // (supd32_kernel, xrows32_kernel): C32 = fp32 copy of the rows of C below each square (written by the panel product),
  // S32 = the running sums, X32 = the finished rows of C^-1 outside the squares (the squares themselves stay in X, fp64).
  // All (np x np) fp32.
  float* C32;
  float* X32;           // finished rows of C^-1 (outside the squares)
  float* S32;           // the running sums S
  double pivot_min;     // pivots at or below this count as failed (0: the positive-definiteness test of the estimators)
};
typedef __attribute__((address_space(1))) float gfloat;
// does this factor accumulate S in fp64 inside the sweep (the round-1 form: curv_chol_factor_inverse, whose fp64 outputs
// feed INF's chain), or in fp32 off the chain (KFAC.invert)?
__device__ __host__ __forceinline__ bool s_in_sweep(const InvDev& d) { return d.X32 == nullptr; }
// 0: fp32 inverse off the chain (KFAC.invert), 1: fp64 inverse inside the sweep, 2: right-hand side mode (Zm)
__device__ __host__ __forceinline__ int s_kind(const InvDev& d) { return d.X32 != nullptr ? 0 : (d.Zm != nullptr ? 2 : 1); }

// block -> (factor, local tile) for per-factor tile counts cnt(f) that depend on the step
template <typename CountFn>
__device__ __forceinline__ bool locate(const InvDev* __restrict__ t, int nf, int bid, CountFn cnt, int& f,
                                       int& local) {
  const int lane = threadIdx.x & 63;
  int base = 0;
  for (int f0 = 0; f0 < nf; f0 += 64) {
    const int ff = f0 + lane;
    const int c = (ff < nf) ? cnt(t[ff]) : 0;
    int incl = c;                       // inclusive wave scan
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o, 64);
      if (lane >= o) incl += v;
    }
    const int total = __builtin_amdgcn_readfirstlane(__shfl(incl, 63, 64));   // uniform: keeps the callers' control flow scalar
    if (bid < base + total) {
      const unsigned long long m = __ballot(bid < base + incl);
      const int l = __ffsll((long long)m) - 1;
      f = __builtin_amdgcn_readfirstlane(f0 + l);               // wave-uniform by construction: let the
      local = __builtin_amdgcn_readfirstlane(bid - (base + __shfl(incl - c, l, 64)));   // compiler know it
      return true;
    }
    base += total;
  }
  return false;
}

// ------------------------------------------------------------------------------------------------
// (0) W = J (sqrt_s F + sqrt_n I) J in fp64, formed with the reference's fp32 rounding sequence;
//     identity tail on the padding; lower triangle only.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(INV_THREADS)
inv_prepare_kernel(const InvDev* __restrict__ t, int nf, int* __restrict__ flags) {
  int f, tile;
  if (!locate(t, nf, blockIdx.x, [](const InvDev& d) { return d.P * (d.P + 1) / 2; }, f, tile)) return;
  const InvDev& d = t[f];
  int bi = 0;
  while (tile > bi) { tile -= bi + 1; ++bi; }       // lower-triangular enumeration: row bi has bi+1 blocks
  const int bj = tile;
  const int n = d.n, np = d.np, rev = d.reverse;
  const float ss = d.sqrt_s, sn = d.sqrt_n;
  const gfloat* F = (const gfloat*)d.F;
  gdouble* W = (gdouble*)d.W;
  if (bi == 0 && bj == 0 && threadIdx.x == 0) *d.info = 0;
  if (bi == 0 && bj == 0 && threadIdx.x < SQ_FLAGS) flags[(long long)f * SQ_FLAGS + threadIdx.x] = 0;
  if (d.Zm != nullptr) {
    // right-hand side mode: the work matrix starts as R (lower triangle, zeros above inside the diagonal tiles and in
    // the padding; tiles above the diagonal are never read)
    const gdouble* Rg = (const gdouble*)d.R;
    gdouble* Z = (gdouble*)d.Zm;
    const int w0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), c0 = threadIdx.x & 63;
    const int jj = bj * NB + c0;
#pragma unroll 4
    for (int u = 0; u < 16; ++u) {
      const int i = bi * NB + w0 + 4 * u;
      Z[(long long)i * np + jj] = (i < n && jj <= i) ? Rg[(long long)i * n + jj] : 0.0;
    }
  }
  if (d.f64_in) {
    // fp64 input (already symmetric by construction; symmetrised again at no cost): W = (M + M^T) / 2 + add * I
    const gdouble* M = (const gdouble*)d.F;
    const int w0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), c0 = threadIdx.x & 63;
    const int jj = bj * NB + c0;
#pragma unroll 4
    for (int u = 0; u < 16; ++u) {
      const int i = bi * NB + w0 + 4 * u;
      double v;
      if (i < n && jj < n) {
        v = 0.5 * (M[(long long)i * n + jj] + M[(long long)jj * n + i]);
        if (i == jj) v += (double)sn;
      } else {
        v = (i == jj) ? 1.0 : 0.0;
      }
      W[(long long)i * np + jj] = v;
    }
    return;
  }
  // The symmetrisation needs F[ri][rj] and its mirror F[rj][ri]: the mirror block is read row-wise (coalesced)
  // into LDS and consumed transposed, instead of 64 lanes striding through 64 rows of F.  A wave owns the rows
  // w, w + 4, ... of the block, so every row base is wave-uniform; all 32 loads of a lane are issued before
  // the first use (HBM-bound kernel: 12 B per element of the lower triangle).
  __shared__ float Tm[NB][NB + 1];
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), c = threadIdx.x & 63;
  const int j = bj * NB + c, rj = rev ? n - 1 - j : j;           // direct block: column of this lane
  const int mi = bi * NB + c, rmi = rev ? n - 1 - mi : mi;       // mirror block: column of this lane (i range)
  float fd[16], fm[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int r = w + 4 * u;
    const int i = bi * NB + r, ri = rev ? n - 1 - i : i;         // direct row
    const int mj = bj * NB + r, rmj = rev ? n - 1 - mj : mj;     // mirror row (j range)
    fd[u] = (i < n && j < n) ? F[(long long)ri * n + rj] : 0.0f;
    fm[u] = (mi < n && mj < n) ? F[(long long)rmj * n + rmi] : 0.0f;
  }
#pragma unroll
  for (int u = 0; u < 16; ++u) Tm[w + 4 * u][c] = fm[u];
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int r = w + 4 * u;
    const int i = bi * NB + r;
    double v;
    if (i < n && j < n) {
      // reg = s**0.5 * F + diag(n**0.5); reg = (reg + reg.t()) / 2   (curvatures.py:368-375), in fp32
      float a = __fmul_rn(ss, fd[u]);
      float b = __fmul_rn(ss, Tm[c][r]);
      if (i == j) { a = __fadd_rn(a, sn); b = __fadd_rn(b, sn); }
      v = (double)__fmul_rn(__fadd_rn(a, b), 0.5f);
    } else {
      v = (i == j) ? 1.0 : 0.0;
    }
    W[(long long)i * np + j] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// (2) trailing update A[i][j] -= A[i][k] A[j][k]^T for k < j <= i and
// (4) S[i][j] (+)= C[i][k] X[k][j] for i > k, j <= k, as two-level (inner / outer panel) tile lists
// ------------------------------------------------------------------------------------------------
// Tile sets of the two-level sweep.  Outer panels of NBO block columns: inside a panel the updates of
// step k are restricted to the panel (K = 64 products), everything beyond the panel is updated once per
// panel with K = NBO * 64, which divides the read-modify-write traffic of the trailing matrix by NBO.
__device__ __host__ __forceinline__ long long inner_tiles(int P, int k, int k0, int kend) {
  const int ke = kend < P ? kend : P;
  long long tr = 0;
  for (int j = k + 1; j < ke; ++j) tr += ke - j;
  const long long sr = (ke - k - 1 > 0) ? (long long)(ke - k - 1) * (k - k0 + 1) : 0;
  return tr + sr;
}
// Outer tile lists are enumerated in SB x SB super-blocks (consecutive work items = one super-block, mapped to one
// XCD, whose L2 then holds the super-block's 2 SB operand panels).  Round 4, same-box A/B on the ResNet-50 factors:
// SB = 16 8.7 ms, 8 (rounds 1-3) 8.32, 4 8.12, 3 8.08, 2 8.12 - the kernel does not live on L2 locality (75 % hit rate
// either way); what the smaller super-blocks save are the workgroups that find their tile beyond the matrix edge or
// above the diagonal and exit.
constexpr int SB = 4;
// far part of an outer update: rows/cols from row0 on (trailing) and rows from row0 on x cols < kend (S)
// (s_part: the factor accumulates S in this sweep - see s_in_sweep)
__device__ __host__ __forceinline__ long long outer_tiles(int P, int kend, int row0, bool s_part) {
  const long long r = P - row0;
  if (r <= 0) return 0;
  const long long nsb = (r + SB - 1) / SB, ncb = s_part ? (kend + SB - 1) / SB : 0;
  return (nsb * (nsb + 1) / 2 + nsb * ncb) * (SB * SB);
}
// a "strip" of an outer update: trailing tiles of the block columns [lo, hi) and S tiles of the block rows
// [lo, hi) (S columns < kend).  [kend, kend + 4) is what the next panel's chain touches ("near").
__device__ __host__ __forceinline__ long long strip_tiles(int P, int kend, int lo, int hi, bool s_part) {
  const int re = hi < P ? hi : P;
  long long n = 0;
  for (int j = lo; j < re; ++j) n += P - j;
  if (s_part && re > lo) n += (long long)(re - lo) * kend;
  return n;
}

__device__ __forceinline__ void store_sub(gdouble* C, int np, const f64x4 (&acc)[2][2], int wm, int wn, int lane,
                                          int mode) {   // mode 0: C -= acc, 1: C = acc, 2: C += acc
  store_acc(C, np, acc, wm, wn, lane, mode);
}

__device__ __forceinline__ void factor_invert_64(double* Ds, double* Is, int* bad, int pivot_base, double pivot_min, int trace_base = -1);
__device__ __forceinline__ void lds_sub_acc(double* s, const f64x4 (&acc)[2][2], int wm, int wn, int lane);

// (2i)/(4i) inner updates of step k, restricted to the outer panel [.., kend).  The workgroup that
// finishes the next diagonal block A[k+1][k+1] factorises it on the spot and stores X_{k+1,k+1}: step
// k + 1 then starts with its panel solve, one launch (and one 64x64 factorisation latency) less on the
// critical path of every step that is not the first of an outer panel.
__global__ void __launch_bounds__(INV_THREADS)
inner_update_kernel(const InvDev* __restrict__ t, int nf, int k, int k0, int kend) {
  __shared__ double As[NB * LDA], Bs[NB * LDA];
  int f, local;
  if (!locate(t, nf, blockIdx.x, [k, k0, kend](const InvDev& d) { return (int)inner_tiles(d.P, k, k0, kend); }, f, local)) return;
  const InvDev& d = t[f];
  const int np = d.np, P = d.P, ke = kend < P ? kend : P;
  gdouble* W = (gdouble*)d.W;
  gdouble* X = (gdouble*)d.X;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
  f64x4 acc[2][2] = {};
  int j = k + 1;
  bool trailing = false;
  for (; j < ke; ++j) {
    if (local < ke - j) { trailing = true; break; }
    local -= ke - j;
  }
  if (trailing) {
    const int i = j + local;                                       // A[i][j] -= A[i][k] A[j][k]^T
    load_block(W + (long long)i * NB * np + k * NB, np, As);
    if (i != j) load_block(W + (long long)j * NB * np + k * NB, np, Bs);
    __syncthreads();
    mma_64<true>(As, (i != j) ? Bs : As, wm, wn, lane, acc);
    if (i == j && j == k + 1) {
      // next diagonal block: A_jj - acc is final.  Factorise it here (As/Bs become the two work tiles).
      __shared__ int bad;
      __syncthreads();                                             // all waves are done reading As
      if (threadIdx.x == 0) bad = 0;
      load_block(W + (long long)j * NB * np + j * NB, np, As);
      __syncthreads();
      {
        const int c16 = lane & 15, rq = lane >> 4;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              As[(32 * wm + 16 * m + rq + 4 * q) * LDA + 32 * wn + 16 * n + c16] -= acc[m][n][q];
      }
      __syncthreads();
      factor_invert_64(As, Bs, &bad, j * NB, d.pivot_min);
      store_block(X + (long long)j * NB * np + j * NB, np, Bs);
      if (threadIdx.x == 0 && bad != 0) atomicCAS(d.info, 0, bad);
    } else {
      store_sub(W + (long long)i * NB * np + j * NB, np, acc, wm, wn, lane, 0);
    }
  } else {
    const int w = k - k0 + 1;                                      // S[i][jj] (+)= C[i][k] X[k][jj], k0 <= jj <= k
    const int a = local / w, jj = k0 + local - a * w;
    const int i = k + 1 + a;
    load_block(W + (long long)i * NB * np + k * NB, np, As);
    load_block(X + (long long)k * NB * np + jj * NB, np, Bs);
    __syncthreads();
    mma_64<false>(As, Bs, wm, wn, lane, acc);
    store_sub(X + (long long)i * NB * np + jj * NB, np, acc, wm, wn, lane, jj == k ? 1 : 2);
  }
}

// (2o)/(4o) once per outer panel [k0, kend): everything beyond the panel, K = (kend - k0) * 64.
// K advances in short steps through two small LDS operand tiles; the next step is fetched into registers while
// the MFMAs of the current one run.  What this latency-bound streaming kernel is short of is workgroups per CU
// (loads in flight, somebody else's MFMAs to run while one workgroup waits), not L2 bandwidth or tile size:
// K step 64 (two 33 KB tiles, 2 per CU) -> 32 (17 KB, 4 per CU): 14.8 -> 12.8 ms for the ResNet-50 factors at
// the time; 32 -> 16 (8.7 KB tiles, 96 VGPRs, 5 per CU): 9.6 -> 9.25 ms.
constexpr int OKS = 16;                // K step (elements); with 1024 threads a step is one load per thread
constexpr int OPA = OKS + 1;           // LDS pitch of a [64][OKS] operand (rows K-contiguous)
// One 64x64 output tile; loads are SGPR base + 32-bit lane offset.  Shared by the outer updates and the panel
// products: every kernel built on it has the same footprint (96 VGPRs, 17.4 KB LDS), so a workgroup of one fits
// exactly the slot a retiring workgroup of another frees.
struct TileJob {                        // everything wave-uniform
  const gbyte* a0;                      // A: rows K-contiguous, element (r, ke) at a0 + (r * np + ke) * 8
  const gbyte* b0;                      // B: bt ? rows K-contiguous like A : element (ke, c) at b0 + (ke * np + c) * 8
  gdouble* C;                           // output tile, pitch np
  int np, ke0, ke1, mode;               // K range in elements; mode of store_acc
  bool bt, same;                        // same: B is A (diagonal tile of a symmetric update)
  gfloat* C32 = nullptr;                // also store the tile as fp32 here (pitch np)
};
// fp32 twin of store_acc: tile (pitch ld floats) = sign * acc
__device__ __forceinline__ void store_acc_f32(gfloat* __restrict__ C, int ld, const f64x4 (&acc)[2][2], int wm, int wn, int lane,
                                              float sign) {
  const int c16 = lane & 15, rq = lane >> 4;
  const unsigned voff = (unsigned)(((long long)(32 * wm + rq) * ld + 32 * wn + c16) * 4);
  gbyte* base = (gbyte*)C;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(gfloat*)(base + ((long long)(16 * m + 4 * q) * ld + 16 * n) * 4 + voff) = sign * (float)acc[m][n][q];
}
// WV x WV waves per workgroup.  WV = 2 (256 threads, a 32x32 quadrant per wave) is the throughput form: <= 128 VGPRs,
// four workgroups per CU hide each other's latencies.  When a launch has fewer workgroups than the GPU has CUs -
// the near update and the panel product of a single large factor, both on the chain's critical path - nobody
// hides anything and a lone workgroup is bound by its own serial issue (LDS operand read -> 4 dependent MFMAs
// of 64 cycles per group of four K elements: 45 us for a K = 256 tile; keeping four K steps of loads in flight did not
// change that).  WV = 4 (1024 threads, one 16x16 MFMA tile per wave) divides that serial part by four.
// MODE 0: B rows K-contiguous (trailing update / rows-below product), 1: the same with B = A (diagonal tile),
// 2: B as [k][col] (S accumulation / columns-left product).  The K loop is compiled once per mode: with the
// flags tested inside it, every step carried ~13 scalar branches, 9 v_cndmask and 16 v_mov next to its 16 MFMAs,
// and VALU instructions do not overlap with another wave's MFMAs on this chip (tools/micro/f64_mfma_overlap.hip:
// a wave issuing MFMAs back to back starves the VALU work of the other wave of its SIMD completely).
template <int WV, int MODE>
__device__ __forceinline__ void tile_product_impl(const TileJob& o, double* __restrict__ As, double* __restrict__ Bs) {
  constexpr int THREADS = 64 * WV * WV, T = 4 / WV, LPT = NB * OKS / THREADS;   // MFMA tiles per wave edge, loads per thread
  const int np = o.np;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WV, wn = wave % WV;
  constexpr bool trailing = MODE != 2, same = MODE == 1;
  const int r16 = lane & 15, kq = lane >> 4;
  double ra[LPT], rb[LPT];
  const unsigned voff_k = (unsigned)(((long long)(tid / OKS) * np + (tid % OKS)) * 8);   // [rows][OKS k] operands
  constexpr int BE = 8;                                                                // bytes per element of the [k][col] operand
  const unsigned voff_n = (unsigned)(((long long)(tid >> 6) * np + (tid & 63)) * BE);   // [OKS k][64 cols] operand
  const long long step_k = (long long)(THREADS / OKS) * np * 8, step_n = (long long)(THREADS / 64) * np * BE;
  auto fetch = [&](int ke) __attribute__((always_inline)) {                            // ke: first K element of the step
    const gbyte* ga = o.a0 + (long long)ke * 8;
    const gbyte* gb = trailing ? o.b0 + (long long)ke * 8 : o.b0 + (long long)ke * np * BE;
#pragma unroll
    for (int u = 0; u < LPT; ++u) {
      ra[u] = *(const gdouble*)(ga + u * step_k + voff_k);
      if (trailing) rb[u] = same ? 0.0 : *(const gdouble*)(gb + u * step_k + voff_k);
      else rb[u] = *(const gdouble*)(gb + u * step_n + voff_n);
    }
  };
  f64x4 acc[T][T] = {};
  const int ke0 = o.ke0, ke1 = o.ke1;
  fetch(ke0);
  for (int ke = ke0; ke < ke1; ke += OKS) {
#pragma unroll
    for (int u = 0; u < LPT; ++u) {
      const int e = tid + u * THREADS;
      As[(e / OKS) * OPA + (e % OKS)] = ra[u];
      if (trailing) { if (!same) Bs[(e / OKS) * OPA + (e % OKS)] = rb[u]; }
      else Bs[(e >> 6) * LDA + (e & 63)] = rb[u];
    }
    __syncthreads();
    if (ke + OKS < ke1) fetch(ke + OKS);
    const double* Bt = same ? As : Bs;
#pragma unroll 4
    for (int ks = 0; ks < OKS / 4; ++ks) {
      const int k = 4 * ks + kq;
      double a[T], b[T];
#pragma unroll
      for (int m = 0; m < T; ++m) a[m] = As[(16 * T * wm + 16 * m + r16) * OPA + k];
#pragma unroll
      for (int n = 0; n < T; ++n)
        b[n] = trailing ? Bt[(16 * T * wn + 16 * n + r16) * OPA + k] : Bs[k * LDA + 16 * T * wn + 16 * n + r16];
#pragma unroll
      for (int m = 0; m < T; ++m)
#pragma unroll
        for (int n = 0; n < T; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
    }
    __syncthreads();
  }
  if constexpr (WV == 2) {
    store_acc(o.C, np, acc, wm, wn, lane, o.mode);
    if (o.C32 != nullptr) store_acc_f32(o.C32, np, acc, wm, wn, lane, 1.0f);
  } else {
    // one 16x16 tile per wave: rows 16 wm + rq + 4 q, column 16 wn + c16
    const int c16 = lane & 15, rq = lane >> 4;
    if (o.C32 != nullptr) {
      gbyte* b32 = (gbyte*)o.C32;
      const unsigned v32 = (unsigned)(((long long)(16 * wm + rq) * np + 16 * wn + c16) * 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) *(gfloat*)(b32 + (long long)(4 * q) * np * 4 + v32) = (float)acc[0][0][q];
    }
    gbyte* base = (gbyte*)o.C;
    const unsigned voff = (unsigned)(((long long)(16 * wm + rq) * np + 16 * wn + c16) * 8);
    double old[4];
    if (o.mode == 0 || o.mode == 2) {
#pragma unroll
      for (int q = 0; q < 4; ++q) old[q] = *(const gdouble*)(base + (long long)(4 * q) * np * 8 + voff);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double v = acc[0][0][q];
      const double out = o.mode == 0 ? old[q] - v : o.mode == 1 ? v : o.mode == 2 ? old[q] + v : -v;
      *(gdouble*)(base + (long long)(4 * q) * np * 8 + voff) = out;
    }
  }
}

template <int WV>
__device__ __forceinline__ void tile_product_k32(const TileJob& o, double* __restrict__ As, double* __restrict__ Bs) {
  if (!o.bt) tile_product_impl<WV, 2>(o, As, Bs);
  else if (o.same) tile_product_impl<WV, 1>(o, As, Bs);
  else tile_product_impl<WV, 0>(o, As, Bs);
}

template <int WV>
__device__ __forceinline__ void outer_update_body(const InvDev* __restrict__ t, int nf, int k0, int kend, int lo, int hi,
                                                  int strip, int n_items, double* __restrict__ As, double* __restrict__ Bs) {
  bool trailing;
  int i, j, f, local;
  const bool s_only = strip == 2;            // far launch of the S tiles alone (the trailing tiles went to outer_update_dma_kernel)
  if (strip == 1) {
    // strip: block columns / rows [lo, hi)
    if (!locate(t, nf, blockIdx.x, [kend, lo, hi](const InvDev& d) { return (int)strip_tiles(d.P, kend, lo, hi, s_in_sweep(d)); }, f, local))
      return;
    const int P = t[f].P, re = hi < P ? hi : P;
    trailing = false;
    j = lo;
    for (; j < re; ++j) {
      if (local < P - j) { trailing = true; break; }
      local -= P - j;
    }
    if (trailing) {
      i = j + local;
    } else {
      const int a = local / kend;
      i = lo + a; j = local - a * kend;
    }
  } else {
    const int row0 = lo;
    // far part: everything from block row / column lo on.  XCD grouping: workgroups with equal blockIdx % 8 share an XCD; give each XCD whole super-blocks
    int item;
    {
      const int bid = blockIdx.x, xcd = bid & 7, jj = bid >> 3;
      item = ((jj / (SB * SB)) * 8 + xcd) * (SB * SB) + (jj % (SB * SB));
    }
    if (item >= n_items) return;
    if (!locate(t, nf, item, [kend, row0, s_only](const InvDev& d) {
          const int all = (int)outer_tiles(d.P, kend, row0, s_in_sweep(d));
          return s_only ? all - (int)outer_tiles(d.P, kend, row0, false) : all; }, f, local)) return;
    const int r = t[f].P - row0;
    const int nsb = (r + SB - 1) / SB;
    const int n_trail_sb = nsb * (nsb + 1) / 2;
    const int sb = local / (SB * SB) + (s_only ? n_trail_sb : 0), in = local % (SB * SB);
    const int di = in / SB, dj = in - di * SB;
    if (sb < n_trail_sb) {
      int a = 0, tl = sb;
      while (tl > a) { tl -= a + 1; ++a; }
      const int ri = a * SB + di, rj = tl * SB + dj;          // relative to row0
      if (ri >= r || rj > ri) return;
      trailing = true; i = row0 + ri; j = row0 + rj;
    } else {
      const int s2 = sb - n_trail_sb;
      const int ncb = (kend + SB - 1) / SB;
      const int a = s2 / ncb, cb = s2 - a * ncb;
      const int ri = a * SB + di; j = cb * SB + dj;
      if (ri >= r || j >= kend) return;
      trailing = false; i = row0 + ri;
    }
  }
  const InvDev& d = t[f];
  const int np = d.np;
  const gdouble* W = (const gdouble*)d.W;
  gdouble* X = (gdouble*)d.X;
  TileJob o;
  o.np = np;
  o.bt = trailing;
  o.same = trailing && i == j;
  o.a0 = (const gbyte*)(W + (long long)i * NB * np);
  // right-hand side mode: the accumulation runs on Zm (= R - S, so it SUBTRACTS), not on X
  gdouble* Sx = d.Zm != nullptr ? (gdouble*)d.Zm : X;
  o.b0 = trailing ? (const gbyte*)(W + (long long)j * NB * np) : (const gbyte*)(Sx + j * NB);
  o.ke0 = (trailing ? k0 : (j > k0 ? j : k0)) * NB;
  o.ke1 = kend * NB;
  o.C = trailing ? (gdouble*)d.W + (long long)i * NB * np + j * NB : Sx + (long long)i * NB * np + j * NB;
  o.mode = trailing ? 0 : (d.Zm != nullptr ? 0 : (j >= k0 ? 1 : 2));
  tile_product_k32<WV>(o, As, Bs);
}
__global__ void __launch_bounds__(INV_THREADS, 3)
outer_update_kernel(const InvDev* __restrict__ t, int nf, int k0, int kend, int lo, int hi, int strip, int n_items) {
  __shared__ double As[NB * OPA], Bs[NB * OPA > OKS * LDA ? NB * OPA : OKS * LDA];
  if (!strip) KT_BEGIN(254)
  outer_update_body<2>(t, nf, k0, kend, lo, hi, strip, n_items, As, Bs);
  if (!strip) KT_END(255)
}
// ------------------------------------------------------------------------------------------------
// The far update of KFAC.invert's sweeps (trailing tiles only: their inverse is accumulated elsewhere, in fp32) with
// LDS-DMA staging (round 6).  outer_update_kernel stages its operands through registers: per K step of 16 a burst of
// global loads, a pass of LDS stores, two barriers, and operand reads of 8 bytes issued right in front of the MFMAs that
// wait for them (0.51 of the fp64 pipe inside a whole-model sweep, 47 TFLOP/s on one panel of three 4608-wide factors).
// Here (tools/micro/far_update_forms.hip: 58 TFLOP/s on the same panel, the same bits):
//   * buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction, into a double-buffered image [64 rows][8 x 16 B] per operand
//     (16 k per stage); the 16-byte slots XOR-swizzled by (row >> 1) & 7 on the SOURCE side (the DMA writes lane-linear
//     bytes), so that the 16 rows x one slot of an operand read cover all 64 banks;
//   * operands by ds_read_b128: two k per read; lane quarter kq multiplies k = 8 h + 2 kq + d at MFMA (h, d) of a stage -
//     a product only needs both operands to agree on the order of its k;
//   * one barrier per stage, no staging registers, no LDS stores: 66 registers, 32 KiB of LDS, four workgroups per CU.
// Tiles, super-block order and the read-modify-write epilogue are outer_update_body's; fp64 sums of the same products in
// another order (the K order inside a stage differs): results agree to rounding, not bit for bit, with the register-staged
// form - every launch form of a sweep uses this kernel for its far updates, so sharded and unsharded runs still agree.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) char lds_char;
typedef double f64x2 __attribute__((ext_vector_type(2)));
namespace fard {
constexpr int ROW_B = 128, TILE_B = NB * ROW_B, KS = 16;
}
__global__ void __launch_bounds__(INV_THREADS, 4)
outer_update_dma_kernel(const InvDev* __restrict__ t, int nf, int k0, int kend, int row0, int n_items) {
  using namespace fard;
  __shared__ __attribute__((aligned(1024))) char smem[4 * TILE_B];     // [A buf0][A buf1][B buf0][B buf1]
  lds_char* lds = (lds_char*)smem;
  int item;
  {
    const int bid = blockIdx.x, xcd = bid & 7, jj = bid >> 3;           // whole super-blocks per XCD (see outer_update_body)
    item = ((jj / (SB * SB)) * 8 + xcd) * (SB * SB) + (jj % (SB * SB));
  }
  if (item >= n_items) return;
  int f, local;
  if (!locate(t, nf, item, [kend, row0](const InvDev& d) { return (int)outer_tiles(d.P, kend, row0, false); }, f, local)) return;
  const InvDev& d = t[f];
  int i, j;
  {
    const int r = d.P - row0;
    const int sb = local / (SB * SB), in = local - sb * (SB * SB);
    const int di = in / SB, dj = in - di * SB;
    int a = 0, tl = sb;
    while (tl > a) { tl -= a + 1; ++a; }
    const int ri = a * SB + di, rj = tl * SB + dj;                      // relative to row0
    if (ri >= r || rj > ri) return;
    i = row0 + ri; j = row0 + rj;
  }
  const int np = d.np;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
  const int r16 = lane & 15, kq = lane >> 4;
  const bool same = i == j;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)d.W, 0, (unsigned)((long long)np * np * 8), 0x00020000);
  // DMA lane geometry: piece p (8 rows x 128 B) of an operand = rows 8 p ..; this wave moves pieces `wave` and `wave + 4`
  // (32 rows apart: the same swizzle key)
  const int drow = 8 * wave + (lane >> 3);
  const int voff = (drow * np) * 8 + (((lane & 7) ^ ((drow >> 1) & 7)) << 4);
  const int soff_a = (i * NB * np) * 8, soff_b = (j * NB * np) * 8, half_b = 32 * np * 8;
  auto issue = [&](int ke, unsigned buf) {
    const int kb = ke * 8;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds + buf + wave * 1024), 16, voff, soff_a + kb, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds + buf + (wave + 4) * 1024), 16, voff, soff_a + half_b + kb, 0, 0);
    if (!same) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds + 2 * TILE_B + buf + wave * 1024), 16, voff, soff_b + kb, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds + 2 * TILE_B + buf + (wave + 4) * 1024), 16, voff, soff_b + half_b + kb, 0, 0);
    }
  };
  // operand addresses: block m (16 rows), half h: row * 128 + ((kq + 4 h) ^ key) * 16
  unsigned addr_a[2], addr_b[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int ra = 32 * wm + 16 * m + r16, rb = 32 * wn + 16 * m + r16;
    addr_a[m] = ra * ROW_B + ((kq ^ ((ra >> 1) & 7)) << 4);
    addr_b[m] = (same ? 0 : 2 * TILE_B) + rb * ROW_B + ((kq ^ ((rb >> 1) & 7)) << 4);
  }
  f64x4 acc[2][2] = {};
  const int ke0 = k0 * NB, n_st = (kend - k0) * NB / KS;
  issue(ke0, 0);
  for (int st = 0; st < n_st; ++st) {
    __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0): this wave's pieces of stage st have landed
    __syncthreads();                           // everyone's have; everyone is done reading the other buffer
    const unsigned buf = (st & 1) * TILE_B;
    if (st + 1 < n_st) issue(ke0 + KS * (st + 1), TILE_B - buf);
    f64x2 av[2][2], bv[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        av[m][h] = *reinterpret_cast<const __attribute__((address_space(3))) f64x2*>(lds + buf + (addr_a[m] ^ (h << 6)));
        bv[m][h] = *reinterpret_cast<const __attribute__((address_space(3))) f64x2*>(lds + buf + (addr_b[m] ^ (h << 6)));
      }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int dd = 0; dd < 2; ++dd)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m][h][dd], bv[n][h][dd], acc[m][n], 0, 0, 0);
  }
  store_acc((gdouble*)d.W + (long long)i * NB * np + j * NB, np, acc, wm, wn, lane, 0);
}

__global__ void __launch_bounds__(1024)
outer_update_wide_kernel(const InvDev* __restrict__ t, int nf, int k0, int kend, int lo, int hi, int strip, int n_items) {
  __shared__ double As[NB * OPA], Bs[NB * OPA > OKS * LDA ? NB * OPA : OKS * LDA];
  KT_BEGIN(252)
  outer_update_body<4>(t, nf, k0, kend, lo, hi, strip, n_items, As, Bs);
  KT_END(253)
}

// ------------------------------------------------------------------------------------------------
// 64x64 block in LDS: Ds <- L (lower Cholesky factor, zeros above), Is <- L^-1.  Blocked by 16:
//   panel factorisation by ONE wave with the 64 rows in registers (lane = row) and v_readlane
//   broadcasts: no barriers inside a panel; trailing updates and the block forward substitution of the
//   inverse are 16x16x16 f64-MFMA tile products.  ~10 barriers in total instead of ~200.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double readlane_f64(double v, int src_lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

// acc += A (16x16, element (i,k) at Ap[i*lda + k]) * B; BT: B element (k,j) at Bp[j*ldb + k], else Bp[k*ldb + j]
template <bool BT>
__device__ __forceinline__ void mma16(const double* Ap, int lda, const double* Bp, int ldb, int lane, f64x4& acc) {
  const int r16 = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int kk = 4 * ks + kq;
    const double a = Ap[r16 * lda + kk];
    const double b = BT ? Bp[r16 * ldb + kk] : Bp[kk * ldb + r16];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
}

// 1 / sqrt(x) for a positive finite double: hardware estimate (v_rsq_f64, ~2^-26) + two Newton steps.
// The library sqrt() and division it replaces sit on the serial pivot chain of every column (~400 cycles
// there, ~100 here); the result is within an ulp or two of the correctly rounded one, five orders of
// magnitude below the parity tolerance.
__device__ __forceinline__ double rsqrt_pos(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  y = y * fma(-hx * y, y, 1.5);
  y = y * fma(-hx * y, y, 1.5);
  return y;
}

// 1 / x for a positive finite double: hardware estimate (v_rcp_f64, ~2^-23) + one cubic step
// y (1 + e + e^2), e = 1 - x y  (error e^3 ~ 2^-69): three dependent instructions behind the estimate.
__device__ __forceinline__ double rcp_pos(double x) {
  const double y = __builtin_amdgcn_rcp(x);
  const double e = fma(-x, y, 1.0);
  return fma(y, fma(e, e, e), y);
}

// `pivot_min`: a pivot at or below it is reported as "not positive definite" (0 for the estimators; a caller that looks for
// the numerical rank of a Gram matrix - the low-rank eigensolver, ops.eigh - passes its threshold through curv_cholinv_desc)
__device__ __forceinline__ void factor_invert_64(double* Ds, double* Is, int* bad, int pivot_base, double pivot_min, int trace_base) {
  (void)trace_base;
  __shared__ double Sc[4][16 * 17];
  __shared__ double dinv_s[NB];        // 1 / L_cc: the triangular inverse divides by the same pivots
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, kq = lane >> 4;
  // The inverse follows the factorisation ONE PANEL BEHIND, on the waves that idle while wave 0 runs a panel's
  // column loop: during panel p wave 1 inverts diagonal block p - 1 (forward substitution) and completes row p - 1 of
  // X; during the last panel waves 2 and 3 also pre-accumulate what row 3 of X can already use.  Behind the last panel
  // only X_33, one more MFMA term and one product per block remain (3.8 k instead of 7.9 k cycles).
  // lane j < 16 of the calling wave owns column j of X_bb.  Right-looking: as soon as x[q] is known every later row's
  // partial sum takes its term, so the chain from x[q] to x[q + 1] is one multiply-add and one multiply; the column
  // of L a step needs is fetched one step ahead.
  auto diag_inverse = [&](int b) {
    if (lane < 16) {
      const double* Lb = Ds + 16 * b * LDA + 16 * b;
      double x[16], sum[16], col[16], dv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) { sum[r] = 0.0; col[r] = r > 0 ? Lb[r * LDA] : 0.0; dv[r] = dinv_s[16 * b + r]; }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        x[q] = ((q == lane ? 1.0 : 0.0) - sum[q]) * dv[q];
        double nxt[16];
#pragma unroll
        for (int r = q + 2; r < 16; ++r) nxt[r] = Lb[r * LDA + q + 1];
#pragma unroll
        for (int r = q + 1; r < 16; ++r) sum[r] += col[r] * x[q];
#pragma unroll
        for (int r = q + 2; r < 16; ++r) col[r] = nxt[r];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) Is[(16 * b + r) * LDA + 16 * b + lane] = x[r];
    }
  };
  // what one wave writes to LDS it may read back (other lanes) without a workgroup barrier: LDS operations of a wave
  // complete in order; the fences only keep the compiler from reordering them
  auto wave_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // acc += sum_{k = k_lo}^{k_hi - 1} L_ik X_kb        (16x16 blocks)
  auto block_sum = [&](int i, int b, int k_lo, int k_hi, f64x4& acc) {
    for (int kk = k_lo; kk < k_hi; ++kk)
      mma16<false>(Ds + 16 * i * LDA + 16 * kk, LDA, Is + 16 * kk * LDA + 16 * b, LDA, lane, acc);
  };
  // X_ib = -X_ii acc, staged through this wave's private scratch
  auto finish_block = [&](int i, int b, const f64x4& acc) {
#pragma unroll
    for (int r = 0; r < 4; ++r) Sc[wave][(kq + 4 * r) * 17 + r16] = acc[r];
    wave_sync();
    f64x4 out = {0.0, 0.0, 0.0, 0.0};
    mma16<false>(Is + 16 * i * LDA + 16 * i, LDA, Sc[wave], 17, lane, out);
#pragma unroll
    for (int r = 0; r < 4; ++r) Is[(16 * i + kq + 4 * r) * LDA + 16 * b + r16] = -out[r];
  };
  f64x4 y3 = {0.0, 0.0, 0.0, 0.0};         // this wave's partial sum for row 3 of X (waves 1..3 -> blocks 2, 0, 1)
  const int b3 = wave == 2 ? 0 : wave == 3 ? 1 : 2;
  // (unrolled: 35 KB of straight-line code, and the first factorisation of a launch runs it from a cold instruction cache -
  // 18 us instead of 8, tools/sq_trace.py.  Rolled (-DCURV_FI_UNROLL=1: 13 KB; chol_square_kernel 64 -> 42 KB) the steady state
  // loses more than the cold start gains: one 4608^2 2.48 -> 2.57 ms, 2304^2 1.08 -> 1.19, three 4608^2 and a whole model equal)
#ifndef CURV_FI_UNROLL
#define CURV_FI_UNROLL 4
#endif
#pragma unroll CURV_FI_UNROLL
  for (int p = 0; p < 4; ++p) {
    const int c0 = 16 * p;
    if (wave != 0) {
      if (p == 0) {
        // all of Is is cleared once (its tiles above the diagonal stay zero; the others are overwritten)
        for (int e = tid - 64; e < NB * NB; e += MMA_THREADS - 64) Is[(e >> 6) * LDA + (e & 63)] = 0.0;
      } else if (wave == 1) {
        diag_inverse(p - 1);
        wave_sync();
        for (int b = 0; b < p - 1; ++b) {             // row p - 1 of X: blocks left of the diagonal
          f64x4 acc = {0.0, 0.0, 0.0, 0.0};
          block_sum(p - 1, b, b, p - 1, acc);
          finish_block(p - 1, b, acc);
          wave_sync();
        }
      } else if (p == 3) {
        block_sum(3, b3, b3, 2, y3);                   // wave 2: L_30 X_00 + L_31 X_10; wave 3: L_31 X_11
      }
    }
    if (wave == 0) {
      // The 16 columns of the panel as ONE straight-line block (no store, no branch and no exec-mask change between
      // the columns), in root-free form: the columns stay UNSCALED (u_c = L_c sqrt(d_c)) until the panel is done, so that
      // the serial recurrence runs over the pivots alone,
      //     d_{c+1} = (a_{c+1,c+1} - earlier columns) - u_c[c+1]^2 / d_c,
      // i.e. per column one multiply-add on the pivot's lane, one broadcast, and a reciprocal (hardware estimate + one
      // cubic correction step: 4 dependent instructions).  The rank-1 updates, the squares the next pivot needs and the
      // bookkeeping hang off that chain and fill its latency; the pivots are tested and their 16 square roots taken at
      // once at the end (lane c owns d_c), and the columns scaled by them.  The form with one rsqrt per column had 20 dependent instructions per
      // column, six of them scalar (the pivot test selected the rsqrt's input): ~350 cycles per column.
      double row[16];
      int mine_lo = 0, mine_hi = 0x3ff00000;      // lane c < 16 collects the pivot d_c of its column
#pragma unroll
      for (int c = 0; c < 16; ++c) row[c] = Ds[lane * LDA + c0 + c];
      double d = readlane_f64(row[0], c0), rd = rcp_pos(d);
      // one column; the column index is a compile-time constant (v_writelane takes its lane as an immediate)
      auto column = [&](auto cc) {
        constexpr int c = decltype(cc)::value;
        asm("v_writelane_b32 %0, %1, %2" : "+v"(mine_lo) : "s"(__double2loint(d)), "n"(c));
        asm("v_writelane_b32 %0, %1, %2" : "+v"(mine_hi) : "s"(__double2hiint(d)), "n"(c));
        if (c + 1 < 16) {
          // the chain: next pivot from this column's reciprocal (only lane c0 + c + 1 of pv is meaningful)
          const double pv = fma(-(row[c] * row[c]), rd, row[c + 1 < 16 ? c + 1 : c]);
          d = readlane_f64(pv, c0 + c + 1);
        }
        const double t = row[c] * rd;                      // multipliers u_c[q] / d_c, lane q
        if (c + 1 < 16) rd = rcp_pos(d);
        // broadcasts issued one update ahead: two instructions between a v_readlane and the use of its SGPR
        double m_next = c + 1 < 16 ? readlane_f64(t, c0 + c + 1) : 0.0;
#pragma unroll
        for (int q = c + 1; q < 16; ++q) {
          const double m = m_next;
          if (q + 1 < 16) m_next = readlane_f64(t, c0 + q + 1);
          row[q] -= row[c] * m;
        }
        __builtin_amdgcn_sched_barrier(0);
      };
#define CURV_COL(C) column(std::integral_constant<int, C>{});
      CURV_COL(0) CURV_COL(1) CURV_COL(2) CURV_COL(3) CURV_COL(4) CURV_COL(5) CURV_COL(6) CURV_COL(7)
      CURV_COL(8) CURV_COL(9) CURV_COL(10) CURV_COL(11) CURV_COL(12) CURV_COL(13) CURV_COL(14) CURV_COL(15)
#undef CURV_COL
      // pivot test, once per panel (false for NaN too; a failed factor turns into NaN / garbage and is reported)
      const double mine = __hiloint2double(mine_hi, mine_lo);
      const unsigned long long failed = __ballot(lane < 16 && !(mine > pivot_min && mine < 1.0e300));
      const int first_bad = failed != 0 ? pivot_base + c0 + __ffsll((long long)failed) : 0;
      // L_c = u_c / sqrt(d_c): the reciprocal roots of all 16 pivots at once, handed round through dinv_s
      const double rs = rsqrt_pos(lane < 16 ? mine : 1.0);
      if (lane < 16) dinv_s[c0 + lane] = rs;
      wave_sync();
#pragma unroll
      for (int c = 0; c < 16; ++c) Ds[lane * LDA + c0 + c] = row[c] * dinv_s[c0 + c];
      if (first_bad != 0 && lane == 0 && *bad == 0) *bad = first_bad;
    }
    __syncthreads();
    // trailing update of the 16x16 tiles (ti, tj), p < tj <= ti: T -= A21_ti A21_tj^T  (K = 16)
    {
      int tcount = 0;
      for (int ti = p + 1; ti < 4; ++ti)
        for (int tj = p + 1; tj <= ti; ++tj, ++tcount) {
          if ((tcount & 3) != wave) continue;
          f64x4 acc = {0.0, 0.0, 0.0, 0.0};
          mma16<true>(Ds + 16 * ti * LDA + c0, LDA, Ds + 16 * tj * LDA + c0, LDA, lane, acc);
          double* C = Ds + 16 * ti * LDA + 16 * tj;
#pragma unroll
          for (int r = 0; r < 4; ++r) C[(kq + 4 * r) * LDA + r16] -= acc[r];
        }
    }
    __syncthreads();
    FIT(p)
  }
  // behind the last panel: X_33 on wave 0 while the others add the terms of row 3 that needed row 2 of X
  if (wave == 0) {
    diag_inverse(3);
  } else {
    block_sum(3, b3, 2, 3, y3);                        // + L_32 X_2b
#pragma unroll
    for (int r = 0; r < 4; ++r) Sc[wave][(kq + 4 * r) * 17 + r16] = y3[r];
  }
  __syncthreads();
  if (wave != 0) {
    f64x4 out = {0.0, 0.0, 0.0, 0.0};
    mma16<false>(Is + 16 * 3 * LDA + 16 * 3, LDA, Sc[wave], 17, lane, out);
#pragma unroll
    for (int r = 0; r < 4; ++r) Is[(16 * 3 + kq + 4 * r) * LDA + 16 * b3 + r16] = -out[r];
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// (1a) diagonal block of step k: one workgroup per factor factorises A_kk in LDS and stores
//      X_kk = L_kk^-1 (the only thing steps (1b) and (3) need; L_kk itself is never used again)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(INV_THREADS)
chol_diag_kernel(const InvDev* __restrict__ t, int nf, int k) {
  __shared__ double Ds[NB * LDA];     // A_kk -> L_kk (lower)
  __shared__ double Is[NB * LDA];     // L_kk^-1 (lower, zeros above)
  __shared__ int bad;
  int f, local;
  if (!locate(t, nf, blockIdx.x, [k](const InvDev& d) { return d.P > k ? 1 : 0; }, f, local)) return;
  const InvDev& d = t[f];
  const int np = d.np, tid = threadIdx.x;
  gdouble* W = (gdouble*)d.W;
  if (tid == 0) bad = 0;
  load_block(W + (long long)k * NB * np + k * NB, np, Ds);
  __syncthreads();
  factor_invert_64(Ds, Is, &bad, k * NB, d.pivot_min);
  store_block((gdouble*)d.X + (long long)k * NB * np + k * NB, np, Is);
  if (tid == 0 && bad != 0) atomicCAS(d.info, 0, bad);
}

// ------------------------------------------------------------------------------------------------
// (1b) panel solve  A[i][k] <- A[i][k] X_kk^T            for i > k   (P - k - 1 tiles per factor), and
// (3)  row k of the inverse  X[k][j] = -X_kk S[k][j]      for j < k   (k tiles per factor),
//      one launch: both need nothing but X_kk from (1a); every workgroup owns the tile it rewrites
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(INV_THREADS)
chol_panel_kernel(const InvDev* __restrict__ t, int nf, int k, int k0, int kend) {
  __shared__ double As[NB * LDA], Bs[NB * LDA];
  int f, local;
  if (!locate(t, nf, blockIdx.x,
              [k, k0, kend](const InvDev& d) { return d.P > k ? (kend < d.P ? kend : d.P) - k0 - 1 : 0; }, f, local))
    return;
  const InvDev& d = t[f];
  const int np = d.np, n_solve = (kend < d.P ? kend : d.P) - k - 1;
  gdouble* W = (gdouble*)d.W;
  gdouble* X = (gdouble*)d.X;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
  f64x4 acc[2][2] = {};
  if (local < n_solve) {
    const int i = k + 1 + local;
    load_block(W + (long long)i * NB * np + k * NB, np, As);            // A_ik as [row][kk]
    load_block(X + (long long)k * NB * np + k * NB, np, Bs);            // X_kk rows are the K-contiguous operand
    __syncthreads();
    mma_64<true>(As, Bs, wm, wn, lane, acc);
    store_acc(W + (long long)i * NB * np + k * NB, np, acc, wm, wn, lane, 1);
  } else {
    const int j = k0 + local - n_solve;
    load_block(X + (long long)k * NB * np + k * NB, np, As);            // X_kk as [row][kk]
    load_block(X + (long long)k * NB * np + j * NB, np, Bs);            // S_kj as [kk][col]
    __syncthreads();
    mma_64<false>(As, Bs, wm, wn, lane, acc);
    store_acc(X + (long long)k * NB * np + j * NB, np, acc, wm, wn, lane, 3);
  }
}

// ------------------------------------------------------------------------------------------------
// The block square of an outer panel in ONE launch (few factors: the chain of a large factor, a layer-sharded
// rank).  The per-step launches above put three kernel boundaries between two 64x64 factorisations
// (panel solve, inner update + next factorisation); here the workgroups of a factor wait for each other
// through flags in global memory instead:
//   runner (row 0's workgroup) walks the diagonal: X_rr = factor-and-invert of A''_rr, then - with X_rr still in
//     LDS - the one solve and the one update the next diagonal block is waiting for,
//       L_{r+1,r} = T_{r+1,r} X_rr^T,    A'_{r+1,r+1} = A''_{r+1,r+1} - L_{r+1,r} L_{r+1,r}^T;
//   helper r (rows 1..3) works left-looking and one step ahead of the runner: its row's other tiles
//       L_{r,m} = (A_{r,m} - sum_{j<m} L_{r,j} L_{m,j}^T) X_mm^T  (m <= r - 2),   T_{r,r-1},   A''_rr,
//     and COLUMN r - 1 of the square's inverse, X_{i,j} = -X_ii sum_{m=j}^{i-1} L_{i,m} X_{m,j}.
// Tiles that cross workgroups are written and read with agent-coherent accesses (sc1: no cache maintenance); a
// hand-off (store, flag, poll, load) takes ~5 us (tools/micro/wg_handoff.hip), which is why nothing on the runner's
// path waits for one that was not posted a step earlier.  The helpers wait for workgroups with a lower block index of
// the launch (dispatched before them); the RUNNER (the lowest index of its factor) also waits for flags its helpers
// post (SQ_FT), i.e. for workgroups with a HIGHER index: forward progress needs all workgroups of a factor (at most 4,
// 109 KB of LDS each: one per CU) resident at the same time, which a whole MI355X always grants but a CU-masked stream,
// a partitioned device or heavy contention from other streams may not.  Every wait is therefore bounded: a lost flag
// ends in status word -1 ("inter-workgroup hand-off timed out", ops.check_chol_info), not in a hung queue, and
// CURV_LATENCY_MAX=0 selects the per-step launches that need no co-residency.
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ void load_block_coh(const gdouble* __restrict__ g, int ld, double* __restrict__ s) {
  const int tid = threadIdx.x, r0 = tid >> 6, c = tid & 63;
  const long long step = 4ll * ld;
  const double* base = (const double*)g + (long long)r0 * ld + c;
  double v[NB * NB / MMA_THREADS];
#pragma unroll
  for (int u = 0; u < NB * NB / MMA_THREADS; ++u)
    v[u] = __hip_atomic_load(base + u * step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int u = 0; u < NB * NB / MMA_THREADS; ++u) s[(r0 + 4 * u) * LDA + c] = v[u];
}
__device__ __forceinline__ void store_block_coh(gdouble* __restrict__ g, int ld, const double* __restrict__ s) {
  const int tid = threadIdx.x, r0 = tid >> 6, c = tid & 63;
  const long long step = 4ll * ld;
  double* base = (double*)g + (long long)r0 * ld + c;
#pragma unroll
  for (int u = 0; u < NB * NB / MMA_THREADS; ++u)
    __hip_atomic_store(base + u * step, s[(r0 + 4 * u) * LDA + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// C = sign * acc, coherent
__device__ __forceinline__ void store_acc_coh(gdouble* __restrict__ C, int ld, const f64x4 (&acc)[2][2], int wm, int wn,
                                              int lane, double sign) {
  const int c16 = lane & 15, rq = lane >> 4;
  double* base = (double*)C + (long long)(32 * wm + rq) * ld + 32 * wn + c16;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        __hip_atomic_store(base + (long long)(16 * m + 4 * q) * ld + 16 * n, sign * acc[m][n][q], __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}
// LDS tile [row][col] -= acc (wave quadrant layout)
__device__ __forceinline__ void lds_sub_acc(double* s, const f64x4 (&acc)[2][2], int wm, int wn, int lane) {
  const int c16 = lane & 15, rq = lane >> 4;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) s[(32 * wm + 16 * m + rq + 4 * q) * LDA + 32 * wn + 16 * n + c16] -= acc[m][n][q];
}
// mma_64 (mma64.h) for a workgroup that is alone on its CU: every operand of the 64-deep product is read from LDS
// first (a lane owns 16 consecutive k of its rows: k = 16 (lane >> 4) + ks, pairs of ds_read2_b64), then the 64 MFMAs
// run back to back - 1.2 us instead of 2.8 for mma_64, whose four-step groups wait for their LDS reads one after the
// other when no second wave hides them.
template <bool BT>
__device__ __forceinline__ void mma_64_pre(const double* __restrict__ As, const double* __restrict__ Bs, int wm, int wn, int lane,
                                           f64x4 (&acc)[2][2]) {
  const int r16 = lane & 15, kq = lane >> 4;
  double a[2][16], b[2][16];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) a[m][ks] = As[(32 * wm + 16 * m + r16) * LDA + 16 * kq + ks];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
      b[n][ks] = BT ? Bs[(32 * wn + 16 * n + r16) * LDA + 16 * kq + ks] : Bs[(16 * kq + ks) * LDA + 32 * wn + 16 * n + r16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m][ks], b[n][ks], acc[m][n], 0, 0, 0);
}
__device__ __forceinline__ void zero_acc(f64x4 (&acc)[2][2]) {
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[m][n] = f64x4{0.0, 0.0, 0.0, 0.0};
}
// everything this workgroup stored is visible device-wide, then the flag
__device__ __forceinline__ void sq_signal(int* flag, int stamp) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sq_wait(int* flag, int stamp, int* lost) {
  if (threadIdx.x == 0) {
    int n = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < stamp) {
      __builtin_amdgcn_s_sleep(2);
      if (++n > SQ_SPIN_LIMIT || *lost) { *lost = 1; break; }
    }
  }
  __syncthreads();
}

__global__ void __launch_bounds__(INV_THREADS)
chol_square_kernel(const InvDev* __restrict__ t, int nf, int k0, int kend, int* __restrict__ flags, int stamp) {
  __shared__ double Ds[NB * LDA], Is[NB * LDA], Bf[NB * LDA];
  __shared__ int bad, lost;
  int f, r;
  if (!locate(t, nf, blockIdx.x, [k0, kend](const InvDev& d) { return d.P > k0 ? (kend < d.P ? kend : d.P) - k0 : 0; }, f, r))
    return;
  const InvDev& d = t[f];
  const int np = d.np, tid = threadIdx.x;
  const int nbf = (kend < d.P ? kend : d.P) - k0;
  gdouble* W = (gdouble*)d.W;
  gdouble* X = (gdouble*)d.X;
  int* fl = flags + (long long)f * SQ_FLAGS;
  const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  auto Wt = [&](int i, int j) { return W + (long long)(k0 + i) * NB * np + (k0 + j) * NB; };   // tiles of the square
  auto Xt = [&](int i, int j) { return X + (long long)(k0 + i) * NB * np + (k0 + j) * NB; };
  if (tid == 0) { bad = 0; lost = 0; }
  __syncthreads();
  f64x4 acc[2][2];
  if (r == 0) {
    // ---------------- runner ----------------
    // (ONE call site of factor_invert_64: inlined twice its 35 KB made this kernel 100 KB of code, and the first two
    // factorisations of every launch ran 11 us instead of 8 - instruction-cache misses, 64 KB shared by two CUs)
    SQT(0)
#ifdef CURV_SQ_TRACE
    if (tid == 0 && f == 0 && stamp == CURV_SQ_TRACE + 1) g_sq_trace[238] = wall_clock64();
#endif
    load_block(Wt(0, 0), np, Ds);
    if (nbf > 1) load_block(Wt(1, 0), np, Bf);                    // T_{1,0} = A_{1,0}: fetched ahead of the first step
    __syncthreads();
    SQT(1)
    for (int q = 0; q < nbf; ++q) {
      if (q >= 1) {
        if (q >= 2) {
          sq_wait(fl + SQ_FT + q, stamp, &lost);
          SQT(8 * q + 0)
          load_block_coh(Wt(q, q - 1), np, Bf);                   // T_{q,q-1}
        }
        load_block_coh(Wt(q, q), np, Ds);                         // A''_qq
        __syncthreads();
        SQT(8 * q + 1)
        zero_acc(acc);
        mma_64_pre<true>(Bf, Is, wm, wn, lane, acc);              // L_{q,q-1} = T X_{q-1,q-1}^T
        SQT(8 * q + 2)
        store_acc_coh(Wt(q, q - 1), np, acc, wm, wn, lane, 1.0);
        __syncthreads();                                          // every wave is done reading Bf
        acc_to_lds(acc, wm, wn, lane, Bf);
        __syncthreads();
        zero_acc(acc);
        mma_64_pre<true>(Bf, Bf, wm, wn, lane, acc);
        lds_sub_acc(Ds, acc, wm, wn, lane);                       // A'_qq
        SQT(8 * q + 3)
        sq_signal(fl + SQ_FLC + q, stamp);                        // L_{q,q-1} is out (also the barrier the step needs)
        SQT(8 * q + 4)
      }
#ifdef CURV_SQ_TRACE
      factor_invert_64(Ds, Is, &bad, (k0 + q) * NB, d.pivot_min, (f == 0 && stamp == CURV_SQ_TRACE) ? 112 + 4 * q : -1);
#else
      factor_invert_64(Ds, Is, &bad, (k0 + q) * NB, d.pivot_min);
#endif
      SQT(q == 0 ? 2 : 8 * q + 5)
      store_block_coh(Xt(q, q), np, Is);
      sq_signal(fl + SQ_FD + q, stamp);
      SQT(q == 0 ? 3 : 8 * q + 6)
    }
    if (tid == 0 && bad != 0) atomicCAS(d.info, 0, bad);
    if (tid == 0 && lost != 0) atomicCAS(d.info, 0, -1);
    return;
  }
  // ---------------- helper of row r ----------------
  double* Af = Ds;                                                // two staging tiles
  SQT(64 * r + 0)
  // phase A: the tiles of this row, T_{r,r-1} and A''_rr (row 1 has none: T_{1,0} = A_{1,0}, A''_11 = A_11).  Every
  // product is formed as soon as its operands exist ("right-looking" inside the row): behind the runner's flag FLC[r-1]
  // - posted one factorisation before the runner wants T_{r,r-1} - only ONE product is left.
  double* Cf = Is;                                                // third tile: this row's newest L_{r,m}
  if (r >= 2) {
    f64x4 accA[2][2], accT0[2][2], accT1[2][2];                   // A''_rr, T_{r,1}, T_{r,2}: running sums
    zero_acc(accA); zero_acc(accT0); zero_acc(accT1);
    auto push = [&](int mp, int m) {                              // accT[mp - 1] += L_{r,m} L_{mp,m}^T, L_{r,m} in Cf
      load_block_coh(Wt(mp, m), np, Bf);
      __syncthreads();
      if (mp == 1) mma_64_pre<true>(Cf, Bf, wm, wn, lane, accT0);
      else mma_64_pre<true>(Cf, Bf, wm, wn, lane, accT1);
      __syncthreads();
    };
    for (int m = 0; m <= r - 2; ++m) {
      SQT(64 * r + 8 + 4 * m)
      load_block(Wt(r, m), np, Af);                               // A_{r,m}: written by an earlier launch
      __syncthreads();
      if (m == 1) lds_sub_acc(Af, accT0, wm, wn, lane);           // T_{r,m}  (m <= r - 2 <= 1)
      sq_wait(fl + SQ_FD + m, stamp, &lost);
      SQT(64 * r + 9 + 4 * m)
      load_block_coh(Xt(m, m), np, Bf);
      __syncthreads();
      zero_acc(acc);
      mma_64_pre<true>(Af, Bf, wm, wn, lane, acc);                // L_{r,m}
      SQT(64 * r + 10 + 4 * m)
      store_acc_coh(Wt(r, m), np, acc, wm, wn, lane, 1.0);
      acc_to_lds(acc, wm, wn, lane, Cf);
      if (m == r - 2) sq_signal(fl + SQ_FLROW + r, stamp);        // (also the barrier behind the LDS writes)
      else __syncthreads();
      mma_64_pre<true>(Cf, Cf, wm, wn, lane, accA);
      __syncthreads();
      for (int mp = m + 1; mp <= r - 1; ++mp) {
        if (m == r - 2 && mp == r - 1) break;                     // needs the runner's L_{r-1,r-2}: last, below
        sq_wait(fl + (mp == m + 1 ? SQ_FLC : SQ_FLROW) + mp, stamp, &lost);
        push(mp, m);
      }
    }
    load_block(Wt(r, r), np, Bf);
    __syncthreads();
    lds_sub_acc(Bf, accA, wm, wn, lane);
    __syncthreads();
    store_block_coh(Wt(r, r), np, Bf);                            // A''_rr, in place
    load_block(Wt(r, r - 1), np, Af);                             // A_{r,r-1}
    sq_wait(fl + SQ_FLC + r - 1, stamp, &lost);                   // (barrier: Bf is free again)
    push(r - 1, r - 2);
    if (r == 2) lds_sub_acc(Af, accT0, wm, wn, lane);
    else lds_sub_acc(Af, accT1, wm, wn, lane);
    __syncthreads();
    store_block_coh(Wt(r, r - 1), np, Af);                        // T_{r,r-1}, in place
    sq_signal(fl + SQ_FT + r, stamp);
    SQT(64 * r + 1)
  }
  // phase C: COLUMN r - 1 of the square's inverse, top to bottom,
  //     X_{i,j} = -X_ii S_i,   S_i = sum_{m=j}^{i-1} L_{i,m} X_{m,j}      (j = r - 1 < i < nbf),
  // whose X_{m,j} are this workgroup's own earlier results: the columns need nothing from each other, and the three
  // tiles of the square's last row are finished by three workgroups side by side once the runner posts X_33.
  {
    const int j = r - 1;
    for (int i = r; i < nbf; ++i) {
      f64x4 S[2][2];
      zero_acc(S);
      for (int m = j; m < i; ++m) {
        // L_{i,m}: the runner's tile (m = i - 1) or helper i's (FLROW[i]: posted with the last of them)
        if (m == i - 1) sq_wait(fl + SQ_FLC + i, stamp, &lost);
        else sq_wait(fl + SQ_FLROW + i, stamp, &lost);
        if (m == j) sq_wait(fl + SQ_FD + j, stamp, &lost);          // X_jj; the others are this workgroup's own
        load_block_coh(Wt(i, m), np, Af);
        load_block_coh(Xt(m, j), np, Bf);
        __syncthreads();
        mma_64_pre<false>(Af, Bf, wm, wn, lane, S);
        __syncthreads();
      }
      SQT(64 * r + 2 + 16 * (i - r))
      acc_to_lds(S, wm, wn, lane, Af);
      sq_wait(fl + SQ_FD + i, stamp, &lost);                        // (also the barrier behind the LDS writes)
      SQT(64 * r + 3 + 16 * (i - r))
      load_block_coh(Xt(i, i), np, Bf);
      __syncthreads();
      zero_acc(acc);
      mma_64_pre<false>(Bf, Af, wm, wn, lane, acc);
      store_acc_coh(Xt(i, j), np, acc, wm, wn, lane, -1.0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // read back by this workgroup for the next row
      __syncthreads();
      SQT(64 * r + 4 + 16 * (i - r))
    }
  }
  if (tid == 0 && lost != 0) atomicCAS(d.info, 0, -1);
}

// ------------------------------------------------------------------------------------------------
// After the chain of diagonal steps has factorised the nb x nb block square of outer panel [k0, kend)
// (L_sq in W, X_sq = L_sq^-1 in X), everything else of the panel is two triangular products:
//   (1c) rows below the square:   W[i][k0 + c] <- sum_{k <= c} W[i][k0 + k] X_sq[c][k]^T      i >= kend
//   (3c) columns left of it:      X[k0 + r][j] <- - sum_{k <= r} X_sq[r][k] S[k0 + k][j]      j <  k0
// One workgroup per row block / column block walks its nb outputs in DESCENDING order: output c only
// reads inputs k <= c, so the in-place update never overwrites something still needed.  Compared with
// doing these rows step by step (nb panel solves + nb (nb - 1) / 2 rank-64 updates, each a read-modify-
// write of 64x64 tiles) the row panel is read and written once.
// ------------------------------------------------------------------------------------------------
// Jobs of a panel product launch per factor (kind: s_kind): the block rows below the square; for the factors that carry
// their inverse in the sweep also the block columns left of the square (right-hand side mode: the block columns of Zm).
// (The columns left of the square of KFAC.invert's fp32 inverse are xrows32_kernel's, off the chain.)
__device__ __host__ __forceinline__ int panel_jobs(int P, int kind, int k0, int kend) {
  if (P <= k0) return 0;
  return (P > kend ? P - kend : 0) + (kind == 2 ? (kend < P ? kend : P) : kind == 1 ? k0 : 0);
}
template <int WV>
__device__ __forceinline__ void panel_product_body(const InvDev* __restrict__ t, int nf, int k0, int kend,
                                                   double* __restrict__ As, double* __restrict__ Bs) {
  int f, local;
  if (!locate(t, nf, blockIdx.x, [k0, kend](const InvDev& d) { return panel_jobs(d.P, s_kind(d), k0, kend); }, f, local))
    return;
  const InvDev& d = t[f];
  const int np = d.np, nb = (kend < d.P ? kend : d.P) - k0;
  gdouble* W = (gdouble*)d.W;
  gdouble* X = (gdouble*)d.X;
  TileJob o;
  o.np = np;
  o.same = false;
  o.ke0 = 0;
  const int n_below = d.P > kend ? d.P - kend : 0;
  const bool below = local < n_below;
  const int i = kend + local, j = local - n_below;
  o.bt = below;
  o.mode = below ? 1 : 3;
  // right-hand side mode: the columns-left part works on Zm instead of X - rows of the panel <- + X_sq (R - S) - and covers
  // the panel's own column blocks as well (Z_sq = X_sq R_sq: block column j starts at block row j - k0 of the square)
  gdouble* Zm = (gdouble*)d.Zm;
  const bool rhs = !below && Zm != nullptr;
  gdouble* S = rhs ? Zm : X;
  if (rhs) { o.mode = 1; o.ke0 = (j > k0 ? j - k0 : 0) * NB; }
  gfloat* C32 = below ? (gfloat*)d.C32 : nullptr;      // the rows below the square also go to the fp32 copy of C
  for (int c = nb - 1; c >= 0; --c) {      // descending: output c reads only inputs k <= c
    if (rhs && k0 + c < j) break;          // above the diagonal of the solution: never read
    const gdouble* xsq = X + (long long)(k0 + c) * NB * np + k0 * NB;              // block row c of X_sq
    o.a0 = below ? (const gbyte*)(W + (long long)i * NB * np + k0 * NB) : (const gbyte*)xsq;
    o.b0 = below ? (const gbyte*)xsq : (const gbyte*)(S + (long long)k0 * NB * np + j * NB);
    o.C = below ? W + (long long)i * NB * np + (k0 + c) * NB : S + (long long)(k0 + c) * NB * np + j * NB;
    o.C32 = C32 != nullptr ? C32 + (long long)i * NB * np + (k0 + c) * NB : nullptr;
    o.ke1 = (c + 1) * NB;
    tile_product_k32<WV>(o, As, Bs);
  }
}

// Quarter form of the panel product, for calls that are bound by their chain (chol_sweep_group).  A job of
// panel_product_body (one block row below the square, or one block column left of it) is cut into four workgroups of
// four waves: rows 16 q.. of the block row (whose products are independent row by row), or columns 16 q.. of the block
// column.  fp64 MFMAs are slow enough (~100 cycles each) that a lone 1024-thread workgroup is bound by its CU's four
// matrix pipes (ten 64-deep products: 640 MFMAs per SIMD, 28 us); spread over four CUs the same waves have a pipe
// each.  The workgroup's own operand - 16 rows x 256 of W, or 256 x 16 columns of S - is read ONCE into LDS, which also
// removes the in-place ordering between the outputs; the ten tiles of X_sq stream through a second LDS tile, the next
// ones in registers while the MFMAs of the current one run.
constexpr int PQ_PITCH = 4 * NB + 1;                 // [16][257]: rows of the workgroup's slice of W
constexpr int PQ_OWN = 16 * PQ_PITCH > 4 * NB * 17 ? 16 * PQ_PITCH : 4 * NB * 17;   // or [256][17]: columns of S
// `below` is a template parameter: tested inside the K loop it put a scalar branch in front of every operand read
template <bool below>
__device__ __forceinline__ void panel_quarter_body(const InvDev& d, int k0, int nb, int i, int j, int q,
                                                   double* __restrict__ Own, double* __restrict__ Ts) {
  const int np = d.np;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, kq = lane >> 4;
  PQT(240)
  const gbyte* Wb = (const gbyte*)d.W;
  const gbyte* Xb = (const gbyte*)d.X;
  // the workgroup's own operand
  if (below) {
    // W[i*64 + 16 q + r][k0*64 + k], r < 16, k < 64 nb: element (r = tid / 64 + 4 u, k = tid % 64 + 64 v)
    const gbyte* g = Wb + (((long long)i * NB + 16 * q) * np + (long long)k0 * NB) * 8;
    for (int v = 0; v < nb; ++v) {
      double x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) x[u] = *(const gdouble*)(g + ((long long)((tid >> 6) + 4 * u) * np + (tid & 63) + 64 * v) * 8);
#pragma unroll
      for (int u = 0; u < 4; ++u) Own[((tid >> 6) + 4 * u) * PQ_PITCH + (tid & 63) + 64 * v] = x[u];
    }
  } else {
    // S[k0*64 + k][j*64 + 16 q + c], k < 64 nb, c < 16: element (k = tid / 16 + 16 u, c = tid % 16)
    const gbyte* g = Xb + ((long long)k0 * NB * np + (long long)j * NB + 16 * q) * 8;
    for (int v = 0; v < nb; ++v) {
      double x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        x[u] = *(const gdouble*)(g + ((long long)((tid >> 4) + 16 * u + 64 * v) * np + (tid & 15)) * 8);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) Own[((tid >> 4) + 16 * u + 64 * v) * 17 + (tid & 15)] = x[u];
    }
  }
  // stream of X_sq tiles: (c, kb), c = nb - 1 .. 0, kb = 0 .. c; tile rows = block row c of X_sq, K-contiguous.
  // Three tiles are in flight in registers: the square kernel has just written X_sq through to memory, every tile
  // is a miss in this XCD's L2 (~2.5 us), and with one tile in flight the ten of them cost 29 us.
  double rt0[16], rt1[16], rt2[16];
  auto fetch = [&](int c, int kb, double (&rt)[16]) __attribute__((always_inline)) {
    const gbyte* g = Xb + (((long long)(k0 + c) * NB) * np + (long long)(k0 + kb) * NB) * 8;
#pragma unroll
    for (int u = 0; u < 16; ++u) rt[u] = *(const gdouble*)(g + ((long long)((tid >> 6) + 4 * u) * np + (tid & 63)) * 8);
  };
  auto next = [](int& c, int& kb) { if (++kb > c) { --c; kb = 0; } };
  int c = nb - 1, kb = 0;                 // tile being multiplied
  int cf = c, kf = kb;                    // next tile to fetch
  fetch(cf, kf, rt0); next(cf, kf);
  if (cf >= 0) { fetch(cf, kf, rt1); next(cf, kf); }
  if (cf >= 0) { fetch(cf, kf, rt2); next(cf, kf); }
  // four accumulators, every fourth group of four k each: back-to-back MFMAs into ONE accumulator wait for each other
  // (a 64-deep product took 2.9 us that way, 16 dependent fp64 MFMAs), independent ones pipeline
  f64x4 acc[4] = {};
  PQT(241)
  [[maybe_unused]] int pq_n = 0;
  // one tile: registers -> LDS, refill the registers with the tile three ahead, multiply (no register rotation: a
  // move would wait for the loads it copies)
  auto step = [&](double (&rt)[16]) __attribute__((always_inline)) {
    __syncthreads();                                   // the previous tile's readers are done (and Own is written)
#pragma unroll
    for (int u = 0; u < 16; ++u) Ts[((tid >> 6) + 4 * u) * LDA + (tid & 63)] = rt[u];
    __syncthreads();
    if (cf >= 0) { fetch(cf, kf, rt); next(cf, kf); }
#pragma unroll
    for (int ks = 0; ks < NB / 4; ++ks) {
      const int k = 4 * ks + kq;
      // below: out[r][col] += W[r][kb 64 + k] X_sq[c][col][k]      left: out[row][cc] += X_sq[c][row][k] S[kb 64 + k][cc]
      const double a = below ? Own[r16 * PQ_PITCH + 64 * kb + k] : Ts[(16 * wave + r16) * LDA + k];
      const double b = below ? Ts[(16 * wave + r16) * LDA + k] : Own[(64 * kb + k) * 17 + r16];
      acc[ks & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[ks & 3], 0, 0, 0);
    }
    if (kb == c) {
      // 16x16 tile of output block c: rows 16 wm + rq + 4 qq, column 16 wn + c16
      const int wm = below ? q : wave, wn = below ? wave : q;
      const int c16 = lane & 15, rq = lane >> 4;
      gbyte* base = below ? (gbyte*)d.W + (((long long)i * NB) * np + (long long)(k0 + c) * NB) * 8
                          : (gbyte*)d.X + (((long long)(k0 + c) * NB) * np + (long long)j * NB) * 8;
      const unsigned voff = (unsigned)(((long long)(16 * wm + rq) * np + 16 * wn + c16) * 8);
      // (below) the fp32 copy of C for the off-chain inverse
      gbyte* base32 = (below && d.C32 != nullptr) ? (gbyte*)d.C32 + (((long long)i * NB) * np + (long long)(k0 + c) * NB) * 4 : nullptr;
      const unsigned voff32 = (unsigned)(((long long)(16 * wm + rq) * np + 16 * wn + c16) * 4);
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const double v = (acc[0][qq] + acc[1][qq]) + (acc[2][qq] + acc[3][qq]);
        *(gdouble*)(base + (long long)(4 * qq) * np * 8 + voff) = below ? v : -v;
        if (below && base32 != nullptr) *(gfloat*)(base32 + (long long)(4 * qq) * np * 4 + voff32) = (float)v;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = f64x4{0.0, 0.0, 0.0, 0.0};
    }
    next(c, kb);
    PQT(242 + pq_n)
    ++pq_n;
  };
  while (c >= 0) {
    step(rt0);
    if (c < 0) break;
    step(rt1);
    if (c < 0) break;
    step(rt2);
  }
}
__global__ void __launch_bounds__(INV_THREADS)
panel_product_quarter_kernel(const InvDev* __restrict__ t, int nf, int k0, int kend) {
  __shared__ double Own[PQ_OWN], Ts[NB * LDA];
  const int job = (blockIdx.x >> 5) * 8 + (blockIdx.x & 7), q = (blockIdx.x >> 3) & 3;   // the four quarters of a job on one XCD
  int f, local;
  KT_BEGIN(236)
  if (!locate(t, nf, job, [k0, kend](const InvDev& d) { return panel_jobs(d.P, s_kind(d), k0, kend); }, f, local)) return;
  const InvDev& d = t[f];
  const int nb = (kend < d.P ? kend : d.P) - k0;
  const int n_below = d.P > kend ? d.P - kend : 0;
  if (local < n_below) panel_quarter_body<true>(d, k0, nb, kend + local, 0, q, Own, Ts);
  else panel_quarter_body<false>(d, k0, nb, 0, local - n_below, q, Own, Ts);
  KT_END(237)
}
__global__ void __launch_bounds__(INV_THREADS, 3)
panel_product_kernel(const InvDev* __restrict__ t, int nf, int k0, int kend) {
  __shared__ double As[NB * OPA], Bs[NB * OPA > OKS * LDA ? NB * OPA : OKS * LDA];
  panel_product_body<2>(t, nf, k0, kend, As, Bs);
}
__global__ void __launch_bounds__(1024)
panel_product_wide_kernel(const InvDev* __restrict__ t, int nf, int k0, int kend) {
  __shared__ double As[NB * OPA], Bs[NB * OPA > OKS * LDA ? NB * OPA : OKS * LDA];
  panel_product_body<4>(t, nf, k0, kend, As, Bs);
}

// ------------------------------------------------------------------------------------------------
// (4') The triangular inverse X = C^-1 OUTSIDE the block squares, for KFAC.invert: in fp32 and off the chain.
//     S32[i][j] (+)= C32[i][panel] X[panel][j]      for block rows i below the panel, block columns j <= panel
// (the forward substitution of C X = I, right-looking, one K = 256 product per tile and panel), followed one panel later
// by the columns-left product with the square's fp64 inverse (xrows32_kernel), which turns the S rows of the next
// panel into rows of X.  Everything the chain computes - the Cholesky factor, the inverses of the 256 x 256 block squares -
// stays fp64; only these sums, n^3 / 3 of the sweep's (2/3) n^3 flops, run on v_mfma_f32_32x32x2_f32 at twice the fp64
// rate and half the bytes.  Forward substitution is far better conditioned than the factorisation: on damped ResNet
// factors (condition 1e4 .. 2e5) the fp32 sums cost 1e-7 .. 6e-7 relative Frobenius error of L against 2e-8 for the
// all-fp64 sweep (both below what the rounding of L itself to fp32 contributes to any product with it), while a fp32
// trailing update of the factorisation would cost cond x 6e-8 (DESIGN K2).
// The launches run on the far-update stream behind the far update of their panel; the chain (panel product, near
// update) carries no part of the inverse any more.
// One 64 x 64 tile per workgroup, 4 waves x one 32x32x2 accumulator; K advances in steps of 32 through register-staged
// LDS tiles (16-byte loads: a row of C32 is K-contiguous, a row of X32 column-contiguous).  The B operand of the panel's
// OWN block columns is the fp64 square X_sq (lower triangular: K starts at the column's block).
// ------------------------------------------------------------------------------------------------
constexpr int SKS = 32;                // K step
constexpr int SPA = SKS + 1;           // [64 rows][SKS] pitch (floats)
constexpr int SPB = NB + 32;           // [SKS][64 cols] pitch: the two lane halves of a 32x32x2 operand read 32 banks apart
// One 64 x 64 fp32 tile: out (=, +=) sign * A[64 x K] B[K x 64], K = [ke0, ke1) in steps of 32.  A rows are K-contiguous
// (fp32, or fp64 converted on the way into LDS), B is [k][col] (fp32 or fp64); everything wave-uniform.
struct Tile32 {
  const gbyte* a;        // element (row, ke) at a + (row * np + ke) * (a64 ? 8 : 4)
  const gbyte* b;        // element (ke, col) at b + (ke * np + col) * (b64 ? 8 : 4)
  gfloat* out;           // tile, pitch np
  int np, ke0, ke1;
  bool a64, b64, accumulate, negate;
};
__device__ __forceinline__ void tile32(const Tile32& o, float* __restrict__ As, float* __restrict__ Bs) {
  const int np = o.np;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int l32 = lane & 31, h = lane >> 5;
  // A: thread -> row = tid / 8 + 32 u, k = 4 (tid % 8);  B: thread -> k = tid / 16 + 16 u, col = 4 (tid % 16)
  const long long a_off = (long long)(tid >> 3) * np + 4 * (tid & 7), a_step = 32ll * np;
  const long long b_off = (long long)(tid >> 4) * np + 4 * (tid & 15), b_step = 16ll * np;
  typedef const __attribute__((address_space(1))) f32x4 gf4;
  typedef const __attribute__((address_space(1))) f64x4 gd4;
  f32x4 ra[2], rb[2];
  auto fetch = [&](int ke) __attribute__((always_inline)) {
    if (o.a64) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const f64x4 v = *(gd4*)(o.a + (a_off + u * a_step + ke) * 8);
        ra[u] = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
      }
    } else {
#pragma unroll
      for (int u = 0; u < 2; ++u) ra[u] = *(gf4*)(o.a + (a_off + u * a_step + ke) * 4);
    }
    if (o.b64) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const f64x4 v = *(gd4*)(o.b + (b_off + u * b_step + (long long)ke * np) * 8);
        rb[u] = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
      }
    } else {
#pragma unroll
      for (int u = 0; u < 2; ++u) rb[u] = *(gf4*)(o.b + (b_off + u * b_step + (long long)ke * np) * 4);
    }
  };
  f32x16 acc = {0};
  fetch(o.ke0);
  for (int ke = o.ke0; ke < o.ke1; ke += SKS) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        As[((tid >> 3) + 32 * u) * SPA + 4 * (tid & 7) + e] = ra[u][e];
        Bs[((tid >> 4) + 16 * u) * SPB + 4 * (tid & 15) + e] = rb[u][e];
      }
    __syncthreads();
    if (ke + SKS < o.ke1) fetch(ke + SKS);
#pragma unroll
    for (int kk = 0; kk < SKS; kk += 2) {
      const float av = As[(32 * wm + l32) * SPA + kk + h];
      const float bv = Bs[(kk + h) * SPB + 32 * wn + l32];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  // C/D map of the 32x32 block: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  gbyte* base = (gbyte*)(o.out + (long long)(32 * wm + 4 * h) * np + 32 * wn + l32);
  if (!o.accumulate) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg)
      *(gfloat*)(base + (long long)((reg & 3) + 8 * (reg >> 2)) * np * 4) = o.negate ? -acc[reg] : acc[reg];
  } else {
    float old[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) old[reg] = *(const gfloat*)(base + (long long)((reg & 3) + 8 * (reg >> 2)) * np * 4);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) *(gfloat*)(base + (long long)((reg & 3) + 8 * (reg >> 2)) * np * 4) = old[reg] + acc[reg];
  }
}
// S update of panel [k0, kend): tiles (i, j), i >= kend, j < kend, in SB x SB super-blocks (see outer_tiles)
__device__ __host__ __forceinline__ long long supd_tiles(int P, int kind, int kend) {
  if (kind != 0 || P <= kend) return 0;
  const long long nsb = (P - kend + SB - 1) / SB, ncb = (kend + SB - 1) / SB;
  return nsb * ncb * (SB * SB);
}
__global__ void __launch_bounds__(INV_THREADS, 3)
supd32_kernel(const InvDev* __restrict__ t, int nf, int k0, int kend, int n_items) {
  __shared__ float As[NB * SPA], Bs[SKS * SPB];
  int item;
  {
    const int bid = blockIdx.x, xcd = bid & 7, jj = bid >> 3;       // whole super-blocks per XCD (see outer_update_body)
    item = ((jj / (SB * SB)) * 8 + xcd) * (SB * SB) + (jj % (SB * SB));
  }
  if (item >= n_items) return;
  int f, local;
  if (!locate(t, nf, item, [kend](const InvDev& d) { return (int)supd_tiles(d.P, s_kind(d), kend); }, f, local)) return;
  const InvDev& d = t[f];
  const int np = d.np, r = d.P - kend;
  const int ncb = (kend + SB - 1) / SB;
  const int sb = local / (SB * SB), in = local - sb * (SB * SB);
  const int a = sb / ncb, cb = sb - a * ncb;
  const int ri = a * SB + in / SB, j = cb * SB + in % SB;
  if (ri >= r || j >= kend) return;
  const int i = kend + ri;
  const bool own = j >= k0;                                       // B = the fp64 square; first contribution to S[i][j]
  Tile32 o;
  o.np = np;
  o.a = (const gbyte*)((const gfloat*)d.C32 + (long long)i * NB * np);
  o.a64 = false;
  o.b = own ? (const gbyte*)((const gdouble*)d.X + j * NB) : (const gbyte*)((const gfloat*)d.X32 + j * NB);
  o.b64 = own;
  o.ke0 = (own ? j : k0) * NB;
  o.ke1 = kend * NB;
  o.out = (gfloat*)d.S32 + (long long)i * NB * np + j * NB;
  o.accumulate = !own;
  o.negate = false;
  tile32(o, As, Bs);
}
// the rows of X of panel [k0, kend) left of its square: X32[k0 + c][j] = - sum_{k <= c} X_sq[c][k] S32[k0 + k][j], j < k0
// (the output has a buffer of its own, so the block rows c of a panel are independent tiles)
__device__ __host__ __forceinline__ long long xrow_tiles(int P, int kind, int k0, int kend) {
  if (kind != 0 || P <= k0) return 0;
  return (long long)((kend < P ? kend : P) - k0) * k0;
}
__global__ void __launch_bounds__(INV_THREADS, 3)
xrows32_kernel(const InvDev* __restrict__ t, int nf, int k0, int kend) {
  __shared__ float As[NB * SPA], Bs[SKS * SPB];
  int f, local;
  if (!locate(t, nf, blockIdx.x, [k0, kend](const InvDev& d) { return (int)xrow_tiles(d.P, s_kind(d), k0, kend); }, f, local)) return;
  const InvDev& d = t[f];
  const int np = d.np, nb = (kend < d.P ? kend : d.P) - k0;
  const int j = local / nb, c = nb - 1 - (local - j * nb);          // longest K first
  Tile32 o;
  o.np = np;
  o.a = (const gbyte*)((const gdouble*)d.X + (long long)(k0 + c) * NB * np);
  o.a64 = true;
  o.b = (const gbyte*)((const gfloat*)d.S32 + j * NB);
  o.b64 = false;
  o.ke0 = k0 * NB;
  o.ke1 = (k0 + c + 1) * NB;
  o.out = (gfloat*)d.X32 + (long long)(k0 + c) * NB * np + j * NB;
  o.accumulate = false;
  o.negate = true;
  tile32(o, As, Bs);
}

// ------------------------------------------------------------------------------------------------
// (5) L[i][j] = (float) X[n-1-j][n-1-i] for j <= i, 0 above the diagonal (64x64 tiles via LDS)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(INV_THREADS)
inv_finalize_kernel(const InvDev* __restrict__ t, int nf, int sq) {   // sq: edge of the block squares (elements)
  __shared__ float tile[NB][NB + 1];
  int f, local;
  if (!locate(t, nf, blockIdx.x, [](const InvDev& d) { return d.P * d.P; }, f, local)) return;
  const InvDev& d = t[f];
  const int n = d.n, np = d.np, q = d.P;
  const int ti = local / q, tj = local - ti * q;       // output tile (rows ti*64.., cols tj*64..)
  const gdouble* X = (const gdouble*)d.X;
  const gfloat* X32 = (const gfloat*)d.X32;
  gfloat* L = (gfloat*)d.L;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), c = threadIdx.x & 63;   // a wave owns rows w, w + 4, ...
  if (!d.reverse) {                                          // plain copy of the lower triangle, fp64
    gdouble* Xo = (gdouble*)d.Xout;
    const gdouble* S = d.Zm != nullptr ? (const gdouble*)d.Zm : X;     // the solution of the right-hand side mode, or C^-1
#pragma unroll 4
    for (int u = 0; u < 16; ++u) {
      const int i = ti * NB + w + 4 * u, j = tj * NB + c;
      if (i < n && j < n) {
        double v = (j <= i) ? S[(long long)i * np + j] : 0.0;
        if (d.r_minus && j <= i) v = ((const gdouble*)d.R)[(long long)i * n + j] - v;
        Xo[(long long)i * n + j] = v;
      }
    }
    return;
  }
  if (tj > ti) {
#pragma unroll 4
    for (int u = 0; u < 16; ++u) {
      const int i = ti * NB + w + 4 * u, j = tj * NB + c;
      if (i < n && j < n) L[(long long)i * n + j] = 0.0f;
    }
    return;
  }
  // source element for output (i, j) is X[n-1-j][n-1-i]: rows of X are read coalesced along its columns
  // (all 16 loads of a lane in flight), transposed through LDS, and L is written row-wise
  double xv[16];
  const int i_l = ti * NB + c;                               // output row   -> source col n-1-i
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int j = tj * NB + w + 4 * u;                       // output column -> source row n-1-j
    // (the 256 x 256 block squares of C^-1 are fp64 in X, everything between them fp32 in X32)
    const int rs = n - 1 - j, cs = n - 1 - i_l;
    const bool live = i_l < n && j < n && j <= i_l;
    if (X32 == nullptr || rs / sq == cs / sq) xv[u] = live ? X[(long long)rs * np + cs] : 0.0;
    else xv[u] = live ? (double)X32[(long long)rs * np + cs] : 0.0;
  }
#pragma unroll
  for (int u = 0; u < 16; ++u) tile[w + 4 * u][c] = (float)xv[u];
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int i = ti * NB + w + 4 * u, j = tj * NB + c;
    if (i < n && j < n) L[(long long)i * n + j] = tile[c][w + 4 * u];
  }
}

constexpr int INV_UPLOAD_CHUNK = 28;
struct InvChunk { InvDev f[INV_UPLOAD_CHUNK]; };
static_assert(sizeof(InvChunk) <= 3840, "kernel argument block must stay below 4 KB");

__global__ void __launch_bounds__(256) inv_upload_kernel(InvDev* __restrict__ table, InvChunk chunk, int count) {
  const int words = count * (int)(sizeof(InvDev) / 4);
  const int* in = reinterpret_cast<const int*>(&chunk);
  int* out = reinterpret_cast<int*>(table);
  for (int w = threadIdx.x; w < words; w += blockDim.x) out[w] = in[w];
}

static size_t inv_table_bytes(int n) { return align_up((size_t)std::max(n, 1) * sizeof(InvDev), 256); }
static size_t inv_flags_bytes(int n) { return align_up((size_t)std::max(n, 1) * SQ_FLAGS * sizeof(int), 256); }

// Extra streams of a sweep, one set per device, created on first use:
//   side[g]  far part of the outer updates of factor group g (runs beside the next panel's chain of
//            small diagonal-step launches), with its fork/join events;
//   aux      the sweep of the second factor group (see chol_sweep).
struct SideStream {
  hipStream_t stream = nullptr;
  hipEvent_t ev_main[2] = {nullptr, nullptr};
  hipEvent_t ev_side[2] = {nullptr, nullptr};
  hipEvent_t ev_tail = nullptr;          // the last row panel of the triangular inverse is done
};
struct StreamSet {
  SideStream side[2];
  hipStream_t aux = nullptr;             // the large group's chain: high priority, all CUs
  hipStream_t masked = nullptr;          // the small group's sweep (an internal stream of its own: measured 3 %
                                         // faster than running it on the caller's stream; confining it to a
                                         // CU subset with hipExtStreamCreateWithCUMask did not help)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join2 = nullptr;
  hipEvent_t ev_chain[2] = {nullptr, nullptr};      // a group's last chain step is enqueued (its finalize pass is not)
};
// CU-masked streams are destroyed explicitly when the process exits: left to the runtime's own teardown they
// crashed inside __cxa_finalize when a profiler (rocprofv3) was attached.  The handler is registered after the
// HIP runtime initialised, so it runs before the runtime's exit handlers.
static std::mutex g_masked_mutex;
static std::vector<std::pair<int, hipStream_t>> g_masked_streams;
static void destroy_masked_streams() {
  std::lock_guard<std::mutex> lock(g_masked_mutex);
  for (auto& e : g_masked_streams) {
    if (hipSetDevice(e.first) == hipSuccess) {
      (void)hipStreamSynchronize(e.second);
      (void)hipStreamDestroy(e.second);
    }
  }
  g_masked_streams.clear();
}

static int stream_set(StreamSet** out) {
  static thread_local std::vector<std::pair<int, StreamSet>> cache;
  int dev = 0;
  CURV_HIP_CHECK(hipGetDevice(&dev));
  for (auto& e : cache) if (e.first == dev) { *out = &e.second; return CURV_OK; }
  StreamSet s;
  int plo = 0, phi = 0;
  CURV_HIP_CHECK(hipDeviceGetStreamPriorityRange(&plo, &phi));
  // the large group's chain of short launches is the critical path of the sweep: highest priority
  static const int aux_prio = getenv("CURV_AUX_PRIO") ? atoi(getenv("CURV_AUX_PRIO")) : 1;
  CURV_HIP_CHECK(hipEventCreateWithFlags(&s.ev_fork, hipEventDisableTiming));
  CURV_HIP_CHECK(hipEventCreateWithFlags(&s.ev_join, hipEventDisableTiming));
  // the far updates are throughput work: lowest priority, so that the latency-critical chain launches of
  // the other streams get workgroup slots first
  int prio_low = 0, prio_high = 0;
  CURV_HIP_CHECK(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
  CURV_HIP_CHECK(hipEventCreateWithFlags(&s.ev_join2, hipEventDisableTiming));
  for (int g = 0; g < 2; ++g) CURV_HIP_CHECK(hipEventCreateWithFlags(&s.ev_chain[g], hipEventDisableTiming));
  // The far updates fill every workgroup slot they can get (4 per CU), and a retiring far workgroup frees
  // half the LDS a chain kernel's workgroup needs: the slot is refilled before a second one retires, and
  // stream priorities do not reserve anything - traced: a 9-workgroup chol_panel launch waited 200-260 us
  // for the far update beside it to drain.  So the streams that carry wide, throughput-bound launches may not use the
  // last CURV_FREE_CUS CUs (mask bits are dealt round-robin over the XCDs: 2 CUs of every XCD), which the chains' small
  // launches then find free.
  auto wide_stream = [&](hipStream_t* out) -> int {
    static const int free_cus = getenv("CURV_FREE_CUS") ? atoi(getenv("CURV_FREE_CUS")) : 32;
    hipDeviceProp_t prop;
    int dev_id = 0;
    CURV_HIP_CHECK(hipGetDevice(&dev_id));
    CURV_HIP_CHECK(hipGetDeviceProperties(&prop, dev_id));
    const int n_cu = prop.multiProcessorCount;
    if (free_cus > 0 && 2 * free_cus <= n_cu) {
      std::vector<uint32_t> mask((size_t)cdiv(n_cu, 32), 0u);
      for (int c = 0; c < n_cu - free_cus; ++c) mask[c >> 5] |= 1u << (c & 31);
      if (hipExtStreamCreateWithCUMask(out, (uint32_t)mask.size(), mask.data()) == hipSuccess) {
        std::lock_guard<std::mutex> lock(g_masked_mutex);
        if (g_masked_streams.empty()) atexit(destroy_masked_streams);
        g_masked_streams.emplace_back(dev_id, *out);
        return CURV_OK;
      }
      (void)hipGetLastError();            // a runtime without CU masks: plain low-priority stream below
    }
    CURV_HIP_CHECK(hipStreamCreateWithPriority(out, hipStreamNonBlocking, prio_low));
    return CURV_OK;
  };
  // Test hook (CURV_STREAM_ORDER; tests/test_invert_gpu.py, tools/stream_sensitivity.py, LAB_NOTEBOOK R5.6): creation order of
  // the set's streams with dummies between them: a = large chain, m = small chain, 0 / 1 = far-update streams, x / p / h / l =
  // unused CU-masked / plain / high-priority / low-priority stream (the dummies live as long as the process)
  if (const char* order = getenv("CURV_STREAM_ORDER")) {
    for (const char* c = order; *c; ++c) {
      hipStream_t dummy = nullptr;
      switch (*c) {
        case 'a': CURV_HIP_CHECK(hipStreamCreateWithPriority(&s.aux, hipStreamNonBlocking, phi)); break;
        case 'm': CURV_HIP_CHECK(hipStreamCreateWithFlags(&s.masked, hipStreamNonBlocking)); break;
        case '0': { const int rc = wide_stream(&s.side[0].stream); if (rc != CURV_OK) return rc; } break;
        case '1': { const int rc = wide_stream(&s.side[1].stream); if (rc != CURV_OK) return rc; } break;
        case 'x': { const int rc = wide_stream(&dummy); if (rc != CURV_OK) return rc; } break;
        case 'p': CURV_HIP_CHECK(hipStreamCreateWithFlags(&dummy, hipStreamNonBlocking)); break;
        case 'h': CURV_HIP_CHECK(hipStreamCreateWithPriority(&dummy, hipStreamNonBlocking, phi)); break;
        case 'l': CURV_HIP_CHECK(hipStreamCreateWithPriority(&dummy, hipStreamNonBlocking, plo)); break;
        default: break;
      }
    }
  }
  if (s.aux == nullptr) CURV_HIP_CHECK(hipStreamCreateWithPriority(&s.aux, hipStreamNonBlocking, aux_prio ? phi : plo));
  // The small group's chain: a plain stream.  Measured alternatives (round 5, ResNet-50 factors): CU-masked like the far
  // updates' streams (the reserved CUs then belong to the large group's chain alone) 7.5 -> 8.5 ms -
  // the small group's own chain starves beside the far updates; on the LOW priority level (the runtime keeps a pool of
  // hardware queues per level, so the stream would not share a queue with normal streams the process created earlier):
  // no effect on the creation-order sensitivity, 6.9 / 6.9 / 11.0 ms for none / three streams after / three before the
  // set, as with a normal stream.  What the four busy streams of a sweep need is four different hardware pipes; a
  // fifth busy stream of any kind (CU-masked, low priority, shared between the groups) costs 4-5 ms.
  if (s.masked == nullptr) CURV_HIP_CHECK(hipStreamCreateWithFlags(&s.masked, hipStreamNonBlocking));
  for (int g = 0; g < 2; ++g) {
    if (s.side[g].stream == nullptr) { const int rc = wide_stream(&s.side[g].stream); if (rc != CURV_OK) return rc; }
    for (int i = 0; i < 2; ++i) {
      CURV_HIP_CHECK(hipEventCreateWithFlags(&s.side[g].ev_main[i], hipEventDisableTiming));
      CURV_HIP_CHECK(hipEventCreateWithFlags(&s.side[g].ev_side[i], hipEventDisableTiming));
    }
    CURV_HIP_CHECK(hipEventCreateWithFlags(&s.side[g].ev_tail, hipEventDisableTiming));
  }
  cache.emplace_back(dev, s);
  *out = &cache.back().second;
  return CURV_OK;
}

// The factor build (syrk.hip) runs its register-staged kernel on a side stream.  It borrows this set's `masked`
// stream instead of creating one more: HIP maps streams onto a handful of hardware queues, and a sixth stream in the
// process changed that mapping for the sweep's own streams (invert() of the ResNet-50 factors 8.3 -> 9.2 ms with an
// extra stream created by the factor build).  The two uses never overlap: both are ordered on the caller's stream.
int curv_internal_side_stream(hipStream_t* out) {
  StreamSet* ss = nullptr;
  const int rc = stream_set(&ss);
  if (rc != CURV_OK) return rc;
  *out = ss->masked;
  return CURV_OK;
}

}  // namespace curv

using namespace curv;

#ifdef CURV_SQ_TRACE
extern "C" int curv_debug_sq_trace_reset() {
  long long zeros[256] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(curv::g_sq_trace), zeros, sizeof(zeros)) == hipSuccess ? 0 : 1;
}
extern "C" int curv_debug_sq_trace(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(curv::g_sq_trace), 256 * sizeof(long long)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int curv_init_streams(void) {
  StreamSet* ss = nullptr;
  return stream_set(&ss);
}

extern "C" size_t curv_chol_inv_workspace_bytes(const curv_inv_desc* descs, int n_factors) {
  size_t total = 2 * inv_table_bytes(n_factors) + 2 * inv_flags_bytes(n_factors);
  for (int i = 0; i < n_factors; ++i) {
    if (descs[i].n <= 0) return 0;
    const size_t np = (size_t)cdiv(descs[i].n, NB) * NB;
    total += 2 * np * np * sizeof(double) + 3 * np * np * sizeof(float);     // W, X; C32, X32, S32
  }
  return total;
}

// One batched sweep over the factors of `tab` (work matrices already assigned), everything enqueued on `stream` except
// the far outer updates and the fp32 inverse, which go to side->stream.  The sweep is ENQUEUED panel by panel
// (begin / panel_step / end), so that the host can interleave the panels of two groups: a hipStreamWaitEvent issued
// after a whole sweep has been enqueued waits for that stream's tail, not for the event's position in it (measured: the
// second group of a whole-model inversion started when the first one's last panels ran, whatever event it waited for;
// tools/trace_buckets.py), so a group that is to start beside another one must be enqueued beside it.
// block columns per outer panel (GroupSweep::begin has the measurements)
static int sweep_nbo(bool latency_bound) {
  static const int nbo_env = getenv("CURV_NBO") ? atoi(getenv("CURV_NBO")) : 0;
  return latency_bound ? 4 : (nbo_env > 0 ? nbo_env : 6);
}
struct GroupSweep {
  hipStream_t stream;
  SideStream* side;
  const std::vector<InvDev>* tabp;
  InvDev* table;
  int* flags;
  bool latency_bound;
  int n_factors = 0, Pmax = 0, NBO = 4, k0 = 0, panel = 0;
  long long fin_tiles = 0, quarter_prod = 0;
  bool use_square = false, far_pending = false, inv_pending = false, capturing = false;

  GroupSweep(hipStream_t st, SideStream* sd, const std::vector<InvDev>& tab, InvDev* tb, int* fl, bool lb)
      : stream(st), side(sd), tabp(&tab), table(tb), flags(fl), latency_bound(lb) {}
  bool done() const { return k0 >= Pmax; }
  int panels() const { return cdiv(Pmax, NBO); }

  int begin() {
    const std::vector<InvDev>& tab = *tabp;
    n_factors = (int)tab.size();
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    CURV_HIP_CHECK(hipStreamIsCapturing(stream, &cap));
    capturing = cap != hipStreamCaptureStatusNone;      // (events ride on launches only outside a capture)
    long long prep_tiles = 0;
    for (const InvDev& d : tab) {
      Pmax = std::max(Pmax, d.P);
      prep_tiles += (long long)d.P * (d.P + 1) / 2;
      fin_tiles += (long long)d.P * d.P;
    }
    for (int b = 0; b < n_factors; b += INV_UPLOAD_CHUNK) {
      InvChunk chunk;
      const int count = std::min(INV_UPLOAD_CHUNK, n_factors - b);
      memset(&chunk, 0, sizeof(chunk));
      memcpy(chunk.f, tab.data() + b, (size_t)count * sizeof(InvDev));
      hipLaunchKernelGGL(inv_upload_kernel, dim3(1), dim3(256), 0, stream, table + b, chunk, count);
      CURV_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(inv_prepare_kernel, dim3((unsigned)prep_tiles), dim3(INV_THREADS), 0, stream, table, n_factors, flags);
    CURV_LAUNCH_CHECK();
    // outer panel: 4 block columns = 256 for the chain-bound forms (the square kernel is built for 4); 6 = 384 for the
    // per-step launches of a whole model - same-box sweeps on the ResNet-50 factors, two rounds each: NBO 4 / 5 / 6 / 7 =
    // 7.5 / 7.2 / 6.8 / 7.3 ms (8: 7.4, 12: 7.3): fewer panel products, near updates and cross-stream hand-offs per block
    // step, and 72 = 12 x 6, 36 = 6 x 6: the widest factors end on a full panel
    NBO = sweep_nbo(latency_bound);
    return CURV_OK;
  }

  // one outer panel: its chain, the near update, and on the side stream the far update and the fp32 inverse
  int panel_step() {
    const std::vector<InvDev>& tab = *tabp;
    // Per panel: the chain of diagonal steps (small, latency-bound launches) runs on the caller's stream,
    // then the near part of the outer update (the block columns / rows the NEXT chain touches); the rest of
    // the outer update goes to a second stream and overlaps the following chains (see below).
    // launches narrower than the GPU take the 1024-thread form of the tile kernels (see tile_product_k32)
    static const long long wide_near = getenv("CURV_WIDE_NEAR") ? atoll(getenv("CURV_WIDE_NEAR")) : 512;
    static const long long wide_prod = getenv("CURV_WIDE_PROD") ? atoll(getenv("CURV_WIDE_PROD")) : 256;
    // ... and below this many jobs the quarter form of the panel product (panel_product_quarter_kernel)
    static const long long quarter_prod_env = getenv("CURV_QUARTER_PROD") ? atoll(getenv("CURV_QUARTER_PROD")) : 512;
    quarter_prod = latency_bound ? quarter_prod_env : 0;
    // A call with few factors (a layer-sharded rank, a single large factor) is bound by the latency of its chain: the block
    // square of a panel then goes into one launch whose workgroups hand tiles to each other (chol_square_kernel), and the
    // panel product takes the quarter form.  A whole model on one GPU gains nothing from them (round 3: 8.40 ms with the
    // per-step launches, 8.55 with the square kernel for the large group; round 5: 6.7-6.8 vs 6.7-7.0), its shards over
    // 2 / 4 / 8 ranks do (tools/emulate_sharding.py, round 3: step 9.6 / 6.3 / 4.4 -> 9.2 / 5.6 / 3.8 ms).
    use_square = latency_bound;
    const int kend = k0 + NBO, row0 = kend + NBO;
    long long prod_tiles = 0;
    if (use_square) {
      long long sq_wgs = 0;
      for (const InvDev& d : tab)
        if (d.P > k0) sq_wgs += std::min(kend, d.P) - k0;
      hipLaunchKernelGGL(chol_square_kernel, dim3((unsigned)sq_wgs), dim3(INV_THREADS), 0, stream, table, n_factors, k0, kend,
                         flags, panel + 1);
      CURV_LAUNCH_CHECK();
    }
    for (int k = k0; !use_square && k < std::min(kend, Pmax); ++k) {
      long long diag_tiles = 0, panel_tiles = 0, upd_tiles = 0;
      for (const InvDev& d : tab) {
        if (d.P > k) { ++diag_tiles; panel_tiles += std::min(kend, d.P) - k0 - 1; }
        upd_tiles += inner_tiles(d.P, k, k0, kend);
      }
      if (k == k0) {    // later diagonal blocks of the panel are factorised by the inner update of step k - 1
        hipLaunchKernelGGL(chol_diag_kernel, dim3((unsigned)diag_tiles), dim3(INV_THREADS), 0, stream, table, n_factors, k);
        CURV_LAUNCH_CHECK();
      }
      if (panel_tiles > 0) {
        hipLaunchKernelGGL(chol_panel_kernel, dim3((unsigned)panel_tiles), dim3(INV_THREADS), 0, stream, table, n_factors, k, k0, kend);
        CURV_LAUNCH_CHECK();
      }
      if (upd_tiles > 0) {
        hipLaunchKernelGGL(inner_update_kernel, dim3((unsigned)upd_tiles), dim3(INV_THREADS), 0, stream, table, n_factors, k, k0, kend);
        CURV_LAUNCH_CHECK();
      }
    }
    // the launch form of a panel product by its number of jobs
    // `done` (optional): recorded as the launch's own completion (hipExtLaunchKernelGGL's stop event) - no marker packet
    // of its own in the chain's queue
    auto launch_product = [&](hipStream_t st, long long jobs, hipEvent_t done) -> int {
      const InvDev* tb = table;
      if (jobs <= quarter_prod)
        hipExtLaunchKernelGGL(panel_product_quarter_kernel, dim3((unsigned)(cdivll(jobs, 8) * 32)), dim3(INV_THREADS), 0, st,
                              nullptr, done, 0, tb, n_factors, k0, kend);
      else if (jobs <= wide_prod)
        hipExtLaunchKernelGGL(panel_product_wide_kernel, dim3((unsigned)jobs), dim3(1024), 0, st, nullptr, done, 0, tb, n_factors,
                              k0, kend);
      else
        hipExtLaunchKernelGGL(panel_product_kernel, dim3((unsigned)jobs), dim3(INV_THREADS), 0, st, nullptr, done, 0, tb,
                              n_factors, k0, kend);
      CURV_LAUNCH_CHECK();
      return CURV_OK;
    };
    long long inv_jobs = 0, inv_tiles = 0;    // fp32 inverse: columns-left products of this panel, then its S update
    for (const InvDev& d : tab) {
      prod_tiles += panel_jobs(d.P, s_kind(d), k0, kend);
      inv_jobs += xrow_tiles(d.P, s_kind(d), k0, kend);
      inv_tiles += supd_tiles(d.P, s_kind(d), kend);
    }
    static const int ext_events = getenv("CURV_EXT_EVENTS") ? atoi(getenv("CURV_EXT_EVENTS")) : 1;
    bool fork_recorded = false;           // ev_main rides on the panel product's completion
    if (prod_tiles > 0) {   // rows below the square (right-hand side mode: and the columns of Zm): one triangular product each
      fork_recorded = ext_events != 0 && !capturing;
      const int rc = launch_product(stream, prod_tiles, fork_recorded ? side->ev_main[panel & 1] : nullptr);
      if (rc != CURV_OK) return rc;
    }
    // Outer update of this panel in two parts: near = the strip the next chain touches (this stream, on the
    // critical path), far = everything beyond, on the side stream beside the next panel's chain.  The two
    // meet again at the next near part, which rewrites tiles the far part has written.  (Splitting off a
    // "mid" strip so that a far update has two chain periods before anything waits for it was measured
    // 2-4 % slower: the far updates are throughput-bound, not waited for.)
    long long near_tiles = 0, far_tiles = 0;
    bool any_s_part = false;
    for (const InvDev& d : tab) {
      near_tiles += strip_tiles(d.P, kend, kend, row0, s_in_sweep(d));
      far_tiles += outer_tiles(d.P, kend, row0, s_in_sweep(d));
      any_s_part = any_s_part || s_in_sweep(d);
    }
    hipStream_t inv_st = side->stream;
    // (Measured and removed, LAB_NOTEBOOK R5.3 / R5.7: the near update on the side stream in front of the far update -
    // ResNet-50 factors 7.75 -> 8.66 ms, the chain pays two cross-stream waits per panel; the fp32 inverse on a stream of
    // its own - every additional busy hardware queue costs far more than it brings, 7.5 -> 11.1 ms.)
    const bool inv_work = inv_jobs > 0 || inv_tiles > 0;
    const bool side_work = far_tiles > 0 || inv_work;
    // fork: the side stream's work needs this panel's chain.  CURV_FORK_AFTER_NEAR=1 forks BEHIND the near update (a far
    // update launched beside it takes every CU but the reserved ones and the near update then runs in rounds on those: one
    // 4608^2 40 -> 14 us per panel) - measured neutral: the side stream's work moves 25 us later and lands on the next panel
    // product instead (17 -> 45 us); one 4608^2 2.63 vs 2.66 ms, whole model 6.87 vs 6.84 ms over six pairs
    static const int fork_late_env = getenv("CURV_FORK_AFTER_NEAR") ? atoi(getenv("CURV_FORK_AFTER_NEAR")) : 0;
    const bool fork_late = fork_late_env != 0;
    auto fork = [&]() -> int {
      if (side_work) {
        if (!fork_recorded || fork_late) CURV_HIP_CHECK(hipEventRecord(side->ev_main[panel & 1], stream));
        CURV_HIP_CHECK(hipStreamWaitEvent(side->stream, side->ev_main[panel & 1], 0));
      }
      return CURV_OK;
    };
    if (!fork_late) { const int rc = fork(); if (rc != CURV_OK) return rc; }
    if (near_tiles > 0) {
      if (far_pending) {                         // join: the previous far part wrote the tiles updated here
        CURV_HIP_CHECK(hipStreamWaitEvent(stream, side->ev_side[(panel + 1) & 1], 0));
      }
      far_pending = false;
      if (near_tiles <= wide_near)
        hipLaunchKernelGGL(outer_update_wide_kernel, dim3((unsigned)near_tiles), dim3(1024), 0, stream, table, n_factors,
                           k0, kend, kend, row0, 1, (int)near_tiles);
      else
        hipLaunchKernelGGL(outer_update_kernel, dim3((unsigned)near_tiles), dim3(INV_THREADS), 0, stream, table, n_factors, k0,
                           kend, kend, row0, 1, (int)near_tiles);
      CURV_LAUNCH_CHECK();
    }
    if (fork_late) { const int rc = fork(); if (rc != CURV_OK) return rc; }
    if (far_tiles > 0) {
      const long long grid = cdivll(far_tiles, 8 * SB * SB) * 8 * SB * SB;
      const bool ride = ext_events != 0 && !capturing;
      const InvDev* tb = table;
      // trailing tiles only (KFAC.invert: the inverse is accumulated in fp32 off the chain): the LDS-DMA form
      static const int far_dma = getenv("CURV_FAR_DMA") ? atoi(getenv("CURV_FAR_DMA")) : 1;
      if (far_dma && !any_s_part)
        hipExtLaunchKernelGGL(outer_update_dma_kernel, dim3((unsigned)grid), dim3(INV_THREADS), 0, side->stream, nullptr,
                              ride ? side->ev_side[panel & 1] : nullptr, 0, tb, n_factors, k0, kend, row0, (int)far_tiles);
      else if (far_dma) {
        // a sweep that accumulates its inverse itself (curv_chol_factor_inverse, INF's fp64 chain): the trailing tiles on the
        // LDS-DMA form, the S tiles (B operand as [k][col]) on the register-staged one, one launch each
        long long trail = 0;
        for (const InvDev& d : tab) trail += outer_tiles(d.P, kend, row0, false);
        const long long s_tiles = far_tiles - trail;
        if (trail > 0) {
          const long long g1 = cdivll(trail, 8 * SB * SB) * 8 * SB * SB;
          hipExtLaunchKernelGGL(outer_update_dma_kernel, dim3((unsigned)g1), dim3(INV_THREADS), 0, side->stream, nullptr,
                                (ride && s_tiles == 0) ? side->ev_side[panel & 1] : nullptr, 0, tb, n_factors, k0, kend, row0, (int)trail);
          CURV_LAUNCH_CHECK();
        }
        if (s_tiles > 0) {
          const long long g2 = cdivll(s_tiles, 8 * SB * SB) * 8 * SB * SB;
          hipExtLaunchKernelGGL(outer_update_kernel, dim3((unsigned)g2), dim3(INV_THREADS), 0, side->stream, nullptr,
                                ride ? side->ev_side[panel & 1] : nullptr, 0, tb, n_factors, k0, kend, row0, 0, 2, (int)s_tiles);
        }
      } else
        hipExtLaunchKernelGGL(outer_update_kernel, dim3((unsigned)grid), dim3(INV_THREADS), 0, side->stream, nullptr,
                              ride ? side->ev_side[panel & 1] : nullptr, 0, tb, n_factors, k0, kend, row0, 0, 0, (int)far_tiles);
      CURV_LAUNCH_CHECK();
      if (!ride) CURV_HIP_CHECK(hipEventRecord(side->ev_side[panel & 1], side->stream));
      far_pending = true;
    }
    // the fp32 inverse on its stream: the rows of this panel x X_sq (needs the square, i.e. this panel's chain, and the S
    // updates of all earlier panels: stream order), then this panel's S update of the rows below
    if (inv_jobs > 0) {
      hipLaunchKernelGGL(xrows32_kernel, dim3((unsigned)inv_jobs), dim3(INV_THREADS), 0, inv_st, table, n_factors, k0, kend);
      CURV_LAUNCH_CHECK();
      inv_pending = true;
    }
    if (inv_tiles > 0) {
      const long long grid = cdivll(inv_tiles, 8 * SB * SB) * 8 * SB * SB;
      hipLaunchKernelGGL(supd32_kernel, dim3((unsigned)grid), dim3(INV_THREADS), 0, inv_st, table, n_factors, k0, kend,
                         (int)inv_tiles);
      CURV_LAUNCH_CHECK();
      inv_pending = true;
    }
    k0 += NBO;
    ++panel;
    return CURV_OK;
  }

  int end(hipEvent_t chain_done = nullptr) {
    // the status words are final here: every factorisation step is enqueued on `stream`, the finalize pass does not touch them
    if (chain_done != nullptr) CURV_HIP_CHECK(hipEventRecord(chain_done, stream));
    if (far_pending) CURV_HIP_CHECK(hipStreamWaitEvent(stream, side->ev_side[(panel + 1) & 1], 0));
    if (inv_pending) {
      CURV_HIP_CHECK(hipEventRecord(side->ev_tail, side->stream));
      CURV_HIP_CHECK(hipStreamWaitEvent(stream, side->ev_tail, 0));
    }
    hipLaunchKernelGGL(inv_finalize_kernel, dim3((unsigned)fin_tiles), dim3(INV_THREADS), 0, stream, table, n_factors, NBO * NB);
    CURV_LAUNCH_CHECK();
    return CURV_OK;
  }
};

static int chol_sweep_group(hipStream_t stream, SideStream* side, const std::vector<InvDev>& tab, InvDev* table, int* flags,
                            bool latency_bound, hipEvent_t chain_done = nullptr) {
  GroupSweep g(stream, side, tab, table, flags, latency_bound);
  int rc = g.begin();
  while (rc == CURV_OK && !g.done()) rc = g.panel_step();
  return rc == CURV_OK ? g.end(chain_done) : rc;
}

// Factors advance in lock step (step k touches every factor with more than k blocks), so a sweep over
// all of them has the launch count of the largest factor and, in its first panels, kernels as wide as
// the whole model.  The large factors (more than SPLIT_P blocks: their chain of ~85 us steps is the
// critical path) and the many small ones are therefore swept as two groups on two streams; the GPU
// interleaves the small group's wide, throughput-bound kernels with the large group's short ones.
constexpr int SPLIT_P = 16;

// `early` (optional): the status words of the call are copied to pinned host memory as soon as the last factorisation
// step of every factor has run - BEFORE the finalize passes - and `ev` is recorded behind that copy: a host that waits
// for `ev` instead of for the caller's stream gets the verdict while the finalize passes (0.15 ms for a ResNet-50) still
// run, and prepares its next launches in their shadow.  The copy travels on the small group's far-update stream, idle by
// then: one more stream in the process would change the mapping of the sweep's streams onto hardware queues (see
// curv_internal_side_stream).
struct EarlyStatus { int* host; const int* dev; size_t bytes; hipEvent_t ev; };

// The caller's stream joins the sweep through hipStreamWaitEvent, i.e. a barrier packet at the head of its hardware
// queue for as long as the sweep runs - and a queue that sits on an unsatisfied barrier slows down every other queue on
// its PIPE of the command processor (queues are dealt onto four pipes in creation order).  Measured on the ResNet-50
// factors with the set's streams created in all 24 orders behind 0 / 2 other queues (LAB_NOTEBOOK R5.6): 6.8-7.1 ms unless
// a chain's queue is the (4 k + 3)-th of the process - the caller's pipe -, then 9.0-9.5 ms (small group's chain there)
// or 10.3-11.0 ms (large group's).  This is the "creation order" sensitivity of rounds 3-5: three streams created in
// front of the set put the large chain on the caller's pipe.  With an early verdict the host waits for the status words
// anyway, so the join is enqueued AFTER that wait - the caller's queue then holds its barrier for the finalize passes
// only (6.9 / 6.8 / 10.9 -> 6.8 / 6.8 / 7.3 ms for none / three streams after / three before the set).  The plain entry
// point (no host wait: graph capture, check=False) joins at once and keeps the sensitivity.
static int await_verdict(const EarlyStatus* early) {
  static const int late_join = getenv("CURV_LATE_JOIN") ? atoi(getenv("CURV_LATE_JOIN")) : 1;
  if (early != nullptr && late_join) CURV_HIP_CHECK(hipEventSynchronize(early->ev));
  return CURV_OK;
}

static int chol_sweep(hipStream_t stream, std::vector<InvDev>& tab, void* workspace, size_t workspace_bytes,
                      const char* who, const EarlyStatus* early = nullptr) {
  const int n_factors = (int)tab.size();
  static const int latency_max = getenv("CURV_LATENCY_MAX") ? atoi(getenv("CURV_LATENCY_MAX")) : 64;
  bool any_rhs = false;
  for (const InvDev& d : tab) any_rhs = any_rhs || d.R != nullptr;
  // (the right-hand side mode lives in the per-step kernels: the chain-bound forms keep their operands elsewhere)
  const bool latency_bound = n_factors <= latency_max && !any_rhs;
  size_t need = 2 * inv_table_bytes(n_factors) + 2 * inv_flags_bytes(n_factors);
  for (const InvDev& d : tab)
    need += (d.R != nullptr ? 3 : 2) * (size_t)d.np * d.np * sizeof(double) + (d.reverse ? 3 * (size_t)d.np * d.np * sizeof(float) : 0);
  if (workspace == nullptr || workspace_bytes < need) {
    set_error("%s: workspace too small (%zu < %zu bytes)", who, workspace_bytes, need);
    return CURV_ERR_WORKSPACE;
  }
  char* p = reinterpret_cast<char*>(workspace) + 2 * inv_table_bytes(n_factors);
  int* flags0 = reinterpret_cast<int*>(p);
  int* flags1 = reinterpret_cast<int*>(p + inv_flags_bytes(n_factors));
  p += 2 * inv_flags_bytes(n_factors);
  std::vector<InvDev> big, small;
  int Pmax = 0;
  for (const InvDev& d : tab) Pmax = std::max(Pmax, d.P);
  // the large group holds the factors that share the longest chain (more than half of the largest block
  // count): measured on ResNet-50, {4608} vs the rest beats {2048, 2304, 4608} vs the rest by 4 %
  const int split = std::max(SPLIT_P, Pmax / 2);
  // a chain-bound call is swept as ONE group: the square kernel gives every factor its own workgroups, so the small
  // factors do not widen the large one's chain launches, and half the launches, events and waits remain (the host needs
  // 0.5 ms to enqueue the two sweeps of an 18-factor shard whose kernels take 0.6 ms)
  static const int one_group_env = getenv("CURV_ONE_GROUP") ? atoi(getenv("CURV_ONE_GROUP")) : 1;
  static const int force_one = getenv("CURV_FORCE_ONE_GROUP") ? atoi(getenv("CURV_FORCE_ONE_GROUP")) : 0;
  const bool one_group = (latency_bound && one_group_env != 0) || force_one != 0;
  for (InvDev& d : tab) {
    d.W = reinterpret_cast<double*>(p); p += (size_t)d.np * d.np * sizeof(double);
    d.X = reinterpret_cast<double*>(p); p += (size_t)d.np * d.np * sizeof(double);
    if (d.R != nullptr) { d.Zm = reinterpret_cast<double*>(p); p += (size_t)d.np * d.np * sizeof(double); }
    if (d.reverse) {                     // KFAC.invert: the inverse outside the block squares in fp32, off the chain
      d.C32 = reinterpret_cast<float*>(p); p += (size_t)d.np * d.np * sizeof(float);
      d.X32 = reinterpret_cast<float*>(p); p += (size_t)d.np * d.np * sizeof(float);
      d.S32 = reinterpret_cast<float*>(p); p += (size_t)d.np * d.np * sizeof(float);
    }
    (d.P > split || one_group ? big : small).push_back(d);
  }
  StreamSet* ss = nullptr;
  { const int rc = stream_set(&ss); if (rc != CURV_OK) return rc; }
  InvDev* table0 = reinterpret_cast<InvDev*>(workspace);
  InvDev* table1 = reinterpret_cast<InvDev*>(reinterpret_cast<char*>(workspace) + inv_table_bytes(n_factors));
  // Under stream capture (curvature_amd/graph.py) the chains run on the caller's stream and only the far updates fork
  // from it: hipStreamEndCapture of ROCm 7.2 crashed on every capture in which two forked streams depend on each other
  // (the chain's internal stream and its far-update stream: every call with a factor wider than 512), while forks that
  // only exchange events with the capturing stream itself are fine.  The two groups then run one after the other.
  hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
  CURV_HIP_CHECK(hipStreamIsCapturing(stream, &capture));
  if (capture != hipStreamCaptureStatusNone) {
    if (early != nullptr) {
      set_error("%s: the early status read-back is a host copy and cannot be captured", who);
      return CURV_ERR_INVALID;
    }
    if (!big.empty()) {
      const int rc = chol_sweep_group(stream, &ss->side[0], big, table0, flags0, latency_bound);
      if (rc != CURV_OK) return rc;
    }
    if (!small.empty()) return chol_sweep_group(stream, &ss->side[1], small, big.empty() ? table0 : table1,
                                                big.empty() ? flags0 : flags1, latency_bound);
    return CURV_OK;
  }
  if (big.empty() || small.empty()) {
    // one group: still swept on an internal stream.  The CU-masked side streams are created by an API that
    // has no "non-blocking" flag, so they synchronise implicitly with the legacy default stream - which the
    // caller's stream may be.
    CURV_HIP_CHECK(hipEventRecord(ss->ev_fork, stream));
    CURV_HIP_CHECK(hipStreamWaitEvent(ss->aux, ss->ev_fork, 0));
    const int rc1 = chol_sweep_group(ss->aux, &ss->side[0], big.empty() ? small : big, table0, flags0, latency_bound,
                                     early ? ss->ev_chain[0] : nullptr);
    if (rc1 != CURV_OK) return rc1;
    if (early != nullptr) {
      CURV_HIP_CHECK(hipStreamWaitEvent(ss->side[0].stream, ss->ev_chain[0], 0));
      CURV_HIP_CHECK(hipMemcpyAsync(early->host, early->dev, early->bytes, hipMemcpyDeviceToHost, ss->side[0].stream));
      CURV_HIP_CHECK(hipEventRecord(early->ev, ss->side[0].stream));
    }
    { const int rc = await_verdict(early); if (rc != CURV_OK) return rc; }
    CURV_HIP_CHECK(hipEventRecord(ss->ev_join, ss->aux));
    CURV_HIP_CHECK(hipStreamWaitEvent(stream, ss->ev_join, 0));
    return CURV_OK;
  }
  CURV_HIP_CHECK(hipEventRecord(ss->ev_fork, stream));
  CURV_HIP_CHECK(hipStreamWaitEvent(ss->aux, ss->ev_fork, 0));
  CURV_HIP_CHECK(hipStreamWaitEvent(ss->masked, ss->ev_fork, 0));
  // The large group is bound by its far updates in its first panels and by its chain in its last ones (the far
  // work shrinks with (P^2 - k^2), the chain does not): the small group starts when the large one has done
  // `start_panel` panels, so that its throughput work fills the large group's chain-bound tail.
  static const int start_frac = getenv("CURV_SMALL_START") ? atoi(getenv("CURV_SMALL_START")) : 30;  // percent of the panels
  // (ResNet-50: 0 -> 9.25 ms, 20 -> 9.2, 30 -> 9.0, 40 -> 9.2, 50 -> 9.5)
  const int n_panels = cdiv(Pmax, sweep_nbo(latency_bound));
  long long far0 = 0;                          // far tiles of the large group's first panel
  for (const InvDev& d : big) far0 += outer_tiles(d.P, 4, 8, s_in_sweep(d));
  // only a far-bound large group has such a tail to fill (a chain step is ~54 us, a far tile ~0.04 us of the
  // whole GPU): with [2048 | 1024, 512, 256] the delay costs 8 %
  const int start_panel = far0 >= 5000 ? std::min(n_panels - 1, n_panels * start_frac / 100) : 0;
  // (the chain-bound forms for the large group of a whole model: 6.7-7.0 vs 6.7-6.8 ms, LAB_NOTEBOOK R5.3 - not kept)
  // The two sweeps are enqueued panel by panel, the small group's panel t - start_panel behind the large group's panel t:
  // the event the small group's stream waits for (the large group has done `start_panel` panels) is then the large
  // group's stream TAIL at the moment of the wait, which is what hipStreamWaitEvent effectively waits for.
  GroupSweep gb(ss->aux, &ss->side[0], big, table0, flags0, latency_bound);
  // CURV_SMALL_ONE_STREAM=1: the small group's far updates and fp32 inverse on its chain's stream (three busy streams)
  static const int small_one = getenv("CURV_SMALL_ONE_STREAM") ? atoi(getenv("CURV_SMALL_ONE_STREAM")) : 0;
  SideStream side_small = ss->side[1];
  if (small_one) side_small.stream = ss->masked;
  GroupSweep gs(ss->masked, small_one ? &side_small : &ss->side[1], small, table1, flags1, latency_bound);
  int rc = gb.begin();
  if (rc != CURV_OK) return rc;
  bool small_started = false;
  for (int t = 0; !gb.done() || !small_started || !gs.done(); ++t) {
    if (!small_started && (t >= start_panel || gb.done())) {
      if (t > 0) {
        CURV_HIP_CHECK(hipEventRecord(ss->ev_join2, ss->aux));
        CURV_HIP_CHECK(hipStreamWaitEvent(ss->masked, ss->ev_join2, 0));
      }
      rc = gs.begin();
      if (rc != CURV_OK) return rc;
      small_started = true;
    }
    if (!gb.done()) { rc = gb.panel_step(); if (rc != CURV_OK) return rc; }
    if (small_started && !gs.done()) { rc = gs.panel_step(); if (rc != CURV_OK) return rc; }
  }
  rc = gb.end(early ? ss->ev_chain[0] : nullptr);
  if (rc != CURV_OK) return rc;
  rc = gs.end(early ? ss->ev_chain[1] : nullptr);
  if (rc != CURV_OK) return rc;
  if (early != nullptr) {
    CURV_HIP_CHECK(hipStreamWaitEvent(ss->side[1].stream, ss->ev_chain[0], 0));
    CURV_HIP_CHECK(hipStreamWaitEvent(ss->side[1].stream, ss->ev_chain[1], 0));
    CURV_HIP_CHECK(hipMemcpyAsync(early->host, early->dev, early->bytes, hipMemcpyDeviceToHost, ss->side[1].stream));
    CURV_HIP_CHECK(hipEventRecord(early->ev, ss->side[1].stream));
  }
  { const int rc = await_verdict(early); if (rc != CURV_OK) return rc; }
  CURV_HIP_CHECK(hipEventRecord(ss->ev_join, ss->aux));
  CURV_HIP_CHECK(hipStreamWaitEvent(stream, ss->ev_join, 0));
  CURV_HIP_CHECK(hipEventRecord(ss->ev_join2, ss->masked));
  CURV_HIP_CHECK(hipStreamWaitEvent(stream, ss->ev_join2, 0));
  return CURV_OK;
}

static int chol_inv_lower_impl(void* stream_, const curv_inv_desc* descs, int n_factors, int* info,
                               void* workspace, size_t workspace_bytes, const EarlyStatus* early);

extern "C" int curv_chol_inv_lower(void* stream_, const curv_inv_desc* descs, int n_factors, int* info,
                                   void* workspace, size_t workspace_bytes) {
  return chol_inv_lower_impl(stream_, descs, n_factors, info, workspace, workspace_bytes, nullptr);
}

extern "C" int curv_chol_inv_lower_status(void* stream_, const curv_inv_desc* descs, int n_factors, int* info,
                                          void* workspace, size_t workspace_bytes, int* host_status, void* ev_status) {
  CURV_REQUIRE(host_status != nullptr && ev_status != nullptr, "curv_chol_inv_lower_status: null argument");
  if (n_factors == 0) return CURV_OK;
  EarlyStatus early{host_status, info, (size_t)n_factors * sizeof(int), (hipEvent_t)ev_status};
  return chol_inv_lower_impl(stream_, descs, n_factors, info, workspace, workspace_bytes, &early);
}

static int chol_inv_lower_impl(void* stream_, const curv_inv_desc* descs, int n_factors, int* info,
                               void* workspace, size_t workspace_bytes, const EarlyStatus* early) {
  if (n_factors == 0) return CURV_OK;
  CURV_REQUIRE(descs != nullptr && info != nullptr, "curv_chol_inv_lower: null argument");
  std::vector<InvDev> tab(n_factors);
  for (int i = 0; i < n_factors; ++i) {
    const curv_inv_desc& s = descs[i];
    CURV_REQUIRE(s.n > 0, "curv_chol_inv_lower: factor %d: empty", i);
    CURV_REQUIRE(s.F != nullptr && s.L != nullptr, "curv_chol_inv_lower: factor %d: null pointer", i);
    CURV_REQUIRE(s.multiply >= 0.0 && s.add >= 0.0, "curv_chol_inv_lower: factor %d: negative hyper-parameter", i);
    InvDev& d = tab[i];
    memset(&d, 0, sizeof(d));
    d.F = s.F; d.L = s.L; d.n = s.n;
    d.P = cdiv(s.n, NB); d.np = d.P * NB;
    d.info = info + i;
    d.sqrt_s = (float)sqrt(s.multiply);      // s ** 0.5 in double, then the fp32 tensor multiply
    d.sqrt_n = (float)sqrt(s.add);
    d.reverse = 1;
  }
  return chol_sweep((hipStream_t)stream_, tab, workspace, workspace_bytes, "curv_chol_inv_lower", early);
}

extern "C" size_t curv_chol_factor_inverse_workspace_bytes(const curv_cholinv_desc* descs, int n) {
  size_t total = 2 * inv_table_bytes(n) + 2 * inv_flags_bytes(n);
  for (int i = 0; i < n; ++i) {
    if (descs[i].n <= 0) return 0;
    const size_t np = (size_t)cdiv(descs[i].n, NB) * NB;
    total += (descs[i].R != nullptr ? 3 : 2) * np * np * sizeof(double);
  }
  return total;
}

extern "C" int curv_chol_factor_inverse(void* stream_, const curv_cholinv_desc* descs, int n_mats, int* info,
                                        void* workspace, size_t workspace_bytes) {
  if (n_mats == 0) return CURV_OK;
  CURV_REQUIRE(descs != nullptr && info != nullptr, "curv_chol_factor_inverse: null argument");
  std::vector<InvDev> tab(n_mats);
  for (int i = 0; i < n_mats; ++i) {
    const curv_cholinv_desc& s = descs[i];
    CURV_REQUIRE(s.n > 0 && s.M != nullptr && s.X != nullptr, "curv_chol_factor_inverse: matrix %d invalid", i);
    InvDev& d = tab[i];
    memset(&d, 0, sizeof(d));
    d.F = reinterpret_cast<const float*>(s.M); d.Xout = s.X; d.n = s.n;
    d.f64_in = s.m_is_f64 ? 1 : 0;
    d.P = cdiv(s.n, NB); d.np = d.P * NB;
    d.info = info + i;
    d.sqrt_s = 1.0f;
    d.sqrt_n = (float)s.diag_add;            // added in fp32 like the reference's `vtv + eye` (:567)
    d.reverse = 0;
    d.pivot_min = s.pivot_min > 0.0 ? s.pivot_min : 0.0;
    d.R = s.R;                               // (Zm: a third work matrix, placed by chol_sweep)
    d.r_minus = (s.R != nullptr && s.r_minus) ? 1 : 0;
  }
  return chol_sweep((hipStream_t)stream_, tab, workspace, workspace_bytes, "curv_chol_factor_inverse");
}
