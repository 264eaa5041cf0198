// Batched, strided fp32 GEMM on the f32 MFMA with the fused epilogues the estimators need, plus the
// Philox normal generator of the samplers.
//
//   C = epilogue( alpha * op(A) op(B) )      op(.) expressed through element strides
//
// Used for (reference lines in curvature/curvatures.py):
//   KFAC.sample   (L_A z L_G^T)^T = L_G z^T L_A^T                      :387-392
//   EFB.update    Lambda += (U_G^T grad U_A)**2                        :427-431
//   EFB.sample    (U_A (z * inv^T) U_G^T)^T                            :457-460
//   INF           the small dense products of update / pre_sampler / sampler   :487-600
// One launch covers any number of independent products (one per layer): the work list is
// (descriptor, 64x64 output tile), decoded on the device from the descriptor table.
#include "common.h"
#include "gemm_nt.h"

#include <algorithm>
#include <vector>

namespace curv {

constexpr int GT = 64;                 // output tile edge
constexpr int GK = 16;                 // K depth per stage
constexpr int GP = GT + 1;             // LDS pitch of a [k][row] operand image


__device__ __forceinline__ int gemm_find(const GemmDev* __restrict__ t, int n, int id) {
  const int lane = threadIdx.x & 63;
  int count = 0;
  for (int f0 = 0; f0 < n; f0 += 64) {
    const int f = f0 + lane;
    bool le = false;
    if (f < n) le = t[f].tile_base <= id;
    count += __popcll(__ballot(le));
  }
  return __builtin_amdgcn_readfirstlane(count - 1);
}

// One output tile of TMv x TMv: 4 waves as 2 x 2, each wave (TMv/2) x (TMv/2) = BL x BL blocks of 32x32.
template <int TMv>
__device__ __forceinline__ void gemm_tile(const GemmDev& d, int local, float* lds) {
  constexpr int BL = TMv / 64;                 // 32x32 blocks per wave edge
  constexpr int PITCH = TMv + 1;
  constexpr int PER_T = TMv * GK / GEMM_THREADS;   // staged elements per thread per operand
  float* As0 = lds;                            // [2][GK * PITCH]
  float* Bs0 = lds + 2 * GK * PITCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, h = lane >> 5;
  int tm = local / d.tiles_n, tn = local - tm * d.tiles_n;
  // triangular operands cut K per tile: hand out the long tiles first, so the tail of the launch is made of
  // the short ones
  if (d.tri == CURV_TRI_A_LOWER) tm = (d.M + TMv - 1) / TMv - 1 - tm;
  else if (d.tri == CURV_TRI_B_UPPER) tn = d.tiles_n - 1 - tn;
  const int i0 = tm * TMv, j0 = tn * TMv;
  const int M = d.M, N = d.N;
  int K = d.K;
  if (d.tri == CURV_TRI_A_LOWER) K = min(K, i0 + TMv);        // A[i][k] = 0 for k > i
  else if (d.tri == CURV_TRI_B_UPPER) K = min(K, j0 + TMv);   // B[k][j] = 0 for k > j
  const gfl* A = (const gfl*)d.A;
  const gfl* B = (const gfl*)d.B;
  const long long a_rs = d.a_rs, a_cs = d.a_cs, b_rs = d.b_rs, b_cs = d.b_cs;
  // staging map: lanes run along the unit-stride dimension of the operand so that global reads coalesce
  const bool a_kfast = (a_cs == 1), b_kfast = (b_rs == 1);
  int ar[PER_T], ak[PER_T], bc[PER_T], bk[PER_T];
  // Per-thread element offsets inside the tile's operand panels are computed once (32-bit: the host checks
  // that a tile spans less than 2^31 elements); a stage adds only the wave-uniform k0 * stride to the panel
  // base, so every load is SGPR base + 32-bit lane offset.  (Per-element 64-bit i * rs + k * cs index
  // arithmetic was ~1300 VALU cycles per stage and wave, next to 2048 MFMA cycles.)
  int oa[PER_T], ob[PER_T];
  unsigned va = 0, vb = 0;                        // row / column inside the matrix?
#pragma unroll
  for (int u = 0; u < PER_T; ++u) {
    const int e = tid + u * GEMM_THREADS;
    if (a_kfast) { ak[u] = e & 15; ar[u] = e >> 4; } else { ar[u] = e % TMv; ak[u] = e / TMv; }
    if (b_kfast) { bk[u] = e & 15; bc[u] = e >> 4; } else { bc[u] = e % TMv; bk[u] = e / TMv; }
    oa[u] = (int)(ar[u] * a_rs + ak[u] * a_cs);
    ob[u] = (int)(bk[u] * b_rs + bc[u] * b_cs);
    if (i0 + ar[u] < M) va |= 1u << u;
    if (j0 + bc[u] < N) vb |= 1u << u;
  }
  const bool interior = (i0 + TMv <= M) && (j0 + TMv <= N);
  const gfl* At = A + (long long)i0 * a_rs;
  const gfl* Bt = B + (long long)j0 * b_cs;
  float ra[PER_T], rb[PER_T];
  auto fetch = [&](int k0) __attribute__((always_inline)) {
    const gfl* Ak = At + (long long)k0 * a_cs;
    const gfl* Bk = Bt + (long long)k0 * b_rs;
    if (interior && k0 + GK <= K) {               // tile and stage inside the matrices (wave-uniform): plain loads.
      // (A predicated load costs an exec-mask save / branch / restore and a v_mov around it: ~100 extra
      // instructions per stage of 32 MFMAs, and vector instructions do not hide behind MFMAs, LAB_NOTEBOOK.md section 8.)
#pragma unroll
      for (int u = 0; u < PER_T; ++u) {
        ra[u] = Ak[oa[u]];
        rb[u] = Bk[ob[u]];
      }
    } else if (k0 + GK <= K) {                    // whole stage inside K
#pragma unroll
      for (int u = 0; u < PER_T; ++u) {
        ra[u] = ((va >> u) & 1) ? Ak[oa[u]] : 0.0f;
        rb[u] = ((vb >> u) & 1) ? Bk[ob[u]] : 0.0f;
      }
    } else {
#pragma unroll
      for (int u = 0; u < PER_T; ++u) {
        ra[u] = (((va >> u) & 1) && k0 + ak[u] < K) ? Ak[oa[u]] : 0.0f;
        rb[u] = (((vb >> u) & 1) && k0 + bk[u] < K) ? Bk[ob[u]] : 0.0f;
      }
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int u = 0; u < PER_T; ++u) {
      As0[buf * GK * PITCH + ak[u] * PITCH + ar[u]] = ra[u];
      Bs0[buf * GK * PITCH + bk[u] * PITCH + bc[u]] = rb[u];
    }
  };

  f32x16 acc[BL][BL];
#pragma unroll
  for (int m = 0; m < BL; ++m)
#pragma unroll
    for (int n = 0; n < BL; ++n) acc[m][n] = f32x16{0};
  const int nk = (K + GK - 1) / GK;
  fetch(0);
  stash(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) fetch((kt + 1) * GK);          // in flight during the MFMAs below
    const float* as = As0 + buf * GK * PITCH + (TMv / 2) * wm + r32;
    const float* bs = Bs0 + buf * GK * PITCH + (TMv / 2) * wn + r32;
#pragma unroll
    for (int kp = 0; kp < GK / 2; ++kp) {
      float a[BL], b[BL];
#pragma unroll
      for (int m = 0; m < BL; ++m) a[m] = as[(2 * kp + h) * PITCH + 32 * m];
#pragma unroll
      for (int n = 0; n < BL; ++n) b[n] = bs[(2 * kp + h) * PITCH + 32 * n];
#pragma unroll
      for (int m = 0; m < BL; ++m)
#pragma unroll
        for (int n = 0; n < BL; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[n], acc[m][n], 0, 0, 0);
    }
    if (kt + 1 < nk) stash(buf ^ 1);
    __syncthreads();
  }

  // epilogue; C/D map of the 32x32 block: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  gfl* C = (gfl*)d.C;
  const gfl* E = (const gfl*)d.E;
  const gfl* F = (const gfl*)d.F;
  const float alpha = d.alpha, beta = d.beta;
  const int ep = d.epilogue;
#pragma unroll
  for (int m = 0; m < BL; ++m)
#pragma unroll
    for (int n = 0; n < BL; ++n) {
      const int j = j0 + (TMv / 2) * wn + 32 * n + r32;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int i = i0 + (TMv / 2) * wm + 32 * m + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        if (i < M && j < N) {
          const long long ci = i * d.c_rs + j * d.c_cs;
          float v = alpha * acc[m][n][reg];
          if (ep == CURV_EPI_SQUARE) v = alpha * acc[m][n][reg] * acc[m][n][reg];
          else if (ep == CURV_EPI_MUL_E) v *= E[i * d.e_rs + j * d.e_cs];
          else if (ep == CURV_EPI_ADD_E) v += E[i * d.e_rs + j * d.e_cs];
          else if (ep == CURV_EPI_MUL_E_ADD_F) v = v * E[i * d.e_rs + j * d.e_cs] + F[i * d.f_rs + j * d.f_cs];
          if (beta != 0.0f) v += beta * C[ci];
          C[ci] = v;
        }
      }
    }
}

__global__ void __launch_bounds__(GEMM_THREADS)
gemm_f32_kernel(const GemmDev* __restrict__ table, int n_desc) {
  __shared__ float lds[4 * GK * (128 + 1)];     // A and B images, double buffered, sized for the 128 tile
  const int f = gemm_find(table, n_desc, blockIdx.x);
  const GemmDev& d = table[f];
  const int local = blockIdx.x - d.tile_base;
  if (d.tm == 128) gemm_tile<128>(d, local, lds);
  else gemm_tile<64>(d, local, lds);
}

__global__ void __launch_bounds__(GEMM_THREADS, nt::WGS)
gemm_nt_kernel(const GemmDev* __restrict__ table, int n_desc, float* __restrict__ slabs, const int* __restrict__ order) {
  __shared__ __attribute__((aligned(1024))) char smem[nt::LDS_B];
  // `order`: the launch's tiles by descending K (host-sorted): workgroups are dispatched in index order, so the launch
  // ends with its shortest tiles whatever product they belong to
  const int id = order != nullptr ? order[blockIdx.x] : (int)blockIdx.x;
  const int f = gemm_find(table, n_desc, id);
  const GemmDev& d = table[f];
  gemm_nt_tile(d, id - d.tile_base, (lds_char_t*)smem, slabs);
}

// Split products: one workgroup per output tile sums the tile's partial slabs in slice order (deterministic) and
// applies the epilogue.  Slices beyond the tile's (triangle-cut) K range were never written and are not read.
__global__ void __launch_bounds__(256)
gemm_nt_reduce_kernel(const GemmDev* __restrict__ table, int n_desc, const float* __restrict__ slabs) {
  constexpr int TM = nt::TM;
  const int lane = threadIdx.x & 63;
  int f = -1;
  for (int f0 = 0; f0 < n_desc; f0 += 64) {            // the split product with the largest red_base <= blockIdx
    const int g = f0 + lane;
    const bool hit = g < n_desc && table[g].red_base >= 0 && table[g].red_base <= (int)blockIdx.x;
    const unsigned long long m = __ballot(hit);
    if (m) f = f0 + 63 - __builtin_clzll(m);
  }
  f = __builtin_amdgcn_readfirstlane(f);
  const GemmDev& d = table[f];
  const int tile = blockIdx.x - d.red_base;
  int tm = tile / d.tiles_n, tn = tile - tm * d.tiles_n;
  if (d.tri == CURV_TRI_A_LOWER) tm = (d.M + TM - 1) / TM - 1 - tm;
  else if (d.tri == CURV_TRI_B_UPPER) tn = d.tiles_n - 1 - tn;
  const int i0 = tm * TM, j0 = tn * TM;
  int K = d.K;
  if (d.tri == CURV_TRI_A_LOWER) K = min(K, i0 + TM);
  else if (d.tri == CURV_TRI_B_UPPER) K = min(K, j0 + TM);
  const int valid = (K + d.kslice - 1) / d.kslice;
  const float* base = slabs + d.slab_base + (long long)tile * d.n_slices * (TM * TM);
  for (int e = threadIdx.x; e < TM * TM / 4; e += 256) {
    const int r = e / (TM / 4), c = (e - r * (TM / 4)) * 4;
    const int i = i0 + r;
    if (i >= d.M) continue;
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int sl = 0; sl < valid; ++sl) v += *reinterpret_cast<const f32x4*>(base + (long long)sl * (TM * TM) + r * TM + c);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (j0 + c + q < d.N) nt_epilogue(d, i, j0 + c + q, v[q]);
  }
}

// ------------------------------------------------------------------------------------------------
// NT products of a SMALL launch (LeNet-scale models: every matrix at most a few hundred wide).  A 128-wide tile of such
// a product runs on ONE CU at 1/256 of the chip's MFMA rate and pays a memory round trip per 32-k stage: 27 + 50 us
// for the two sampler launches of LeNet-5 (profiles/r04_lenet_trace.txt).  Here a workgroup owns one 32 x 32 output
// block and its four waves split the K range in four contiguous quarters; each lane reads its operand rows straight
// from global memory, 16 bytes (4 k values = the inputs of 4 MFMA steps) per load, up to 16 loads per operand in
// flight at once - no LDS staging, no per-stage barrier: one memory round trip per 128 k of a wave.  The four partial
// blocks meet in LDS and are summed in wave order (bit-reproducible); wave w finishes registers 4w .. 4w+3.
// Rows beyond M / N are clamped to the last row (never stored); k beyond K is zeroed; a triangular operand cuts K.
// Operand rows need only 4-byte alignment (a 401-wide factor, a column slice of a wider buffer): the 16-byte loads are
// dword-aligned global loads, which gfx950 serves (as do the buffer loads of gemm_nt_kernel on the same operands).
// ------------------------------------------------------------------------------------------------
namespace sm {
constexpr int T = 32;            // output block edge
constexpr int CHUNKS = 16;       // 8-k chunks per wave and pass (16 B per lane and operand each)
constexpr int MAX_K = 4 * 8 * CHUNKS;   // K range one pass covers
}  // namespace sm

__global__ void __launch_bounds__(GEMM_THREADS)
gemm_nt_small_kernel(const GemmDev* __restrict__ table, int n_desc) {
  using namespace sm;
  __shared__ float part[4][16][64];
  const int f = gemm_find(table, n_desc, blockIdx.x);
  const GemmDev& d = table[f];
  const int local = blockIdx.x - d.tile_base;
  const int tm = local / d.tiles_n, tn = local - tm * d.tiles_n;
  const int i0 = tm * T, j0 = tn * T, M = d.M, N = d.N;
  int K = d.K;
  if (d.tri == CURV_TRI_A_LOWER) K = min(K, i0 + T);
  else if (d.tri == CURV_TRI_B_UPPER) K = min(K, j0 + T);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31, h = lane >> 5;
  const gfl* arow = (const gfl*)d.A + (long long)min(i0 + r32, M - 1) * d.a_rs;
  const gfl* brow = (const gfl*)d.B + (long long)min(j0 + r32, N - 1) * d.b_cs;
  f32x16 acc = {0};
  for (int p0 = 0; p0 < K; p0 += MAX_K) {
    const int Kp = min(K - p0, MAX_K);                       // this pass
    const int Kq = (((Kp + 3) >> 2) + 7) & ~7;               // a wave's quarter, whole chunks
    const int kb = p0 + wave * Kq, ke = min(p0 + Kp, kb + Kq);
    const int nch = __builtin_amdgcn_readfirstlane(ke > kb ? (ke - kb + 7) >> 3 : 0);
    f32x4 a[CHUNKS], b[CHUNKS];
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
      if (c < nch) {
        const int k = kb + 8 * c + 4 * h;
        if (kb + 8 * c + 8 <= ke) {                           // whole chunk (wave-uniform): 16-byte loads
          a[c] = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(arow + k);
          b[c] = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(brow + k);
        } else {                                              // the chunk that holds the end of K
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool ok = k + e < ke;
            const int kk = ok ? k + e : kb;
            const float va = arow[kk], vb = brow[kk];
            a[c][e] = ok ? va : 0.0f;
            b[c][e] = ok ? vb : 0.0f;
          }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
      if (c < nch) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][e], b[c][e], acc, 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) part[wave][reg][lane] = acc[reg];
  __syncthreads();
  // C/D map of the 32x32 block: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  const int j = j0 + r32;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int reg = 4 * wave + q;
    const float v = ((part[0][reg][lane] + part[1][reg][lane]) + part[2][reg][lane]) + part[3][reg][lane];
    const int i = i0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
    if (i < M && j < N) nt_epilogue(d, i, j, v);
  }
}

// ------------------------------------------------------------------------------------------------
// The same batched strided GEMM in fp64 (v_mfma_f64_16x16x4_f64) for the ill-conditioned products of
// INF.pre_sampler (L_c = A^-T (I - B^-1) A^-1).  alpha/beta only, no fused epilogue.
// ------------------------------------------------------------------------------------------------
struct Gemm64Dev {
  const double* A;
  const double* B;
  double* C;
  long long a_rs, a_cs, b_rs, b_cs, c_rs, c_cs;
  int M, N, K;
  int tiles_n, tile_base, tri;
  double alpha, beta;
  const double* E;             // optional: C = alpha acc + beta E
  const float* rs;             // optional fp32 output C32[i][j] = (float)(alpha rs[i] cs[j] acc)
  const float* cs;
  float* C32;
};

typedef __attribute__((address_space(1))) double gdbl;

// C[i][j] = alpha acc + beta (E ? E : C)[i][j], or the scaled fp32 form
__device__ __forceinline__ void gemm64_store(const Gemm64Dev& d, int i, int j, double acc) {
  typedef __attribute__((address_space(1))) float gflt;
  const long long ci = i * d.c_rs + j * d.c_cs;
  if (d.C32 != nullptr) {
    double v = d.alpha * acc;
    if (d.rs != nullptr) v *= (double)((const gflt*)d.rs)[i];
    if (d.cs != nullptr) v *= (double)((const gflt*)d.cs)[j];
    ((gflt*)d.C32)[ci] = (float)v;
    return;
  }
  double v = d.alpha * acc;
  if (d.beta != 0.0) v += d.beta * (d.E != nullptr ? ((const gdbl*)d.E)[ci] : ((const gdbl*)d.C)[ci]);
  ((gdbl*)d.C)[ci] = v;
}

// descriptors travel as kernel arguments, G64_BATCH per launch: a ResNet-scale INF.invert hands over 54 products of
// 500-1100 tiles each, and four per launch (the first form) ended every launch with a partly filled round of its
// longest tiles - 14 tails per call
constexpr int G64_BATCH = 24;
struct Gemm64Table { Gemm64Dev d[G64_BATCH]; };
static_assert(sizeof(Gemm64Table) <= 3840, "kernel argument block must stay below 4 KB");

// workgroup -> (descriptor, tile): tile_base is ascending; lane l looks at descriptor l
__device__ __forceinline__ int gemm64_find(const Gemm64Table& tab, int n_desc, int bid) {
  const int lane = threadIdx.x & 63;
  const bool le = lane < n_desc && tab.d[lane < G64_BATCH ? lane : 0].tile_base <= bid;
  return __builtin_amdgcn_readfirstlane(__popcll(__ballot(le)) - 1);
}
// tile order inside a product: longest K range first (triangular operands cut it per tile), so that a launch ends
// with short tiles
__device__ __forceinline__ void gemm64_tile(const Gemm64Dev& d, int local, int T, int& tm, int& tn) {
  tm = local / d.tiles_n;
  tn = local - tm * d.tiles_n;
  if (d.tri & 1) tm = (d.M + T - 1) / T - 1 - tm;       // A lower: K ends at i0 + T
  if (d.tri & 8) tn = d.tiles_n - 1 - tn;               // B upper: K ends at j0 + T
}

__global__ void __launch_bounds__(GEMM_THREADS)
gemm_f64_kernel(const Gemm64Table tab, int n_desc) {
  __shared__ double As[GK * GP];     // [k][row]
  __shared__ double Bs[GK * GP];     // [k][col]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r16 = lane & 15, kq = lane >> 4;
  const Gemm64Dev& d = tab.d[gemm64_find(tab, n_desc, blockIdx.x)];
  const int local = blockIdx.x - d.tile_base;
  int tm, tn;
  gemm64_tile(d, local, GT, tm, tn);
  const int i0 = tm * GT, j0 = tn * GT;
  if ((d.tri & 16) && j0 > i0 + GT - 1) return;   // C lower: the tile lies strictly above the diagonal
  const int M = d.M, N = d.N, K = d.K;
  const gdbl* A = (const gdbl*)d.A;
  const gdbl* B = (const gdbl*)d.B;
  const bool a_kfast = (d.a_cs == 1), b_kfast = (d.b_rs == 1);
  f64x4 acc[2][2] = {};
  // staging map (lanes along the unit-stride dimension) and element offsets, computed once; the next K stage is
  // fetched into registers while the MFMAs of the current one run (these products sit on INF.invert's serial
  // path with a handful of tiles each: without the prefetch every stage paid a full load round trip)
  int ar[4], ak[4], bc[4], bk[4];
  long long oa[4], ob[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = tid + u * GEMM_THREADS;
    if (a_kfast) { ak[u] = e & 15; ar[u] = e >> 4; } else { ar[u] = e & 63; ak[u] = e >> 6; }
    if (b_kfast) { bk[u] = e & 15; bc[u] = e >> 4; } else { bc[u] = e & 63; bk[u] = e >> 6; }
    oa[u] = (long long)(i0 + ar[u]) * d.a_rs + (long long)ak[u] * d.a_cs;
    ob[u] = (long long)bk[u] * d.b_rs + (long long)(j0 + bc[u]) * d.b_cs;
  }
  double ra[4], rb[4];
  auto fetch = [&](int k0) __attribute__((always_inline)) {
    const gdbl* Ak = A + (long long)k0 * d.a_cs;
    const gdbl* Bk = B + (long long)k0 * d.b_rs;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      ra[u] = (i0 + ar[u] < M && k0 + ak[u] < K) ? Ak[oa[u]] : 0.0;
      rb[u] = (j0 + bc[u] < N && k0 + bk[u] < K) ? Bk[ob[u]] : 0.0;
    }
  };
  // triangular operands (CURV_TRI64_*): the K range that can contribute to this tile
  int k_lo = 0, k_hi = K;
  if (d.tri & 1) k_hi = min(k_hi, i0 + GT);      // A lower: a[i][k] = 0 for k > i
  if (d.tri & 2) k_lo = max(k_lo, i0);           // A upper: a[i][k] = 0 for k < i
  if (d.tri & 4) k_lo = max(k_lo, j0);           // B lower: b[k][j] = 0 for k < j
  if (d.tri & 8) k_hi = min(k_hi, j0 + GT);      // B upper: b[k][j] = 0 for k > j
  k_lo &= ~(GK - 1);
  if (k_lo >= k_hi && d.beta == 1.0 && d.E == nullptr && d.C32 == nullptr) return;     // nothing to add
  if (k_lo < k_hi) fetch(k_lo);
  for (int k0 = k_lo; k0 < k_hi; k0 += GK) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      As[ak[u] * GP + ar[u]] = ra[u];
      Bs[bk[u] * GP + bc[u]] = rb[u];
    }
    __syncthreads();
    if (k0 + GK < k_hi) fetch(k0 + GK);
#pragma unroll
    for (int ks = 0; ks < GK / 4; ++ks) {
      const int k = 4 * ks + kq;
      double a[2], b[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) a[m] = As[k * GP + 32 * wm + 16 * m + r16];
#pragma unroll
      for (int n = 0; n < 2; ++n) b[n] = Bs[k * GP + 32 * wn + 16 * n + r16];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + 32 * wm + 16 * m + kq + 4 * r, j = j0 + 32 * wn + 16 * n + r16;
        if (i < M && j < N) gemm64_store(d, i, j, acc[m][n][r]);
      }
}

// The macro-tile form of gemm_f64_kernel for large products (INF.pre_sampler on ResNet-scale layers: ab = 3400-4284):
// 128 x 128 output tile, four waves of 64 x 64 = 4 x 4 MFMA tiles each, K steps of 16 with the next step's 16 loads per
// thread in flight during the 64 MFMAs of the current one, two workgroups per CU.  16 flops per operand byte instead
// of 8: the 64 x 64 tile loop is bound by where its operands come from (47 TFLOP/s from HBM, 66 from L2), this one
// is not (61-65 either way, profiles/r04_micro_mfma_f64.txt).  Same descriptors, strides, triangular K cuts and
// alpha / beta epilogue as gemm_f64_kernel.
constexpr int G64M_T = 128, G64M_P = G64M_T + 1;
__global__ void __launch_bounds__(GEMM_THREADS, 2)
gemm_f64_macro_kernel(const Gemm64Table tab, int n_desc) {
  __shared__ double As[GK * G64M_P];     // [k][row]
  __shared__ double Bs[GK * G64M_P];     // [k][col]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r16 = lane & 15, kq = lane >> 4;
  const Gemm64Dev& d = tab.d[gemm64_find(tab, n_desc, blockIdx.x)];
  const int local = blockIdx.x - d.tile_base;
  int tm, tn;
  gemm64_tile(d, local, G64M_T, tm, tn);
  const int i0 = tm * G64M_T, j0 = tn * G64M_T;
  if ((d.tri & 16) && j0 > i0 + G64M_T - 1) return;   // C lower: the tile lies strictly above the diagonal
  const int M = d.M, N = d.N, K = d.K;
  const gdbl* A = (const gdbl*)d.A;
  const gdbl* B = (const gdbl*)d.B;
  // staging maps: element u of a thread is (row ar0 + u a_dr, k ak0 + u a_dk); lanes run along the operand's unit stride
  const bool a_kfast = (d.a_cs == 1), b_kfast = (d.b_rs == 1);
  const int ar0 = a_kfast ? tid >> 4 : tid & 127, ak0 = a_kfast ? tid & 15 : tid >> 7;
  const int a_dr = a_kfast ? 16 : 0, a_dk = a_kfast ? 0 : 2;
  const int bc0 = b_kfast ? tid >> 4 : tid & 127, bk0 = b_kfast ? tid & 15 : tid >> 7;
  const int b_dc = b_kfast ? 16 : 0, b_dk = b_kfast ? 0 : 2;
  const long long oa0 = (long long)(i0 + ar0) * d.a_rs + (long long)ak0 * d.a_cs;
  const long long ob0 = (long long)bk0 * d.b_rs + (long long)(j0 + bc0) * d.b_cs;
  const long long sa = (long long)a_dr * d.a_rs + (long long)a_dk * d.a_cs;
  const long long sb = (long long)b_dk * d.b_rs + (long long)b_dc * d.b_cs;
  unsigned va = 0, vb = 0;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    if (i0 + ar0 + u * a_dr < M) va |= 1u << u;
    if (j0 + bc0 + u * b_dc < N) vb |= 1u << u;
  }
  int k_lo = 0, k_hi = K;
  if (d.tri & 1) k_hi = min(k_hi, i0 + G64M_T);      // A lower: a[i][k] = 0 for k > i
  if (d.tri & 2) k_lo = max(k_lo, i0);               // A upper: a[i][k] = 0 for k < i
  if (d.tri & 4) k_lo = max(k_lo, j0);               // B lower: b[k][j] = 0 for k < j
  if (d.tri & 8) k_hi = min(k_hi, j0 + G64M_T);      // B upper: b[k][j] = 0 for k > j
  k_lo &= ~(GK - 1);
  if (k_lo >= k_hi && d.beta == 1.0 && d.E == nullptr && d.C32 == nullptr) return;     // nothing to add
  double ra[8], rb[8];
  auto fetch = [&](int k0) __attribute__((always_inline)) {
    const gdbl* Ak = A + (long long)k0 * d.a_cs;
    const gdbl* Bk = B + (long long)k0 * d.b_rs;
    if (va == 0xffu && vb == 0xffu && k0 + GK <= k_hi) {        // interior step (wave-uniform in practice: plain loads)
#pragma unroll
      for (int u = 0; u < 8; ++u) { ra[u] = Ak[oa0 + u * sa]; rb[u] = Bk[ob0 + u * sb]; }
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        ra[u] = (((va >> u) & 1) && k0 + ak0 + u * a_dk < k_hi) ? Ak[oa0 + u * sa] : 0.0;
        rb[u] = (((vb >> u) & 1) && k0 + bk0 + u * b_dk < k_hi) ? Bk[ob0 + u * sb] : 0.0;
      }
    }
  };
  f64x4 acc[4][4] = {};
  if (k_lo < k_hi) fetch(k_lo);
  for (int k0 = k_lo; k0 < k_hi; k0 += GK) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      As[(ak0 + u * a_dk) * G64M_P + ar0 + u * a_dr] = ra[u];
      Bs[(bk0 + u * b_dk) * G64M_P + bc0 + u * b_dc] = rb[u];
    }
    __syncthreads();
    if (k0 + GK < k_hi) fetch(k0 + GK);
#pragma unroll
    for (int ks = 0; ks < GK / 4; ++ks) {
      const int k = 4 * ks + kq;
      double a[4], b[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) a[m] = As[k * G64M_P + 64 * wm + 16 * m + r16];
#pragma unroll
      for (int n = 0; n < 4; ++n) b[n] = Bs[k * G64M_P + 64 * wn + 16 * n + r16];
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + 64 * wm + 16 * m + kq + 4 * r, j = j0 + 64 * wn + 16 * n + r16;
        if (i < M && j < N) gemm64_store(d, i, j, acc[m][n][r]);
      }
}

constexpr int ORDER_UPLOAD_CHUNK = 944;
struct OrderChunk { int v[ORDER_UPLOAD_CHUNK]; };
__global__ void __launch_bounds__(256) order_upload_kernel(int* __restrict__ dst, OrderChunk chunk, int count) {
  for (int w = threadIdx.x; w < count; w += blockDim.x) dst[w] = chunk.v[w];
}

constexpr int GEMM_UPLOAD_CHUNK = 17;
struct GemmChunk { GemmDev f[GEMM_UPLOAD_CHUNK]; };
static_assert(sizeof(GemmChunk) <= 3840, "kernel argument block must stay below 4 KB");

__global__ void __launch_bounds__(256) gemm_upload_kernel(GemmDev* __restrict__ table, GemmChunk chunk, int count) {
  const int words = count * (int)(sizeof(GemmDev) / 4);
  const int* in = reinterpret_cast<const int*>(&chunk);
  int* out = reinterpret_cast<int*>(table);
  for (int w = threadIdx.x; w < words; w += blockDim.x) out[w] = in[w];
}

// ------------------------------------------------------------------------------------------------
// Standard normal noise: Philox4x32-10 counter-based generator + Box-Muller, 4 values per counter.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
  const unsigned long long p0 = 0xD2511F53ull * c[0];
  const unsigned long long p1 = 0xCD9E8D57ull * c[2];
  const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0;
  const unsigned n1 = (unsigned)p1;
  const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1;
  const unsigned n3 = (unsigned)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__global__ void __launch_bounds__(256)
randn_kernel(float* __restrict__ out, long long count, unsigned long long seed, unsigned long long offset,
             const unsigned long long* __restrict__ counter) {
  if (counter != nullptr) offset += *counter;           // device-side stream position (HIP-graph replays advance it)
  const long long nquad = (count + 3) >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < nquad; q += stride) {
    const unsigned long long ctr = offset + (unsigned long long)q;
    unsigned c[4] = {(unsigned)ctr, (unsigned)(ctr >> 32), 0u, 0u};
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      philox_round(c, k0, k1);
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
    float z[4];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const float u1 = ((float)(c[2 * p] >> 8) + 1.0f) * (1.0f / 16777216.0f);      // (0, 1]
      const float u2 = (float)(c[2 * p + 1] >> 8) * (1.0f / 16777216.0f);           // [0, 1)
      const float r = sqrtf(-2.0f * logf(u1));
      float sn, cs;
      sincosf(6.283185307179586f * u2, &sn, &cs);
      z[2 * p] = r * cs;
      z[2 * p + 1] = r * sn;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * q + e < count) out[4 * q + e] = z[e];
  }
}

// the draw is over: advance the device-side stream position (its own launch, ordered behind every reader of the value)
__global__ void counter_add_kernel(unsigned long long* counter, unsigned long long inc) { *counter += inc; }

}  // namespace curv

using namespace curv;

extern "C" size_t curv_gemm_workspace_bytes(int n_desc) {
  return align_up((size_t)std::max(n_desc, 1) * sizeof(GemmDev), 256);
}

// K slices of an NT product.  A launch with fewer tiles than the chip has workgroup slots lasts as long as its
// longest tile: a layer-sharded rank that samples one 512 x 4608 layer runs 144 tiles with K up to 4608.  Such
// launches - and only such: with a whole model's tiles in flight slicing buys nothing (measured) - cut their long
// products into slices of NT_KSLICE, summed deterministically by a second launch that applies the epilogue.
constexpr int NT_KSLICE = 768;
constexpr long long NT_SPLIT_BELOW_TILES = 1024;       // 2 workgroup slots per CU x 256 CUs x 2
// Thin products: one output column (N == 1) with K-contiguous operand rows - the bias column of a sampled layer,
// (m x n) (n x 1).  As a 64x64-tile GEMM that is cdiv(m, 64) workgroups walking all of K one after the other (0.14 ms
// for ResNet-50's fc bias, 1 % MFMA utilisation); here one wave per output row reads its row and the vector with
// 16-byte loads and reduces across lanes: bound by the bytes of A.
__global__ void __launch_bounds__(256) gemv_rows_kernel(const GemmDev* __restrict__ t, int nf) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  long long row = (long long)blockIdx.x * 4 + wave;
  int f = 0;
  while (f < nf && row >= t[f].M) { row -= (t[f].M + 3) / 4 * 4; ++f; }       // rows of a job are padded to whole workgroups
  if (f >= nf || row >= t[f].M || row < 0) return;
  const GemmDev& d = t[f];
  const gfl* a = (const gfl*)d.A + row * d.a_rs;
  const gfl* b = (const gfl*)d.B;
  int K = d.K;
  if (d.tri == CURV_TRI_A_LOWER) K = min(K, (int)row + 1);                     // row i of a lower-triangular A ends at i
  float acc = 0.0f;
  for (int k = lane; k < K; k += 64) acc += a[k] * b[k];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) nt_epilogue(d, (int)row, 0, acc);
}

static bool gemv_eligible(const curv_gemm_desc& s) {
  return s.N == 1 && s.M >= 1 && s.a_cs == 1 && s.b_rs == 1 && s.tri != CURV_TRI_B_UPPER;
}

static bool nt_eligible(const curv_gemm_desc& s) {
  const long long a_ext = ((long long)(s.M - 1) * s.a_rs + s.K) * 4, b_ext = ((long long)(s.N - 1) * s.b_cs + s.K) * 4;
  return s.a_cs == 1 && s.b_rs == 1 && s.M >= 64 && s.N >= 64 && s.K >= 8 && a_ext < (1LL << 31) - 64 &&   // (31 bits: bit 31 of a lane offset marks "no fetch", gemm_nt.h)
         b_ext < (1LL << 31) - 64;
}
static long long nt_tiles_of(const curv_gemm_desc* descs, int n_desc) {
  long long t = 0;
  for (int i = 0; i < n_desc; ++i)
    if (descs[i].M > 0 && descs[i].N > 0 && nt_eligible(descs[i])) t += (long long)cdiv(descs[i].M, 128) * cdiv(descs[i].N, 128);
  return t;
}
static int nt_slices(const curv_gemm_desc& s, bool underfilled) {
  return (underfilled && nt_eligible(s) && s.K >= 2 * NT_KSLICE) ? cdiv(s.K, NT_KSLICE) : 1;
}

// A small launch (see gemm_nt_small_kernel): every product in NT layout and short, and so few 128-wide tiles that most
// of the chip would idle behind them
constexpr long long SMALL_MAX_BLOCKS = 2048;      // 32 x 32 blocks: 8 workgroups per CU
constexpr long long SMALL_MAX_TILES128 = 64;
static bool small_launch(const curv_gemm_desc* descs, int n_desc) {
  long long blocks = 0, tiles128 = 0;
  for (int i = 0; i < n_desc; ++i) {
    const curv_gemm_desc& s = descs[i];
    if (s.M <= 0 || s.N <= 0) continue;
    if (s.a_cs != 1 || s.b_rs != 1 || s.K > 2 * sm::MAX_K || s.a_rs < 0 || s.b_cs < 0) return false;
    blocks += (long long)cdiv(s.M, sm::T) * cdiv(s.N, sm::T);
    tiles128 += (long long)cdiv(s.M, 128) * cdiv(s.N, 128);
  }
  return blocks > 0 && blocks <= SMALL_MAX_BLOCKS && tiles128 <= SMALL_MAX_TILES128;
}

static size_t nt_order_bytes(const curv_gemm_desc* descs, int n_desc) {       // the K-sorted tile list of an unsplit NT launch
  return align_up((size_t)nt_tiles_of(descs, n_desc) * sizeof(int), 256);
}

extern "C" size_t curv_gemm_workspace_bytes_for(const curv_gemm_desc* descs, int n_desc) {
  size_t total = curv_gemm_workspace_bytes(n_desc);
  const bool underfilled = nt_tiles_of(descs, n_desc) < NT_SPLIT_BELOW_TILES;
  if (!underfilled) return total + nt_order_bytes(descs, n_desc);
  for (int i = 0; i < n_desc; ++i) {
    const curv_gemm_desc& s = descs[i];
    if (s.M <= 0 || s.N <= 0) continue;
    const int sl = nt_slices(s, underfilled);
    if (sl > 1) total += (size_t)cdiv(s.M, 128) * cdiv(s.N, 128) * sl * 128 * 128 * sizeof(float);
  }
  return total;
}

static int gemm_batched_impl(void* stream_, const curv_gemm_desc* descs, int n_desc, void* workspace,
                             size_t workspace_bytes, unsigned flags) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_desc == 0) return CURV_OK;
  CURV_REQUIRE(descs != nullptr, "curv_gemm_batched: null descriptor array");
  if (workspace == nullptr || workspace_bytes < curv_gemm_workspace_bytes(n_desc)) {
    set_error("curv_gemm_batched: workspace too small");
    return CURV_ERR_WORKSPACE;
  }
  std::vector<GemmDev> tab, tab_nt, tab_v;      // work lists: the general kernel, the NT / LDS-DMA kernel, thin products
  tab.reserve(n_desc);
  long long tiles = 0, tiles_nt = 0, red_tiles = 0, slab_floats = 0, gemv_wgs = 0;
  const bool small = small_launch(descs, n_desc);      // everything goes to gemm_nt_small_kernel (work list `tab`)
  // K slicing needs the slab area behind the table: only with a workspace sized by curv_gemm_workspace_bytes_for
  const bool underfilled = nt_tiles_of(descs, n_desc) < NT_SPLIT_BELOW_TILES;
  const bool may_split = underfilled && workspace_bytes >= curv_gemm_workspace_bytes_for(descs, n_desc);
  for (int i = 0; i < n_desc; ++i) {
    const curv_gemm_desc& s = descs[i];
    CURV_REQUIRE(s.M >= 0 && s.N >= 0 && s.K >= 0, "curv_gemm_batched: desc %d: negative shape", i);
    if (s.M == 0 || s.N == 0) continue;
    CURV_REQUIRE(s.C != nullptr && (s.K == 0 || (s.A != nullptr && s.B != nullptr)),
                 "curv_gemm_batched: desc %d: null pointer", i);
    CURV_REQUIRE(s.epilogue >= 0 && s.epilogue <= CURV_EPI_MUL_E_ADD_F, "curv_gemm_batched: desc %d: bad epilogue", i);
    CURV_REQUIRE(s.epilogue < CURV_EPI_MUL_E || s.E != nullptr, "curv_gemm_batched: desc %d: epilogue needs E", i);
    CURV_REQUIRE(s.epilogue != CURV_EPI_MUL_E_ADD_F || s.F != nullptr, "curv_gemm_batched: desc %d: epilogue needs F", i);
    GemmDev d;
    memset(&d, 0, sizeof(d));
    d.A = s.A; d.B = s.B; d.C = s.C; d.E = s.E; d.F = s.F;
    d.f_rs = s.f_rs; d.f_cs = s.f_cs;
    d.a_rs = s.a_rs; d.a_cs = s.a_cs; d.b_rs = s.b_rs; d.b_cs = s.b_cs;
    d.c_rs = s.c_rs; d.c_cs = s.c_cs; d.e_rs = s.e_rs; d.e_cs = s.e_cs;
    d.M = s.M; d.N = s.N; d.K = s.K;
    d.epilogue = s.epilogue; d.alpha = s.alpha; d.beta = s.beta;
    CURV_REQUIRE(s.tri >= 0 && s.tri <= CURV_TRI_B_UPPER, "curv_gemm_batched: desc %d: bad tri flag", i);
    d.tri = s.tri;
    d.tm = (s.M >= 96 && s.N >= 96) ? 128 : 64;
    {   // the kernel addresses a tile's operand panels with 32-bit element offsets (non-negative strides)
      const long long lim = (1LL << 31) - 1;
      CURV_REQUIRE(s.a_rs >= 0 && s.a_cs >= 0 && s.b_rs >= 0 && s.b_cs >= 0, "curv_gemm_batched: desc %d: negative stride", i);
      CURV_REQUIRE((long long)d.tm * s.a_rs + 16 * s.a_cs < lim && (long long)d.tm * s.b_cs + 16 * s.b_rs < lim,
                   "curv_gemm_batched: desc %d: operand stride too large", i);
    }
    // NT products with K-contiguous rows on both sides and at least one full-width tile edge go to the LDS-DMA
    // kernel; their operand extents must fit a buffer descriptor (32-bit byte offsets)
    const long long a_ext = ((long long)(s.M - 1) * s.a_rs + s.K) * 4, b_ext = ((long long)(s.N - 1) * s.b_cs + s.K) * 4;
    if (small) {
      d.tm = sm::T;
      d.tiles_n = cdiv(s.N, sm::T);
      d.tile_base = (int)tiles;
      tiles += (long long)cdiv(s.M, sm::T) * d.tiles_n;
      tab.push_back(d);
      continue;
    }
    if (gemv_eligible(s)) {
      gemv_wgs += (s.M + 3) / 4;
      tab_v.push_back(d);
      continue;
    }
    const bool is_nt = nt_eligible(s);
    if (is_nt) {
      d.tm = 128;
      d.a_bytes = (unsigned)a_ext; d.b_bytes = (unsigned)b_ext;
      d.tiles_n = cdiv(s.N, 128);
      d.tile_base = (int)tiles_nt;
      const long long nt_tiles = (long long)cdiv(s.M, 128) * d.tiles_n;
      d.n_slices = may_split ? nt_slices(s, true) : 1;
      d.kslice = NT_KSLICE;
      d.red_base = -1;
      if (d.n_slices > 1) {
        d.red_base = (int)red_tiles;
        d.slab_base = slab_floats;
        red_tiles += nt_tiles;
        slab_floats += nt_tiles * d.n_slices * 128 * 128;
      }
      tiles_nt += nt_tiles * d.n_slices;
      CURV_REQUIRE(tiles_nt < (1LL << 30), "curv_gemm_batched: too many tiles");
      tab_nt.push_back(d);
      continue;
    }
    d.tiles_n = cdiv(s.N, d.tm);
    d.tile_base = (int)tiles;
    tiles += (long long)cdiv(s.M, d.tm) * d.tiles_n;
    CURV_REQUIRE(tiles < (1LL << 30), "curv_gemm_batched: too many tiles");
    tab.push_back(d);
  }
  if (tab.empty() && tab_nt.empty() && tab_v.empty()) return CURV_OK;
  GemmDev* table = reinterpret_cast<GemmDev*>(workspace);
  const int n = (int)tab.size(), n_nt = (int)tab_nt.size(), n_v = (int)tab_v.size();
  std::vector<GemmDev> all(tab);
  all.insert(all.end(), tab_nt.begin(), tab_nt.end());
  all.insert(all.end(), tab_v.begin(), tab_v.end());
  // CURV_GEMM_TABLE_RESIDENT: the caller replays the very same descriptor array into a workspace nobody else has
  // written since the previous call - the device table is still there
  for (int b = 0; b < n + n_nt + n_v && !(flags & CURV_GEMM_TABLE_RESIDENT); b += GEMM_UPLOAD_CHUNK) {
    GemmChunk chunk;
    const int count = std::min(GEMM_UPLOAD_CHUNK, n + n_nt + n_v - b);
    memset(&chunk, 0, sizeof(chunk));
    memcpy(chunk.f, all.data() + b, (size_t)count * sizeof(GemmDev));
    hipLaunchKernelGGL(gemm_upload_kernel, dim3(1), dim3(256), 0, stream, table + b, chunk, count);
    CURV_LAUNCH_CHECK();
  }
  float* slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + curv_gemm_workspace_bytes(n_desc));
  // a full launch (several tiles per workgroup slot, no K slicing): its tiles in descending K order, in the space the
  // slabs of an underfilled launch would take
  const int* order = nullptr;
  if (n_nt > 1 && !underfilled && workspace_bytes >= curv_gemm_workspace_bytes_for(descs, n_desc)) {
    int* dev_order = reinterpret_cast<int*>(slabs);
    order = dev_order;
    if (!(flags & CURV_GEMM_TABLE_RESIDENT)) {
      std::vector<std::pair<int, int>> keyed;               // (K of the tile, global tile id)
      keyed.reserve((size_t)tiles_nt);
      for (const GemmDev& d : tab_nt) {
        const int tiles_m = cdiv(d.M, 128);
        for (int local = 0; local < tiles_m * d.tiles_n; ++local) {
          int tm = local / d.tiles_n, tn = local - tm * d.tiles_n;
          if (d.tri == CURV_TRI_A_LOWER) tm = tiles_m - 1 - tm;
          else if (d.tri == CURV_TRI_B_UPPER) tn = d.tiles_n - 1 - tn;
          int K = d.K;
          if (d.tri == CURV_TRI_A_LOWER) K = std::min(K, tm * 128 + 128);
          else if (d.tri == CURV_TRI_B_UPPER) K = std::min(K, tn * 128 + 128);
          keyed.emplace_back(K, d.tile_base + local);
        }
      }
      std::stable_sort(keyed.begin(), keyed.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first > b.first; });
      for (size_t b = 0; b < keyed.size(); b += ORDER_UPLOAD_CHUNK) {
        OrderChunk chunk;
        const int count = (int)std::min<size_t>(ORDER_UPLOAD_CHUNK, keyed.size() - b);
        for (int k = 0; k < count; ++k) chunk.v[k] = keyed[b + k].second;
        hipLaunchKernelGGL(order_upload_kernel, dim3(1), dim3(256), 0, stream, dev_order + b, chunk, count);
        CURV_LAUNCH_CHECK();
      }
    }
  }
  if (n_nt > 0) {
    hipLaunchKernelGGL(gemm_nt_kernel, dim3((unsigned)tiles_nt), dim3(GEMM_THREADS), 0, stream, table + n, n_nt, slabs, order);
    CURV_LAUNCH_CHECK();
    if (red_tiles > 0) {
      hipLaunchKernelGGL(gemm_nt_reduce_kernel, dim3((unsigned)red_tiles), dim3(256), 0, stream, table + n, n_nt, slabs);
      CURV_LAUNCH_CHECK();
    }
  }
  if (n > 0 && small) {
    hipLaunchKernelGGL(gemm_nt_small_kernel, dim3((unsigned)tiles), dim3(GEMM_THREADS), 0, stream, table, n);
    CURV_LAUNCH_CHECK();
  } else if (n > 0) {
    hipLaunchKernelGGL(gemm_f32_kernel, dim3((unsigned)tiles), dim3(GEMM_THREADS), 0, stream, table, n);
    CURV_LAUNCH_CHECK();
  }
  if (n_v > 0) {
    hipLaunchKernelGGL(gemv_rows_kernel, dim3((unsigned)gemv_wgs), dim3(256), 0, stream, table + n + n_nt, n_v);
    CURV_LAUNCH_CHECK();
  }
  return CURV_OK;
}

extern "C" int curv_gemm_batched(void* stream, const curv_gemm_desc* descs, int n_desc, void* workspace,
                                 size_t workspace_bytes) {
  return gemm_batched_impl(stream, descs, n_desc, workspace, workspace_bytes, 0u);
}

extern "C" int curv_gemm_batched_ex(void* stream, const curv_gemm_desc* descs, int n_desc, void* workspace,
                                    size_t workspace_bytes, unsigned flags) {
  return gemm_batched_impl(stream, descs, n_desc, workspace, workspace_bytes, flags);
}

extern "C" int curv_randn(void* stream, float* out, long long count, unsigned long long seed,
                          unsigned long long offset) {
  if (count <= 0) return CURV_OK;
  CURV_REQUIRE(out != nullptr, "curv_randn: null pointer");
  long long blocks = cdivll(cdivll(count, 4), 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(randn_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, count, seed,
                     offset, (const unsigned long long*)nullptr);
  CURV_LAUNCH_CHECK();
  return CURV_OK;
}

extern "C" int curv_randn_counter(void* stream, float* out, long long count, unsigned long long seed,
                                  unsigned long long* counter) {
  if (count <= 0) return CURV_OK;
  CURV_REQUIRE(out != nullptr && counter != nullptr, "curv_randn_counter: null pointer");
  long long blocks = cdivll(cdivll(count, 4), 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(randn_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, count, seed,
                     0ull, (const unsigned long long*)counter);
  CURV_LAUNCH_CHECK();
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter,
                     (unsigned long long)((count + 3) >> 2));
  CURV_LAUNCH_CHECK();
  return CURV_OK;
}

// products with both output edges >= G64_MACRO_MIN go to the 128 x 128 macro-tile kernel
#ifndef CURV_G64_MACRO_MIN
#define CURV_G64_MACRO_MIN 1024
#endif
constexpr int G64_MACRO_MIN = CURV_G64_MACRO_MIN;
extern "C" int curv_gemm_f64_batched(void* stream_, const curv_gemm64_desc* descs, int n_desc) {
  hipStream_t stream = (hipStream_t)stream_;
  CURV_REQUIRE(n_desc >= 0 && (n_desc == 0 || descs), "curv_gemm_f64_batched: bad arguments");
  for (int i = 0; i < n_desc; ++i) {
    const curv_gemm64_desc& s = descs[i];
    CURV_REQUIRE(s.M > 0 && s.N > 0 && s.K >= 0 && s.A && s.B && (s.C || s.C32), "curv_gemm_f64_batched: desc %d invalid", i);
    CURV_REQUIRE(s.C32 == nullptr || (s.beta == 0.0 && s.E == nullptr), "curv_gemm_f64_batched: desc %d: the fp32 output takes no beta / E", i);
    CURV_REQUIRE(s.E == nullptr || (s.C != nullptr && s.E != s.C), "curv_gemm_f64_batched: desc %d: E needs an output C of its own", i);
    CURV_REQUIRE((s.tri & ~31) == 0 && (s.tri & 3) != 3 && (s.tri & 12) != 12, "curv_gemm_f64_batched: desc %d: bad tri flags", i);
    CURV_REQUIRE(!(s.tri & CURV_TRI64_C_LOWER) || s.M == s.N, "curv_gemm_f64_batched: desc %d: CURV_TRI64_C_LOWER needs a square product", i);
  }
  // launches of up to G64_BATCH descriptors (they travel as kernel arguments), per tile size, in the caller's order
  for (int pass = 0; pass < 2; ++pass) {
    const bool macro = pass == 1;
    const int T = macro ? G64M_T : GT;
    Gemm64Table tab;
    Gemm64Dev* d = tab.d;
    int cnt = 0;
    long long tiles = 0;
    auto flush = [&]() -> int {
      if (cnt == 0) return CURV_OK;
      if (macro) hipLaunchKernelGGL(gemm_f64_macro_kernel, dim3((unsigned)tiles), dim3(GEMM_THREADS), 0, stream, tab, cnt);
      else hipLaunchKernelGGL(gemm_f64_kernel, dim3((unsigned)tiles), dim3(GEMM_THREADS), 0, stream, tab, cnt);
      CURV_LAUNCH_CHECK();
      cnt = 0; tiles = 0;
      return CURV_OK;
    };
    memset(&tab, 0, sizeof(tab));
    for (int i = 0; i < n_desc; ++i) {
      const curv_gemm64_desc& s = descs[i];
      if ((s.M >= G64_MACRO_MIN && s.N >= G64_MACRO_MIN) != macro) continue;
      Gemm64Dev& o = d[cnt];
      o.A = s.A; o.B = s.B; o.C = s.C;
      o.a_rs = s.a_rs; o.a_cs = s.a_cs; o.b_rs = s.b_rs; o.b_cs = s.b_cs; o.c_rs = s.c_rs; o.c_cs = s.c_cs;
      o.M = s.M; o.N = s.N; o.K = s.K; o.alpha = s.alpha; o.beta = s.beta;
      o.tri = s.tri;
      o.E = s.E; o.rs = s.row_scale; o.cs = s.col_scale; o.C32 = s.C32;
      o.tiles_n = cdiv(s.N, T);
      o.tile_base = (int)tiles;
      tiles += (long long)cdiv(s.M, T) * o.tiles_n;
      CURV_REQUIRE(tiles < (1LL << 30), "curv_gemm_f64_batched: too many tiles");
      if (++cnt == G64_BATCH || tiles > (1LL << 24)) { const int rc = flush(); if (rc != CURV_OK) return rc; memset(&tab, 0, sizeof(tab)); }
    }
    const int rc = flush();
    if (rc != CURV_OK) return rc;
  }
  return CURV_OK;
}
