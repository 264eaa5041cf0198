// Which tile form suits the FAR update of the blocked Cholesky sweep (csrc/invert.hip, outer_update_kernel)?
//   C[i][j] -= sum_{k < 256} A[i][k] A[j][k]      for all tiles row0 <= j <= i of three 4608-wide fp64 matrices
// i.e. one outer panel's trailing update of the chain-bound group of ResNet-50 (K = 256 per pass: short tiles, a
// read-modify-write epilogue, a launch of a few rounds).  Forms: the shipped 64x64 tile on 4 waves, and 128x128 tiles on
// 4 / 8 / 16 waves.  Prints TFLOP/s of the executed tiles (diagonal tiles counted in full).
//   hipcc -O3 --offload-arch=gfx950 -o far_update_forms far_update_forms.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include <algorithm>
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

// ---- LDS-DMA form of the 64x64 tile (round 6): buffer_load ... lds straight into a double-buffered, XOR-swizzled
// [64 rows][8 x 16 B] image per operand (16 k per stage), ds_read_b128 operand reads (two k per read: lane quarter kq takes
// k = 8 h + 2 kq + d - both operands agree on the order), one barrier per stage, no staging registers or LDS stores
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) char lds_char;
typedef double f64x2 __attribute__((ext_vector_type(2)));
template <int MINWG, bool FORCE_SAME = false, bool EARLY_C = false>
__global__ void __launch_bounds__(256, MINWG)
far_dma(double* __restrict__ W0, long long mat_stride, int np, int row0_blk, int tiles_per_mat, int Kel) {
  constexpr int ROW_B = 128, TILE_B = 64 * ROW_B;
  __shared__ __attribute__((aligned(1024))) char smem[4 * TILE_B];     // [A buf0][A buf1][B buf0][B buf1]
  lds_char* lds = (lds_char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
  const int r16 = lane & 15, kq = lane >> 4;
  const int mat = blockIdx.x / tiles_per_mat;
  int t = blockIdx.x - mat * tiles_per_mat;
  int a = 0;
  while (t > a) { t -= a + 1; ++a; }
  double* W = W0 + mat * mat_stride;
  const int i0 = (row0_blk + a) * 64, j0 = (row0_blk + t) * 64;
  const bool same = FORCE_SAME || a == t;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (unsigned)((long long)np * np * 8), 0x00020000);
  // DMA lane geometry: piece p (8 rows x 128 B) of an operand = rows 8 p ..; this wave moves pieces `wave` and `wave + 4`
  const int prow = lane >> 3, pslot = lane & 7;
  const int drow = 8 * wave + prow;                                     // (+ 32 for the second piece: same key)
  const int dkey = (drow >> 1) & 7;
  const int voff = (drow * np) * 8 + ((pslot ^ dkey) << 4);
  const int soff_a = (i0 * np) * 8, soff_b = (j0 * np) * 8, half_b = 32 * np * 8;
  auto issue = [&](int ke, unsigned buf) {
    const int kb = ke * 8;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds + buf + wave * 1024), 16, voff, soff_a + kb, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds + buf + (wave + 4) * 1024), 16, voff, soff_a + half_b + kb, 0, 0);
    if (!same) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds + 2 * TILE_B + buf + wave * 1024), 16, voff, soff_b + kb, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds + 2 * TILE_B + buf + (wave + 4) * 1024), 16, voff, soff_b + half_b + kb, 0, 0);
    }
  };
  // operand addresses: block b (16 rows), half h: row * 128 + ((kq + 4 h) ^ key) * 16
  unsigned addr_a[2], addr_b[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int ra = 32 * wm + 16 * m + r16, rb = 32 * wn + 16 * m + r16;
    addr_a[m] = ra * ROW_B + ((kq ^ ((ra >> 1) & 7)) << 4);
    addr_b[m] = (same ? 0 : 2 * TILE_B) + rb * ROW_B + ((kq ^ ((rb >> 1) & 7)) << 4);
  }
  f64x4 acc[2][2] = {};
  const int n_st = Kel / 16;
  gdouble* C = (gdouble*)(W + (long long)i0 * np + j0);
  double oldc[2][2][4];
  if (EARLY_C) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) oldc[m][n][q] = C[(long long)(32 * wm + 16 * m + kq + 4 * q) * np + 32 * wn + 16 * n + r16];
  }
  issue(0, 0);
  for (int st = 0; st < n_st; ++st) {
    __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0)
    __syncthreads();
    const unsigned buf = (st & 1) * TILE_B;
    if (st + 1 < n_st) issue(16 * (st + 1), TILE_B - buf);
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
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m][h][d], bv[n][h][d], acc[m][n], 0, 0, 0);
  }
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    double old[2][4];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) old[n][q] = EARLY_C ? oldc[m][n][q] : C[(long long)(32 * wm + 16 * m + kq + 4 * q) * np + 32 * wn + 16 * n + r16];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) C[(long long)(32 * wm + 16 * m + kq + 4 * q) * np + 32 * wn + 16 * n + r16] = old[n][q] - acc[m][n][q];
  }
}

// ---- the LDS-DMA form as PERSISTENT workgroups: a workgroup walks tiles blockIdx.x, + gridDim.x, ...; the first stage of the
// next tile is fetched behind the last MFMAs of the current one, in front of its read-modify-write epilogue
template <int MINWG>
__global__ void __launch_bounds__(256, MINWG)
far_dma_p(double* __restrict__ W0, long long mat_stride, int np, int row0_blk, int tiles_per_mat, int n_tiles, int Kel) {
  constexpr int ROW_B = 128, TILE_B = 64 * ROW_B;
  __shared__ __attribute__((aligned(1024))) char smem[4 * TILE_B];
  lds_char* lds = (lds_char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
  const int r16 = lane & 15, kq = lane >> 4;
  const int drow = 8 * wave + (lane >> 3);
  const int voff = (drow * np) * 8 + (((lane & 7) ^ ((drow >> 1) & 7)) << 4);
  const int half_b = 32 * np * 8;
  unsigned addr_a[2], addr_b0[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int ra = 32 * wm + 16 * m + r16, rb = 32 * wn + 16 * m + r16;
    addr_a[m] = ra * ROW_B + ((kq ^ ((ra >> 1) & 7)) << 4);
    addr_b0[m] = rb * ROW_B + ((kq ^ ((rb >> 1) & 7)) << 4);
  }
  auto decode = [&](int tile, int& mat, int& i0, int& j0) {
    mat = tile / tiles_per_mat;
    int t = tile - mat * tiles_per_mat, a = 0;
    while (t > a) { t -= a + 1; ++a; }
    i0 = (row0_blk + a) * 64; j0 = (row0_blk + t) * 64;
  };
  auto issue = [&](const __amdgpu_buffer_rsrc_t& rs, int i0, int j0, int ke, unsigned buf) {
    const int kb = ke * 8, sa = (i0 * np) * 8, sb = (j0 * np) * 8;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds + buf + wave * 1024), 16, voff, sa + kb, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds + buf + (wave + 4) * 1024), 16, voff, sa + half_b + kb, 0, 0);
    if (i0 != j0) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds + 2 * TILE_B + buf + wave * 1024), 16, voff, sb + kb, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds + 2 * TILE_B + buf + (wave + 4) * 1024), 16, voff, sb + half_b + kb, 0, 0);
    }
  };
  const int n_st = Kel / 16;
  int tile = blockIdx.x;
  if (tile >= n_tiles) return;
  int mat, i0, j0;
  decode(tile, mat, i0, j0);
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(W0 + mat * mat_stride), 0, (unsigned)((long long)np * np * 8), 0x00020000);
  unsigned par = 0;                              // buffer of the stage about to be consumed
  issue(rs, i0, j0, 0, 0);
  while (true) {
    const bool same = i0 == j0;
    f64x4 acc[2][2] = {};
    const int next = tile + gridDim.x;
    int nmat = 0, ni0 = 0, nj0 = 0;
    if (next < n_tiles) decode(next, nmat, ni0, nj0);
    for (int st = 0; st < n_st; ++st) {
      __builtin_amdgcn_s_waitcnt(0x0f70);
      __syncthreads();
      const unsigned buf = par * TILE_B;
      par ^= 1;
      if (st + 1 < n_st) issue(rs, i0, j0, 16 * (st + 1), par * TILE_B);
      else if (next < n_tiles) {
        const __amdgpu_buffer_rsrc_t nrs = __builtin_amdgcn_make_buffer_rsrc((void*)(W0 + nmat * mat_stride), 0, (unsigned)((long long)np * np * 8), 0x00020000);
        issue(nrs, ni0, nj0, 0, par * TILE_B);
      }
      f64x2 av[2][2], bv[2][2];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          av[m][h] = *reinterpret_cast<const __attribute__((address_space(3))) f64x2*>(lds + buf + (addr_a[m] ^ (h << 6)));
          bv[m][h] = *reinterpret_cast<const __attribute__((address_space(3))) f64x2*>(lds + buf + (same ? 0 : 2 * TILE_B) + (addr_b0[m] ^ (h << 6)));
        }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m][h][d], bv[n][h][d], acc[m][n], 0, 0, 0);
    }
    gdouble* C = (gdouble*)(W0 + mat * mat_stride + (long long)i0 * np + j0);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      double old[2][4];
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) old[n][q] = C[(long long)(32 * wm + 16 * m + kq + 4 * q) * np + 32 * wn + 16 * n + r16];
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) C[(long long)(32 * wm + 16 * m + kq + 4 * q) * np + 32 * wn + 16 * n + r16] = old[n][q] - acc[m][n][q];
    }
    if (next >= n_tiles) break;
    tile = next; mat = nmat; i0 = ni0; j0 = nj0;
    rs = __builtin_amdgcn_make_buffer_rsrc((void*)(W0 + mat * mat_stride), 0, (unsigned)((long long)np * np * 8), 0x00020000);
  }
}

// ---- LDS-DMA form with THREE stage buffers: the pieces of stage st + 2 are issued at stage st and have two stages to land
// (a stage is only 16 MFMAs = 1024 cycles per wave: with two buffers a wave meets its own DMA's latency at every barrier)
template <int MINWG>
__global__ void __launch_bounds__(256, MINWG)
far_dma3(double* __restrict__ W0, long long mat_stride, int np, int row0_blk, int tiles_per_mat, int Kel) {
  constexpr int ROW_B = 128, TILE_B = 64 * ROW_B, NBUF = 3;
  __shared__ __attribute__((aligned(1024))) char smem[2 * NBUF * TILE_B];     // [A x NBUF][B x NBUF]
  lds_char* lds = (lds_char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
  const int r16 = lane & 15, kq = lane >> 4;
  const int mat = blockIdx.x / tiles_per_mat;
  int t = blockIdx.x - mat * tiles_per_mat;
  int a = 0;
  while (t > a) { t -= a + 1; ++a; }
  double* W = W0 + mat * mat_stride;
  const int i0 = (row0_blk + a) * 64, j0 = (row0_blk + t) * 64;
  const bool same = a == t;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (unsigned)((long long)np * np * 8), 0x00020000);
  const int drow = 8 * wave + (lane >> 3);
  const int voff = (drow * np) * 8 + (((lane & 7) ^ ((drow >> 1) & 7)) << 4);
  const int soff_a = (i0 * np) * 8, soff_b = (j0 * np) * 8, half_b = 32 * np * 8;
  auto issue = [&](int ke, unsigned buf) {
    const int kb = ke * 8;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds + buf + wave * 1024), 16, voff, soff_a + kb, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds + buf + (wave + 4) * 1024), 16, voff, soff_a + half_b + kb, 0, 0);
    if (!same) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds + NBUF * TILE_B + buf + wave * 1024), 16, voff, soff_b + kb, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds + NBUF * TILE_B + buf + (wave + 4) * 1024), 16, voff, soff_b + half_b + kb, 0, 0);
    }
  };
  unsigned addr_a[2], addr_b[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int ra = 32 * wm + 16 * m + r16, rb = 32 * wn + 16 * m + r16;
    addr_a[m] = ra * ROW_B + ((kq ^ ((ra >> 1) & 7)) << 4);
    addr_b[m] = (same ? 0 : NBUF * TILE_B) + rb * ROW_B + ((kq ^ ((rb >> 1) & 7)) << 4);
  }
  f64x4 acc[2][2] = {};
  const int n_st = Kel / 16;
  issue(0, 0);
  if (n_st > 1) issue(16, TILE_B);
  unsigned cur = 0;                            // buffer of stage st
  for (int st = 0; st < n_st; ++st) {
    // all but the newest issue (stage st + 1) have landed: 4 (2 on a diagonal tile) pieces may still fly
    if (st + 1 < n_st) { if (same) __builtin_amdgcn_s_waitcnt(0x0f72); else __builtin_amdgcn_s_waitcnt(0x0f74); }
    else __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    const unsigned buf = cur * TILE_B;
    const unsigned nxt2 = (cur >= 1 ? cur - 1 : NBUF - 1);      // the buffer stage st - 1 used: free now
    if (st + 2 < n_st) issue(16 * (st + 2), nxt2 * TILE_B);
    cur = cur + 1 == NBUF ? 0 : cur + 1;
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
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m][h][d], bv[n][h][d], acc[m][n], 0, 0, 0);
  }
  gdouble* C = (gdouble*)(W + (long long)i0 * np + j0);
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    double old[2][4];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) old[n][q] = C[(long long)(32 * wm + 16 * m + kq + 4 * q) * np + 32 * wn + 16 * n + r16];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) C[(long long)(32 * wm + 16 * m + kq + 4 * q) * np + 32 * wn + 16 * n + r16] = old[n][q] - acc[m][n][q];
  }
}

// fp64 reference of one tile's update (for the check): out = sum_k A[i0 + r][k] A[j0 + c][k]
__global__ void ref_tile(const double* W, int np, int i0, int j0, int Kel, double* out) {
  const int r = blockIdx.x, c = threadIdx.x;
  double s = 0;
  for (int k = 0; k < Kel; ++k) s += W[(long long)(i0 + r) * np + k] * W[(long long)(j0 + c) * np + k];
  out[r * 64 + c] = s;
}

int main() {
  const int np = 4608, P = np / 64, mats = 3, Kel = 256, row0 = 8;
  double* W;
  const size_t bytes = (size_t)mats * np * np * 8;
  hipMalloc(&W, bytes);
  hipMemset(W, 0, bytes);
  {  // random K columns (the first 512 of every row), zero elsewhere: the update's operands
    std::vector<double> h((size_t)np * 512);
    unsigned st = 12345u;
    for (auto& v : h) { st = st * 1664525u + 1013904223u; v = ((st >> 8) & 0xffff) / 65536.0 - 0.4; }
    for (int m = 0; m < mats; ++m) hipMemcpy2D(W + (size_t)m * np * np, (size_t)np * 8, h.data(), 512 * 8, 512 * 8, np, hipMemcpyHostToDevice);
  }
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
  run("64x64 LDS-DMA tile, 4 waves, 3/CU", 64, [&](int g, int t, int r) { hipLaunchKernelGGL((far_dma<3>), dim3(g), dim3(256), 0, 0, W, (long long)np * np, np, row0, t, Kel); });
  run("64x64 LDS-DMA tile, 4 waves, 4/CU", 64, [&](int g, int t, int r) { hipLaunchKernelGGL((far_dma<4>), dim3(g), dim3(256), 0, 0, W, (long long)np * np, np, row0, t, Kel); });
  run("64x64 LDS-DMA tile, 4 waves, 5/CU", 64, [&](int g, int t, int r) { hipLaunchKernelGGL((far_dma<5>), dim3(g), dim3(256), 0, 0, W, (long long)np * np, np, row0, t, Kel); });
  run("64x64 LDS-DMA, 4/CU, ONLY the A panel staged (wrong sums: half the operand traffic)", 64, [&](int g, int t, int r) { hipLaunchKernelGGL((far_dma<4, true>), dim3(g), dim3(256), 0, 0, W, (long long)np * np, np, row0, t, Kel); });
  run("64x64 LDS-DMA, old C loaded at the start, 3/CU", 64, [&](int g, int t, int r) { hipLaunchKernelGGL((far_dma<3, false, true>), dim3(g), dim3(256), 0, 0, W, (long long)np * np, np, row0, t, Kel); });
  run("64x64 LDS-DMA, old C loaded at the start, 4/CU", 64, [&](int g, int t, int r) { hipLaunchKernelGGL((far_dma<4, false, true>), dim3(g), dim3(256), 0, 0, W, (long long)np * np, np, row0, t, Kel); });
  run("64x64 LDS-DMA, three stage buffers, 2/CU", 64, [&](int g, int t, int r) { hipLaunchKernelGGL((far_dma3<2>), dim3(g), dim3(256), 0, 0, W, (long long)np * np, np, row0, t, Kel); });
  run("64x64 LDS-DMA, three stage buffers, 3/CU", 64, [&](int g, int t, int r) { hipLaunchKernelGGL((far_dma3<3>), dim3(g), dim3(256), 0, 0, W, (long long)np * np, np, row0, t, Kel); });
  for (int wgs : {3, 4, 5}) {
    char name[96];
    snprintf(name, sizeof(name), "64x64 LDS-DMA persistent, %d/CU", wgs);
    run(name, 64, [&](int g, int t, int r) {
      if (wgs == 3) hipLaunchKernelGGL((far_dma_p<3>), dim3(256 * 3), dim3(256), 0, 0, W, (long long)np * np, np, row0, t, g, Kel);
      else if (wgs == 4) hipLaunchKernelGGL((far_dma_p<4>), dim3(256 * 4), dim3(256), 0, 0, W, (long long)np * np, np, row0, t, g, Kel);
      else hipLaunchKernelGGL((far_dma_p<5>), dim3(256 * 5), dim3(256), 0, 0, W, (long long)np * np, np, row0, t, g, Kel);
    });
  }
  {  // check: one launch of each 64x64 form on a fresh matrix, tile (a, t) = (5, 2) and the diagonal tile (3, 3)
    for (int form = 0; form < 4; ++form) {
      hipMemset(W + (size_t)(row0 * 64) * np, 0, 8);   // (C starts as whatever the runs above left: compare the DIFFERENCE of one launch)
      std::vector<double> before(64 * 64), after(64 * 64), ref(64 * 64);
      const int rows = P - row0, tiles = rows * (rows + 1) / 2;
      double* dref; hipMalloc(&dref, 64 * 64 * 8);
      double worst = 0;
      for (int which = 0; which < 2; ++which) {
        const int a = which ? 3 : 5, tt = which ? 3 : 2;
        const int i0 = (row0 + a) * 64, j0 = (row0 + tt) * 64;
        hipMemcpy2D(before.data(), 64 * 8, W + (size_t)i0 * np + j0, (size_t)np * 8, 64 * 8, 64, hipMemcpyDeviceToHost);
        if (form == 0) hipLaunchKernelGGL((far<64, 2, 2, 3>), dim3(tiles), dim3(256), 0, 0, W, (long long)np * np, np, row0, rows, tiles, Kel);
        else if (form == 1) hipLaunchKernelGGL((far_dma<4>), dim3(tiles), dim3(256), 0, 0, W, (long long)np * np, np, row0, tiles, Kel);
        else if (form == 2) hipLaunchKernelGGL((far_dma_p<4>), dim3(1024), dim3(256), 0, 0, W, (long long)np * np, np, row0, tiles, tiles, Kel);
        else hipLaunchKernelGGL((far_dma3<3>), dim3(tiles), dim3(256), 0, 0, W, (long long)np * np, np, row0, tiles, Kel);
        hipMemcpy2D(after.data(), 64 * 8, W + (size_t)i0 * np + j0, (size_t)np * 8, 64 * 8, 64, hipMemcpyDeviceToHost);
        ref_tile<<<64, 64>>>(W, np, i0, j0, Kel, dref);
        hipMemcpy(ref.data(), dref, 64 * 64 * 8, hipMemcpyDeviceToHost);
        for (int e = 0; e < 64 * 64; ++e) worst = std::max(worst, std::abs((before[e] - after[e]) - ref[e]));
      }
      printf("check %s: max |(C_before - C_after) - A A^T| = %.3e\n", form == 0 ? "shipped" : form == 1 ? "LDS-DMA" : form == 2 ? "LDS-DMA persistent" : "LDS-DMA 3 buffers", worst);
      hipFree(dref);
    }
  }
  // a long-K control: the same forms with K = 2048 (steady state)
  return 0;
}
