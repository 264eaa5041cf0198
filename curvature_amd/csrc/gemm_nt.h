// The 128 x 128 NT tile product of the fp32 GEMMs (gemm.hip: samplers, EFB) as a header, because the inversion sweep
// (invert.hip: the fp32 accumulation of the triangular inverse) runs its rank-K updates through the same tile code.
#pragma once
#include "common.h"

namespace curv {

constexpr int GEMM_THREADS = 256;

struct GemmDev {
  const float* A;
  const float* B;
  float* C;
  const float* E;         // optional elementwise operand of the epilogue
  const float* F;         // optional second (additive) operand: CURV_EPI_MUL_E_ADD_F
  long long a_rs, a_cs;   // op(A) is M x K: element (i, k) at A[i*a_rs + k*a_cs]
  long long b_rs, b_cs;   // op(B) is K x N: element (k, j) at B[k*b_rs + j*b_cs]
  long long c_rs, c_cs;
  long long e_rs, e_cs;
  long long f_rs, f_cs;
  int M, N, K;
  int epilogue;
  float alpha, beta;
  int tiles_n, tile_base;
  int tm;                 // tile edge: 64 or 128
  int tri;                // CURV_TRI_*: triangular operand -> shorter K range per tile
  unsigned a_bytes, b_bytes;   // NT kernel: extents of the two operands (buffer range check)
  // NT kernel, split K: a tile's K range is cut into slices of kslice elements; item = tile * n_slices + slice, raw
  // partial tiles go to slabs and gemm_nt_reduce_kernel sums them in slice order and applies the epilogue
  int kslice, n_slices;        // n_slices <= 1: no split
  int red_base, pad2;          // first workgroup of this product in the reduce launch (-1: not split)
  long long slab_base;         // floats into the slab area
};

typedef __attribute__((address_space(1))) float gfl;

// ------------------------------------------------------------------------------------------------
// NT products with K-contiguous operands on both sides (A[i][k] at A + i a_rs + k, B[k][j] at B + j b_cs + k):
// every product of KFAC.sample_and_replace (L_G z^T, then (.) L_A^T) and of EFB.sample.  Staged like the flat factor
// build (syrk_flat.hip): buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction, into a double-buffered
// [128 rows][8 x 16 B] image per operand, 16-byte slots XOR-swizzled by (row >> 1) & 7 on the SOURCE side, operands
// back by conflict-free ds_read_b128 (one read = 4 k values of one row = the input of 4 MFMAs); lane half h takes k
// group 2 j + h of a step; no staging registers, no LDS store pass; 64 KiB of LDS, two workgroups per CU.
// Rows beyond M / N are clamped to the last row (their results are never stored); k beyond K is zeroed in the last
// step; triangular operands cut the K range per tile (what lies beyond the cut inside the last step is stored zeros).
// ------------------------------------------------------------------------------------------------
#ifndef CURV_NT_KC
#define CURV_NT_KC 32
#endif
namespace nt {
constexpr int TM = 128, KC = CURV_NT_KC, ROW_B = KC * 4, SLOTS = KC / 4, STEPS = KC / 8, RPP = 1024 / ROW_B;
constexpr int PIECES = TM / RPP / 4, PANEL_B = TM * ROW_B, LDS_B = 4 * PANEL_B, NP = 2 * PIECES;
constexpr int PPS = (NP + STEPS / 2 - 1) / (STEPS / 2);
static_assert(PPS <= 4, "at most one DMA piece per MFMA group");
constexpr int KEY_SHIFT = SLOTS == 8 ? 1 : 2, LANES_PER_ROW_SHIFT = SLOTS == 8 ? 3 : 2;   // see syrk_flat.hip
constexpr int WGS = KC == 32 ? 2 : 4;       // workgroups per CU (64 / 32 KiB of LDS)
static_assert(SLOTS == 8 || SLOTS == 4, "stage rows of 32 or 16 k values");
}  // namespace nt
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(3))) char lds_char_t;

// C[i][j] = epilogue(alpha * acc) [+ beta * C[i][j]]
__device__ __forceinline__ void nt_epilogue(const GemmDev& d, int i, int j, float acc) {
  gfl* C = (gfl*)d.C;
  const gfl* E = (const gfl*)d.E;
  const gfl* F = (const gfl*)d.F;
  const int ep = d.epilogue;
  const long long ci = i * d.c_rs + j * d.c_cs;
  float v = d.alpha * acc;
  if (ep == CURV_EPI_SQUARE) v = d.alpha * acc * acc;
  else if (ep == CURV_EPI_MUL_E) v *= E[i * d.e_rs + j * d.e_cs];
  else if (ep == CURV_EPI_ADD_E) v += E[i * d.e_rs + j * d.e_cs];
  else if (ep == CURV_EPI_MUL_E_ADD_F) v = v * E[i * d.e_rs + j * d.e_cs] + F[i * d.f_rs + j * d.f_cs];
  if (d.beta != 0.0f) v += d.beta * C[ci];
  C[ci] = v;
}

__device__ __forceinline__ void gemm_nt_tile(const GemmDev& d, int local, lds_char_t* lds, float* __restrict__ slabs) {
  using namespace nt;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const bool split = d.n_slices > 1;
  const int item = local;
  int slice = 0;
  if (split) { slice = local % d.n_slices; local /= d.n_slices; }
  int tm = local / d.tiles_n, tn = local - tm * d.tiles_n;
  if (d.tri == CURV_TRI_A_LOWER) tm = (d.M + TM - 1) / TM - 1 - tm;        // long tiles first
  else if (d.tri == CURV_TRI_B_UPPER) tn = d.tiles_n - 1 - tn;
  const int i0 = tm * TM, j0 = tn * TM, M = d.M, N = d.N;
  int K = d.K;
  const bool cut = (d.tri == CURV_TRI_A_LOWER && i0 + TM < K) || (d.tri == CURV_TRI_B_UPPER && j0 + TM < K);
  if (d.tri == CURV_TRI_A_LOWER) K = min(K, i0 + TM);
  else if (d.tri == CURV_TRI_B_UPPER) K = min(K, j0 + TM);
  // this item's part of the K range: [kb, K) (K becomes the slice's end)
  int kb = 0;
  bool inner = false;                                // a slice that ends inside the tile's K range: whole steps
  if (split) {
    kb = slice * d.kslice;
    if (kb >= K) return;                             // the triangle cut this slice away
    inner = kb + d.kslice < K;
    K = min(K, kb + d.kslice);
  }
  (void)cut; (void)inner;
  const int n_stages = (K - kb + KC - 1) / KC;
  // the last stage holds k values at or behind K when the range is no multiple of KC (never behind a triangular cut or
  // inside a split: those end on whole tiles / slices): what the DMA leaves there is zeroed in the operand registers

  // DMA lane geometry (see syrk_flat.hip): piece `slot` of this wave covers panel rows 32 slot + 8 wave + (lane >> 3)
  const int rsub = RPP * wave + (lane >> LANES_PER_ROW_SHIFT);
  const int g_lane = (lane & (SLOTS - 1)) ^ ((rsub >> KEY_SHIFT) & (SLOTS - 1));
  // per-piece row offsets: rows beyond the matrix are clamped to its last row
  int voff_a[PIECES], voff_b[PIECES];
#pragma unroll
  for (int p = 0; p < PIECES; ++p) {
    const int ra = min(i0 + 4 * RPP * p + rsub, M - 1), rb = min(j0 + 4 * RPP * p + rsub, N - 1);
    voff_a[p] = (int)(((long long)ra * d.a_rs + 4 * g_lane) * 4);
    voff_b[p] = (int)(((long long)rb * d.b_cs + 4 * g_lane) * 4);
  }
  const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)d.A, 0, d.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)d.B, 0, d.b_bytes, 0x00020000);

  unsigned addr[4][STEPS];
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    const int R = ((o < 2) ? 64 * wm : 64 * wn) + (o & 1) * 32 + r32;
    const unsigned pbase = (o < 2) ? 0u : 2u * PANEL_B;
    const int rkey = (R >> KEY_SHIFT) & (SLOTS - 1);
#pragma unroll
    for (int j = 0; j < STEPS; ++j) addr[o][j] = pbase + R * ROW_B + (((2 * j + h) ^ rkey) << 4);
  }
  f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};

  // Every stage runs as straight-line code (round 6, as syrk_flat.hip: the round-5 form carried step counts, the k tail and
  // the DMA predicate as run-time conditions, i.e. a scalar branch around every group of MFMAs and an exec-mask change
  // around every piece).  A lane whose 16-byte group lies at or behind K - or any lane behind the item's last stage -
  // carries an out-of-range voffset instead (the descriptor's range check drops the fetch; operand extents stay below
  // 2^31 bytes: launch_gemm_nt's eligibility).
  constexpr int OOB = (int)0x80000000;
  auto issue = [&](int i, bool live, int k0n, unsigned nbuf) {
    const int p = i / PIECES, slot = i % PIECES;
    const unsigned lbase = (p ? 2u * PANEL_B : 0u) + nbuf + (unsigned)(RPP * wave + 4 * RPP * slot) * ROW_B;
    if (p == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lds_void_t*)(lds + lbase), 16, live ? voff_a[slot] : OOB, k0n * 4, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (lds_void_t*)(lds + lbase), 16, live ? voff_b[slot] : OOB, k0n * 4, 0, 0);
  };
  if (n_stages > 0) {
    const bool live = kb + 4 * g_lane < K;
#pragma unroll
    for (int i = 0; i < NP; ++i) issue(i, live, kb, 0u);
  }
  for (int t = 0; t < n_stages; ++t) {
    __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0): this wave's DMA of stage t has landed
    __syncthreads();
    const bool more = t + 1 < n_stages;
    const int k0n = kb + (t + 1) * KC;
    const bool live_n = more && k0n + 4 * g_lane < K;
    const unsigned buf = (unsigned)(t & 1) * PANEL_B, nbuf = PANEL_B - buf;
    const int kbase = kb + t * KC;
    const bool tail_stage = kbase + KC > K;    // (the last stage of a range that is no multiple of KC)
    auto rd = [&](int o, int j) { return *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(lds + addr[o][j] + buf); };
    auto mask_step = [&](int j, f32x4& xa0, f32x4& xa1, f32x4& xb0, f32x4& xb1) {
      if (tail_stage) {
        asm volatile("; k tail" ::: "memory");             // keeps this a branch around a VALU-only block
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool gone = kbase + 4 * (2 * j + h) + e >= K;
          xa0[e] = gone ? 0.0f : xa0[e]; xa1[e] = gone ? 0.0f : xa1[e];
          xb0[e] = gone ? 0.0f : xb0[e]; xb1[e] = gone ? 0.0f : xb1[e];
        }
      }
    };
    f32x4 a0 = rd(0, 0), a1 = rd(1, 0), b0 = rd(2, 0), b1 = rd(3, 0);
#pragma unroll
    for (int j = 0; j < STEPS; ++j) {
      mask_step(j, a0, a1, b0, b1);
      f32x4 na0, na1, nb0, nb1;
      if (j + 1 < STEPS) { na0 = rd(0, j + 1); na1 = rd(1, j + 1); nb0 = rd(2, j + 1); nb1 = rd(3, j + 1); }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], c00, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b1[e], c01, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b0[e], c10, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], c11, 0, 0, 0);
        if (e < PPS && PPS * j + e < NP) issue(PPS * j + e, live_n, k0n, nbuf);     // one piece behind a group of MFMAs
      }
      if (j + 1 < STEPS) { a0 = na0; a1 = na1; b0 = nb0; b1 = nb1; }
    }
  }

  // C/D map of the 32x32 block: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  if (split) {
    // raw partial tile, row-major 128 x 128, to this item's slab
    gfl* slab = (gfl*)slabs + d.slab_base + (long long)item * (TM * TM);
    auto store_raw = [&](const f32x16& acc, int m, int n) {
      const int c = 64 * wn + 32 * n + r32;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int r = 64 * wm + 32 * m + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        slab[r * TM + c] = acc[reg];
      }
    };
    store_raw(c00, 0, 0);
    store_raw(c01, 0, 1);
    store_raw(c10, 1, 0);
    store_raw(c11, 1, 1);
    return;
  }
  auto store_block = [&](const f32x16& acc, int m, int n) {
    const int j = j0 + 64 * wn + 32 * n + r32;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int i = i0 + 64 * wm + 32 * m + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      if (i < M && j < N) nt_epilogue(d, i, j, acc[reg]);
    }
  };
  store_block(c00, 0, 0);
  store_block(c01, 0, 1);
  store_block(c10, 1, 0);
  store_block(c11, 1, 1);
}

}  // namespace curv
