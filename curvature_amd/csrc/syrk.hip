// KFAC factor build on gfx950: grouped, symmetric, implicit-im2col SYRK on the fp32 MFMA.
//
//   dst (+)= scale * X X^T,   X = unfold(src) (+ ones row)       (curvature/curvatures.py:329-350)
//
// Design (see DESIGN.md section "K1"):
//   * one launch covers every Kronecker factor of a model.  The work list is implicit:
//       item -> (factor, k-slice, upper-triangular 64x64 tile), decoded on the device from a
//     small descriptor table, so nothing but that table is uploaded per call;
//   * a workgroup stages the RAW activation patch of the channels its two 64-row panels touch in LDS
//     (never the unfolded matrix: a 3x3 conv reads each input pixel once per chunk instead of nine
//     times) and MFMA operands are gathered from the patch with per-lane addresses
//       addr(i, k) = lane_base(i) + ktab[k],
//     lane_base encodes (channel, kh, kw), ktab[k] encodes (sample, out row, out col) of the chunk;
//     row/plane strides are padded so that 32 consecutive unfolded rows hit 32 distinct banks;
//   * the bias row of ones and the zero padding rows are two constant LDS words;
//   * the four waves of a workgroup split the K range of a chunk and each owns the full 64x64 tile as
//     2x2 v_mfma_f32_32x32x2_f32 blocks (diagonal tiles skip the redundant lower-left block);
//   * partial tiles go to fp32 slabs and a second kernel sums the k-slices in a fixed order, applies
//     the scale and adds into the factor and its mirror image: deterministic, exactly symmetric.
#include "common.h"
#include "../../include/curv_hip.h"

#include <algorithm>
#include <vector>

namespace curv {

constexpr int SYRK_THREADS = 256;
constexpr int TM = 64;                 // tile edge (rows of X per panel)
constexpr int XCD_GROUP = 32;          // consecutive items that share an XCD
constexpr int KTAB_MAX = 1024;         // k values per chunk
constexpr int ROWTAB_MAX = 512;        // patch rows per panel per chunk
constexpr int PATCH_WORDS = 16384;     // both panels; also the 4 x (64x64) cross-wave reduce scratch
constexpr int PANEL_WORDS = PATCH_WORDS / 2;
// LDS word offsets
constexpr int ZERO_OFF = 0;
constexpr int ONE_OFF = 1;
constexpr int KTAB_OFF = 16;
constexpr int ROWTAB_OFF = KTAB_OFF + KTAB_MAX;
constexpr int PATCH_OFF = ROWTAB_OFF + 3 * ROWTAB_MAX;
constexpr int SMEM_WORDS = PATCH_OFF + PATCH_WORDS;   // 18960 words = 75840 B -> 2 workgroups per CU

struct FactorDev {
  const float* src;
  float* dst;
  int N, C, H, W;          // source geometry (1x1/stride-1 convs arrive flattened to H = 1)
  int kh, kw, sh, sw, ph, pw;
  int Ho, Wo;
  int khkw;
  int rows, dim, has_bias;
  int compact;             // kh == kw == 1: patch holds only the sampled pixels
  int NS, R, Wc;           // chunk extent: samples, output rows, output cols
  int n_rg, n_cg;          // chunk grid (rows, cols); samples outermost
  int n_chunks;
  int RS, PS, SS, nch;     // LDS strides in words, channels per panel
  int P, n_tiles;
  int cpi, n_slices;       // chunks per item, k-slices
  int item_base, n_items;
  int tile_base;
  int first;
  float scale;
  int pad0;
  long long slab_base;     // in floats
};

__device__ __forceinline__ int find_segment(const FactorDev* __restrict__ descs, int n_factors, int id,
                                            bool by_tile) {
  // largest f with base[f] <= id; bases are ascending.  One ballot per 64 factors.
  const int lane = threadIdx.x & 63;
  int count = 0;
  for (int f0 = 0; f0 < n_factors; f0 += 64) {
    const int f = f0 + lane;
    bool le = false;
    if (f < n_factors) le = (by_tile ? descs[f].tile_base : descs[f].item_base) <= id;
    count += __popcll(__ballot(le));
  }
  return __builtin_amdgcn_readfirstlane(count - 1);
}

__device__ __forceinline__ void decode_tile(int t, int P, int& ti, int& tj) {
  ti = 0;
  while (t >= P - ti) { t -= P - ti; ++ti; }
  tj = ti + t;
}

__global__ void __launch_bounds__(SYRK_THREADS, 2)
syrk_patch_kernel(const FactorDev* __restrict__ descs, int n_factors, int n_items,
                  float* __restrict__ slabs) {
  __shared__ __attribute__((aligned(16))) int smem[SMEM_WORDS];
  float* fs = reinterpret_cast<float*>(smem);
  int* ktab = smem + KTAB_OFF;
  int* rowtab = smem + ROWTAB_OFF;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31;
  const int h = lane >> 5;

  // XCD-aware item order: workgroups that share an XCD (equal blockIdx % 8) take every 8th group of
  // XCD_GROUP consecutive items, i.e. neighbouring tiles of one k-slice of one factor, so the panels
  // they stage hit that XCD's L2, while every XCD still sees an even mix of all factors.
  int item;
  {
    const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
    item = ((j / XCD_GROUP) * 8 + xcd) * XCD_GROUP + (j % XCD_GROUP);
  }
  if (item >= n_items) return;
  const int f = find_segment(descs, n_factors, item, false);
  const FactorDev& d = descs[f];

  const int local = item - d.item_base;
  const int n_tiles = d.n_tiles;
  const int slice = local / n_tiles;
  const int tile = local - slice * n_tiles;
  int ti, tj;
  decode_tile(tile, d.P, ti, tj);
  const bool diag = (ti == tj);
  const int i0 = ti * TM, j0 = tj * TM;

  const float* __restrict__ src = d.src;
  const int N = d.N, C = d.C, H = d.H, W = d.W;
  const int kh = d.kh, kw = d.kw, sh = d.sh, sw = d.sw, ph = d.ph, pw = d.pw;
  const int Ho = d.Ho, Wo = d.Wo, khkw = d.khkw, rows = d.rows, has_bias = d.has_bias;
  const int compact = d.compact;
  const int NS = d.NS, R = d.R, Wc = d.Wc, n_rg = d.n_rg, n_cg = d.n_cg, n_chunks = d.n_chunks;
  const int RS = d.RS, PS = d.PS, SS = d.SS, nch = d.nch;
  const int HW = H * W;

  const int c_lo_i = i0 / khkw, c_lo_j = j0 / khkw;
  const int off_i = 0, off_j = diag ? 0 : PANEL_WORDS;

  if (tid == 0) { fs[ZERO_OFF] = 0.0f; fs[ONE_OFF] = 1.0f; }

  // Per-lane operand rows: A0/A1 = panel i rows r32, 32 + r32; B0/B1 = panel j.
  int base[4], kmask[4];
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    const int i = ((o < 2) ? i0 : j0) + (o & 1) * 32 + r32;
    const int c_lo = (o < 2) ? c_lo_i : c_lo_j;
    const int poff = (o < 2) ? off_i : off_j;
    if (i < rows) {
      const int c = i / khkw;
      const int rem = i - c * khkw;
      const int a = rem / kw;
      const int b = rem - a * kw;
      base[o] = PATCH_OFF + poff + (c - c_lo) * PS + a * RS + b;
      kmask[o] = -1;
    } else if (i == rows && has_bias) {
      base[o] = ONE_OFF;
      kmask[o] = 0;
    } else {
      base[o] = ZERO_OFF;
      kmask[o] = 0;
    }
  }

  f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};

  const int cy = compact ? RS : sh * RS;     // LDS step per output row / col
  const int cx = compact ? 1 : sw;
  const int gy = compact ? sh : 1;           // source step per patch row / col
  const int gx = compact ? sw : 1;

  const int ch_begin = slice * d.cpi;
  const int ch_end = min(ch_begin + d.cpi, n_chunks);
  int cur_ns = -1, cur_ra = -1, cur_wa = -1;

  for (int ch = ch_begin; ch < ch_end; ++ch) {
    const int cg = ch % n_cg;
    const int t1 = ch / n_cg;
    const int rg = t1 % n_rg;
    const int sg = t1 / n_rg;
    const int s0 = sg * NS, ns = min(NS, N - s0);
    const int oh0 = rg * R, ra = min(R, Ho - oh0);
    const int ow0 = cg * Wc, wa = min(Wc, Wo - ow0);
    const int kc = ns * ra * wa;
    const int npairs = (kc + 1) >> 1;
    const int rows_in = compact ? ra : (ra - 1) * sh + kh;
    const int cols_in = compact ? wa : (wa - 1) * sw + kw;
    const int total_rows = ns * nch * rows_in;

    if (ns != cur_ns || ra != cur_ra || wa != cur_wa) {
      cur_ns = ns; cur_ra = ra; cur_wa = wa;
      const int rw = ra * wa;
      for (int k = tid; k < 2 * npairs; k += SYRK_THREADS) {
        int v = 0;
        if (k < kc) {
          const int s = k / rw;
          const int rem = k - s * rw;
          const int r = rem / wa;
          const int w = rem - r * wa;
          v = s * SS + r * cy + w * cx;
        }
        ktab[k] = v;
      }
      for (int p = tid; p < total_rows; p += SYRK_THREADS) {
        const int y = p % rows_in;
        const int t2 = p / rows_in;
        const int cc = t2 % nch;
        const int s = t2 / nch;
        rowtab[3 * p + 0] = s * SS + cc * PS + y * RS;
        rowtab[3 * p + 1] = (s * C + cc) * HW + y * gy * W;
        rowtab[3 * p + 2] = (cc << 16) | y;
      }
      __syncthreads();
    }

    // ---- stage the raw patch of both panels ----
    {
      int lx_shift = 0;
      while ((1 << lx_shift) < cols_in && lx_shift < 6) ++lx_shift;
      const int LX = 1 << lx_shift;
      const int row_step = SYRK_THREADS >> lx_shift;
      const int x0 = tid & (LX - 1);
      const int ih_base = oh0 * sh - ph;
      const int iw_base = ow0 * sw - pw;
      const int n_panels = diag ? 1 : 2;
      for (int pnl = 0; pnl < n_panels; ++pnl) {
        const int c_lo = pnl ? c_lo_j : c_lo_i;
        const int nch_p = min(nch, C - c_lo);
        const long long gbase = ((long long)s0 * C + c_lo) * HW + (long long)ih_base * W + iw_base;
        float* lbase = fs + PATCH_OFF + (pnl ? off_j : off_i);
        for (int p = tid >> lx_shift; p < total_rows; p += row_step) {
          const int lo = rowtab[3 * p + 0];
          const int go = rowtab[3 * p + 1];
          const int cyv = rowtab[3 * p + 2];
          const int cc = cyv >> 16;
          const int y = cyv & 0xffff;
          if (cc >= nch_p) continue;
          const int ih = ih_base + y * gy;
          const bool rowok = (unsigned)ih < (unsigned)H;
          const float* g = src + gbase + go;
          float* l = lbase + lo;
          for (int x = x0; x < cols_in; x += LX) {
            const int iw = iw_base + x * gx;
            float v = 0.0f;
            if (rowok && (unsigned)iw < (unsigned)W) v = g[x * gx];
            l[x] = v;
          }
        }
      }
    }
    __syncthreads();

    // ---- MFMA over this wave's share of the chunk's k pairs ----
    for (int p = wave; p < npairs; p += 4) {
      const int k = 2 * p + h;
      const int koff = ktab[k];
      const bool valid = k < kc;
      float a0 = fs[base[0] + (koff & kmask[0])];
      float a1 = fs[base[1] + (koff & kmask[1])];
      float b0 = fs[base[2] + (koff & kmask[2])];
      float b1 = fs[base[3] + (koff & kmask[3])];
      if (!valid) { a0 = 0.0f; a1 = 0.0f; b0 = 0.0f; b1 = 0.0f; }
      acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc00, 0, 0, 0);
      acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc01, 0, 0, 0);
      if (!diag) acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc10, 0, 0, 0);
      acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc11, 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- cross-wave reduction of the four K shares, then one coalesced slab write ----
  float* red = fs + PATCH_OFF + wave * (TM * TM);
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
    red[row * TM + r32] = acc00[reg];
    red[row * TM + 32 + r32] = acc01[reg];
    red[(32 + row) * TM + r32] = acc10[reg];
    red[(32 + row) * TM + 32 + r32] = acc11[reg];
  }
  __syncthreads();
  {
    const f32x4* r0 = reinterpret_cast<const f32x4*>(fs + PATCH_OFF);
    f32x4* slab = reinterpret_cast<f32x4*>(slabs + d.slab_base + (long long)local * (TM * TM));
    for (int e = tid; e < TM * TM / 4; e += SYRK_THREADS) {
      f32x4 v = r0[e] + r0[TM * TM / 4 + e] + r0[2 * (TM * TM / 4) + e] + r0[3 * (TM * TM / 4) + e];
      slab[e] = v;
    }
  }
}

// Sum the k-slices of one tile in slice order, scale, and add into the factor and its mirror.
__global__ void __launch_bounds__(SYRK_THREADS)
syrk_reduce_kernel(const FactorDev* __restrict__ descs, int n_factors, const float* __restrict__ slabs) {
  __shared__ float tile[TM][TM + 1];
  const int tid = threadIdx.x;
  const int f = find_segment(descs, n_factors, blockIdx.x, true);
  const FactorDev& d = descs[f];
  const int t = blockIdx.x - d.tile_base;
  int ti, tj;
  decode_tile(t, d.P, ti, tj);
  const bool diag = (ti == tj);
  const int i0 = ti * TM, j0 = tj * TM, dim = d.dim;
  const int n_tiles = d.n_tiles, n_slices = d.n_slices;
  const float scale = d.scale;
  const bool first = d.first != 0;
  float* __restrict__ dst = d.dst;

  const f32x4* s4 = reinterpret_cast<const f32x4*>(slabs + d.slab_base + (long long)t * (TM * TM));
  const long long slice_stride4 = (long long)n_tiles * (TM * TM / 4);
  for (int e = tid; e < TM * TM / 4; e += SYRK_THREADS) {
    f32x4 v = s4[e];
    for (int s = 1; s < n_slices; ++s) v += s4[s * slice_stride4 + e];
    const int r = e >> 4, c = (e & 15) << 2;
    tile[r][c + 0] = v.x * scale;
    tile[r][c + 1] = v.y * scale;
    tile[r][c + 2] = v.z * scale;
    tile[r][c + 3] = v.w * scale;
  }
  __syncthreads();
  for (int e = tid; e < TM * TM; e += SYRK_THREADS) {
    const int r = e >> 6, c = e & 63;
    const int gi = i0 + r, gj = j0 + c;
    if (gi < dim && gj < dim) {
      const float v = diag ? tile[min(r, c)][max(r, c)] : tile[r][c];
      const long long idx = (long long)gi * dim + gj;
      dst[idx] = first ? v : dst[idx] + v;
    }
  }
  if (!diag) {
    for (int e = tid; e < TM * TM; e += SYRK_THREADS) {
      const int r = e >> 6, c = e & 63;      // r indexes panel j, c panel i
      const int gi = j0 + r, gj = i0 + c;
      if (gi < dim && gj < dim) {
        const float v = tile[c][r];
        const long long idx = (long long)gi * dim + gj;
        dst[idx] = first ? v : dst[idx] + v;
      }
    }
  }
}

// The descriptor table travels as kernel arguments (copied by the runtime at launch time), so the
// call is fully asynchronous and needs neither pinned staging memory nor a stream synchronisation.
constexpr int UPLOAD_CHUNK = 16;
struct TableChunk { FactorDev f[UPLOAD_CHUNK]; };
static_assert(sizeof(TableChunk) <= 3840, "kernel argument block must stay below 4 KB");
static_assert(sizeof(FactorDev) % 8 == 0, "FactorDev must be 8-byte granular");

__global__ void __launch_bounds__(256)
upload_table_kernel(FactorDev* __restrict__ table, TableChunk chunk, int count) {
  const int words = count * (int)(sizeof(FactorDev) / 4);
  const int* in = reinterpret_cast<const int*>(&chunk);
  int* out = reinterpret_cast<int*>(table);
  for (int w = threadIdx.x; w < words; w += blockDim.x) out[w] = in[w];
}

// ---------------------------------------------------------------------------------------------
// Host-side planning
// ---------------------------------------------------------------------------------------------
static int round_mod32(int v, int m) {   // smallest u >= v with u = m (mod 32)
  m &= 31;
  int u = (v & ~31) + m;
  if (u < v) u += 32;
  return u;
}

struct ChunkGeom { int rows_in, cols_in, RS, PS, SS; };

static bool chunk_fits(const FactorDev& f, int NS, int R, int Wc, ChunkGeom& g) {
  g.rows_in = f.compact ? R : (R - 1) * f.sh + f.kh;
  g.cols_in = f.compact ? Wc : (Wc - 1) * f.sw + f.kw;
  g.RS = f.compact ? g.cols_in : round_mod32(g.cols_in, f.kw);
  g.PS = round_mod32(g.rows_in * g.RS, f.khkw);
  g.SS = f.nch * g.PS;
  if ((long long)NS * g.SS > PANEL_WORDS) return false;
  if ((long long)NS * f.nch * g.rows_in > ROWTAB_MAX) return false;
  if ((long long)NS * R * Wc > KTAB_MAX) return false;
  if (g.rows_in > 0xffff) return false;
  return true;
}

struct Plan {
  std::vector<FactorDev> f;
  int n_items = 0;
  int n_tiles = 0;
  long long slab_floats = 0;
};

static int make_plan(const curv_factor_desc* descs, int n, Plan& plan) {
  CURV_REQUIRE(n >= 0 && (n == 0 || descs != nullptr), "curv_kfac: bad descriptor array");
  plan.f.resize(n);
  std::vector<double> chunk_cost(n);   // MFMA wave-cycles of one (tile, chunk), averaged over tiles
  double total_cost = 0.0;
  for (int i = 0; i < n; ++i) {
    const curv_factor_desc& s = descs[i];
    FactorDev& f = plan.f[i];
    CURV_REQUIRE(s.N > 0 && s.C > 0 && s.H > 0 && s.W > 0, "curv_kfac: factor %d: empty source", i);
    CURV_REQUIRE(s.kh > 0 && s.kw > 0 && s.sh > 0 && s.sw > 0 && s.ph >= 0 && s.pw >= 0,
                 "curv_kfac: factor %d: bad kernel geometry", i);
    CURV_REQUIRE(s.src != nullptr && s.dst != nullptr, "curv_kfac: factor %d: null pointer", i);
    f.src = s.src; f.dst = s.dst;
    f.N = s.N; f.C = s.C; f.H = s.H; f.W = s.W;
    f.kh = s.kh; f.kw = s.kw; f.sh = s.sh; f.sw = s.sw; f.ph = s.ph; f.pw = s.pw;
    CURV_REQUIRE(s.H + 2 * s.ph >= s.kh && s.W + 2 * s.pw >= s.kw, "curv_kfac: factor %d: kernel larger than input", i);
    f.Ho = (s.H + 2 * s.ph - s.kh) / s.sh + 1;
    f.Wo = (s.W + 2 * s.pw - s.kw) / s.sw + 1;
    f.khkw = s.kh * s.kw;
    f.compact = (s.kh == 1 && s.kw == 1) ? 1 : 0;
    if (f.compact && s.sh == 1 && s.sw == 1 && s.ph == 0 && s.pw == 0) {
      // pure per-pixel factor (1x1 conv, grad_output, Linear): one long row per (sample, channel)
      CURV_REQUIRE((long long)s.H * s.W < (1LL << 30), "curv_kfac: factor %d: plane too large", i);
      f.W = s.H * s.W; f.H = 1; f.Ho = 1; f.Wo = f.W;
    }
    f.rows = s.C * f.khkw;
    f.has_bias = s.has_bias ? 1 : 0;
    f.dim = f.rows + f.has_bias;
    f.first = s.first;
    f.scale = s.scale;
    f.pad0 = 0;
    CURV_REQUIRE((long long)f.N * f.C * f.H * f.W < (1LL << 31), "curv_kfac: factor %d: source too large", i);
    f.nch = std::min(f.C, (f.khkw + TM - 2) / f.khkw + 1);

    // chunk extent: full-width rows if they fit, then as many rows, then as many samples
    ChunkGeom g;
    int Wc = f.Wo, R = 1, NS = 1;
    if (!chunk_fits(f, 1, 1, Wc, g)) {
      CURV_REQUIRE(chunk_fits(f, 1, 1, 1, g), "curv_kfac: factor %d: no chunk shape fits the LDS budget", i);
      int lo = 1, hi = Wc;                       // fits(lo), !fits(hi); fits is monotone in Wc
      while (hi - lo > 1) {
        const int mid = (lo + hi) / 2;
        if (chunk_fits(f, 1, 1, mid, g)) lo = mid; else hi = mid;
      }
      Wc = lo;
      if (Wc > 4) Wc &= ~3;                      // keep column groups 16-B aligned
    }
    if (Wc == f.Wo) {
      while (R < f.Ho && chunk_fits(f, 1, R + 1, Wc, g)) ++R;
      if (R == f.Ho) while (NS < f.N && chunk_fits(f, NS + 1, R, Wc, g)) ++NS;
    }
    chunk_fits(f, NS, R, Wc, g);
    f.NS = NS; f.R = R; f.Wc = Wc;
    f.RS = g.RS; f.PS = g.PS; f.SS = g.SS;
    f.n_rg = cdiv(f.Ho, R);
    f.n_cg = cdiv(f.Wo, Wc);
    f.n_chunks = cdiv(f.N, NS) * f.n_rg * f.n_cg;
    f.P = cdiv(f.dim, TM);
    f.n_tiles = f.P * (f.P + 1) / 2;
    const double kc = (double)NS * R * Wc;
    const double blocks = (4.0 * (f.n_tiles - f.P) + 3.0 * f.P) / f.n_tiles;
    chunk_cost[i] = kc / 8.0 * blocks * 64.0 + 600.0;   // + staging / barrier overhead
    total_cost += chunk_cost[i] * f.n_tiles * f.n_chunks;
  }
  // k-slicing: aim at ~16 items per workgroup slot (2 per CU) so that the tail of the launch is
  // a few percent, while keeping the slab traffic (16 KB per item) negligible.
  const double target = std::max(total_cost / (512.0 * 16.0), 1.0);
  long long items = 0, tiles = 0, slab = 0;
  for (int i = 0; i < n; ++i) {
    FactorDev& f = plan.f[i];
    int cpi = (int)(target / chunk_cost[i] + 0.5);
    cpi = std::max(1, std::min(cpi, f.n_chunks));
    f.cpi = cpi;
    f.n_slices = cdiv(f.n_chunks, cpi);
    f.cpi = cdiv(f.n_chunks, f.n_slices);          // even out the slices
    f.n_slices = cdiv(f.n_chunks, f.cpi);
    f.n_items = f.n_slices * f.n_tiles;
    f.item_base = (int)items;
    f.tile_base = (int)tiles;
    f.slab_base = slab;
    items += f.n_items;
    tiles += f.n_tiles;
    slab += (long long)f.n_items * TM * TM;
    CURV_REQUIRE(items < (1LL << 30) && tiles < (1LL << 30), "curv_kfac: too many work items");
  }
  plan.n_items = (int)items;
  plan.n_tiles = (int)tiles;
  plan.slab_floats = slab;
  return CURV_OK;
}

static size_t table_bytes(int n) { return align_up((size_t)std::max(n, 1) * sizeof(FactorDev), 256); }

}  // namespace curv

using namespace curv;

extern "C" size_t curv_kfac_workspace_bytes(const curv_factor_desc* descs, int n_factors) {
  Plan plan;
  // pointers are not dereferenced by the planner, but it insists on non-null ones
  std::vector<curv_factor_desc> tmp(descs, descs + (n_factors > 0 ? n_factors : 0));
  for (auto& t : tmp) {
    if (!t.src) t.src = reinterpret_cast<const float*>(16);
    if (!t.dst) t.dst = reinterpret_cast<float*>(16);
  }
  if (make_plan(tmp.data(), n_factors, plan) != CURV_OK) return 0;
  return table_bytes(n_factors) + (size_t)plan.slab_floats * sizeof(float);
}

extern "C" int curv_kfac_plan_info(const curv_factor_desc* descs, int n_factors, long long* out) {
  Plan plan;
  std::vector<curv_factor_desc> tmp(descs, descs + (n_factors > 0 ? n_factors : 0));
  for (auto& t : tmp) {
    if (!t.src) t.src = reinterpret_cast<const float*>(16);
    if (!t.dst) t.dst = reinterpret_cast<float*>(16);
  }
  int rc = make_plan(tmp.data(), n_factors, plan);
  if (rc != CURV_OK) return rc;
  for (int i = 0; i < n_factors; ++i) {
    const FactorDev& f = plan.f[i];
    long long* o = out + (size_t)i * CURV_PLAN_INFO_FIELDS;
    o[0] = f.dim; o[1] = f.Ho; o[2] = f.Wo; o[3] = f.NS; o[4] = f.R; o[5] = f.Wc;
    o[6] = f.n_chunks; o[7] = f.RS; o[8] = f.PS; o[9] = f.SS; o[10] = f.nch;
    o[11] = f.n_tiles; o[12] = f.cpi; o[13] = f.n_slices; o[14] = f.n_items; o[15] = f.item_base;
  }
  return CURV_OK;
}

extern "C" int curv_kfac_accumulate(void* stream_, const curv_factor_desc* descs, int n_factors,
                                    void* workspace, size_t workspace_bytes) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_factors == 0) return CURV_OK;
  Plan plan;
  int rc = make_plan(descs, n_factors, plan);
  if (rc != CURV_OK) return rc;
  const size_t tb = table_bytes(n_factors);
  const size_t need = tb + (size_t)plan.slab_floats * sizeof(float);
  if (workspace == nullptr || workspace_bytes < need) {
    set_error("curv_kfac_accumulate: workspace too small (%zu < %zu bytes)", workspace_bytes, need);
    return CURV_ERR_WORKSPACE;
  }
  FactorDev* table = reinterpret_cast<FactorDev*>(workspace);
  float* slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + tb);
  for (int b = 0; b < n_factors; b += UPLOAD_CHUNK) {
    TableChunk chunk;
    const int count = std::min(UPLOAD_CHUNK, n_factors - b);
    memset(&chunk, 0, sizeof(chunk));
    memcpy(chunk.f, plan.f.data() + b, (size_t)count * sizeof(FactorDev));
    hipLaunchKernelGGL(upload_table_kernel, dim3(1), dim3(256), 0, stream, table + b, chunk, count);
    CURV_LAUNCH_CHECK();
  }
  const int grid = cdiv(plan.n_items, 8 * XCD_GROUP) * 8 * XCD_GROUP;
  hipLaunchKernelGGL(syrk_patch_kernel, dim3(grid), dim3(SYRK_THREADS), 0, stream, table, n_factors,
                     plan.n_items, slabs);
  CURV_LAUNCH_CHECK();
  hipLaunchKernelGGL(syrk_reduce_kernel, dim3(plan.n_tiles), dim3(SYRK_THREADS), 0, stream, table,
                     n_factors, slabs);
  CURV_LAUNCH_CHECK();
  return CURV_OK;
}
