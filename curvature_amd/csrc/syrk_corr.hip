// KFAC A factor of a 3x3 / stride 1 / padding 1 convolution from SHIFTED CORRELATIONS on gfx950.
//
// The factor is the Gram matrix of the im2col rows (curvature/curvatures.py:329-337):
//   A[(c, kh, kw), (c', kh', kw')] = sum over samples and output pixels (y, x) of
//                                    X[c, y + kh - 1, x + kw - 1] * X[c', y + kh' - 1, x + kw' - 1]      (zero outside).
// With u = y + kh - 1, v = x + kw - 1 the summand is X[c, u, v] * X[c', u + dh, v + dw] for the RELATIVE shift
// (dh, dw) = (kh' - kh, kw' - kw): the 81 (45 by symmetry) C x C blocks of A share their products, and differ only in
// which border row / column of (u, v) the window of (kh, kw) leaves out.  For the blocks p = (kh, kw) >= q = (kh', kw')
// (row-major order; the rest is the transpose) there are 13 shifts, all with dh < 0 or (dh = 0, dw <= 0), and
//   block(p, q) = F[dh, dw] - [kh = 0] RB[dw] - [kh = 2] RT[dw] - [kw = 0] CR[dh] - [kw = 2] CL[dh] + corner terms
// with   F[dh, dw][c, c'] = sum_{n, u, v} X[c, u, v] X[c', u + dh, v + dw]        (whole image: 13 correlations)
//        RB / RT [dw]     = the same sum over the bottom / top row only (needed for dh = 0 only: 3 + 3)
//        CR / CL [dh]     = over the right / left column only           (needed for dw = 0 only: 3 + 3)
//        4 corner terms   = sum_n X[c, corner] X[c', corner]            (shift (0, 0) only)
// (tools / tests: the decomposition is checked against F.unfold in fp64).  29 C x C matrices with together
// 13 x C^2 x (N H W) multiply-adds instead of 40.5 x C^2 x (N H W) for the 45 blocks computed one by one: 3.1 x fewer
// flops, and every one of them is a plain "rows times shifted rows" product that the LDS-DMA kernel of syrk_flat.hip
// runs (operand panels at different offsets of the same rows), which is also the more efficient of the two kernels.
//
// Data flow per eligible layer (all inside curv_kfac_accumulate, in its workspace):
//   corr_prep_kernel      X (N, C, H, W) -> Xp (N, C, H, W + 2): two zero columns behind every image row, so that a flat
//                         offset dh * (W + 2) + dw never pairs pixels of different rows; and the gathered border rows /
//                         columns / corner pixels as (C, samples x length) arrays with the same zero separators
//   syrk_flat_kernel      29 "virtual factors" per layer in the ordinary work list of the LDS-DMA kernel
//   syrk_reduce_kernel    k-slices summed into the 29 component matrices
//   corr_assemble_kernel  dst (+)= scale * A, both triangles, through LDS tiles so that the 9-interleaved rows of A
//                         are written as contiguous runs
#include <algorithm>
#include <cstring>
#include <vector>

#include "syrk_plan.h"

namespace curv {

namespace corr {
constexpr int CT = 8;                        // channels per assembly tile edge: a (9 CT) x (9 CT) output tile (16: 63 KB of LDS, 238 -> 510 us)
constexpr int OUT = 9 * CT;
constexpr int LEAD = 4;                      // zero floats in front of every gathered row (negative shifts of sample 0)
constexpr int XP_LEAD = 4;                   // 64 channels: zero floats in front of the padded copy (offset -1 of its first row)

// component index of F[dh, dw]: dh = 0: dw = 0, -1, -2 -> 0..2; dh = -1: dw = -2..2 -> 3..7; dh = -2 -> 8..12
__host__ __device__ __forceinline__ int f_index(int dh, int dw) { return dh == 0 ? -dw : 3 + 5 * (-dh - 1) + (dw + 2); }
constexpr int RB0 = 13, RT0 = 16, CR0 = 19, CL0 = 22, PT0 = 25;      // + |dw| / |dh|; corners: BR, BL, TR, TL
}  // namespace corr

struct CorrDev {
  const float* src;
  float* dst;
  float* xp;
  float* rowb; float* rowt; float* colr; float* coll; float* pt;
  const float* comp;
  int N, C, H, W, Wp, Hq;
  int row_pitch, col_pitch, pt_pitch;
  int xp_pitch, xp_lead;     // floats per (sample, channel) row of Xp (>= H Wp: zero tail) and zero floats in front of Xp
  int comp_pitch;            // row pitch of the component matrices; component k starts at comp + comp_at[k]
  int comp_at[CORR_COMPONENTS];
  int first;
  float scale;
  long long prep_base;       // first workgroup of this layer in the prep grid
  int tile_base;             // first workgroup of this layer in the assembly grid
  unsigned g_magic, h_magic, c_magic;       // ceil(2^32 / d) for W / G, H, C
  int G, pad_;               // source floats per access of the padding pass (4 / 2 / 1)
};
constexpr int CORR_CHUNK = 13;
struct CorrChunk { CorrDev l[CORR_CHUNK]; };
static_assert(sizeof(CorrChunk) <= 3840, "kernel argument block must stay below 4 KB");

typedef __attribute__((address_space(1))) float gfl;

// one workgroup per PREP_SEG consecutive access groups of a layer's source (one exact division per workgroup, multiply-high
// arithmetic per group); the border rows / columns / corners are scattered from the same values
constexpr int PREP_SEG = 2048;
__device__ __forceinline__ int corr_divu(int x, int d, unsigned magic) { return d == 1 ? x : (int)__umulhi((unsigned)x, magic); }
static unsigned corr_magic(int d) { return (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); }

// G consecutive floats of a SOURCE row per thread and access (G = 4 / 2 / 1: the widest that divides W and the source
// alignment allows - 16-byte loads for the 56- and 28-wide images that make up three quarters of the bytes), U groups per
// thread and pass with all loads issued before the first store; the thread that holds a row's last group also writes the
// row's two zero columns and its share of the zero tail.  (Round 5: one 4-byte element of Xp per thread and access, 2.6 TB/s.)
template <int G>
__device__ __forceinline__ void corr_prep_body(const CorrDev& d, long long g_first, int n_groups) {
  constexpr int U = 4;             // (segments of 1024-8192 groups, 4 or 8 groups per pass: the same update() within 0.5 %)
  const int Wp = d.Wp, H = d.H, W = d.W, C = d.C;
  const int gpr = W / G;                                     // groups per source row
  const unsigned g_magic = d.g_magic, h_magic = d.h_magic, c_magic = d.c_magic;
  gfl* xp = (gfl*)d.xp;
  const int plane = H * Wp, tail = d.xp_pitch - plane;
  const gfl* __restrict__ src = (const gfl*)d.src;
  typedef float fg __attribute__((ext_vector_type(G == 1 ? 2 : G)));      // (G = 1: scalar accesses below)
  typedef __attribute__((address_space(1))) fg gfg;
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(1))) f2 gf2;
  const long long row_first = g_first / gpr;                 // one exact division per workgroup
  const int goff0 = (int)(g_first - row_first * gpr);
  // everything but the Xp store of one element (n, c, u, v) - a pixel or one of the two zero columns behind its row:
  // the gathered border rows / columns / corners and the zero leads, as the factor's strips read them
  auto emit = [&](int sc, int n, int c, int u, int v, float val) {
    if (tail > 0 && v >= W) {
      // packed pair tiles read every row up to two image rows past its end: zero tail, written by the 2 H elements
      // that are the row's padding columns
      for (int q = 2 * u + (v - W); q < tail; q += 2 * H) xp[(long long)sc * d.xp_pitch + plane + q] = 0.0f;
      if (sc == 0 && u == 0) for (int q = v - W; q < d.xp_lead; q += 2) xp[q - d.xp_lead] = 0.0f;
    }
    // The gathered arrays are padded to their pitch (a multiple of four floats + 4): the LDS-DMA kernel reads whole 4-pixel
    // groups, i.e. up to three floats behind a row's last element, and zeroes only ONE operand side there (flat_body) - the
    // other side must be finite, so whoever writes a row's last element also zeroes the gap behind it (round 6: these gaps
    // were left as the workspace held them, and a NaN bit pattern from an earlier fp64 use turned 0 * x into NaN)
    const bool last = n == d.N - 1;
    if (u == H - 1) {
      float* row = d.rowb + (long long)c * d.row_pitch;
      row[corr::LEAD + n * Wp + v] = val;
      if (last && v == W + 1) for (int q = corr::LEAD + d.N * Wp; q < d.row_pitch; ++q) row[q] = 0.0f;
    }
    if (u == 0) {
      float* row = d.rowt + (long long)c * d.row_pitch;
      row[corr::LEAD + n * Wp + v] = val;
      if (last && v == W + 1) for (int q = corr::LEAD + d.N * Wp; q < d.row_pitch; ++q) row[q] = 0.0f;
    }
    if (v == W - 1 || v == 0) {
      float* colbase = (v == 0 ? d.coll : d.colr) + (long long)c * d.col_pitch;
      float* col = colbase + corr::LEAD + n * d.Hq;
      col[u] = val;
      if (u == H - 1) {
        col[H] = 0.0f; col[H + 1] = 0.0f;
        if (last) for (int q = corr::LEAD + d.N * d.Hq; q < d.col_pitch; ++q) colbase[q] = 0.0f;
      }
    }
    if ((u == 0 || u == H - 1) && (v == 0 || v == W - 1)) {
      const int kk = (u == 0 ? 2 : 0) + (v == 0 ? 1 : 0);           // BR, BL, TR, TL
      float* prow = d.pt + ((long long)kk * C + c) * d.pt_pitch;
      prow[n] = val;
      if (last) for (int q = d.N; q < d.pt_pitch; ++q) prow[q] = 0.0f;
    }
    if (n == 0 && u == 0 && v < corr::LEAD) {
      d.rowb[(long long)c * d.row_pitch + v] = 0.0f;
      d.rowt[(long long)c * d.row_pitch + v] = 0.0f;
      d.colr[(long long)c * d.col_pitch + v] = 0.0f;
      d.coll[(long long)c * d.col_pitch + v] = 0.0f;
    }
  };
  for (int t0 = threadIdx.x; t0 < n_groups; t0 += 256 * U) {
    int r_[U], v_[U];
    float val_[U][G];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int off = goff0 + min(t0 + 256 * k, n_groups - 1);      // (passes beyond the segment repeat its last group)
      const int drow = gpr == 1 ? off : (int)__umulhi((unsigned)off, g_magic);
      v_[k] = (off - drow * gpr) * G;
      r_[k] = (int)row_first + drow;                                // (n * C + c) * H + u
      const gfl* sp = src + (long long)r_[k] * W + v_[k];
      if (G == 1) val_[k][0] = *sp;
      else {
        const fg x = *(const gfg*)sp;
#pragma unroll
        for (int e = 0; e < G; ++e) val_[k][e] = x[e];
      }
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      if (t0 + 256 * k >= n_groups) break;
      const int r = r_[k], v = v_[k];
      const int sc = corr_divu(r, H, h_magic), u = r - sc * H;
      const int n = corr_divu(sc, C, c_magic), c = sc - n * C;
      gfl* row = xp + (long long)sc * d.xp_pitch + u * Wp;
      // (Wp = W + 2 and v are even for G >= 2, the pitch a multiple of four: 8-byte stores)
      if (G == 1) row[v] = val_[k][0];
      else {
#pragma unroll
        for (int e = 0; e < G; e += 2) *(gf2*)(row + v + e) = f2{val_[k][e], val_[k][e + 1]};
      }
      const bool edge = v == 0 || v + G == W || u == 0 || u == H - 1;
      if (edge) {
#pragma unroll
        for (int e = 0; e < G; ++e) emit(sc, n, c, u, v + e, val_[k][e]);
      }
      if (v + G == W) {                                             // the row's last group: its two zero columns
        if (G == 1) { row[W] = 0.0f; row[W + 1] = 0.0f; }
        else *(gf2*)(row + W) = f2{0.0f, 0.0f};
        emit(sc, n, c, u, W, 0.0f);
        emit(sc, n, c, u, W + 1, 0.0f);
      }
    }
  }
}
__global__ void __launch_bounds__(256) corr_prep_kernel(CorrChunk chunk, int count, long long total) {
  (void)total;
  int l = 0;
  while (l + 1 < count && chunk.l[l + 1].prep_base <= (long long)blockIdx.x) ++l;
  const CorrDev& d = chunk.l[l];
  const int G = d.G;
  const long long groups = (long long)d.N * d.C * d.H * (d.W / G);
  const long long g0 = ((long long)blockIdx.x - d.prep_base) * PREP_SEG;
  const int cnt = (int)min((long long)PREP_SEG, groups - g0);
  if (G == 4) corr_prep_body<4>(d, g0, cnt);
  else if (G == 2) corr_prep_body<2>(d, g0, cnt);
  else corr_prep_body<1>(d, g0, cnt);
}


// dst tile (OUT x OUT) of channel tile (cb, cb2): rows (c, p), columns (c', q).  p >= q reads the components at
// [c][c'], p < q is the transposed block (q, p) and reads them at [c'][c] (the mirrored channel tile).
__global__ void __launch_bounds__(256) corr_assemble_kernel(CorrChunk chunk, int count) {
  using namespace corr;
  __shared__ float ta[29][CT][CT + 1];        // components at channel tile (cb, cb2)
  __shared__ float tb[29][CT][CT + 1];        // ... at (cb2, cb)
  int l = 0;
  while (l + 1 < count && chunk.l[l + 1].tile_base <= (int)blockIdx.x) ++l;
  const CorrDev& d = chunk.l[l];
  const int C = d.C, nct = C / CT;
  // Workgroups b, b + 8, b + 16, b + 24 share an XCD (and its L2); they take four NEIGHBOURING tiles of a row of channel
  // tiles, whose 8-float component rows are the four quarters of one 128-byte line (one tile per workgroup in launch
  // order fetched every such line four times, each time on another XCD)
  int t = blockIdx.x - d.tile_base;
  if (t < ((nct * nct) & ~31)) { const int w = t & 31; t = (t & ~31) + ((w & 7) << 2) + (w >> 3); }
  const int cb = t / nct, cb2 = t - cb * nct;
  {
    // component tiles: 8-float rows = two 16-byte loads (component offsets, pitches and tile columns are multiples of four
    // floats), all of a thread's loads issued before its first LDS store
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(1))) f4 gf4;
    constexpr int HALVES = 29 * CT * 2, PER = (HALVES + 255) / 256;      // 16-byte pieces per operand side
    f4 va[PER], vb[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = min((int)threadIdx.x + 256 * i, HALVES - 1);
      const int k = e / (CT * 2), rc = e - k * (CT * 2), r = rc >> 1, c = (rc & 1) * 4;
      const gfl* m = (const gfl*)d.comp + d.comp_at[k];
      va[i] = *(const gf4*)(m + (long long)(cb * CT + r) * d.comp_pitch + cb2 * CT + c);
      vb[i] = *(const gf4*)(m + (long long)(cb2 * CT + r) * d.comp_pitch + cb * CT + c);
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = (int)threadIdx.x + 256 * i;
      if (e < HALVES) {
        const int k = e / (CT * 2), rc = e - k * (CT * 2), r = rc >> 1, c = (rc & 1) * 4;
#pragma unroll
        for (int x = 0; x < 4; ++x) { ta[k][r][c + x] = va[i][x]; tb[k][r][c + x] = vb[i][x]; }
      }
    }
  }
  __syncthreads();
  const int dim = 9 * C;
  const float scale = d.scale;
  const bool first = d.first != 0;
  gfl* dst = (gfl*)d.dst;
  // four CONSECUTIVE columns per thread (one 16-byte access to the accumulated factor each way: tile rows are 288 bytes,
  // 16-byte aligned since 9 C and 72 are multiples of four), three such groups per thread and pass, the loads of the
  // accumulated factor issued first so that they land while the components are combined (round 5: one 4-byte element
  // per access, 20 dependent round trips per thread: 229-266 us for 0.8 GB)
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(1))) f4 gf4;
  constexpr int GROUPS = OUT * OUT / 4, GPR = OUT / 4;      // 16-byte groups per tile / per tile row
  static_assert(OUT % 4 == 0, "tile rows in whole 16-byte groups");
  auto element = [&](int R, int Q) -> float {
    int c = R / 9, p = R - 9 * c, c2 = Q / 9, q = Q - 9 * c2;
    const bool lower = p >= q;
    const float (*T)[CT][CT + 1] = lower ? ta : tb;
    if (!lower) { int s_ = p; p = q; q = s_; s_ = c; c = c2; c2 = s_; }      // block (q, p) transposed
    const int kh = p / 3, kw = p - 3 * kh, kh2 = q / 3, kw2 = q - 3 * kh2;
    const int dh = kh2 - kh, dw = kw2 - kw;
    float v = T[f_index(dh, dw)][c][c2];
    if (dh == 0) {
      if (kh == 0) v -= T[RB0 - dw][c][c2];
      if (kh == 2) v -= T[RT0 - dw][c][c2];
    }
    if (dw == 0) {
      if (kw == 0) v -= T[CR0 - dh][c][c2];
      if (kw == 2) v -= T[CL0 - dh][c][c2];
    }
    if (dh == 0 && dw == 0 && kh != 1 && kw != 1) v += T[PT0 + (kh == 2 ? 2 : 0) + (kw == 2 ? 1 : 0)][c][c2];
    return v * scale;
  };
  constexpr int U = 3;
  for (int g0 = threadIdx.x; g0 < GROUPS; g0 += 256 * U) {
    long long o_[U];
    f4 old_[U], v_[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int g = min(g0 + 256 * u, GROUPS - 1);
      const int R = g / GPR, Q = 4 * (g - R * GPR);
      o_[u] = (long long)(cb * OUT + R) * dim + cb2 * OUT + Q;
      old_[u] = first ? f4{0.0f, 0.0f, 0.0f, 0.0f} : *(const gf4*)(dst + o_[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int g = min(g0 + 256 * u, GROUPS - 1);
      const int R = g / GPR, Q = 4 * (g - R * GPR);
#pragma unroll
      for (int x = 0; x < 4; ++x) v_[u][x] = element(R, Q + x);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (g0 + 256 * u < GROUPS) *(gf4*)(dst + o_[u]) = first ? v_[u] : old_[u] + v_[u];
  }
}

// ------------------------------------------------------------------------------------------------ host side
static long long round_up4(long long v) { return (v + 3) & ~3LL; }

bool syrk_corr_eligible(const curv_factor_desc& s) {
  if (!(s.kh == 3 && s.kw == 3 && s.sh == 1 && s.sw == 1 && s.ph == 1 && s.pw == 1) || s.has_bias) return false;
  // whole 128-row tiles of channels, or exactly 64 channels (packed pair tiles)
  if (!(s.C == 64 || (s.C >= 128 && s.C % 128 == 0)) || s.N < 8 || s.H < 3 || s.W < 3) return false;
  // every operand array addressable by one buffer descriptor of the LDS-DMA kernel
  if ((long long)s.N * s.C * ((long long)s.H * (s.W + 2) + 2 * (s.W + 2) + 16) * 4 >= (1LL << 32) - 4096) return false;
  if ((long long)s.N * s.C * s.H * std::max(s.H, s.C) >= (1LL << 32)) return false;     // row numbers are divided by multiply-high
  return true;
}

// 64 channels: a 128x128 tile of the LDS-DMA kernel holds FOUR shifted correlations.  Panel j is the image twice, at
// offsets (0, -1); panel i the image at two offsets (a1, a2): block (hi, hj) = sum_p X[c][p + a_hi] X[c'][p + b_hj] is
// the correlation with relative offset a_hi - b_hj, i.e. a row offset a covers the offsets {a, a + 1}.  The 13 offsets
// 0..2, Wp-2..Wp+2, 2Wp-2..2Wp+2 take 8 row offsets = 4 tiles (16 blocks, 3 of them duplicates) instead of the 45
// blocks of the factor computed one by one; the border strips one tile per strip array, the corner products two
// symmetric 128-row tiles over [corner k; corner k + 1].  Every row of Xp ends in a zero tail and the array starts
// behind XP_LEAD zeros, so all four blocks of a tile sum over the same K = H Wp positions.
static void syrk_corr_expand_half(const curv_factor_desc& s, std::vector<FactorDev>& f, CorrLayer& L, long long& area_floats) {
  using namespace corr;
  auto take = [&](long long n) { const long long o = area_floats; area_floats += round_up4(n) + 64; return o; };
  const int Wp = L.Wp, plane = s.H * Wp;
  L.xp_pitch = (int)round_up4((long long)plane + 2 * Wp + 12);
  L.comp_pitch = 128;
  L.n_vf = 10;
  L.comp_off = take(10LL * 128 * 128);
  for (int k = 0; k < CORR_COMPONENTS; ++k) L.comp_at[k] = -1;
  int tile = 0;
  auto pair_tile = [&](long long src_off, int samples, int pitch, int K, int lead, int a1, int a2, bool pad_k = false) {
    FactorDev v;
    memset(&v, 0, sizeof(v));
    // (the rows the whole-image tiles read end in a zero tail: K rounded up to whole 4-pixel groups adds zeros to the sums
    // and spares the LDS-DMA kernel its row-end masks; Wo keeps the true K for the flop count)
    v.N = samples; v.C = 64; v.H = 1; v.W = pad_k ? (int)round_up4(K) : K;
    v.kh = v.kw = v.sh = v.sw = 1;
    v.Ho = 1; v.Wo = K; v.khkw = 1;
    v.rows = v.dim = 128;
    v.compact = 1;
    v.first = 1; v.scale = 1.0f;
    v.dma = 1;
    v.pitch = pitch; v.nonsym = 1; v.half = 1;
    v.off_i = lead + a1; v.off_i2 = lead + a2; v.off_j = lead; v.off_j2 = lead - 1;
    v.TM = 128; v.P = 1; v.n_tiles = 1;
    v.n_chunks = syrk_flat_chunks(samples, v.W);
    v.src = reinterpret_cast<const float*>(src_off);
    v.dst = reinterpret_cast<float*>(L.comp_off + (long long)tile * 128 * 128);
    f.push_back(v);
    return tile++;
  };
  auto block_at = [&](int t, int hi, int hj) { return t * 128 * 128 + hi * 64 * 128 + hj * 64; };
  // whole-image correlations
  const int rows8[4][2] = {{0, 1}, {Wp - 2, Wp}, {Wp + 1, 2 * Wp - 2}, {2 * Wp, 2 * Wp + 1}};
  for (int t4 = 0; t4 < 4; ++t4) {
    const int t = pair_tile(L.xp_off, s.N, L.xp_pitch, plane, XP_LEAD, rows8[t4][0], rows8[t4][1], true);
    f.back().group_n = 4; f.back().group_pos = t4;
    for (int hi = 0; hi < 2; ++hi)
      for (int hj = 0; hj < 2; ++hj) {
        const int delta = rows8[t4][hi] + hj;                  // = -(dh Wp + dw)
        const int m = (delta + Wp / 2) / Wp, dh = -m, dw = -(delta - m * Wp);
        if (dw < -2 || dw > 2 || m > 2 || (m == 0 && dw > 0)) continue;
        L.comp_at[f_index(dh, dw)] = block_at(t, hi, hj);
      }
  }
  // border strips: offsets 0, 1, 2 from the row offsets (0, 1)
  const long long strip_src[4] = {L.rowb_off, L.rowt_off, L.colr_off, L.coll_off};
  const int strip_comp[4] = {RB0, RT0, CR0, CL0};
  for (int k = 0; k < 4; ++k) {
    const bool rows = k < 2;
    const int t = pair_tile(strip_src[k], 1, rows ? L.row_pitch : L.col_pitch, rows ? s.N * Wp : s.N * L.Hq, LEAD, 0, 1);
    L.comp_at[strip_comp[k] + 0] = block_at(t, 0, 0);
    L.comp_at[strip_comp[k] + 1] = block_at(t, 0, 1);
    L.comp_at[strip_comp[k] + 2] = block_at(t, 1, 1);
  }
  // corner products: [corner k; corner k + 1] as one symmetric 128-row factor, the products on its diagonal blocks
  for (int k = 0; k < 4; k += 2) {
    FactorDev v;
    memset(&v, 0, sizeof(v));
    v.N = 1; v.C = 128; v.H = 1; v.W = s.N;
    v.kh = v.kw = v.sh = v.sw = 1;
    v.Ho = 1; v.Wo = s.N; v.khkw = 1;
    v.rows = v.dim = 128;
    v.compact = 1;
    v.first = 1; v.scale = 1.0f;
    v.dma = 1;
    v.pitch = L.pt_pitch;
    v.TM = 128; v.P = 1; v.n_tiles = 1;
    v.n_chunks = syrk_flat_chunks(1, s.N);
    v.src = reinterpret_cast<const float*>(L.pt_off + (long long)k * 64 * L.pt_pitch);
    v.dst = reinterpret_cast<float*>(L.comp_off + (long long)tile * 128 * 128);
    f.push_back(v);
    L.comp_at[PT0 + k] = block_at(tile, 0, 0);
    L.comp_at[PT0 + k + 1] = block_at(tile, 1, 1);
    ++tile;
  }
}

// Append the 29 virtual factors of one eligible user factor to `f` (pointers are filled in by syrk_corr_bind) and
// reserve its part of the correlation area (`area_floats` advances).
void syrk_corr_expand(const curv_factor_desc& s, int user, std::vector<FactorDev>& f, CorrLayer& L,
                      long long& area_floats) {
  using namespace corr;
  memset(&L, 0, sizeof(L));
  L.user = user;
  L.vf0 = (int)f.size();
  L.N = s.N; L.C = s.C; L.H = s.H; L.W = s.W; L.Wp = s.W + 2; L.Hq = s.H + 2;
  L.row_pitch = (int)round_up4(LEAD + (long long)s.N * L.Wp) + 4;
  L.col_pitch = (int)round_up4(LEAD + (long long)s.N * L.Hq) + 4;
  L.pt_pitch = (int)round_up4(s.N) + 4;
  auto take = [&](long long n) { const long long o = area_floats; area_floats += round_up4(n) + 64; return o; };
  // 64 channels: rows with a zero tail of two image rows (+ DMA slack) and XP_LEAD zeros in front of the array
  // (128 channels and more: a zero tail of one group behind every row, so that a correlation's K = plane - delta rounds up to
  // whole 4-pixel groups: the shifted panel reads zeros there and the LDS-DMA kernel needs no row-end masks)
  const int xp_pitch = s.C == 64 ? (int)round_up4((long long)s.H * L.Wp + 2 * L.Wp + 12) : (int)round_up4((long long)s.H * L.Wp + 4);
  L.xp_off = take((long long)s.N * s.C * xp_pitch + (s.C == 64 ? XP_LEAD : 0));
  L.rowb_off = take((long long)s.C * L.row_pitch);
  L.rowt_off = take((long long)s.C * L.row_pitch);
  L.colr_off = take((long long)s.C * L.col_pitch);
  L.coll_off = take((long long)s.C * L.col_pitch);
  L.pt_off = take(4LL * s.C * L.pt_pitch);
  if (s.C == 64) { syrk_corr_expand_half(s, f, L, area_floats); return; }
  L.n_vf = CORR_COMPONENTS;
  L.xp_pitch = xp_pitch;
  L.comp_pitch = s.C;
  for (int k = 0; k < CORR_COMPONENTS; ++k) L.comp_at[k] = k * s.C * s.C;
  L.comp_off = take((long long)CORR_COMPONENTS * s.C * s.C);

  auto add = [&](int comp, long long src_off, int samples, int pitch, int K, int off_i, int off_j, bool nonsym, bool pad_k = false) {
    FactorDev v;
    memset(&v, 0, sizeof(v));
    v.N = samples; v.C = s.C; v.H = 1; v.W = pad_k ? (int)round_up4(K) : K;
    v.kh = v.kw = v.sh = v.sw = 1;
    v.Ho = 1; v.Wo = K; v.khkw = 1;
    v.rows = v.dim = s.C;
    v.compact = 1;
    v.first = 1; v.scale = 1.0f;
    v.dma = 1;
    v.pitch = pitch; v.off_i = off_i; v.off_j = off_j; v.nonsym = nonsym ? 1 : 0;
    v.TM = 128;
    v.P = s.C / 128;
    v.n_tiles = nonsym ? v.P * v.P : v.P * (v.P + 1) / 2;
    v.n_chunks = syrk_flat_chunks(samples, v.W);
    // src / dst: offsets into the correlation area for now (syrk_corr_bind turns them into pointers)
    v.src = reinterpret_cast<const float*>(src_off);
    v.dst = reinterpret_cast<float*>(L.comp_off + (long long)comp * s.C * s.C);
    f.push_back(v);
  };
  const int plane = s.H * L.Wp;
  // the 13 whole-image correlations, in component order
  for (int k = 0; k < 13; ++k) {
    int dh, dw;
    if (k < 3) { dh = 0; dw = -k; } else { dh = -1 - (k - 3) / 5; dw = (k - 3) % 5 - 2; }
    const int delta = -(dh * L.Wp + dw);
    add(f_index(dh, dw), L.xp_off, s.N, L.xp_pitch, plane - delta, delta, 0, delta != 0, true);
    if (k >= 1) { f.back().group_n = 12; f.back().group_pos = k - 1; }      // the 12 non-symmetric ones: one item range
  }
  for (int a = 0; a < 3; ++a) add(RB0 + a, L.rowb_off, 1, L.row_pitch, s.N * L.Wp, LEAD, LEAD - a, a != 0);
  for (int a = 0; a < 3; ++a) add(RT0 + a, L.rowt_off, 1, L.row_pitch, s.N * L.Wp, LEAD, LEAD - a, a != 0);
  for (int a = 0; a < 3; ++a) add(CR0 + a, L.colr_off, 1, L.col_pitch, s.N * L.Hq, LEAD, LEAD - a, a != 0);
  for (int a = 0; a < 3; ++a) add(CL0 + a, L.coll_off, 1, L.col_pitch, s.N * L.Hq, LEAD, LEAD - a, a != 0);
  for (int k = 0; k < 4; ++k) add(PT0 + k, L.pt_off + (long long)k * s.C * L.pt_pitch, 1, L.pt_pitch, s.N, 0, 0, false);
}

// area-relative offsets of a layer's virtual factors -> device pointers
void syrk_corr_bind(const CorrLayer& L, std::vector<FactorDev>& f, float* area) {
  for (int k = 0; k < L.n_vf; ++k) {
    FactorDev& v = f[L.vf0 + k];
    v.src = area + reinterpret_cast<intptr_t>(v.src);
    v.dst = area + reinterpret_cast<intptr_t>(v.dst);
  }
}

static void fill_dev(const CorrLayer& L, const FactorDev& user, float* area, CorrDev& d) {
  memset(&d, 0, sizeof(d));
  d.src = user.src; d.dst = user.dst;
  d.xp_lead = L.C == 64 ? corr::XP_LEAD : 0;
  d.xp = area + L.xp_off + d.xp_lead;
  d.xp_pitch = L.xp_pitch;
  d.comp_pitch = L.comp_pitch;
  for (int k = 0; k < CORR_COMPONENTS; ++k) d.comp_at[k] = L.comp_at[k];
  d.rowb = area + L.rowb_off; d.rowt = area + L.rowt_off;
  d.colr = area + L.colr_off; d.coll = area + L.coll_off;
  d.pt = area + L.pt_off;
  d.comp = area + L.comp_off;
  d.N = L.N; d.C = L.C; d.H = L.H; d.W = L.W; d.Wp = L.Wp; d.Hq = L.Hq;
  d.row_pitch = L.row_pitch; d.col_pitch = L.col_pitch; d.pt_pitch = L.pt_pitch;
  d.first = user.first; d.scale = user.scale;
  d.G = ((L.W & 3) == 0 && (reinterpret_cast<uintptr_t>(user.src) & 15) == 0) ? 4
        : ((L.W & 1) == 0 && (reinterpret_cast<uintptr_t>(user.src) & 7) == 0) ? 2 : 1;
  d.g_magic = corr_magic(L.W / d.G); d.h_magic = corr_magic(L.H); d.c_magic = corr_magic(L.C);
}

int launch_corr_prep(hipStream_t stream, const std::vector<CorrLayer>& layers, const std::vector<FactorDev>& f,
                     float* area) {
  for (size_t b = 0; b < layers.size(); b += CORR_CHUNK) {
    CorrChunk chunk;
    memset(&chunk, 0, sizeof(chunk));
    const int count = (int)std::min<size_t>(CORR_CHUNK, layers.size() - b);
    long long total = 0;
    for (int k = 0; k < count; ++k) {
      const CorrLayer& L = layers[b + k];
      fill_dev(L, f[L.user], area, chunk.l[k]);
      chunk.l[k].prep_base = total;
      total += ((long long)L.N * L.C * L.H * (L.W / chunk.l[k].G) + PREP_SEG - 1) / PREP_SEG;
    }
    CURV_REQUIRE(total < (1LL << 31), "curv_kfac: too many padding segments");
    hipLaunchKernelGGL(corr_prep_kernel, dim3((unsigned)total), dim3(256), 0, stream, chunk, count, total);
    CURV_LAUNCH_CHECK();
  }
  return CURV_OK;
}

int launch_corr_assemble(hipStream_t stream, const std::vector<CorrLayer>& layers, const std::vector<FactorDev>& f,
                         float* area) {
  for (size_t b = 0; b < layers.size(); b += CORR_CHUNK) {
    CorrChunk chunk;
    memset(&chunk, 0, sizeof(chunk));
    const int count = (int)std::min<size_t>(CORR_CHUNK, layers.size() - b);
    int tiles = 0;
    for (int k = 0; k < count; ++k) {
      const CorrLayer& L = layers[b + k];
      fill_dev(L, f[L.user], area, chunk.l[k]);
      chunk.l[k].tile_base = tiles;
      tiles += (L.C / corr::CT) * (L.C / corr::CT);
    }
    hipLaunchKernelGGL(corr_assemble_kernel, dim3(tiles), dim3(256), 0, stream, chunk, count);
    CURV_LAUNCH_CHECK();
  }
  return CURV_OK;
}

}  // namespace curv
