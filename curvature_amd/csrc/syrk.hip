// KFAC factor build on gfx950: grouped, symmetric, implicit-im2col SYRK on the fp32 MFMA.
//
//   dst (+)= scale * X X^T,   X = unfold(src) (+ ones row)       (curvature/curvatures.py:329-350)
//
// Design (see DESIGN.md section "K1"):
//   * one launch covers every Kronecker factor of a model.  The work list is implicit:
//       item -> (factor, k-slice, upper-triangular tile), decoded on the device from a small
//     descriptor table, so nothing but that table is uploaded per call;
//   * a workgroup stages the RAW activation patch of the channels its two row panels touch in LDS
//     (never the unfolded matrix: a 3x3 conv reads each input pixel once per chunk instead of nine
//     times) and MFMA operands are gathered from the patch with per-lane addresses
//       addr(i, k) = lane_base(i) + koff(k),
//     lane_base encodes (channel, kh, kw), koff (a scalar) encodes (sample, out row, out col) of the chunk;
//     along an output row koff advances by a constant, so one address serves several MFMA steps through the
//     immediate offsets of the LDS reads;
//     row/plane strides are padded so that 32 consecutive unfolded rows hit 32 distinct banks;
//   * the bias row of ones and the zero padding rows are two constant LDS words;
//   * staging, convolutions with kh x kw > 1 whose chunks span the full output width (every 3x3 / 5x5 / 7x7 layer
//     of the benchmark networks): a pass in front of the kernel (syrk_pre.hip) writes the source once more in
//     PATCH-IMAGE order - per (chunk, sample, channel) one zero-padded plane with the LDS strides - so that the
//     image of a panel is one contiguous run and the kernel fills it with buffer_load_dwordx4 ... lds (1 KiB per
//     wave-instruction, no staging registers, no address arithmetic, no LDS store pass); the staging phase of a
//     chunk shrinks from ~11 k to ~2 k wave-cycles, which the other workgroup on the CU covers with its MFMAs;
//   * staging, everything else (flattened per-pixel factors that are not whole 128-row tiles, strided 1x1
//     convolutions, chunks narrower than the output): raw buffer loads of chunk t+1 are issued into registers
//     before the MFMA loop of chunk t and written to LDS after it; padding lanes carry an out-of-range
//     offset and get their zeros from the hardware range check;
//   * tile 64x64 (four waves split the K range of a chunk, each owning the full tile) for small or
//     awkward dims, tile 128x128 (2x2 waves of 64x64) for large ones; every wave works on 2x2
//     v_mfma_f32_32x32x2_f32 blocks and skips the redundant lower-left block on the diagonal;
//   * partial tiles go to fp32 slabs and a second kernel sums the k-slices in a fixed order, applies
//     the scale and adds into the factor and its mirror image: deterministic, exactly symmetric.
#include "common.h"
#include "syrk_plan.h"

#include <algorithm>
#include <type_traits>
#include <cmath>
#include <cstdlib>
#include <vector>

#ifndef CURV_SINGLE_MAX
#define CURV_SINGLE_MAX 5.0            // a 128x128 tile whose K range costs at most this many target item lengths stays unsliced
#endif
#ifndef CURV_ITEM_FLOOR
#define CURV_ITEM_FLOOR 40000          // shortest work item (MFMA CU-cycles) a small launch is cut into
#endif
#ifndef CURV_MAX_CHAIN_PX
#define CURV_MAX_CHAIN_PX 3072         // longest fp32 accumulation chain, in k values (pixels x samples)
#endif
#ifndef CURV_PRE_PANEL_WORDS
#define CURV_PRE_PANEL_WORDS 6528
#define CURV_PRE_WGS 3
#endif

namespace curv {

constexpr double MAX_CHAIN_PX = CURV_MAX_CHAIN_PX;
constexpr int GU = 2;                  // k steps (MFMA groups) per operand fetch: one address per operand row serves GU steps
constexpr int PANEL_WORDS = 8704;      // LDS words per panel patch
constexpr int PATCH_WORDS = 2 * PANEL_WORDS;   // >= 4 x (64x64) cross-wave reduce scratch
constexpr int STAGE_SLOTS = 32;        // staging registers per panel per lane (floats)
constexpr int PANEL_SLOT_ELEMS = STAGE_SLOTS * SYRK_THREADS;   // padded patch elements per panel
// LDS word offsets
constexpr int ZERO_OFF = 0;            // 16 zero words (padding rows read their GU step elements from here)
constexpr int ONE_OFF = 16;            // 16 one words (the bias row of ones)
constexpr int PATCH_OFF = 32;
constexpr int SMEM_WORDS = PATCH_OFF + PATCH_WORDS;   // 18464 words = 73856 B -> 2 workgroups per CU
static_assert(PATCH_WORDS >= 4 * 64 * 64, "reduce scratch must fit the patch region");
static_assert(2 * SMEM_WORDS * 4 <= 160 * 1024, "two workgroups per CU");
// the pre-tiled variant (LDS-DMA staging, no staging registers): smaller panels, THREE workgroups per CU - while one
// waits for its DMA pieces or sits at a barrier, two others keep the matrix pipe of every SIMD busy
constexpr int PRE_PANEL_WORDS = CURV_PRE_PANEL_WORDS;
constexpr int PRE_WGS = CURV_PRE_WGS;
constexpr int PRE_SMEM_WORDS = PATCH_OFF + 2 * PRE_PANEL_WORDS;
static_assert(2 * PRE_PANEL_WORDS >= 2 * 64 * 64, "two-round reduce scratch must fit the patch region");
static_assert(PRE_WGS * PRE_SMEM_WORDS * 4 <= 160 * 1024, "workgroups per CU of the pre-tiled variant");

typedef __attribute__((address_space(1))) float gfloat;      // global-address-space views
typedef __attribute__((address_space(1))) f32x4 gf32x4;
typedef __attribute__((address_space(1))) char gchar;
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Position and extent of one K chunk.
struct Chunk {
  int s0, ns, oh0, ra, ow0, wa, rows_in, cols_in, ih_base, iw_base;
  int plane0;                          // pre-tiled source: first (sample) plane group of the chunk
};

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) char lds_char;

template <int TMv, bool PRE>
__device__ __forceinline__ void syrk_body(const FactorDev& d, const int local, float* __restrict__ slabs,
                                          const float* __restrict__ zeros_, int* smem, lds_char* l3) {
  (void)zeros_;            // (zeroed pad of the workspace: unused since padding comes from the buffer range check)
  float* fs = reinterpret_cast<float*>(smem);
  constexpr int PW = PRE ? PRE_PANEL_WORDS : PANEL_WORDS;          // LDS words per panel image
  constexpr int SMW = PATCH_OFF + 2 * PW;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31;
  const int h = lane >> 5;

  const int n_tiles = d.n_tiles;
  const int slice = local / n_tiles;
  const int tile = local - slice * n_tiles;
  int ti, tj;
  decode_tile(tile, d.P, ti, tj);
  const bool diag = (ti == tj);
  const int i0 = ti * TMv, j0 = tj * TMv;

  // wave roles
  // 128x128 tiles: wave (wm, wn) owns the 64x64 quadrant (wm, wn), i.e. 2x2 MFMA blocks.  On a diagonal
  // tile the quadrant (1, 0) is redundant and the diagonal quadrants skip their lower-left block, so the
  // work is 3 + 4 + 0 + 3 blocks: the otherwise idle wave takes the right block column of quadrant (0, 1)
  // (part 3) and the wave of that quadrant keeps the left one (part 2): 3 + 2 + 2 + 3.
  //   part 0: all four blocks   1: all but the lower-left   2: left block column   3: right block column
  int wm_ = (TMv == 128) ? (wave >> 1) : 0;
  int wn_ = (TMv == 128) ? (wave & 1) : 0;
  int part_ = 0;
  if (diag) {
    if (wm_ == wn_) part_ = 1;
    else if (TMv == 128) { part_ = (wm_ == 0) ? 2 : 3; wm_ = 0; wn_ = 1; }
  }
  const int wm = wm_, wn = wn_, part = part_;
  const int kfirst = (TMv == 128) ? 0 : wave;      // first k pair of this wave within a chunk
  constexpr int KSTRIDE = (TMv == 128) ? 1 : 4;
  constexpr bool idle = false;

  const int N = d.N, C = d.C, H = d.H, W = d.W;
  const int kh = d.kh, kw = d.kw, sh = d.sh, sw = d.sw, ph = d.ph, pw = d.pw;
  const int Ho = d.Ho, Wo = d.Wo, khkw = d.khkw, rows = d.rows, has_bias = d.has_bias;
  const int compact = d.compact, vec4 = d.vec4;
  const bool flat1 = d.flat && !vec4;                // flattened per-pixel factor staged with scalar loads
  const int NS = d.NS, R = d.R, Wc = d.Wc, n_rg = d.n_rg, n_cg = d.n_cg, n_chunks = d.n_chunks;
  const int RS = d.RS, PS = d.PS, SS = d.SS, nch = d.nch, cshift = d.cshift;
  const int HW = H * W;

  const int c_lo_i = i0 / khkw, c_lo_j = j0 / khkw;
  const int off_j = diag ? 0 : PW;
  const int n_panels = diag ? 1 : 2;

  const int cy = compact ? RS : sh * RS;     // LDS step per output row / col
  const int cx = compact ? 1 : sw;
  const int gy = compact ? sh : 1;           // source step per patch row / col
  const int gx = compact ? sw : 1;

  // full-chunk patch extent (tables are built for it once; ragged chunks mask the excess)
  const int rows_in_full = compact ? R : (R - 1) * sh + kh;

  // every LDS word a masked run element may touch must be finite (0 * NaN would poison the tile)
  for (int w = tid; w < SMW; w += SYRK_THREADS) fs[w] = (w >= ONE_OFF && w < ONE_OFF + 16) ? 1.0f : 0.0f;

  // Per-lane operand rows: A0/A1 = panel i rows (64 wm) + r32, + 32; B0/B1 = panel j rows (64 wn) + ...
  int base[4], kmask[4];
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    const int i = ((o < 2) ? i0 + 64 * wm : j0 + 64 * wn) + (o & 1) * 32 + r32;
    const int c_lo = (o < 2) ? c_lo_i : c_lo_j;
    const int poff = (o < 2) ? 0 : off_j;
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

  // staging lane geometry (constant per lane: 256 % padded row length == 0)
  const int cmask = (1 << cshift) - 1;
  const int lx_ = tid & cmask;                // x (general) or x4 (vec4) of this lane
  const int prow0_ = tid >> cshift;
  const int prow_step = SYRK_THREADS >> cshift;

  auto decode_chunk = [&](int ch) {
    Chunk c;
    const int cg = ch % n_cg;
    const int t1 = ch / n_cg;
    const int rg = t1 % n_rg;
    const int sg = t1 / n_rg;
    c.s0 = sg * NS; c.ns = min(NS, N - c.s0);
    c.oh0 = rg * R; c.ra = min(R, Ho - c.oh0);
    c.ow0 = cg * Wc; c.wa = min(Wc, Wo - c.ow0);
    c.rows_in = compact ? c.ra : (c.ra - 1) * sh + kh;
    c.cols_in = compact ? c.wa : (c.wa - 1) * sw + kw;
    c.ih_base = c.oh0 * sh - ph;
    c.iw_base = c.ow0 * sw - pw;
    c.plane0 = ch * NS;
    return c;
  };
  // pre-tiled variant: chunks are full-width (n_cg == 1) and an item walks them in order, so the (sample group, row
  // group) pair is carried along instead of being divided out of the chunk number every time
  auto next_chunk_pre = [&](const Chunk& p, int ch) {
    Chunk c = p;
    c.oh0 = p.oh0 + R;
    if (c.oh0 >= Ho) { c.oh0 = 0; c.s0 = p.s0 + NS; c.ns = min(NS, N - c.s0); }
    c.ra = min(R, Ho - c.oh0);
    c.rows_in = (c.ra - 1) * sh + kh;
    c.ih_base = c.oh0 * sh - ph;
    c.plane0 = ch * NS;
    return c;
  };

  // ---- staging: global -> registers (before the MFMA loop) -> LDS (after it) ----
  // All source reads are raw buffer loads against a descriptor of the chunk's samples:
  //   address = base + soffset (SGPR: channel plane, advanced by SALU per slot) + voffset (VGPR: the
  //   lane's position inside a plane, constant across the channels it walks),
  // and a lane that holds padding, or lies outside a ragged chunk, carries voffset = OOB: the range
  // check returns 0 for it, so halo zeros cost no instruction and no read ever leaves the tensor.
  // A slot is therefore ~2 instructions on either side (s_add + buffer_load, v_add + ds_write); the
  // wave issues one instruction every ~4 cycles, so this count is what the staging phase costs.
  constexpr int OOB = (int)0x80000000;
  const int sample_bytes = C * HW * 4;
  auto make_rsrc = [&](const Chunk& c) {
    const long long left = (long long)(N - c.s0) * sample_bytes;
    const unsigned nrec = (unsigned)min(left, 0x7ffff000ll);   // < OOB marker, >= any offset of the chunk
    return __builtin_amdgcn_make_buffer_rsrc((void*)(d.src + (long long)c.s0 * C * HW), 0, nrec, 0x00020000);
  };
  auto bload = [&](__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
  };

  // general path: a lane owns (folded row, x) positions of the patch plane and walks channel groups
  struct RowGeo { int voff, laddr; bool inr; };
  const unsigned rmagic = d.rmagic;
  const int rshift = d.rshift;
  const int rstep = 1 << rshift;                       // folded rows per pass
  const int CL = prow_step >> rshift;                  // channels staged side by side by the row lanes
  const int n_cgs = (nch + CL - 1) / CL;               // channel groups = slots per row pass; the sample
                                                       // stride SS covers n_cgs * CL planes
  auto row_geo = [&](const Chunk& c, int k, int lx, int prow0) {
    RowGeo g;
    const int ccl = prow0 >> rshift;
    const int rr = (prow0 & (rstep - 1)) + k * rstep;
    const int s = (rows_in_full == 1) ? rr : (int)__umulhi((unsigned)rr, rmagic);
    const int y = rr - s * rows_in_full;
    const int ih = c.ih_base + y * gy;
    const int iw = c.iw_base + lx * gx;
    g.inr = s < c.ns && y < c.rows_in && lx < c.cols_in;
    const bool ok = g.inr && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
    g.voff = ok ? ((s * C + ccl) * HW + ih * W + iw) * 4 : OOB;
    g.laddr = s * SS + ccl * PS + y * RS + lx;
    return g;
  };

  float st[2 * STAGE_SLOTS];

  // NCG > 0: channel groups per row pass known at compile time (slot -> (pass, group) is static and the
  // loop is straight-line); NCG == 0: run-time counters with a uniform branch per slot.
  auto issue_general = [&](auto ncg_tag, const Chunk& c, __amdgpu_buffer_rsrc_t rs, int pnl, int c_lo, int lx,
                           int prow0) {
    constexpr int NCG = decltype(ncg_tag)::value;
    const int step = CL * HW * 4;
    int soff0 = c_lo * HW * 4;
    asm volatile("" : "+s"(soff0));   // keep the per-slot offsets out of loop-invariant hoisting
    if constexpr (NCG > 0) {
#pragma unroll
      for (int k = 0; k < STAGE_SLOTS / NCG; ++k) {
        const RowGeo g = row_geo(c, k, lx, prow0);
        int soff = soff0;
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) {
          st[pnl * STAGE_SLOTS + k * NCG + cg] = bload(rs, g.voff, soff);
          soff += step;
        }
      }
    } else {
      int cg = 0, k = 0, soff = soff0;
      RowGeo g = row_geo(c, 0, lx, prow0);
#pragma unroll
      for (int j = 0; j < STAGE_SLOTS; ++j) {
        st[pnl * STAGE_SLOTS + j] = bload(rs, g.voff, soff);
        soff += step;
        if (++cg == n_cgs) { cg = 0; soff = soff0; ++k; g = row_geo(c, k, lx, prow0); }
      }
    }
  };
  auto store_general = [&](auto ncg_tag, const Chunk& c, float* lbase, int pnl, int lx, int prow0) {
    constexpr int NCG = decltype(ncg_tag)::value;
    int lstep = CL * PS;
    asm volatile("" : "+s"(lstep));
    if constexpr (NCG > 0) {
#pragma unroll
      for (int k = 0; k < STAGE_SLOTS / NCG; ++k) {
        const RowGeo g = row_geo(c, k, lx, prow0);
        if (g.inr) {
          float* l = lbase + g.laddr;
#pragma unroll
          for (int cg = 0; cg < NCG; ++cg) { *l = st[pnl * STAGE_SLOTS + k * NCG + cg]; l += lstep; }
        }
      }
    } else {
      int cg = 0, k = 0;
      RowGeo g = row_geo(c, 0, lx, prow0);
#pragma unroll
      for (int j = 0; j < STAGE_SLOTS; ++j) {
        if (g.inr) lbase[g.laddr + cg * lstep] = st[pnl * STAGE_SLOTS + j];
        if (++cg == n_cgs) { cg = 0; ++k; g = row_geo(c, k, lx, prow0); }
      }
    }
  };

  // flattened per-pixel factors: patch rows are (sample, channel) pairs, lanes run along the pixels.
  // Slot j holds rows j * prow_step + prow0; the planner guarantees prow_step | nch, so the sample and
  // the first channel of a slot are wave-uniform.
  const int flat_rows = NS * nch;

  // register staging path of this factor as one scalar: 0 float4 flat, 1 scalar flat, 7-9 general (16 / 8 / other
  // channel groups); the pre-tiled variant stages by LDS-DMA (dma_stage) and has none of this
  const int path_id = __builtin_amdgcn_readfirstlane(
      vec4 ? 0 : flat1 ? 1 : n_cgs == 16 ? 7 : n_cgs == 8 ? 8 : 9);
  auto issue_loads = [&](const Chunk& c) {
    if constexpr (PRE) { (void)c; return; } else {          // LDS-DMA staging: everything happens in store_stage
    // lane geometry made opaque per call: otherwise per-slot values are hoisted out of the chunk loop
    // and pinned in registers, which spills
    int lx = lx_, prow0 = prow0_;
    asm volatile("" : "+v"(lx), "+v"(prow0));
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(c);
    // one scalar code for the staging path, re-read per chunk: the individual tests (vec4, flat1,
    // n_cgs == 16, ...) are loop invariants the compiler turns into a dozen 64-bit lane masks, spills, and
    // reloads with v_readlane in every chunk - vector instructions that wait for the other wave's MFMAs
    int path = path_id;
    asm volatile("" : "+s"(path));
#pragma unroll
    for (int pnl = 0; pnl < 2; ++pnl) {
      if (pnl < n_panels) {
        const int c_lo = pnl ? c_lo_j : c_lo_i;
        if (path == 0) {
          const int voff = (lx < (c.wa >> 2)) ? (prow0 * HW + c.iw_base + 4 * lx) * 4 : OOB;
          // slot j holds channel rows j * prow_step + prow0 of the folded (sample, channel) index: the
          // plane offset advances by a constant per slot, plus the jump to the next sample at a wrap
          int soff = c_lo * HW * 4, cb = 0, rows = 0;
          asm volatile("" : "+s"(soff), "+s"(cb), "+s"(rows));   // keeps the slot conditions out of the loop-invariant (spilled) set
          const int step = prow_step * HW * 4, wrap = (C - nch) * HW * 4;
#pragma unroll
          for (int j = 0; j < STAGE_SLOTS / 4; ++j) {
            if (rows < flat_rows) {
              const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
              st[pnl * STAGE_SLOTS + 4 * j + 0] = v.x;
              st[pnl * STAGE_SLOTS + 4 * j + 1] = v.y;
              st[pnl * STAGE_SLOTS + 4 * j + 2] = v.z;
              st[pnl * STAGE_SLOTS + 4 * j + 3] = v.w;
            } else {
              // the slot is not stored either: tell the compiler its registers hold nothing worth keeping, or it
              // copies the previous chunk's values around the skipped load (4 v_mov per load, in the middle of an
              // issue phase whose vector instructions wait for the other wave's MFMAs)
              asm volatile("" : "=v"(st[pnl * STAGE_SLOTS + 4 * j + 0]), "=v"(st[pnl * STAGE_SLOTS + 4 * j + 1]),
                                "=v"(st[pnl * STAGE_SLOTS + 4 * j + 2]), "=v"(st[pnl * STAGE_SLOTS + 4 * j + 3]));
            }
            rows += prow_step; soff += step; cb += prow_step;
            if (cb == nch) { cb = 0; soff += wrap; }
          }
        } else if (path == 1) {
          const int voff = (lx < c.wa) ? (prow0 * HW + c.iw_base + lx) * 4 : OOB;
          int soff = c_lo * HW * 4, cb = 0, rows = 0;
          asm volatile("" : "+s"(soff), "+s"(cb), "+s"(rows));   // keeps the slot conditions out of the loop-invariant (spilled) set
          const int step = prow_step * HW * 4, wrap = (C - nch) * HW * 4;
#pragma unroll
          for (int j = 0; j < STAGE_SLOTS; ++j) {
            if (rows < flat_rows) st[pnl * STAGE_SLOTS + j] = bload(rs, voff, soff);
            else asm volatile("" : "=v"(st[pnl * STAGE_SLOTS + j]));      // see the float4 path
            rows += prow_step; soff += step; cb += prow_step;
            if (cb == nch) { cb = 0; soff += wrap; }
          }
        } else if (path == 7) {
          issue_general(std::integral_constant<int, 16>{}, c, rs, pnl, c_lo, lx, prow0);
        } else if (path == 8) {
          issue_general(std::integral_constant<int, 8>{}, c, rs, pnl, c_lo, lx, prow0);
        } else {
          issue_general(std::integral_constant<int, 0>{}, c, rs, pnl, c_lo, lx, prow0);
        }
      }
    }
    }
  };

  // pre-tiled source: the image of (panel, sample) is the run of SS words that starts at plane (plane0 + s) * C + c_lo
  // of the copy; the four waves take its 1 KiB pieces in turn (lane l of a piece moves bytes [16 l, 16 l + 16)), the
  // last piece is cut to the lanes inside the run.  Words of the image past the run (and the 16 slack words behind the
  // panel) keep the zeros of the initial fill.
  const int ppr = d.ppr, tail_lanes = d.tail_lanes;
  auto dma_stage = [&](const Chunk& c) {
    const long long bytes = ((long long)n_chunks * NS * C + nch) * PS * 4 + 64;
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)d.src, 0, (unsigned)min(bytes, 0xfffff000ll), 0x00020000);
    const int voff = lane * 16;
    const int plane_b = PS * 4;
    for (int pnl = 0; pnl < n_panels; ++pnl) {
      const int c_lo = pnl ? c_lo_j : c_lo_i;
      const unsigned lpanel = (unsigned)(PATCH_OFF + (pnl ? off_j : 0)) * 4u;
      for (int s = 0; s < c.ns; ++s) {
        const int run = ((c.plane0 + s) * C + c_lo) * plane_b;            // bytes (the planner keeps the copy < 2 GB)
        const unsigned lrun = lpanel + (unsigned)(s * SS) * 4u;
        for (int j = wave; j < ppr; j += 4) {
          if (j < ppr - 1 || lane < tail_lanes)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_void*)(l3 + lrun + j * 1024), 16, voff, run + j * 1024, 0, 0);
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);                    // vmcnt(0): this wave's pieces have landed
  };

  // registers -> LDS patch
  auto store_stage = [&](const Chunk& c) {
    if constexpr (PRE) { dma_stage(c); return; } else {
    int lx = lx_, prow0 = prow0_;
    asm volatile("" : "+v"(lx), "+v"(prow0));
    int path = path_id;                                   // see issue_loads
    asm volatile("" : "+s"(path));
#pragma unroll
    for (int pnl = 0; pnl < 2; ++pnl) {
      if (pnl < n_panels) {
        float* lbase = fs + PATCH_OFF + (pnl ? off_j : 0);
        if (path == 0) {
          if (lx < (c.wa >> 2)) {
            float* l = lbase + prow0 * PS + 4 * lx;
            int rows = 0, lstep = prow_step * PS;     // SS = nch * PS: the sample wrap needs no extra step
            asm volatile("" : "+s"(rows), "+s"(lstep));
#pragma unroll
            for (int j = 0; j < STAGE_SLOTS / 4; ++j) {
              if (rows < flat_rows) {
                l[0] = st[pnl * STAGE_SLOTS + 4 * j + 0];
                l[1] = st[pnl * STAGE_SLOTS + 4 * j + 1];
                l[2] = st[pnl * STAGE_SLOTS + 4 * j + 2];
                l[3] = st[pnl * STAGE_SLOTS + 4 * j + 3];
              }
              rows += prow_step; l += lstep;
            }
          }
        } else if (path == 1) {
          if (lx < c.wa) {
            float* l = lbase + prow0 * PS + lx;
            int rows = 0, lstep = prow_step * PS;     // SS = nch * PS: the sample wrap needs no extra step
            asm volatile("" : "+s"(rows), "+s"(lstep));
#pragma unroll
            for (int j = 0; j < STAGE_SLOTS; ++j) {
              if (rows < flat_rows) *l = st[pnl * STAGE_SLOTS + j];
              rows += prow_step; l += lstep;
            }
          }
        } else if (path == 7) {
          store_general(std::integral_constant<int, 16>{}, c, lbase, pnl, lx, prow0);
        } else if (path == 8) {
          store_general(std::integral_constant<int, 8>{}, c, lbase, pnl, lx, prow0);
        } else {
          store_general(std::integral_constant<int, 0>{}, c, lbase, pnl, lx, prow0);
        }
      }
    }
    }
  };

  f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};

  // ---- MFMA loop over a chunk ----
  // The chunk's k index is (sample s, output row r, output column w); the LDS word of operand row i at k is
  //   lane_base(i) + s SS + r cy + w cx.
  // An MFMA consumes two k values, one per lane half h.  They are paired
  //   along the output row   (wa even, or nothing better): k = (s, r, 2 j + h)   step 2 cx per MFMA group
  //   along the samples      (wa odd, ns even):            k = (2 s' + h, r, j)   step cx
  // so that every lane walks an output row with a CONSTANT stride: one address per operand row (a scalar k offset
  // folded into the lane's base by one v_mad) serves GU consecutive steps through the immediate offsets of the LDS
  // reads.  The loop runs over units = (row run, group of GU steps); its only vector-ALU work is those four
  // address instructions per unit of up to 4 GU MFMAs (the table-driven loop it replaces spent ~2 per MFMA on
  // rebuilding addresses, and vector instructions do not hide behind another wave's MFMAs on this chip).
  // With an odd row length and an odd sample count the last k of a row has no partner: it gets a unit of its own
  // (one step) in which the h = 1 lanes of the A operand read the constant zero words instead of the patch.
  int bbase[4], kmul_[4];                      // operand row bases as LDS byte addresses; 1 for patch rows, 0 for the
#pragma unroll                                 // bias / padding rows (their address ignores k)
  for (int o = 0; o < 4; ++o) { bbase[o] = base[o] * 4; kmul_[o] = kmask[o] & 1; }

  struct Ops { float a0[GU], a1[GU], b0[GU], b1[GU]; };
  auto load_ops = [&](auto ws_tag, Ops& op, int koff, const int (&hb)[4], const int (&kmul)[4], int ws_rt) {
    constexpr int WS = decltype(ws_tag)::value;          // words between consecutive steps of a lane (0: run time)
    const char* lds = reinterpret_cast<const char*>(fs);
    const unsigned kk = (unsigned)koff & 0xffffffu;      // scalar byte offset of the unit's first k
    const int pa0 = (int)__umul24(kk, (unsigned)kmul[0]) + hb[0];        // one v_mad_u32_u24 each: the lane's base
    const int pa1 = (int)__umul24(kk, (unsigned)kmul[1]) + hb[1];        // (+ its half's offset) + k offset, or,
    const int pb0 = (int)__umul24(kk, (unsigned)kmul[2]) + hb[2];        // for bias / padding rows, the base alone
    const int pb1 = (int)__umul24(kk, (unsigned)kmul[3]) + hb[3];
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      // WS > 0: the step offsets are immediates of the LDS reads.  Bias / padding rows read their GU steps from
      // the 16 constant words at ONE_OFF / ZERO_OFF (GU * WS <= 16 words)
      const int so = (WS > 0) ? u * WS * 4 : u * ws_rt * 4;
      op.a0[u] = *reinterpret_cast<const float*>(lds + pa0 + (WS > 0 ? so : (int)__umul24((unsigned)so, (unsigned)kmul[0])));
      op.a1[u] = *reinterpret_cast<const float*>(lds + pa1 + (WS > 0 ? so : (int)__umul24((unsigned)so, (unsigned)kmul[1])));
      op.b0[u] = *reinterpret_cast<const float*>(lds + pb0 + (WS > 0 ? so : (int)__umul24((unsigned)so, (unsigned)kmul[2])));
      op.b1[u] = *reinterpret_cast<const float*>(lds + pb1 + (WS > 0 ? so : (int)__umul24((unsigned)so, (unsigned)kmul[3])));
    }
  };
  // Which of the wave's four 32x32 blocks it accumulates (diagonal tiles: see `part` above) is decided by four
  // wave-uniform flags, i.e. scalar branches around the MFMAs (a handful of scalar instructions next to 64-cycle
  // matrix instructions) instead of one compiled loop body per role: with 4 roles x 4 strides x 2 tile sizes
  // inlined, the register allocator spilled thousands of vector registers.
  const bool do00 = part != 3, do01 = part != 2, do10 = (part == 0 || part == 2), do11 = part != 2;
  auto compute_ops = [&](Ops& op, int cnt) {
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      if (u < cnt) {
        const float a0 = op.a0[u], a1 = op.a1[u];
        if (do00) acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, op.b0[u], acc00, 0, 0, 0);
        if (do01) acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, op.b1[u], acc01, 0, 0, 0);
        if (do10) acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, op.b0[u], acc10, 0, 0, 0);
        if (do11) acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, op.b1[u], acc11, 0, 0, 0);
      }
    }
  };

  auto mfma_chunk = [&](auto ws_tag, const Chunk& c, bool by_sample) {
    const int ws_rt = by_sample ? cx : 2 * cx;
    const int nsteps = by_sample ? c.wa : c.wa >> 1;             // MFMA groups (k pairs) per row run
    const int ngroups = (nsteps + GU - 1) / GU;                  // units per row run
    const int n_runs = (by_sample ? c.ns >> 1 : c.ns) * c.ra;    // row runs: (sample [pair], output row)
    const int s_bytes = (by_sample ? 2 * SS : SS) * 4, r_bytes = cy * 4, g_bytes = GU * ws_rt * 4;
    const int hoff = h ? (by_sample ? SS : cx) * 4 : 0;
    int hb[4];                                                   // lane bases with the lane half's k offset folded in
#pragma unroll
    for (int o = 0; o < 4; ++o) hb[o] = bbase[o] + (hoff & kmask[o]);
    const int n_units = n_runs * ngroups;
    // unit -> (sample index, row, group) kept as scalar counters; this wave takes every KSTRIDE-th unit
    int unit = kfirst;
    if (unit < n_units) {
      int g = unit % ngroups, rr = unit / ngroups;
      int r = rr % c.ra, si = rr / c.ra;
      auto advance = [&]() {
        unit += KSTRIDE;
        g += KSTRIDE;
        while (g >= ngroups) { g -= ngroups; if (++r == c.ra) { r = 0; ++si; } }
      };
      Ops A, B;
      load_ops(ws_tag, A, si * s_bytes + r * r_bytes + g * g_bytes, hb, kmul_, ws_rt);
      while (true) {
        const int cnt_a = min(GU, nsteps - g * GU);
        advance();
        const bool more_b = unit < n_units;
        // retire A's reads (issued a whole unit ago) before B's are issued: the MFMAs below then never wait on LDS
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
        if (more_b) load_ops(ws_tag, B, si * s_bytes + r * r_bytes + g * g_bytes, hb, kmul_, ws_rt);
        compute_ops(A, cnt_a);
        if (!more_b) break;
        const int cnt_b = min(GU, nsteps - g * GU);
        advance();
        const bool more_a = unit < n_units;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        if (more_a) load_ops(ws_tag, A, si * s_bytes + r * r_bytes + g * g_bytes, hb, kmul_, ws_rt);
        compute_ops(B, cnt_b);
        if (!more_a) break;
      }
    }
    if (!by_sample && (c.wa & 1)) {
      // odd row length, no sample to pair with: the last column of every row, one step per row run, with the
      // h = 1 lanes of the A operand reading the constant zero words.  Rare (odd widths with an odd sample
      // count), short, and kept out of the loop above so that its operands stay free of selects.
      const char* lds = reinterpret_cast<const char*>(fs);
      const int tail_bytes = (c.wa - 1) * cx * 4;
      for (int rr = kfirst; rr < n_runs; rr += KSTRIDE) {
        const int r = rr % c.ra, si = rr / c.ra;
        const unsigned kk = (unsigned)(si * s_bytes + r * r_bytes + tail_bytes) & 0xffffffu;
        const int pa0 = h ? ZERO_OFF * 4 : (int)__umul24(kk, (unsigned)kmul_[0]) + bbase[0];
        const int pa1 = h ? ZERO_OFF * 4 : (int)__umul24(kk, (unsigned)kmul_[1]) + bbase[1];
        const int pb0 = (int)__umul24(kk, (unsigned)kmul_[2]) + bbase[2];
        const int pb1 = (int)__umul24(kk, (unsigned)kmul_[3]) + bbase[3];
        const float a0 = *reinterpret_cast<const float*>(lds + pa0), a1 = *reinterpret_cast<const float*>(lds + pa1);
        const float b0 = *reinterpret_cast<const float*>(lds + pb0), b1 = *reinterpret_cast<const float*>(lds + pb1);
        if (do00) acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc00, 0, 0, 0);
        if (do01) acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc01, 0, 0, 0);
        if (do10) acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc10, 0, 0, 0);
        if (do11) acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc11, 0, 0, 0);
      }
    }
  };

  // sliced factor: chunks [slice cpi, + cpi) into a slab; unsliced (direct, 128x128 tiles only) factor: all chunks, the
  // accumulators flushed into the factor behind every cpi of them (d.direct segments, see syrk_plan.h)
  const int nseg = (TMv == 128) ? d.direct : 0;
  const int ch_begin = nseg ? 0 : slice * d.cpi;
  const int ch_end = nseg ? n_chunks : min(ch_begin + d.cpi, n_chunks);
  int seg = 0, seg_end = nseg ? min(d.cpi, ch_end) : ch_end + 1;

  __syncthreads();                       // ZERO/ONE visible
  Chunk cur = decode_chunk(min(ch_begin, n_chunks - 1));
  if (ch_begin < ch_end) issue_loads(cur);

  for (int ch = ch_begin; ch < ch_end; ++ch) {
    // (all waves are past the MFMA loop of the previous chunk here)
#ifdef CURV_DIAG
    if (!(d.pad0 & 10) || ch == ch_begin) store_stage(cur);
#else
    store_stage(cur);
#endif
    __syncthreads();

    const Chunk work = cur;
    if (ch + 1 < ch_end) {
      if constexpr (PRE) cur = next_chunk_pre(work, ch + 1);
      else cur = decode_chunk(ch + 1);
#ifdef CURV_DIAG
      if (!(d.pad0 & 6)) issue_loads(cur);
#else
      issue_loads(cur);                                     // in flight during the MFMA loop below
#endif
    }

    // ---- MFMA over this wave's share of the chunk ----
#ifdef CURV_DIAG
    if (!(d.pad0 & 1)) {
#else
    {
#endif
      const bool by_sample = (work.wa & 1) && !(work.ns & 1) && work.ns > 1;
      const int ws = by_sample ? cx : 2 * cx;              // words between consecutive steps: 1, 2 or 4
      if (ws == 1) mfma_chunk(std::integral_constant<int, 1>{}, work, by_sample);
      else if (ws == 2) mfma_chunk(std::integral_constant<int, 2>{}, work, by_sample);
      else if (ws == 4) mfma_chunk(std::integral_constant<int, 4>{}, work, by_sample);
      else mfma_chunk(std::integral_constant<int, 0>{}, work, by_sample);     // strides > 2: step offsets at run time
    }
    __syncthreads();
    if constexpr (TMv == 128) {
      if (ch + 1 == seg_end) {
        // end of a segment of an unsliced item: the wave's quadrant goes straight into the factor (scaled, added) and,
        // behind the last segment, into its mirror image; the next chunk's loads are in flight meanwhile
        direct_store_quadrant(d, part, i0 + 64 * wm, j0 + 64 * wn, r32, h, acc00, acc01, acc10, acc11, seg == 0,
                              ch + 1 == ch_end);
        acc00 = 0.0f; acc01 = 0.0f; acc10 = 0.0f; acc11 = 0.0f;
        ++seg;
        seg_end = min(seg_end + d.cpi, ch_end);
      }
    }
  }
  if (nseg) return;

  gfloat* slab = (gfloat*)slabs + d.slab_base + (long long)local * (TMv * TMv);
  if (TMv == 64 && PRE) {
    // cross-wave reduction of the four K shares in two rounds (the smaller patch region holds two 64x64 tiles):
    // waves 2, 3 park theirs, waves 0, 1 add them to their own and park the sums, then one coalesced slab write
    auto park = [&](float* red) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        red[row * 64 + r32] = acc00[reg];
        red[row * 64 + 32 + r32] = acc01[reg];
        red[(32 + row) * 64 + r32] = acc10[reg];
        red[(32 + row) * 64 + 32 + r32] = acc11[reg];
      }
    };
    float* red = fs + PATCH_OFF + (wave & 1) * (64 * 64);
    if (wave >= 2) park(red);
    __syncthreads();
    if (wave < 2) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        acc00[reg] += red[row * 64 + r32];
        acc01[reg] += red[row * 64 + 32 + r32];
        acc10[reg] += red[(32 + row) * 64 + r32];
        acc11[reg] += red[(32 + row) * 64 + 32 + r32];
      }
      park(red);
    }
    __syncthreads();
    const f32x4* r0 = reinterpret_cast<const f32x4*>(fs + PATCH_OFF);
    gf32x4* slab4 = reinterpret_cast<gf32x4*>(slab);
    for (int e = tid; e < 64 * 64 / 4; e += SYRK_THREADS) slab4[e] = r0[e] + r0[1024 + e];
  } else if (TMv == 64) {
    // cross-wave reduction of the four K shares, then one coalesced slab write
    float* red = fs + PATCH_OFF + wave * (64 * 64);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
      red[row * 64 + r32] = acc00[reg];
      red[row * 64 + 32 + r32] = acc01[reg];
      red[(32 + row) * 64 + r32] = acc10[reg];
      red[(32 + row) * 64 + 32 + r32] = acc11[reg];
    }
    __syncthreads();
    const f32x4* r0 = reinterpret_cast<const f32x4*>(fs + PATCH_OFF);
    gf32x4* slab4 = reinterpret_cast<gf32x4*>(slab);
    for (int e = tid; e < 64 * 64 / 4; e += SYRK_THREADS)
      slab4[e] = r0[e] + r0[1024 + e] + r0[2048 + e] + r0[3072 + e];
  } else if (!idle) {
    // each wave owns one 64x64 quadrant of the 128x128 slab
    gfloat* q = slab + (64 * wm) * 128 + 64 * wn;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
      if (part != 3) q[row * 128 + r32] = acc00[reg];
      if (part != 2) q[row * 128 + 32 + r32] = acc01[reg];
      if (part != 3) q[(32 + row) * 128 + r32] = acc10[reg];
      if (part != 2) q[(32 + row) * 128 + 32 + r32] = acc11[reg];
    }
  }
}

__global__ void __launch_bounds__(SYRK_THREADS, 2)
syrk_patch_kernel(const FactorDev* __restrict__ descs, int n_factors, int n_items,
                  float* __restrict__ slabs, const float* __restrict__ zeros) {
  __shared__ __attribute__((aligned(16))) int smem[SMEM_WORDS];
  // XCD-aware item order: workgroups that share an XCD (equal blockIdx % 8) take every 8th group of
  // XCD_GROUP consecutive items, i.e. neighbouring tiles of one k-slice of one factor, so the panels
  // they stage hit that XCD's L2, while every XCD still sees an even mix of all factors.
  const int item = xcd_item(blockIdx.x);
  if (item >= n_items) return;
  const int f = find_segment(descs, n_factors, item, false);
  const FactorDev& d = descs[f];
  const int local = item - d.item_base;
  if (d.TM == 128) syrk_body<128, false>(d, local, slabs, zeros, smem, (lds_char*)smem);
  else syrk_body<64, false>(d, local, slabs, zeros, smem, (lds_char*)smem);
}

// the same body with LDS-DMA staging from the pre-tiled copies (syrk_pre.hip): its own work list
__global__ void __launch_bounds__(SYRK_THREADS, PRE_WGS)
syrk_pre_kernel(const FactorDev* __restrict__ descs, int n_factors, int n_items, float* __restrict__ slabs) {
  __shared__ __attribute__((aligned(1024))) int smem[PRE_SMEM_WORDS];
  const int item = xcd_item(blockIdx.x);
  if (item >= n_items) return;
  const int f = find_segment(descs, n_factors, item, false);
  const FactorDev& d = descs[f];
  const int local = item - d.item_base;
  if (d.TM == 128) syrk_body<128, true>(d, local, slabs, nullptr, smem, (lds_char*)smem);
  else syrk_body<64, true>(d, local, slabs, nullptr, smem, (lds_char*)smem);
}

// Sum the k-slices of one 64x64 sub-tile in slice order, scale, add into the factor and its mirror.
// (descs2 / n2 / split: a second work list behind the first one in the same launch - workgroups from `split` on serve it;
// two consecutive launches of 700 and 190 workgroups took 92 + 113 us, one of 890 takes the longer of the two)
__global__ void __launch_bounds__(SYRK_THREADS)
syrk_reduce_kernel(const FactorDev* __restrict__ descs, int n_factors, const float* __restrict__ slabs,
                   const FactorDev* __restrict__ descs2, int n2, int split) {
  __shared__ float tile[64][65];
  const int tid = threadIdx.x;
  int bid = blockIdx.x;
  if (descs2 != nullptr && bid >= split) { descs = descs2; n_factors = n2; bid -= split; }
  const int f = find_segment(descs, n_factors, bid, true);
  const FactorDev& d = descs[f];
  const int TMv = d.TM;
  const int q = TMv >> 6;                       // sub-tiles per tile edge
  const int sub = bid - d.sub_base;
  const int t = sub / (q * q);
  const int qq = sub - t * (q * q);
  const int qi = qq / q, qj = qq - qi * q;
  int ti, tj;
  decode_tile_of(d, t, ti, tj);
  const int bi = ti * q + qi, bj = tj * q + qj;  // 64-granular block coordinates
  const bool nonsym = d.nonsym != 0;            // correlation (syrk_corr.hip): every block is its own, no mirror
  if (bi > bj && !nonsym) return;               // mirror of (bj, bi)
  const bool diag = (bi == bj) && !nonsym;
  const int i0 = bi * 64, j0 = bj * 64, dim = d.dim;
  if (i0 >= dim || j0 >= dim) return;
  const int n_tiles = d.n_tiles, n_slices = d.n_slices;
  const float scale = d.scale;
  const bool first = d.first != 0;
  gfloat* __restrict__ dst = (gfloat*)d.dst;

  const float* s0 = slabs + d.slab_base + (long long)t * (TMv * TMv) + (qi * 64) * TMv + qj * 64;
  const long long slice_stride = (long long)n_tiles * (TMv * TMv);
  // The slices are added in slice order per element (bit-reproducible) with the loads of a thread's four elements x
  // eight slices in flight together: a tile with 30-40 slices paid a memory round trip per slice and element
  // otherwise (48 us for LeNet's conv factors).  Rows / columns beyond the factor's edge are neither read nor summed
  // (a 26-wide factor uses a sixth of its 64 x 64 block).
  {
    const float* p[4];
    f32x4 v[4];
    bool live[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + u * SYRK_THREADS;
      const int r = e >> 4, c = (e & 15) << 2;
      live[u] = i0 + r < dim && j0 + c < dim;            // (the mirror pass reads the same elements, transposed)
      p[u] = s0 + r * TMv + c;
      v[u] = live[u] ? *reinterpret_cast<const f32x4*>(p[u]) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    int s = 1;
    for (; s + 8 <= n_slices; s += 8) {
      f32x4 w[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (live[u]) {
#pragma unroll
          for (int q = 0; q < 8; ++q) w[u][q] = *reinterpret_cast<const f32x4*>(p[u] + (long long)(s + q) * slice_stride);
        }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (live[u]) {
#pragma unroll
          for (int q = 0; q < 8; ++q) v[u] += w[u][q];
        }
    }
    for (; s < n_slices; ++s) {
      f32x4 w[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) if (live[u]) w[u] = *reinterpret_cast<const f32x4*>(p[u] + (long long)s * slice_stride);
#pragma unroll
      for (int u = 0; u < 4; ++u) if (live[u]) v[u] += w[u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + u * SYRK_THREADS;
      const int r = e >> 4, c = (e & 15) << 2;
      tile[r][c + 0] = v[u].x * scale;
      tile[r][c + 1] = v[u].y * scale;
      tile[r][c + 2] = v[u].z * scale;
      tile[r][c + 3] = v[u].w * scale;
    }
  }
  __syncthreads();
  // the factor's 16 + 16 elements of a thread are loaded before the first one is stored: one element per pass waited for
  // its own load (32 dependent round trips per thread: most of the launch's 90 us)
  constexpr int PASSES = 64 * 64 / SYRK_THREADS;
  {
    float old[PASSES];
#pragma unroll
    for (int u = 0; u < PASSES; ++u) {
      const int e = tid + u * SYRK_THREADS, r = e >> 6, c = e & 63;
      const int gi = i0 + r, gj = j0 + c;
      old[u] = (!first && gi < dim && gj < dim) ? dst[(long long)gi * dim + gj] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < PASSES; ++u) {
      const int e = tid + u * SYRK_THREADS, r = e >> 6, c = e & 63;
      const int gi = i0 + r, gj = j0 + c;
      if (gi < dim && gj < dim) {
        const float v = diag ? tile[min(r, c)][max(r, c)] : tile[r][c];
        dst[(long long)gi * dim + gj] = first ? v : old[u] + v;
      }
    }
  }
  if (!diag && !nonsym) {
    float old[PASSES];
#pragma unroll
    for (int u = 0; u < PASSES; ++u) {
      const int e = tid + u * SYRK_THREADS, r = e >> 6, c = e & 63;      // r indexes panel j, c panel i
      const int gi = j0 + r, gj = i0 + c;
      old[u] = (!first && gi < dim && gj < dim) ? dst[(long long)gi * dim + gj] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < PASSES; ++u) {
      const int e = tid + u * SYRK_THREADS, r = e >> 6, c = e & 63;
      const int gi = j0 + r, gj = i0 + c;
      if (gi < dim && gj < dim) dst[(long long)gi * dim + gj] = first ? tile[c][r] : old[u] + tile[c][r];
    }
  }
}

// The descriptor table travels as kernel arguments (copied by the runtime at launch time), so the
// call is fully asynchronous and needs neither pinned staging memory nor a stream synchronisation.
constexpr int UPLOAD_CHUNK = 13;
constexpr int ZERO_PAD_FLOATS = 64;    // dummy load target of masked staging slots
struct TableChunk { FactorDev f[UPLOAD_CHUNK]; };
static_assert(sizeof(TableChunk) <= 3840, "kernel argument block must stay below 4 KB");

__global__ void __launch_bounds__(256)
upload_table_kernel(FactorDev* __restrict__ table, TableChunk chunk, int count, float* __restrict__ zeros) {
  const int words = count * (int)(sizeof(FactorDev) / 4);
  const int* in = reinterpret_cast<const int*>(&chunk);
  int* out = reinterpret_cast<int*>(table);
  for (int w = threadIdx.x; w < words; w += blockDim.x) out[w] = in[w];
  if (zeros != nullptr && threadIdx.x < ZERO_PAD_FLOATS) zeros[threadIdx.x] = 0.0f;
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

static int ceil_log2(int v) {
  int s = 0;
  while ((1 << s) < v) ++s;
  return s;
}

struct ChunkGeom { int rows_in, cols_in, RS, PS, SS, cshift, rshift, pre; };

// Bank multiplicity of the per-lane operand gather for LDS strides RS = r, PS = p (mod 32): the 32
// lanes of a half-wave read rows i .. i+31 of the unfolded matrix, i.e. words c*PS + a*RS + b with
// (c, a, b) the (channel, kh, kw) digits of i.  (r, p) = (kw, kh*kw) is conflict-free by construction;
// other residues are accepted when no bank is hit more than twice, which lets small feature maps use
// tighter strides (a 9-wide patch row padded to 35 words wastes 4x LDS).
struct ConflictTable {
  int kh = 0, kw = 0;
  unsigned char mult[32][32];
};

static const ConflictTable& conflict_table(int kh, int kw) {
  static thread_local std::vector<ConflictTable> memo;   // pure function cache
  for (const auto& t : memo) if (t.kh == kh && t.kw == kw) return t;
  ConflictTable t;
  t.kh = kh; t.kw = kw;
  const int q = kh * kw;
  for (int r = 0; r < 32; ++r) {
    for (int p = 0; p < 32; ++p) {
      int worst = 0;
      for (int phase = 0; phase < q; ++phase) {
        int hist[32] = {0};
        for (int l = 0; l < 32; ++l) {
          const int i = phase + l;
          const int c = i / q, rem = i % q, a = rem / kw, b = rem % kw;
          const int bank = (c * p + a * r + b) & 31;
          worst = std::max(worst, ++hist[bank]);
        }
      }
      t.mult[r][p] = (unsigned char)std::min(worst, 255);
    }
  }
  memo.push_back(t);
  return memo.back();
}

static void pick_strides(const FactorDev& f, int rows_in, int cols_in, int& RS, int& PS) {
  const ConflictTable& t = conflict_table(f.kh, f.kw);
  int best_ps[3] = {0, 1 << 30, 1 << 30}, best_rs[3] = {0, 0, 0};
  for (int r = 0; r < 32; ++r) {
    const int rs = round_mod32(cols_in, r);
    for (int p = 0; p < 32; ++p) {
      const int m = t.mult[r][p];
      if (m > 2) continue;
      const int ps = round_mod32(rows_in * rs, p);
      if (ps < best_ps[m]) { best_ps[m] = ps; best_rs[m] = rs; }
    }
  }
  // prefer conflict-free unless the 2-way layout is markedly smaller
  const int m = (best_ps[2] * 5 < best_ps[1] * 4) ? 2 : 1;
  RS = best_rs[m];
  PS = best_ps[m];
}

// A launch that is small as a whole (LeNet: 10 factors, 0.1 GFLOP) is bound by its longest work item, and an item is
// at least one chunk: such launches cap the chunk at SMALL_CHUNK_PX k values instead of filling the LDS panel
// (conv1 of LeNet-5: 5 samples x 28 x 28 = 3920 px per chunk, 20 items of 64 us -> 1 sample, 50 items of 2 chunks).
#ifndef CURV_SMALL_CHUNK_PX
#define CURV_SMALL_CHUNK_PX 1024
#endif
static thread_local int g_chunk_px_cap = 4096;      // set by build_plan for the plan it builds (chunk_fits has no other caller)

static bool chunk_fits(const FactorDev& f, int NS, int R, int Wc, ChunkGeom& g) {
  g.rows_in = f.compact ? R : (R - 1) * f.sh + f.kh;
  g.cols_in = f.compact ? Wc : (Wc - 1) * f.sw + f.kw;
  if (f.compact) {
    g.RS = g.cols_in;
    g.PS = round_mod32(g.rows_in * g.RS, 1);
  } else {
    pick_strides(f, g.rows_in, g.cols_in, g.RS, g.PS);
  }
  g.SS = f.nch * g.PS;
  g.rshift = 0;
  g.pre = 0;
  const bool pre = !f.compact && Wc == f.Wo;                        // full-width chunk of a kh x kw > 1 convolution
  if (pre) g.SS = (g.SS + 3) & ~3;                                  // DMA pieces are whole 16-byte lanes
  if ((long long)NS * g.SS + 16 > (pre ? PRE_PANEL_WORDS : PANEL_WORDS)) return false;   // + slack for padded run elements
  if ((long long)NS * R * Wc > 4096) return false;
  if ((long long)NS * R * Wc > g_chunk_px_cap && NS > 1) return false;      // (small launches: fewer samples per chunk)
  if (NS > 127 || g.rows_in > 0xffff) return false;
  if ((long long)NS * f.C * f.H * f.W * 4 > 0x7fff0000ll) return false;   // buffer offsets of a chunk: 31 bits
  const long long prow = (long long)NS * f.nch * g.rows_in;
  // flat modes: a slot covers 256 >> cshift (sample, channel) rows and must not straddle samples:
  // nch is a power of two there, so it is enough to keep at least 256 / nch lanes along the pixels
  const int min_cshift = f.flat ? std::max(0, 8 - ceil_log2(f.nch)) : 0;
  if (f.vec4) {
    if (Wc % 4 != 0) return false;
    g.cshift = std::max(ceil_log2(Wc / 4), min_cshift);
    if (g.cshift > 6) return false;                                    // <= 64 lanes along x4
    if ((prow << g.cshift) * 4 > PANEL_SLOT_ELEMS) return false;       // float4 slots per lane
  } else {
    if (pre) {
      // staged by LDS-DMA from the pre-tiled copy (syrk_pre.hip): no staging registers to budget; the copy
      // ((chunks x NS x C + nch) planes of PS floats) must stay addressable with 31-bit byte offsets
      const long long chunks = cdivll(f.N, NS) * cdivll(f.Ho, R);
      if (((chunks * NS * f.C + f.nch) * g.PS + 64) * 4 > 0x7ff00000ll) return false;
      if (chunks * NS * f.C * f.C >= (1LL << 32)) return false;        // plane numbers are divided by multiply-high
      g.pre = 1;
      g.cshift = 0;
      return true;
    }
    g.cshift = std::max(ceil_log2(g.cols_in), min_cshift);
    if (g.cshift > 8) return false;                                    // <= 256 lanes along x
    if (f.flat) {
      if ((prow << g.cshift) > PANEL_SLOT_ELEMS) return false;         // rows = (sample, channel)
    } else {
      // row lanes = 256 >> cshift: 2^rshift of them walk the folded (sample, row) index, the rest
      // stage that many channels side by side; slots = row passes x channel groups
      const int rl = SYRK_THREADS >> g.cshift;
      g.rshift = std::min(ceil_log2(NS * g.rows_in), ceil_log2(rl));
      const int CL = rl >> g.rshift;
      const int n_cgs = cdiv(f.nch, CL);
      if ((long long)cdiv(NS * g.rows_in, 1 << g.rshift) * n_cgs > STAGE_SLOTS) return false;
      g.SS = n_cgs * CL * g.PS;                                          // planes of the last, partial group
      if ((long long)NS * g.SS + 16 > PANEL_WORDS) return false;
    }
  }
  return true;
}

struct Plan {
  std::vector<FactorDev> f;             // caller order
  // two kernels, two work lists: [0] the implicit-im2col patch kernel below, [1] the LDS-DMA kernel of
  // syrk_flat.hip (flattened per-pixel factors); each with its own device table (factors by descending work per
  // item) and item / sub-tile numbering, sharing one slab buffer
  std::vector<int> order[3];           // [2]: the patch kernel's pre-tiled variant (syrk_pre_kernel, LDS-DMA staging)
  int n_items[3] = {0, 0, 0};
  int n_sub[3] = {0, 0, 0};
  long long slab_floats = 0;
  // f[0 .. n_user) are the caller's factors; 3x3 / stride 1 / pad 1 ones (dma = 2) are built from shifted
  // correlations (syrk_corr.hip): their virtual factors follow in f[n_user ...) and join the LDS-DMA work list
  int n_user = 0;
  std::vector<CorrLayer> corr;
  long long area_floats = 0;
};

static int build_plan(const curv_factor_desc* descs, int n, Plan& plan);

// The plan depends on the geometry of the factors only (plus the 16-byte alignment of the sources); a
// training loop asks for the same one every step, twice per step (workspace size, then the launch).  The
// last plan is kept per thread and re-used with the pointers / scale / first flag patched in (the chunk
// search costs ~240 us per call for a ResNet-50 otherwise).
static int make_plan(const curv_factor_desc* descs, int n, Plan& plan) {
  CURV_REQUIRE(n >= 0 && (n == 0 || descs != nullptr), "curv_kfac: bad descriptor array");
  static thread_local std::vector<int> cached_key;
  static thread_local Plan cached_plan;
  std::vector<int> key;
  key.reserve((size_t)n * 12 + 1);
#ifdef CURV_DIAG
  { const char* ab = getenv("CURV_SYRK_ABLATE"); key.push_back(ab ? atoi(ab) : 0); }
#endif
  for (int i = 0; i < n; ++i) {
    const curv_factor_desc& s = descs[i];
    const int vals[12] = {s.N, s.C, s.H, s.W, s.kh, s.kw, s.sh, s.sw, s.ph, s.pw, s.has_bias,
                          (int)(reinterpret_cast<uintptr_t>(s.src) & 15)};
    key.insert(key.end(), vals, vals + 12);
  }
  if (key == cached_key && cached_plan.n_user == n) {
    plan = cached_plan;
    for (int i = 0; i < n; ++i) {
      CURV_REQUIRE(descs[i].src != nullptr && descs[i].dst != nullptr, "curv_kfac: factor %d: null pointer", i);
      plan.f[i].src = descs[i].src; plan.f[i].dst = descs[i].dst;
      plan.f[i].first = descs[i].first; plan.f[i].scale = descs[i].scale;
    }
    return CURV_OK;
  }
  const int rc = build_plan(descs, n, plan);
  if (rc == CURV_OK) { cached_key = key; cached_plan = plan; }
  return rc;
}

static int build_plan(const curv_factor_desc* descs, int n, Plan& plan) {
  plan.f.resize(n);
  plan.n_user = n;
  plan.corr.clear();
  plan.area_floats = 0;
  std::vector<double> chunk_cost(n);   // MFMA CU-cycles of one (tile, chunk)
  std::vector<double> chunk_px(n, 1.0); // k values (samples x output pixels) of one chunk
  double total_cost = 0.0;
  {
    // size of the launch in MFMA CU-cycles (64 x 64 x k = 32 k), before any chunk is planned
    double estimate = 0.0;
    for (int i = 0; i < n; ++i) {
      const curv_factor_desc& s = descs[i];
      if (s.N <= 0 || s.C <= 0 || s.kh <= 0 || s.kw <= 0 || s.sh <= 0 || s.sw <= 0) continue;   // rejected below
      const double dim = (double)s.C * s.kh * s.kw + (s.has_bias ? 1 : 0);
      const double blocks = std::ceil(dim / 64.0);
      const double ho = (s.H + 2.0 * s.ph - s.kh) / s.sh + 1, wo = (s.W + 2.0 * s.pw - s.kw) / s.sw + 1;
      estimate += 32.0 * blocks * (blocks + 1) / 2 * s.N * std::max(ho, 1.0) * std::max(wo, 1.0);
    }
    g_chunk_px_cap = estimate < 512.0 * 16.0 * CURV_ITEM_FLOOR ? CURV_SMALL_CHUNK_PX : 4096;
  }
  for (int i = 0; i < n; ++i) {
    const curv_factor_desc& s = descs[i];
    FactorDev& f = plan.f[i];
    memset(&f, 0, sizeof(f));
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
    bool flattened = false;
    if (f.compact && s.sh == 1 && s.sw == 1 && s.ph == 0 && s.pw == 0) {
      // pure per-pixel factor (1x1 conv, grad_output, Linear): one long row per (sample, channel)
      CURV_REQUIRE((long long)s.H * s.W < (1LL << 30), "curv_kfac: factor %d: plane too large", i);
      f.W = s.H * s.W; f.H = 1; f.Ho = 1; f.Wo = f.W;
      flattened = true;
    }
    f.rows = s.C * f.khkw;
    f.has_bias = s.has_bias ? 1 : 0;
    f.dim = f.rows + f.has_bias;
    f.first = s.first;
    f.scale = s.scale;
#ifdef CURV_DIAG   // diagnostic builds only (tools/make_prof_build.py): ablation switches for the phase profile
    { const char* ab = getenv("CURV_SYRK_ABLATE"); f.pad0 = ab ? atoi(ab) : 0; }
#endif
    CURV_REQUIRE((long long)f.N * f.C * f.H * f.W < (1LL << 31), "curv_kfac: factor %d: source too large", i);

    if (syrk_corr_eligible(s)) {       // no work items of its own: expanded into virtual factors below
      f.dma = 2;
      chunk_cost[i] = 0.0;
      continue;
    }
    if (!flattened && !f.has_bias && !syrk_corr_eligible(s)) {
      // Everything else without a bias row whose width suits the LDS-DMA kernel - 1x1 with a stride (X = src[:, :, ::sh,
      // ::sw]: the sampled pixels lie sh * sw apart in memory, nothing an LDS-DMA piece can fetch) and kh x kw > 1
      // convolutions that the shifted correlations do not cover (stride 2: ResNet-50's layer2-4.0.conv2) - is UNFOLDED once:
      // a pass of its own writes X = unfold(src) as a compact (N, C kh kw, Ho Wo) copy into the workspace area
      // (curvature/curvatures.py:329 materialises the same matrix; ResNet-50: 45 + 202 MB) and the factor joins the LDS-DMA
      // work list as a flattened factor of C kh kw rows.  On the register-staged kernel the strided 1x1 items ran at a few
      // percent of the matrix pipe and kept it resident - holding half of their CUs' registers - for 5 ms beside the LDS-DMA
      // kernel; the implicit-im2col kernel gathers every MFMA operand from the LDS image by 4-byte reads between scalar
      // branches (0.60 of the pipe, round 6: 1.24 ms for what the LDS-DMA kernel does in 0.9).
      static const int sub_on = getenv("CURV_FLAT_SUB") ? atoi(getenv("CURV_FLAT_SUB")) : 1;
      FactorDev g = f;
      g.C = f.rows; g.H = 1; g.W = f.Ho * f.Wo; g.kh = g.kw = g.sh = g.sw = 1; g.ph = g.pw = 0; g.compact = 1;
      if (sub_on && (long long)f.Ho * f.Wo < (1LL << 30) && (long long)f.N * f.rows * f.Ho * f.Wo < (1LL << 29) &&
          syrk_flat_eligible(g, reinterpret_cast<const void*>(16))) {
        f.sub = 1;
        f.C = g.C; f.H = 1; f.W = g.W; f.Ho = 1; f.Wo = g.W;
        f.kh = f.kw = f.sh = f.sw = 1; f.ph = f.pw = 0; f.khkw = 1; f.compact = 1;
        f.xq_off = plan.area_floats;
        plan.area_floats += ((long long)f.N * f.C * f.W + 63) & ~63LL;
        flattened = true;
      }
    }
    if (flattened && (f.sub || syrk_flat_eligible(f, s.src))) {
      // LDS-DMA kernel: whole 128-row tiles, K in stages of <= 32 pixels of one sample (syrk_flat.hip)
      f.dma = 1;
      f.pitch = f.W;
      f.TM = 128;
      f.P = cdiv(f.dim, 128);             // (the last tile row / column may be ragged: syrk_flat_eligible)
      f.n_tiles = f.P * (f.P + 1) / 2;
      f.n_chunks = syrk_flat_chunks(f.N, f.W);
      chunk_cost[i] = 16.0 * 32.0 * 4.0 + 800.0;                  // 128x128x16 = 2048 CU-cycles; + barrier / DMA wait
      chunk_px[i] = 16.0;
      total_cost += chunk_cost[i] * f.n_tiles * f.n_chunks;
      continue;
    }

    // tile edge: 128 where the padding it adds is small, 64 otherwise
    f.TM = (f.dim >= 256 && cdiv(f.dim, 128) * 128 <= f.dim + f.dim / 14) ? 128 : 64;
    f.nch = std::min(f.C, (f.khkw + f.TM - 2) / f.khkw + 1);
    f.flat = (flattened && (f.nch & (f.nch - 1)) == 0) ? 1 : 0;
    f.vec4 = (f.flat && f.W % 4 == 0 && f.W >= 4 && (reinterpret_cast<uintptr_t>(s.src) & 15) == 0) ? 1 : 0;

    // chunk extent: full-width rows if they fit, then as many rows, then as many samples
    ChunkGeom g;
    int Wc = f.Wo, R = 1, NS = 1;
    if (!chunk_fits(f, 1, 1, Wc, g)) {
      if (f.vec4 && !chunk_fits(f, 1, 1, 4, g)) f.vec4 = 0;
      if (f.flat && !f.vec4 && !chunk_fits(f, 1, 1, 1, g)) f.flat = 0;
      const int unit = f.vec4 ? 4 : 1;
      CURV_REQUIRE(chunk_fits(f, 1, 1, unit, g), "curv_kfac: factor %d: no chunk shape fits the LDS budget", i);
      int lo = 1, hi = cdiv(Wc, unit);           // in units; fits(lo), !fits(hi); monotone
      while (hi - lo > 1) {
        const int mid = (lo + hi) / 2;
        if (chunk_fits(f, 1, 1, mid * unit, g)) lo = mid; else hi = mid;
      }
      Wc = lo * unit;
      if (!f.vec4 && Wc > 4) Wc &= ~3;
      const int groups = cdiv(f.Wo, Wc);         // even out the column groups
      const int even = (cdiv(f.Wo, groups) + 3) & ~3;
      if (even <= Wc) Wc = even;
    }
    if (Wc == f.Wo) {
      while (R < f.Ho && chunk_fits(f, 1, R + 1, Wc, g)) ++R;
      if (R < f.Ho) R = cdiv(f.Ho, cdiv(f.Ho, R));          // even out the row groups
      if (R == f.Ho) while (NS < f.N && chunk_fits(f, NS + 1, R, Wc, g)) ++NS;
      if (NS > 1 && NS < f.N) NS = cdiv(f.N, cdiv(f.N, NS)); // even out the sample groups
    }
    CURV_REQUIRE(chunk_fits(f, NS, R, Wc, g), "curv_kfac: factor %d: internal chunk planning error", i);
    f.NS = NS; f.R = R; f.Wc = Wc;
    f.RS = g.RS; f.PS = g.PS; f.SS = g.SS; f.cshift = g.cshift; f.rshift = g.rshift; f.pre = g.pre;
    if (f.pre) {
      f.ppr = cdiv(f.SS, 256);                                   // 1 KiB pieces per (panel, sample) run
      f.tail_lanes = (f.SS - 256 * (f.ppr - 1)) / 4;             // SS is a multiple of 4 words
    }
    f.rmagic = (unsigned)(((1ull << 32) + (unsigned)g.rows_in - 1) / (unsigned)g.rows_in);   // unused when rows_in == 1
    f.n_rg = cdiv(f.Ho, R);
    f.n_cg = cdiv(f.Wo, Wc);
    f.n_chunks = cdiv(f.N, NS) * f.n_rg * f.n_cg;
    if (f.pre) {                                                  // its pre-tiled copy lives in the workspace area
      f.xq_off = plan.area_floats;
      plan.area_floats += (syrk_pre_floats(f) + 63) & ~63LL;
    }
    f.P = cdiv(f.dim, f.TM);
    f.n_tiles = f.P * (f.P + 1) / 2;
    const double kc = (double)NS * R * (Wc + (Wc & 1));
    const double q = f.TM / 64.0;
    chunk_cost[i] = kc * 32.0 * q * q + 1500.0;   // 64x64xk = 32 k CU-cycles; + staging / barriers
    chunk_px[i] = std::max(1.0, (double)NS * R * Wc);
    total_cost += chunk_cost[i] * f.n_tiles * f.n_chunks;
  }
  for (int i = 0; i < n; ++i) {
    if (plan.f[i].dma != 2) continue;
    CorrLayer layer;
    syrk_corr_expand(descs[i], i, plan.f, layer, plan.area_floats);
    plan.corr.push_back(layer);
    for (int k = 0; k < layer.n_vf; ++k) {
      const FactorDev& v = plan.f[layer.vf0 + k];
      // (scaled by 1.5 / 2.0, i.e. dispatched earlier and sliced finer: flat kernel 4.06 -> 3.95 / 3.99 ms, pre-tiled kernel
      // 1.30 -> 1.39 / 1.38 ms - the target item length moves with the total: a wash)
      const double cost = 16.0 * 32.0 * 4.0 + 800.0;
      chunk_cost.push_back(cost);
      chunk_px.push_back(16.0);
      total_cost += cost * v.n_tiles * v.n_chunks;
    }
  }
  const int n_all = (int)plan.f.size();
  // k-slicing: aim at ~16 items per workgroup slot (2 per CU) so that the tail of the launch is
  // a few percent, while keeping the slab traffic negligible.
  // (floor: a launch that is small as a whole must not be cut into items whose slab traffic exceeds their work)
  const double target = std::max(total_cost / (512.0 * 16.0), (double)CURV_ITEM_FLOOR);
  std::vector<double> item_cost(n_all, 0.0);
  for (int i = 0; i < n_all; ++i) {
    FactorDev& f = plan.f[i];
    if (f.dma == 2) continue;
    int cpi = (int)(target / chunk_cost[i] + 0.5);
    cpi = std::max(1, std::min(cpi, f.n_chunks));
    // 128x128 tiles whose whole K range is a few target lengths are NOT sliced: the item then owns its tile, and its
    // epilogue scales, adds into the factor and writes the mirror tile itself (FactorDev::direct) - no slab, no
    // reduce pass over it.  These items are the longest of their list and are dispatched first (longest-first list
    // scheduling), so they cost no balance; slicing stays for what it is needed for, the few-tile / huge-K factors.
    const bool unsliced = f.TM == 128 && (long long)f.dim * f.dim * 4 < (1LL << 31) &&     // (31-bit offsets of the direct epilogue)
                          (cpi >= f.n_chunks || chunk_cost[i] * f.n_chunks <= CURV_SINGLE_MAX * target);
    // Bounded accumulation chains: the fp32 MFMA accumulates with a small systematic (truncation-like) bias that grows
    // with the number of steps a sum runs through (measured on ResNet-50 gradients: -1.0e-5 of a diagonal entry after
    // 3136 steps of two pixels).  No k-slice, and no serial segment of an unsliced item, sums more than MAX_CHAIN_PX
    // pixels in one chain (a 64x64 tile's four waves split its K range: four chains), whatever the batch size.
    const int chain_cap = std::max(1, (int)((f.TM == 64 ? 4.0 : 1.0) * MAX_CHAIN_PX / chunk_px[i]));
    if (unsliced) {
      const int nseg = cdiv(f.n_chunks, chain_cap);
      f.cpi = cdiv(f.n_chunks, nseg);              // chunks per serial segment
      f.direct = cdiv(f.n_chunks, f.cpi);
      f.n_slices = 1;
      f.n_items = f.n_tiles;
      f.n_sub = 0;
      item_cost[i] = chunk_cost[i] * f.n_chunks;
      continue;
    }
    cpi = std::min(cpi, chain_cap);
    f.n_slices = cdiv(f.n_chunks, cpi);
    f.cpi = cdiv(f.n_chunks, f.n_slices);          // even out the slices
    f.n_slices = cdiv(f.n_chunks, f.cpi);
    f.n_items = f.n_slices * f.n_tiles;
    f.direct = 0;
    f.n_sub = f.n_tiles * (f.TM / 64) * (f.TM / 64);
    item_cost[i] = chunk_cost[i] * f.cpi;
  }
  // groups (FactorDev::group_n): one slicing for all members - that of the member with the most chunks; a member
  // with fewer chunks finds its last slices empty and writes zero slabs for them
  for (int i = 0; i < n_all; ++i) {
    if (plan.f[i].group_n <= 0 || plan.f[i].group_pos != 0) continue;
    const int gn = plan.f[i].group_n;
    int lead = i;
    for (int j = i; j < i + gn; ++j) if (plan.f[j].n_chunks > plan.f[lead].n_chunks) lead = j;
    double cost = 0.0;
    for (int j = i; j < i + gn; ++j) {
      FactorDev& f = plan.f[j];
      f.cpi = plan.f[lead].cpi;
      f.n_slices = plan.f[lead].n_slices;
      f.direct = plan.f[lead].direct;
      f.n_sub = f.direct ? 0 : f.n_tiles * (f.TM / 64) * (f.TM / 64);
      f.n_items = j == i ? gn * f.n_slices * f.n_tiles : 0;       // the range belongs to the first member
      cost = std::max(cost, chunk_cost[j] * (f.direct ? f.n_chunks : f.cpi));
    }
    for (int j = i; j < i + gn; ++j) item_cost[j] = cost;         // equal keys: the stable sort keeps them together
  }
  // Work items are dispatched in index order: the longest items go first, so that the tail of the
  // launch is made of the shortest ones (the resident workgroups drain within one short item).
  long long slab = 0;
  for (int k = 0; k < 3; ++k) {
    plan.order[k].clear();
    for (int i = 0; i < n_all; ++i) {
      const FactorDev& v = plan.f[i];
      const int list = v.dma == 2 ? -1 : v.dma == 1 ? 1 : v.pre ? 2 : 0;
      if (list == k) plan.order[k].push_back(i);
    }
    std::stable_sort(plan.order[k].begin(), plan.order[k].end(), [&](int a, int b) { return item_cost[a] > item_cost[b]; });
    // Graded items: a launch ends when its last item does, and with ~16 equal items per workgroup slot the slots
    // drain over the length of one item (measured: 6-10 % of both big kernels at a quarter of the slots).  The
    // factors that are dispatched last are therefore cut finer: half-length items past 70 % of the list's work,
    // quarter-length items past 90 % (a few more slabs for the reduce pass, none of it in the bulk of the launch).
    if (k != 0 && plan.order[k].size() > 1) {
      double list_cost = 0.0, seen = 0.0;
      for (int idx : plan.order[k]) if (plan.f[idx].n_items > 0 || plan.f[idx].group_n > 0)
        list_cost += chunk_cost[idx] * plan.f[idx].n_tiles * plan.f[idx].n_chunks;
      if (list_cost > 512.0 * 8.0 * target) {          // a launch of several items per slot: otherwise nothing to grade
        for (int idx : plan.order[k]) {
          FactorDev& f = plan.f[idx];
          const double mine = chunk_cost[idx] * f.n_tiles * f.n_chunks;
          const double start = seen / list_cost;
          seen += mine;
          if (f.group_n > 0 || f.direct || start < 0.7) continue;  // (groups share one slicing, unsliced items own their tiles: left alone)
          const double t = start < 0.9 ? target * 0.5 : target * 0.25;
          int cpi = (int)(t / chunk_cost[idx] + 0.5);
          cpi = std::max(1, std::min(cpi, f.n_chunks));
          if (cpi >= f.cpi) continue;
          f.n_slices = cdiv(f.n_chunks, cpi);
          f.cpi = cdiv(f.n_chunks, f.n_slices);
          f.n_slices = cdiv(f.n_chunks, f.cpi);
          f.n_items = f.n_slices * f.n_tiles;
          item_cost[idx] = chunk_cost[idx] * f.cpi;
        }
        std::stable_sort(plan.order[k].begin(), plan.order[k].end(), [&](int a, int b) { return item_cost[a] > item_cost[b]; });
      }
    }
    long long items = 0, subs = 0;
    for (int idx : plan.order[k]) {
      FactorDev& f = plan.f[idx];
      // members of a group sit behind its first entry, whose range they share (bases stay ascending)
      f.item_base = f.group_n > 0 && f.group_pos > 0 ? plan.f[idx - f.group_pos].item_base + f.group_pos : (int)items;
      f.sub_base = (int)subs;
      f.slab_base = slab;
      items += f.n_items;
      subs += f.n_sub;
      if (!f.direct) slab += (long long)f.n_slices * f.n_tiles * f.TM * f.TM;
      CURV_REQUIRE(items < (1LL << 30) && subs < (1LL << 30), "curv_kfac: too many work items");
    }
    plan.n_items[k] = (int)items;
    plan.n_sub[k] = (int)subs;
    if (getenv("CURV_PLAN_DUMP")) {       // diagnostics: the dispatch order of a list (stderr)
      for (int idx : plan.order[k]) {
        const FactorDev& f = plan.f[idx];
        fprintf(stderr, "list %d idx %4d dim %5d W %6d tiles %4d chunks %6d cpi %4d slices %3d items %5d direct %d group %d/%d nonsym %d cost/item %.0f chunk_cost %.0f\n",
                k, idx, f.dim, f.W, f.n_tiles, f.n_chunks, f.cpi, f.n_slices, f.n_items, f.direct, f.group_pos, f.group_n, f.nonsym,
                item_cost[idx], chunk_cost[idx]);
      }
    }
  }
  plan.slab_floats = slab;
  return CURV_OK;
}

static size_t table_bytes(int n) {   // descriptor table + the zeroed dummy-load pad
  return align_up((size_t)std::max(n, 1) * sizeof(FactorDev), 256) + 256;
}

static int plan_without_pointers(const curv_factor_desc* descs, int n_factors, Plan& plan) {
  // pointers are not dereferenced by the planner, but it insists on non-null ones
  std::vector<curv_factor_desc> tmp(descs, descs + (n_factors > 0 ? n_factors : 0));
  for (auto& t : tmp) {
    if (!t.src) t.src = reinterpret_cast<const float*>(16);
    if (!t.dst) t.dst = reinterpret_cast<float*>(16);
  }
  return make_plan(tmp.data(), n_factors, plan);
}

}  // namespace curv

using namespace curv;

static size_t slab_bytes(const Plan& plan) { return align_up((size_t)plan.slab_floats * sizeof(float), 256); }

extern "C" size_t curv_kfac_workspace_bytes(const curv_factor_desc* descs, int n_factors) {
  Plan plan;
  if (plan_without_pointers(descs, n_factors, plan) != CURV_OK) return 0;
  const size_t grouped = table_bytes((int)plan.f.size()) + slab_bytes(plan) + (size_t)plan.area_floats * sizeof(float);
  return std::max(grouped, kfac_small_workspace_bytes(descs, n_factors));      // (either path may take the call)
}

extern "C" int curv_kfac_path_for(const curv_factor_desc* descs, int n_factors) {
  if (descs == nullptr || n_factors <= 0) return CURV_PATH_GROUPED;
  return kfac_path_for(descs, n_factors);
}

extern "C" int curv_kfac_plan_info(const curv_factor_desc* descs, int n_factors, long long* out) {
  Plan plan;
  int rc = plan_without_pointers(descs, n_factors, plan);
  if (rc != CURV_OK) return rc;
  for (int i = 0; i < n_factors; ++i) {
    const FactorDev& f = plan.f[i];
    long long* o = out + (size_t)i * CURV_PLAN_INFO_FIELDS;
    o[0] = f.dim; o[1] = f.Ho; o[2] = f.Wo; o[3] = f.NS; o[4] = f.R; o[5] = f.Wc;
    o[6] = f.n_chunks; o[7] = f.RS; o[8] = f.PS; o[9] = f.SS; o[10] = f.nch;
    o[11] = f.n_tiles; o[12] = f.cpi; o[13] = f.n_slices; o[14] = f.n_items; o[15] = f.item_base;
    o[16] = f.TM; o[17] = f.vec4; o[18] = f.cshift; o[19] = f.n_sub; o[20] = f.direct; o[21] = f.rshift; o[22] = f.pre;
    o[23] = f.dma;
    auto flops_of = [](const FactorDev& v) {
      const double K = (double)v.N * v.Ho * v.Wo;
      return v.nonsym ? 2.0 * v.dim * v.dim * K : (double)v.dim * (v.dim + 1.0) * K;
    };
    double fl = 0.0;
    if (f.dma == 2) {
      for (const CorrLayer& layer : plan.corr)
        if (layer.user == i)
          for (int k = 0; k < layer.n_vf; ++k) fl += flops_of(plan.f[layer.vf0 + k]);
    } else {
      fl = flops_of(f);
    }
    o[24] = (long long)fl;
  }
  return CURV_OK;
}

// side stream for the register-staged patch kernel (few, long, latency-bound items: flattened factors narrower than a
// 128-row tile, strided 1x1 convolutions, Linear layers): it runs beside the padding / pre-tiling passes and the two
// LDS-DMA kernels instead of in front of them with a tail of its own
struct SyrkStreams { hipStream_t side = nullptr; hipEvent_t fork = nullptr, join = nullptr, join_r = nullptr; };
static int syrk_streams(SyrkStreams** out) {
  static thread_local std::vector<std::pair<int, SyrkStreams>> cache;
  int dev = 0;
  CURV_HIP_CHECK(hipGetDevice(&dev));
  for (auto& e : cache) if (e.first == dev) { *out = &e.second; return CURV_OK; }
  SyrkStreams s;
  // borrowed from the inversion sweep's stream set, not created: a stream of its own cost invert() 0.9 ms (invert.hip)
  const int rcs = curv_internal_side_stream(&s.side);
  if (rcs != CURV_OK) return rcs;
  CURV_HIP_CHECK(hipEventCreateWithFlags(&s.fork, hipEventDisableTiming));
  CURV_HIP_CHECK(hipEventCreateWithFlags(&s.join, hipEventDisableTiming));
  CURV_HIP_CHECK(hipEventCreateWithFlags(&s.join_r, hipEventDisableTiming));
  cache.emplace_back(dev, s);
  *out = &cache.back().second;
  return CURV_OK;
}

// what the device table of a workspace held after the previous call (CURV_KFAC_TABLE_RESIDENT: the caller vouches
// that nobody else wrote to the head of that workspace since): unchanged argument blocks are not uploaded again
struct TableShadow { const void* ws = nullptr; std::vector<FactorDev> rows; };

static int kfac_accumulate_impl(void* stream_, const curv_factor_desc* descs, int n_factors, void* workspace,
                                size_t workspace_bytes, unsigned flags, void* ev_start, void* ev_stop) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_factors == 0) return CURV_OK;
  Plan plan;
  int rc = make_plan(descs, n_factors, plan);
  if (rc != CURV_OK) return rc;
  const size_t tb = table_bytes((int)plan.f.size());
  const size_t need = tb + slab_bytes(plan) + (size_t)plan.area_floats * sizeof(float);
  if (workspace == nullptr || workspace_bytes < need) {
    set_error("curv_kfac_accumulate: workspace too small (%zu < %zu bytes)", workspace_bytes, need);
    return CURV_ERR_WORKSPACE;
  }
  FactorDev* table = reinterpret_cast<FactorDev*>(workspace);
  float* slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + tb);
  float* zeros = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + tb - 256);
  float* area = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + tb + slab_bytes(plan));
  for (const CorrLayer& layer : plan.corr) syrk_corr_bind(layer, plan.f, area);
  // device table: the patch kernel's factors, then the LDS-DMA kernel's (the caller's and the virtual ones), then
  // those of the patch kernel's pre-tiled variant
  std::vector<int> all(plan.order[0]);
  all.insert(all.end(), plan.order[1].begin(), plan.order[1].end());
  all.insert(all.end(), plan.order[2].begin(), plan.order[2].end());
  const int n_table = (int)all.size();
  static thread_local TableShadow shadow;
  {
    // a small launch (LeNet scale) is built by the two launches of syrk_small.hip instead; they use the head of the
    // workspace for their slabs, so whatever table the shadow remembers there is gone
    const int rcs = kfac_accumulate_small(stream, descs, n_factors, workspace, workspace_bytes, ev_start, ev_stop);
    if (rcs != CURV_ERR_WORKSPACE) { shadow.ws = nullptr; shadow.rows.clear(); return rcs; }
  }
  const bool resident = (flags & CURV_KFAC_TABLE_RESIDENT) && shadow.ws == workspace;
  if (!resident) shadow.rows.clear();
  shadow.ws = workspace;
  {
    // the shadow describes exactly this call's table: everything behind table_bytes(n_table) is scratch of this call
    // (zero pad, slabs, correlation area), so rows a LONGER previous table left there are gone - forget them, or the
    // next long table would match them and skip their upload
    FactorDev none;
    memset(&none, 0xff, sizeof(none));
    shadow.rows.resize(n_table, none);
  }
  for (int b = 0; b < n_table; b += UPLOAD_CHUNK) {
    TableChunk chunk;
    const int count = std::min(UPLOAD_CHUNK, n_table - b);
    memset(&chunk, 0, sizeof(chunk));
    for (int k = 0; k < count; ++k) {
      chunk.f[k] = plan.f[all[b + k]];
      if (chunk.f[k].pre || chunk.f[k].sub) chunk.f[k].src = area + chunk.f[k].xq_off;   // the kernel stages from the pre-tiled / compact copy
      chunk.f[k].xq_off = 0;
    }
    if (resident && memcmp(&shadow.rows[b], chunk.f, (size_t)count * sizeof(FactorDev)) == 0) continue;
    memcpy(&shadow.rows[b], chunk.f, (size_t)count * sizeof(FactorDev));
    hipLaunchKernelGGL(upload_table_kernel, dim3(1), dim3(256), 0, stream, table + b, chunk, count,
                       b == 0 ? zeros : nullptr);
    CURV_LAUNCH_CHECK();
  }
  const int n0 = (int)plan.order[0].size(), n1 = (int)plan.order[1].size(), n2 = (int)plan.order[2].size();
  if (ev_start) CURV_HIP_CHECK(hipEventRecord((hipEvent_t)ev_start, stream));
  // launch order on the caller's stream: padding pass, pre-tiling pass, pre-tiled patch kernel, LDS-DMA kernel; the
  // register-staged patch kernel beside them on the side stream (when there is anything for it to run beside)
  if (!plan.corr.empty()) {
    const int rcp = launch_corr_prep(stream, plan.corr, plan.f, area);
    if (rcp != CURV_OK) return rcp;
  }
  if (n2 > 0) {
    const int rcq = launch_patch_prep(stream, plan.f, plan.order[2], area);
    if (rcq != CURV_OK) return rcq;
  }
  {
    const int rcs = launch_sub_prep(stream, plan.f, plan.n_user, descs, area);
    if (rcs != CURV_OK) return rcs;
  }
  // the register-staged kernel starts beside the MFMA kernels, behind the two HBM-bound passes (beside those it
  // slowed them down by more than it gained)
  SyrkStreams* ss = nullptr;
  const bool fork = plan.n_items[0] > 0 && (plan.n_items[1] > 0 || plan.n_items[2] > 0);
  if (fork) {
    rc = syrk_streams(&ss);
    if (rc != CURV_OK) return rc;
    CURV_HIP_CHECK(hipEventRecord(ss->fork, stream));
    CURV_HIP_CHECK(hipStreamWaitEvent(ss->side, ss->fork, 0));
  }
  auto reduce = [&](hipStream_t on, int list, int first, int count) -> int {
    if (plan.n_sub[list] <= 0) return CURV_OK;
    hipLaunchKernelGGL(syrk_reduce_kernel, dim3(plan.n_sub[list]), dim3(SYRK_THREADS), 0, on, table + first, count, slabs,
                       (const FactorDev*)nullptr, 0, 0);
    CURV_LAUNCH_CHECK();
    return CURV_OK;
  };
  if (plan.n_items[0] > 0) {
    const int grid = cdiv(plan.n_items[0], 8 * XCD_GROUP) * 8 * XCD_GROUP;
    hipLaunchKernelGGL(syrk_patch_kernel, dim3(grid), dim3(SYRK_THREADS), 0, fork ? ss->side : stream, table, n0,
                       plan.n_items[0], slabs, zeros);
    CURV_LAUNCH_CHECK();
    if (fork) {
      CURV_HIP_CHECK(hipEventRecord(ss->join, ss->side));
      // its k-slices are summed on the side stream too, beside the MFMA kernels of the caller's stream
      if ((rc = reduce(ss->side, 0, 0, n0)) != CURV_OK) return rc;
    }
  }
  if (n2 > 0) {
    const int grid = cdiv(plan.n_items[2], 8 * XCD_GROUP) * 8 * XCD_GROUP;
    hipLaunchKernelGGL(syrk_pre_kernel, dim3(grid), dim3(SYRK_THREADS), 0, stream, table + n0 + n1, n2, plan.n_items[2], slabs);
    CURV_LAUNCH_CHECK();
  }
  // (summing the pre-tiled kernel's k-slices on the side stream as well, beside the LDS-DMA kernel, was measured on one
  // box: update() 6.85 -> 6.79 ms, the MFMA kernels 5.98 -> 6.05 ms - a wash; they stay behind the kernels)
  if (fork) CURV_HIP_CHECK(hipEventRecord(ss->join_r, ss->side));
  if (plan.n_items[1] > 0) {
    const int rc1 = launch_syrk_flat(stream, table + n0, n1, plan.n_items[1], slabs);
    if (rc1 != CURV_OK) return rc1;
  }
  if (!fork && (rc = reduce(stream, 0, 0, n0)) != CURV_OK) return rc;
  if (plan.n_sub[1] > 0 && plan.n_sub[2] > 0) {      // the k-slices of both MFMA kernels of this stream in one launch
    hipLaunchKernelGGL(syrk_reduce_kernel, dim3(plan.n_sub[1] + plan.n_sub[2]), dim3(SYRK_THREADS), 0, stream, table + n0, n1, slabs,
                       (const FactorDev*)(table + n0 + n1), n2, plan.n_sub[1]);
    CURV_LAUNCH_CHECK();
  } else {
    if ((rc = reduce(stream, 1, n0, n1)) != CURV_OK) return rc;
    if ((rc = reduce(stream, 2, n0 + n1, n2)) != CURV_OK) return rc;
  }
  if (!plan.corr.empty() && (rc = launch_corr_assemble(stream, plan.corr, plan.f, area)) != CURV_OK) return rc;
  // the side stream (register-staged MFMA kernel + its reduce pass) joins behind everything the caller's stream had to
  // do itself: none of the passes above reads what the side stream writes (LeNet-5: the two chains are 60 us each, and
  // ran one after the other when the join came first)
  // (join_r is recorded behind join on the same stream: one wait covers both)
  if (fork) CURV_HIP_CHECK(hipStreamWaitEvent(stream, ss->join_r, 0));
  // the timed window (bench.py's roofline) spans the WHOLE build: padding / pre-tiling passes, the MFMA kernels, the
  // k-slice reduction of the sliced factors and the assembly of the 3x3 factors
  if (ev_stop) CURV_HIP_CHECK(hipEventRecord((hipEvent_t)ev_stop, stream));
  return CURV_OK;
}

extern "C" int curv_kfac_accumulate(void* stream, const curv_factor_desc* descs, int n_factors, void* workspace,
                                    size_t workspace_bytes) {
  return kfac_accumulate_impl(stream, descs, n_factors, workspace, workspace_bytes, 0u, nullptr, nullptr);
}

extern "C" int curv_kfac_accumulate_ex(void* stream, const curv_factor_desc* descs, int n_factors, void* workspace,
                                       size_t workspace_bytes, unsigned flags, void* ev_start, void* ev_stop) {
  return kfac_accumulate_impl(stream, descs, n_factors, workspace, workspace_bytes, flags, ev_start, ev_stop);
}

extern "C" int curv_kfac_accumulate_timed(void* stream, const curv_factor_desc* descs, int n_factors,
                                          void* workspace, size_t workspace_bytes, void* ev_start, void* ev_stop) {
  return kfac_accumulate_impl(stream, descs, n_factors, workspace, workspace_bytes, 0u, ev_start, ev_stop);
}
