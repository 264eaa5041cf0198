// KFAC factor build for factors whose rows are CONTIGUOUS in memory on gfx950 - flattened per-pixel factors (1x1 stride-1
// convolutions' A side and every G side, curvature/curvatures.py:329-350 with kernel 1x1: X_s = src[s] is a (C x HW)
// row-major matrix per sample), the components of the shifted correlations (syrk_corr.hip) and, since round 6, the unfolded
// copies of stride-2 3x3 and strided 1x1 convolutions (unfold_prep_kernel below) -
//   slab(tile, slice) = sum over the slice's (sample, pixel) of X[rows_i][k] X[rows_j][k]
// for every upper-triangular 128x128 tile: five sixths of a ResNet-50's factor-build window.  There is no im2col reuse in
// such a factor: a tile streams 2 x 128 rows x K floats, and staging them through registers (load burst, LDS store pass,
// barrier) left the matrix pipe idle two thirds of the time.  Here:
//   * LDS-DMA staging: buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction, straight from global memory into a
//     double-buffered LDS image - no staging registers, no store pass, no vector-ALU work; the pieces of stage t + 1 are
//     issued behind the first MFMA groups of stage t and have the rest of the stage to land;
//   * the image is [128 rows][4 x 16 B] (16 pixels per stage; round 2: 8 x 16 B) with the 16-byte slots XOR-swizzled by
//     (row >> 2) & 3 (16 consecutive rows x one slot must cover 16 different 4-bank groups):
//     the DMA writes lane-linear bytes, so the swizzle is applied to the per-lane SOURCE address, and operands are
//     read back with conflict-free ds_read_b128 (one read = one operand row x 4 pixels = the A or B input of 4 MFMAs);
//   * the k order inside a stage is whatever suits the reads (lane half h takes pixel group 2 j + h): a SYRK only
//     needs both operands to agree on it;
//   * every stage is full and straight-line code (the K range is the stream of the factor's 4-pixel groups, flat_body);
//     every lane's read addresses are computed once per work item, the steady state carries a dozen vector-ALU
//     instructions per 32 MFMAs (the lane's place in the stream);
//   * 32 KiB of LDS and <= 128 registers: FOUR workgroups per CU, each covering the others' barriers and DMA waits
//     (round 3: 32-pixel stages, 64 KiB, two per CU -> 16-pixel stages, four per CU: the MFMA kernels of a ResNet-50
//     update() 5.89 -> 5.75 ms on one box; the stand-alone prototype's trend - 64-pixel stages at one per CU 0.42-0.50
//     of peak, 32 at two 0.57-0.60 - continues.  The same change made gemm_nt_kernel, whose tiles have short K ranges,
//     slower: 1.45 -> 1.55 ms per sample, so that kernel keeps its 32-wide stages).
// Work items, slabs and the wave roles on diagonal tiles are those of syrk.hip; syrk_reduce_kernel sums the slabs.
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
#include "syrk_plan.h"

namespace curv {

#ifndef CURV_FLAT_KC
#define CURV_FLAT_KC 16
#endif

namespace flat {
constexpr int TM = 128;
constexpr int KC = CURV_FLAT_KC;            // pixels per stage row
constexpr int ROW_B = KC * 4;               // 128 B
constexpr int SLOTS = KC / 4;               // 16-byte slots per row
constexpr int STEPS = KC / 8;               // MFMA steps (8 pixels: 4 per lane half) per full stage
constexpr int RPP = 1024 / ROW_B;           // rows per DMA piece (one wave-instruction)
constexpr int PIECES = TM / RPP / 4;        // pieces per panel per wave
constexpr int PANEL_B = TM * ROW_B;         // 16 KiB
constexpr int LDS_B = 4 * PANEL_B;          // [Pi buf0][Pi buf1][Pj buf0][Pj buf1]
constexpr int NP = 2 * PIECES;              // pieces per stage per wave
constexpr int PPS = (NP + STEPS / 2 - 1) / (STEPS / 2);   // pieces per step when issued during the first half of a stage
static_assert(PPS <= 4, "at most one DMA piece per MFMA group");
// 16-byte slots of 16 consecutive rows must fall into 16 different 4-bank groups: rows R .. R + 16 / SLOTS - 1 share
// a key, the slot is XORed with it
constexpr int KEY_SHIFT = SLOTS == 8 ? 1 : SLOTS == 4 ? 2 : 0;
constexpr int LANES_PER_ROW_SHIFT = SLOTS == 8 ? 3 : 2;
constexpr int WGS = KC == 32 ? 2 : 4;       // workgroups per CU (LDS: 64 / 32 KiB)
static_assert(SLOTS == 8 || SLOTS == 4, "stage rows of 32 or 16 pixels");
}  // namespace flat

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(1))) float gfloat_t;

// The K range of a factor is the stream of its 4-pixel GROUPS: gps = ceil(W / 4) per (sample, channel) row, sample after
// sample, N gps in all; a stage takes four consecutive groups of the stream (16 pixels) - also across a sample boundary -
// so that every stage but the factor's last is full, whatever W is (round 5 cut every sample into its own stages: 49
// pixels = 2 + 2 + 2 + 1 steps of eight, each count a run-time condition inside the MFMA sequence).
int syrk_flat_chunks(int N, int W) {
  const long long groups = (long long)N * std::max((W + 3) / 4, flat::SLOTS);   // (rows shorter than a stage: flat_body)
  return (int)((groups + flat::SLOTS - 1) / flat::SLOTS);
}

bool syrk_flat_eligible(const FactorDev& f, const void* src) {
  // flattened per-pixel factor (1x1, stride 1, no padding: H = 1, W = pixels per (sample, channel) row), no bias row, and
  // the whole tensor addressable by one buffer descriptor with 31-bit byte offsets (bit 31 of a lane's offset marks "no
  // fetch": flat_body).  The last tile row / column of a factor
  // may be ragged (DenseNet-121 / 161: 64 + 32 k / 96 + 48 k channels; any multiple of 16 from 96 on): the panel rows behind the factor's edge are the next sample's first
  // channels (zeros behind the tensor's end: the descriptor's range check) - finite values whose products land in tile rows
  // and columns that neither epilogue stores (direct_store_block masks them, syrk_reduce_kernel does not read them)
  static const int ragged = getenv("CURV_FLAT_RAGGED") ? atoi(getenv("CURV_FLAT_RAGGED")) : 1;
  if (!(f.compact && f.H == 1 && f.kh == 1 && f.kw == 1 && f.sh == 1 && f.sw == 1 && f.ph == 0 && f.pw == 0)) return false;
  if (f.has_bias || f.dim < (ragged ? 96 : flat::TM) || f.dim % (ragged ? 16 : flat::TM) != 0) return false;
  if (f.W < 8) return false;
  if ((long long)f.N * f.C * f.W * 4 >= (1LL << 31) - 4096) return false;
  return (reinterpret_cast<uintptr_t>(src) & 3) == 0;
}

template <int PART>
__device__ __forceinline__ void flat_mfma_step(const f32x4& a0, const f32x4& a1, const f32x4& b0, const f32x4& b1,
                                               f32x16& c00, f32x16& c01, f32x16& c10, f32x16& c11) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (PART != 3) c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], c00, 0, 0, 0);
    if (PART != 2) c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b1[e], c01, 0, 0, 0);
    if (PART == 0 || PART == 2) c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b0[e], c10, 0, 0, 0);
    if (PART != 2) c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], c11, 0, 0, 0);
  }
}

template <int PART>
__device__ __forceinline__ void flat_body(const FactorDev& d, int local, float* __restrict__ slabs, lds_char* lds) {
  using namespace flat;
  static_assert(KC == 16 && STEPS == 2 && PIECES == 2, "the straight-line stage below is written for 16-pixel stages");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31, h = lane >> 5;
  const int slice = local / d.n_tiles, tile = local - slice * d.n_tiles;
  int ti, tj;
  decode_tile_of(d, tile, ti, tj);
  const bool diag = (ti == tj) && !d.nonsym;       // a correlation's diagonal tiles have two different operand panels
  // wave roles as in syrk.hip: wave (wm, wn) owns a 64x64 quadrant; on a diagonal tile the redundant quadrant's
  // wave takes half of quadrant (0, 1) (parts 2 / 3) and the diagonal quadrants skip their lower-left block (part 1)
  int wm = wave >> 1, wn = wave & 1;
  if (PART >= 2) { wm = 0; wn = 1; }
  const int i0 = ti * TM, j0 = tj * TM;
  const int W = d.W, C = d.C, pitch = d.pitch;
  // groups per (sample, channel) row of the stream: ceil(W / 4), at least four (rows of fewer than 13 pixels - strips of the
  // smallest images - get phantom groups behind their end: a stage then never wraps more than once)
  const int gps = max((W + 3) >> 2, SLOTS);
  const int g_tot = d.N * gps;                              // groups of the whole K range
  // ---- DMA lane geometry: piece `slot` of this wave covers panel rows 64 slot + 16 wave + (lane >> 2); the lane's
  // physical 16-byte slot (lane & 3) holds the stage's logical group g_lane = slot ^ ((row >> 2) & 3), which does not
  // depend on the piece index (pieces of a wave are 64 rows apart).  Group G of the stream = (sample G / gps, group G % gps):
  // a lane follows its own group from stage to stage (G += 4) and carries the (sample, row, pixel) part of the address in
  // its voffset; tile row, panel offset and piece are scalars that do not change within an item.  A lane with nothing
  // to fetch - its group lies behind the K range, or the item has no further stage - carries an out-of-range voffset
  // instead of a cleared exec bit (the range check of the descriptor drops the fetch): no predicate, no branch.
  constexpr int OOB = (int)0x80000000;                      // + soffset < 2^31 (syrk_flat_eligible) stays out of range
  const int rsub = RPP * wave + (lane >> LANES_PER_ROW_SHIFT);
  const int g_lane = (lane & (SLOTS - 1)) ^ ((rsub >> KEY_SHIFT) & (SLOTS - 1));
  const unsigned total_b = (unsigned)((long long)d.N * C * pitch * 4);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)d.src, 0, total_b, 0x00020000);

  // ---- operand addresses: row R of a panel, step j: R * 64 + ((2 j + h) ^ ((R >> 2) & 3)) * 16
  unsigned addr[4][STEPS];
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    const int R = ((o < 2) ? 64 * wm : 64 * wn) + (o & 1) * 32 + r32;
    const unsigned pbase = (o < 2 || diag) ? 0u : 2u * PANEL_B;
    const int rkey = (R >> KEY_SHIFT) & (SLOTS - 1);
#pragma unroll
    for (int j = 0; j < STEPS; ++j) addr[o][j] = pbase + R * ROW_B + (((2 * j + h) ^ rkey) << 4);
  }

  f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};
  // sliced factor: stages [slice cpi, + cpi) into a slab; unsliced (direct) factor: all stages, the accumulators
  // flushed into the factor behind every cpi of them (d.direct segments: bounded fp32 accumulation chains)
  const int nseg = d.direct;
  const int t0 = nseg ? 0 : slice * d.cpi, t1 = nseg ? d.n_chunks : min(t0 + d.cpi, d.n_chunks);
  int seg = 0, seg_end = nseg ? min(d.cpi, t1) : t1 + 1;
  constexpr int N_PANELS = PART == 0 ? 2 : 1;               // PART != 0 <=> diagonal tile of a symmetric factor: one panel

  // packed pair tile (d.half): rows 64 .. 127 of a panel are the 64 channel rows again, at the panel's second offset
  const int hadj[2] = {d.half ? (d.off_i2 - d.off_i - 64 * pitch) * 4 : 0, d.half ? (d.off_j2 - d.off_j - 64 * pitch) * 4 : 0};
  const int soff[2] = {(i0 * pitch + d.off_i) * 4, (j0 * pitch + d.off_j) * 4};
  // this lane's group at stage t0 (one division per item), then incrementally: four groups on, or - behind a row's last
  // groups - into the next sample's row
  int G = SLOTS * t0 + g_lane, gi, voff_l;
  {
    const int smp = G / gps;
    gi = G - smp * gps;
    voff_l = ((smp * C + rsub) * pitch + 4 * gi) * 4;
  }
  const int wrap_b = (C * pitch + 4 * (SLOTS - gps)) * 4;
  auto lane_voff = [&](bool live) { return (live && G < g_tot) ? voff_l : OOB; };
  auto advance = [&]() {
    G += SLOTS; gi += SLOTS;
    const bool w = gi >= gps;
    gi -= w ? gps : 0;
    voff_l += w ? wrap_b : SLOTS * 16;
  };
  auto piece = [&](int i, int voff, unsigned buf) {        // piece i = (panel i / PIECES, row group i % PIECES)
    const int p = i / PIECES, slot = i % PIECES;
    if (p < N_PANELS) {
      const unsigned lbase = (p ? 2u * PANEL_B : 0u) + buf + (unsigned)(RPP * wave + 4 * RPP * slot) * ROW_B;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds + lbase), 16, voff,
                                               soff[p] + slot * 4 * RPP * pitch * 4 + (slot * 4 * RPP >= 64 ? hadj[p] : 0), 0, 0);
    }
  };
  // ---- the reader's view of the stream: at step j this lane half multiplies group Gr = 4 t + 2 j + h, which is group
  // rr[j] = Gr mod gps of its row.  Operand values that are not part of the sum - pixels behind the end of a row (W not a
  // multiple of 4, phantom groups: what the DMA fetched there belongs to the next row) and whole groups behind the K range
  // (LDS keeps what an earlier stage left there) - are zeroed in the operand registers, in a block that only the stages
  // holding such a group enter (ru = 4 t mod gps, scalar: the stage's first group).
  const bool maskable = (W & 3) != 0 || gps != ((W + 3) >> 2);
  const int g_full = W >> 2;                               // groups of a row whose four pixels all exist
  int rr[STEPS];
#pragma unroll
  for (int j = 0; j < STEPS; ++j) rr[j] = (SLOTS * t0 + 2 * j + h) % gps;
  int ru = (SLOTS * t0) % gps;
  const int t_end = (g_tot - 1) / SLOTS;                   // the stage that holds the stream's last group
  const bool short_end = (g_tot & (SLOTS - 1)) != 0;

  {
    const int v0 = lane_voff(true);
#pragma unroll
    for (int i = 0; i < NP; ++i) piece(i, v0, (unsigned)(t0 & 1) * PANEL_B);
    advance();
  }
  for (int t = t0; t < t1; ++t) {
    __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0): this wave's DMA of stage t has landed
    __syncthreads();                           // everyone's has; everyone is done reading the other buffer
    const int voff_n = lane_voff(t + 1 < t1);  // stage t + 1 (all lanes out of range behind the item's last stage)
    advance();
    const unsigned buf = (unsigned)(t & 1) * PANEL_B, nbuf = PANEL_B - buf;
    auto rd = [&](int o, int j) { return *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(lds + addr[o][j] + buf); };
    // straight line: per step four operand reads and 16 MFMAs, the four pieces of stage t + 1 behind the first step's MFMA
    // groups (the last piece has more than half of the stage's MFMA time to land before the wait at the top).  A step
    // whose lane halves hold a row end (or lie behind the stream's end) zeroes those operand values first - a VALU-only
    // block per step, so that the second step's reads still issue under the first step's MFMAs (one block for the whole
    // stage put all eight reads in front of the first MFMA: DenseNet-121 update() 7.6 -> 8.2 ms)
    const bool tail_stage = (maskable && ru + SLOTS > g_full) || (short_end && t == t_end);
    auto mask_step = [&](int j, f32x4& xa0, f32x4& xa1, f32x4& xb0, f32x4& xb1) {
      if (tail_stage) {
        asm volatile("; stream tail" ::: "memory");        // keeps this a branch around a VALU-only block
        const bool gone = SLOTS * t + 2 * j + h >= g_tot;  // the whole group lies behind the K range
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool k = 4 * rr[j] + e >= W || gone;
          xa0[e] = k ? 0.0f : xa0[e]; xa1[e] = k ? 0.0f : xa1[e];
          // the B side only where LDS may hold anything (behind the stream's end nothing was fetched); behind a row's end
          // it holds the next row's pixels, whose products with the A side's zeros vanish
          xb0[e] = gone ? 0.0f : xb0[e]; xb1[e] = gone ? 0.0f : xb1[e];
        }
      }
    };
    f32x4 a0 = rd(0, 0), a1 = rd(1, 0), b0 = rd(2, 0), b1 = rd(3, 0);
    mask_step(0, a0, a1, b0, b1);
    f32x4 na0 = rd(0, 1), na1 = rd(1, 1), nb0 = rd(2, 1), nb1 = rd(3, 1);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (PART != 3) c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], c00, 0, 0, 0);
      if (PART != 2) c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b1[e], c01, 0, 0, 0);
      if (PART == 0 || PART == 2) c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b0[e], c10, 0, 0, 0);
      if (PART != 2) c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], c11, 0, 0, 0);
      piece(e, voff_n, nbuf);                  // one LDS-DMA piece behind a group of MFMAs: its issue cost hides under them
    }
    mask_step(1, na0, na1, nb0, nb1);
    flat_mfma_step<PART>(na0, na1, nb0, nb1, c00, c01, c10, c11);
    ru += SLOTS;
    ru -= ru >= gps ? gps : 0;
#pragma unroll
    for (int j = 0; j < STEPS; ++j) {
      rr[j] += SLOTS;
      rr[j] -= rr[j] >= gps ? gps : 0;
    }
    if (t + 1 == seg_end) {
      // end of a segment of an unsliced item: scale, add into the factor, (last segment) write the mirror tile; the
      // DMA of the next stage is in flight meanwhile
      direct_store_quadrant(d, PART, i0 + 64 * wm, j0 + 64 * wn, r32, h, c00, c01, c10, c11, seg == 0, t + 1 == t1);
      c00 = 0.0f; c01 = 0.0f; c10 = 0.0f; c11 = 0.0f;
      ++seg;
      seg_end = min(seg_end + d.cpi, t1);
    }
  }
  if (nseg) return;
  gfloat_t* slab = (gfloat_t*)slabs + d.slab_base + (long long)local * (TM * TM);
  gfloat_t* q = slab + (64 * wm) * 128 + 64 * wn;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
    if (PART != 3) q[row * 128 + r32] = c00[reg];
    if (PART != 2) q[row * 128 + 32 + r32] = c01[reg];
    if (PART != 3) q[(32 + row) * 128 + r32] = c10[reg];
    if (PART != 2) q[(32 + row) * 128 + 32 + r32] = c11[reg];
  }
}

__global__ void __launch_bounds__(SYRK_THREADS, flat::WGS)
syrk_flat_kernel(const FactorDev* __restrict__ descs, int n_factors, int n_items, float* __restrict__ slabs) {
  __shared__ __attribute__((aligned(1024))) char smem[flat::LDS_B];
  const int item = xcd_item(blockIdx.x);
  if (item >= n_items) return;
  int f = find_segment(descs, n_factors, item, false);
  int local = item - descs[f].item_base;
  if (descs[f].group_n > 0) {                  // a group's shared range: item -> (k-slice, member, tile)
    const int head = f - descs[f].group_pos;
    const int gi = item - descs[head].item_base, nt = descs[head].n_tiles, gt = descs[head].group_n * nt;
    const int slice = gi / gt, rem = gi - slice * gt, member = rem / nt;
    f = head + member;
    local = slice * nt + (rem - member * nt);
  }
  const FactorDev& d = descs[f];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = local % d.n_tiles;
  int ti, tj;
  decode_tile_of(d, tile, ti, tj);
  int part = 0;
  if (ti == tj && !d.nonsym) {
    const int wm = wave >> 1, wn = wave & 1;
    part = (wm == wn) ? 1 : (wm == 0 ? 2 : 3);
  }
  part = __builtin_amdgcn_readfirstlane(part);
  lds_char* l3 = (lds_char*)smem;
  if (part == 0) flat_body<0>(d, local, slabs, l3);
  else if (part == 1) flat_body<1>(d, local, slabs, l3);
  else if (part == 2) flat_body<2>(d, local, slabs, l3);
  else flat_body<3>(d, local, slabs, l3);
}

// ------------------------------------------------------------------------------------------------
// unfolded copies of the sources of `sub` factors (curvature/curvatures.py:329, F.unfold: rows (c, kh, kw), columns (oh, ow)):
//   out[n][(c, i, j)][oh][ow] = src[n][c][oh sh + i - ph][ow sw + j - pw]   (0 outside the image)
// ------------------------------------------------------------------------------------------------
struct SubDev {
  const float* src;
  float* out;
  int C, H, W, kh, kw, sh, sw, ph, pw, Ho, Wo;
  int wg_base;           // first workgroup of this factor in the grid
  unsigned words;        // N * C kh kw * Ho * Wo (< 2^29: syrk_flat_eligible)
  unsigned m_wo, m_ho, m_rows, m_kk, m_kw;     // ceil(2^32 / d) for Wo, Ho, C kh kw, kh kw, kw
};
// x / d for x < 2^31 through the multiplier ceil(2^32 / d): the product's high word is the quotient or one above it
__device__ __forceinline__ unsigned sub_div(unsigned x, unsigned d, unsigned magic) {
  if (d == 1) return x;
  unsigned q = __umulhi(x, magic);
  return q * d > x ? q - 1 : q;
}
static unsigned sub_magic(int d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); }
constexpr int SUB_CHUNK = 16;
constexpr int SUB_SEG = 2048;  // output words per workgroup: 8 per thread, stores coalesced
struct SubChunk { SubDev f[SUB_CHUNK]; };
static_assert(sizeof(SubChunk) <= 3840, "kernel argument block must stay below 4 KB");

__global__ void __launch_bounds__(256) unfold_prep_kernel(SubChunk chunk, int count) {
  int l = 0;
  while (l + 1 < count && chunk.f[l + 1].wg_base <= (int)blockIdx.x) ++l;
  const SubDev& d = chunk.f[l];
  const unsigned Wo = d.Wo, Ho = d.Ho, kk = d.kh * d.kw, rows = d.C * kk;
  const unsigned e0 = (blockIdx.x - d.wg_base) * SUB_SEG;
  const gfloat_t* src = (const gfloat_t*)d.src;
  gfloat_t* out = (gfloat_t*)d.out;
  // eight elements per thread, all eight loads issued before the first store (the nine rows of a channel read the same
  // input lines: the source is fetched from HBM once); 32-bit index arithmetic with multiply-high divisions (the first
  // form, with five 64-bit divisions per element, ran at 1.2 TB/s)
  constexpr int U = SUB_SEG / 256;
  float v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const unsigned e = e0 + threadIdx.x + 256 * u;
    v[u] = 0.0f;
    if (e < d.words) {
      const unsigned r = sub_div(e, Wo, d.m_wo), ow = e - r * Wo;           // r: (sample, row of X, oh)
      const unsigned q = sub_div(r, Ho, d.m_ho), oh = r - q * Ho;           // q: (sample, row of X)
      const unsigned n = sub_div(q, rows, d.m_rows), row = q - n * rows;
      const unsigned c = sub_div(row, kk, d.m_kk), ij = row - c * kk;
      const unsigned i = sub_div(ij, d.kw, d.m_kw), j = ij - i * d.kw;
      const int ih = (int)(oh * d.sh + i) - d.ph, iw = (int)(ow * d.sw + j) - d.pw;
      if (ih >= 0 && ih < d.H && iw >= 0 && iw < d.W) v[u] = src[((long long)(n * d.C + c) * d.H + ih) * d.W + iw];
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const unsigned e = e0 + threadIdx.x + 256 * u;
    if (e < d.words) out[e] = v[u];
  }
}

int launch_sub_prep(hipStream_t stream, const std::vector<FactorDev>& f, int n_user, const curv_factor_desc* descs, float* area) {
  std::vector<int> which;
  for (int i = 0; i < n_user; ++i) if (f[i].sub) which.push_back(i);
  for (size_t b = 0; b < which.size(); b += SUB_CHUNK) {
    SubChunk chunk;
    memset(&chunk, 0, sizeof(chunk));
    const int count = (int)std::min<size_t>(SUB_CHUNK, which.size() - b);
    long long wgs = 0;
    for (int k = 0; k < count; ++k) {
      const int i = which[b + k];
      const curv_factor_desc& s = descs[i];
      SubDev& d = chunk.f[k];
      d.src = s.src;
      d.out = area + f[i].xq_off;
      d.C = s.C; d.H = s.H; d.W = s.W; d.kh = s.kh; d.kw = s.kw; d.sh = s.sh; d.sw = s.sw; d.ph = s.ph; d.pw = s.pw;
      d.Ho = (s.H + 2 * s.ph - s.kh) / s.sh + 1; d.Wo = (s.W + 2 * s.pw - s.kw) / s.sw + 1;
      const long long words = (long long)s.N * s.C * s.kh * s.kw * d.Ho * d.Wo;
      CURV_REQUIRE(words < (1LL << 31), "curv_kfac: unfolded source too large");
      d.words = (unsigned)words;
      d.m_wo = sub_magic(d.Wo); d.m_ho = sub_magic(d.Ho); d.m_rows = sub_magic(s.C * s.kh * s.kw);
      d.m_kk = sub_magic(s.kh * s.kw); d.m_kw = sub_magic(s.kw);
      d.wg_base = (int)wgs;
      wgs += cdivll(words, SUB_SEG);
      CURV_REQUIRE(wgs < (1LL << 31), "curv_kfac: unfolded sources too large for one pass");
    }
    hipLaunchKernelGGL(unfold_prep_kernel, dim3((unsigned)wgs), dim3(256), 0, stream, chunk, count);
    CURV_LAUNCH_CHECK();
  }
  return CURV_OK;
}

int launch_syrk_flat(hipStream_t stream, const FactorDev* table, int n_factors, int n_items, float* slabs) {
  if (n_items <= 0) return CURV_OK;
  const int grid = cdiv(n_items, 8 * XCD_GROUP) * 8 * XCD_GROUP;
  hipLaunchKernelGGL(syrk_flat_kernel, dim3(grid), dim3(SYRK_THREADS), 0, stream, table, n_factors, n_items, slabs);
  CURV_LAUNCH_CHECK();
  return CURV_OK;
}

}  // namespace curv
