// Shared between the two factor-build kernels (syrk.hip: implicit-im2col patch kernel; syrk_flat.hip: LDS-DMA kernel
// for flattened per-pixel factors): the device descriptor of one Kronecker factor and the work-list helpers.
#pragma once
#include <vector>

#include "common.h"

namespace curv {

constexpr int SYRK_THREADS = 256;
constexpr int XCD_GROUP = 32;          // consecutive items that share an XCD

struct FactorDev {
  const float* src;
  float* dst;
  int N, C, H, W;          // source geometry (1x1/stride-1 convs arrive flattened to H = 1)
  int kh, kw, sh, sw, ph, pw;
  int Ho, Wo;
  int khkw;
  int rows, dim, has_bias;
  int compact;             // kh == kw == 1: patch holds only the sampled pixels
  int vec4;                // flattened per-pixel factor with 16-B aligned rows: float4 staging
  int TM;                  // tile edge: 64 or 128
  int NS, R, Wc;           // chunk extent: samples, output rows, output cols
  int n_rg, n_cg;          // chunk grid (rows, cols); samples outermost
  int n_chunks;
  int RS, PS, SS, nch;     // LDS strides in words, channels per panel
  int cshift;              // log2 of the padded patch row length (lanes along x)
  int direct;              // > 0: unsliced 128x128 items whose epilogue scales, adds into dst and writes the mirror tile
                           // itself (direct_store_block below): no slab, no sub-tiles in the reduce pass.  The value
                           // is the number of SERIAL segments (of cpi chunks each) the item cuts its K range into:
                           // the accumulators are flushed into dst behind every segment, which bounds the length of
                           // an fp32 accumulation chain (see MAX_CHAIN_PX in syrk.hip)
  int P, n_tiles;
  int cpi, n_slices;       // chunks per item, k-slices
  int item_base, n_items;
  int sub_base, n_sub;     // 64x64 sub-tiles for the reduce kernel
  int first;
  float scale;
  int pad0;
  int rshift;              // general staging: a lane group of 2^rshift folded rows; the other row lanes split channels
  int flat;                // flattened per-pixel factor whose (sample, channel) rows are staged slot-regularly
  unsigned rmagic;         // ceil(2^32 / patch rows per sample): folded (sample, row) index -> sample
  int pre;                 // full-width chunks of a kh x kw > 1 convolution: the patch images are staged by LDS-DMA from a
                           // pre-tiled, zero-padded copy of the source (syrk_pre.hip), one contiguous run of SS words per
                           // (chunk, sample, panel); `src` of the device table entry points at that copy
  int ppr, tail_lanes;     // ... 1 KiB DMA pieces per run, and the lanes of the last piece that lie inside the run
  int dma;                 // 1: built by syrk_flat_kernel (n_chunks = stages, cpi = stages per item);
                           // 2: a 3x3 factor assembled from shifted correlations (syrk_corr.hip): no work items of its own
  // syrk_flat_kernel operands: row r of sample s starts at src + (s * C + r) * pitch floats; panel i reads
  // [off_i, off_i + W) of its rows, panel j [off_j, off_j + W) of its rows.  Plain flattened factors: pitch = W,
  // offsets 0.  nonsym: a correlation X_i X_j^T between differently shifted rows - every (ti, tj) tile is computed
  // and nothing is mirrored.
  int pitch, off_i, off_j, nonsym;
  // half: a PACKED pair tile for 64-channel sources (syrk_corr.hip): the 128 rows of a panel are the source's 64
  // channel rows twice, rows 64 .. 127 at the second offset (off_i2 / off_j2), so the tile's four 64x64 blocks are four
  // different shifted correlations of one 64-channel image
  int half, off_i2, off_j2, pad1;
  // group: group_n consecutive table entries (this one is number group_pos) with equal n_tiles / cpi / n_slices share
  // one item range, enumerated (k-slice, member, tile): the workgroups that stream the same slice of one source - the
  // shifted correlations of a layer - are then neighbours in launch order and meet in an XCD's L2.  The range starts
  // at the first member's item_base; member j carries item_base + j only to keep the table's bases ascending.
  int group_n, group_pos;
  // sub: a strided 1x1 or a kh x kw > 1 convolution whose unfolded matrix X (C kh kw rows of Ho Wo pixels per sample) is
  // written out once (unfold_prep_kernel, into the workspace area at xq_off) and then built by the LDS-DMA kernel as a
  // flattened factor: `src` of the device table entry points at the copy, N / C (= rows) / W (= Ho Wo) describe it
  int sub, pad2;
  long long slab_base;     // in floats
  long long xq_off;        // pre: offset of the pre-tiled copy in the workspace area, in floats (host side only)
};
static_assert(sizeof(FactorDev) % 8 == 0, "FactorDev must be 8-byte granular");

__device__ __forceinline__ int find_segment(const FactorDev* __restrict__ descs, int n_factors, int id,
                                            bool by_sub) {
  // largest f with base[f] <= id; bases are ascending.  One ballot per 64 factors.
  const int lane = threadIdx.x & 63;
  int count = 0;
  for (int f0 = 0; f0 < n_factors; f0 += 64) {
    const int f = f0 + lane;
    bool le = false;
    if (f < n_factors) le = (by_sub ? descs[f].sub_base : descs[f].item_base) <= id;
    count += __popcll(__ballot(le));
  }
  return __builtin_amdgcn_readfirstlane(count - 1);
}

// Tile t of a factor's upper triangle (ti <= tj) of P x P tiles.  Order: blocks of TB_H x TB_W tiles, block row by block
// row, row-major inside a block - so that XCD_GROUP = 32 consecutive work items of one k-slice (one XCD, launched together)
// stream TB_H + TB_W = 12 different operand panels instead of the ~33 of a plain row-major walk over a wide triangle: an
// XCD's L2 is filled with every panel once per block, not once per tile row (round 6; factors of at most 8 tile rows keep
// their old order).  Scalar loops over at most P^2 / 32 blocks per work item.
constexpr int TB_H = 4, TB_W = 8;
__device__ __forceinline__ void decode_tile(int t, int P, int& ti, int& tj) {
  for (int r0 = 0; r0 < P; r0 += TB_H) {
    const int r1 = min(r0 + TB_H, P);
    for (int c0 = (r0 / TB_W) * TB_W; c0 < P; c0 += TB_W) {
      const int c1 = min(c0 + TB_W, P);
      int cnt = 0;                                   // tiles of the block on or above the diagonal
      for (int r = r0; r < r1; ++r) cnt += max(0, c1 - max(c0, r));
      if (t < cnt) {
        for (int r = r0; r < r1; ++r) {
          const int lo = max(c0, r), w = max(0, c1 - lo);
          if (t < w) { ti = r; tj = lo + t; return; }
          t -= w;
        }
      }
      t -= cnt;
    }
  }
  ti = tj = 0;                                       // (t < P (P + 1) / 2: not reached)
}
// ... and of a full P x P grid (a correlation between differently shifted rows), in the same blocks
__device__ __forceinline__ void decode_tile_full(int t, int P, int& ti, int& tj) {
  const int ncb = (P + TB_W - 1) / TB_W;
  for (int r0 = 0; r0 < P; r0 += TB_H) {
    const int h = min(TB_H, P - r0), row_tiles = h * P;
    if (t < row_tiles) {
      for (int cb = 0; cb < ncb; ++cb) {
        const int c0 = cb * TB_W, w = min(TB_W, P - c0), cnt = h * w;
        if (t < cnt) { ti = r0 + t / w; tj = c0 + t % w; return; }
        t -= cnt;
      }
    }
    t -= row_tiles;
  }
  ti = tj = 0;
}


// XCD-aware item order: workgroups that share an XCD (equal blockIdx % 8) take every 8th group of XCD_GROUP
// consecutive items, i.e. neighbouring tiles of one k-slice of one factor, so the panels they stage hit that XCD's
// L2, while every XCD still sees an even mix of all factors.
__device__ __forceinline__ int xcd_item(int bid) {
  const int xcd = bid & 7, j = bid >> 3;
  return ((j / XCD_GROUP) * 8 + xcd) * XCD_GROUP + (j % XCD_GROUP);
}

// Direct epilogue of an unsliced work item (FactorDev::direct): one 32x32 MFMA block of the wave's quadrant, whose
// lane (r32, h) holds acc[reg] = element (row(reg, h), r32) with row = (reg & 3) + 8 (reg >> 2) + 4 h, goes straight to
//   dst[gi][gj] (+)= scale * acc      and, for a symmetric factor, the same VALUE to dst[gj][gi]
// (exactly symmetric by construction; the old value is read from the upper position only).  Only one work item touches
// a tile and its mirror image, so there is nothing to order.  A block on the factor's diagonal writes its upper
// triangle and mirrors that (both triangles of an MFMA block are computed, only one is used - as the reduce pass does).
// Same arithmetic as syrk_reduce_kernel with one slice: (acc * scale) rounded, then dst + that.
// All accesses are raw buffer operations on a descriptor of the factor: address = base + scalar offset (the row of a
// register, SALU) + per-lane offset (ONE vector register for the upper position, one for the mirror), so the epilogue
// needs no 64-bit per-register addresses (an earlier form with pointers spilled 250-800 bytes per lane inside the
// MFMA kernels); masked elements (ragged edge, wrong triangle of a diagonal block) carry an out-of-range lane offset
// and are dropped by the hardware.  Upper stores: 2 rows x 32 consecutive floats per instruction.  Mirror stores: four
// consecutive rows of a lane are four consecutive floats of a mirror row -> 16-byte stores when the factor's rows are
// 16-byte granular (the four stores of a block complete 128-byte lines in L2).  Eight registers are in flight at a time.
struct DirectDst {
  __amdgpu_buffer_rsrc_t rs;
  int dim;
  float scale;
  bool vec_ok;
};
__device__ __forceinline__ DirectDst direct_dst(const FactorDev& d) {
  DirectDst t;
  t.dim = d.dim;
  t.scale = d.scale;
  // the planner keeps dim^2 * 4 < 2^31 for direct factors: every byte offset fits 31 bits, 0x80000000 is out of range
  t.rs = __builtin_amdgcn_make_buffer_rsrc((void*)d.dst, 0, (unsigned)d.dim * (unsigned)d.dim * 4u, 0x00020000);
  t.vec_ok = (((unsigned)d.dim & 3u) | (unsigned)(reinterpret_cast<uintptr_t>(d.dst) & 15)) == 0;
  return t;
}

__device__ __forceinline__ void direct_store_block(const DirectDst& t, const f32x16& acc, int gi0, int gj0, int r32,
                                                   int h, bool diag_block, bool first, bool mirror) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  constexpr int OOB = (int)0x80000000;
  const int dim = t.dim;
  const float scale = t.scale;
  const int gj = gj0 + r32;
  const bool plain = gi0 + 32 <= dim && gj0 + 32 <= dim && !diag_block;     // no element of the block is masked
  const int vu = (gj < dim) ? ((4 * h) * dim + gj) * 4 : OOB;                // + row offset (scalar)
  const int vm = (gj < dim) ? (gj * dim + 4 * h) * 4 : OOB;                  // + column offset (scalar)
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    float out[8];
    int vo[8], so[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int row0 = (k & 3) + 8 * (2 * b + (k >> 2));           // register 8 b + k: row = row0 + 4 h
      so[k] = (gi0 + row0) * dim * 4;
      vo[k] = vu;
      if (!plain) {
        const int row = row0 + 4 * h;
        if (gi0 + row >= dim || (diag_block && row > r32)) vo[k] = OOB;
      }
      out[k] = first ? 0.0f : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(t.rs, vo[k], so[k], 0));
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float v = __fmul_rn(acc[8 * b + k], scale);            // no fma: the reduce pass rounds the product too
      out[k] = first ? v : __fadd_rn(out[k], v);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, out[k]), t.rs, vo[k], so[k], 0);
    }
    if (mirror) {
      if (plain && t.vec_ok) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const u32x4 v = {__builtin_bit_cast(unsigned, out[4 * g]), __builtin_bit_cast(unsigned, out[4 * g + 1]),
                           __builtin_bit_cast(unsigned, out[4 * g + 2]), __builtin_bit_cast(unsigned, out[4 * g + 3])};
          __builtin_amdgcn_raw_buffer_store_b128(v, t.rs, vm, (gi0 + 8 * (2 * b + g)) * 4, 0);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int row0 = (k & 3) + 8 * (2 * b + (k >> 2));
          int v = vm;
          if (!plain) {
            const int row = row0 + 4 * h;
            if (gi0 + row >= dim || (diag_block && row >= r32)) v = OOB;
          }
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, out[k]), t.rs, v, (gi0 + row0) * 4, 0);
        }
      }
    }
  }
}

// the wave's 64x64 quadrant at (qi0, qj0) of a 128x128 tile: parts as in syrk.hip / syrk_flat.hip
//   0: all four blocks   1: diagonal quadrant (upper three blocks)   2: left block column   3: right block column
// first_seg / last_seg: position of the flushed segment in the item's serial segments: the first one honours the
// factor's `first` flag (overwrite), the others add; only the last one - whose upper tile then holds the final values -
// writes the mirror
__device__ __forceinline__ void direct_store_quadrant(const FactorDev& d, int part, int qi0, int qj0, int r32, int h,
                                                      const f32x16& c00, const f32x16& c01, const f32x16& c10,
                                                      const f32x16& c11, bool first_seg, bool last_seg) {
  // the flush sits inside the K loop and everything it computes is loop-invariant: without these barriers the
  // compiler hoists the per-register lane offsets and masks out of the loop and spills them (100-250 bytes per lane)
  asm volatile("" : "+v"(r32), "+v"(h));
  asm volatile("" : "+s"(qi0), "+s"(qj0));
  const bool first = d.first != 0 && first_seg;
  const bool mirror = !d.nonsym && last_seg;
  const bool dq = part == 1;
  const DirectDst t = direct_dst(d);
  if (part != 3) direct_store_block(t, c00, qi0, qj0, r32, h, dq, first, mirror);
  if (part != 2) direct_store_block(t, c01, qi0, qj0 + 32, r32, h, false, first, mirror);
  if (part == 0 || part == 2) direct_store_block(t, c10, qi0 + 32, qj0, r32, h, false, first, mirror);
  if (part != 2) direct_store_block(t, c11, qi0 + 32, qj0 + 32, r32, h, dq, first, mirror);
}

// syrk_small.hip: the two-launch build of a small launch (LeNet scale).  kfac_small_workspace_bytes: 0 if the launch is
// not a small one; kfac_accumulate_small: CURV_ERR_WORKSPACE (no error text) if it is not, or if the workspace does not
// hold its slabs - the caller then takes the grouped path.
size_t kfac_small_workspace_bytes(const curv_factor_desc* descs, int n);
int kfac_path_for(const curv_factor_desc* descs, int n);
int kfac_accumulate_small(hipStream_t stream, const curv_factor_desc* descs, int n, void* workspace, size_t workspace_bytes,
                          void* ev_start, void* ev_stop);

// invert.hip: one of the library's internal streams (per thread and device), lent to the factor build's side work
int curv_internal_side_stream(hipStream_t* out);

// syrk_flat.hip: host launcher of the LDS-DMA kernel over a device table of dma factors
int launch_syrk_flat(hipStream_t stream, const FactorDev* table, int n_factors, int n_items, float* slabs);
// the unfolded copies of `sub` factors (one pass in front of the LDS-DMA kernel): descs = the caller's descriptors
int launch_sub_prep(hipStream_t stream, const std::vector<FactorDev>& f, int n_user, const curv_factor_desc* descs, float* area);
// host-side eligibility / stage count of the LDS-DMA kernel
bool syrk_flat_eligible(const FactorDev& f, const void* src);
int syrk_flat_chunks(int N, int W);     // stages of 16 pixels over the factor's stream of 4-pixel groups

__device__ __forceinline__ void decode_tile_of(const FactorDev& d, int t, int& ti, int& tj) {
  if (d.nonsym) decode_tile_full(t, d.P, ti, tj);
  else decode_tile(t, d.P, ti, tj);
}

// syrk_pre.hip: pre-tiled copies of the sources of `pre` factors (one pass in front of the patch kernel)
long long syrk_pre_floats(const FactorDev& f);
int launch_patch_prep(hipStream_t stream, const std::vector<FactorDev>& f, const std::vector<int>& which, float* area);

// syrk_corr.hip: 3x3 / stride 1 / padding 1 factors assembled from shifted correlations
constexpr int CORR_COMPONENTS = 29;
struct CorrLayer {
  int user;                  // index of the user factor in Plan::f
  int vf0, n_vf;             // its virtual factors are Plan::f[vf0 .. vf0 + n_vf): the 29 components one by one, or
                             // (64 channels) 10 packed pair tiles that hold them as 64x64 blocks
  int xp_pitch;              // floats between consecutive (sample, channel) rows of the padded copy
  int comp_pitch;            // row pitch of a component matrix, and where each of the 29 starts (floats from comp_off)
  int comp_at[CORR_COMPONENTS];
  int N, C, H, W, Wp, Hq;
  int row_pitch, col_pitch, pt_pitch;
  long long xp_off, rowb_off, rowt_off, colr_off, coll_off, pt_off, comp_off;    // floats from the area base
};
bool syrk_corr_eligible(const curv_factor_desc& s);
void syrk_corr_expand(const curv_factor_desc& s, int user, std::vector<FactorDev>& f, CorrLayer& layer,
                      long long& area_floats);
void syrk_corr_bind(const CorrLayer& layer, std::vector<FactorDev>& f, float* area);
int launch_corr_prep(hipStream_t stream, const std::vector<CorrLayer>& layers, const std::vector<FactorDev>& f,
                     float* area);
int launch_corr_assemble(hipStream_t stream, const std::vector<CorrLayer>& layers, const std::vector<FactorDev>& f,
                         float* area);

}  // namespace curv
