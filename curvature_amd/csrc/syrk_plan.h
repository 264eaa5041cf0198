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
  int RL;                  // (reserved; the k loop no longer works with table-driven runs)
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
  // group: group_n consecutive table entries (this one is number group_pos) with equal n_tiles / cpi / n_slices share
  // one item range, enumerated (k-slice, member, tile): the workgroups that stream the same slice of one source - the
  // shifted correlations of a layer - are then neighbours in launch order and meet in an XCD's L2.  The range starts
  // at the first member's item_base; member j carries item_base + j only to keep the table's bases ascending.
  int group_n, group_pos;
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

__device__ __forceinline__ void decode_tile(int t, int P, int& ti, int& tj) {
  ti = 0;
  while (t >= P - ti) { t -= P - ti; ++ti; }
  tj = ti + t;
}


// XCD-aware item order: workgroups that share an XCD (equal blockIdx % 8) take every 8th group of XCD_GROUP
// consecutive items, i.e. neighbouring tiles of one k-slice of one factor, so the panels they stage hit that XCD's
// L2, while every XCD still sees an even mix of all factors.
__device__ __forceinline__ int xcd_item(int bid) {
  const int xcd = bid & 7, j = bid >> 3;
  return ((j / XCD_GROUP) * 8 + xcd) * XCD_GROUP + (j % XCD_GROUP);
}

// invert.hip: one of the library's internal streams (per thread and device), lent to the factor build's side work
int curv_internal_side_stream(hipStream_t* out);

// syrk_flat.hip: host launcher of the LDS-DMA kernel over a device table of dma factors
int launch_syrk_flat(hipStream_t stream, const FactorDev* table, int n_factors, int n_items, float* slabs);
// host-side eligibility / stage count of the LDS-DMA kernel
bool syrk_flat_eligible(const FactorDev& f, const void* src);
int syrk_flat_stages(int HW);

__device__ __forceinline__ void decode_tile_of(const FactorDev& d, int t, int& ti, int& tj) {
  if (d.nonsym) { ti = t / d.P; tj = t - ti * d.P; }
  else decode_tile(t, d.P, ti, tj);
}

// syrk_pre.hip: pre-tiled copies of the sources of `pre` factors (one pass in front of the patch kernel)
long long syrk_pre_floats(const FactorDev& f);
int launch_patch_prep(hipStream_t stream, const std::vector<FactorDev>& f, const std::vector<int>& which, float* area);

// syrk_corr.hip: 3x3 / stride 1 / padding 1 factors assembled from shifted correlations
constexpr int CORR_COMPONENTS = 29;
struct CorrLayer {
  int user;                  // index of the user factor in Plan::f
  int vf0;                   // its CORR_COMPONENTS virtual factors are Plan::f[vf0 ...]
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
