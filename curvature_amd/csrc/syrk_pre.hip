// Pre-tiled source copies for the implicit-im2col factor build (syrk.hip) on gfx950.
//
// A convolution with kh x kw > 1 is built from LDS patch images: per (chunk = sample group x output-row group,
// sample, channel) one plane of `rows_in` input rows at row pitch RS (the halo columns and the rows outside the
// image are zeros; RS and the plane stride PS are chosen by the planner for conflict-free operand gathers), the
// channels of a panel side by side at stride PS (curvature/curvatures.py:329-335: this is what F.unfold reads,
// before it is unfolded).  Gathering such an image from the (N, C, H, W) tensor inside the SYRK kernel cost ~11 k
// wave-cycles of address arithmetic, loads and LDS stores per chunk next to ~23 k cycles of MFMA work.  This pass
// writes the source once in image order instead,
//     Xq[chunk][sample of the chunk][channel][PS]          (chunk = sample group * n_rg + row group),
// so that the image of ANY panel (channels c_lo .. c_lo + nch of one sample) is one contiguous run of nch * PS floats
// which the kernel moves with buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction.  Rows shared by neighbouring
// row groups are stored twice ((R - 1) sh + kh rows per R output rows); the pass is one read of the source and one
// write of 1.2-1.6 x its size, HBM-bound.
#include <algorithm>
#include <cstring>
#include <vector>

#include "syrk_plan.h"

namespace curv {

struct PreDev {
  const float* src;
  float* xq;
  int N, C, H, W;
  int NS, R, n_rg;
  int sh, ph, pw;
  int rows_in, RS, PS;
  unsigned rs_magic;          // ceil(2^32 / RS): w / RS for w < 2^16
  unsigned ps_magic;          // ceil(2^32 / PS): r / PS for r < 2^16
  unsigned c_magic, ns_magic, rg_magic;   // ceil(2^32 / d) for C, NS, n_rg (quotients of numbers whose product with d < 2^32)
  int n_planes;               // chunks * NS * C
  int seg_base;               // first workgroup (= SEG-word segment of the copy) of this factor in the grid
  int pad;
  long long words;            // n_planes * PS
};
constexpr int PRE_CHUNK = 32;
constexpr int SEG = 4096;     // output words per workgroup: 16 per thread, coalesced
struct PreChunk { PreDev f[PRE_CHUNK]; };
static_assert(sizeof(PreChunk) <= 3840, "kernel argument block must stay below 4 KB");

typedef __attribute__((address_space(1))) float gfl_t;

__device__ __forceinline__ int divu(int x, int d, unsigned magic) { return d == 1 ? x : (int)__umulhi((unsigned)x, magic); }
static unsigned magic_of(int d) { return (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); }

// one workgroup per SEG consecutive words of a factor's copy: the plane of the segment's first word comes from one
// exact division per workgroup, everything after it from 32-bit multiply-high arithmetic on small offsets
__global__ void __launch_bounds__(256) patch_prep_kernel(PreChunk chunk, int count) {
  int l = 0;
  while (l + 1 < count && chunk.f[l + 1].seg_base <= (int)blockIdx.x) ++l;
  const PreDev& d = chunk.f[l];
  const long long w0 = (long long)(blockIdx.x - d.seg_base) * SEG;
  const int PS = d.PS, RS = d.RS, C = d.C, H = d.H, W = d.W, NS = d.NS, n_rg = d.n_rg;
  const int plane0 = (int)(w0 / PS);
  const int r0 = (int)(w0 - (long long)plane0 * PS);
  gfl_t* out = (gfl_t*)d.xq + w0;
  const int n = (int)min((long long)SEG, d.words - w0);
  // four words per thread and pass, the four loads issued before the first store (one word per pass left 1 KB per
  // workgroup in flight)
#ifndef CURV_PREP_U
#define CURV_PREP_U 4
#endif
  constexpr int U = CURV_PREP_U;
  for (int t0 = threadIdx.x; t0 < n; t0 += 256 * U) {
    float val[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int r = r0 + min(t0 + 256 * k, n - 1);           // < SEG + PS (passes beyond n repeat its last word)
      const int dp = (int)__umulhi((unsigned)r, d.ps_magic);
      const int w = r - dp * PS;
      const int plane = plane0 + dp;
      const int cs = divu(plane, C, d.c_magic);             // chunk * NS + sample of the chunk
      const int c = plane - cs * C;
      const int ch = divu(cs, NS, d.ns_magic), s_in = cs - ch * NS;
      const int sg = divu(ch, n_rg, d.rg_magic), rg = ch - sg * n_rg;
      const int s_ = sg * NS + s_in;
      const int y = divu(w, RS, d.rs_magic);
      const int x = w - y * RS;
      const int ih = rg * d.R * d.sh - d.ph + y, iw = x - d.pw;
      float v = 0.0f;
      if (s_ < d.N && y < d.rows_in && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
        v = d.src[(((long long)s_ * C + c) * H + ih) * W + iw];
      val[k] = v;
    }
#pragma unroll
    for (int k = 0; k < U; ++k)
      if (t0 + 256 * k < n) out[t0 + 256 * k] = val[k];
  }
  // the tail behind the last plane: a panel whose channel range ends past C reads up to nch planes (+ one DMA lane)
  // beyond it; those values are never MFMA operands of a live row, but must be finite
  if (w0 + SEG >= d.words) {
    gfl_t* tail = (gfl_t*)d.xq + d.words;
    for (int t = threadIdx.x; t < d.pad; t += 256) tail[t] = 0.0f;
  }
}

long long syrk_pre_floats(const FactorDev& f) {
  return ((long long)f.n_chunks * f.NS * f.C + f.nch) * f.PS + 64;
}

int launch_patch_prep(hipStream_t stream, const std::vector<FactorDev>& f, const std::vector<int>& which, float* area) {
  for (size_t b = 0; b < which.size(); b += PRE_CHUNK) {
    PreChunk chunk;
    memset(&chunk, 0, sizeof(chunk));
    const int count = (int)std::min<size_t>(PRE_CHUNK, which.size() - b);
    long long segs = 0;
    for (int k = 0; k < count; ++k) {
      const FactorDev& v = f[which[b + k]];
      PreDev& d = chunk.f[k];
      d.src = v.src;
      d.xq = area + v.xq_off;
      d.N = v.N; d.C = v.C; d.H = v.H; d.W = v.W;
      d.NS = v.NS; d.R = v.R; d.n_rg = v.n_rg;
      d.sh = v.sh; d.ph = v.ph; d.pw = v.pw;
      d.rows_in = (v.R - 1) * v.sh + v.kh;
      d.RS = v.RS; d.PS = v.PS;
      d.rs_magic = magic_of(v.RS);
      d.ps_magic = magic_of(v.PS);
      d.c_magic = magic_of(v.C); d.ns_magic = magic_of(v.NS); d.rg_magic = magic_of(v.n_rg);
      CURV_REQUIRE((long long)v.n_chunks * v.NS * v.C * v.C < (1LL << 32), "curv_kfac: pre-tiled copy has too many planes");
      d.n_planes = v.n_chunks * v.NS * v.C;
      d.words = (long long)d.n_planes * v.PS;
      d.seg_base = (int)segs;
      d.pad = v.nch * v.PS + 64;
      segs += (d.words + SEG - 1) / SEG;
      CURV_REQUIRE(segs < (1LL << 31), "curv_kfac: too many patch segments");
    }
    hipLaunchKernelGGL(patch_prep_kernel, dim3((unsigned)segs), dim3(256), 0, stream, chunk, count);
    CURV_LAUNCH_CHECK();
  }
  return CURV_OK;
}

}  // namespace curv
