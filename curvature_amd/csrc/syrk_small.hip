// KFAC factor build of a SMALL launch (LeNet-scale: ten factors, 0.6 GFLOP) on gfx950: two launches.
//
// The grouped build of syrk.hip is made for models whose factors fill the chip: LDS-staged patch images, k-slices of tens
// of thousands of pixels, three MFMA kernels that are big straight-line programs.  On a launch this small every
// workgroup runs one short item, pays for fetching that program cold and for filling and staging 70 KB of LDS, and the
// launch lasts as long as its slowest item: 35-50 us per factor CLASS whatever its size (a 121-wide Linear factor with
// K = 100: 45 us; profiles/r04_lenet_trace.txt), five launches and two streams per update().
//
// Here a workgroup owns one 32 x 32 block (bi <= bj) of one factor and one slice of its K range (samples x output
// pixels); its four waves split the slice.  A lane gathers its two operand values of a step straight from the source
// tensor - row i = (channel, kh, kw) of the unfolded matrix at pixel k = (sample, y, x) is
// src[sample][channel][y sh + kh - ph][x sw + kw - pw], zero outside, 1 for the bias row (curvatures.py:329-343) -
// eight steps of loads in flight behind eight MFMAs; no LDS staging, no chunk planning.  The partial
// blocks go to slabs; syrk_small_reduce_kernel sums a block's slices in a fixed order (bit-reproducible), scales, adds
// into the factor and writes the mirror block.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "syrk_plan.h"

namespace curv {

struct SmallDev {
  const float* src;
  float* dst;
  int N, C, H, W, kh, kw, sh, sw, ph, pw, Ho, Wo;
  int rows, dim, has_bias, first;
  float scale;
  int nb, n_pairs, n_slices, kslice, K;
  int wg_base, red_base;
  long long slab_base;                   // floats
};
constexpr int SMALL_CHUNK = 24;
struct SmallChunk { SmallDev f[SMALL_CHUNK]; };
static_assert(sizeof(SmallChunk) <= 3840, "kernel argument block must stay below 4 KB");

typedef __attribute__((address_space(1))) float gfl;

__device__ __forceinline__ void small_pair(int nb, int pair, int& bi, int& bj) {
  bi = 0;
  while (pair >= nb - bi) { pair -= nb - bi; ++bi; }
  bj = bi + pair;
}

// one operand row of the unfolded matrix: kind 0 = patch row (plane offset, kernel offsets), 1 = ones (bias), 2 = zero
struct SmallRow { int kind, plane, oy, ox; };
__device__ __forceinline__ SmallRow small_row(const SmallDev& d, int i) {
  SmallRow r;
  if (i < d.rows) {
    const int khkw = d.kh * d.kw, c = i / khkw, rem = i - c * khkw, a = rem / d.kw, b = rem - a * d.kw;
    r.kind = 0; r.plane = c * d.H * d.W; r.oy = a - d.ph; r.ox = b - d.pw;
  } else {
    r.kind = (i == d.rows && d.has_bias) ? 1 : 2; r.plane = 0; r.oy = 0; r.ox = 0;
  }
  return r;
}

constexpr int SMALL_U = 8;               // steps of loads in flight
constexpr int SMALL_MAX_KSLICE = 2048;   // k values of a slice: the pixel table of a workgroup (16 KB); a wave sums a 1/8 of it in one chain
constexpr int SMALL_WAVES = 8;           // waves per workgroup: they split the slice's K range (16 accumulator registers / 8)

__global__ void __launch_bounds__(64 * SMALL_WAVES)
syrk_small_kernel(SmallChunk chunk, int count, float* __restrict__ slabs) {
  __shared__ float part[SMALL_WAVES][16][64];
  __shared__ int2 ktab[SMALL_MAX_KSLICE];
  int f = 0;
  while (f + 1 < count && chunk.f[f + 1].wg_base <= (int)blockIdx.x) ++f;
  const SmallDev& d = chunk.f[f];
  const int local = blockIdx.x - d.wg_base;
  const int pair = local / d.n_slices, slice = local - pair * d.n_slices;
  int bi, bj;
  small_pair(d.nb, pair, bi, bj);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31, h = lane >> 5;
  // this wave's share of the slice (whole k pairs)
  const int ks0 = slice * d.kslice, ks1 = min(d.K, ks0 + d.kslice);
  const int q = (((ks1 - ks0 + SMALL_WAVES - 1) / SMALL_WAVES) + 1) & ~1;
  const int kb = ks0 + wave * q, ke = min(ks1, kb + q);
  const SmallRow ra = small_row(d, 32 * bi + r32), rb = small_row(d, 32 * bj + r32);
  const int H = d.H, W = d.W, Ho = d.Ho, Wo = d.Wo, sh = d.sh, sw = d.sw, sample = d.C * H * W;
  const gfl* src = (const gfl*)d.src;
  // lane constants of the two rows: offset of the row's window corner inside a sample, and the window's displacement
  const int ca = ra.plane + ra.oy * W + ra.ox, cb = rb.plane + rb.oy * W + rb.ox;
  const bool diag = bi == bj;                              // (wave-uniform: a diagonal block has one operand)
  // Pixel table of the slice: entry k - ks0 = {offset of pixel k's window origin in the tensor, y sh << 16 | x sw} (two
  // integer divisions per pixel, once per workgroup; carrying (sample, y, x) through every step in registers cost ~90
  // instructions per step).  The launch is bound by this per-element work all the same - a table read, two window tests,
  // an address and a select per operand value, ~50 instructions per step beside its one MFMA: LeNet-5's ten factors are
  // 165 k wave-steps, 29 us, of which 6 remain with the K loop compiled out and 23-25 with either the loads or the MFMAs
  // compiled out.  Staging zero-padded images in LDS, as the grouped kernels do, would remove the tests; not built.
  for (int e = tid; e < ks1 - ks0; e += 64 * SMALL_WAVES) {
    const int k = ks0 + e, n = k / (Ho * Wo), rem = k - n * (Ho * Wo), y = rem / Wo, x = rem - y * Wo;
    ktab[e] = make_int2(n * sample + y * sh * W + x * sw, (y * sh) << 16 | (x * sw));
  }
  __syncthreads();
  f32x16 acc = {0};
  if (kb < ke) {
    const int nsteps = (ke - kb + 1) >> 1;
    // Two register sets: the loads of the next SMALL_U steps are in flight behind the MFMAs of the current ones.  A step
    // does not look at what it loaded: the load is always issued (element 0 of the tensor when the lane has nothing to
    // read) and the "zero / one / loaded" choice is made from per-step mask bits when the value is consumed
    // (`v = ok ? src[..] : c` made the compiler wait for memory inside every step).
    struct Batch { float a[SMALL_U], b[SMALL_U]; unsigned oka, okb, live; };
    auto fetch = [&](Batch& t, int s0) __attribute__((always_inline)) {
      t.oka = t.okb = t.live = 0u;
#pragma unroll
      for (int u = 0; u < SMALL_U; ++u) {
        const int kk = kb + 2 * (s0 + u) + h;
        const bool live = s0 + u < nsteps && kk < ke;
        const int2 px = ktab[live ? kk - ks0 : 0];
        const int ys = px.y >> 16, xs = px.y & 0xffff;
        const bool oka = live && ra.kind == 0 && (unsigned)(ys + ra.oy) < (unsigned)H && (unsigned)(xs + ra.ox) < (unsigned)W;
        const bool okb = live && rb.kind == 0 && (unsigned)(ys + rb.oy) < (unsigned)H && (unsigned)(xs + rb.ox) < (unsigned)W;
        t.a[u] = src[oka ? px.x + ca : 0];
        if (!diag) t.b[u] = src[okb ? px.x + cb : 0];
        t.oka |= (oka ? 1u : 0u) << u; t.okb |= (okb ? 1u : 0u) << u; t.live |= (live ? 1u : 0u) << u;
      }
    };
    const float one_a = ra.kind == 1 ? 1.0f : 0.0f, one_b = rb.kind == 1 ? 1.0f : 0.0f;
    auto compute = [&](const Batch& t, int s0) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < SMALL_U; ++u) {
        if (s0 + u < nsteps) {
          const bool live = (t.live >> u) & 1u;
          const float va = ((t.oka >> u) & 1u) ? t.a[u] : (live ? one_a : 0.0f);
          const float vb = diag ? va : (((t.okb >> u) & 1u) ? t.b[u] : (live ? one_b : 0.0f));
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(va, vb, acc, 0, 0, 0);
        }
      }
    };
    Batch t0, t1;
    fetch(t0, 0);
    for (int s0 = 0; s0 < nsteps; s0 += 2 * SMALL_U) {
      if (s0 + SMALL_U < nsteps) fetch(t1, s0 + SMALL_U);
      compute(t0, s0);
      if (s0 + 2 * SMALL_U < nsteps) fetch(t0, s0 + 2 * SMALL_U);
      if (s0 + SMALL_U < nsteps) compute(t1, s0 + SMALL_U);
    }
  }
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) part[wave][reg][lane] = acc[reg];
  __syncthreads();
  // C/D map of the 32x32 block: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5); wave w finishes
  // registers 2 w, 2 w + 1 (the partial blocks summed in wave order)
  gfl* slab = (gfl*)slabs + d.slab_base + ((long long)pair * d.n_slices + slice) * 1024;
#pragma unroll
  for (int qq = 0; qq < 16 / SMALL_WAVES; ++qq) {
    const int reg = (16 / SMALL_WAVES) * wave + qq;
    float v = part[0][reg][lane];
#pragma unroll
    for (int p = 1; p < SMALL_WAVES; ++p) v += part[p][reg][lane];
    slab[((reg & 3) + 8 * (reg >> 2) + 4 * h) * 32 + r32] = v;
  }
}

// One workgroup of 1024 threads per block: thread group g (of four) sums the g-th quarter of the block's slices, 16 loads
// per element in flight (a slab written by another XCD is a 2 us round trip: summing 64 slices eight at a time took 15 us);
// the four partial sums meet in LDS and are added in group order, so the result is a fixed function of the slabs.
__global__ void __launch_bounds__(1024)
syrk_small_reduce_kernel(SmallChunk chunk, int count, const float* __restrict__ slabs) {
  __shared__ float partial[3][1024];
  int f = 0;
  while (f + 1 < count && chunk.f[f + 1].red_base <= (int)blockIdx.x) ++f;
  const SmallDev& d = chunk.f[f];
  const int pair = blockIdx.x - d.red_base;
  int bi, bj;
  small_pair(d.nb, pair, bi, bj);
  const int dim = d.dim, n_slices = d.n_slices;
  const float scale = d.scale;
  const bool first = d.first != 0;
  const float* base = slabs + d.slab_base + (long long)pair * n_slices * 1024;
  gfl* dst = (gfl*)d.dst;
  const int g = threadIdx.x >> 8, t = threadIdx.x & 255;
  const int per = (n_slices + 3) >> 2, s_lo = g * per, s_hi = min(n_slices, s_lo + per);
  float v[4];
  bool live[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = t + 256 * u;
    live[u] = 32 * bi + (e >> 5) < dim && 32 * bj + (e & 31) < dim;
    v[u] = 0.0f;
  }
  int s = s_lo;
  for (; s + 16 <= s_hi; s += 16) {
    float w[4][16];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (live[u]) {
#pragma unroll
        for (int k = 0; k < 16; ++k) w[u][k] = base[(long long)(s + k) * 1024 + t + 256 * u];
      }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (live[u]) {
#pragma unroll
        for (int k = 0; k < 16; ++k) v[u] += w[u][k];
      }
  }
  for (; s < s_hi; ++s) {
#pragma unroll
    for (int u = 0; u < 4; ++u) if (live[u]) v[u] += base[(long long)s * 1024 + t + 256 * u];
  }
  if (g > 0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) partial[g - 1][t + 256 * u] = v[u];
  }
  __syncthreads();
  if (g > 0) return;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (!live[u]) continue;
    const int e = t + 256 * u, gi = 32 * bi + (e >> 5), gj = 32 * bj + (e & 31);
    const float out = (((v[u] + partial[0][e]) + partial[1][e]) + partial[2][e]) * scale;
    const long long idx = (long long)gi * dim + gj;
    dst[idx] = first ? out : dst[idx] + out;
    if (bi != bj) {
      const long long mdx = (long long)gj * dim + gi;
      dst[mdx] = first ? out : dst[mdx] + out;
    }
  }
}

// ------------------------------------------------------------------------------------------------ host side
// (CURV_SMALL_MAX_FLOP, include/curv_hip.h: executed multiply-add flops of the launch up to which it takes this form)
constexpr int SMALL_KSLICE = 320;        // k values per slice (LeNet-5 at N = 100: 764 workgroups, ~3 per CU)
constexpr int SMALL_MAX_SLICES = 64;     // the reduce pass walks a block's slices eight at a time: a round trip each

struct SmallPlan { std::vector<SmallDev> f; long long wgs = 0, red_wgs = 0, slab_floats = 0; };

static bool small_plan(const curv_factor_desc* descs, int n, SmallPlan& plan) {
  const char* env = getenv("CURV_KFAC_SMALL");
  if (env != nullptr && atoi(env) == 0) return false;
  if (n <= 0 || n > 4 * SMALL_CHUNK) return false;
  // the caller may have decided for the whole (unsharded) model: a rank's share must not switch forms on its own size
  bool forced = n > 0;
  for (int i = 0; i < n; ++i) {
    if (descs[i].path_hint == CURV_PATH_GROUPED) return false;
    forced = forced && descs[i].path_hint == CURV_PATH_SMALL;
  }
  plan.f.resize(n);
  double flop = 0.0, work = 0.0;
  for (int i = 0; i < n; ++i) {
    const curv_factor_desc& s = descs[i];
    SmallDev& d = plan.f[i];
    memset(&d, 0, sizeof(d));
    if (s.N <= 0 || s.C <= 0 || s.H <= 0 || s.W <= 0 || s.kh <= 0 || s.kw <= 0 || s.sh <= 0 || s.sw <= 0 || s.ph < 0 || s.pw < 0)
      return false;                                                       // (the grouped path reports the error)
    if (s.H + 2 * s.ph < s.kh || s.W + 2 * s.pw < s.kw) return false;
    if ((long long)s.N * s.C * s.H * s.W >= (1LL << 31) || s.H >= (1 << 15) || s.W >= (1 << 15)) return false;   // (pixel table packs y sh, x sw in 16 bits)
    d.src = s.src; d.dst = s.dst;
    d.N = s.N; d.C = s.C; d.H = s.H; d.W = s.W; d.kh = s.kh; d.kw = s.kw; d.sh = s.sh; d.sw = s.sw; d.ph = s.ph; d.pw = s.pw;
    d.Ho = (s.H + 2 * s.ph - s.kh) / s.sh + 1;
    d.Wo = (s.W + 2 * s.pw - s.kw) / s.sw + 1;
    d.rows = s.C * s.kh * s.kw;
    d.has_bias = s.has_bias ? 1 : 0;
    d.dim = d.rows + d.has_bias;
    d.first = s.first; d.scale = s.scale;
    const long long K = (long long)s.N * d.Ho * d.Wo;
    if (K >= (1LL << 30) || (long long)d.dim * d.dim >= (1LL << 31)) return false;
    d.K = (int)K;
    d.nb = cdiv(d.dim, 32);
    d.n_pairs = d.nb * (d.nb + 1) / 2;
    flop += 2.0 * 1024.0 * d.n_pairs * (double)K;
    work += (double)d.n_pairs * (double)K;
    if (!forced && flop > CURV_SMALL_MAX_FLOP) return false;
  }
  // one slicing rule per FACTOR (slices of ~SMALL_KSLICE k values, at most SMALL_MAX_SLICES of them): what a factor's
  // sums look like does not depend on what else is in the launch - a layer-sharded rank gets the bits of the unsharded run
  const long long px = SMALL_KSLICE;
  (void)work;
  for (int i = 0; i < n; ++i) {
    SmallDev& d = plan.f[i];
    d.n_slices = (int)std::min<long long>(cdivll(d.K, px), SMALL_MAX_SLICES);
    d.kslice = (cdiv(d.K, d.n_slices) + 7) & ~7;
    if (d.kslice > SMALL_MAX_KSLICE) return false;            // a factor this long is not a small launch's
    d.n_slices = cdiv(d.K, d.kslice);
    d.wg_base = (int)plan.wgs;
    d.red_base = (int)plan.red_wgs;
    d.slab_base = plan.slab_floats;
    plan.wgs += (long long)d.n_pairs * d.n_slices;
    plan.red_wgs += d.n_pairs;
    plan.slab_floats += (long long)d.n_pairs * d.n_slices * 1024;
    if (plan.wgs >= (1LL << 24)) return false;
  }
  return true;
}

// which form a launch of exactly these factors takes on its own (path hints ignored): every gate of small_plan - flops,
// job count, slice length, workgroup count - evaluated on the UNSHARDED model, for the ranks of a layer-sharded run
int kfac_path_for(const curv_factor_desc* descs, int n) {
  std::vector<curv_factor_desc> plain(descs, descs + n);
  for (curv_factor_desc& d : plain) d.path_hint = CURV_PATH_AUTO;
  SmallPlan plan;
  return small_plan(plain.data(), n, plan) ? CURV_PATH_SMALL : CURV_PATH_GROUPED;
}

// bytes of workspace the small path needs for these factors; 0: the launch is not a small one
size_t kfac_small_workspace_bytes(const curv_factor_desc* descs, int n) {
  SmallPlan plan;
  if (!small_plan(descs, n, plan)) return 0;
  return align_up((size_t)plan.slab_floats * sizeof(float), 256);
}

// CURV_OK: done.  CURV_ERR_WORKSPACE (without an error text): not a small launch, or the workspace does not hold its
// slabs - the caller takes the grouped path.
int kfac_accumulate_small(hipStream_t stream, const curv_factor_desc* descs, int n, void* workspace, size_t workspace_bytes,
                          void* ev_start, void* ev_stop) {
  SmallPlan plan;
  if (!small_plan(descs, n, plan)) return CURV_ERR_WORKSPACE;
  if (workspace == nullptr || workspace_bytes < (size_t)plan.slab_floats * sizeof(float)) return CURV_ERR_WORKSPACE;
  for (int i = 0; i < n; ++i)
    CURV_REQUIRE(descs[i].src != nullptr && descs[i].dst != nullptr, "curv_kfac: factor %d: null pointer", i);
  float* slabs = reinterpret_cast<float*>(workspace);
  if (ev_start) CURV_HIP_CHECK(hipEventRecord((hipEvent_t)ev_start, stream));
  for (int b = 0; b < n; b += SMALL_CHUNK) {
    SmallChunk chunk;
    memset(&chunk, 0, sizeof(chunk));
    const int count = std::min(SMALL_CHUNK, n - b);
    long long wgs = 0, red = 0;
    for (int k = 0; k < count; ++k) {
      chunk.f[k] = plan.f[b + k];
      chunk.f[k].wg_base = (int)wgs;
      chunk.f[k].red_base = (int)red;
      wgs += (long long)chunk.f[k].n_pairs * chunk.f[k].n_slices;
      red += chunk.f[k].n_pairs;
    }
    hipLaunchKernelGGL(syrk_small_kernel, dim3((unsigned)wgs), dim3(64 * SMALL_WAVES), 0, stream, chunk, count, slabs);
    CURV_LAUNCH_CHECK();
    hipLaunchKernelGGL(syrk_small_reduce_kernel, dim3((unsigned)red), dim3(1024), 0, stream, chunk, count, (const float*)slabs);
    CURV_LAUNCH_CHECK();
  }
  if (ev_stop) CURV_HIP_CHECK(hipEventRecord((hipEvent_t)ev_stop, stream));
  return CURV_OK;
}

}  // namespace curv
