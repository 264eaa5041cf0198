// HBM-bound elementwise kernels of the Diagonal / EFB / INF estimators.
// Reference arithmetic: curvature/curvatures.py:141-193 (Diagonal), :431-434 (EFB diags),
// :449 (EFB invert), :523-526 (INF invert).
#include <cstring>

#include "common.h"
#include "../../include/curv_hip.h"

namespace curv {

// One grid-stride sweep, 16 B per lane where alignment allows.
static inline dim3 sweep_grid(long long count, int per_thread) {
  long long blocks = cdivll(count, 256LL * per_thread);
  if (blocks > 2048) blocks = 2048;   // 256 CUs x 8 blocks, grid-stride the rest
  if (blocks < 1) blocks = 1;
  return dim3((unsigned)blocks);
}

// (s*v + n)^(-1/2) with the reference's rounding sequence: mul, add, reciprocal, sqrt (no fma contraction).
__device__ __forceinline__ float rsq_aff(float s, float v, float n) {
  return __fsqrt_rn(__frcp_rn(__fadd_rn(__fmul_rn(s, v), n)));
}

// out[i] = (s * v[i] + n)^(-1/2)
__global__ void __launch_bounds__(256)
rsqrt_affine_kernel(const float* __restrict__ v, float s, float n, float* __restrict__ out,
                    long long count) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool vec = ((reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  if (vec) {
    const long long nvec = count >> 2;
    const float4* v4 = reinterpret_cast<const float4*>(v);
    float4* o4 = reinterpret_cast<float4*>(out);
    for (long long j = i; j < nvec; j += stride) {
      float4 a = v4[j];
      float4 r;
      // reciprocal().sqrt() in the reference: 1/x then sqrt, both correctly rounded fp32
      r.x = rsq_aff(s, a.x, n);
      r.y = rsq_aff(s, a.y, n);
      r.z = rsq_aff(s, a.z, n);
      r.w = rsq_aff(s, a.w, n);
      o4[j] = r;
    }
    for (long long j = (nvec << 2) + i; j < count; j += stride) out[j] = rsq_aff(s, v[j], n);
  } else {
    for (long long j = i; j < count; j += stride) out[j] = rsq_aff(s, v[j], n);
  }
}

// state[r, c] (+)= bs * grad(r, c)^2 where grad = [gw (rows x cols_w) | gb (rows)] (bias column last).
__global__ void __launch_bounds__(256)
sq_accumulate_kernel(const float* __restrict__ gw, const float* __restrict__ gb, int rows, int cols_w,
                     float bs, float* __restrict__ state, int first) {
  const int cols = cols_w + (gb != nullptr ? 1 : 0);
  const long long count = (long long)rows * cols;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += stride) {
    const int r = (int)(j / cols);
    const int c = (int)(j - (long long)r * cols);
    const float g = (c < cols_w) ? gw[(long long)r * cols_w + c] : gb[r];
    const float val = g * g * bs;
    state[j] = first ? val : state[j] + val;
  }
}

// The same for many layers in one launch: descriptors travel in the kernel arguments; a workgroup finds its layer by
// the prefix of 1024-element blocks (first_block ascending).
struct SqDev {
  const float* gw; const float* gb; float* state;
  int rows, cols_w, first, first_block;
};
constexpr int SQ_CHUNK = 96;
struct SqChunk { SqDev d[SQ_CHUNK]; };
static_assert(sizeof(SqChunk) <= 3840, "kernel argument block must stay below 4 KB");
__global__ void __launch_bounds__(256) sq_accumulate_batched_kernel(SqChunk chunk, int count, float bs) {
  int lo = 0, hi = count - 1;                       // last layer with first_block <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (chunk.d[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const SqDev& d = chunk.d[lo];
  const int cols = d.cols_w + (d.gb != nullptr ? 1 : 0);
  const long long n = (long long)d.rows * cols;
  const long long j0 = (long long)((int)blockIdx.x - d.first_block) * 1024 + threadIdx.x;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long long j = j0 + 256 * u;
    if (j < n) {
      const int r = (int)(j / cols);
      const int c = (int)(j - (long long)r * cols);
      const float g = (c < d.cols_w) ? d.gw[(long long)r * d.cols_w + c] : d.gb[r];
      const float val = g * g * bs;
      d.state[j] = d.first ? val : d.state[j] + val;
    }
  }
}

// v[i] = max(v[i], 0) in place (INF.invert clamps the correction term on `state`).
__global__ void __launch_bounds__(256) clamp_min0_kernel(float* __restrict__ v, long long count) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += stride) {
    const float a = v[j];
    v[j] = a < 0.0f ? 0.0f : a;
  }
}

// out[i] = sqrt(s * v[i])
__global__ void __launch_bounds__(256)
sqrt_scale_kernel(const float* __restrict__ v, float s, float* __restrict__ out, long long count) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += stride)
    out[j] = sqrtf(s * v[j]);
}

// out[i] = a[i] * b[i]
__global__ void __launch_bounds__(256)
mul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
           long long count) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += stride)
    out[j] = a[j] * b[j];
}

// out[(i*br + k)][(j*bc + l)] = a[i][j] * b[k][l]   (utils.kron)
__global__ void __launch_bounds__(256)
kron_kernel(const float* __restrict__ a, int ar, int ac, const float* __restrict__ b, int br, int bc,
            float* __restrict__ out) {
  const long long cols = (long long)ac * bc, count = (long long)ar * br * cols;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += stride) {
    const long long row = e / cols, col = e - row * cols;
    const int i = (int)(row / br), k = (int)(row - (long long)i * br);
    const int j = (int)(col / bc), l = (int)(col - (long long)j * bc);
    out[e] = a[(long long)i * ac + j] * b[(long long)k * bc + l];
  }
}

}  // namespace curv

using namespace curv;

extern "C" int curv_kron(void* stream, const float* a, int ar, int ac, const float* b, int br, int bc,
                         float* out) {
  CURV_REQUIRE(ar >= 0 && ac >= 0 && br >= 0 && bc >= 0, "curv_kron: negative shape");
  const long long count = (long long)ar * br * ac * bc;
  if (count == 0) return CURV_OK;
  CURV_REQUIRE(a && b && out, "curv_kron: null pointer");
  hipLaunchKernelGGL(kron_kernel, sweep_grid(count, 1), dim3(256), 0, (hipStream_t)stream, a, ar, ac, b, br, bc, out);
  CURV_LAUNCH_CHECK();
  return CURV_OK;
}

extern "C" int curv_rsqrt_affine(void* stream, const float* v, double s, double n, float* out,
                                 long long count) {
  CURV_REQUIRE(count >= 0, "curv_rsqrt_affine: negative count");
  if (count == 0) return CURV_OK;
  CURV_REQUIRE(v && out, "curv_rsqrt_affine: null pointer");
  hipLaunchKernelGGL(rsqrt_affine_kernel, sweep_grid(count, 4), dim3(256), 0, (hipStream_t)stream, v,
                     (float)s, (float)n, out, count);
  CURV_LAUNCH_CHECK();
  return CURV_OK;
}

extern "C" int curv_sq_accumulate(void* stream, const float* grad_w, const float* grad_b, int rows,
                                  int cols_w, double batch_size, float* state, int first) {
  CURV_REQUIRE(rows >= 0 && cols_w >= 0, "curv_sq_accumulate: negative shape");
  const long long count = (long long)rows * (cols_w + (grad_b ? 1 : 0));
  if (count == 0) return CURV_OK;
  CURV_REQUIRE(grad_w && state, "curv_sq_accumulate: null pointer");
  hipLaunchKernelGGL(sq_accumulate_kernel, sweep_grid(count, 1), dim3(256), 0, (hipStream_t)stream,
                     grad_w, grad_b, rows, cols_w, (float)batch_size, state, first);
  CURV_LAUNCH_CHECK();
  return CURV_OK;
}

extern "C" int curv_sq_accumulate_batched(void* stream, const curv_sq_desc* descs, int n, double batch_size) {
  CURV_REQUIRE(n >= 0 && (n == 0 || descs != nullptr), "curv_sq_accumulate_batched: bad arguments");
  for (int b = 0; b < n; b += SQ_CHUNK) {
    SqChunk chunk;
    memset(&chunk, 0, sizeof(chunk));
    int count = 0;
    long long blocks = 0;
    for (int i = b; i < n && i < b + SQ_CHUNK; ++i) {
      const curv_sq_desc& s = descs[i];
      CURV_REQUIRE(s.rows >= 0 && s.cols_w >= 0, "curv_sq_accumulate_batched: desc %d: negative shape", i);
      const long long elems = (long long)s.rows * (s.cols_w + (s.grad_b ? 1 : 0));
      if (elems == 0) continue;
      CURV_REQUIRE(s.grad_w && s.state, "curv_sq_accumulate_batched: desc %d: null pointer", i);
      SqDev& d = chunk.d[count++];
      d.gw = s.grad_w; d.gb = s.grad_b; d.state = s.state;
      d.rows = s.rows; d.cols_w = s.cols_w; d.first = s.first;
      d.first_block = (int)blocks;
      blocks += (elems + 1023) / 1024;
      CURV_REQUIRE(blocks < (1LL << 31), "curv_sq_accumulate_batched: too many elements");
    }
    if (count == 0) continue;
    hipLaunchKernelGGL(sq_accumulate_batched_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, chunk,
                       count, (float)batch_size);
    CURV_LAUNCH_CHECK();
  }
  return CURV_OK;
}

extern "C" int curv_clamp_min0(void* stream, float* v, long long count) {
  if (count <= 0) return CURV_OK;
  CURV_REQUIRE(v, "curv_clamp_min0: null pointer");
  hipLaunchKernelGGL(clamp_min0_kernel, sweep_grid(count, 1), dim3(256), 0, (hipStream_t)stream, v, count);
  CURV_LAUNCH_CHECK();
  return CURV_OK;
}

extern "C" int curv_sqrt_scale(void* stream, const float* v, double s, float* out, long long count) {
  if (count <= 0) return CURV_OK;
  CURV_REQUIRE(v && out, "curv_sqrt_scale: null pointer");
  hipLaunchKernelGGL(sqrt_scale_kernel, sweep_grid(count, 1), dim3(256), 0, (hipStream_t)stream, v,
                     (float)s, out, count);
  CURV_LAUNCH_CHECK();
  return CURV_OK;
}

extern "C" int curv_mul(void* stream, const float* a, const float* b, float* out, long long count) {
  if (count <= 0) return CURV_OK;
  CURV_REQUIRE(a && b && out, "curv_mul: null pointer");
  hipLaunchKernelGGL(mul_kernel, sweep_grid(count, 1), dim3(256), 0, (hipStream_t)stream, a, b, out, count);
  CURV_LAUNCH_CHECK();
  return CURV_OK;
}

// ------------------------------------------------------------------------------------------------
// Batched device-to-device copy (see curv_copy_batched): descriptors travel in the kernel arguments,
// one workgroup per 64 KB piece.
// ------------------------------------------------------------------------------------------------
namespace curv {
constexpr int COPY_MAX = 96;                       // buffers per launch: the argument block stays below 4 KB
constexpr unsigned long long COPY_PIECE = 65536;
struct CopyArgs {
  void* dst[COPY_MAX];
  const void* src[COPY_MAX];
  unsigned long long bytes[COPY_MAX];
  int wg_base[COPY_MAX + 1];
};
static_assert(sizeof(CopyArgs) <= 3840, "kernel argument block must stay below 4 KB");

__global__ void __launch_bounds__(256) copy_batched_kernel(CopyArgs a, int n) {
  const int lane = threadIdx.x & 63, bid = (int)blockIdx.x;
  int t = 0;
  for (int b0 = 0; b0 < n; b0 += 64) {             // largest t with wg_base[t] <= bid (bases ascend)
    const int i = b0 + lane;
    const bool le = i < n && a.wg_base[i] <= bid;
    t += __popcll(__ballot(le));
  }
  t = __builtin_amdgcn_readfirstlane(t - 1);
  const unsigned long long off = (unsigned long long)(bid - a.wg_base[t]) * COPY_PIECE;
  const unsigned long long left = a.bytes[t] - off;
  const unsigned long long len = left < COPY_PIECE ? left : COPY_PIECE;
  char* d = reinterpret_cast<char*>(a.dst[t]) + off;
  const char* s = reinterpret_cast<const char*>(a.src[t]) + off;
  const unsigned long long mis = reinterpret_cast<unsigned long long>(d) | reinterpret_cast<unsigned long long>(s);
  if ((mis & 15) == 0) {
    const unsigned long long n16 = len >> 4;
    for (unsigned long long i = threadIdx.x; i < n16; i += 256)
      reinterpret_cast<uint4*>(d)[i] = reinterpret_cast<const uint4*>(s)[i];
    for (unsigned long long i = (n16 << 4) + threadIdx.x; i < len; i += 256) d[i] = s[i];
  } else if ((mis & 3) == 0) {
    const unsigned long long n4 = len >> 2;
    for (unsigned long long i = threadIdx.x; i < n4; i += 256)
      reinterpret_cast<unsigned*>(d)[i] = reinterpret_cast<const unsigned*>(s)[i];
    for (unsigned long long i = (n4 << 2) + threadIdx.x; i < len; i += 256) d[i] = s[i];
  } else {
    for (unsigned long long i = threadIdx.x; i < len; i += 256) d[i] = s[i];
  }
}
}  // namespace curv

extern "C" int curv_copy_batched(void* stream, const curv_copy_desc* descs, int n) {
  using namespace curv;
  CURV_REQUIRE(n >= 0 && (n == 0 || descs != nullptr), "curv_copy_batched: bad arguments");
  int i = 0;
  while (i < n) {
    CopyArgs a;
    memset(&a, 0, sizeof(a));
    int cnt = 0, wgs = 0;
    while (i < n && cnt < COPY_MAX) {
      const curv_copy_desc& c = descs[i++];
      if (c.bytes == 0) continue;
      CURV_REQUIRE(c.dst != nullptr && c.src != nullptr, "curv_copy_batched: buffer %d: null pointer", i - 1);
      const unsigned long long pieces = (c.bytes + COPY_PIECE - 1) / COPY_PIECE;
      CURV_REQUIRE(pieces < (1ull << 24), "curv_copy_batched: buffer %d too large", i - 1);
      a.dst[cnt] = c.dst; a.src[cnt] = c.src; a.bytes[cnt] = c.bytes; a.wg_base[cnt] = wgs;
      wgs += (int)pieces;
      ++cnt;
    }
    if (cnt == 0) continue;
    a.wg_base[cnt] = wgs;
    hipLaunchKernelGGL(copy_batched_kernel, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, a, cnt);
    CURV_LAUNCH_CHECK();
  }
  return CURV_OK;
}
