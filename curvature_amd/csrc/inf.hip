// INF (sparse information form) helpers on gfx950 (curvature/curvatures.py:463-672): the integer index
// work of `_dim_reduction` and the small structured products of `pre_sampler`, none of which ever
// materialises a Kronecker product.
#include "common.h"

namespace curv {

typedef __attribute__((address_space(1))) float gflt;

// ------------------------------------------------------------------------------------------------
// Top-`rank` selection of |lambda| and the row / column index sets it induces (curvatures.py:617-634):
//   top = argsort(-|lambda|)[:rank];  I = unique(top // m);  J = unique(top % m)   (ascending)
// One 1024-thread workgroup per layer: a 4-pass radix select on the bit pattern of |x| finds the value
// of the rank-th largest magnitude, everything strictly larger is selected, ties at the threshold are
// taken by a fixed deterministic rule until `rank` elements are selected (the reference's unstable
// argsort leaves tie order unspecified).  Integer work: results are exact.
// ------------------------------------------------------------------------------------------------
constexpr int SEL_THREADS = 1024;

struct SelDev {
  const float* lam;       // n*m values, index i*m + j
  long long* I;           // out: up to n row indices
  long long* J;           // out: up to m col indices
  int* counts;            // out: {a, b}
  int n, m, rank, pad;
};

constexpr int SEL_BATCH = 32;          // layers per launch (one workgroup each), descriptors by value
struct SelBatch { SelDev d[SEL_BATCH]; };
static_assert(sizeof(SelBatch) <= 3584, "kernel argument block must stay below 4 KB");

__global__ void __launch_bounds__(SEL_THREADS)
inf_select_kernel(SelBatch batch, int n_desc) {
  __shared__ unsigned hist[256];
  __shared__ unsigned s_prefix, s_remaining;
  __shared__ unsigned char rowflag[8192], colflag[8192];
  __shared__ int s_scan[SEL_THREADS];
  const SelDev& d = batch.d[blockIdx.x];
  const int tid = threadIdx.x;
  const long long total = (long long)d.n * d.m;
  const gflt* lam = (const gflt*)d.lam;
  const unsigned rank = (unsigned)min((long long)d.rank, total);

  // radix select: find threshold key T = the rank-th largest key (key = bits of |x|, monotone for >= 0)
  unsigned prefix = 0, remaining = rank;      // keys with (key >> shift+8 << shift+8) == prefix still candidates
  for (int pass = 3; pass >= 0; --pass) {
    const int shift = 8 * pass;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const unsigned himask = (pass == 3) ? 0u : (0xffffffffu << (shift + 8));
    for (long long e = tid; e < total; e += SEL_THREADS) {
      const unsigned key = __float_as_uint(lam[e]) & 0x7fffffffu;
      if ((key & himask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned rem = remaining, b = 255;
      for (;; --b) {                            // walk buckets from the largest digit down
        if (hist[b] >= rem || b == 0) break;
        rem -= hist[b];
      }
      s_prefix = prefix | (b << shift);
      s_remaining = rem;                        // how many of the elements with this digit are still needed
    }
    __syncthreads();
    prefix = s_prefix;
    remaining = s_remaining;
    __syncthreads();
  }
  const unsigned T = prefix;                    // threshold key; `remaining` ties at T are selected

  for (int e = tid; e < 8192; e += SEL_THREADS) { rowflag[e] = 0; colflag[e] = 0; }
  __syncthreads();
  // strictly larger keys: always selected
  for (long long e = tid; e < total; e += SEL_THREADS) {
    const unsigned key = __float_as_uint(lam[e]) & 0x7fffffffu;
    if (key > T) { rowflag[e / d.m] = 1; colflag[e % d.m] = 1; }
  }
  // ties at T: a deterministic rule (thread-major order over the strided sweep), `remaining` of them
  {
    int mine = 0;
    for (long long e = tid; e < total; e += SEL_THREADS)
      mine += ((__float_as_uint(lam[e]) & 0x7fffffffu) == T) ? 1 : 0;
    s_scan[tid] = mine;
    __syncthreads();
    for (int o = 1; o < SEL_THREADS; o <<= 1) {          // Hillis-Steele inclusive scan
      const int v = (tid >= o) ? s_scan[tid - o] : 0;
      __syncthreads();
      s_scan[tid] += v;
      __syncthreads();
    }
    unsigned taken = (unsigned)(s_scan[tid] - mine);     // ties owned by lower thread ids
    for (long long e = tid; e < total && taken < remaining; e += SEL_THREADS) {
      if ((__float_as_uint(lam[e]) & 0x7fffffffu) == T) {
        rowflag[e / d.m] = 1;
        colflag[e % d.m] = 1;
        ++taken;
      }
    }
  }
  __syncthreads();
  // compaction (ascending) by one thread per output list: n, m <= 8192, runs once per layer
  if (tid == 0) {
    int a = 0;
    for (int i = 0; i < d.n; ++i) if (rowflag[i]) d.I[a++] = i;
    d.counts[0] = a;
  }
  if (tid == 64) {
    int b = 0;
    for (int j = 0; j < d.m; ++j) if (colflag[j]) d.J[b++] = j;
    d.counts[1] = b;
  }
}

// out[p][i*a + k] = U[p][i] * U[p][k]   (rows of the Khatri-Rao square used by the closed-form vtv);
// T = double: the product of two fp32 values is exact
template <typename T>
__global__ void __launch_bounds__(256)
colpairs_kernel(const float* __restrict__ U, int n, int a, long long u_rs, T* __restrict__ out) {
  const long long total = (long long)n * a * a;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int p = (int)(e / ((long long)a * a));
    const int rem = (int)(e - (long long)p * a * a);
    const int i = rem / a, k = rem - i * a;
    out[e] = (T)U[p * u_rs + i] * (T)U[p * u_rs + k];
  }
}

// out[i] = (double) v[i]^2
__global__ void __launch_bounds__(256)
square_f64_kernel(const float* __restrict__ v, double* __restrict__ out, long long count) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += stride) {
    const double x = (double)v[e];
    out[e] = x * x;
  }
}

// M[i][j] = (M[i][j] or src[i][j]) * dl[i] * dr[j]; src may be fp64 (P_c = diag(sigma) L_c diag(sigma))
template <typename T>
__global__ void __launch_bounds__(256)
diag_scale_kernel(const T* __restrict__ src, float* __restrict__ dst, const float* __restrict__ dl,
                  const float* __restrict__ dr, int rows, int cols) {
  const long long total = (long long)rows * cols;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int i = (int)(e / cols), j = (int)(e - (long long)i * cols);
    dst[e] = (float)((double)src[e] * (double)dl[i] * (double)dr[j]);
  }
}

// V4[(i,k),(j,l)] (a*a x b*b) -> vtv[(i,j),(k,l)] (ab x ab) scaled by sigma, then symmetrised:
//   vtv[x][y] = (w(x,y) + w(y,x)) / 2,  w((i,j),(k,l)) = sigma[i*b+j] sigma[k*b+l] V4[(i*a+k)][(j*b+l)]
template <typename T>
__global__ void __launch_bounds__(256)
vtv_assemble_kernel(const T* __restrict__ V4, const float* __restrict__ sigma, int a, int b,
                    T* __restrict__ vtv) {
  const int ab = a * b;
  const long long total = (long long)ab * ab;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long ld4 = (long long)b * b;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int x = (int)(e / ab), y = (int)(e - (long long)x * ab);
    const int i = x / b, j = x - i * b, k = y / b, l = y - k * b;
    const T sxy = (T)sigma[x] * (T)sigma[y];
    const T w1 = sxy * V4[(long long)(i * a + k) * ld4 + (j * b + l)];
    const T w2 = sxy * V4[(long long)(k * a + i) * ld4 + (l * b + j)];
    vtv[e] = (w1 + w2) * (T)0.5;
  }
}

// The column pairs of colpairs are symmetric in (i, k), and so is everything built from them: the packed forms keep
// i <= k only - column t(i, k) = i a - i (i - 1) / 2 + (k - i) of a (n, a (a + 1) / 2) matrix - which halves the first
// product of the closed-form V_s^T V_s and quarters the second.
__device__ __forceinline__ int packed_pair(int i, int k, int a) {      // any order of (i, k)
  const int lo = i < k ? i : k, hi = i < k ? k : i;
  return lo * a - lo * (lo - 1) / 2 + (hi - lo);
}
__global__ void __launch_bounds__(256)
colpairs_sym_kernel(const float* __restrict__ U, int n, int a, long long u_rs, double* __restrict__ out) {
  const int ap = a * (a + 1) / 2;
  const long long total = (long long)n * a * a;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int p = (int)(e / ((long long)a * a));
    const int rem = (int)(e - (long long)p * a * a);
    const int i = rem / a, k = rem - i * a;
    if (i <= k) out[(long long)p * ap + packed_pair(i, k, a)] = (double)U[p * u_rs + i] * (double)U[p * u_rs + k];
  }
}
// vtv[(i,j),(k,l)] = sigma[i*b+j] sigma[k*b+l] V4p[t_a(i,k)][t_b(j,l)]  (symmetric by construction)
__global__ void __launch_bounds__(256)
vtv_assemble_sym_kernel(const double* __restrict__ V4p, const float* __restrict__ sigma, int a, int b,
                        double* __restrict__ vtv) {
  const int ab = a * b;
  const long long total = (long long)ab * ab;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long ld4 = (long long)b * (b + 1) / 2;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int x = (int)(e / ab), y = (int)(e - (long long)x * ab);
    const int i = x / b, j = x - i * b, k = y / b, l = y - k * b;
    vtv[e] = (double)sigma[x] * (double)sigma[y] * V4p[(long long)packed_pair(i, k, a) * ld4 + packed_pair(j, l, b)];
  }
}

// out[i][j] = A(i,j) * B(i,j) on strided 2-D views
__global__ void __launch_bounds__(256)
mul2d_kernel(const float* __restrict__ A, long long a_rs, long long a_cs, const float* __restrict__ B,
             long long b_rs, long long b_cs, float* __restrict__ out, int rows, int cols) {
  const long long total = (long long)rows * cols;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int i = (int)(e / cols), j = (int)(e - (long long)i * cols);
    out[e] = A[i * a_rs + j * a_cs] * B[i * b_rs + j * b_cs];
  }
}

// out[r][c] = src[(ri ? ri[r] : r) * rs + (ci ? ci[c] : c) * cs]: transposes (strides), torch.index_select on
// either axis (int64 index lists, as inf_select_kernel writes them) or both, as one contiguous write
__global__ void __launch_bounds__(256)
gather2d_kernel(const float* __restrict__ src, long long rs, long long cs, const long long* __restrict__ ri,
                const long long* __restrict__ ci, float* __restrict__ out, int rows, int cols) {
  const long long total = (long long)rows * cols;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int r = (int)(e / cols), c = (int)(e - (long long)r * cols);
    const long long sr = ri ? ri[r] : r, sc = ci ? ci[c] : c;
    out[e] = src[sr * rs + sc * cs];
  }
}

static inline dim3 grid_for(long long count) {
  long long blocks = cdivll(count, 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  return dim3((unsigned)blocks);
}

}  // namespace curv

using namespace curv;

extern "C" int curv_inf_select(void* stream, const curv_select_desc* descs, int n_desc) {
  CURV_REQUIRE(n_desc >= 0 && (n_desc == 0 || descs), "curv_inf_select: bad arguments");
  for (int base = 0; base < n_desc; base += SEL_BATCH) {
    SelBatch batch;
    SelDev* d = batch.d;
    memset(&batch, 0, sizeof(batch));
    const int cnt = n_desc - base < SEL_BATCH ? n_desc - base : SEL_BATCH;
    for (int i = 0; i < cnt; ++i) {
      const curv_select_desc& s = descs[base + i];
      CURV_REQUIRE(s.lambda_vec && s.I && s.J && s.counts, "curv_inf_select: desc %d: null pointer", base + i);
      CURV_REQUIRE(s.n > 0 && s.m > 0 && s.n <= 8192 && s.m <= 8192 && s.rank > 0,
                   "curv_inf_select: desc %d: sizes out of range (n, m <= 8192)", base + i);
      d[i].lam = s.lambda_vec; d[i].I = (long long*)s.I; d[i].J = (long long*)s.J; d[i].counts = s.counts;
      d[i].n = s.n; d[i].m = s.m; d[i].rank = s.rank;
    }
    hipLaunchKernelGGL(inf_select_kernel, dim3(cnt), dim3(SEL_THREADS), 0, (hipStream_t)stream, batch, cnt);
    CURV_LAUNCH_CHECK();
  }
  return CURV_OK;
}

extern "C" int curv_colpairs(void* stream, const float* U, int n, int a, long long u_row_stride, float* out) {
  if (n <= 0 || a <= 0) return CURV_OK;
  CURV_REQUIRE(U && out, "curv_colpairs: null pointer");
  hipLaunchKernelGGL(colpairs_kernel<float>, grid_for((long long)n * a * a), dim3(256), 0, (hipStream_t)stream, U, n, a,
                     u_row_stride, out);
  CURV_LAUNCH_CHECK();
  return CURV_OK;
}

extern "C" int curv_colpairs_f64(void* stream, const float* U, int n, int a, long long u_row_stride, double* out) {
  if (n <= 0 || a <= 0) return CURV_OK;
  CURV_REQUIRE(U && out, "curv_colpairs_f64: null pointer");
  hipLaunchKernelGGL(colpairs_kernel<double>, grid_for((long long)n * a * a), dim3(256), 0, (hipStream_t)stream, U, n, a,
                     u_row_stride, out);
  CURV_LAUNCH_CHECK();
  return CURV_OK;
}

extern "C" int curv_square_f64(void* stream, const float* v, double* out, long long count) {
  if (count <= 0) return CURV_OK;
  CURV_REQUIRE(v && out, "curv_square_f64: null pointer");
  hipLaunchKernelGGL(square_f64_kernel, grid_for(count), dim3(256), 0, (hipStream_t)stream, v, out, count);
  CURV_LAUNCH_CHECK();
  return CURV_OK;
}

extern "C" int curv_diag_scale(void* stream, const void* src, int src_is_f64, float* dst, const float* dl,
                               const float* dr, int rows, int cols) {
  if (rows <= 0 || cols <= 0) return CURV_OK;
  CURV_REQUIRE(src && dst && dl && dr, "curv_diag_scale: null pointer");
  if (src_is_f64)
    hipLaunchKernelGGL(diag_scale_kernel<double>, grid_for((long long)rows * cols), dim3(256), 0, (hipStream_t)stream,
                       (const double*)src, dst, dl, dr, rows, cols);
  else
    hipLaunchKernelGGL(diag_scale_kernel<float>, grid_for((long long)rows * cols), dim3(256), 0, (hipStream_t)stream,
                       (const float*)src, dst, dl, dr, rows, cols);
  CURV_LAUNCH_CHECK();
  return CURV_OK;
}

extern "C" int curv_inf_vtv_assemble(void* stream, const float* V4, const float* sigma, int a, int b, float* vtv) {
  if (a <= 0 || b <= 0) return CURV_OK;
  CURV_REQUIRE(V4 && sigma && vtv, "curv_inf_vtv_assemble: null pointer");
  hipLaunchKernelGGL(vtv_assemble_kernel<float>, grid_for((long long)a * b * a * b), dim3(256), 0, (hipStream_t)stream, V4,
                     sigma, a, b, vtv);
  CURV_LAUNCH_CHECK();
  return CURV_OK;
}

extern "C" int curv_colpairs_sym_f64(void* stream, const float* U, int n, int a, long long u_row_stride, double* out) {
  if (n <= 0 || a <= 0) return CURV_OK;
  CURV_REQUIRE(U && out, "curv_colpairs_sym_f64: null pointer");
  hipLaunchKernelGGL(colpairs_sym_kernel, grid_for((long long)n * a * a), dim3(256), 0, (hipStream_t)stream, U, n, a,
                     u_row_stride, out);
  CURV_LAUNCH_CHECK();
  return CURV_OK;
}
extern "C" int curv_inf_vtv_assemble_sym_f64(void* stream, const double* V4p, const float* sigma, int a, int b, double* vtv) {
  if (a <= 0 || b <= 0) return CURV_OK;
  CURV_REQUIRE(V4p && sigma && vtv, "curv_inf_vtv_assemble_sym_f64: null pointer");
  hipLaunchKernelGGL(vtv_assemble_sym_kernel, grid_for((long long)a * b * a * b), dim3(256), 0, (hipStream_t)stream, V4p,
                     sigma, a, b, vtv);
  CURV_LAUNCH_CHECK();
  return CURV_OK;
}
extern "C" int curv_inf_vtv_assemble_f64(void* stream, const double* V4, const float* sigma, int a, int b, double* vtv) {
  if (a <= 0 || b <= 0) return CURV_OK;
  CURV_REQUIRE(V4 && sigma && vtv, "curv_inf_vtv_assemble_f64: null pointer");
  hipLaunchKernelGGL(vtv_assemble_kernel<double>, grid_for((long long)a * b * a * b), dim3(256), 0, (hipStream_t)stream, V4,
                     sigma, a, b, vtv);
  CURV_LAUNCH_CHECK();
  return CURV_OK;
}

extern "C" int curv_gather2d(void* stream, const float* src, long long src_rs, long long src_cs,
                             const int64_t* row_index, const int64_t* col_index, float* out, int rows, int cols) {
  if (rows <= 0 || cols <= 0) return CURV_OK;
  CURV_REQUIRE(src && out, "curv_gather2d: null pointer");
  hipLaunchKernelGGL(gather2d_kernel, grid_for((long long)rows * cols), dim3(256), 0, (hipStream_t)stream, src, src_rs,
                     src_cs, (const long long*)row_index, (const long long*)col_index, out, rows, cols);
  CURV_LAUNCH_CHECK();
  return CURV_OK;
}

extern "C" int curv_mul2d(void* stream, const float* A, long long a_rs, long long a_cs, const float* B,
                          long long b_rs, long long b_cs, float* out, int rows, int cols) {
  if (rows <= 0 || cols <= 0) return CURV_OK;
  CURV_REQUIRE(A && B && out, "curv_mul2d: null pointer");
  hipLaunchKernelGGL(mul2d_kernel, grid_for((long long)rows * cols), dim3(256), 0, (hipStream_t)stream, A, a_rs,
                     a_cs, B, b_rs, b_cs, out, rows, cols);
  CURV_LAUNCH_CHECK();
  return CURV_OK;
}
