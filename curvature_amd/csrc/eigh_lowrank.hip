// curv_syevd: the eigensolver's entry point (utils.get_eigenvectors, curvature/utils.py:45-60).
//
// Wide symmetric matrices whose numerical rank is below half their width - the Kronecker factor of a layer with more rows
// than samples went into it (ResNet-50's 4608-wide factors at N = 32: rank 1568) - are decomposed WITHOUT iterating on their
// null space (round 5: Python glue over the ABI, ops._eigh_lowrank; round 6: here, so that every caller of the C entry
// point gets it).  With S = sym(F) in fp64:
//   Y = S Omega (n x r, r = n / 2 rounded down to 64, Omega Gaussian)        range of F, columns in general position
//   G = Y^T Y, Cholesky with a pivot threshold   -> k = numerical rank (the first pivot that falls to the threshold)
//   Q = orth(Y[:, :k]), Q = orth(S Q)            three Cholesky-QR passes each (G's condition is F's squared); one step of
//                                                subspace iteration
//   B = Q^T S Q (k x k);  accept iff ||S||^2 - ||B||^2 <= (3e-6 ||S||)^2        (P = Q Q^T is an orthogonal projector)
//   B = W diag(lam) W^T                          the block-Jacobi iteration (eigh.hip) on a matrix (k / n)^3 the size
//   U = [ Z | Q W ],  w = [ 0 | lam ], sorted    Z = an orthonormal basis of the complement of Q (Gaussian, projected,
//                                                two Cholesky-QR passes): S Z is below the residual bar by construction
// Every product is curv_gemm_f64_batched, every factorisation curv_chol_factor_inverse; a matrix that fails any test (rank
// >= n / 2 - 8, a failed factorisation, residual above the bar) is left to the iteration on the whole matrix, like every
// matrix narrower than 2048.  Deterministic: the Gaussian matrices depend on n only.
#include "common.h"

#include <algorithm>
#include <cstdlib>
#include <numeric>
#include <vector>

namespace curv {

constexpr double LOWRANK_PROBE = 0.5;          // columns of the range finder over the width: ranks up to this share take the path
constexpr int LOWRANK_MIN_N = 2048;            // narrower matrices converge in a few cheap sweeps anyway
constexpr double LOWRANK_GRAM_PIVOT = 1e-13;   // relative pivot of the Gram matrix below which a direction counts as noise (3e-7 of ||F||)
constexpr double LOWRANK_RESIDUAL = 3e-6;      // ||F - P F P|| / ||F|| the projection must reach (the iteration's own bar above 1024: 5e-6)
constexpr int LR_BLOCKS = 1024;                // partial sums of the squared norms (summed in a fixed order)

static double lowrank_probe() {
  const char* e = getenv("CURV_EIGH_PROBE");
  return e ? atof(e) : LOWRANK_PROBE;
}
static bool lowrank_enabled() {
  const char* e = getenv("CURV_EIGH_LOWRANK");
  return !(e && e[0] == '0' && e[1] == 0);
}
static int probe_columns(int n) { return (int)(n * lowrank_probe()) / 64 * 64; }

// S = (F + F^T) / 2 in fp64, and the block's share of sum S^2
__global__ void __launch_bounds__(256) lr_sym_f32_kernel(const float* __restrict__ F, double* __restrict__ S, int n,
                                                         double* __restrict__ partial) {
  __shared__ double red[256];
  const long long total = (long long)n * n;
  double acc = 0.0;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int i = (int)(e / n), j = (int)(e - (long long)i * n);
    const double v = ((double)F[e] + (double)F[(long long)j * n + i]) * 0.5;
    S[e] = v;
    acc += v * v;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
// B <- (B + B^T) / 2 (out of place), its fp32 copy, and the block's share of sum B^2
__global__ void __launch_bounds__(256) lr_sym_f64_kernel(const double* __restrict__ Bin, double* __restrict__ Bout,
                                                         float* __restrict__ B32, int k, double* __restrict__ partial) {
  __shared__ double red[256];
  const long long total = (long long)k * k;
  double acc = 0.0;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int i = (int)(e / k), j = (int)(e - (long long)i * k);
    const double v = (Bin[e] + Bin[(long long)j * k + i]) * 0.5;
    Bout[e] = v;
    B32[e] = (float)v;
    acc += v * v;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
// out[0] = sum of `count` partial sums, in index order
__global__ void __launch_bounds__(256) lr_sum_kernel(const double* __restrict__ partial, int count, double* __restrict__ out) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int e = threadIdx.x; e < count; e += 256) acc += partial[e];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}
// out[0] = max of the diagonal of an (r x r) matrix
__global__ void __launch_bounds__(256) lr_diag_max_kernel(const double* __restrict__ G, int r, double* __restrict__ out) {
  __shared__ double red[256];
  double m = -1.0e300;
  for (int e = threadIdx.x; e < r; e += 256) m = fmax(m, G[(long long)e * r + e]);
  red[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}
__global__ void __launch_bounds__(256) lr_widen_kernel(const float* __restrict__ src, double* __restrict__ dst, long long count) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < count; e += (long long)gridDim.x * 256) dst[e] = (double)src[e];
}
// U[:, j] = column order[j] of [ Z (n x (n - k), fp64) | QW (n x k, fp32) ],  w[j] = the matching entry of [ 0 | lam ]
__global__ void __launch_bounds__(256) lr_assemble_kernel(float* __restrict__ U, float* __restrict__ w, const double* __restrict__ Z,
                                                          const float* __restrict__ QW, const float* __restrict__ lam,
                                                          const int* __restrict__ order, int n, int k) {
  const long long total = (long long)n * n;
  const int nz = n - k;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int i = (int)(e / n), j = (int)(e - (long long)i * n);
    const int c = order[j];
    U[e] = c < nz ? (float)Z[(long long)i * nz + c] : QW[(long long)i * k + (c - nz)];
    if (i == 0 && w != nullptr) w[j] = c < nz ? 0.0f : lam[c - nz];
  }
}

// workspace carving: everything 256-byte aligned
struct Bump {
  char* base;
  size_t off;
  template <typename T> T* take(size_t count) {
    off = align_up(off, 256);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += count * sizeof(T);
    return p;
  }
};
// one matrix on the projection path
struct LowRank {
  int pos;                 // position in the caller's descriptor array
  int n, r, k;
  const float* F;
  float* U;
  float* w;
  double *S, *Qa, *Qb, *SQ, *gram, *X, *B, *Bs, *W64, *scal;
  float *B32, *W32, *lam, *QW32;
  int* order;
  // first-phase buffers (Omega, Y, G, its factor) and third-phase buffers (Z, T, its Gram matrix, the assembled basis) share
  // one region: the first are dead when the third begin
  char* scratch;
  float* Om32; double *Om, *Y, *G, *XG;
  float* Z32; double *Zb, *T, *gramZ, *XZ;
  double* Za;              // (aliases S: dead once B is formed)
  double nS2;
  bool alive;
};
static size_t lowrank_scratch_bytes(size_t n, size_t r) {
  const size_t a = align_up(n * r * 4, 256) + 2 * align_up(n * r * 8, 256) + 2 * align_up(r * r * 8, 256);
  const size_t c = align_up(n * n * 4, 256) + align_up(n * n * 8, 256) + align_up(r * n * 8, 256) + 2 * align_up(n * n * 8, 256);
  return std::max(a, c);
}
static void lowrank_carve(Bump& m, LowRank& q) {
  const size_t n = q.n, r = q.r;
  q.S = m.take<double>(n * n);
  q.Qa = m.take<double>(n * r); q.Qb = m.take<double>(n * r); q.SQ = m.take<double>(n * r);
  q.gram = m.take<double>(r * r); q.X = m.take<double>(r * r);
  q.B = m.take<double>(r * r); q.Bs = m.take<double>(r * r); q.W64 = m.take<double>(r * r);
  q.B32 = m.take<float>(r * r); q.W32 = m.take<float>(r * r); q.lam = m.take<float>(r); q.QW32 = m.take<float>(n * r);
  q.order = m.take<int>(n);
  q.scal = m.take<double>(LR_BLOCKS + 8);
  q.scratch = m.take<char>(lowrank_scratch_bytes(n, r));
  Bump a{q.scratch, 0};
  q.Om32 = a.take<float>(n * r); q.Om = a.take<double>(n * r); q.Y = a.take<double>(n * r);
  q.G = a.take<double>(r * r); q.XG = a.take<double>(r * r);
  Bump c{q.scratch, 0};
  q.Z32 = c.take<float>(n * n); q.Zb = c.take<double>(n * n); q.T = c.take<double>(r * n);
  q.gramZ = c.take<double>(n * n); q.XZ = c.take<double>(n * n);
  q.Za = q.S;
}

static curv_gemm64_desc gemm64(const double* A, long long a_rs, long long a_cs, const double* B, long long b_rs, long long b_cs,
                               double* C, long long c_rs, int M, int N, int K, int tri = 0, double alpha = 1.0, double beta = 0.0,
                               const double* E = nullptr) {
  curv_gemm64_desc d;
  memset(&d, 0, sizeof(d));
  d.A = A; d.B = B; d.C = C;
  d.a_rs = a_rs; d.a_cs = a_cs; d.b_rs = b_rs; d.b_cs = b_cs; d.c_rs = c_rs; d.c_cs = 1;
  d.M = M; d.N = N; d.K = K; d.tri = tri; d.alpha = alpha; d.beta = beta; d.E = E;
  return d;
}

// the wide matrices of a call: sizes of the shared pieces behind the per-matrix regions
struct LowRankShared { size_t chol_bytes, jacobi_bytes; };
static LowRankShared lowrank_shared(const std::vector<LowRank>& ms) {
  // the largest factorisation batch is the complement's Gram matrices ((n - k)^2 <= n^2 each), the largest Jacobi batch the
  // projected problems (k <= r)
  std::vector<curv_cholinv_desc> cd(ms.size());
  std::vector<curv_eigh_desc> ed(ms.size());
  for (size_t i = 0; i < ms.size(); ++i) {
    memset(&cd[i], 0, sizeof(cd[i]));
    cd[i].n = ms[i].n; cd[i].m_is_f64 = 1;
    memset(&ed[i], 0, sizeof(ed[i]));
    ed[i].n = ms[i].r;
  }
  LowRankShared s;
  s.chol_bytes = curv_chol_factor_inverse_workspace_bytes(cd.data(), (int)cd.size());
  s.jacobi_bytes = syevd_jacobi_workspace_bytes(ed.data(), (int)ed.size());
  return s;
}

static std::vector<LowRank> lowrank_candidates(const curv_eigh_desc* descs, int n_mats) {
  std::vector<LowRank> ms;
  if (!lowrank_enabled()) return ms;
  for (int i = 0; i < n_mats; ++i) {
    if (descs[i].n < LOWRANK_MIN_N) continue;
    LowRank q;
    memset(&q, 0, sizeof(q));
    q.pos = i; q.n = descs[i].n; q.r = probe_columns(q.n);
    if (q.r < 64) continue;
    q.F = descs[i].F; q.U = descs[i].U; q.w = descs[i].w;
    q.alive = true;
    ms.push_back(q);
  }
  return ms;
}
static size_t lowrank_workspace_bytes(const curv_eigh_desc* descs, int n_mats) {
  std::vector<LowRank> ms = lowrank_candidates(descs, n_mats);
  if (ms.empty()) return 0;
  Bump m{nullptr, 0};
  for (LowRank& q : ms) lowrank_carve(m, q);
  const LowRankShared s = lowrank_shared(ms);
  m.take<int>(ms.size() + 64);
  m.take<char>(s.chol_bytes);
  m.take<char>(s.jacobi_bytes);
  return align_up(m.off, 256) + 256;
}

// Cholesky-QR of the columns `c` (n x k, row pitch ldc) of every live matrix: gram = c^T c (unless given), X = chol(gram)^-1,
// out = c X^T.  A matrix whose Gram matrix is not positive definite drops out.
static int lowrank_cholqr(hipStream_t stream, std::vector<LowRank*>& live, const std::vector<const double*>& c,
                          const std::vector<long long>& ldc, const std::vector<int>& cols, const std::vector<double*>& gram,
                          const std::vector<double*>& X, const std::vector<double*>& out, bool grams_given, int* info_dev,
                          void* chol_ws, size_t chol_bytes, std::vector<char>& ok) {
  const int m = (int)live.size();
  ok.assign(m, 1);
  if (m == 0) return CURV_OK;
  std::vector<curv_gemm64_desc> jobs;
  if (!grams_given) {
    for (int i = 0; i < m; ++i)
      jobs.push_back(gemm64(c[i], 1, ldc[i], c[i], ldc[i], 1, gram[i], cols[i], cols[i], cols[i], live[i]->n));
    const int rc = curv_gemm_f64_batched(stream, jobs.data(), m);
    if (rc != CURV_OK) return rc;
  }
  std::vector<curv_cholinv_desc> cd(m);
  for (int i = 0; i < m; ++i) {
    memset(&cd[i], 0, sizeof(cd[i]));
    cd[i].M = gram[i]; cd[i].X = X[i]; cd[i].n = cols[i]; cd[i].m_is_f64 = 1; cd[i].diag_add = 0.0;
  }
  int rc = curv_chol_factor_inverse(stream, cd.data(), m, info_dev, chol_ws, chol_bytes);
  if (rc != CURV_OK) return rc;
  std::vector<int> info(m);
  CURV_HIP_CHECK(hipMemcpyAsync(info.data(), info_dev, (size_t)m * sizeof(int), hipMemcpyDeviceToHost, stream));
  CURV_HIP_CHECK(hipStreamSynchronize(stream));
  jobs.clear();
  for (int i = 0; i < m; ++i)     // out = c X^T: X^T is upper triangular
    jobs.push_back(gemm64(c[i], ldc[i], 1, X[i], 1, cols[i], out[i], cols[i], live[i]->n, cols[i], cols[i], CURV_TRI64_B_UPPER));
  rc = curv_gemm_f64_batched(stream, jobs.data(), m);
  if (rc != CURV_OK) return rc;
  for (int i = 0; i < m; ++i) ok[i] = info[i] == 0;
  return CURV_OK;
}

// the matrices that passed the last test stay on the path; the others are left to the iteration on the whole matrix
static void keep_live(std::vector<LowRank*>& live, const std::vector<char>& ok) {
  std::vector<LowRank*> next;
  for (size_t i = 0; i < live.size(); ++i) {
    if (ok[i]) next.push_back(live[i]);
    else live[i]->alive = false;
  }
  live.swap(next);
}

// The projection path on the wide matrices of a call; on return q.alive says which were decomposed (their U / w are written).
static int lowrank_run(hipStream_t stream, std::vector<LowRank>& ms, char* ws, int* sweeps_done) {
  Bump mem{ws, 0};
  for (LowRank& q : ms) lowrank_carve(mem, q);
  const LowRankShared sh = lowrank_shared(ms);
  int* info_dev = mem.take<int>(ms.size() + 64);
  void* chol_ws = mem.take<char>(sh.chol_bytes);
  void* jac_ws = mem.take<char>(sh.jacobi_bytes);
  const int M = (int)ms.size();
  std::vector<curv_gemm64_desc> jobs;
  // S = sym(F), ||S||^2, Y = S Omega, G = Y^T Y
  for (LowRank& q : ms) {
    hipLaunchKernelGGL(lr_sym_f32_kernel, dim3(LR_BLOCKS), dim3(256), 0, stream, q.F, q.S, q.n, q.scal + 8);
    CURV_LAUNCH_CHECK();
    hipLaunchKernelGGL(lr_sum_kernel, dim3(1), dim3(256), 0, stream, (const double*)(q.scal + 8), LR_BLOCKS, q.scal);
    CURV_LAUNCH_CHECK();
    int rc = curv_randn(stream, q.Om32, (long long)q.n * q.r, 0x5EED0000ull + (unsigned long long)q.n, 0);
    if (rc != CURV_OK) return rc;
    hipLaunchKernelGGL(lr_widen_kernel, dim3(2048), dim3(256), 0, stream, (const float*)q.Om32, q.Om, (long long)q.n * q.r);
    CURV_LAUNCH_CHECK();
    jobs.push_back(gemm64(q.S, q.n, 1, q.Om, q.r, 1, q.Y, q.r, q.n, q.r, q.n));
  }
  int rc = curv_gemm_f64_batched(stream, jobs.data(), M);
  if (rc != CURV_OK) return rc;
  jobs.clear();
  for (LowRank& q : ms) jobs.push_back(gemm64(q.Y, 1, q.r, q.Y, q.r, 1, q.G, q.r, q.r, q.r, q.n));
  rc = curv_gemm_f64_batched(stream, jobs.data(), M);
  if (rc != CURV_OK) return rc;
  for (LowRank& q : ms) {
    hipLaunchKernelGGL(lr_diag_max_kernel, dim3(1), dim3(256), 0, stream, (const double*)q.G, q.r, q.scal + 1);
    CURV_LAUNCH_CHECK();
  }
  std::vector<double> head(2 * M);
  for (int i = 0; i < M; ++i)
    CURV_HIP_CHECK(hipMemcpyAsync(&head[2 * i], ms[i].scal, 2 * sizeof(double), hipMemcpyDeviceToHost, stream));
  CURV_HIP_CHECK(hipStreamSynchronize(stream));
  {
    std::vector<curv_cholinv_desc> cd(M);
    for (int i = 0; i < M; ++i) {
      ms[i].nS2 = head[2 * i];
      memset(&cd[i], 0, sizeof(cd[i]));
      cd[i].M = ms[i].G; cd[i].X = ms[i].XG; cd[i].n = ms[i].r; cd[i].m_is_f64 = 1;
      cd[i].pivot_min = head[2 * i + 1] * LOWRANK_GRAM_PIVOT;
    }
    rc = curv_chol_factor_inverse(stream, cd.data(), M, info_dev, chol_ws, sh.chol_bytes);
    if (rc != CURV_OK) return rc;
  }
  std::vector<int> info(M);
  CURV_HIP_CHECK(hipMemcpyAsync(info.data(), info_dev, (size_t)M * sizeof(int), hipMemcpyDeviceToHost, stream));
  CURV_HIP_CHECK(hipStreamSynchronize(stream));
  std::vector<LowRank*> live;
  for (int i = 0; i < M; ++i) {
    LowRank& q = ms[i];
    q.k = info[i] > 0 ? info[i] - 1 : q.r;
    if (info[i] >= 0 && q.k >= 16 && q.k < q.r - 8) live.push_back(&q);
    else q.alive = false;
  }
  if (live.empty()) return CURV_OK;
  std::vector<char> ok;
  auto vec = [&](auto fn) { std::vector<decltype(fn(*live[0]))> v; for (LowRank* q : live) v.push_back(fn(*q)); return v; };
  // Q = orth(Y[:, :k]): the Gram matrix of the first pass is the leading block of G
  for (LowRank* q : live)
    CURV_HIP_CHECK(hipMemcpy2DAsync(q->gram, (size_t)q->k * 8, q->G, (size_t)q->r * 8, (size_t)q->k * 8, q->k, hipMemcpyDeviceToDevice, stream));
  rc = lowrank_cholqr(stream, live, vec([](LowRank& q) { return (const double*)q.Y; }), vec([](LowRank& q) { return (long long)q.r; }),
                      vec([](LowRank& q) { return q.k; }), vec([](LowRank& q) { return q.gram; }), vec([](LowRank& q) { return q.X; }),
                      vec([](LowRank& q) { return q.Qa; }), true, info_dev, chol_ws, sh.chol_bytes, ok);
  if (rc != CURV_OK) return rc;
  // (the current basis of a matrix is Qa, the other buffer Qb; swapped after every pass)
  auto pass = [&]() -> int {
    keep_live(live, ok);
    if (live.empty()) return CURV_OK;
    const int r2 = lowrank_cholqr(stream, live, vec([](LowRank& q) { return (const double*)q.Qa; }), vec([](LowRank& q) { return (long long)q.k; }),
                                  vec([](LowRank& q) { return q.k; }), vec([](LowRank& q) { return q.gram; }),
                                  vec([](LowRank& q) { return q.X; }), vec([](LowRank& q) { return q.Qb; }), false, info_dev, chol_ws,
                                  sh.chol_bytes, ok);
    for (LowRank* q : live) std::swap(q->Qa, q->Qb);
    return r2;
  };
  for (int it = 0; it < 2; ++it) {
    rc = pass();
    if (rc != CURV_OK) return rc;
    if (live.empty()) return CURV_OK;
  }
  keep_live(live, ok);
  if (live.empty()) return CURV_OK;
  // one step of subspace iteration, Q <- orth(S Q): without it the basis of a matrix whose rank was hit exactly (no
  // oversampling: a sharp drop of the pivots) is only as good as the k x k Gaussian mixing matrix is conditioned
  jobs.clear();
  for (LowRank* q : live) jobs.push_back(gemm64(q->S, q->n, 1, q->Qa, q->k, 1, q->Qb, q->k, q->n, q->k, q->n));
  rc = curv_gemm_f64_batched(stream, jobs.data(), (int)live.size());
  if (rc != CURV_OK) return rc;
  for (LowRank* q : live) std::swap(q->Qa, q->Qb);
  ok.assign(live.size(), 1);
  for (int it = 0; it < 3; ++it) {
    rc = pass();
    if (rc != CURV_OK) return rc;
    if (live.empty()) return CURV_OK;
    keep_live(live, ok);
    if (live.empty()) return CURV_OK;
    ok.assign(live.size(), 1);
  }
  // B = Q^T S Q, the residual of the projection
  jobs.clear();
  for (LowRank* q : live) jobs.push_back(gemm64(q->S, q->n, 1, q->Qa, q->k, 1, q->SQ, q->k, q->n, q->k, q->n));
  rc = curv_gemm_f64_batched(stream, jobs.data(), (int)live.size());
  if (rc != CURV_OK) return rc;
  jobs.clear();
  for (LowRank* q : live) jobs.push_back(gemm64(q->Qa, 1, q->k, q->SQ, q->k, 1, q->B, q->k, q->k, q->k, q->n));
  rc = curv_gemm_f64_batched(stream, jobs.data(), (int)live.size());
  if (rc != CURV_OK) return rc;
  std::vector<double> nb2(live.size());
  for (size_t i = 0; i < live.size(); ++i) {
    LowRank* q = live[i];
    hipLaunchKernelGGL(lr_sym_f64_kernel, dim3(LR_BLOCKS), dim3(256), 0, stream, (const double*)q->B, q->Bs, q->B32, q->k, q->scal + 8);
    CURV_LAUNCH_CHECK();
    hipLaunchKernelGGL(lr_sum_kernel, dim3(1), dim3(256), 0, stream, (const double*)(q->scal + 8), LR_BLOCKS, q->scal + 2);
    CURV_LAUNCH_CHECK();
    CURV_HIP_CHECK(hipMemcpyAsync(&nb2[i], q->scal + 2, sizeof(double), hipMemcpyDeviceToHost, stream));
  }
  CURV_HIP_CHECK(hipStreamSynchronize(stream));
  for (size_t i = 0; i < live.size(); ++i) {
    const double res2 = live[i]->nS2 - nb2[i], bar = live[i]->nS2 * (LOWRANK_RESIDUAL * LOWRANK_RESIDUAL);
    ok[i] = res2 <= bar;
  }
  keep_live(live, ok);
  if (live.empty()) return CURV_OK;
  // the small problems: the block-Jacobi iteration (they have full rank by construction)
  {
    std::vector<curv_eigh_desc> ed(live.size());
    for (size_t i = 0; i < live.size(); ++i) {
      memset(&ed[i], 0, sizeof(ed[i]));
      ed[i].F = live[i]->B32; ed[i].U = live[i]->W32; ed[i].w = live[i]->lam; ed[i].n = live[i]->k;
    }
    int sw = 0;
    rc = syevd_jacobi(stream, ed.data(), (int)ed.size(), jac_ws, sh.jacobi_bytes, 0, 0.0, &sw);
    if (sweeps_done) *sweeps_done = std::max(*sweeps_done, sw);
    if (rc != CURV_OK) return rc;
  }
  // complement of Q: Gaussian, projected, orthonormalised twice.  (Z alternates between Za - the dead S - and Zb.)
  for (LowRank* q : live) {
    const long long cnt = (long long)q->n * (q->n - q->k);
    rc = curv_randn(stream, q->Z32, cnt, 0x5EED8000ull + (unsigned long long)q->n, 0);
    if (rc != CURV_OK) return rc;
    hipLaunchKernelGGL(lr_widen_kernel, dim3(2048), dim3(256), 0, stream, (const float*)q->Z32, q->Za, cnt);
    CURV_LAUNCH_CHECK();
  }
  for (int it = 0; it < 2; ++it) {
    jobs.clear();
    for (LowRank* q : live) jobs.push_back(gemm64(q->Qa, 1, q->k, q->Za, q->n - q->k, 1, q->T, q->n - q->k, q->k, q->n - q->k, q->n));
    rc = curv_gemm_f64_batched(stream, jobs.data(), (int)live.size());
    if (rc != CURV_OK) return rc;
    jobs.clear();
    for (LowRank* q : live)
      jobs.push_back(gemm64(q->Qa, q->k, 1, q->T, q->n - q->k, 1, q->Zb, q->n - q->k, q->n, q->n - q->k, q->k, 0, -1.0, 1.0, q->Za));
    rc = curv_gemm_f64_batched(stream, jobs.data(), (int)live.size());
    if (rc != CURV_OK) return rc;
    rc = lowrank_cholqr(stream, live, vec([](LowRank& q) { return (const double*)q.Zb; }), vec([](LowRank& q) { return (long long)(q.n - q.k); }),
                        vec([](LowRank& q) { return q.n - q.k; }), vec([](LowRank& q) { return q.gramZ; }),
                        vec([](LowRank& q) { return q.XZ; }), vec([](LowRank& q) { return q.Za; }), false, info_dev, chol_ws, sh.chol_bytes, ok);
    if (rc != CURV_OK) return rc;
    keep_live(live, ok);
    if (live.empty()) return CURV_OK;
  }
  // U = [ Z | Q W ] in ascending order of [ 0 | lam ] (stable: the order torch.sort(stable=True) gives)
  jobs.clear();
  for (LowRank* q : live) {
    hipLaunchKernelGGL(lr_widen_kernel, dim3(2048), dim3(256), 0, stream, (const float*)q->W32, q->W64, (long long)q->k * q->k);
    CURV_LAUNCH_CHECK();
    curv_gemm64_desc d = gemm64(q->Qa, q->k, 1, q->W64, q->k, 1, nullptr, q->k, q->n, q->k, q->k);
    d.C32 = q->QW32;
    jobs.push_back(d);
  }
  rc = curv_gemm_f64_batched(stream, jobs.data(), (int)live.size());
  if (rc != CURV_OK) return rc;
  std::vector<std::vector<float>> lam(live.size());
  for (size_t i = 0; i < live.size(); ++i) {
    lam[i].resize(live[i]->k);
    CURV_HIP_CHECK(hipMemcpyAsync(lam[i].data(), live[i]->lam, (size_t)live[i]->k * sizeof(float), hipMemcpyDeviceToHost, stream));
  }
  CURV_HIP_CHECK(hipStreamSynchronize(stream));
  std::vector<std::vector<int>> order(live.size());
  for (size_t i = 0; i < live.size(); ++i) {
    LowRank* q = live[i];
    const int n = q->n, nz = n - q->k;
    std::vector<float> wcat(n, 0.0f);
    for (int j = 0; j < q->k; ++j) wcat[nz + j] = lam[i][j];
    order[i].resize(n);
    std::iota(order[i].begin(), order[i].end(), 0);
    std::stable_sort(order[i].begin(), order[i].end(), [&](int a, int b) { return wcat[a] < wcat[b]; });
    CURV_HIP_CHECK(hipMemcpyAsync(q->order, order[i].data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(lr_assemble_kernel, dim3(2048), dim3(256), 0, stream, q->U, q->w, (const double*)q->Za, (const float*)q->QW32,
                       (const float*)q->lam, (const int*)q->order, n, q->k);
    CURV_LAUNCH_CHECK();
  }
  CURV_HIP_CHECK(hipStreamSynchronize(stream));      // (`order` is host memory of this frame)
  return CURV_OK;
}

}  // namespace curv

using namespace curv;

extern "C" size_t curv_syevd_workspace_bytes(const curv_eigh_desc* descs, int n_mats) {
  const size_t jac = syevd_jacobi_workspace_bytes(descs, n_mats);
  if (jac == 0) return 0;
  return align_up(jac, 256) + lowrank_workspace_bytes(descs, n_mats);
}

extern "C" int curv_syevd_ex(void* stream_, const curv_eigh_desc* descs, int n_mats, void* workspace, size_t workspace_bytes,
                             int max_sweeps, double tol, int* sweeps_done, int* ranks) {
  hipStream_t stream = (hipStream_t)stream_;
  if (sweeps_done) *sweeps_done = 0;
  if (ranks) for (int i = 0; i < n_mats; ++i) ranks[i] = 0;
  if (n_mats == 0) return CURV_OK;
  CURV_REQUIRE(descs != nullptr, "curv_syevd: null descriptor array");
  const size_t jac = syevd_jacobi_workspace_bytes(descs, n_mats);
  CURV_REQUIRE(jac != 0, "curv_syevd: matrix size out of range");
  std::vector<char> decomposed(n_mats, 0);
  // the projection applies to the default iteration only (an explicit tolerance or sweep count asks for the plain one)
  if (max_sweeps <= 0 && tol <= 0.0) {
    std::vector<LowRank> ms = lowrank_candidates(descs, n_mats);
    if (!ms.empty()) {
      const size_t need = align_up(jac, 256) + lowrank_workspace_bytes(descs, n_mats);
      if (workspace == nullptr || workspace_bytes < need) {
        set_error("curv_syevd: workspace too small (%zu < %zu bytes)", workspace_bytes, need);
        return CURV_ERR_WORKSPACE;
      }
      for (const LowRank& q : ms) CURV_REQUIRE(q.F != nullptr && q.U != nullptr, "curv_syevd: matrix %d: null pointer", q.pos);
      int sw = 0;
      const int rc = lowrank_run(stream, ms, reinterpret_cast<char*>(workspace) + align_up(jac, 256), &sw);
      if (rc != CURV_OK) return rc;
      if (sweeps_done) *sweeps_done = sw;
      for (const LowRank& q : ms)
        if (q.alive) { decomposed[q.pos] = 1; if (ranks) ranks[q.pos] = q.k; }
    }
  }
  std::vector<curv_eigh_desc> rest;
  for (int i = 0; i < n_mats; ++i) if (!decomposed[i]) rest.push_back(descs[i]);
  if (rest.empty()) return CURV_OK;
  int sw = 0;
  const int rc = syevd_jacobi(stream, rest.data(), (int)rest.size(), workspace, workspace_bytes, max_sweeps, tol, &sw);
  if (sweeps_done) *sweeps_done = std::max(*sweeps_done, sw);
  return rc;
}

extern "C" int curv_syevd(void* stream, const curv_eigh_desc* descs, int n_mats, void* workspace, size_t workspace_bytes,
                          int max_sweeps, double tol, int* sweeps_done) {
  return curv_syevd_ex(stream, descs, n_mats, workspace, workspace_bytes, max_sweeps, tol, sweeps_done, nullptr);
}
