// Development harness of the 16x16x4 LDS-DMA factor-build kernel (round 6): symmetric tile enumeration with BALANCED
// diagonal tiles.  A diagonal 128 x 128 tile has 36 upper 16 x 16 blocks: nine per wave (the 32x32x2 form gives its waves
// 3 / 2 / 2 / 3 of 4 blocks per step: the tile takes 3/4 of a full tile's time for 10/16 of its flops; here 9/16).
//   MODE 0: 32x32x2, roles 3/2/2/3 (the round-5 product kernel's arithmetic)     MODE 1: 16x16x4, nine blocks per wave
//   hipcc -O3 --offload-arch=gfx950 tools/micro/flat16_probe.hip -o tools/micro/flat16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <chrono>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(1))) float gfloat;

constexpr int THREADS = 256, TM = 128, ROW_B = 64, RPP = 16, PIECES = 2, PANEL_B = TM * ROW_B, LDS_B = 4 * PANEL_B, NP = 4;

struct Flat {
  const float* src;
  float* slabs;
  int N, C, HW, P, n_tiles, SPS, total_stages, spi, n_slices, n_items;
};

__device__ __forceinline__ void decode_tile(int t, int P, int& ti, int& tj) {
  ti = 0;
  while (t >= P - ti) { t -= P - ti; ++ti; }
  tj = ti + t;
}

// wave roles of the 16x16x4 form.  ROLE 0: off-diagonal tile, wave (wm, wn) owns the 4 x 4 blocks of its quadrant.
// ROLES 1..4: waves 0..3 of a diagonal tile: `rd` = the 16-row blocks of the (single) panel the wave reads, `pa` / `pb` =
// its nine block products as indices into `rd` (block row rd[pa], block column rd[pb], all on or above the diagonal).
template <int ROLE> struct Role;
template <> struct Role<0> { static constexpr int NRD = 8, NPAIR = 16; };
template <> struct Role<1> { static constexpr int NRD = 4, NPAIR = 9;
  static constexpr int rd[8] = {0, 1, 2, 3, 0, 0, 0, 0};
  static constexpr int pa[16] = {0, 0, 0, 0, 1, 1, 1, 2, 2}, pb[16] = {0, 1, 2, 3, 1, 2, 3, 2, 3}; };
template <> struct Role<2> { static constexpr int NRD = 7, NPAIR = 9;
  static constexpr int rd[8] = {0, 1, 4, 5, 6, 7, 3, 0};
  static constexpr int pa[16] = {0, 0, 0, 0, 1, 1, 1, 1, 6}, pb[16] = {2, 3, 4, 5, 2, 3, 4, 5, 6}; };
template <> struct Role<3> { static constexpr int NRD = 6, NPAIR = 9;
  static constexpr int rd[8] = {2, 3, 4, 5, 6, 7, 0, 0};
  static constexpr int pa[16] = {0, 0, 0, 0, 1, 1, 1, 1, 2}, pb[16] = {2, 3, 4, 5, 2, 3, 4, 5, 2}; };
template <> struct Role<4> { static constexpr int NRD = 4, NPAIR = 9;
  static constexpr int rd[8] = {4, 5, 6, 7, 0, 0, 0, 0};
  static constexpr int pa[16] = {0, 0, 0, 1, 1, 1, 2, 2, 3}, pb[16] = {1, 2, 3, 1, 2, 3, 2, 3, 3}; };

struct Ctx {
  int item, slice, ti, tj, wave, lane, i0, j0;
  bool diag;
};

template <typename Body>
__device__ __forceinline__ void stage_loop(const Flat& d, const Ctx& c, lds_char* lds, Body body) {
  const int HW = d.HW, C = d.C, wave = c.wave, lane = c.lane;
  const int rsub = RPP * wave + (lane >> 2);
  const int g_lane = (lane & 3) ^ ((rsub >> 2) & 3);
  const int voff = (rsub * HW + 4 * g_lane) * 4;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)d.src, 0, (unsigned)((long long)d.N * C * HW * 4), 0x00020000);
  const int t0 = c.slice * d.spi, t1 = min(t0 + d.spi, d.total_stages);
  const int n_panels = c.diag ? 1 : 2;
  int n_soff[2] = {0, 0};
  unsigned n_buf = 0;
  auto plan_next = [&](int t) {
    const int s = t / d.SPS, q = t - s * d.SPS;
    n_buf = (unsigned)(t & 1) * PANEL_B;
    n_soff[0] = ((s * C + c.i0) * HW + 16 * q) * 4;
    n_soff[1] = ((s * C + c.j0) * HW + 16 * q) * 4;
  };
  auto piece = [&](int i) {
    const int p = i / PIECES, slot = i % PIECES;
    if (p < n_panels) {
      const unsigned lbase = (p ? 2u * PANEL_B : 0u) + n_buf + (unsigned)(RPP * wave + 4 * RPP * slot) * ROW_B;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds + lbase), 16, voff, n_soff[p] + slot * 4 * RPP * HW * 4, 0, 0);
    }
  };
  plan_next(t0);
#pragma unroll
  for (int i = 0; i < NP; ++i) piece(i);
  for (int t = t0; t < t1; ++t) {
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    const bool more = t + 1 < t1;
    if (more) plan_next(t + 1);
    body((unsigned)(t & 1) * PANEL_B, [&](int e) { if (more) piece(e); });
  }
}

// MODE 1 body
template <int ROLE>
__device__ __forceinline__ void body16(const Flat& d, const Ctx& c, lds_char* lds) {
  using R = Role<ROLE>;
  constexpr int NRD = R::NRD, NPAIR = R::NPAIR;
  const int lane = c.lane, wm = c.wave >> 1, wn = c.wave & 1;
  const int r16 = lane & 15, kq = lane >> 4;
  unsigned addr[NRD];
#pragma unroll
  for (int x = 0; x < NRD; ++x) {
    int b; unsigned pbase = 0;
    if constexpr (ROLE == 0) { b = x < 4 ? 4 * wm + x : 4 * wn + (x - 4); pbase = x < 4 ? 0u : 2u * PANEL_B; }
    else b = R::rd[x];
    const int Rr = 16 * b + r16;
    addr[x] = pbase + Rr * ROW_B + ((kq ^ ((Rr >> 2) & 3)) << 4);
  }
  f32x4 acc[NPAIR];
#pragma unroll
  for (int p = 0; p < NPAIR; ++p) acc[p] = 0.0f;
  stage_loop(d, c, lds, [&](unsigned buf, auto hook) {
    f32x4 v[NRD];
#pragma unroll
    for (int x = 0; x < NRD; ++x) v[x] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(lds + addr[x] + buf);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int p = 0; p < NPAIR; ++p) {
        int xa, xb;
        if constexpr (ROLE == 0) { xa = p >> 2; xb = 4 + (p & 3); } else { xa = R::pa[p]; xb = R::pb[p]; }
        acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[xa][e], v[xb][e], acc[p], 0, 0, 0);
      }
      hook(e);
    }
  });
  gfloat* q = (gfloat*)d.slabs + (long long)c.item * (TM * TM);
#pragma unroll
  for (int p = 0; p < NPAIR; ++p) {
    int ba, bb;
    if constexpr (ROLE == 0) { ba = 4 * wm + (p >> 2); bb = 4 * wn + (p & 3); } else { ba = R::rd[R::pa[p]]; bb = R::rd[R::pb[p]]; }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) q[(16 * ba + 4 * kq + reg) * 128 + 16 * bb + r16] = acc[p][reg];
  }
}

// MODE 0 body: the round-5 arithmetic.  PART 0 all four blocks, 1 diagonal quadrant (upper three), 2 / 3 halves of quadrant (0, 1)
template <int PART>
__device__ __forceinline__ void body32(const Flat& d, const Ctx& c, lds_char* lds) {
  const int lane = c.lane;
  int wm = c.wave >> 1, wn = c.wave & 1;
  if (PART >= 2) { wm = 0; wn = 1; }
  const int r32 = lane & 31, h = lane >> 5;
  unsigned addr[4][2];
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    const int Rr = ((o < 2) ? 64 * wm : 64 * wn) + (o & 1) * 32 + r32;
    const unsigned pbase = (o < 2 || c.diag) ? 0u : 2u * PANEL_B;
#pragma unroll
    for (int j = 0; j < 2; ++j) addr[o][j] = pbase + Rr * ROW_B + (((2 * j + h) ^ ((Rr >> 2) & 3)) << 4);
  }
  f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};
  stage_loop(d, c, lds, [&](unsigned buf, auto hook) {
    auto rd = [&](int o, int j) { return *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(lds + addr[o][j] + buf); };
    f32x4 a0 = rd(0, 0), a1 = rd(1, 0), b0 = rd(2, 0), b1 = rd(3, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x4 na0, na1, nb0, nb1;
      if (j == 0) { na0 = rd(0, 1); na1 = rd(1, 1); nb0 = rd(2, 1); nb1 = rd(3, 1); }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (PART != 3) c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], c00, 0, 0, 0);
        if (PART != 2) c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b1[e], c01, 0, 0, 0);
        if (PART == 0 || PART == 2) c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b0[e], c10, 0, 0, 0);
        if (PART != 2) c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], c11, 0, 0, 0);
        if (j == 0) hook(e);
      }
      if (j == 0) { a0 = na0; a1 = na1; b0 = nb0; b1 = nb1; }
    }
  });
  gfloat* q = (gfloat*)d.slabs + (long long)c.item * (TM * TM) + (64 * wm) * 128 + 64 * wn;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
    if (PART != 3) q[row * 128 + r32] = c00[reg];
    if (PART != 2) q[row * 128 + 32 + r32] = c01[reg];
    if (PART != 3) q[(32 + row) * 128 + r32] = c10[reg];
    if (PART != 2) q[(32 + row) * 128 + 32 + r32] = c11[reg];
  }
}

template <int MODE>
__global__ void __launch_bounds__(THREADS, 4) probe_kernel(Flat d) {
  __shared__ __attribute__((aligned(1024))) char smem[LDS_B];
  lds_char* lds = (lds_char*)smem;
  Ctx c;
  {
    const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
    c.item = ((j / 32) * 8 + xcd) * 32 + (j % 32);
  }
  if (c.item >= d.n_items) return;
  c.lane = threadIdx.x & 63;
  c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  c.slice = c.item / d.n_tiles;
  decode_tile(c.item - c.slice * d.n_tiles, d.P, c.ti, c.tj);
  c.diag = c.ti == c.tj;
  c.i0 = c.ti * TM; c.j0 = c.tj * TM;
  int role = 0;
  if (c.diag) {
    if (MODE == 0) { const int wm = c.wave >> 1, wn = c.wave & 1; role = (wm == wn) ? 1 : (wm == 0 ? 2 : 3); }
    else role = 1 + c.wave;
  }
  role = __builtin_amdgcn_readfirstlane(role);
  if (MODE == 0) {
    if (role == 0) body32<0>(d, c, lds); else if (role == 1) body32<1>(d, c, lds);
    else if (role == 2) body32<2>(d, c, lds); else body32<3>(d, c, lds);
  } else {
    if (role == 0) body16<0>(d, c, lds); else if (role == 1) body16<1>(d, c, lds); else if (role == 2) body16<2>(d, c, lds);
    else if (role == 3) body16<3>(d, c, lds); else body16<4>(d, c, lds);
  }
}

__global__ void ref_kernel(const float* src, double* out, int N, int C, int HW, int i0, int j0) {
  const int i = i0 + blockIdx.x * 16 + threadIdx.x / 16, j = j0 + blockIdx.y * 16 + threadIdx.x % 16;
  double acc = 0;
  for (int s = 0; s < N; ++s) {
    const float* a = src + ((long long)s * C + i) * HW;
    const float* b = src + ((long long)s * C + j) * HW;
    for (int p = 0; p < HW; ++p) acc += (double)a[p] * b[p];
  }
  out[(i - i0) * TM + (j - j0)] = acc;
}


template <int MODE>
static double run(int N, int C, int HW, int target_items, float* src) {
  Flat d;
  memset(&d, 0, sizeof(d));
  d.N = N; d.C = C; d.HW = HW;
  d.P = C / TM; d.n_tiles = d.P * (d.P + 1) / 2;
  d.SPS = HW / 16;
  d.total_stages = N * d.SPS;
  int slices = std::max(1, std::min(d.total_stages, (target_items + d.n_tiles - 1) / d.n_tiles));
  d.spi = (d.total_stages + slices - 1) / slices;
  d.n_slices = (d.total_stages + d.spi - 1) / d.spi;
  d.n_items = d.n_slices * d.n_tiles;
  float* slabs;
  hipMalloc(&slabs, (size_t)d.n_items * TM * TM * 4);
  hipMemset(slabs, 0, (size_t)d.n_items * TM * TM * 4);
  d.src = src; d.slabs = slabs;
  const int grid = (d.n_items + 255) / 256 * 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe_kernel<MODE><<<grid, THREADS>>>(d);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); exit(1); }
  auto w0 = std::chrono::steady_clock::now();
  int warm = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() < 1.5) {
    for (int k = 0; k < 50; ++k) probe_kernel<MODE><<<grid, THREADS>>>(d);
    hipDeviceSynchronize();
    warm += 50;
  }
  const int reps = 40;
  hipEventRecord(e0);
  for (int k = 0; k < reps; ++k) probe_kernel<MODE><<<grid, THREADS>>>(d);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double flops = (double)C * (C + 1.0) * (double)N * HW;
  // correctness: the last diagonal tile and tile (0, P - 1), upper triangle
  double worst = 0;
  for (int which = 0; which < 2; ++which) {
    const int ti = which ? 0 : d.P - 1, tj = d.P - 1;
    int tile = 0;
    for (int a = 0; a < ti; ++a) tile += d.P - a;
    tile += tj - ti;
    std::vector<double> acc(TM * TM, 0.0), r(TM * TM);
    std::vector<float> part(TM * TM);
    for (int s = 0; s < d.n_slices; ++s) {
      hipMemcpy(part.data(), slabs + ((size_t)s * d.n_tiles + tile) * TM * TM, TM * TM * 4, hipMemcpyDeviceToHost);
      for (int e = 0; e < TM * TM; ++e) acc[e] += part[e];
    }
    double* ref;
    hipMalloc(&ref, TM * TM * 8);
    ref_kernel<<<dim3(8, 8), 256>>>(src, ref, N, C, HW, ti * TM, tj * TM);
    hipMemcpy(r.data(), ref, TM * TM * 8, hipMemcpyDeviceToHost);
    hipFree(ref);
    double num = 0, den = 0;
    for (int i = 0; i < TM; ++i)
      for (int j = 0; j < TM; ++j) {
        if (ti == tj && j < i) continue;
        num += (acc[i * TM + j] - r[i * TM + j]) * (acc[i * TM + j] - r[i * TM + j]); den += r[i * TM + j] * r[i * TM + j];
      }
    worst = std::max(worst, std::sqrt(num / den));
  }
  printf("%s C=%4d HW=%4d N=%d: tiles %d items %d (spi %d) warm %d  %.3f ms  %.1f TFLOP/s executed n(n+1)K (%.3f of 157.3)  rel err %.2e\n",
         MODE == 0 ? "32x32x2 3/2/2/3" : "16x16x4 9/9/9/9", C, HW, N, d.n_tiles, d.n_items, d.spi, warm, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3, worst);
  hipFree(slabs);
  return ms;
}

int main(int argc, char** argv) {
  const int target = argc > 1 ? atoi(argv[1]) : 2048;
  const int cases[][3] = {{1024, 784, 128}, {2048, 192, 512}, {512, 3136, 128}, {256, 3136, 128}, {4096, 64, 512}};
  const int first = argc > 2 ? atoi(argv[2]) : 0, last = argc > 3 ? atoi(argv[3]) : 5;
  for (int ci = first; ci < last && ci < 5; ++ci) {
    const int C = cases[ci][0], HW = cases[ci][1], N = cases[ci][2];
    const size_t elems = (size_t)N * C * HW;
    std::vector<float> h(elems);
    unsigned st = 12345u + C * 7 + HW;
    for (size_t i = 0; i < elems; ++i) { st = st * 1664525u + 1013904223u; h[i] = ((st >> 8) & 0xffff) / 65536.0f - 0.3f; }
    float* src;
    hipMalloc(&src, elems * 4);
    hipMemcpy(src, h.data(), elems * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
      run<0>(N, C, HW, target * (N / 32), src);
      run<1>(N, C, HW, target * (N / 32), src);
    }
    hipFree(src);
  }
  return 0;
}
