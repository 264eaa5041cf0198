// SURVEY section 7, H1 decided by measurement: the LDS-DMA factor-build stage on SPLIT-bf16 operands instead of native fp32 MFMA.
//   x = hi + mid + lo  (three bf16 planes: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid))
//   3 products: hi hi + hi mid + mid hi                     (relative error of a product ~2^-16)
//   6 products: + mid mid + hi lo + lo hi                   (~2^-24: fp32 level)
// v_mfma_f32_32x32x16_bf16 runs at 16x the rate of v_mfma_f32_32x32x2_f32 per flop, so 6 passes cost 0.375 of the fp32 MFMA
// time - at 6 instead of 4 operand bytes per element, and behind a pass that writes the planes (timed here too).
// Same harness as flat_shape_probe.hip: full 128 x 128 tiles X_i X_j^T of one flattened factor, inputs streamed from HBM,
// ~1.5 s of back-to-back launches before the timed ones; error of one tile against an fp64 reference.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/micro/flat_bf16_probe.hip -o tools/micro/flat_bf16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <chrono>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(1))) float gfloat;

constexpr int THREADS = 256, TM = 128, KC = 16, ROW_B = 32, IMG_B = TM * ROW_B;     // one plane of one panel per stage: 4 KiB

struct Flat {
  const __bf16* planes;      // [plane][N][C][HW]
  float* slabs;
  long long plane_elems;
  int N, C, HW, P, n_tiles, SPS, total_stages, spi, n_slices, n_items;
};

__global__ void split_kernel(const float* __restrict__ x, __bf16* __restrict__ planes, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  const __bf16 hi = (__bf16)v;
  const float r1 = v - (float)hi;
  const __bf16 mid = (__bf16)r1;
  const float r2 = r1 - (float)mid;
  planes[i] = hi; planes[n + i] = mid; planes[2 * n + i] = (__bf16)r2;
}

template <int NPL>
__device__ __forceinline__ void probe_body(const Flat& d) {
  // LDS: [buffer][panel][plane][128 rows x 32 B]
  __shared__ __attribute__((aligned(1024))) char smem[2 * 2 * NPL * IMG_B];
  lds_char* lds = (lds_char*)smem;
  int item;
  {
    const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
    item = ((j / 32) * 8 + xcd) * 32 + (j % 32);
  }
  if (item >= d.n_items) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, h = lane >> 5;
  const int slice = item / d.n_tiles, tile = item - slice * d.n_tiles;
  const int ti = tile / d.P, tj = tile - ti * d.P;
  const int i0 = ti * TM, j0 = tj * TM, HW = d.HW, C = d.C;
  // DMA: piece = 32 rows x 32 B of one plane of one panel; this wave moves piece `wave` of every (panel, plane).  Lane ->
  // row 32 wave + (lane >> 1), physical half (lane & 1) holds logical half (lane & 1) ^ ((row >> 3) & 1)
  const int drow = 32 * wave + (lane >> 1);
  const int lhalf = (lane & 1) ^ ((drow >> 3) & 1);
  const int voff = (drow * HW + 8 * lhalf) * 2;
  __amdgpu_buffer_rsrc_t rs[NPL];
#pragma unroll
  for (int p = 0; p < NPL; ++p)
    rs[p] = __builtin_amdgcn_make_buffer_rsrc((void*)(d.planes + p * d.plane_elems), 0, (unsigned)(d.plane_elems * 2), 0x00020000);
  auto issue = [&](int t, unsigned buf) {
    const int s = t / d.SPS, q = t - s * d.SPS;
    const int sa = ((s * C + i0) * HW + KC * q) * 2, sb = ((s * C + j0) * HW + KC * q) * 2;
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[p], (lds_void*)(lds + buf + (0 * NPL + p) * IMG_B + wave * 1024), 16, voff, sa, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[p], (lds_void*)(lds + buf + (1 * NPL + p) * IMG_B + wave * 1024), 16, voff, sb, 0, 0);
    }
  };
  unsigned addr_a[2], addr_b[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int ra = 64 * wm + 32 * m + r32, rb = 64 * wn + 32 * m + r32;
    addr_a[m] = ra * ROW_B + ((h ^ ((ra >> 3) & 1)) << 4);
    addr_b[m] = NPL * IMG_B + rb * ROW_B + ((h ^ ((rb >> 3) & 1)) << 4);
  }
  f32x16 c[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) c[m][n] = 0.0f;
  const int t0 = slice * d.spi, t1 = min(t0 + d.spi, d.total_stages);
  constexpr unsigned BUF_B = 2 * NPL * IMG_B;
  issue(t0, 0);
  for (int t = t0; t < t1; ++t) {
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    const unsigned buf = (unsigned)((t - t0) & 1) * BUF_B;
    if (t + 1 < t1) issue(t + 1, BUF_B - buf);
    bf16x8 a[NPL][2], b[NPL][2];
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        a[p][m] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(lds + buf + p * IMG_B + addr_a[m]);
        b[p][m] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(lds + buf + p * IMG_B + addr_b[m]);
      }
    // plane pairs, smallest terms first
    constexpr int NPAIR = NPL == 3 ? 6 : 3;
    constexpr int pa[6] = {0, 2, 1, 0, 1, 0}, pb[6] = {2, 0, 1, 1, 0, 0};       // 6 products
    constexpr int qa[3] = {0, 1, 0}, qb[3] = {1, 0, 0};                         // 3 products
#pragma unroll
    for (int k = 0; k < NPAIR; ++k) {
      const int x = NPL == 3 ? pa[k] : qa[k], y = NPL == 3 ? pb[k] : qb[k];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) c[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[x][m], b[y][n], c[m][n], 0, 0, 0);
    }
  }
  gfloat* q = (gfloat*)d.slabs + (long long)item * (TM * TM) + (64 * wm) * 128 + 64 * wn;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        q[(32 * m + row) * 128 + 32 * n + r32] = c[m][n][reg];
      }
}

__global__ void __launch_bounds__(THREADS, 3) probe_kernel2(Flat d) { probe_body<2>(d); }
__global__ void __launch_bounds__(THREADS, 3) probe_kernel3(Flat d) { probe_body<3>(d); }

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

template <int NPL>
static void run(int N, int C, int HW, int target_items, float* src, __bf16* planes) {
  Flat d;
  memset(&d, 0, sizeof(d));
  d.N = N; d.C = C; d.HW = HW;
  d.planes = planes; d.plane_elems = (long long)N * C * HW;
  d.P = C / TM; d.n_tiles = d.P * d.P;
  d.SPS = HW / KC;
  d.total_stages = N * d.SPS;
  int slices = std::max(1, std::min(d.total_stages, (target_items + d.n_tiles - 1) / d.n_tiles));
  d.spi = (d.total_stages + slices - 1) / slices;
  d.n_slices = (d.total_stages + d.spi - 1) / d.spi;
  d.n_items = d.n_slices * d.n_tiles;
  float* slabs;
  hipMalloc(&slabs, (size_t)d.n_items * TM * TM * 4);
  d.slabs = slabs;
  const int grid = (d.n_items + 255) / 256 * 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  (NPL == 3 ? probe_kernel3<<<grid, THREADS>>>(d) : probe_kernel2<<<grid, THREADS>>>(d));
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); exit(1); }
  auto w0 = std::chrono::steady_clock::now();
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() < 1.5) {
    for (int k = 0; k < 50; ++k) (NPL == 3 ? probe_kernel3<<<grid, THREADS>>>(d) : probe_kernel2<<<grid, THREADS>>>(d));
    hipDeviceSynchronize();
  }
  const int reps = 40;
  hipEventRecord(e0);
  for (int k = 0; k < reps; ++k) (NPL == 3 ? probe_kernel3<<<grid, THREADS>>>(d) : probe_kernel2<<<grid, THREADS>>>(d));
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double flops = 2.0 * C * C * (double)N * HW;
  const int ti = d.P - 1, tj = 1 % d.P, tile = ti * d.P + tj;
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
  for (int e = 0; e < TM * TM; ++e) { num += (acc[e] - r[e]) * (acc[e] - r[e]); den += r[e] * r[e]; }
  const int passes = NPL == 3 ? 6 : 3;
  printf("%d bf16 products C=%4d HW=%4d N=%d: items %d (spi %d)  %.3f ms  %.1f TFLOP/s of the fp32 product (%.3f of 157.3; %.3f of its own "
         "roof 2500 / %d = %.0f)  rel err vs fp64 %.2e\n", passes, C, HW, N, d.n_items, d.spi, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3,
         flops / ms / 1e9 / (2500.0 / passes), passes, 2500.0 / passes, std::sqrt(num / den));
  hipFree(slabs);
}

int main(int argc, char** argv) {
  const int target = argc > 1 ? atoi(argv[1]) : 4096;
  const int cases[][3] = {{1024, 784, 128}, {2048, 192, 512}, {512, 3136, 128}};
  for (auto& cs : cases) {
    const int C = cs[0], HW = cs[1], N = cs[2];
    const size_t elems = (size_t)N * C * HW;
    std::vector<float> h(elems);
    unsigned st = 12345u + C * 7 + HW;
    for (size_t i = 0; i < elems; ++i) { st = st * 1664525u + 1013904223u; h[i] = ((st >> 8) & 0xffff) / 65536.0f - 0.3f; }
    float* src;
    __bf16* planes;
    hipMalloc(&src, elems * 4);
    hipMalloc(&planes, elems * 6);
    hipMemcpy(src, h.data(), elems * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    split_kernel<<<(unsigned)((elems + 255) / 256), 256>>>(src, planes, (long long)elems);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int k = 0; k < 10; ++k) split_kernel<<<(unsigned)((elems + 255) / 256), 256>>>(src, planes, (long long)elems);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("split pass: %.0f MB of fp32 -> three bf16 planes: %.3f ms (%.2f TB/s of 10 B per element)\n", elems * 4 / 1e6, ms / 10, elems * 10.0 / (ms / 10) / 1e9);
    run<2>(N, C, HW, target * (N / 32), src, planes);
    run<3>(N, C, HW, target * (N / 32), src, planes);
    hipFree(src); hipFree(planes);
  }
  return 0;
}
