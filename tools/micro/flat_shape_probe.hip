// Does the fp32 MFMA SHAPE move the LDS-DMA factor-build kernel under sustained load?  (MI355X_MICROARCH.md, DVFS give-back
// item 7: where the chip holds its clock down, the clock can depend on the MFMA shape; syrk_flat_kernel runs at 2.07 GHz.)
// The product kernel's stage geometry (16-pixel stages, [128 rows][4 x 16 B] XOR-swizzled double-buffered images, four
// workgroups per CU, buffer_load ... lds staging, ds_read_b128 operands) on full 128 x 128 tiles X_i X_j^T, with
//   SHAPE 0: v_mfma_f32_32x32x2_f32  (wave quadrant 64 x 64 = 2 x 2 blocks, a read = 32 rows x 4 pixels of a lane half)
//   SHAPE 1: v_mfma_f32_16x16x4_f32  (wave quadrant 64 x 64 = 4 x 4 blocks, a read = 16 rows x 4 pixels of a lane quarter)
// Same LDS bytes per flop, same DMA, same accumulator registers; the 16x16x4 form reads and writes half as many
// accumulator values per multiply-add.  Each case runs back to back for ~1.5 s before it is timed (regulated clock).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/flat_shape_probe.hip -o tools/micro/flat_shape_probe
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

constexpr int THREADS = 256, TM = 128, KC = 16, ROW_B = 64, RPP = 16, PIECES = 2, PANEL_B = TM * ROW_B, LDS_B = 4 * PANEL_B, NP = 4;

struct Flat {
  const float* src;
  float* slabs;
  int N, C, HW, P, n_tiles, SPS, total_stages, spi, n_slices, n_items;
};

template <int SHAPE>
__global__ void __launch_bounds__(THREADS, 4) probe_kernel(Flat d) {
  __shared__ __attribute__((aligned(1024))) char smem[LDS_B];
  lds_char* lds = (lds_char*)smem;
  int item;
  {
    const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
    item = ((j / 32) * 8 + xcd) * 32 + (j % 32);
  }
  if (item >= d.n_items) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slice = item / d.n_tiles, tile = item - slice * d.n_tiles;
  const int ti = tile / d.P, tj = tile - ti * d.P;
  const int wm = wave >> 1, wn = wave & 1;
  const int i0 = ti * TM, j0 = tj * TM, HW = d.HW, C = d.C;
  // DMA lane geometry: piece `slot` of this wave covers panel rows 64 slot + 16 wave + (lane >> 2); physical 16-byte
  // slot (lane & 3) holds logical pixel group (lane & 3) ^ ((row >> 2) & 3)
  const int rsub = RPP * wave + (lane >> 2);
  const int g_lane = (lane & 3) ^ ((rsub >> 2) & 3);
  const int voff = (rsub * HW + 4 * g_lane) * 4;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)d.src, 0, (unsigned)((long long)d.N * C * HW * 4), 0x00020000);

  constexpr int NO = SHAPE == 0 ? 2 : 4;            // operand blocks per side
  constexpr int NR = SHAPE == 0 ? 2 : 1;            // reads per block per stage
  unsigned addr_a[NO][NR], addr_b[NO][NR];
#pragma unroll
  for (int o = 0; o < NO; ++o)
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      int Ra, Rb, g;
      if (SHAPE == 0) { Ra = 64 * wm + 32 * o + (lane & 31); Rb = 64 * wn + 32 * o + (lane & 31); g = 2 * j + (lane >> 5); }
      else { Ra = 64 * wm + 16 * o + (lane & 15); Rb = 64 * wn + 16 * o + (lane & 15); g = lane >> 4; }
      addr_a[o][j] = Ra * ROW_B + ((g ^ ((Ra >> 2) & 3)) << 4);
      addr_b[o][j] = 2u * PANEL_B + Rb * ROW_B + ((g ^ ((Rb >> 2) & 3)) << 4);
    }
  f32x16 c32[SHAPE == 0 ? 4 : 1];
  f32x4 c16[SHAPE == 0 ? 1 : 16];
#pragma unroll
  for (auto& c : c32) c = 0.0f;
#pragma unroll
  for (auto& c : c16) c = 0.0f;

  const int t0 = slice * d.spi, t1 = min(t0 + d.spi, d.total_stages);
  int n_soff[2] = {0, 0};
  unsigned n_buf = 0;
  auto plan_next = [&](int t) {
    const int s = t / d.SPS, q = t - s * d.SPS;
    n_buf = (unsigned)(t & 1) * PANEL_B;
    n_soff[0] = ((s * C + i0) * HW + 16 * q) * 4;
    n_soff[1] = ((s * C + j0) * HW + 16 * q) * 4;
  };
  auto piece = [&](int i) {
    const int p = i / PIECES, slot = i % PIECES;
    const unsigned lbase = (p ? 2u * PANEL_B : 0u) + n_buf + (unsigned)(RPP * wave + 4 * RPP * slot) * ROW_B;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds + lbase), 16, voff, n_soff[p] + slot * 4 * RPP * HW * 4, 0, 0);
  };
  plan_next(t0);
#pragma unroll
  for (int i = 0; i < NP; ++i) piece(i);
  for (int t = t0; t < t1; ++t) {
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    const bool more = t + 1 < t1;
    if (more) plan_next(t + 1);
    const unsigned buf = (unsigned)(t & 1) * PANEL_B;
    auto rd = [&](unsigned a) { return *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(lds + a + buf); };
    if (SHAPE == 0) {
      f32x4 a0 = rd(addr_a[0][0]), a1 = rd(addr_a[1][0]), b0 = rd(addr_b[0][0]), b1 = rd(addr_b[1][0]);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x4 na0, na1, nb0, nb1;
        if (j == 0) { na0 = rd(addr_a[0][NR - 1]); na1 = rd(addr_a[1][NR - 1]); nb0 = rd(addr_b[0][NR - 1]); nb1 = rd(addr_b[1][NR - 1]); }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          c32[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], c32[0], 0, 0, 0);
          c32[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b1[e], c32[1], 0, 0, 0);
          c32[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b0[e], c32[2], 0, 0, 0);
          c32[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], c32[3], 0, 0, 0);
          if (more && j == 0) piece(e);
        }
        if (j == 0) { a0 = na0; a1 = na1; b0 = nb0; b1 = nb1; }
      }
    } else {
      f32x4 a[4], b[4];
#pragma unroll
      for (int o = 0; o < 4; ++o) { a[o] = rd(addr_a[o][0]); b[o] = rd(addr_b[o][0]); }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n)
            c16[(SHAPE == 0 ? 0 : 4 * m + n)] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][e], b[n][e], c16[(SHAPE == 0 ? 0 : 4 * m + n)], 0, 0, 0);
        if (more) piece(e);
      }
    }
  }
  gfloat* q = (gfloat*)d.slabs + (long long)item * (TM * TM) + (64 * wm) * 128 + 64 * wn;
  if (SHAPE == 0) {
    const int r32 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
      q[row * 128 + r32] = c32[0][reg];
      q[row * 128 + 32 + r32] = c32[SHAPE == 0 ? 1 : 0][reg];
      q[(32 + row) * 128 + r32] = c32[SHAPE == 0 ? 2 : 0][reg];
      q[(32 + row) * 128 + 32 + r32] = c32[SHAPE == 0 ? 3 : 0][reg];
    }
  } else {
    const int r16 = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) q[(16 * m + 4 * kq + reg) * 128 + 16 * n + r16] = c16[SHAPE == 0 ? 0 : 4 * m + n][reg];
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

template <int SHAPE>
static double run(int N, int C, int HW, int target_items, float* src, const std::vector<float>& h) {
  Flat d;
  memset(&d, 0, sizeof(d));
  d.N = N; d.C = C; d.HW = HW;
  d.P = C / TM; d.n_tiles = d.P * d.P;
  d.SPS = HW / 16;
  d.total_stages = N * d.SPS;
  int slices = std::max(1, std::min(d.total_stages, (target_items + d.n_tiles - 1) / d.n_tiles));
  d.spi = (d.total_stages + slices - 1) / slices;
  d.n_slices = (d.total_stages + d.spi - 1) / d.spi;
  d.n_items = d.n_slices * d.n_tiles;
  float* slabs;
  hipMalloc(&slabs, (size_t)d.n_items * TM * TM * 4);
  d.src = src; d.slabs = slabs;
  const int grid = (d.n_items + 255) / 256 * 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe_kernel<SHAPE><<<grid, THREADS>>>(d);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); exit(1); }
  // sustained load: ~1.5 s back to back, then 20 timed launches in one event pair
  auto w0 = std::chrono::steady_clock::now();
  int warm = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() < 1.5) {
    for (int k = 0; k < 50; ++k) probe_kernel<SHAPE><<<grid, THREADS>>>(d);
    hipDeviceSynchronize();
    warm += 50;
  }
  const int reps = 40;
  hipEventRecord(e0);
  for (int k = 0; k < reps; ++k) probe_kernel<SHAPE><<<grid, THREADS>>>(d);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double flops = 2.0 * C * C * (double)N * HW;
  // correctness: tile (P - 1, 1 % P)
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
  printf("%s C=%4d HW=%4d N=%d: items %d (spi %d) warm %d launches  %.3f ms  %.1f TFLOP/s (%.3f of 157.3)  rel err %.2e\n",
         SHAPE == 0 ? "32x32x2" : "16x16x4", C, HW, N, d.n_items, d.spi, warm, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3, std::sqrt(num / den));
  hipFree(slabs);
  return ms;
}

int main(int argc, char** argv) {
  const int target = argc > 1 ? atoi(argv[1]) : 4096;
  // {C, HW, N}: N = 32 keeps the inputs (50 - 205 MB) inside the 256 MB MALL across launches; larger N streams them from HBM
  const int cases[][3] = {{1024, 784, 32}, {2048, 192, 32}, {512, 3136, 32}, {1024, 784, 128}, {1024, 784, 512}, {2048, 192, 512}, {512, 3136, 128}};
  const int first = argc > 2 ? atoi(argv[2]) : 0, last = argc > 3 ? atoi(argv[3]) : 7;
  for (int ci = first; ci < last && ci < 7; ++ci) {
    const int C = cases[ci][0], HW = cases[ci][1], N = cases[ci][2];
    const size_t elems = (size_t)N * C * HW;
    std::vector<float> h(elems);
    unsigned st = 12345u + C * 7 + HW;
    for (size_t i = 0; i < elems; ++i) { st = st * 1664525u + 1013904223u; h[i] = ((st >> 8) & 0xffff) / 65536.0f - 0.3f; }
    float* src;
    hipMalloc(&src, elems * 4);
    hipMemcpy(src, h.data(), elems * 4, hipMemcpyHostToDevice);
    printf("inputs %.0f MB\n", elems * 4 / 1e6);
    for (int rep = 0; rep < 2; ++rep) {
      run<0>(N, C, HW, target * (N / 32), src, h);
      run<1>(N, C, HW, target * (N / 32), src, h);
    }
    hipFree(src);
  }
  return 0;
}
