// Prototype of the LDS-DMA factor-build kernel for "flat" factors (1x1 stride-1 convs, grad_output factors):
//   X_s = src[s] is a (C x HW) row-major matrix per sample, slab(tile, slice) = sum over the slice's (s, pixel) of
//   X[rows_i][k] X[rows_j][k]  for every upper-triangular 128x128 tile.
// Staging: buffer_load_dwordx4 ... lds (1 KiB per wave-instruction) into an XOR-swizzled [128 rows][16 x 16 B] image,
// double buffered; operands by ds_read_b128 through 32 precomputed per-lane addresses; no vector ALU work in steady state.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/flat_syrk_proto.hip -o tools/micro/flat_syrk_proto
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) float gfloat;

constexpr int THREADS = 256;
constexpr int TM = 128;
// KC pixels per stage row: 64 (256-byte rows, 128 KiB LDS, one workgroup per CU) or 32 (128-byte rows, 64 KiB, two per CU)
template <int KC> struct Geo {
  static constexpr int ROW_B = KC * 4;                  // LDS bytes per row per stage
  static constexpr int SLOTS = KC / 4;                  // 16-byte slots per row
  static constexpr int STEPS = KC / 8;                  // MFMA steps (8 pixels) per full stage
  static constexpr int ROWS_PER_PIECE = 1024 / ROW_B;   // rows covered by one 1 KiB DMA wave-instruction
  static constexpr int PIECES = TM / ROWS_PER_PIECE / 4;   // pieces per panel per wave
  static constexpr int PANEL_B = TM * ROW_B;
  static constexpr int LDS_B = 4 * PANEL_B;             // [Pi buf0][Pi buf1][Pj buf0][Pj buf1]
};

struct Flat {
  const float* src;
  float* slabs;
  int N, C, HW;
  int P, n_tiles;
  int TS, SPS, base_steps, rem_steps, nv_last;   // steps (8 px) per sample, stages per sample, balanced split, valid px in the last step
  int total_stages, spi, n_slices, n_items;       // stages per item
};

__device__ __forceinline__ void decode_tile(int t, int P, int& ti, int& tj) {
  ti = 0;
  while (t >= P - ti) { t -= P - ti; ++ti; }
  tj = ti + t;
}

template <int PART, typename Hook>
__device__ __forceinline__ void mfma_step(const f32x4& a0, const f32x4& a1, const f32x4& b0, const f32x4& b1,
                                          f32x16& c00, f32x16& c01, f32x16& c10, f32x16& c11, int ne, Hook hook) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (e < ne) {
      if (PART != 3) c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], c00, 0, 0, 0);
      if (PART != 2) c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b1[e], c01, 0, 0, 0);
      if (PART == 0 || PART == 2) c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b0[e], c10, 0, 0, 0);
      if (PART != 2) c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], c11, 0, 0, 0);
    }
    hook(e);                      // one LDS-DMA piece behind a group of MFMAs: its issue cost hides under them
  }
}

template <int KC, int PART>
__device__ __forceinline__ void flat_body(const Flat& d, int local, __attribute__((address_space(3))) char* lds) {
  using G = Geo<KC>;
  constexpr int ROW_B = G::ROW_B, PANEL_B = G::PANEL_B, STEPS = G::STEPS, PIECES = G::PIECES, RPP = G::ROWS_PER_PIECE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31, h = lane >> 5;
  const int slice = local / d.n_tiles, tile = local - slice * d.n_tiles;
  int ti, tj;
  decode_tile(tile, d.P, ti, tj);
  const bool diag = (ti == tj);
  int wm = wave >> 1, wn = wave & 1;
  if (PART >= 2) { wm = 0; wn = 1; }
  const int i0 = ti * TM, j0 = tj * TM;
  const int HW = d.HW, C = d.C;

  // ---- DMA lane geometry: load slot i of this wave covers panel rows 16 i + 4 wave + (lane >> 4); the lane's
  // physical 16-byte slot (lane & 15) holds logical pixel group g = slot ^ (row & 15)
  // KC = 64: a piece is 4 rows x 16 slots, slot ^ (row & 15); KC = 32: 8 rows x 8 slots, slot ^ ((row >> 1) & 7).
  // Either way the XOR key of a lane does not depend on the piece index (pieces of a wave are 16 / 32 rows apart).
  const int rloc = (KC == 64) ? (lane >> 4) : (lane >> 3);           // row inside the piece
  const int rsub = RPP * wave + rloc;                                 // row inside the group of 4 pieces (one per wave)
  const int key = (KC == 64) ? (rsub & 15) : ((rsub >> 1) & 7);
  const int g_lane = (lane & (G::SLOTS - 1)) ^ key;
  const int voff = (rsub * HW + 4 * g_lane) * 4;
  const long long total_b = (long long)d.N * C * HW * 4;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)d.src, 0, (unsigned)total_b, 0x00020000);

  // ---- operand addresses: row R of a panel, step j: R * 256 + ((2 j + h) ^ (R & 15)) * 16
  unsigned addr[4][STEPS];
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    const int R = ((o < 2) ? 64 * wm : 64 * wn) + (o & 1) * 32 + r32;
    const unsigned pbase = (o < 2 || diag) ? 0u : 2u * PANEL_B;
    const int rkey = (KC == 64) ? (R & 15) : ((R >> 1) & 7);
#pragma unroll
    for (int j = 0; j < STEPS; ++j) addr[o][j] = pbase + R * ROW_B + (((2 * j + h) ^ rkey) << 4);
  }

  f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};
  const int t0 = slice * d.spi, t1 = min(t0 + d.spi, d.total_stages);
  const int n_panels = diag ? 1 : 2;

  auto stage_geo = [&](int t, int& s, int& px0, int& nsteps, bool& last) {
    s = t / d.SPS;
    const int q = t - s * d.SPS;
    nsteps = d.base_steps + (q < d.rem_steps ? 1 : 0);
    px0 = 8 * (q * d.base_steps + min(q, d.rem_steps));
    last = (q == d.SPS - 1);
  };
  // next stage's DMA geometry (scalars) and the issue of one piece: piece i = (panel i >> 3, row group i & 7)
  int n_soff[2] = {0, 0};
  int n_gmax = 0;
  unsigned n_buf = 0;
  auto plan_next = [&](int t) {
    int s, px0, nsteps; bool last;
    stage_geo(t, s, px0, nsteps, last);
    n_gmax = last ? (HW - px0 + 3) / 4 : 2 * nsteps;               // pixel groups this stage needs
    n_buf = (unsigned)(t & 1) * PANEL_B;
    n_soff[0] = ((s * C + i0) * HW + px0) * 4;
    n_soff[1] = ((s * C + j0) * HW + px0) * 4;
  };
  auto piece = [&](int i) {
    const int p = i / PIECES, slot = i % PIECES;
    if (p < n_panels && g_lane < n_gmax) {
      const unsigned lbase = (p ? 2u * PANEL_B : 0u) + n_buf + (unsigned)(RPP * wave + 4 * RPP * slot) * ROW_B;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds + lbase), 16, voff, n_soff[p] + slot * 4 * RPP * HW * 4, 0, 0);
    }
  };
  constexpr int NP = 2 * PIECES;             // pieces per stage per wave
  constexpr int PPS = (NP + STEPS / 2 - 1) / (STEPS / 2) > 4 ? 4 : (NP + STEPS / 2 - 1) / (STEPS / 2);   // per step, in the first half

  plan_next(t0);
#pragma unroll
  for (int i = 0; i < NP; ++i) piece(i);
  for (int t = t0; t < t1; ++t) {
    __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0): this wave's DMA of stage t has landed
    __syncthreads();                           // everyone's has; everyone is done reading the other buffer
    const bool more = t + 1 < t1;
    if (more) plan_next(t + 1);
    int s, px0, nsteps; bool last;
    stage_geo(t, s, px0, nsteps, last);
    const unsigned buf = (unsigned)(t & 1) * PANEL_B;
    auto rd = [&](int o, int j) { return *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(lds + addr[o][j] + buf); };
    // software pipeline over the stage's steps: operands of step j + 1 are read while the MFMAs of step j issue;
    // the 16 DMA pieces of stage t + 1 are issued two per step, each behind a group of MFMAs
    f32x4 a0 = rd(0, 0), a1 = rd(1, 0), b0 = rd(2, 0), b1 = rd(3, 0);
    int next_piece = 0;
#pragma unroll
    for (int j = 0; j < STEPS; ++j) {
      if (j < nsteps) {
        f32x4 na0, na1, nb0, nb1;
        if (j + 1 < STEPS && j + 1 < nsteps) { na0 = rd(0, j + 1); na1 = rd(1, j + 1); nb0 = rd(2, j + 1); nb1 = rd(3, j + 1); }
        int ne = 4;
        if (last && j == nsteps - 1 && d.nv_last < 8) {
          // the sample's final step: only nv_last of its 8 pixels exist
          ne = min(4, d.nv_last);
          asm volatile("; sample tail" ::: "memory");      // keeps this a branch: if-converted, its 16 selects ran at every step
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool ok = (4 * h + e) < d.nv_last;
            a0[e] = ok ? a0[e] : 0.0f; a1[e] = ok ? a1[e] : 0.0f;
            b0[e] = ok ? b0[e] : 0.0f; b1[e] = ok ? b1[e] : 0.0f;
          }
        }
        // all 16 pieces go out during the first four steps: the last one then still has half of the stage's
        // MFMA time to land before the wait at the top of the next stage
        mfma_step<PART>(a0, a1, b0, b1, c00, c01, c10, c11, ne,
                        [&](int k) { if (more && k < PPS && PPS * j + k < NP) piece(PPS * j + k); });
        next_piece = min(NP, PPS * j + PPS);
        if (j + 1 < STEPS && j + 1 < nsteps) { a0 = na0; a1 = na1; b0 = nb0; b1 = nb1; }
      }
    }
    if (more) {
#pragma unroll
      for (int i = 0; i < NP; ++i) if (i >= next_piece) piece(i);      // short stages: the rest
    }
  }

  gfloat* slab = (gfloat*)d.slabs + (long long)local * (TM * TM);
  gfloat* q = slab + (64 * wm) * 128 + 64 * wn;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
    if (PART != 3) q[row * 128 + r32] = c00[reg];
    if (PART != 2) q[row * 128 + 32 + r32] = c01[reg];
    if (PART != 3) q[(32 + row) * 128 + r32] = c10[reg];
    if (PART != 2) q[(32 + row) * 128 + 32 + r32] = c11[reg];
  }
}

template <int KC>
__global__ void __launch_bounds__(THREADS, KC == 64 ? 1 : 2) flat_syrk_kernel(Flat d) {
  __shared__ __attribute__((aligned(1024))) char lds[Geo<KC>::LDS_B];
  int item;
  {
    const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
    item = ((j / 32) * 8 + xcd) * 32 + (j % 32);
  }
  if (item >= d.n_items) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = item % d.n_tiles;
  int ti, tj;
  decode_tile(tile, d.P, ti, tj);
  int part = 0;
  if (ti == tj) {
    const int wm = wave >> 1, wn = wave & 1;
    part = (wm == wn) ? 1 : (wm == 0 ? 2 : 3);
  }
  part = __builtin_amdgcn_readfirstlane(part);
  auto* l3 = (__attribute__((address_space(3))) char*)lds;
  if (part == 0) flat_body<KC, 0>(d, item, l3);
  else if (part == 1) flat_body<KC, 1>(d, item, l3);
  else if (part == 2) flat_body<KC, 2>(d, item, l3);
  else flat_body<KC, 3>(d, item, l3);
}

// reference: one thread per (i, j) of the upper triangle, fp64 accumulation
__global__ void ref_kernel(const float* src, double* out, int N, int C, int HW, int i0, int j0, int n) {
  const int i = i0 + blockIdx.x * 16 + threadIdx.x / 16, j = j0 + blockIdx.y * 16 + threadIdx.x % 16;
  if (i >= i0 + n || j >= j0 + n) return;
  double acc = 0;
  for (int s = 0; s < N; ++s) {
    const float* a = src + ((long long)s * C + i) * HW;
    const float* b = src + ((long long)s * C + j) * HW;
    for (int p = 0; p < HW; ++p) acc += (double)a[p] * b[p];
  }
  out[(i - i0) * n + (j - j0)] = acc;
}

template <int KC>
static void run(int N, int C, int HW, int target_items) {
  Flat d;
  memset(&d, 0, sizeof(d));
  d.N = N; d.C = C; d.HW = HW;
  d.P = C / TM; d.n_tiles = d.P * (d.P + 1) / 2;
  d.TS = (HW + 7) / 8;
  d.SPS = (d.TS + KC / 8 - 1) / (KC / 8);
  d.base_steps = d.TS / d.SPS; d.rem_steps = d.TS % d.SPS;
  d.nv_last = HW - 8 * (d.TS - 1);
  d.total_stages = N * d.SPS;
  int slices = std::max(1, std::min(d.total_stages, (target_items + d.n_tiles - 1) / d.n_tiles));
  d.spi = (d.total_stages + slices - 1) / slices;
  d.n_slices = (d.total_stages + d.spi - 1) / d.spi;
  d.n_items = d.n_slices * d.n_tiles;
  const size_t elems = (size_t)N * C * HW;
  std::vector<float> h(elems);
  unsigned st = 12345u + C * 7 + HW;
  for (size_t i = 0; i < elems; ++i) { st = st * 1664525u + 1013904223u; h[i] = ((st >> 8) & 0xffff) / 65536.0f - 0.3f; }
  float* src; float* slabs;
  hipMalloc(&src, elems * 4);
  hipMemcpy(src, h.data(), elems * 4, hipMemcpyHostToDevice);
  hipMalloc(&slabs, (size_t)d.n_items * TM * TM * 4);
  hipMemset(slabs, 0, (size_t)d.n_items * TM * TM * 4);
  d.src = src; d.slabs = slabs;
  const int grid = (d.n_items + 255) / 256 * 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  flat_syrk_kernel<KC><<<grid, THREADS>>>(d);
  hipError_t err = hipDeviceSynchronize();
  if (err != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(err)); exit(1); }
  float best = 1e30f, sum = 0;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0);
    flat_syrk_kernel<KC><<<grid, THREADS>>>(d);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = std::min(best, ms);
    if (rep) sum += ms;
  }
  const double exec_flops = (double)C * (C + 1.0) * N * HW;
  // correctness: last diagonal tile and the (0, P-1) tile against the fp64 reference
  double worst = 0;
  for (int which = 0; which < 2; ++which) {
    const int ti = which ? 0 : d.P - 1, tj = d.P - 1;
    int tile = 0;
    for (int a = 0; a < ti; ++a) tile += d.P - a;
    tile += tj - ti;
    std::vector<double> acc(TM * TM, 0.0);
    std::vector<float> part(TM * TM);
    for (int s = 0; s < d.n_slices; ++s) {
      hipMemcpy(part.data(), slabs + ((size_t)s * d.n_tiles + tile) * TM * TM, TM * TM * 4, hipMemcpyDeviceToHost);
      for (int e = 0; e < TM * TM; ++e) acc[e] += part[e];
    }
    double* ref;
    hipMalloc(&ref, TM * TM * 8);
    ref_kernel<<<dim3(8, 8), 256>>>(src, ref, N, C, HW, ti * TM, tj * TM, TM);
    std::vector<double> r(TM * TM);
    hipMemcpy(r.data(), ref, TM * TM * 8, hipMemcpyDeviceToHost);
    hipFree(ref);
    double num = 0, den = 0;
    for (int i = 0; i < TM; ++i)
      for (int j = 0; j < TM; ++j) {
        if (ti == tj && j < i) continue;
        if (ti == tj && (i / 32) > (j / 32)) continue;
        const double dlt = acc[i * TM + j] - r[i * TM + j];
        num += dlt * dlt; den += r[i * TM + j] * r[i * TM + j];
      }
    worst = std::max(worst, std::sqrt(num / den));
  }
  printf("KC=%d C=%4d HW=%4d N=%d: tiles %d slices %d items %d (spi %d, SPS %d, steps %d+%d)  best %.3f ms avg %.3f ms  %.1f TF executed (%.3f of 157.3)  rel err %.2e\n",
         KC, C, HW, N, d.n_tiles, d.n_slices, d.n_items, d.spi, d.SPS, d.base_steps, d.rem_steps, best, sum / 5, exec_flops / best / 1e9,
         exec_flops / best / 1e9 / 157.3, worst);
  hipFree(src); hipFree(slabs);
}

template <int KC> static void all(int target) {
  run<KC>(32, 1024, 196, target);
  run<KC>(32, 256, 3136, target);
  run<KC>(32, 512, 784, target);
  run<KC>(32, 2048, 49, target);
  run<KC>(32, 2048, 196, target);
  run<KC>(32, 512, 3136, target);
}

int main(int argc, char** argv) {
  const int target = argc > 1 ? atoi(argv[1]) : 1024;
  all<64>(target);
  all<32>(target);
  return 0;
}
