// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access pattern of the factor-build kernels (MI355X_MICROARCH.md,
// "HBM": "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
// Every byte of a 1 GiB buffer (four times the Infinity Cache) is read exactly once by `buffer_load ... lds`, 16 bytes per lane:
//   mode 0: a wave instruction reads 1 KiB contiguous (the pattern the guide's factor of 2 was measured on)
//   mode 1: a wave instruction reads 16 rows x 64 bytes, rows `pitch` bytes apart, and walks along the rows in steps of
//           64 bytes - syrk_flat_kernel's stage (128 channel rows x 16 pixels): each 128-byte line is touched by two
//           consecutive stages of the same workgroup, half a line each
//   mode 2: as mode 1, but only the FIRST 64 bytes of every 128-byte line are ever read (half the bytes)
// Run once plainly (prints GB/s of the bytes read) and once under  rocprofv3 --kernel-trace --pmc FETCH_SIZE.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/fetch_calib.hip -o tools/micro/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) char lds_char;

constexpr int THREADS = 256, STAGE_B = 8192, NBUF = 4;
constexpr long long TOTAL = 1ll << 30;
constexpr int ROWS = 8192, ROW_B = (int)(TOTAL / ROWS), PITCH = ROW_B + 256;   // 128 KiB of every row read, rows 128.25 KiB apart
constexpr long long ALLOC = (long long)ROWS * PITCH;
constexpr int WG_ROWS = 128, KSLICES = 8, SLICE_B = ROW_B / KSLICES;   // 16 KiB of every row per workgroup

template <int MODE>
__global__ void __launch_bounds__(THREADS, 4) read_kernel(const char* __restrict__ src, int* __restrict__ sink) {
  __shared__ __attribute__((aligned(1024))) char smem[NBUF * STAGE_B];
  lds_char* lds = (lds_char*)smem;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (unsigned)ALLOC, 0x00020000);
  const int wg = blockIdx.x;
  int n_stages;
  unsigned voff0, voff1, step;
  if (MODE == 0) {
    // 2 MiB contiguous per workgroup, a stage = 8 KiB contiguous, a wave instruction 1 KiB
    const unsigned base = (unsigned)wg * (2u << 20);
    voff0 = base + wave * 1024 + lane * 16;
    voff1 = voff0 + 4096;
    step = STAGE_B;
    n_stages = (2 << 20) / STAGE_B;
  } else {
    const int rg = wg / KSLICES, ks = wg - rg * KSLICES;
    const int row = rg * WG_ROWS + 16 * wave + (lane >> 2);
    voff0 = (unsigned)row * PITCH + ks * SLICE_B + (lane & 3) * 16;
    voff1 = voff0 + 64u * PITCH;
    step = MODE == 1 ? 64 : 128;
    n_stages = SLICE_B / step;
  }
  for (int t = 0; t < n_stages; ++t) {
    const unsigned buf = (unsigned)(t & (NBUF - 1)) * STAGE_B;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds + buf + wave * 1024), 16, voff0, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds + buf + 4096 + wave * 1024), 16, voff1, 0, 0, 0);
    voff0 += step;
    voff1 += step;
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (sink != nullptr && threadIdx.x == 0 && smem[wg & 1023] == 77) sink[0] = 1;
}

int main(int argc, char** argv) {
  char* buf;
  int* sink;
  if (hipMalloc(&buf, ALLOC) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
  hipMemset(buf, 1, ALLOC);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = (int)(TOTAL / (2 << 20));       // 512 workgroups, 2 MiB each (modes 0, 1); mode 2 reads half of it
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(read_kernel<0>, dim3(grid), dim3(THREADS), 0, 0, buf, sink);
      else if (mode == 1) hipLaunchKernelGGL(read_kernel<1>, dim3(grid), dim3(THREADS), 0, 0, buf, sink);
      else hipLaunchKernelGGL(read_kernel<2>, dim3(grid), dim3(THREADS), 0, 0, buf, sink);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double bytes = mode == 2 ? TOTAL / 2.0 : (double)TOTAL;
      printf("mode %d: %.3f ms, %.0f GB/s of the bytes read (%.3f GB)\n", mode, ms, bytes / ms / 1e6, bytes / 1e9);
    }
  }
  if (hipDeviceSynchronize() != hipSuccess) return 2;
  return 0;
}
