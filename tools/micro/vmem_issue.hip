// How many cycles does a wave spend ISSUING a burst of loads (not waiting for the data)?
// 512 workgroups x 256 threads (2 per CU like syrk_patch_kernel); each wave issues `nload` loads per round
// into distinct registers, measures clock64() around the issue burst, then waits and consumes them.
// mode 0: buffer_load_dword, 4 rows x 16 lanes per wave (the 14x14 3x3 pattern)
// mode 1: buffer_load_dwordx4, 64 consecutive float4 (the flattened 1x1 pattern)
// mode 2: like 0 while the odd workgroups spin in an MFMA loop (same SIMDs: 2 workgroups per CU)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ unsigned long long g_cyc[4];

template <int MODE>
__global__ void __launch_bounds__(256, 2) k(const float* src, float* out, int rounds, long long nbytes, int foot) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)nbytes, 0x00020000);
  float acc = 0;
  if (MODE == 2 && (blockIdx.x & 1)) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = tid * 0.001f, y = 1.0f;
    for (int r = 0; r < rounds * 40; ++r) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + tid] = a0[0] + a1[1] + a2[2] + a3[3];
    return;
  }
  unsigned long long issue = 0, total = 0;
  const int plane = 196 * 4;
  for (int r = 0; r < rounds; ++r) {
    const int base = ((blockIdx.x * 37 + r * 11 + wave * 5) % 4096) * 64 * plane % foot;
    float st[64];
    const long long t0 = clock64();
    if (MODE == 1) {
      const int voff = lane * 16;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, base + j * 1024 * 4, 0));
        st[4 * j] = v.x; st[4 * j + 1] = v.y; st[4 * j + 2] = v.z; st[4 * j + 3] = v.w;
      }
    } else {
      const int voff = ((lane >> 4) * 14 + (lane & 15)) * 4 + wave * 4 * 14 * 4;
#pragma unroll
      for (int j = 0; j < 64; ++j)
        st[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, base + j * plane, 0));
    }
    const long long t1 = clock64();
#pragma unroll
    for (int j = 0; j < 64; ++j) acc += st[j];
    const long long t2 = clock64();
    issue += t1 - t0; total += t2 - t0;
  }
  out[blockIdx.x * 256 + tid] = acc;
  if (lane == 0) { atomicAdd(&g_cyc[0], issue); atomicAdd(&g_cyc[1], total); atomicAdd(&g_cyc[2], 1ull); }
}

template <int MODE>
void run(const char* name, const float* src, float* out, long long nbytes, int foot) {
  unsigned long long z[4] = {0, 0, 0, 0}, r[4];
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_cyc), z, sizeof(z));
  const int rounds = 50;
  k<MODE><<<512, 256>>>(src, out, rounds, nbytes, foot);
  (void)hipDeviceSynchronize();
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_cyc), z, sizeof(z));
  k<MODE><<<512, 256>>>(src, out, rounds, nbytes, foot);
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(r, HIP_SYMBOL(g_cyc), sizeof(r));
  const double n = (double)r[2] * rounds;
  const int nl = MODE == 1 ? 16 : 64;
  printf("%-52s issue %.0f cycles per burst (%.1f per load), burst+wait %.0f\n", name, r[0] / n, r[0] / n / nl, r[1] / n);
}

int main() {
  const long long nbytes = 1ll << 30;
  float *src, *out;
  (void)hipMalloc(&src, nbytes); (void)hipMemset(src, 0, nbytes);
  (void)hipMalloc(&out, 512 * 256 * 4);
  for (int foot : {1 << 20, 16 << 20, 512 << 20}) {
    printf("footprint %d MiB\n", foot >> 20);
    run<0>("64 x buffer_load_dword, 4 rows x 16 lanes", src, out, nbytes, foot);
    run<1>("16 x buffer_load_dwordx4, contiguous", src, out, nbytes, foot);
    run<2>("64 x buffer_load_dword next to an MFMA-bound wave", src, out, nbytes, foot);
  }
  return 0;
}
