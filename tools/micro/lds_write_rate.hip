// LDS write cost on gfx950 for the flavours the staging store phase could use: cycles for one wave-group
// (4 waves) to store 64 KB (64 floats per lane) with b32, write2_b32, b64 and b128 instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ unsigned long long g_c[4];

template <int MODE>
__global__ void __launch_bounds__(256, 2) k(float* out, int rounds) {
  __shared__ __attribute__((aligned(16))) float lds[18000];
  const int tid = threadIdx.x, lane = tid & 63;
  float v[64];
  for (int j = 0; j < 64; ++j) v[j] = tid * 0.5f + j;
  unsigned long long tot = 0;
  for (int r = 0; r < rounds; ++r) {
    __syncthreads();
    const long long t0 = clock64();
    if (MODE == 0) {          // 64 x ds_write_b32, lanes consecutive
      float* l = lds + tid;
#pragma unroll
      for (int j = 0; j < 64; ++j) l[j * 256] = v[j];
    } else if (MODE == 1) {   // 32 x ds_write2_b32 (two adjacent words)
      float* l = lds + 2 * tid;
      int step = 512 + (r & 0); asm volatile("" : "+s"(step));
#pragma unroll
      for (int j = 0; j < 32; ++j) { l[0] = v[2 * j]; l[1] = v[2 * j + 1]; l += step; }
    } else if (MODE == 2) {   // 32 x ds_write_b64
      f32x2* l = reinterpret_cast<f32x2*>(lds) + tid;
#pragma unroll
      for (int j = 0; j < 32; ++j) { f32x2 x = {v[2 * j], v[2 * j + 1]}; l[j * 256] = x; }
    } else {                  // 16 x ds_write_b128
      f32x4* l = reinterpret_cast<f32x4*>(lds) + tid;
#pragma unroll
      for (int j = 0; j < 16; ++j) { f32x4 x = {v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]}; l[j * 256] = x; }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    tot += clock64() - t0;
#pragma unroll
    for (int j = 0; j < 64; ++j) v[j] += lds[(tid * 7 + j) & 16383];
  }
  out[blockIdx.x * 256 + tid] = v[3];
  if (lane == 0) { atomicAdd(&g_c[0], tot); atomicAdd(&g_c[1], 1ull); }
}

template <int MODE>
void run(const char* name, float* out) {
  unsigned long long z[4] = {0, 0, 0, 0}, r[4];
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_c), z, sizeof(z));
  const int rounds = 100;
  k<MODE><<<256, 256>>>(out, rounds);     // one workgroup per CU
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(r, HIP_SYMBOL(g_c), sizeof(r));
  printf("%-28s %6.0f cycles per 64 KB (4 waves)  = %.1f B/clk/CU\n", name, (double)r[0] / r[1] / rounds,
         65536.0 / ((double)r[0] / r[1] / rounds));
}

int main() {
  float* out;
  (void)hipMalloc(&out, 512 * 256 * 4);
  run<0>("64 x ds_write_b32", out);
  run<1>("32 x ds_write2_b32", out);
  run<2>("32 x ds_write_b64", out);
  run<3>("16 x ds_write_b128", out);
  return 0;
}
