// Does a wave's LDS-store phase (v_add + ds_write2 x 32) slow down when the other wave of its SIMD runs an
// MFMA + ds_read loop?  512 workgroups x 256 threads, 2 per CU.  Even workgroups: store phase in a loop,
// timed with clock64.  Odd workgroups: mode 0 exit at once, mode 1 run the MFMA/LDS-read loop.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ unsigned long long g_c[4];

__global__ void __launch_bounds__(256, 2) k(float* out, int rounds, int mode, int prio) {
  __shared__ float lds[18000];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 18000; i += 256) lds[i] = 0.001f * (i & 15);
  __syncthreads();
  if (blockIdx.x & 1) {
    if (mode == 0) return;
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    const float* p = lds + (lane & 31) * 65 + (lane >> 5);
    for (int r = 0; r < rounds * 24; ++r) {
      const int o = (r & 15) * 4;
      const float x0 = p[o], x1 = p[o + 2080], y0 = p[o + 4160], y1 = p[o + 6240];
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y0, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y1, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y0, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y1, a3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + tid] = a0[0] + a1[1] + a2[2] + a3[3];
    return;
  }
  if (prio) __builtin_amdgcn_s_setprio(3);
  float v[64];
  for (int j = 0; j < 64; ++j) v[j] = tid * 0.5f + j;
  unsigned long long tot = 0;
  for (int r = 0; r < rounds; ++r) {
    const long long t0 = clock64();
    float* l = lds + 8704 + (tid >> 3) * 16 + (tid & 7) * 2;
    int step = 259 + (r & 1);
    asm volatile("" : "+s"(step));
#pragma unroll
    for (int j = 0; j < 32; ++j) { l[0] = v[2 * j]; l[1] = v[2 * j + 1]; l += step; if (j == 15) l -= 16 * step - 512; }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    const long long t1 = clock64();
    tot += t1 - t0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 64; ++j) v[j] += 1.0f;
  }
  out[blockIdx.x * 256 + tid] = v[3];
  if (lane == 0) { atomicAdd(&g_c[0], tot); atomicAdd(&g_c[1], 1ull); }
}

int main() {
  float* out;
  (void)hipMalloc(&out, 512 * 256 * 4);
  for (int prio = 0; prio < 2; ++prio)
    for (int mode = 0; mode < 2; ++mode) {
      unsigned long long z[4] = {0, 0, 0, 0}, r[4];
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_c), z, sizeof(z));
      const int rounds = 200;
      k<<<512, 256>>>(out, rounds, mode, prio);
      (void)hipDeviceSynchronize();
      (void)hipMemcpyFromSymbol(r, HIP_SYMBOL(g_c), sizeof(r));
      printf("store phase (32 x ds_write2 + v_add): %.0f cycles  [%s, setprio %d]\n", (double)r[0] / r[1] / rounds,
             mode ? "other workgroup of the CU in an MFMA + ds_read loop" : "alone on the CU", prio);
    }
  return 0;
}
