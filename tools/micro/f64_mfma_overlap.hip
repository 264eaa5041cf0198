// Does v_mfma_f64_16x16x4_f64 leave room for the other waves of its SIMD?  (On gfx950 its rate equals the
// vector-FP64 rate, which raises the suspicion that it runs on the vector ALUs.)
// 256 workgroups x 512 threads = 2 waves per SIMD on every CU.  Waves 0-3 (one per SIMD) run a chain of
// independent f64 MFMAs, waves 4-7 run one of: nothing / FP32 VALU / FP64 VALU / LDS reads / global loads.
// Reported: wall cycles of the MFMA waves and of the other waves, alone and together.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ unsigned long long g_c[8];

__global__ void __launch_bounds__(512) k(double* out, const double* src, int rounds, int mfma_on, int other) {
  __shared__ double lds[8192];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 8192; i += 512) lds[i] = 1e-3 * (i & 31);
  __syncthreads();
  const long long t0 = clock64();
  if (wave < 4) {
    if (!mfma_on) return;
    if (mfma_on == 2) {                   // the fp32 MFMA of the factor build, same 64-cycle occupancy
      f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
      float xf = 1.0f + lane * 1e-3f, yf = 0.5f;
      for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(xf, yf, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(xf, yf, c1, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(xf, yf, c2, 0, 0, 0);
          c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(xf, yf, c3, 0, 0, 0);
        }
      }
      const long long t1 = clock64();
      out[blockIdx.x * 512 + tid] = c0[0] + c1[1] + c2[2] + c3[3];
      if (lane == 0) { atomicAdd(&g_c[0], (unsigned long long)(t1 - t0)); atomicAdd(&g_c[1], 1ull); }
      return;
    }
    f64x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x = 1.0 + lane * 1e-3, y = 0.5;
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
      }
    }
    const long long t1 = clock64();
    out[blockIdx.x * 512 + tid] = a0[0] + a1[1] + a2[2] + a3[3];
    if (lane == 0) { atomicAdd(&g_c[0], (unsigned long long)(t1 - t0)); atomicAdd(&g_c[1], 1ull); }
    return;
  }
  if (other == 0) return;
  double acc = 0.0;
  float facc = 0.0f;
  if (other == 1) {                       // FP32 VALU: 64 dependent-free FMAs per round
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = tid + j;
    for (int r = 0; r < rounds; ++r)
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] * 1.0001f + 0.5f;
    for (int j = 0; j < 8; ++j) facc += v[j];
  } else if (other == 2) {                // FP64 VALU
    double v[8];
    for (int j = 0; j < 8; ++j) v[j] = tid + j;
    for (int r = 0; r < rounds; ++r)
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] * 1.0001 + 0.5;
    for (int j = 0; j < 8; ++j) acc += v[j];
  } else if (other == 3) {                // LDS reads: 64 ds_read_b64 per round
    const double* p = lds + lane;
    for (int r = 0; r < rounds; ++r)
#pragma unroll
      for (int u = 0; u < 64; ++u) acc += p[(u * 64 + r) & 8191 & ~63];
  } else {                                // global loads: 16 dwordx2 per round, L2-resident
    const double* p = src + (blockIdx.x * 512 + tid);
    for (int r = 0; r < rounds; ++r)
#pragma unroll
      for (int u = 0; u < 16; ++u) acc += p[((r * 16 + u) & 63) * 131072];
  }
  const long long t1 = clock64();
  out[blockIdx.x * 512 + tid] = acc + facc;
  if (lane == 0) { atomicAdd(&g_c[2], (unsigned long long)(t1 - t0)); atomicAdd(&g_c[3], 1ull); }
}

int main() {
  double *out, *src;
  (void)hipMalloc(&out, 256 * 512 * 8);
  (void)hipMalloc(&src, 64ull * 131072 * 8 + 256 * 512 * 8);
  (void)hipMemset(src, 0, 64ull * 131072 * 8 + 256 * 512 * 8);
  const char* names[5] = {"nothing", "FP32 VALU (64 v_fma_f32 / round)", "FP64 VALU (64 v_fma_f64 / round)",
                          "LDS (64 ds_read_b64 / round)", "global loads (16 dwordx2 / round)"};
  const int rounds = 400;
  for (int other = 0; other < 5; ++other)
    for (int mfma_on = 0; mfma_on < 3; ++mfma_on) {
      if (!mfma_on && other == 0) continue;
      unsigned long long z[8] = {0}, r[8];
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_c), z, sizeof(z));
      k<<<256, 512>>>(out, src, rounds, mfma_on, other);
      (void)hipDeviceSynchronize();
      (void)hipMemcpyFromSymbol(r, HIP_SYMBOL(g_c), sizeof(r));
      printf("MFMA waves %s, other waves: %-36s | per round: MFMA waves %7.0f cycles (32 MFMAs = 2048 at full rate), others %7.0f\n",
             mfma_on == 2 ? "f32" : mfma_on ? "f64" : "off", names[other], r[1] ? (double)r[0] / r[1] / rounds : 0.0, r[3] ? (double)r[2] / r[3] / rounds : 0.0);
    }
  return 0;
}
