// Cost of the ways to hand one lane's fp64 value to a whole wave, as used by the 16-column panel loop of
// factor_invert_64 (csrc/invert.hip): cycles per (broadcast + v_fma_f64) for one wave alone on its SIMD.
// Diagnostics only.   hipcc --offload-arch=gfx950 -O3 lane_bcast_cost.hip -o lane_bcast_cost
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double readlane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
template <int MODE>
__global__ void __launch_bounds__(64) k(double* out, long long* cyc, int iters) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  __shared__ double sh_[64];
  __shared__ d2 sh2_[32];
  volatile __attribute__((address_space(3))) double* sh = (volatile __attribute__((address_space(3))) double*)sh_;
  volatile __attribute__((address_space(3))) d2* sh2 = (volatile __attribute__((address_space(3))) d2*)sh2_;
  const int lane = threadIdx.x;
  double r[16], t = 1.0 + lane * 1e-3;
  for (int i = 0; i < 16; ++i) r[i] = lane + i;
  sh[lane] = t;
  if (lane < 32) { d2 w = {t, t + 1.0}; sh2[lane] = w; }
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {          // 2 readlanes + fma per value
#pragma unroll
      for (int q = 0; q < 16; ++q) r[q] -= r[(q + 1) & 15] * readlane_f64(t, q);
    } else if (MODE == 1) {   // independent fma only
#pragma unroll
      for (int q = 0; q < 16; ++q) r[q] = fma(-t, r[q], r[q]);
    } else if (MODE == 2) {   // dependent fma chain
#pragma unroll
      for (int q = 0; q < 16; ++q) t = fma(-t, t, r[q]);
    } else if (MODE == 3) {   // LDS broadcast read (uniform address) + fma
#pragma unroll
      for (int q = 0; q < 16; ++q) r[q] -= r[(q + 1) & 15] * sh[q];
    } else if (MODE == 4) {   // v_fmac_f64_dpp row_newbcast
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        switch (q) {
#define C(Q) case Q: asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #Q " row_mask:0xf bank_mask:0xf" : "+v"(r[Q]) : "v"(t), "v"(r[(Q + 1) & 15])); break;
          C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15)
#undef C
        }
      }
    } else if (MODE == 5) {   // dependent chain: readlane -> rcp -> 3 fma -> fma
      double d = readlane_f64(t, it & 15);
      double y = __builtin_amdgcn_rcp(d);
      double e = fma(-d, y, 1.0);
      y = fma(y, fma(e, e, e), y);
      t = fma(-r[0], y, t);
    } else if (MODE == 6) {   // readlane only (2 per value), results summed on the scalar side
      int acc = 0;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc ^= __builtin_amdgcn_readlane(__double2hiint(r[q]), q) ^ __builtin_amdgcn_readlane(__double2loint(r[q]), q);
      r[0] += acc;
    } else if (MODE == 7) {   // ds_read_b128 broadcast: two values per LDS instruction
#pragma unroll
      for (int q = 0; q < 16; q += 2) {
        const d2 v = sh2[q >> 1];
        r[q] -= r[(q + 1) & 15] * v.x;
        r[q + 1] -= r[(q + 2) & 15] * v.y;
      }
    }
  }
  long long t1 = clock64();
  double s = t;
  for (int i = 0; i < 16; ++i) s += r[i];
  out[lane] = s;
  if (lane == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char* name, int per_iter) {
  double* out; long long* cyc;
  hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
  const int iters = 1000;
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, out, cyc, iters); hipDeviceSynchronize(); }
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-48s %8.1f cycles per item (%d items per iteration)\n", name, (double)c / iters / per_iter, per_iter);
}
int main() {
  run<0>("2 x v_readlane + v_fma_f64 (SGPR operand)", 16);
  run<1>("v_fma_f64, independent", 16);
  run<2>("v_fma_f64, dependent", 16);
  run<3>("ds_read_b64 (uniform address) + v_fma_f64", 16);
  run<7>("ds_read_b128 (uniform address) + 2 v_fma_f64, per value", 16);
  run<4>("v_fmac_f64_dpp row_newbcast", 16);
  run<5>("chain: readlane, rcp, 3 fma, fma", 1);
  run<6>("2 x v_readlane alone", 16);
  return 0;
}
