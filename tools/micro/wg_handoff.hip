// Latency of handing a 64x64 fp64 tile from one workgroup to another inside a running kernel (flag in global memory),
// the building block of a chain kernel whose workgroups wait for each other instead of for the next launch.
// Variants: coherent accesses (relaxed agent-scope atomics = sc1 loads/stores, no cache maintenance) vs ordinary
// accesses bracketed by agent-scope release / acquire fences.  Ping-pong between workgroup 0 and workgroup `peer`
// (blockIdx % 8 = XCD).  Diagnostics only.   hipcc --offload-arch=gfx950 -O3 wg_handoff.hip -o wg_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#define SPIN_LIMIT (1 << 22)
__device__ __forceinline__ bool wait_flag(int* flag, int want) {
  int n = 0;
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
    __builtin_amdgcn_s_sleep(1);
    if (++n > SPIN_LIMIT) return false;
  }
  return true;
}
template <int MODE>
__global__ void __launch_bounds__(256) k(double* tile_a, double* tile_b, int* flags, int peer, int iters, long long* cyc, double* sink) {
  const int tid = threadIdx.x;
  const bool first = blockIdx.x == 0;
  if (!first && (int)blockIdx.x != peer) return;
  __shared__ int ok;
  double acc = 0.0;
  long long t0 = wall_clock64();
  for (int it = 1; it <= iters; ++it) {
    for (int half = 0; half < 2; ++half) {
      const bool producer = (half == 0) == first;
      double* tile = half == 0 ? tile_a : tile_b;
      int* flag = flags + half;
      if (producer) {
        for (int e = tid; e < 4096; e += 256) {
          const double v = (double)(it + e) + acc * 1e-30;
          if (MODE == 0) __hip_atomic_store(tile + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else tile[e] = v;
        }
        if (MODE == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        else __threadfence();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(flag, it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        if (tid == 0) ok = wait_flag(flag, it) ? 1 : 0;
        __syncthreads();
        if (!ok) { if (tid == 0) cyc[1] = -1; return; }
        if (MODE == 1) __threadfence();
        for (int e = tid; e < 4096; e += 256)
          acc += MODE == 0 ? __hip_atomic_load(tile + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tile[e];
      }
    }
  }
  long long t1 = wall_clock64();
  if (tid == 0 && first) cyc[0] = t1 - t0;
  sink[blockIdx.x * 256 + tid] = acc;
}
template <int MODE> void run(const char* name, int peer) {
  double *ta, *tb, *sink; int* flags; long long* cyc;
  hipMalloc(&ta, 4096 * 8); hipMalloc(&tb, 4096 * 8); hipMalloc(&sink, 64 * 256 * 8); hipMalloc(&flags, 8); hipMalloc(&cyc, 16);
  const int iters = 200;
  long long c[2] = {0, 0};
  for (int rep = 0; rep < 2; ++rep) {
    hipMemset(flags, 0, 8); hipMemset(cyc, 0, 16);
    hipLaunchKernelGGL(k<MODE>, dim3(peer + 1), dim3(256), 0, 0, ta, tb, flags, peer, iters, cyc, sink);
    hipDeviceSynchronize();
    hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
  }
  printf("%-44s peer block %2d: %7.2f us per hand-off (write tile, flag, wait, read tile)%s\n", name, peer,
         (double)c[0] / 100.0 / iters / 2, c[1] ? "  TIMED OUT" : "");
  hipFree(ta); hipFree(tb); hipFree(sink); hipFree(flags); hipFree(cyc);
}
int main() {
  for (int peer : {8, 1, 9}) {
    run<0>("coherent accesses (sc1), no cache maintenance", peer);
    run<1>("ordinary accesses + agent-scope fences", peer);
  }
  return 0;
}
