// Probe of hipStreamWaitValue32 / hipStreamWriteValue32 against a resident kernel (LAB_NOTEBOOK R5.18): is the pair supported,
// what does a hand-off cost each way?    hipcc --offload-arch=gfx950 -O3 -o stream_value_probe stream_value_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// resident kernel: for p = 1..n: wait ready >= p (bounded), work a little, publish done = p
__global__ void resident(volatile unsigned* ready, unsigned* done, int n, long long* stamps) {
  for (int p = 1; p <= n; ++p) {
    int spins = 0;
    while (__hip_atomic_load((unsigned*)ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned)p) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1 << 24)) { __hip_atomic_store(done, 0xffffffffu, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); return; }
    }
    stamps[2 * p] = wall_clock64();
    __hip_atomic_store(done, (unsigned)p, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    stamps[2 * p + 1] = wall_clock64();
  }
}
__global__ void tiny(long long* stamp) { *stamp = wall_clock64(); }

int main() {
  int can = 0;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  if (!can) return 0;
  const int n = 20;
  unsigned *ready, *done;
  long long *stamps, *kst;
  CK(hipMalloc(&ready, 256)); CK(hipMalloc(&done, 256));
  CK(hipMalloc(&stamps, (2 * n + 4) * 8)); CK(hipMalloc(&kst, (n + 2) * 8));
  CK(hipMemset(ready, 0, 256)); CK(hipMemset(done, 0, 256)); CK(hipMemset(stamps, 0, (2 * n + 4) * 8)); CK(hipMemset(kst, 0, (n + 2) * 8));
  hipStream_t chain, mid;
  CK(hipStreamCreateWithFlags(&chain, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&mid, hipStreamNonBlocking));
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(resident, dim3(1), dim3(64), 0, chain, (volatile unsigned*)ready, done, n, stamps);
  // mid: release panel 1, then per panel: wait done >= p, run a tiny kernel, release panel p + 1
  CK(hipStreamWriteValue32(mid, ready, 1, 0));
  for (int p = 1; p <= n; ++p) {
    CK(hipStreamWaitValue32(mid, done, (unsigned)p, hipStreamWaitValueGte, 0xffffffffu));
    hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, mid, kst + p);
    CK(hipStreamWriteValue32(mid, ready, (unsigned)(p + 1), 0));
  }
  auto t0 = std::chrono::steady_clock::now();
  CK(hipStreamSynchronize(mid)); CK(hipStreamSynchronize(chain));
  double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  long long hs[2 * n + 4], hk[n + 2];
  CK(hipMemcpy(hs, stamps, sizeof(hs), hipMemcpyDeviceToHost)); CK(hipMemcpy(hk, kst, sizeof(hk), hipMemcpyDeviceToHost));
  printf("host wait %.3f ms\n", ms);
  for (int p = 2; p <= n; ++p)
    printf("panel %2d: resident published done at +0 | tiny kernel on the other stream ran %.1f us later | resident saw ready %.1f us after that kernel\n",
           p - 1, (hk[p - 1] - hs[2 * (p - 1) + 1]) / 100.0, (hs[2 * p] - hk[p - 1]) / 100.0);
  return 0;
}
