#!/usr/bin/env python
"""Where the ~20 us of one 64x64 factor-and-invert (csrc/invert.hip factor_invert_64, on the serial chain of the
invert) go: builds a copy of invert.hip with clock probes in /tmp, runs it on one workgroup and prints the phases.
Diagnostics only."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "curvature_amd", "csrc")


def sub(s, a, b, count=1):
    assert a in s, a
    return s.replace(a, b, count)


def main():
    s = open(os.path.join(CSRC, "invert.hip")).read()
    s = sub(s, "namespace curv {\n", "namespace curv {\n__device__ unsigned long long g_probe[16];\n"
            "#define PROBE(k) { if (threadIdx.x == 0) { long long t_ = clock64(); g_probe[k] += (unsigned long long)(t_ - tprev); tprev = t_; } }\n")
    # probes inside factor_invert_64 (thread 0 = wave 0, lane 0): cycles between consecutive probe points, summed
    k = s.index("__device__ __forceinline__ void factor_invert_64(double* Ds, double* Is, int* bad, int pivot_base) {")
    head, body = s[:k], s[k:]
    body = sub(body, "  const int r16 = lane & 15, kq = lane >> 4;\n", "  const int r16 = lane & 15, kq = lane >> 4;\n  long long tprev = clock64();\n")
    body = sub(body, "    const int c0 = 16 * p;\n", "    const int c0 = 16 * p;\n    PROBE(7)\n")
    body = sub(body, "      if (first_bad != 0 && lane == 0 && *bad == 0) *bad = first_bad;\n    }\n    __syncthreads();\n",
               "      if (first_bad != 0 && lane == 0 && *bad == 0) *bad = first_bad;\n    }\n    PROBE(0)\n    __syncthreads();\n    PROBE(1)\n")
    body = sub(body, "    __syncthreads();\n  }\n  // behind the last panel", "    __syncthreads();\n    PROBE(2)\n  }\n  // behind the last panel")
    body = sub(body, "    diag_inverse(3);\n  } else {", "    diag_inverse(3);\n    PROBE(3)\n  } else {")
    body = sub(body, "  __syncthreads();\n  if (wave != 0) {\n    f64x4 out = {0.0, 0.0, 0.0, 0.0};\n    mma16<false>(Is + 16 * 3 * LDA + 16 * 3",
               "  __syncthreads();\n  PROBE(4)\n  if (wave != 0) {\n    f64x4 out = {0.0, 0.0, 0.0, 0.0};\n    mma16<false>(Is + 16 * 3 * LDA + 16 * 3")
    body = sub(body, "  __syncthreads();\n}\n\n// ------------------------------------------------------------------------------------------------\n// (1a)",
               "  __syncthreads();\n  PROBE(5)\n}\n\n// ------------------------------------------------------------------------------------------------\n// (1a)")
    s = head + body
    s += r'''
namespace curv {
__global__ void __launch_bounds__(INV_THREADS) probe_kernel(const double* __restrict__ A, double* __restrict__ X, unsigned long long* out) {
  __shared__ double Ds[NB * LDA];
  __shared__ double Is[NB * LDA];
  __shared__ int bad;
  const int tid = threadIdx.x;
  if (tid == 0) bad = 0;
  long long t0 = clock64();
  for (int e = tid; e < NB * NB; e += INV_THREADS) Ds[(e >> 6) * LDA + (e & 63)] = A[e];
  __syncthreads();
  long long t1 = clock64();
  factor_invert_64(Ds, Is, &bad, 0);
  long long t2 = clock64();
  for (int e = tid; e < NB * NB; e += INV_THREADS) X[e] = Is[(e >> 6) * LDA + (e & 63)];
  __syncthreads();
  long long t3 = clock64();
  if (tid == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; out[3] = bad; }
}
}
#include <cstdio>
#include <vector>
#include <cmath>
namespace curv { void set_error(const char*, ...) {} }
int main() {
  const int n = 64;
  std::vector<double> A(n * n);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[i * n + j] = (i == j ? 2.0 + 0.01 * i : 0.0) + 0.5 / (1.0 + std::abs(i - j));
  double *dA, *dX; unsigned long long* dout;
  hipMalloc(&dA, n * n * 8); hipMalloc(&dX, n * n * 8); hipMalloc(&dout, 64);
  hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
  for (int it = 0; it < 3; ++it) {
    unsigned long long z[16] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(curv::g_probe), z, sizeof(z));
    hipLaunchKernelGGL(curv::probe_kernel, dim3(1), dim3(curv::INV_THREADS), 0, 0, dA, dX, dout);
    hipDeviceSynchronize();
    unsigned long long o[8], p[16];
    hipMemcpy(o, dout, 32, hipMemcpyDeviceToHost);
    hipMemcpyFromSymbol(p, HIP_SYMBOL(curv::g_probe), sizeof(p));
    printf("run %d: load %llu  factor_invert_64 %llu  store %llu cycles (clock64 = 100 MHz ticks? see below) bad %llu\n", it, o[0], o[1], o[2], o[3]);
    printf("   panel columns (wave 0, 4 panels) %llu | barrier after (incl. waiting for the side work of waves 1-3) %llu | trailing MFMA + barrier %llu | pre-panel %llu | X_33 %llu | barrier %llu | last products + barrier %llu\n",
           p[0], p[1], p[2], p[7], p[3], p[4], p[5]);
  }
  // check: X * L = I where L = chol(A) -> X A X^T = I
  std::vector<double> X(n * n);
  hipMemcpy(X.data(), dX, n * n * 8, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
    double s = 0;
    for (int a = 0; a < n; ++a) for (int b = 0; b < n; ++b) s += X[i * n + a] * A[a * n + b] * X[j * n + b];
    worst = std::fmax(worst, std::fabs(s - (i == j)));
  }
  printf("max |X A X^T - I| = %.3e\n", worst);
  return 0;
}
'''
    src = "/tmp/probe_factor64.hip"
    open(src, "w").write(s)
    exe = "/tmp/probe_factor64"
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, src, "-o", exe]
    subprocess.check_call(cmd)
    if "--build-only" not in sys.argv:
        subprocess.check_call([exe])


if __name__ == "__main__":
    main()
