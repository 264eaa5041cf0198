// Microbenchmark replicating the inner k-run loop of syrk_patch_kernel (operands gathered from LDS through
// per-lane bases + a k-run table) to find what limits its MFMA issue rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VARIANT>
__global__ void __launch_bounds__(256, 2) k(float* out, int chunks, int niter, int conflict) {
  __shared__ int smem[19984];
  float* fs = (float*)smem;
  int* ktab = smem + 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, r32 = lane & 31;
  for (int i = tid; i < 19984; i += 256) fs[i] = (float)((i * 7) & 15) * 0.01f;
  __syncthreads();
  for (int i = tid; i < 1024; i += 256) ktab[i] = ((i * 37) % 4000) | (3 << 20);
  __syncthreads();
  int bbase[4], kmask[4];
  for (int o = 0; o < 4; ++o) {
    // conflict = 0: 32 consecutive words (conflict-free); 1: stride 2 (2-way)
    bbase[o] = 4 * (2600 + o * 2100 + r32 * (conflict ? 2 : 1) * 33 % 2000);
    kmask[o] = -1;
  }
  f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
  const bool skip10 = (blockIdx.x & 15) == 0 && VARIANT != 3;
  struct Ops { float a0[2], a1[2], b0[2], b1[2]; int mask; };
  const char* lds = (const char*)fs;
  auto load_ops = [&](Ops& op, int e) {
    const int koff = (e & 0xfffff) * 4;
    op.mask = e >> 20;
    const int pa0 = bbase[0] + (koff & kmask[0]), pa1 = bbase[1] + (koff & kmask[1]);
    const int pb0 = bbase[2] + (koff & kmask[2]), pb1 = bbase[3] + (koff & kmask[3]);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      op.a0[j] = *(const float*)(lds + pa0 + j * 4);
      op.a1[j] = *(const float*)(lds + pa1 + j * 4);
      op.b0[j] = *(const float*)(lds + pb0 + j * 4);
      op.b1[j] = *(const float*)(lds + pb1 + j * 4);
    }
  };
  auto compute_ops = [&](Ops& op) {
    if (VARIANT != 2) {
      if (__ballot(op.mask != 3) != 0ull) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bool v = (op.mask >> j) & 1;
          op.a0[j] = v ? op.a0[j] : 0.0f;
          op.a1[j] = v ? op.a1[j] : 0.0f;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(op.a0[j], op.b0[j], acc00, 0, 0, 0);
      acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(op.a0[j], op.b1[j], acc01, 0, 0, 0);
      if (!skip10) acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(op.a1[j], op.b0[j], acc10, 0, 0, 0);
      acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(op.a1[j], op.b1[j], acc11, 0, 0, 0);
    }
    if (VARIANT != 1) __builtin_amdgcn_s_waitcnt(0xc07f);
  };
  for (int c = 0; c < chunks; ++c) {
    Ops A, B;
    const int last = 2 * niter - 1;
    int it = 0;
    load_ops(A, ktab[2 * it + h]);
    int e = ktab[min(2 * (it + 1) + h, last)];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    while (true) {
      const int it1 = it + 1;
      const bool n1 = it1 < niter;
      if (n1) { load_ops(B, e); e = ktab[min(2 * (it1 + 1) + h, last)]; }
      compute_ops(A);
      if (!n1) break;
      it = it1 + 1;
      const bool n2 = it < niter;
      if (n2) { load_ops(A, e); e = ktab[min(2 * (it + 1) + h, last)]; }
      compute_ops(B);
      if (!n2) break;
    }
    __syncthreads();
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc00[i] + acc01[i] + acc10[i] + acc11[i];
  out[blockIdx.x * 256 + tid] = s;
}

template <int V>
void run(const char* name, int conflict) {
  float* out;
  (void)hipMalloc(&out, 2048 * 256 * 4);
  const int chunks = 40, niter = 98, grid = 512;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<V><<<grid, 256>>>(out, 2, niter, conflict);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<V><<<grid, 256>>>(out, chunks, niter, conflict);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)grid * 4 * chunks * niter * 8 * 4096.0 * (1.0 - 1.0 / 16 / 4);
  printf("%-46s conflict=%d: %.3f ms  %.1f TFLOP/s\n", name, conflict, ms, flops / ms / 1e9);
  (void)hipFree(out);
}

int main() {
  run<0>("as in syrk.hip", 0);
  run<0>("as in syrk.hip", 1);
  run<1>("no lgkmcnt(0) after the MFMA group", 0);
  run<2>("no mask test / select", 0);
  run<3>("no skip10 branch (never diagonal)", 0);
  return 0;
}
