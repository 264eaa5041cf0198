// Checks the raw-buffer range-check semantics the SYRK staging relies on (gfx950):
//   (1) a lane whose voffset is 0x80000000 reads 0; (2) voffset + soffset past num_records reads 0;
//   (3) in-range lanes read src[(voffset + soffset) / 4].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* src, float* out, int nbytes, int step) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  int voff = threadIdx.x * 4;
  if (threadIdx.x & 1) voff = 0x80000000;
  int so = 0;
  for (int j = 0; j < 8; ++j) {
    out[j * 64 + threadIdx.x] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, so, 0));
    so += step;
  }
  f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16, 0, 0));
  out[8 * 64 + threadIdx.x] = v.x + v.y + v.z + v.w;
}
int main() {
  const int n = 300;   // floats in the buffer; 8 steps of 40 floats walk past the end
  std::vector<float> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = (float)(i + 1);
  float *d, *o;
  (void)hipMalloc(&d, 4096 * 4); (void)hipMalloc(&o, 9 * 64 * 4);
  (void)hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, o, n * 4, 40 * 4);
  std::vector<float> r(9 * 64);
  (void)hipMemcpy(r.data(), o, 9 * 64 * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int j = 0; j < 8; ++j)
    for (int t = 0; t < 64; ++t) {
      const int idx = t + 40 * j;
      const float want = (t & 1) ? 0.f : (idx < n ? (float)(idx + 1) : 0.f);
      if (r[j * 64 + t] != want) { if (bad < 10) printf("j=%d t=%d got %g want %g\n", j, t, r[j * 64 + t], want); ++bad; }
    }
  for (int t = 0; t < 64; ++t) {
    float want = 0;
    for (int e = 0; e < 4; ++e) { const int idx = 4 * t + e; want += (4 * t + 3 < n) ? (float)(idx + 1) : 0.f; }
    // a partially out-of-range x4 load: report what the hardware does
    if (r[8 * 64 + t] != want && 4 * t < n && 4 * t + 3 >= n) printf("partial x4 at t=%d: got %g\n", t, r[8 * 64 + t]);
    else if (r[8 * 64 + t] != want) { printf("x4 t=%d got %g want %g\n", t, r[8 * 64 + t], want); ++bad; }
  }
  printf("buffer_oob: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
  return bad != 0;
}
