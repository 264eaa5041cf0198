// Shared helpers for the gfx950 kernels behind include/curv_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/curv_hip.h"
#include <cstdint>
#include <cstdio>
#include <cstring>

namespace curv {

// Status codes (CURV_OK, CURV_ERR_*) are the macros of include/curv_hip.h.

void set_error(const char* fmt, ...);

#define CURV_HIP_CHECK(expr)                                                         \
  do {                                                                               \
    hipError_t _e = (expr);                                                          \
    if (_e != hipSuccess) {                                                          \
      ::curv::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),       \
                        __FILE__, __LINE__);                                         \
      return CURV_ERR_HIP;                                                   \
    }                                                                                \
  } while (0)

#define CURV_LAUNCH_CHECK() CURV_HIP_CHECK(hipGetLastError())

#define CURV_REQUIRE(cond, ...)                                                      \
  do {                                                                               \
    if (!(cond)) {                                                                   \
      ::curv::set_error(__VA_ARGS__);                                                \
      return CURV_ERR_INVALID;                                               \
    }                                                                                \
  } while (0)

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long long cdivll(long long a, long long b) { return (a + b - 1) / b; }

// eigh.hip: the block-Jacobi iteration on whole matrices (curv_syevd = eigh_lowrank.hip's projection driver in front of it)
size_t syevd_jacobi_workspace_bytes(const curv_eigh_desc* descs, int n_mats);
int syevd_jacobi(hipStream_t stream, const curv_eigh_desc* descs, int n_mats, void* workspace, size_t workspace_bytes,
                 int max_sweeps, double tol, int* sweeps_done);

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

}  // namespace curv
