// Version and error plumbing of the C ABI (include/curv_hip.h).
#include "common.h"
#include "../../include/curv_hip.h"
#include <cstdarg>

namespace curv {
static thread_local char g_error[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}
}  // namespace curv

extern "C" int curv_version(void) { return CURV_ABI_VERSION; }
extern "C" const char* curv_last_error(void) { return curv::g_error; }

// HIP event helpers so that a host without HIP bindings (Python/ctypes) can time a kernel on the
// stream it is launched on (bench.py's roofline leg).
extern "C" void* curv_event_create(void) {
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return (void*)e;
}
extern "C" void curv_event_destroy(void* e) {
  if (e) (void)hipEventDestroy((hipEvent_t)e);
}
extern "C" int curv_event_synchronize(void* e) {
  CURV_REQUIRE(e != nullptr, "curv_event_synchronize: null argument");
  CURV_HIP_CHECK(hipEventSynchronize((hipEvent_t)e));
  return CURV_OK;
}
extern "C" int curv_event_elapsed_ms(void* start, void* stop, float* ms) {
  CURV_REQUIRE(start && stop && ms, "curv_event_elapsed_ms: null argument");
  CURV_HIP_CHECK(hipEventSynchronize((hipEvent_t)stop));
  CURV_HIP_CHECK(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return CURV_OK;
}
