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
