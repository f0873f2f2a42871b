// Error reporting shared by every entry point of libcat_hip.
#include "common.h"
#include <stdarg.h>
#include <string.h>

namespace cat {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return -5;
  }
  return 0;
}
}  // namespace cat

extern "C" {
const char* cat_hip_last_error(void) { return cat::g_err; }
int cat_hip_version(void) { return 1; }
}
