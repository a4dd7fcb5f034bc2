// Error reporting and version of the C-ABI (include/relnet_hip.h).
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

namespace relnet {
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
    return -2;
  }
  return 0;
}
}  // namespace relnet

extern "C" const char* relnet_last_error(void) { return relnet::g_err; }
extern "C" int relnet_version(void) { return 100; }   // 0.1.0
