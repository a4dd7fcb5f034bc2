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

// Compute units of the current device (256 on an MI355X), cached per device: the grid size of the persistent kernels.
long device_cu_count() {
  static int cached[64] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  int& c = cached[dev & 63];
  if (c == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    c = n;
  }
  return c;
}
}  // namespace relnet

extern "C" const char* relnet_last_error(void) { return relnet::g_err; }
// 0 when `stream` is not being captured into a hipGraph, else the id of that capture (unique per capture sequence, hipStreamGetCaptureInfo):
// lets the host side tell two captures on one stream apart (per-capture scratch such as the split-K work area of relnet_gemm_set_workspace)
extern "C" unsigned long long relnet_stream_capture_id(void* stream) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  if (hipStreamGetCaptureInfo((hipStream_t)stream, &st, &id) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return st == hipStreamCaptureStatusActive ? (id ? id : ~0ull) : 0;
}
extern "C" int relnet_version(void) { return 100; }   // 0.1.0
