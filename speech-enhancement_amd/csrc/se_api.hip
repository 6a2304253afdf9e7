// Error plumbing + version for libse_hip.so.
#include "se_common.h"
#include <stdarg.h>

char g_se_err[512] = "";

int se_fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_se_err, sizeof(g_se_err), fmt, ap);
  va_end(ap);
  return -1;
}

int se_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return se_fail("%s: launch failed: %s", what, hipGetErrorString(e));
  return 0;
}

extern "C" int se_version(void) { return 1; }
extern "C" const char* se_last_error(void) { return g_se_err; }
