// Error plumbing + version for libse_hip.so.
#include "se_common.h"
#include <stdarg.h>

// per-thread: the entry points are re-entrant (include/se_hip.h conventions); an error text belongs to the calling thread
thread_local char g_se_err[512] = "";

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

// workspace sizes of the entry points that take a caller-owned workspace (include/se_hip.h)
extern "C" size_t se_norm_prelu_bwd_workspace_bytes(int B, int C, int per_batch) {
  return (B > 0 && C > 0) ? (size_t)(per_batch ? B : 1) * C * 3 * sizeof(double) : 0;
}
extern "C" size_t se_segnorm_workspace_bytes(int nseg) { return nseg > 0 ? (size_t)nseg * 2 * sizeof(double) : 0; }
