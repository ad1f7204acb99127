// common.cpp — error state + ABI version for libenh_hip.so
#include <stdarg.h>
#include <string.h>
#include "common.h"

static thread_local char g_err[512] = "";

void enh_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int enh_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    enh_set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
    return ENH_E_HIP_BASE - (int)e;
  }
  return ENH_OK;
}

extern "C" const char* enh_last_error(void) { return g_err; }
extern "C" int enh_abi_version(void) { return ENH_ABI_VERSION; }
