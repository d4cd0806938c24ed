// Error text, ABI version and device queries of libspr_hip.so.
#include <stdarg.h>
#include <string.h>

#include "common.hpp"

namespace {
thread_local char g_err[512] = "";
}

void spr_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int spr_cached_cus() {
  static int cus = 0;
  if (cus) return cus;
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  cus = prop.multiProcessorCount;
  return cus;
}

extern "C" int spr_abi_version(void) { return 1; }

extern "C" const char *spr_last_error(void) { return g_err; }

extern "C" int spr_device_cus(int *out_cus) {
  SPR_REQUIRE(out_cus != nullptr, SPR_E_INVALID, "spr_device_cus: NULL output");
  const int c = spr_cached_cus();
  SPR_REQUIRE(c > 0, SPR_E_HIP, "spr_device_cus: no HIP device");
  *out_cus = c;
  return SPR_OK;
}
