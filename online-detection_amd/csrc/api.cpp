// Error text, version and device queries of the libodx C ABI (include/odx.h).
#include "odx_common.h"
#include <string.h>

namespace odx {
static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}
}  // namespace odx

extern "C" const char* odx_last_error_string(void) { return odx::g_error; }

extern "C" int odx_version(void) { return 100; }

extern "C" int odx_device_cus(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    odx::set_error("odx_device_cus: no HIP device");
    return ODX_ERR_HIP;
  }
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
    odx::set_error("odx_device_cus: attribute query failed");
    return ODX_ERR_HIP;
  }
  return cus;
}
