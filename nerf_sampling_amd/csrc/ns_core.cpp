// Error string, version and device queries of libnerf_sampling_hip.so.
#include "ns_common.h"

#include <cstring>

namespace ns {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int cu_count() {
  static int cached = -1;
  if (cached >= 0) return cached;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  cached = prop.multiProcessorCount;
  return cached;
}

}  // namespace ns

extern "C" {
const char* ns_last_error(void) { return ns::g_err; }
int ns_version(void) { return 1; }
int ns_device_cu_count(void) { return ns::cu_count(); }
}
