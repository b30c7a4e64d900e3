// Error string, version and device queries of libnerf_sampling_hip.so.
#include "ns_common.h"

#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>

namespace ns {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// per-device caches (a process may drive several GPUs; round-1 code cached the first device's answer)
static std::mutex g_mu;
static std::map<int, int> g_cus;
static std::map<std::pair<const void*, int>, size_t> g_dyn_lds;

int cu_count() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_cus.find(dev);
  if (it != g_cus.end()) return it->second;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  g_cus[dev] = prop.multiProcessorCount;
  return prop.multiProcessorCount;
}

// hipFuncAttributeMaxDynamicSharedMemorySize for `kernel` on the CURRENT device, raised whenever a launch needs more
// than any earlier one did (the LDS size of the MLP kernels grows with the network depth through the bias image)
hipError_t ensure_dynamic_lds(const void* kernel, size_t bytes) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(g_mu);
  size_t& have = g_dyn_lds[std::make_pair(kernel, dev)];
  if (bytes <= have) return hipSuccess;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
  if (e == hipSuccess) have = bytes;
  return e;
}

// Diagnostic switches: read from the environment ONCE (first use), changed afterwards only through ns_debug_set -- no
// getenv on the launch path.
DebugFlags& debug_flags() {
  static DebugFlags f = [] {
    DebugFlags d{};
    const char* v = std::getenv("NS_OB16_GENERIC");
    d.generic_kernels = (v && v[0] == '1') ? 1 : 0;
    v = std::getenv("NS_OB16_TILES");
    d.prod_tiles = (v && (v[0] == '4' || v[0] == '5')) ? v[0] - '0' : 0;
    return d;
  }();
  return f;
}

int& prod_tiles_hint() {
  static thread_local int hint = 0;
  return hint;
}

}  // namespace ns

extern "C" {
const char* ns_last_error(void) { return ns::g_err; }
int ns_debug_set(const char* name, int value) {
  if (!name) { ns::set_error("ns_debug_set: null name"); return NS_E_INVALID; }
  ns::DebugFlags& f = ns::debug_flags();
  if (!std::strcmp(name, "generic_kernels")) { f.generic_kernels = value ? 1 : 0; return NS_OK; }
  if (!std::strcmp(name, "hier_chain")) { f.hier_chain = value ? 1 : 0; return NS_OK; }
  if (!std::strcmp(name, "prod_tiles") && (value == 0 || value == 4 || value == 5)) { f.prod_tiles = value; return NS_OK; }
  ns::set_error("ns_debug_set: unknown switch or value (%s = %d)", name, value);
  return NS_E_INVALID;
}
int ns_version(void) { return 1; }
int ns_device_cu_count(void) { return ns::cu_count(); }
}
