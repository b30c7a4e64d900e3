// Shared host-side helpers for libnerf_sampling_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/nerf_sampling_hip.h"

namespace ns {

void set_error(const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// ceil-div on 64-bit sizes
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// grid for grid-stride elementwise kernels: enough blocks to fill 256 CUs x 8, capped
inline int ew_grid(int64_t n, int block) {
  int64_t g = cdiv(n, block);
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return static_cast<int>(g);
}

// diagnostic switches (ns_debug_set; initial values from NS_OB16_GENERIC / NS_OB16_TILES, read once)
struct DebugFlags {
  int generic_kernels;   // 1: the production network runs the generic, compiler-scheduled kernels (tests compare the two)
  int prod_tiles;        // 4 / 5: tiles per wave of the 16-bit production kernel; 0 = chosen per launch
  int hier_chain;        // 1: ns_render_rays_hierarchical keeps raw [R,N,4] in HBM and composites with the stand-alone kernel
};
DebugFlags& debug_flags();
// Per-thread launch hint of the one-call renderers (ns_render.cpp): 4 = this call's per-sample outputs are about to be
// copied to the host while the NEXT call's MLP kernel runs -- the four-tile production kernel leaves 64+ registers of every
// SIMD free, so the blit kernels ROCm moves pinned device-to-host copies with can run beside it; the five-tile kernel
// (2.5 % faster alone) fills the register file and the copies would wait for the whole persistent grid.  0 = no preference.
int& prod_tiles_hint();

int cu_count();
hipError_t ensure_dynamic_lds(const void* kernel, size_t bytes);

}  // namespace ns

// internal (not part of the public header)
extern "C" int ns_coarse_z_scalar(float near_, float far_, int64_t R, int N, int lindisp, const float* t_rand_dev,
                                  float* z_dev, void* stream);

#define NS_REQUIRE(cond, msg)                    \
  do {                                           \
    if (!(cond)) {                               \
      ns::set_error("%s: %s", __func__, msg);    \
      return NS_E_INVALID;                       \
    }                                            \
  } while (0)

#define NS_HIP(call)                                                           \
  do {                                                                         \
    hipError_t e_ = (call);                                                    \
    if (e_ != hipSuccess) {                                                    \
      ns::set_error("%s: %s -> %s", __func__, #call, hipGetErrorString(e_));   \
      return NS_E_HIP;                                                         \
    }                                                                          \
  } while (0)

#define NS_LAUNCH_CHECK()                                                      \
  do {                                                                         \
    hipError_t e_ = hipGetLastError();                                         \
    if (e_ != hipSuccess) {                                                    \
      ns::set_error("%s: launch -> %s", __func__, hipGetErrorString(e_));      \
      return NS_E_HIP;                                                         \
    }                                                                          \
  } while (0)
