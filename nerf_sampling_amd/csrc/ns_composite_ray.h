// Per-ray alpha compositing as lane-level building blocks: raw2outputs_kernel (ns_composite.hip) and the epilogue of the
// one-kernel renderer (ns_nerf_mlp_ob16.hip) both run exactly this code, so they agree bit for bit.
// raw2alpha + raw2outputs: nerf_utils.py:27-42, sampling_trainer.py:153-230.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace nscomp {

// running sums of one ray, one lane's share
struct RayAccum {
  float carry = 1.0f;   // transmittance entering the current chunk of SW samples
  float r = 0.f, g = 0.f, b = 0.f, depth = 0.f, acc = 0.f;
};

// ‖d‖ as torch.norm computes it on the CPU (fma chain)
__device__ __forceinline__ float ray_norm(float dx, float dy, float dz) {
  return sqrtf(__builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)));
}

// ---- wave-level scans on DPP (gfx9 data-parallel primitives: the shifted operand is read by the VALU instruction itself, no
// LDS round trip; a __shfl is a ds_bpermute with ~100 cycles of latency, and the scan is a dependent chain of them).
// dpp<CTRL>(old, x): lane i receives x of the lane CTRL selects, or `old` where that lane does not exist / the row is masked.
constexpr int kRowShr1 = 0x111, kRowShr2 = 0x112, kRowShr4 = 0x114, kRowShr8 = 0x118, kRowBcast15 = 0x142, kRowBcast31 = 0x143,
              kWaveShr1 = 0x138;
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp(float old, float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, x), CTRL,
                                                               ROW_MASK, 0xF, false));
}
// Inclusive scan over segments of SW consecutive lanes (SW a power of two <= 64, segments aligned), `sub` = lane % SW.
// Kogge-Stone inside the 16-lane rows (row_shr 1, 2, 4, 8), then the row totals: lane 15 of a row into the next row
// (row_bcast15, rows 1 and 3), lane 31 into the upper half (row_bcast31).  OP(a, b): a is the EARLIER operand.
template <int SW, class OP>
__device__ __forceinline__ float seg_scan(float x, float identity, int sub, OP op) {
  auto step = [&](auto ctrl_, int dlt) {
    const float up = dpp<decltype(ctrl_)::value>(identity, x);
    // inside a row the shift itself stops at the row's first lane; segments shorter than a row need the lane test
    if (SW >= 16 || sub >= dlt) x = op(up, x);
  };
  if constexpr (SW >= 2) step(std::integral_constant<int, kRowShr1>{}, 1);
  if constexpr (SW >= 4) step(std::integral_constant<int, kRowShr2>{}, 2);
  if constexpr (SW >= 8) step(std::integral_constant<int, kRowShr4>{}, 4);
  if constexpr (SW >= 16) step(std::integral_constant<int, kRowShr8>{}, 8);
  if constexpr (SW >= 32) x = op(dpp<kRowBcast15, 0xA>(identity, x), x);
  if constexpr (SW >= 64) x = op(dpp<kRowBcast31, 0xC>(identity, x), x);
  return x;
}

// One chunk of SW consecutive samples of a ray, one sample per lane (lane `sub` of the SW-lane group).
//   ok: this lane holds a real sample;  q = raw (rgb, sigma);  zi = its depth;  dist_raw = z[i+1] - z[i] or 1e10 for
//   the last sample (before the ‖d‖ scaling);  noise = raw_noise_std * randn or 0.
// Returns alpha and weight of the lane's sample and accumulates the ray sums.
template <int SW>
__device__ __forceinline__ void composite_chunk(RayAccum& A, bool ok, int sub, float4 q, float zi, float dist_raw,
                                                float norm, float noise, bool has_noise, float& alpha_out,
                                                float& w_out) {
  float alpha = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
  if (ok) {
    const float dist = dist_raw * norm;
    float sigma = q.w;
    if (has_noise) sigma += noise;
    alpha = 1.0f - expf(-fmaxf(sigma, 0.0f) * dist);
    if (sigma != sigma) alpha = sigma;  // relu(NaN) is NaN in torch
    cr = 1.0f / (1.0f + expf(-q.x));
    cg = 1.0f / (1.0f + expf(-q.y));
    cb = 1.0f / (1.0f + expf(-q.z));
  }
  // inclusive product scan of (1 - alpha + 1e-10) over the SW lanes of this ray
  const float p = seg_scan<SW>(ok ? (1.0f - alpha) + 1e-10f : 1.0f, 1.0f, sub, [](float a, float b) { return a * b; });
  float excl = dpp<kWaveShr1>(1.0f, p);          // the previous lane's inclusive product (lane 0: 1)
  if (sub == 0) excl = 1.0f;
  const float T = A.carry * excl;
  const float w = alpha * T;
  if constexpr (SW == 64)                        // (shorter segments are whole rays: nothing is carried)
    A.carry = A.carry * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p), 63));
  if (ok) {
    A.r += w * cr; A.g += w * cg; A.b += w * cb;
    A.depth += w * zi;
    A.acc += w;
  }
  alpha_out = alpha;
  w_out = w;
}

// Reduce the lanes' shares: the LAST lane of each SW-lane group (sub == SW - 1) ends up with the ray's totals in A and its
// disparity in `disp` (inclusive add-scans on DPP: five independent chains of log2 SW instructions, no LDS traffic).
template <int SW>
__device__ __forceinline__ void composite_finish(RayAccum& A, int white_bkgd, float& disp, int sub) {
  auto add = [](float a, float b) { return a + b; };
  A.r = seg_scan<SW>(A.r, 0.0f, sub, add);
  A.g = seg_scan<SW>(A.g, 0.0f, sub, add);
  A.b = seg_scan<SW>(A.b, 0.0f, sub, add);
  A.depth = seg_scan<SW>(A.depth, 0.0f, sub, add);
  A.acc = seg_scan<SW>(A.acc, 0.0f, sub, add);
  const float q = A.depth / (A.acc + 1e-10f);
  disp = 1.0f / ((q != q) ? q : fmaxf(1e-10f, q));   // torch.max(1e-10, q) propagates NaN
  if (white_bkgd) { A.r += 1.0f - A.acc; A.g += 1.0f - A.acc; A.b += 1.0f - A.acc; }
}

}  // namespace nscomp
