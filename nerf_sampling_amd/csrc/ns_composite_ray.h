// Per-ray alpha compositing of raw2outputs_kernel (ns_composite.hip), kept as lane-level building blocks.
// (Round 1 also ran them as an epilogue of the 16-bit NeRF kernel for N == 64 -- a wave's two tiles are one ray --
// bit-identical and raw never reached HBM, but at one wave per SIMD the ~200 epilogue instructions per ray are
// exposed: the kernel grew by exactly the 0.25 ms the separate launch costs, so the fusion was not kept.)
// raw2alpha + raw2outputs: nerf_utils.py:27-42, sampling_trainer.py:153-230.
#pragma once
#include <hip/hip_runtime.h>

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
  float p = ok ? (1.0f - alpha) + 1e-10f : 1.0f;
#pragma unroll
  for (int dlt = 1; dlt < SW; dlt <<= 1) {
    const float up = __shfl_up(p, dlt, SW);
    if (sub >= dlt) p *= up;
  }
  float excl = __shfl_up(p, 1, SW);
  if (sub == 0) excl = 1.0f;
  const float T = A.carry * excl;
  const float w = alpha * T;
  A.carry = A.carry * __shfl(p, SW - 1, SW);
  if (ok) {
    A.r += w * cr; A.g += w * cg; A.b += w * cb;
    A.depth += w * zi;
    A.acc += w;
  }
  alpha_out = alpha;
  w_out = w;
}

// reduce the lanes' shares; lane 0 of the group (at least) ends up with the ray's totals
template <int SW>
__device__ __forceinline__ void composite_finish(RayAccum& A, int white_bkgd, float& disp) {
  if constexpr (SW == 64) {
    // Five sums over 64 lanes.  A plain butterfly is 5 x 6 ds_bpermute shuffles per ray, and those -- not the loads --
    // bound the kernel; here each halving step also halves the lanes a value lives on (r, g, b to the lower half,
    // depth, acc to the upper one, ...), so 9 shuffles do the five reductions and 4 more gather the totals in lane 0.
    const int lane = static_cast<int>(threadIdx.x) & 63;
    const bool up32 = lane & 32, up16 = lane & 16, up8 = lane & 8;
    // step 32: lower half collects r, g, b; upper half depth, acc
    const float x0 = __shfl_xor(up32 ? A.r : A.depth, 32, 64);
    const float x1 = __shfl_xor(up32 ? A.g : A.acc, 32, 64);
    const float x2 = __shfl_xor(A.b, 32, 64);
    float v0 = up32 ? A.depth + x0 : A.r + x0;      // lower: r      upper: depth
    float v1 = up32 ? A.acc + x1 : A.g + x1;        // lower: g      upper: acc
    float v2 = A.b + x2;                            // lower: b      (upper: unused)
    // step 16: lower half: lanes 0-15 keep r, g; 16-31 keep b.  upper half: 32-47 keep depth; 48-63 keep acc
    const float y0 = __shfl_xor(up32 ? (up16 ? v0 : v1) : (up16 ? v0 : v2), 16, 64);
    const float y1 = __shfl_xor(v1, 16, 64);
    //   lane groups now:  [0,16): a = r, b = g   [16,32): a = b   [32,48): a = depth   [48,64): a = acc
    float a = up32 ? (up16 ? v1 + y0 : v0 + y0) : (up16 ? v2 + y0 : v0 + y0);
    float b = v1 + y1;                              // meaningful in [0,16) only
    // step 8: [0,8) keeps r, [8,16) keeps g; the other groups just halve
    const bool first16 = !up32 && !up16;
    const float z0 = __shfl_xor(first16 ? (up8 ? a : b) : a, 8, 64);
    a = first16 ? (up8 ? b + z0 : a + z0) : a + z0;
    // steps 4, 2, 1: one value per lane
    a += __shfl_xor(a, 4, 64);
    a += __shfl_xor(a, 2, 64);
    a += __shfl_xor(a, 1, 64);
    // totals: r in lanes [0,8), g in [8,16), b in [16,32), depth in [32,48), acc in [48,64)
    A.r = a;
    A.g = __shfl(a, 8, 64);
    A.b = __shfl(a, 16, 64);
    A.depth = __shfl(a, 32, 64);
    A.acc = __shfl(a, 48, 64);
  } else {
#pragma unroll
    for (int m = SW >> 1; m > 0; m >>= 1) {
      A.r += __shfl_xor(A.r, m, SW); A.g += __shfl_xor(A.g, m, SW); A.b += __shfl_xor(A.b, m, SW);
      A.depth += __shfl_xor(A.depth, m, SW); A.acc += __shfl_xor(A.acc, m, SW);
    }
  }
  const float q = A.depth / (A.acc + 1e-10f);
  disp = 1.0f / ((q != q) ? q : fmaxf(1e-10f, q));   // torch.max(1e-10, q) propagates NaN
  if (white_bkgd) { A.r += 1.0f - A.acc; A.g += 1.0f - A.acc; A.b += 1.0f - A.acc; }
}

}  // namespace nscomp
