// Sample placement of sample_points_around_mean("uniform") (utils.py:231-241) as lane-level building blocks, shared by the
// stand-alone placement kernels (ns_rays.hip) and the one-kernel renderer (ns_nerf_mlp_ob16.hip), so both place a sample
// with the same fp32 operations (translation units that include this are built with -ffp-contract=off).
#pragma once
#include <hip/hip_runtime.h>

namespace nsplace {

// torch.linspace(start, end, steps)[i] for fp32 (ATen RangeFactories: two-sided evaluation)
__device__ __forceinline__ float linspace_at(float start, float end, int steps, int i) {
  if (steps <= 1) return start;
  const float step = (end - start) / static_cast<float>(steps - 1);
  return (i < steps / 2) ? start + step * static_cast<float>(i)
                         : end - step * static_cast<float>(steps - i - 1);
}
// the same with step = (end - start) / float(steps - 1) precomputed (a correctly rounded fp32 division on either side)
__device__ __forceinline__ float linspace_step(float start, float end, float step, int steps, int i) {
  if (steps <= 1) return start;
  return (i < steps / 2) ? start + step * static_cast<float>(i)
                         : end - step * static_cast<float>(steps - i - 1);
}
__host__ inline float linspace_step_of(float start, float end, int steps) {
  return steps <= 1 ? 0.0f : (end - start) / static_cast<float>(steps - 1);
}

// z[j], j = 0 .. steps, of  clip(sort(cat[m + linspace(-std, std, steps), m]), 2, 6)  (steps = n_samples - 1 >= 1).
// The grid a_i = m + g_i is non-decreasing, so the mean is merged at its rank p = #{i : a_i < m} instead of sorting:
// z[j] = a_j for j < p, m for j == p, a_{j-1} for j > p.  "j < p" <=> a_j < m and "j > p" <=> !(a_{j-1} < m), so no rank
// search is needed.  A NaN mean gives NaN everywhere (the clip would turn it into 2).
__device__ __forceinline__ float uniform_z(float m, float std_, float step, int steps, int j) {
  float v = m;
  if (j < steps) {
    const float b = m + linspace_step(-std_, std_, step, steps, j);
    if (b < m) v = b;
  }
  if (v == m && j >= 1) {          // (v != m: the first branch fired, j < p)
    const float a = m + linspace_step(-std_, std_, step, steps, j - 1);
    if (!(a < m)) v = a;
  }
  v = fminf(fmaxf(v, 2.0f), 6.0f);  // hard-coded clip, utils.py:240
  return (m != m) ? m : v;
}

// z[j] and z[j + 1] of the same ray at once (the one-kernel renderer needs a sample's depth and the distance to the next one):
// the three grid values a_{j-1}, a_j, a_{j+1} are evaluated once, the selections are those of uniform_z -- the same bits.
// Written without branches (both sides of every choice are computed, then selected: the compiler otherwise masks EXEC around each
// two-instruction arm), and with the integer -> float conversions shared: float(k) for k = j - 1, j, j + 1 and float(steps - k - 1)
// are differences of small exact integers, so fj -+ 1 and fs1 - fk are the very values the conversions give.
__device__ __forceinline__ void uniform_z_pair(float m, float std_, float step, int steps, int j, float& z, float& z_next) {
  const int half = steps / 2;
  const float fj = static_cast<float>(j), fs1 = static_cast<float>(steps - 1);
  const bool one = steps <= 1;                       // linspace_step returns `start` for a single step
  auto grid = [&](int k, float fk) {                 // m + linspace_step(-std, std, step, steps, k)
    const float lo = -std_ + step * fk;
    const float hi = std_ - step * (fs1 - fk);
    const bool low = (k < half) | one;
    return m + (low ? lo : hi);
  };
  const float am = grid(j - 1, fj - 1.0f), a0 = grid(j, fj), ap = grid(j + 1, fj + 1.0f);   // (out-of-range k: never selected)
  auto pick = [&](int jj, float below, float here) {                   // uniform_z(jj) from a_{jj-1} = below, a_jj = here
    const bool c1 = (jj < steps) & (here < m);
    float v = c1 ? here : m;
    const bool c2 = (v == m) & (jj >= 1) & !(below < m);
    v = c2 ? below : v;
    v = fminf(fmaxf(v, 2.0f), 6.0f);
    return (m != m) ? m : v;
  };
  z = pick(j, am, a0);
  z_next = pick(j + 1, a0, ap);
}

}  // namespace nsplace
