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

// ‖d‖: the sum as torch.norm forms it on the CPU (fma chain), the root on the transcendental unit (v_sqrt_f32, 1 ulp: the
// correctly rounded sqrtf is that plus two refinement steps and a denormal rescue, 15 instructions per sample for a factor that
// only scales the exponent of alpha)
__device__ __forceinline__ float ray_norm(float dx, float dy, float dz) {
  return __builtin_amdgcn_sqrtf(__builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)));
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

// The same scan for products over segments of whole rows (SW >= 16) with the shift INSIDE the multiply (v_mul_f32_dpp: a lane
// whose source lane does not exist is left as it is -- the multiplication by the identity the generic form spends a v_mov and a
// v_mov_dpp on), one instruction per step.  Each step reads the register the step before wrote: two wait states (s_nop 1) between
// a VALU write and a DPP read of the same register; s_nop 4 in front covers a VALU write of EXEC by the code before.
#define NS_DPP_SHR(n) "row_shr:" #n " row_mask:0xf bank_mask:0xf"
#define NS_DPP_BC15 "row_bcast:15 row_mask:0xa bank_mask:0xf"
#define NS_DPP_BC31 "row_bcast:31 row_mask:0xc bank_mask:0xf"
#define NS_MUL_STEP(ctl) "v_mul_f32_dpp %0, %0, %0 " ctl "\n\ts_nop 1\n\t"
template <int SW>
__device__ __forceinline__ float seg_scan_mul_rows(float x) {
  static_assert(SW == 16 || SW == 32 || SW == 64, "whole 16-lane rows");
  // (ONE asm statement per width: between two statements the compiler may copy the value to another register right in front of
  // the DPP read, a hazard it does not see inside inline asm)
  if constexpr (SW == 16)
    asm volatile("s_nop 4\n\t" NS_MUL_STEP(NS_DPP_SHR(1)) NS_MUL_STEP(NS_DPP_SHR(2)) NS_MUL_STEP(NS_DPP_SHR(4)) NS_MUL_STEP(NS_DPP_SHR(8)) : "+v"(x));
  else if constexpr (SW == 32)
    asm volatile("s_nop 4\n\t" NS_MUL_STEP(NS_DPP_SHR(1)) NS_MUL_STEP(NS_DPP_SHR(2)) NS_MUL_STEP(NS_DPP_SHR(4)) NS_MUL_STEP(NS_DPP_SHR(8))
                 NS_MUL_STEP(NS_DPP_BC15) : "+v"(x));
  else
    asm volatile("s_nop 4\n\t" NS_MUL_STEP(NS_DPP_SHR(1)) NS_MUL_STEP(NS_DPP_SHR(2)) NS_MUL_STEP(NS_DPP_SHR(4)) NS_MUL_STEP(NS_DPP_SHR(8))
                 NS_MUL_STEP(NS_DPP_BC15) NS_MUL_STEP(NS_DPP_BC31) : "+v"(x));
  return x;
}

// Five sums at once, reduced INTO THE LAST LANE of every SW-lane segment (the other lanes end up with partial sums that may
// reach into the segment before: only lane SW - 1 is read).  The operations that reach the last lane are those of seg_scan
// above in the same order -- the same bits -- as one v_add_f32_dpp per step and sum: the five chains are interleaved, so a
// register is read four instructions after it was written and no wait states are needed between the steps.
#define NS_SUM5_STEP(ctl) \
  "v_add_f32_dpp %0, %0, %0 " ctl "\n\tv_add_f32_dpp %1, %1, %1 " ctl "\n\tv_add_f32_dpp %2, %2, %2 " ctl "\n\t" \
  "v_add_f32_dpp %3, %3, %3 " ctl "\n\tv_add_f32_dpp %4, %4, %4 " ctl "\n\t"
#define NS_SUM5_OPS : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e)
template <int SW>
__device__ __forceinline__ void reduce5(float& a, float& b, float& c, float& d, float& e) {
  static_assert(SW == 2 || SW == 4 || SW == 8 || SW == 16 || SW == 32 || SW == 64, "a power of two up to the wave");
  if constexpr (SW == 2) asm volatile("s_nop 4\n\t" NS_SUM5_STEP(NS_DPP_SHR(1)) NS_SUM5_OPS);
  else if constexpr (SW == 4) asm volatile("s_nop 4\n\t" NS_SUM5_STEP(NS_DPP_SHR(1)) NS_SUM5_STEP(NS_DPP_SHR(2)) NS_SUM5_OPS);
  else if constexpr (SW == 8) asm volatile("s_nop 4\n\t" NS_SUM5_STEP(NS_DPP_SHR(1)) NS_SUM5_STEP(NS_DPP_SHR(2)) NS_SUM5_STEP(NS_DPP_SHR(4)) NS_SUM5_OPS);
  else if constexpr (SW == 16)
    asm volatile("s_nop 4\n\t" NS_SUM5_STEP(NS_DPP_SHR(1)) NS_SUM5_STEP(NS_DPP_SHR(2)) NS_SUM5_STEP(NS_DPP_SHR(4)) NS_SUM5_STEP(NS_DPP_SHR(8)) NS_SUM5_OPS);
  else if constexpr (SW == 32)
    asm volatile("s_nop 4\n\t" NS_SUM5_STEP(NS_DPP_SHR(1)) NS_SUM5_STEP(NS_DPP_SHR(2)) NS_SUM5_STEP(NS_DPP_SHR(4)) NS_SUM5_STEP(NS_DPP_SHR(8))
                 NS_SUM5_STEP(NS_DPP_BC15) NS_SUM5_OPS);
  else
    asm volatile("s_nop 4\n\t" NS_SUM5_STEP(NS_DPP_SHR(1)) NS_SUM5_STEP(NS_DPP_SHR(2)) NS_SUM5_STEP(NS_DPP_SHR(4)) NS_SUM5_STEP(NS_DPP_SHR(8))
                 NS_SUM5_STEP(NS_DPP_BC15) NS_SUM5_STEP(NS_DPP_BC31) NS_SUM5_OPS);
}
#undef NS_SUM5_STEP
#undef NS_SUM5_OPS
#undef NS_MUL_STEP

// exp and 1 / x on the transcendental unit (v_exp_f32, v_rcp_f32: 1 ulp each) for compositing.  exp(x) = 2^(x log2 e) with the
// product rounded once: relative error <= (1 + |x| log2 e) 2^-23.  What compositing uses are 1 - exp(-s) and 1 / (1 + exp(-x)),
// whose ABSOLUTE error that leaves at the size of the fp32 result's own rounding for every argument (s e^-s <= 0.37; |x| sigma(x)
// (1 - sigma(x)) <= 0.23; measured against float64: 5.9e-8 and 9.3e-8, tests/test_gpu_kernels.py) -- at 2 instructions instead of
// 14 (range-reduced expf) and 1 instead of 10 (IEEE division).
// +-inf and NaN arguments come out as IEEE says (no inf - inf inside).
__device__ __forceinline__ float exp_tu(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float rcp_tu(float x) { return __builtin_amdgcn_rcpf(x); }

// One chunk of SW consecutive samples of a ray, one sample per lane (lane `sub` of the SW-lane group), in two stages so that a
// ray longer than a chunk can have its chunks evaluated side by side (the one-kernel renderer) or one after the other
// (raw2outputs_kernel) with the same arithmetic:
//   chunk_local: everything that does not depend on the chunks before -- alpha, the colours, the chunk's OWN inclusive
//                transmittance product p (lane SW - 1: the chunk's total) and its exclusive form excl;
//   then, given the transmittance `carry` entering the chunk:  T = carry * excl,  w = alpha * T.
//   ok: this lane holds a real sample;  q = raw (rgb, sigma);  dist_raw = z[i+1] - z[i] or 1e10 for the last sample (before the
//   ‖d‖ scaling);  noise = raw_noise_std * randn or 0.
struct ChunkLocal {
  float alpha, cr, cg, cb, p, excl;
};
// raw2alpha (nerf_utils.py:27-42) and the colour sigmoid of one sample (also evaluated by the selective guard's fix-up kernel:
// one definition, one rounding)
__device__ __forceinline__ float sample_alpha(float sigma, float dist) {
  const float alpha = 1.0f - exp_tu(-fmaxf(sigma, 0.0f) * dist);
  return (sigma != sigma) ? sigma : alpha;      // relu(NaN) is NaN in torch
}
__device__ __forceinline__ float sample_colour(float x) { return rcp_tu(1.0f + exp_tu(-x)); }
template <int SW>
__device__ __forceinline__ ChunkLocal chunk_local(bool ok, int sub, float4 q, float dist_raw, float norm, float noise, bool has_noise) {
  ChunkLocal L;
  L.alpha = 0.f; L.cr = 0.f; L.cg = 0.f; L.cb = 0.f;
  if (ok) {
    const float dist = dist_raw * norm;
    float sigma = q.w;
    if (has_noise) sigma += noise;
    L.alpha = sample_alpha(sigma, dist);
    L.cr = sample_colour(q.x);
    L.cg = sample_colour(q.y);
    L.cb = sample_colour(q.z);
  }
  // inclusive product scan of (1 - alpha + 1e-10) over the SW lanes of this chunk
  const float keep = ok ? (1.0f - L.alpha) + 1e-10f : 1.0f;
  if constexpr (SW >= 16) L.p = seg_scan_mul_rows<SW>(keep);
  else L.p = seg_scan<SW>(keep, 1.0f, sub, [](float a, float b) { return a * b; });
  L.excl = dpp<kWaveShr1>(1.0f, L.p);          // the previous lane's inclusive product (lane 0: 1)
  if (sub == 0) L.excl = 1.0f;
  return L;
}
// the chunk's total transmittance factor (SW == 64: lane 63 of the wave), wave-uniform
__device__ __forceinline__ float chunk_product(const ChunkLocal& L) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, L.p), 63));
}

// Both stages for a chunk whose entering transmittance is A.carry; accumulates the lane's share of the ray sums into A and
// returns alpha and weight of the lane's sample.
template <int SW>
__device__ __forceinline__ void composite_chunk(RayAccum& A, bool ok, int sub, float4 q, float zi, float dist_raw,
                                                float norm, float noise, bool has_noise, float& alpha_out,
                                                float& w_out, float* T_out = nullptr) {
  const ChunkLocal L = chunk_local<SW>(ok, sub, q, dist_raw, norm, noise, has_noise);
  const float T = A.carry * L.excl;
  const float w = L.alpha * T;
  if (T_out) *T_out = T;
  if constexpr (SW == 64) A.carry = A.carry * chunk_product(L);     // (shorter segments are whole rays: nothing is carried)
  if (ok) {
    A.r += w * L.cr; A.g += w * L.cg; A.b += w * L.cb;
    A.depth += w * zi;
    A.acc += w;
  }
  alpha_out = L.alpha;
  w_out = w;
}

// Reduce the lanes' shares: the LAST lane of each SW-lane group (sub == SW - 1) ends up with the group's totals in A
// (inclusive add-scans on DPP: five independent chains of log2 SW instructions, no LDS traffic).
template <int SW>
__device__ __forceinline__ void reduce_sums(RayAccum& A, int sub) {
  (void)sub;
  reduce5<SW>(A.r, A.g, A.b, A.depth, A.acc);
}
// disparity and the white background from a ray's totals (sampling_trainer.py:209-220)
__device__ __forceinline__ void finish_totals(RayAccum& A, int white_bkgd, float& disp) {
  const float q = A.depth * rcp_tu(A.acc + 1e-10f);
  disp = rcp_tu((q != q) ? q : fmaxf(1e-10f, q));    // torch.max(1e-10, q) propagates NaN
  if (white_bkgd) { A.r += 1.0f - A.acc; A.g += 1.0f - A.acc; A.b += 1.0f - A.acc; }
}
// a ray of ONE chunk: the last lane of each SW-lane group ends up with the ray's totals in A and its disparity in `disp`.
// The share of lane SW - 1 -- the ray's LAST sample when N == SW, the one composited with dist = 1e10 -- joins the sums last:
// totals = tree(lanes 0 .. SW - 2) + share(SW - 1).  The selective guard (ns_render_args::guard_threshold) re-evaluates that one
// sample after the kernel and repeats exactly this addition from the tree sums (`tree`, if asked for) it was handed.
template <int SW>
__device__ __forceinline__ void composite_finish(RayAccum& A, int white_bkgd, float& disp, int sub, RayAccum* tree = nullptr) {
  const RayAccum own = A;
  if (sub == SW - 1) { A.r = 0.0f; A.g = 0.0f; A.b = 0.0f; A.depth = 0.0f; A.acc = 0.0f; }
  reduce_sums<SW>(A, sub);
  if (tree) *tree = A;
  A.r = A.r + own.r; A.g = A.g + own.g; A.b = A.b + own.b; A.depth = A.depth + own.depth; A.acc = A.acc + own.acc;
  finish_totals(A, white_bkgd, disp);
}
// ... and the fix-up's side of it: the ray's totals from the tree sums, the transmittance T entering the last sample, that
// sample's raw colour, depth and distance, and its re-evaluated sigma (the operations of composite_chunk + composite_finish)
__device__ __forceinline__ void recomposite_last(RayAccum& A /* in: tree sums, out: totals */, float T, float qx, float qy, float qz,
                                                 float sigma, float z, float dist, int white_bkgd, float& disp, float& w_out) {
  const float w = sample_alpha(sigma, dist) * T;
  RayAccum own;
  own.r = 0.0f + w * sample_colour(qx); own.g = 0.0f + w * sample_colour(qy); own.b = 0.0f + w * sample_colour(qz);
  own.depth = 0.0f + w * z; own.acc = 0.0f + w;
  A.r = A.r + own.r; A.g = A.g + own.g; A.b = A.b + own.b; A.depth = A.depth + own.depth; A.acc = A.acc + own.acc;
  finish_totals(A, white_bkgd, disp);
  w_out = w;
}

// A ray of SEVERAL 64-sample chunks (N > 64): every chunk's sums are reduced on their own and the chunk totals are added in
// chunk order -- so the chunks of a ray can be evaluated by different waves (the one-kernel renderer) or one after the other
// (raw2outputs_kernel) and give the same bits.  `tot` (wave-uniform) += the totals of the chunk whose lane shares are in A.
__device__ __forceinline__ void add_chunk_totals(RayAccum& tot, RayAccum A) {
  reduce_sums<64>(A, static_cast<int>(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))));
  auto last = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63)); };
  tot.r = tot.r + last(A.r); tot.g = tot.g + last(A.g); tot.b = tot.b + last(A.b);
  tot.depth = tot.depth + last(A.depth); tot.acc = tot.acc + last(A.acc);
}

}  // namespace nscomp
