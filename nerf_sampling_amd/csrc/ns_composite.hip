// Per-ray scan/search kernels: alpha compositing (raw2outputs), inverse-CDF importance
// sampling, per-ray argmax.  HBM-bound: every sample is read once (16 B raw + 4 B z) and the
// per-sample outputs written once; the transmittance product is a wave-level segmented scan.
#include "ns_common.h"
#include "ns_composite_ray.h"

namespace {

// compile-time loops over powers of two: F(J), F(J / 2), ..., F(1)  and  F(K), F(2 K), ..., F(KMAX)
template <int J, class F>
__device__ __forceinline__ void static_for_pow2_down(F&& f) {
  if constexpr (J >= 1) { f(std::integral_constant<int, J>{}); static_for_pow2_down<J / 2>(f); }
}
template <int K, int KMAX, class F>
__device__ __forceinline__ void static_for_pow2_up(F&& f) {
  if constexpr (K <= KMAX) { f(std::integral_constant<int, K>{}); static_for_pow2_up<2 * K, KMAX>(f); }
}

__device__ __forceinline__ float linspace01(int steps, int i) {
  if (steps <= 1) return 0.0f;
  const float step = 1.0f / static_cast<float>(steps - 1);
  return (i < steps / 2) ? step * static_cast<float>(i) : 1.0f - step * static_cast<float>(steps - i - 1);
}

// a8: nerf_utils.py:27-42 + sampling_trainer.py:153-230.
// SW lanes cooperate on one ray (SW = power of two <= 64); rays longer than SW samples are
// walked in chunks of SW with the transmittance carried in a register.
template <int SW>
__global__ void __launch_bounds__(256)
raw2outputs_kernel(const float4* __restrict__ raw, const float* __restrict__ z,
                   const float* __restrict__ rays_d, const float* __restrict__ noise, int64_t R, int N,
                   int white_bkgd, float* __restrict__ rgb_out, float* __restrict__ disp_out,
                   float* __restrict__ acc_out, float* __restrict__ depth_out,
                   float* __restrict__ alphas_out, float* __restrict__ weights_out, int64_t rgb_stride,
                   int64_t disp_stride) {
  constexpr int RAYS_PER_BLOCK = 256 / SW;
  const int sub = threadIdx.x % SW;
  const int64_t ray_stride = (int64_t)gridDim.x * RAYS_PER_BLOCK;
  // all lanes of a wave iterate the same number of times (shuffles need full participation)
  const int64_t iters = (R + ray_stride - 1) / ray_stride;
  if (N <= SW) {
    // one chunk per ray: software-pipelined, the next ray's loads are in flight while this one is composited
    struct In { int64_t r; bool ok, live; float4 q; float zi, dist_raw, nz, norm; };
    auto fetch = [&](int64_t it) -> In {
      In x;
      x.r = it * ray_stride + (int64_t)blockIdx.x * RAYS_PER_BLOCK + threadIdx.x / SW;
      x.live = it < iters && x.r < R;
      x.ok = x.live && sub < N;
      x.q = make_float4(0.f, 0.f, 0.f, 0.f);
      x.zi = x.dist_raw = x.nz = x.norm = 0.f;
      if (x.live) x.norm = nscomp::ray_norm(rays_d[x.r * 3], rays_d[x.r * 3 + 1], rays_d[x.r * 3 + 2]);
      if (x.ok) {
        const int64_t e = x.r * N + sub;
        x.q = raw[e];
        x.zi = z[e];
        x.dist_raw = (sub < N - 1) ? z[e + 1] - x.zi : 1e10f;
        if (noise) x.nz = noise[e];
      }
      return x;
    };
    In cur = fetch(0);
    for (int64_t it = 0; it < iters; ++it) {
      const In nxt = fetch(it + 1);
      nscomp::RayAccum A;
      float alpha, w, disp;
      nscomp::composite_chunk<SW>(A, cur.ok, sub, cur.q, cur.zi, cur.dist_raw, cur.norm, cur.nz, noise != nullptr, alpha, w);
      if (cur.ok) {
        const int64_t e = cur.r * N + sub;
        if (alphas_out) alphas_out[e] = alpha;
        if (weights_out) weights_out[e] = w;
      }
      nscomp::composite_finish<SW>(A, white_bkgd, disp, sub);
      if (cur.live && sub == SW - 1) {                     // the group's last lane holds the totals
        const int64_t r = cur.r;
        if (rgb_out) { float* p = rgb_out + r * rgb_stride; p[0] = A.r; p[1] = A.g; p[2] = A.b; }
        if (acc_out) acc_out[r] = A.acc;
        if (depth_out) depth_out[r] = A.depth;
        if (disp_out) disp_out[r * disp_stride] = disp;
      }
      cur = nxt;
    }
    return;
  }
  for (int64_t it = 0; it < iters; ++it) {
    const int64_t r = it * ray_stride + (int64_t)blockIdx.x * RAYS_PER_BLOCK + threadIdx.x / SW;
    const bool live = r < R;
    float norm = 0.f;
    if (live) norm = nscomp::ray_norm(rays_d[r * 3], rays_d[r * 3 + 1], rays_d[r * 3 + 2]);
    // several chunks per ray (only SW == 64 gets here): every chunk's sums are reduced on their own, the chunk totals added
    // in chunk order (ns_composite_ray.h, add_chunk_totals: the arithmetic the one-kernel renderer reproduces chunk-parallel)
    nscomp::RayAccum tot;                       // wave-uniform: tot.carry = transmittance entering the next chunk
    for (int base = 0; SW == 64 && base < N; base += SW) {
      const int i = base + sub;
      const bool ok = live && i < N;
      float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
      float zi = 0.f, dist_raw = 0.f, nz = 0.f;
      if (ok) {
        const int64_t e = r * N + i;
        q = raw[e];
        zi = z[e];
        dist_raw = (i < N - 1) ? z[e + 1] - zi : 1e10f;
        if (noise) nz = noise[e];
      }
      float alpha, w;
      nscomp::RayAccum A;
      A.carry = tot.carry;
      nscomp::composite_chunk<SW>(A, ok, sub, q, zi, dist_raw, norm, nz, noise != nullptr, alpha, w);
      tot.carry = A.carry;
      nscomp::add_chunk_totals(tot, A);
      if (ok) {
        const int64_t e = r * N + i;
        if (alphas_out) alphas_out[e] = alpha;
        if (weights_out) weights_out[e] = w;
      }
    }
    float disp;
    nscomp::finish_totals(tot, white_bkgd, disp);
    if (live && sub == SW - 1) {
      if (rgb_out) { float* p = rgb_out + r * rgb_stride; p[0] = tot.r; p[1] = tot.g; p[2] = tot.b; }
      if (acc_out) acc_out[r] = tot.acc;
      if (depth_out) depth_out[r] = tot.depth;
      if (disp_out) disp_out[r * disp_stride] = disp;
    }
  }
}

// N == 1 (depth_only sampling, the training operator).  The reference builds `dists` from
// z[..., 1:] - z[..., :-1] and a 1e10 column expanded to that EMPTY shape, so with one sample dists,
// alphas and weights are [R, 0]: acc = depth = 0, disp = 1/1e-10, and rgb_map falls into the
// `weights.shape[-1] == 0` branch: rgb_map = sum(sigmoid(raw rgb), -2) (sampling_trainer.py:174-220).
__global__ void __launch_bounds__(256)
raw2outputs_single_kernel(const float4* __restrict__ raw, int64_t R, float* __restrict__ rgb_out,
                          float* __restrict__ disp_out, float* __restrict__ acc_out,
                          float* __restrict__ depth_out, int64_t rgb_stride, int64_t disp_stride) {
  for (int64_t r = blockIdx.x * (int64_t)256 + threadIdx.x; r < R; r += (int64_t)gridDim.x * 256) {
    const float4 q = raw[r];
    if (rgb_out) {
      float* p = rgb_out + r * rgb_stride;
      p[0] = 1.0f / (1.0f + expf(-q.x));
      p[1] = 1.0f / (1.0f + expf(-q.y));
      p[2] = 1.0f / (1.0f + expf(-q.z));
    }
    if (disp_out) disp_out[r * disp_stride] = 1.0f / 1e-10f;
    if (acc_out) acc_out[r] = 0.0f;
    if (depth_out) depth_out[r] = 0.0f;
  }
}

// ---- inverse-CDF sampling, run_nerf_helpers.py:250-293 -----------------------------------------
// shared body: cdf (length nb) from weights (length nb-1), then one inverse lookup
constexpr int kMaxBins = 512;

__device__ __forceinline__ void build_cdf(const float* w, int nb, float* pdf_s, float* cdf_s, int lane) {
  // pdf = (w + 1e-5) / sum;  cdf = [0, cumsum(pdf)]  (cumsum accumulated in double, as ATen's
  // CPU cumsum does for float input)
  float part = 0.f;
  for (int k = lane; k < nb - 1; k += 64) {
    const float v = w[k] + 1e-5f;
    pdf_s[k] = v;
    part += v;
  }
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) part += __shfl_xor(part, m, 64);
  __syncthreads();
  if (lane == 0) {
    double run = 0.0;
    cdf_s[0] = 0.0f;
    for (int k = 0; k < nb - 1; ++k) {
      run += static_cast<double>(pdf_s[k] / part);
      cdf_s[k + 1] = static_cast<float>(run);
    }
  }
  __syncthreads();
}

__device__ __forceinline__ float invert_cdf(const float* cdf_s, const float* bins_s, int nb, float u) {
  // idx = searchsorted(cdf, u, right=True): first k with cdf[k] > u
  int lo = 0, hi = nb;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (cdf_s[mid] > u) hi = mid; else lo = mid + 1;
  }
  const int below = max(lo - 1, 0);
  const int above = min(lo, nb - 1);
  float denom = cdf_s[above] - cdf_s[below];
  if (denom < 1e-5f) denom = 1.0f;
  const float t = (u - cdf_s[below]) / denom;
  return bins_s[below] + t * (bins_s[above] - bins_s[below]);
}

__global__ void __launch_bounds__(64)
sample_pdf_kernel(const float* __restrict__ bins, const float* __restrict__ weights, int64_t R, int Nb,
                  int Nf, const float* __restrict__ u, float* __restrict__ out) {
  __shared__ float pdf_s[kMaxBins], cdf_s[kMaxBins], bins_s[kMaxBins];
  const int lane = threadIdx.x;
  for (int64_t r = blockIdx.x; r < R; r += gridDim.x) {
    for (int k = lane; k < Nb; k += 64) bins_s[k] = bins[r * Nb + k];
    build_cdf(weights + r * (Nb - 1), Nb, pdf_s, cdf_s, lane);
    for (int s = lane; s < Nf; s += 64) {
      const float uu = u ? u[r * Nf + s] : linspace01(Nf, s);
      out[r * Nf + s] = invert_cdf(cdf_s, bins_s, Nb, uu);
    }
    __syncthreads();
  }
}

__device__ __forceinline__ bool sort_less(float a, float b) {
  return !(a != a) && ((b != b) || a < b);
}

// Trainer.py:672-685: z_mid, sample_pdf(z_mid, w[1:-1]), sort(cat[z, samples])
template <int P>
__global__ void __launch_bounds__(64)
importance_z_kernel(const float* __restrict__ z, const float* __restrict__ w, int64_t R, int Nc, int Nf,
                    const float* __restrict__ u, float* __restrict__ out) {
  __shared__ float pdf_s[kMaxBins], cdf_s[kMaxBins], bins_s[kMaxBins], buf[P];
  const int lane = threadIdx.x;
  const int Nb = Nc - 1;
  for (int64_t r = blockIdx.x; r < R; r += gridDim.x) {
    for (int k = lane; k < Nc; k += 64) buf[k] = z[r * Nc + k];
    __syncthreads();
    for (int k = lane; k < Nb; k += 64) bins_s[k] = 0.5f * (buf[k + 1] + buf[k]);
    build_cdf(w + r * Nc + 1, Nb, pdf_s, cdf_s, lane);
    for (int s = lane; s < Nf; s += 64) {
      const float uu = u ? u[r * Nf + s] : linspace01(Nf, s);
      buf[Nc + s] = invert_cdf(cdf_s, bins_s, Nb, uu);
    }
    for (int i = Nc + Nf + lane; i < P; i += 64) buf[i] = __builtin_nanf("");
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int t = lane; t < P / 2; t += 64) {
          const int lo = ((t / j) * 2 * j) + (t % j);
          const int hi = lo + j;
          const bool up = ((lo & k) == 0);
          const float a = buf[lo], b = buf[hi];
          const bool swap = up ? sort_less(b, a) : sort_less(a, b);
          if (swap) { buf[lo] = b; buf[hi] = a; }
        }
        __syncthreads();
      }
    }
    for (int i = lane; i < Nc + Nf; i += 64) out[r * (Nc + Nf) + i] = buf[i];
    __syncthreads();
  }
}

// Same operator, one WAVE per ray with everything but the two lookup tables in registers (Nc <= 64,
// Nc + Nf <= 64 * NR): the CDF is a wave prefix scan in double (every partial sum of <= 64 floats spanning
// < 2^29 is exact in double, so the scan order cannot change ATen's sequential result), the sort is a bitonic
// network over NR registers x 64 lanes (lane exchanges by ds_bpermute-free __shfl_xor, register exchanges in
// place) -- no block barrier and no serial lane-0 loop.  The LDS version above stays for longer rows.
template <int NR>
__global__ void __launch_bounds__(256)
importance_z_wave_kernel(const float* __restrict__ z, const float* __restrict__ w, int64_t R, int Nc, int Nf,
                         const float* __restrict__ u, float* __restrict__ out) {
  __shared__ float cdf_all[4][64], bins_all[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* cdf_s = cdf_all[wv];
  float* bins_s = bins_all[wv];
  const int Nb = Nc - 1, tot = Nc + Nf;
  const int64_t wave = (int64_t)blockIdx.x * 4 + wv, nwaves = (int64_t)gridDim.x * 4;
  for (int64_t r = wave; r < R; r += nwaves) {
    const float zk = lane < Nc ? z[r * Nc + lane] : 0.0f;
    const float zn = __shfl_down(zk, 1, 64);
    // pdf_k = w[k + 1] + 1e-5 for k < Nb - 1; sum by the same xor butterfly as build_cdf()
    const float v = lane < Nb - 1 ? w[r * Nc + 1 + lane] + 1e-5f : 0.0f;
    float part = v;
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) part += __shfl_xor(part, m, 64);
    double run = lane < Nb - 1 ? static_cast<double>(v / part) : 0.0;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
      const double up = __shfl_up(run, m, 64);
      if (lane >= m) run += up;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the previous ray's lookups are done
    if (lane < Nb) bins_s[lane] = 0.5f * (zn + zk);
    if (lane == 0) cdf_s[0] = 0.0f;
    if (lane < Nb - 1) cdf_s[lane + 1] = static_cast<float>(run);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // The 64 NR slots as a BITONIC sequence: the coarse depths ascending in slots [0, Nc), NaN (sorts last) in the middle, the new
    // samples in REVERSE order at the end (sample s in slot 64 NR - 1 - s).  With deterministic draws (u = linspace) the inverse
    // CDF is non-decreasing in u, so the sequence rises, stays at its maximum and falls: ONE merge stage of the network (log2(64 NR)
    // compare-exchange steps) sorts it; the full network (36 steps at NR = 4) runs only when the wave finds either run out of order
    // -- random draws, or a rounding inversion at a bin boundary -- so the result is the sorted row either way.
    constexpr int P = 64 * NR;
    float e[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      const int i = q * 64 + lane;
      if (i < Nc) e[q] = zk;                      // only q == 0: Nc <= 64
      else if (i >= P - Nf) {
        const int sidx = P - 1 - i;
        const float uu = u ? u[r * Nf + sidx] : linspace01(Nf, sidx);
        e[q] = invert_cdf(cdf_s, bins_s, Nb, uu);
      } else e[q] = __builtin_nanf("");
    }
    bool in_order = true;                         // slot i against slot i - 1, inside the two runs
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      const int i = q * 64 + lane;
      float prev = __shfl_up(e[q], 1, 64);
      if (q > 0) { const float carry = __shfl(e[q - 1], 63, 64); if (lane == 0) prev = carry; }
      if (i >= 1 && i < Nc) in_order = in_order && (e[q] >= prev);
      if (i > P - Nf) in_order = in_order && (e[q] <= prev);
    }
    auto stage = [&](auto k_) {                   // one stage of the bitonic network: blocks of k slots, alternating direction
      constexpr int k = decltype(k_)::value;
      static_for_pow2_down<k / 2>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        if constexpr (j >= 64) {
#pragma unroll
          for (int q = 0; q < NR; ++q) {
            const int pq = q ^ (j >> 6);
            if (pq > q) {
              const bool up = (((q * 64 + lane) & k) == 0);
              const float a = e[q], b = e[pq];
              const bool swap = up ? sort_less(b, a) : sort_less(a, b);
              e[q] = swap ? b : a;
              e[pq] = swap ? a : b;
            }
          }
        } else {
#pragma unroll
          for (int q = 0; q < NR; ++q) {
            const int idx = q * 64 + lane;
            const float mine = e[q], other = __shfl_xor(mine, j, 64);
            const bool up = ((idx & k) == 0), lower = ((lane & j) == 0);
            // the lower index of the pair keeps the smaller value when sorting up
            const bool take_min = (up == lower);
            const bool other_less = sort_less(other, mine), mine_less = sort_less(mine, other);
            e[q] = take_min ? (other_less ? other : mine) : (mine_less ? other : mine);
          }
        }
      });
    };
    if (__builtin_amdgcn_ballot_w64(in_order) == __builtin_amdgcn_ballot_w64(true)) {
      stage(std::integral_constant<int, P>{});
    } else {
      static_for_pow2_up<2, P>([&](auto k_) { stage(k_); });
    }
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      const int i = q * 64 + lane;
      if (i < tot) out[r * tot + i] = e[q];
    }
  }
}

// nerf_utils.py:813-819: argmax over samples (first index on ties, NaN counts as largest)
__device__ __forceinline__ bool beats(float v, int i, float bv, int bi) {
  if (i == 0x7fffffff) return false;
  if (bi == 0x7fffffff) return true;
  const bool vn = v != v, bn = bv != bv;
  if (vn != bn) return vn;
  if (vn) return i < bi;
  if (v != bv) return v > bv;
  return i < bi;
}

__global__ void __launch_bounds__(256)
argmax_gather_kernel(const float* __restrict__ w, const float* __restrict__ z, const float4* __restrict__ raw,
                     int64_t R, int N, float* __restrict__ max_z, float* __restrict__ max_w,
                     float* __restrict__ max_rgb) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * 256) >> 6;
  for (int64_t r = wave; r < R; r += nwaves) {
    float best = 0.f;
    int bi = 0x7fffffff;
    for (int i = lane; i < N; i += 64) {
      const float v = w[r * N + i];
      if (beats(v, i, best, bi)) { best = v; bi = i; }
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
      const float ov = __shfl_xor(best, m, 64);
      const int oi = __shfl_xor(bi, m, 64);
      if (beats(ov, oi, best, bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) {
      if (max_w) max_w[r] = best;
      if (max_z) max_z[r] = z[r * N + bi];
      if (max_rgb) {
        const float4 q = raw[r * N + bi];
        max_rgb[r * 3] = 1.0f / (1.0f + expf(-q.x));
        max_rgb[r * 3 + 1] = 1.0f / (1.0f + expf(-q.y));
        max_rgb[r * 3 + 2] = 1.0f / (1.0f + expf(-q.z));
      }
    }
  }
}

template <int SW>
void launch_r2o(const float* raw, const float* z, const float* rays_d, const float* noise, int64_t R, int N,
                int white, float* rgb, float* disp, float* acc, float* depth, float* alphas, float* weights,
                int64_t rgb_stride, int64_t disp_stride, hipStream_t s) {
  const int rays_per_block = 256 / SW;
  int64_t grid = ns::cdiv(R, rays_per_block);
  if (grid > 256 * 16) grid = 256 * 16;
  raw2outputs_kernel<SW><<<static_cast<int>(grid), 256, 0, s>>>(reinterpret_cast<const float4*>(raw), z, rays_d,
                                                               noise, R, N, white, rgb, disp, acc, depth,
                                                               alphas, weights, rgb_stride, disp_stride);
}

}  // namespace

extern "C" {

int ns_raw2outputs(const float* raw_dev, const float* z_dev, const float* rays_d_dev,
                   const float* noise_dev, int64_t R, int N, int white_bkgd, float* rgb_dev,
                   float* disp_dev, float* acc_dev, float* depth_dev, float* alphas_dev,
                   float* weights_dev, void* stream) {
  return ns_raw2outputs_strided(raw_dev, z_dev, rays_d_dev, noise_dev, R, N, white_bkgd, rgb_dev, 3, disp_dev, 1,
                                acc_dev, depth_dev, alphas_dev, weights_dev, stream);
}

int ns_raw2outputs_strided(const float* raw_dev, const float* z_dev, const float* rays_d_dev,
                           const float* noise_dev, int64_t R, int N, int white_bkgd, float* rgb_dev,
                           int64_t rgb_stride, float* disp_dev, int64_t disp_stride, float* acc_dev,
                           float* depth_dev, float* alphas_dev, float* weights_dev, void* stream) {
  NS_REQUIRE(R >= 0 && N >= 1, "bad shape (N == 0 is handled by the caller)");
  NS_REQUIRE(rgb_stride >= 3 && disp_stride >= 1, "output strides (in floats) must be >= 3 for rgb and >= 1 for disp");
  if (R == 0) return NS_OK;
  NS_REQUIRE(raw_dev && z_dev && rays_d_dev, "null input");
  NS_REQUIRE((reinterpret_cast<uintptr_t>(raw_dev) & 15) == 0, "raw must be 16-byte aligned");
  hipStream_t s = ns::as_stream(stream);
#define NS_R2O(SW) launch_r2o<SW>(raw_dev, z_dev, rays_d_dev, noise_dev, R, N, white_bkgd, rgb_dev, disp_dev, \
                                  acc_dev, depth_dev, alphas_dev, weights_dev, rgb_stride, disp_stride, s)
  if (N == 1) {
    // alphas / weights are [R, 0] in the reference for a single sample: nothing to write there
    raw2outputs_single_kernel<<<ns::ew_grid(R, 256), 256, 0, s>>>(reinterpret_cast<const float4*>(raw_dev), R,
                                                                  rgb_dev, disp_dev, acc_dev, depth_dev, rgb_stride,
                                                                  disp_stride);
  } else if (N <= 2) NS_R2O(2);
  else if (N <= 4) NS_R2O(4);
  else if (N <= 8) NS_R2O(8);
  else if (N <= 16) NS_R2O(16);
  else if (N <= 32) NS_R2O(32);
  else NS_R2O(64);
#undef NS_R2O
  NS_LAUNCH_CHECK();
  return NS_OK;
}

int ns_sample_pdf(const float* bins_dev, const float* weights_dev, int64_t R, int Nb, int Nf,
                  const float* u_dev, float* samples_dev, void* stream) {
  NS_REQUIRE(R >= 0 && Nb >= 2 && Nf >= 0, "bad shape");
  NS_REQUIRE(Nb <= kMaxBins, "more than 512 bins is not supported");
  if (R == 0 || Nf == 0) return NS_OK;
  NS_REQUIRE(bins_dev && weights_dev && samples_dev, "null pointer");
  const int grid = static_cast<int>(R < 256 * 32 ? R : 256 * 32);
  sample_pdf_kernel<<<grid, 64, 0, ns::as_stream(stream)>>>(bins_dev, weights_dev, R, Nb, Nf, u_dev, samples_dev);
  NS_LAUNCH_CHECK();
  return NS_OK;
}

int ns_importance_z(const float* z_dev, const float* weights_dev, int64_t R, int Nc, int Nf,
                    const float* u_dev, float* z_out_dev, void* stream) {
  NS_REQUIRE(R >= 0 && Nc >= 3 && Nf >= 0, "bad shape (needs at least 3 coarse samples)");
  NS_REQUIRE(Nc - 1 <= kMaxBins && Nc + Nf <= 2048, "sample counts too large");
  if (R == 0) return NS_OK;
  NS_REQUIRE(z_dev && weights_dev && z_out_dev, "null pointer");
  const int grid = static_cast<int>(R < 256 * 32 ? R : 256 * 32);
  hipStream_t s = ns::as_stream(stream);
  const int tot = Nc + Nf;
  if (Nc <= 64 && tot <= 256) {   // one wave per ray, registers only
    int64_t g = ns::cdiv(R, 4);
    if (g > 256 * 32) g = 256 * 32;
    const int gi = static_cast<int>(g);
    if (tot <= 64) importance_z_wave_kernel<1><<<gi, 256, 0, s>>>(z_dev, weights_dev, R, Nc, Nf, u_dev, z_out_dev);
    else if (tot <= 128) importance_z_wave_kernel<2><<<gi, 256, 0, s>>>(z_dev, weights_dev, R, Nc, Nf, u_dev, z_out_dev);
    else importance_z_wave_kernel<4><<<gi, 256, 0, s>>>(z_dev, weights_dev, R, Nc, Nf, u_dev, z_out_dev);
    NS_LAUNCH_CHECK();
    return NS_OK;
  }
  if (tot <= 64) importance_z_kernel<64><<<grid, 64, 0, s>>>(z_dev, weights_dev, R, Nc, Nf, u_dev, z_out_dev);
  else if (tot <= 128) importance_z_kernel<128><<<grid, 64, 0, s>>>(z_dev, weights_dev, R, Nc, Nf, u_dev, z_out_dev);
  else if (tot <= 256) importance_z_kernel<256><<<grid, 64, 0, s>>>(z_dev, weights_dev, R, Nc, Nf, u_dev, z_out_dev);
  else if (tot <= 512) importance_z_kernel<512><<<grid, 64, 0, s>>>(z_dev, weights_dev, R, Nc, Nf, u_dev, z_out_dev);
  else if (tot <= 1024) importance_z_kernel<1024><<<grid, 64, 0, s>>>(z_dev, weights_dev, R, Nc, Nf, u_dev, z_out_dev);
  else importance_z_kernel<2048><<<grid, 64, 0, s>>>(z_dev, weights_dev, R, Nc, Nf, u_dev, z_out_dev);
  NS_LAUNCH_CHECK();
  return NS_OK;
}

int ns_argmax_gather(const float* weights_dev, const float* z_dev, const float* raw_dev, int64_t R,
                     int N, float* max_z_dev, float* max_w_dev, float* max_rgb_dev, void* stream) {
  NS_REQUIRE(R >= 0 && N >= 1, "bad shape");
  if (R == 0) return NS_OK;
  NS_REQUIRE(weights_dev, "null weights");
  NS_REQUIRE(!max_z_dev || z_dev, "max_z requested without z");
  NS_REQUIRE(!max_rgb_dev || raw_dev, "max_rgb requested without raw");
  int64_t grid = ns::cdiv(R, 4);
  if (grid > 256 * 16) grid = 256 * 16;
  argmax_gather_kernel<<<static_cast<int>(grid), 256, 0, ns::as_stream(stream)>>>(
      weights_dev, z_dev, reinterpret_cast<const float4*>(raw_dev), R, N, max_z_dev, max_w_dev, max_rgb_dev);
  NS_LAUNCH_CHECK();
  return NS_OK;
}

}  // extern "C"

#include "ns_weights.h"
namespace {
__global__ void __launch_bounds__(256)
patch_sigma_last_kernel(float4* __restrict__ raw, const float4* __restrict__ raw_last, int64_t R, int N) {
  for (int64_t r = blockIdx.x * (int64_t)256 + threadIdx.x; r < R; r += (int64_t)gridDim.x * 256)
    raw[r * N + (N - 1)].w = raw_last[r].w;
}
}  // namespace
// sigma of every ray's last sample <- the guard pass's (ns_render_args::nerf_guard; the chain renderer)
// ---- the selective guard (ns_render_args::guard_threshold): the one-kernel renderer left a record per ray whose 16-bit sigma of
// the last sample lies within the threshold of zero (Nerf16Args::fix_rec); the count lives on the device, the launches cover the
// capacity and return at once past it.
__device__ __forceinline__ int64_t fix_ray_of(const float* rec) {
  return static_cast<int64_t>(static_cast<uint64_t>(__builtin_bit_cast(uint32_t, rec[11])) |
                              (static_cast<uint64_t>(__builtin_bit_cast(uint32_t, rec[12])) << 32));
}
// inputs of the flagged rays' last samples, compacted for the fp32-grade network (N = 1 "rays")
__global__ void __launch_bounds__(256)
fix_gather_kernel(const float* __restrict__ rec, const uint32_t* __restrict__ count, int64_t cap, const float* __restrict__ o,
                  const float* __restrict__ d, const float* __restrict__ view, float* __restrict__ o_c, float* __restrict__ d_c,
                  float* __restrict__ view_c, float* __restrict__ z_c) {
  int64_t n = static_cast<int64_t>(*count);
  if (n > cap) n = cap;
  for (int64_t s = blockIdx.x * static_cast<int64_t>(256) + threadIdx.x; s < n; s += static_cast<int64_t>(gridDim.x) * 256) {
    const float* q = rec + s * 16;
    const int64_t r = fix_ray_of(q);
#pragma unroll
    for (int c = 0; c < 3; ++c) { o_c[s * 3 + c] = o[r * 3 + c]; d_c[s * 3 + c] = d[r * 3 + c]; view_c[s * 3 + c] = view[r * 3 + c]; }
    z_c[s] = q[9];
  }
}
// the flagged pixels again: tree sums + the last sample's share with the re-evaluated sigma (nscomp::recomposite_last: the very
// additions composite_finish makes, so a ray whose sigma keeps its sign comes out bit-identical to the every-ray guard)
__global__ void __launch_bounds__(256)
fix_last_sample_kernel(const float* __restrict__ rec, const uint32_t* __restrict__ count, int64_t cap, const float4* __restrict__ raw_c,
                       int N, int white_bkgd, float* __restrict__ rgb, int64_t rgb_stride, float* __restrict__ disp_out,
                       int64_t disp_stride, float* __restrict__ weights) {
  int64_t n = static_cast<int64_t>(*count);
  if (n > cap) n = cap;
  for (int64_t s = blockIdx.x * static_cast<int64_t>(256) + threadIdx.x; s < n; s += static_cast<int64_t>(gridDim.x) * 256) {
    const float* q = rec + s * 16;
    const int64_t r = fix_ray_of(q);
    nscomp::RayAccum A;
    A.r = q[0]; A.g = q[1]; A.b = q[2]; A.depth = q[3]; A.acc = q[4];
    float disp, w;
    nscomp::recomposite_last(A, q[5], q[6], q[7], q[8], raw_c[s].w, q[9], q[10], white_bkgd, disp, w);
    float* p = rgb + r * rgb_stride;
    p[0] = A.r; p[1] = A.g; p[2] = A.b;
    disp_out[r * disp_stride] = disp;
    if (weights) weights[r * N + (N - 1)] = w;
  }
}

int ns_fix_gather(const float* rec_dev, const uint32_t* count_dev, int64_t cap, const float* o_dev, const float* d_dev,
                  const float* view_dev, float* o_c, float* d_c, float* view_c, float* z_c, void* stream) {
  NS_REQUIRE(rec_dev && count_dev && cap >= 0 && o_dev && d_dev && view_dev && o_c && d_c && view_c && z_c, "bad arguments");
  if (cap == 0) return NS_OK;
  fix_gather_kernel<<<ns::ew_grid(cap, 256), 256, 0, ns::as_stream(stream)>>>(rec_dev, count_dev, cap, o_dev, d_dev, view_dev, o_c,
                                                                             d_c, view_c, z_c);
  NS_LAUNCH_CHECK();
  return NS_OK;
}

int ns_fix_last_sample(const float* rec_dev, const uint32_t* count_dev, int64_t cap, const float* raw_c, int N, int white_bkgd,
                       float* rgb_dev, int64_t rgb_stride, float* disp_dev, int64_t disp_stride, float* weights_dev, void* stream) {
  NS_REQUIRE(rec_dev && count_dev && cap >= 0 && raw_c && N >= 2 && rgb_dev && disp_dev, "bad arguments");
  if (cap == 0) return NS_OK;
  fix_last_sample_kernel<<<ns::ew_grid(cap, 256), 256, 0, ns::as_stream(stream)>>>(
      rec_dev, count_dev, cap, reinterpret_cast<const float4*>(raw_c), N, white_bkgd, rgb_dev, rgb_stride, disp_dev, disp_stride,
      weights_dev);
  NS_LAUNCH_CHECK();
  return NS_OK;
}

int ns_patch_sigma_last(float* raw_dev, const float* raw_last_dev, int64_t R, int N, void* stream) {
  NS_REQUIRE(R >= 0 && N >= 1 && raw_dev && raw_last_dev, "bad arguments");
  if (R == 0) return NS_OK;
  patch_sigma_last_kernel<<<ns::ew_grid(R, 256), 256, 0, ns::as_stream(stream)>>>(
      reinterpret_cast<float4*>(raw_dev), reinterpret_cast<const float4*>(raw_last_dev), R, N);
  NS_LAUNCH_CHECK();
  return NS_OK;
}
