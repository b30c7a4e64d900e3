// Ray-side element-wise kernels: camera rays, ray-sphere intersection, positional encoding,
// sample placement, coarse depths.  All HBM-bound, fp32, one thread per output element or per
// ray, grid-stride.  Built with -ffp-contract=off so that mul/add stay separate roundings like
// the reference's eager PyTorch arithmetic.
#include "ns_common.h"
#include "ns_place.h"

namespace {

constexpr int kBlock = 256;

using nsplace::linspace_at;

struct Cam {
  float fx, fy, cx, cy;
  float r[9];  // c2w[:3,:3] row-major
  float t[3];  // c2w[:3,3]
};

// a1: run_nerf_helpers.py:187-202 + nerf_utils.py:156-188
__global__ void __launch_bounds__(kBlock)
get_rays_kernel(Cam cam, int W, int row0, int64_t R, float near_, float far_,
                float* __restrict__ rays_o, float* __restrict__ rays_d,
                float* __restrict__ viewdirs, float* __restrict__ ray_batch) {
  for (int64_t idx = blockIdx.x * (int64_t)kBlock + threadIdx.x; idx < R;
       idx += (int64_t)gridDim.x * kBlock) {
    const int j = row0 + static_cast<int>(idx / W);
    const int i = static_cast<int>(idx % W);
    const float dx = (static_cast<float>(i) - cam.cx) / cam.fx;
    const float dy = -((static_cast<float>(j) - cam.cy) / cam.fy);
    const float dz = -1.0f;
    float d[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
      d[c] = (dx * cam.r[3 * c + 0] + dy * cam.r[3 * c + 1]) + dz * cam.r[3 * c + 2];
    // torch.norm on the CPU accumulates squares as an fma chain (verified bit-for-bit on the golden rays)
    const float nrm = sqrtf(__builtin_fmaf(d[2], d[2], __builtin_fmaf(d[1], d[1], d[0] * d[0])));
    if (rays_o) {
      rays_o[idx * 3 + 0] = cam.t[0]; rays_o[idx * 3 + 1] = cam.t[1]; rays_o[idx * 3 + 2] = cam.t[2];
    }
    if (rays_d) {
      rays_d[idx * 3 + 0] = d[0]; rays_d[idx * 3 + 1] = d[1]; rays_d[idx * 3 + 2] = d[2];
    }
    if (viewdirs) {
      viewdirs[idx * 3 + 0] = d[0] / nrm; viewdirs[idx * 3 + 1] = d[1] / nrm; viewdirs[idx * 3 + 2] = d[2] / nrm;
    }
    if (ray_batch) {
      float* b = ray_batch + idx * 11;
      b[0] = cam.t[0]; b[1] = cam.t[1]; b[2] = cam.t[2];
      b[3] = d[0]; b[4] = d[1]; b[5] = d[2];
      b[6] = near_; b[7] = far_;
      b[8] = d[0] / nrm; b[9] = d[1] / nrm; b[10] = d[2] / nrm;
    }
  }
}

// a2: utils.py:159-217.  t[.,0] is the minus-sqrt root.
__global__ void __launch_bounds__(kBlock)
sphere_kernel(const float* __restrict__ o, const float* __restrict__ d, int64_t R, float radius,
              float* __restrict__ t_out, float* __restrict__ p_out) {
  for (int64_t r = blockIdx.x * (int64_t)kBlock + threadIdx.x; r < R;
       r += (int64_t)gridDim.x * kBlock) {
    const float ox = o[r * 3], oy = o[r * 3 + 1], oz = o[r * 3 + 2];
    const float dx = d[r * 3], dy = d[r * 3 + 1], dz = d[r * 3 + 2];
    const float b = 2.0f * ((dx * ox + dy * oy) + dz * oz);
    const float on = sqrtf(__builtin_fmaf(oz, oz, __builtin_fmaf(oy, oy, ox * ox)));  // torch.norm(o)**2: sqrt, then square
    const float c = on * on - radius * radius;
    const float a = (dx * dx + dy * dy) + dz * dz;
    const float sq = sqrtf(b * b - 4.0f * a * c);           // NaN when the line misses
    const float t0 = (-b - sq) / (2.0f * a);
    const float t1 = (-b + sq) / (2.0f * a);
    if (t_out) { t_out[r * 2] = t0; t_out[r * 2 + 1] = t1; }
    if (p_out) {
      float* p = p_out + r * 6;
      p[0] = ox + t0 * dx; p[1] = oy + t0 * dy; p[2] = oz + t0 * dz;
      p[3] = ox + t1 * dx; p[4] = oy + t1 * dy; p[5] = oz + t1 * dz;
    }
  }
}

__global__ void __launch_bounds__(kBlock)
quadratic_kernel(const float* __restrict__ a, const float* __restrict__ b,
                 const float* __restrict__ c, int64_t n, float* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * kBlock) {
    const float sq = sqrtf(b[i] * b[i] - 4.0f * a[i] * c[i]);
    out[i] = (-b[i] - sq) / (2.0f * a[i]);
    out[n + i] = (-b[i] + sq) / (2.0f * a[i]);
  }
}

// a3: run_nerf_helpers.py:15-63.  One thread per output element so stores are coalesced.
__global__ void __launch_bounds__(kBlock)
posenc_kernel(const float* __restrict__ x, int64_t M, int d, int L, float* __restrict__ out) {
  const int width = d * (1 + 2 * L);
  const int64_t total = M * width;
  for (int64_t e = blockIdx.x * (int64_t)kBlock + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * kBlock) {
    const int64_t m = e / width;
    const int c = static_cast<int>(e % width);
    const int blk = c / d, comp = c % d;
    const float v = x[m * d + comp];
    float r;
    if (blk == 0) {
      r = v;
    } else {
      const int f = blk - 1;
      const float arg = v * exp2f(static_cast<float>(f >> 1));  // freq bands are exact powers of two
      r = (f & 1) ? cosf(arg) : sinf(arg);
    }
    out[e] = r;
  }
}

// a5: utils.py:220-244, values before any sort.  UNIFORM is emitted already sorted + clipped
// (the grid is increasing, so the mean is merged at its rank instead of sorting: nsplace::uniform_z).
__global__ void __launch_bounds__(kBlock)
place_z_kernel(int mode, const float* __restrict__ mean, const float* __restrict__ noise,
               int64_t R, int N, float std_, float* __restrict__ z) {
  const int64_t total = R * N;
  const float step = N > 2 ? (std_ - (-std_)) / static_cast<float>(N - 2) : 0.0f;   // linspace(-std, std, N - 1)
  for (int64_t e = blockIdx.x * (int64_t)kBlock + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * kBlock) {
    const int64_t r = e / N;
    const int j = static_cast<int>(e % N);
    const float m = mean[r];
    float v;
    if (mode == NS_MODE_DEPTH_ONLY) {
      v = m;
    } else if (mode == NS_MODE_GAUSSIAN) {
      v = (j < N - 1) ? m + std_ * noise[r * (N - 1) + j] : m;
    } else {
      v = nsplace::uniform_z(m, std_, step, N - 1, j);   // already sorted + clipped (ns_place.h)
    }
    z[e] = v;
  }
}

// UNIFORM mode, N % 4 == 0: one thread per four consecutive samples of a ray, the store is a float4 (the generic kernel
// above is ~6x off the HBM rate at N = 64).
__global__ void __launch_bounds__(kBlock)
place_z_uniform4_kernel(const float* __restrict__ mean, int64_t R, int N, float std_, float4* __restrict__ z4) {
  const int q = N >> 2, steps = N - 1;
  const float step = steps > 1 ? (std_ - (-std_)) / static_cast<float>(steps - 1) : 0.0f;
  const int64_t total = R * q;
  for (int64_t e = blockIdx.x * (int64_t)kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock) {
    const int64_t r = e / q;
    const int j0 = 4 * static_cast<int>(e - r * q);
    const float m = mean[r];
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = nsplace::uniform_z(m, std_, step, steps, j0 + k);
    z4[e] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// the LAST sample of every ray only (the guard pass of ns_render_rays_*: ns_render_args::nerf_guard)
__global__ void __launch_bounds__(kBlock)
place_last_kernel(const float* __restrict__ mean, int64_t R, int N, float std_, float* __restrict__ z_last) {
  const int steps = N - 1;
  const float step = steps > 1 ? (std_ - (-std_)) / static_cast<float>(steps - 1) : 0.0f;
  for (int64_t r = blockIdx.x * (int64_t)kBlock + threadIdx.x; r < R; r += (int64_t)gridDim.x * kBlock)
    z_last[r] = nsplace::uniform_z(mean[r], std_, step, steps, N - 1);
}

__global__ void __launch_bounds__(kBlock)
points_kernel(const float* __restrict__ o, const float* __restrict__ d,
              const float* __restrict__ z, int64_t R, int N, float* __restrict__ pts) {
  const int64_t total = R * N * 3;
  for (int64_t e = blockIdx.x * (int64_t)kBlock + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * kBlock) {
    const int64_t s = e / 3;
    const int c = static_cast<int>(e % 3);
    const int64_t r = s / N;
    pts[e] = o[r * 3 + c] + d[r * 3 + c] * z[s];
  }
}

// a11: Trainer.py:603-626
__global__ void __launch_bounds__(kBlock)
coarse_z_kernel(const float* __restrict__ near_, const float* __restrict__ far_, float near_s, float far_s,
                int64_t R, int N, int lindisp, const float* __restrict__ t_rand, float* __restrict__ z) {
  const int64_t total = R * N;
  for (int64_t e = blockIdx.x * (int64_t)kBlock + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * kBlock) {
    const int64_t r = e / N;
    const int i = static_cast<int>(e % N);
    const float nr = near_ ? near_[r] : near_s, fr = far_ ? far_[r] : far_s;
    auto zval = [&](int k) {
      const float t = linspace_at(0.0f, 1.0f, N, k);
      return lindisp ? 1.0f / (1.0f / nr * (1.0f - t) + 1.0f / fr * t) : nr * (1.0f - t) + fr * t;
    };
    float v = zval(i);
    if (t_rand) {
      const float upper = (i < N - 1) ? 0.5f * (zval(i + 1) + v) : v;
      const float lower = (i > 0) ? 0.5f * (v + zval(i - 1)) : v;
      v = lower + (upper - lower) * t_rand[e];
    }
    z[e] = v;
  }
}

// torch.sort(x, -1).values per row, one wave per row, bitonic in LDS; NaN sorts last.
__device__ __forceinline__ bool sort_less(float a, float b) {
  return !(a != a) && ((b != b) || a < b);
}

template <int P>
__global__ void __launch_bounds__(64)
sort_rows_kernel(const float* x, int64_t R, int N, float* out) {  // x may alias out
  __shared__ float buf[P];
  const int lane = threadIdx.x;
  for (int64_t r = blockIdx.x; r < R; r += gridDim.x) {
    for (int i = lane; i < P; i += 64) buf[i] = (i < N) ? x[r * N + i] : __builtin_nanf("");
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
    for (int i = lane; i < N; i += 64) out[r * N + i] = buf[i];
    __syncthreads();
  }
}

}  // namespace

extern "C" {

int ns_get_rays(int H, int W, float fx, float fy, float cx, float cy, const float* c2w_host,
                int row0, int row1, float near_, float far_, float* rays_o_dev, float* rays_d_dev,
                float* viewdirs_dev, float* ray_batch_dev, void* stream) {
  NS_REQUIRE(H > 0 && W > 0 && c2w_host, "bad camera");
  NS_REQUIRE(row0 >= 0 && row1 <= H && row0 <= row1, "bad row range");
  const int64_t R = static_cast<int64_t>(row1 - row0) * W;
  if (R == 0) return NS_OK;
  Cam cam{fx, fy, cx, cy, {}, {}};
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) cam.r[3 * r + c] = c2w_host[4 * r + c];
    cam.t[r] = c2w_host[4 * r + 3];
  }
  get_rays_kernel<<<ns::ew_grid(R, kBlock), kBlock, 0, ns::as_stream(stream)>>>(
      cam, W, row0, R, near_, far_, rays_o_dev, rays_d_dev, viewdirs_dev, ray_batch_dev);
  NS_LAUNCH_CHECK();
  return NS_OK;
}

int ns_sphere_intersect(const float* o_dev, const float* d_dev, int64_t R, float radius,
                        float* t_dev, float* pts_dev, void* stream) {
  NS_REQUIRE(R >= 0, "negative ray count");
  if (R == 0) return NS_OK;
  NS_REQUIRE(o_dev && d_dev, "null rays");
  sphere_kernel<<<ns::ew_grid(R, kBlock), kBlock, 0, ns::as_stream(stream)>>>(o_dev, d_dev, R, radius,
                                                                             t_dev, pts_dev);
  NS_LAUNCH_CHECK();
  return NS_OK;
}

int ns_solve_quadratic(const float* a_dev, const float* b_dev, const float* c_dev, int64_t n,
                       float* out_dev, void* stream) {
  NS_REQUIRE(n >= 0, "negative size");
  if (n == 0) return NS_OK;
  NS_REQUIRE(a_dev && b_dev && c_dev && out_dev, "null pointer");
  quadratic_kernel<<<ns::ew_grid(n, kBlock), kBlock, 0, ns::as_stream(stream)>>>(a_dev, b_dev, c_dev, n,
                                                                                out_dev);
  NS_LAUNCH_CHECK();
  return NS_OK;
}

int ns_posenc(const float* x_dev, int64_t M, int d, int n_freqs, float* out_dev, void* stream) {
  NS_REQUIRE(M >= 0 && d > 0 && n_freqs >= 0, "bad shape");
  if (M == 0) return NS_OK;
  NS_REQUIRE(x_dev && out_dev, "null pointer");
  posenc_kernel<<<ns::ew_grid(M * d * (1 + 2 * n_freqs), kBlock), kBlock, 0, ns::as_stream(stream)>>>(
      x_dev, M, d, n_freqs, out_dev);
  NS_LAUNCH_CHECK();
  return NS_OK;
}

int ns_sort_rows(const float* x_dev, int64_t R, int N, float* out_dev, void* stream) {
  NS_REQUIRE(R >= 0 && N >= 0, "bad shape");
  if (R == 0 || N == 0) return NS_OK;
  NS_REQUIRE(x_dev && out_dev, "null pointer");
  NS_REQUIRE(N <= 2048, "rows longer than 2048 are not supported");
  const int grid = static_cast<int>(R < 256 * 32 ? R : 256 * 32);
  hipStream_t s = ns::as_stream(stream);
  if (N <= 64) sort_rows_kernel<64><<<grid, 64, 0, s>>>(x_dev, R, N, out_dev);
  else if (N <= 128) sort_rows_kernel<128><<<grid, 64, 0, s>>>(x_dev, R, N, out_dev);
  else if (N <= 256) sort_rows_kernel<256><<<grid, 64, 0, s>>>(x_dev, R, N, out_dev);
  else if (N <= 512) sort_rows_kernel<512><<<grid, 64, 0, s>>>(x_dev, R, N, out_dev);
  else if (N <= 1024) sort_rows_kernel<1024><<<grid, 64, 0, s>>>(x_dev, R, N, out_dev);
  else sort_rows_kernel<2048><<<grid, 64, 0, s>>>(x_dev, R, N, out_dev);
  NS_LAUNCH_CHECK();
  return NS_OK;
}

int ns_points_along_rays(const float* o_dev, const float* d_dev, const float* z_dev, int64_t R,
                         int N, float* pts_dev, void* stream) {
  NS_REQUIRE(R >= 0 && N >= 0, "bad shape");
  if (R == 0 || N == 0) return NS_OK;
  NS_REQUIRE(o_dev && d_dev && z_dev && pts_dev, "null pointer");
  points_kernel<<<ns::ew_grid(R * N * 3, kBlock), kBlock, 0, ns::as_stream(stream)>>>(o_dev, d_dev, z_dev,
                                                                                      R, N, pts_dev);
  NS_LAUNCH_CHECK();
  return NS_OK;
}

int ns_place_samples(int mode, const float* o_dev, const float* d_dev, const float* mean_dev,
                     const float* noise_dev, int64_t R, int N, float std_, float* pts_dev,
                     float* z_dev, void* stream) {
  NS_REQUIRE(mode == NS_MODE_DEPTH_ONLY || mode == NS_MODE_UNIFORM || mode == NS_MODE_GAUSSIAN,
             "unknown mode");
  if (mode == NS_MODE_DEPTH_ONLY) N = 1;
  NS_REQUIRE(R >= 0 && N >= 1, "bad shape");
  if (R == 0) return NS_OK;
  NS_REQUIRE(mean_dev && z_dev, "mean and z are required");
  NS_REQUIRE(mode != NS_MODE_GAUSSIAN || N == 1 || noise_dev, "gaussian mode needs the noise draws");
  NS_REQUIRE(mode != NS_MODE_UNIFORM || N >= 2, "uniform mode needs n_samples >= 2");
  if (mode == NS_MODE_UNIFORM && (N & 3) == 0 && (reinterpret_cast<uintptr_t>(z_dev) & 15) == 0)
    place_z_uniform4_kernel<<<ns::ew_grid(R * (N >> 2), kBlock), kBlock, 0, ns::as_stream(stream)>>>(
        mean_dev, R, N, std_, reinterpret_cast<float4*>(z_dev));
  else
    place_z_kernel<<<ns::ew_grid(R * N, kBlock), kBlock, 0, ns::as_stream(stream)>>>(mode, mean_dev, noise_dev,
                                                                                    R, N, std_, z_dev);
  NS_LAUNCH_CHECK();
  if (mode == NS_MODE_GAUSSIAN && N > 1) {
    int rc = ns_sort_rows(z_dev, R, N, z_dev, stream);
    if (rc != NS_OK) return rc;
  }
  if (pts_dev) {
    NS_REQUIRE(o_dev && d_dev, "pts requested without rays");
    return ns_points_along_rays(o_dev, d_dev, z_dev, R, N, pts_dev, stream);
  }
  return NS_OK;
}

int ns_coarse_z(const float* near_dev, const float* far_dev, int64_t R, int N, int lindisp,
                const float* t_rand_dev, float* z_dev, void* stream) {
  NS_REQUIRE(R >= 0 && N >= 1, "bad shape");
  if (R == 0) return NS_OK;
  NS_REQUIRE(near_dev && far_dev && z_dev, "null pointer");
  coarse_z_kernel<<<ns::ew_grid(R * N, kBlock), kBlock, 0, ns::as_stream(stream)>>>(near_dev, far_dev, 0.f, 0.f, R, N,
                                                                                   lindisp, t_rand_dev, z_dev);
  NS_LAUNCH_CHECK();
  return NS_OK;
}

// same with one near / far for every ray (used by ns_render_rays_hierarchical)
int ns_coarse_z_scalar(float near_, float far_, int64_t R, int N, int lindisp, const float* t_rand_dev,
                       float* z_dev, void* stream) {
  NS_REQUIRE(R >= 0 && N >= 1, "bad shape");
  if (R == 0) return NS_OK;
  NS_REQUIRE(z_dev, "null pointer");
  coarse_z_kernel<<<ns::ew_grid(R * N, kBlock), kBlock, 0, ns::as_stream(stream)>>>(nullptr, nullptr, near_, far_, R, N,
                                                                                   lindisp, t_rand_dev, z_dev);
  NS_LAUNCH_CHECK();
  return NS_OK;
}

}  // extern "C"

#include "ns_weights.h"
int ns_place_last_sample(const float* mean_dev, int64_t R, int N, float std_, float* z_last_dev, void* stream) {
  NS_REQUIRE(R >= 0 && N >= 2 && mean_dev && z_last_dev, "bad arguments");
  if (R == 0) return NS_OK;
  place_last_kernel<<<ns::ew_grid(R, kBlock), kBlock, 0, ns::as_stream(stream)>>>(mean_dev, R, N, std_, z_last_dev);
  NS_LAUNCH_CHECK();
  return NS_OK;
}
