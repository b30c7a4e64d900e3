// Kernels for the DepthNet training step (Trainer.core_optimization_loop, Trainer.py:506-544): the
// backward of the layers the optimiser actually updates (DepthNet: weights) or differentiates through
// (frozen NeRF: input points only; N=1 compositing: sigmoid).  Batches are 1024 rays (lego.yaml N_rand),
// i.e. a few GFLOP per step -- launch-bound, so one generic strided fp32 MFMA GEMM
// (v_mfma_f32_32x32x2_f32, exact fp32 products) serves forward, grad-input and grad-weight alike.
#include "ns_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// C[i,j] (+)= sum_k A[i*sa0 + k*sa1] * B[j*sb0 + k*sb1]  (+ bias[j]),  i < M, j < N, k < K
//
// The training GEMMs are small (1024 rays x 256 features x K <= 1020) and there are ~140 of them per step, so the
// kernel is built for latency, not for tile reuse: one workgroup = one 32x32 output tile (256-320 workgroups per
// GEMM fill the chip), its 4 waves split K four ways, operands go global -> registers directly in MFMA layout (lane
// (r, h) feeds row r with the k values 8q + 4h + m, m = 0..3, of its slice: no LDS staging, no barrier in the K
// loop, all loads of a slice in flight at once), and the four partial tiles are summed through LDS in a fixed
// order (deterministic).  v_mfma_f32_32x32x2_f32: exact fp32 products.
//
// Epilogues (round 4: a training step was ~360 launches, a third of them 5-us elementwise / reduction kernels around the
// GEMMs): `act` applies an activation to the result (forward layers); `dact` multiplies the result by the derivative of the
// activation whose OUTPUT is dref (the grad-input GEMM of the layer above does the backward through this layer's
// activation); `a_rowsum` receives sum_k A[i, k] from the workgroups of the first output column -- with A = dy^T that is
// the bias gradient, which the grad-weight GEMM reads anyway.
struct GemmEpilogue {
  int act;                 // 0 none, 1 relu, 2 leaky(0.01), 3 sigmoid: C = act(acc + bias)
  int dact;                // 0 none, else C = acc * act'(.) of activation `dact`, evaluated from its output dref[i, j]
  const float* dref;
  int64_t ld_ref;
  float* a_rowsum;         // [M] or null
};

// KT: k values per trip of a wave (32 or 64).  The kernel is bound by the CHAIN of trips -- each one a global-load round
// trip, ~1.5 us -- not by bandwidth or MFMAs: with 64 the grad-weight GEMMs (K = 1024 rows, 256 per wave) take 4 trips
// instead of 8.  Same MFMA order per accumulator, so the results do not depend on KT.
template <bool A_KCONTIG, bool B_KCONTIG, int KT>
__device__ __forceinline__ void gemm_tile(const float* __restrict__ A, int64_t sa0, int64_t sa1, const float* __restrict__ B,
                                          int64_t sb0, int64_t sb1, const float* __restrict__ bias, float* __restrict__ C,
                                          int64_t ldc, int M, int N, int K, int accumulate, const GemmEpilogue& ep) {
  __shared__ float red[4][16][64];
  __shared__ float asum_s[4][64];
  float asum = 0.f;
  const bool want_rowsum = ep.a_rowsum != nullptr && blockIdx.x == 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  const int kq = (((K + 3) / 4) + 7) & ~7;            // K slice per wave, a multiple of 8
  const int kbeg = wave * kq, kend = min(K, kbeg + kq);
  const int i = i0 + r, j = j0 + r;
  const bool iok = i < M, jok = j < N;
  const float* ap = A + static_cast<int64_t>(iok ? i : 0) * sa0;
  const float* bp = B + static_cast<int64_t>(jok ? j : 0) * sb0;
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  // KT k per trip: KT / 2 + KT / 2 independent loads in flight, then KT / 2 MFMAs
  constexpr int NE = KT / 2;
  for (int k0 = kbeg; k0 < kend; k0 += KT) {
    float a[NE], b[NE];
#pragma unroll
    for (int c = 0; c < KT / 8; ++c)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int k = k0 + 8 * c + 4 * h + m;
        const bool kok = k < kend;
        a[4 * c + m] = (iok && kok) ? ap[A_KCONTIG ? k : k * sa1] : 0.f;
        b[4 * c + m] = (jok && kok) ? bp[B_KCONTIG ? k : k * sb1] : 0.f;
      }
#pragma unroll
    for (int e = 0; e < NE; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
    if (want_rowsum) {
#pragma unroll
      for (int e = 0; e < NE; ++e) asum += a[e];
    }
  }
  if (want_rowsum) asum_s[wave][lane] = asum;
#pragma unroll
  for (int q = 0; q < 16; ++q) red[wave][q][lane] = acc[q];
  __syncthreads();
  // wave w finishes accumulator registers 4w .. 4w+3 (rows (q & 3) + 8 (q >> 2) + 4 h of the tile)
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int q = 4 * wave + t;
    const int row = i0 + (q & 3) + 8 * (q >> 2) + 4 * h;
    if (row < M && jok) {
      float v = (red[0][q][lane] + red[1][q][lane]) + (red[2][q][lane] + red[3][q][lane]);
      if (bias) v += bias[j];
      float* c = C + static_cast<int64_t>(row) * ldc + j;
      if (accumulate) v += *c;
      if (ep.act == 1) v = fmaxf(v, 0.f);
      else if (ep.act == 2) v = v > 0.f ? v : 0.01f * v;
      else if (ep.act == 3) v = 1.f / (1.f + expf(-v));
      if (ep.dact) {
        const float y = ep.dref[static_cast<int64_t>(row) * ep.ld_ref + j];
        v *= ep.dact == 1 ? (y > 0.f ? 1.f : 0.f) : ep.dact == 2 ? (y > 0.f ? 1.f : 0.01f) : y * (1.f - y);
      }
      *c = v;
    }
  }
  // row sums of A (fixed order: the two k-halves of each wave, waves 0..3), one thread per row of the tile
  if (want_rowsum && threadIdx.x < 32 && i0 + static_cast<int>(threadIdx.x) < M) {
    const int rr = threadIdx.x;
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) t += asum_s[w][rr] + asum_s[w][rr + 32];
    ep.a_rowsum[i0 + rr] = t;
  }
}

template <bool A_KCONTIG, bool B_KCONTIG, int KT = 32>
__global__ void __launch_bounds__(256)
gemm_strided_kernel(const float* __restrict__ A, int64_t sa0, int64_t sa1, const float* __restrict__ B,
                    int64_t sb0, int64_t sb1, const float* __restrict__ bias, float* __restrict__ C, int64_t ldc,
                    int M, int N, int K, int accumulate, GemmEpilogue ep) {
  gemm_tile<A_KCONTIG, B_KCONTIG, KT>(A, sa0, sa1, B, sb0, sb1, bias, C, ldc, M, N, K, accumulate, ep);
}

// Up to four GEMMs of the same kind in ONE launch (blockIdx.z picks the problem; the grid covers the largest): the three skip
// branches of the DepthNet run the same layer shape side by side, and a training step is bound by its launch count.
struct GemmBatch {
  ns_gemm_problem p[4];
};
template <bool A_KCONTIG, bool B_KCONTIG, int KT>
__global__ void __launch_bounds__(256)
gemm_batched_kernel(GemmBatch batch) {
  const ns_gemm_problem& q = batch.p[blockIdx.z];
  if (static_cast<int>(blockIdx.x) * 32 >= q.N || static_cast<int>(blockIdx.y) * 32 >= q.M) return;   // (workgroup-uniform)
  const GemmEpilogue ep{q.act, q.dact, q.dact_ref_dev, q.ld_ref, q.a_rowsum_dev};
  gemm_tile<A_KCONTIG, B_KCONTIG, KT>(q.A_dev, q.sa0, q.sa1, q.B_dev, q.sb0, q.sb1, q.bias_dev, q.C_dev, q.ldc, q.M, q.N, q.K,
                                      q.accumulate, ep);
}

// column sums: out[j] = sum_i X[i*ld + j]   (bias gradient).  Block = 32 columns x 32 row lanes; coalesced
// 128-B row segments, fixed-order LDS reduction over the row lanes.
__global__ void __launch_bounds__(1024)
colsum_kernel(const float* __restrict__ X, int64_t ld, int M, int N, float* __restrict__ out) {
  __shared__ float part[32][33];
  const int c = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + c;
  float s = 0.f;
  if (j < N)
    for (int i = rl; i < M; i += 32) s += X[i * ld + j];
  part[rl][c] = s;
  __syncthreads();
  if (rl == 0 && j < N) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) t += part[k][c];
    out[j] = t;
  }
}

// activation forward in place / backward: act 0 none, 1 relu, 2 leaky(0.01), 3 sigmoid
__global__ void __launch_bounds__(256)
act_forward_kernel(float* __restrict__ y, int64_t n, int act) {
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = y[i];
    y[i] = act == 1 ? fmaxf(v, 0.f) : act == 2 ? (v > 0.f ? v : 0.01f * v) : act == 3 ? 1.f / (1.f + expf(-v)) : v;
  }
}
// dy *= act'(.) given the activation OUTPUT y
__global__ void __launch_bounds__(256)
act_backward_kernel(float* __restrict__ dy, const float* __restrict__ y, int64_t n, int act) {
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = y[i];
    const float g = act == 1 ? (v > 0.f ? 1.f : 0.f) : act == 2 ? (v > 0.f ? 1.f : 0.01f) : act == 3 ? v * (1.f - v) : 1.f;
    dy[i] *= g;
  }
}

// d/dx of the positional encoding: x [M,d], de [M, d(1+2L)] -> dx [M,d]
__global__ void __launch_bounds__(256)
posenc_backward_kernel(const float* __restrict__ x, const float* __restrict__ de, int64_t M, int d, int L,
                       float* __restrict__ dx) {
  const int width = d * (1 + 2 * L);
  for (int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x; e < M * d; e += (int64_t)gridDim.x * 256) {
    const int64_t m = e / d;
    const int c = static_cast<int>(e % d);
    const float v = x[e];
    const float* g = de + m * width;
    float s = g[c];
    for (int l = 0; l < L; ++l) {
      const float f = exp2f(static_cast<float>(l));
      float sn, cs;
      sincosf(v * f, &sn, &cs);
      s += f * (cs * g[d + 2 * d * l + c] - sn * g[d + 2 * d * l + d + c]);
    }
    dx[e] = s;
  }
}

// dz[r] (+)= sum_c dpts[r,n,c] * d[r,c]   for pts = o + d*z with N samples per ray
__global__ void __launch_bounds__(256)
points_backward_kernel(const float* __restrict__ dpts, const float* __restrict__ d, int64_t R, int N,
                       float* __restrict__ dz) {
  for (int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x; e < R * N; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / N;
    dz[e] = (dpts[e * 3] * d[r * 3] + dpts[e * 3 + 1] * d[r * 3 + 1]) + dpts[e * 3 + 2] * d[r * 3 + 2];
  }
}

// torch.optim.Adam step (no weight decay, no amsgrad), in place
__global__ void __launch_bounds__(256)
adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
            int64_t n, float lr, float b1, float b2, float eps, float bc1, float bc2) {
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = g[i];
    const float mi = m[i] = b1 * m[i] + (1.f - b1) * gi;
    const float vi = v[i] = b2 * v[i] + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    p[i] -= (lr / bc1) * (mi / denom);
  }
}

// the same update with the step counter (and optionally the learning rate) read from device memory: a captured hipGraph
// of the training step replays it unchanged while the bias corrections follow the counter
__global__ void __launch_bounds__(256)
adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                int64_t n, float lr, const float* __restrict__ lr_dev, float b1, float b2, float eps,
                const int* __restrict__ step_dev) {
  const float step = static_cast<float>(*step_dev);
  const float bc1 = 1.f - powf(b1, step), bc2 = 1.f - powf(b2, step);
  if (lr_dev) lr = *lr_dev;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = g[i];
    const float mi = m[i] = b1 * m[i] + (1.f - b1) * gi;
    const float vi = v[i] = b2 * v[i] + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    p[i] -= (lr / bc1) * (mi / denom);
  }
}

__global__ void add_i32_kernel(int* __restrict__ x, int delta) { *x += delta; }

// every parameter tensor of the optimiser in ONE launch: blockIdx.y = tensor, table row = {p, g, m, v, n}
__global__ void __launch_bounds__(256)
adam_multi_dev_kernel(const ns_adam_tensor* __restrict__ table, float lr, const float* __restrict__ lr_dev, float b1,
                      float b2, float eps, const int* __restrict__ step_dev) {
  const ns_adam_tensor t = table[blockIdx.y];
  const float step = static_cast<float>(*step_dev);
  const float bc1 = 1.f - powf(b1, step), bc2 = 1.f - powf(b2, step);
  if (lr_dev) lr = *lr_dev;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < t.n; i += (int64_t)gridDim.x * 256) {
    const float gi = t.g[i];
    const float mi = t.m[i] = b1 * t.m[i] + (1.f - b1) * gi;
    const float vi = t.v[i] = b2 * t.v[i] + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    t.p[i] -= (lr / bc1) * (mi / denom);
  }
}

}  // namespace

extern "C" {

int ns_gemm_strided(const float* A_dev, int64_t sa0, int64_t sa1, const float* B_dev, int64_t sb0, int64_t sb1,
                    const float* bias_dev, float* C_dev, int64_t ldc, int M, int N, int K, int accumulate,
                    void* stream) {
  return ns_gemm_fused(A_dev, sa0, sa1, B_dev, sb0, sb1, bias_dev, C_dev, ldc, M, N, K, accumulate, 0, 0, nullptr, 0,
                       nullptr, stream);
}

int ns_gemm_fused(const float* A_dev, int64_t sa0, int64_t sa1, const float* B_dev, int64_t sb0, int64_t sb1,
                  const float* bias_dev, float* C_dev, int64_t ldc, int M, int N, int K, int accumulate, int act,
                  int dact, const float* dact_ref_dev, int64_t ld_ref, float* a_rowsum_dev, void* stream) {
  NS_REQUIRE(M >= 0 && N >= 0 && K >= 0, "bad shape");
  NS_REQUIRE(act >= 0 && act <= 3 && dact >= 0 && dact <= 3 && (dact == 0 || dact_ref_dev), "bad epilogue");
  if (M == 0 || N == 0) return NS_OK;
  NS_REQUIRE(A_dev && B_dev && C_dev, "null pointer");
  dim3 grid((N + 31) / 32, (M + 31) / 32);
  hipStream_t s = ns::as_stream(stream);
  const GemmEpilogue ep{act, dact, dact_ref_dev, ld_ref, a_rowsum_dev};
  const bool wide = K >= 256;      // >= 64 k per wave: trips of 64
#define NS_GEMM(AK, BK)                                                                                                  \
  do {                                                                                                                   \
    if (wide) gemm_strided_kernel<AK, BK, 64><<<grid, 256, 0, s>>>(A_dev, sa0, sa1, B_dev, sb0, sb1, bias_dev, C_dev, ldc, \
                                                                   M, N, K, accumulate, ep);                               \
    else gemm_strided_kernel<AK, BK, 32><<<grid, 256, 0, s>>>(A_dev, sa0, sa1, B_dev, sb0, sb1, bias_dev, C_dev, ldc, M,   \
                                                              N, K, accumulate, ep);                                       \
  } while (0)
  if (sa1 == 1 && sb1 == 1) NS_GEMM(true, true);
  else if (sa1 == 1) NS_GEMM(true, false);
  else if (sb1 == 1) NS_GEMM(false, true);
  else NS_GEMM(false, false);
#undef NS_GEMM
  NS_LAUNCH_CHECK();
  return NS_OK;
}

int ns_gemm_fused_batched(const ns_gemm_problem* problems_host, int count, void* stream) {
  NS_REQUIRE(problems_host && count >= 1 && count <= 4, "1 to 4 problems per launch");
  GemmBatch batch{};
  int gx = 0, gy = 0, kmax = 0;
  const bool ak = problems_host[0].sa1 == 1, bk = problems_host[0].sb1 == 1;
  for (int b = 0; b < count; ++b) {
    const ns_gemm_problem& q = problems_host[b];
    NS_REQUIRE(q.M >= 1 && q.N >= 1 && q.K >= 0 && q.A_dev && q.B_dev && q.C_dev, "bad problem");
    NS_REQUIRE(q.act >= 0 && q.act <= 3 && q.dact >= 0 && q.dact <= 3 && (q.dact == 0 || q.dact_ref_dev), "bad epilogue");
    NS_REQUIRE((q.sa1 == 1) == ak && (q.sb1 == 1) == bk, "the problems of one launch share their operand layout (k-contiguous or not)");
    batch.p[b] = q;
    gx = gx > (q.N + 31) / 32 ? gx : (q.N + 31) / 32;
    gy = gy > (q.M + 31) / 32 ? gy : (q.M + 31) / 32;
    kmax = kmax > q.K ? kmax : q.K;
  }
  dim3 grid(gx, gy, count);
  hipStream_t s = ns::as_stream(stream);
  const bool wide = kmax >= 256;
#define NS_GEMMB(AK, BK)                                                             \
  do {                                                                               \
    if (wide) gemm_batched_kernel<AK, BK, 64><<<grid, 256, 0, s>>>(batch);           \
    else gemm_batched_kernel<AK, BK, 32><<<grid, 256, 0, s>>>(batch);                \
  } while (0)
  if (ak && bk) NS_GEMMB(true, true);
  else if (ak) NS_GEMMB(true, false);
  else if (bk) NS_GEMMB(false, true);
  else NS_GEMMB(false, false);
#undef NS_GEMMB
  NS_LAUNCH_CHECK();
  return NS_OK;
}

int ns_colsum(const float* X_dev, int64_t ld, int M, int N, float* out_dev, void* stream) {
  NS_REQUIRE(M >= 0 && N >= 0, "bad shape");
  if (N == 0) return NS_OK;
  NS_REQUIRE(X_dev && out_dev, "null pointer");
  colsum_kernel<<<(N + 31) / 32, 1024, 0, ns::as_stream(stream)>>>(X_dev, ld, M, N, out_dev);
  NS_LAUNCH_CHECK();
  return NS_OK;
}

int ns_act_forward(float* y_dev, int64_t n, int act, void* stream) {
  NS_REQUIRE(n >= 0 && act >= 0 && act <= 3, "bad argument");
  if (n == 0 || act == 0) return NS_OK;
  NS_REQUIRE(y_dev, "null pointer");
  act_forward_kernel<<<ns::ew_grid(n, 256), 256, 0, ns::as_stream(stream)>>>(y_dev, n, act);
  NS_LAUNCH_CHECK();
  return NS_OK;
}

int ns_act_backward(float* dy_dev, const float* y_dev, int64_t n, int act, void* stream) {
  NS_REQUIRE(n >= 0 && act >= 0 && act <= 3, "bad argument");
  if (n == 0 || act == 0) return NS_OK;
  NS_REQUIRE(dy_dev && y_dev, "null pointer");
  act_backward_kernel<<<ns::ew_grid(n, 256), 256, 0, ns::as_stream(stream)>>>(dy_dev, y_dev, n, act);
  NS_LAUNCH_CHECK();
  return NS_OK;
}

int ns_posenc_backward(const float* x_dev, const float* de_dev, int64_t M, int d, int n_freqs, float* dx_dev,
                       void* stream) {
  NS_REQUIRE(M >= 0 && d > 0 && n_freqs >= 0, "bad shape");
  if (M == 0) return NS_OK;
  NS_REQUIRE(x_dev && de_dev && dx_dev, "null pointer");
  posenc_backward_kernel<<<ns::ew_grid(M * d, 256), 256, 0, ns::as_stream(stream)>>>(x_dev, de_dev, M, d, n_freqs,
                                                                                    dx_dev);
  NS_LAUNCH_CHECK();
  return NS_OK;
}

int ns_points_backward(const float* dpts_dev, const float* d_dev, int64_t R, int N, float* dz_dev, void* stream) {
  NS_REQUIRE(R >= 0 && N >= 1, "bad shape");
  if (R == 0) return NS_OK;
  NS_REQUIRE(dpts_dev && d_dev && dz_dev, "null pointer");
  points_backward_kernel<<<ns::ew_grid(R * N, 256), 256, 0, ns::as_stream(stream)>>>(dpts_dev, d_dev, R, N, dz_dev);
  NS_LAUNCH_CHECK();
  return NS_OK;
}

int ns_adam_step(float* p_dev, const float* g_dev, float* m_dev, float* v_dev, int64_t n, float lr, float beta1,
                 float beta2, float eps, int step, void* stream) {
  NS_REQUIRE(n >= 0 && step >= 1, "bad argument");
  if (n == 0) return NS_OK;
  NS_REQUIRE(p_dev && g_dev && m_dev && v_dev, "null pointer");
  const float bc1 = 1.f - powf(beta1, static_cast<float>(step));
  const float bc2 = 1.f - powf(beta2, static_cast<float>(step));
  adam_kernel<<<ns::ew_grid(n, 256), 256, 0, ns::as_stream(stream)>>>(p_dev, g_dev, m_dev, v_dev, n, lr, beta1, beta2,
                                                                     eps, bc1, bc2);
  NS_LAUNCH_CHECK();
  return NS_OK;
}

int ns_adam_step_dev(float* p_dev, const float* g_dev, float* m_dev, float* v_dev, int64_t n, float lr,
                     const float* lr_dev, float beta1, float beta2, float eps, const int* step_dev, void* stream) {
  NS_REQUIRE(n >= 0, "bad argument");
  if (n == 0) return NS_OK;
  NS_REQUIRE(p_dev && g_dev && m_dev && v_dev && step_dev, "null pointer");
  adam_dev_kernel<<<ns::ew_grid(n, 256), 256, 0, ns::as_stream(stream)>>>(p_dev, g_dev, m_dev, v_dev, n, lr, lr_dev, beta1,
                                                                         beta2, eps, step_dev);
  NS_LAUNCH_CHECK();
  return NS_OK;
}

int ns_adam_step_multi_dev(const ns_adam_tensor* table_dev, int n_tensors, int64_t max_n, float lr, const float* lr_dev,
                           float beta1, float beta2, float eps, const int* step_dev, void* stream) {
  NS_REQUIRE(n_tensors >= 0 && max_n >= 0, "bad argument");
  if (n_tensors == 0 || max_n == 0) return NS_OK;
  NS_REQUIRE(table_dev && step_dev && n_tensors <= 65535, "null pointer / too many tensors");
  int64_t bx = ns::cdiv(max_n, 256);
  if (bx > 256) bx = 256;                      // grid-stride over the larger tensors
  adam_multi_dev_kernel<<<dim3(static_cast<unsigned>(bx), static_cast<unsigned>(n_tensors)), 256, 0, ns::as_stream(stream)>>>(
      table_dev, lr, lr_dev, beta1, beta2, eps, step_dev);
  NS_LAUNCH_CHECK();
  return NS_OK;
}

int ns_add_i32(int* x_dev, int delta, void* stream) {
  NS_REQUIRE(x_dev, "null pointer");
  add_i32_kernel<<<1, 1, 0, ns::as_stream(stream)>>>(x_dev, delta);
  NS_LAUNCH_CHECK();
  return NS_OK;
}

}  // extern "C"
