// Practical MFMA ceiling of THIS chip: bare v_mfma_f32_32x32x16_bf16 loop, operands in registers,
// random data, every SIMD busy.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(512) bare(const bf16x8* __restrict__ in, float* out, int iters) {
  bf16x8 a = in[threadIdx.x], b = in[512 + threadIdx.x];
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Same loop, but the A and B operands change on every MFMA (8 x 8 register-resident fragments), as they do in any
// real GEMM: the operand buses and multiplier inputs toggle, which the fixed-operand loop above does not pay for.
template <int NACC>
__global__ void __launch_bounds__(512) rotating(const bf16x8* __restrict__ in, float* out, int iters) {
  bf16x8 a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = in[(2 * i) * 512 + threadIdx.x]; b[i] = in[(2 * i + 1) * 512 + threadIdx.x]; }
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + k) & 7], b[(k + 3 * i) & 7], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Same rotating-operand loop on the other dense bf16 shape, v_mfma_f32_16x16x32_bf16 (4-register accumulators)
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(512) rotating16(const bf16x8* __restrict__ in, float* out, int iters) {
  bf16x8 a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = in[(2 * i) * 512 + threadIdx.x]; b[i] = in[(2 * i + 1) * 512 + threadIdx.x]; }
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + k) & 7], b[(k + 3 * i) & 7], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  std::vector<uint16_t> h(16 * 512 * 8);
  srand(1);
  for (auto& v : h) { float f = (rand() / (float)RAND_MAX) * 2.f - 1.f; uint32_t u; std::memcpy(&u, &f, 4); v = u >> 16; }
  bf16x8* din; float* dout;
  hipMalloc(&din, h.size() * 2); hipMalloc(&dout, 256 * 8 * 512 * 4);
  hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int threads : {256, 512}) {
    const int iters = 250000, nacc = 8, grid = 256;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      bare<8><<<grid, threads>>>(din, dout, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double flop = (double)grid * (threads / 64) * iters * nacc * 2.0 * 32 * 32 * 16;
      printf("waves/SIMD %d: %.1f ms, %.1f TFLOP/s (%.1f%% of 2500)\n", threads / 256, ms, flop / ms / 1e9, flop / ms / 1e9 / 25.0);
    }
  }
  // rotating operands: dense random, then with ReLU-like B operands (half the values zero)
  for (int relu = 0; relu < 2; ++relu) {
    if (relu) {
      for (int f = 0; f < 8; ++f)
        for (int i = 0; i < 512 * 8; ++i) { uint16_t& v = h[((2 * f + 1) * 512) * 8 + i]; if (v & 0x8000) v = 0; }
      hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    }
    for (int threads : {256, 512}) {
      const int iters = 31250, nacc = 8, grid = 256;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        rotating<8><<<grid, threads>>>(din, dout, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flop = (double)grid * (threads / 64) * iters * 8 * nacc * 2.0 * 32 * 32 * 16;
        printf("rotating operands%s, waves/SIMD %d: %.1f ms, %.1f TFLOP/s (%.1f%% of 2500)\n", relu ? " (B = relu)" : "", threads / 256, ms, flop / ms / 1e9, flop / ms / 1e9 / 25.0);
      }
    }
  }
  for (int threads : {256, 512}) {
    const int iters = 62500, nacc = 8, grid = 256;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      rotating16<8><<<grid, threads>>>(din, dout, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double flop = (double)grid * (threads / 64) * iters * 8 * nacc * 2.0 * 16 * 16 * 32;
      printf("16x16x32 rotating operands (B = relu), waves/SIMD %d: %.1f ms, %.1f TFLOP/s (%.1f%% of 2500)\n", threads / 256, ms, flop / ms / 1e9, flop / ms / 1e9 / 25.0);
    }
  }
  return 0;
}
