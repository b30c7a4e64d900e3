// DepthNet forward (depth_net.py:117-169) as one persistent MFMA kernel: ray-sphere
// intersection, three positional encodings, the three affine skip branches (the reference never
// applies its LeakyReLU there, depth_net.py:140,148,156), the LeakyReLU trunk and the sigmoid
// head.  One wave owns 32 rays; HBM traffic per ray is 24 B in, 4 B out (+ a per-workgroup stash
// of two branch outputs that stays in L2).
#include "ns_common.h"
#include "ns_mlp_engine.h"
#include "ns_weights.h"

#ifndef NS_DN_WAVES
#define NS_DN_WAVES 8   // waves per workgroup of the 16-bit W = 256 kernel (8: two per SIMD, prologue spills; 4: one per SIMD)
#endif

namespace {

using namespace nsmlp;

struct DepthArgs {
  const char* stream;
  const float* bias;
  uint32_t n_slabs;
  int bias_floats;
  int n_layers;
  const float* o;
  const float* d;
  int64_t R;
  float near_, far_, radius;
  float* z;
  char* scratch;  // [grid][NWAVES][2][NB] blocks of 64 lanes
};

template <class M, int NB>
__device__ __forceinline__ void stash_store(char* base, const typename M::Block (&hcur)[NB], int lane) {
  using Block = typename M::Block;
  static_for<NB>([&](auto b_) {
    constexpr int b = decltype(b_)::value;
    *reinterpret_cast<Block*>(base + (static_cast<size_t>(b) * 64 + lane) * sizeof(Block)) = hcur[b];
  });
}
template <class M, int NB>
__device__ __forceinline__ void stash_load(const char* base, typename M::Block (&hcur)[NB], int lane) {
  using Block = typename M::Block;
  static_for<NB>([&](auto b_) {
    constexpr int b = decltype(b_)::value;
    hcur[b] = *reinterpret_cast<const Block*>(base + (static_cast<size_t>(b) * 64 + lane) * sizeof(Block));
  });
}

// one skip branch: h = e; layer 0 on cat[e, e]; layers >= 1 on cat[h, e]; no activation
template <class M, int NB, int EBLK, class PipeT>
__device__ __forceinline__ void branch(PipeT& ring, f32x16 (&acc)[NB], typename M::Block (&hcur)[NB],
                                       const typename M::Block (&e)[EBLK], const float*& bias, int n_layers,
                                       int h) {
  init_bias<NB>(acc, bias, h); bias += NB * 32;
  consume<M, NB, EBLK>(ring, acc, e);
  consume<M, NB, EBLK>(ring, acc, e);
  to_blocks<M, kNone, NB>(hcur, acc);
  for (int i = 1; i < n_layers; ++i) {
    init_bias<NB>(acc, bias, h); bias += NB * 32;
    consume<M, NB, NB>(ring, acc, hcur);
    consume<M, NB, EBLK>(ring, acc, e);
    to_blocks<M, kNone, NB>(hcur, acc);
  }
}

template <class M, int NB, int NWAVES, bool PRECISE_TRIG>
__global__ void __launch_bounds__(NWAVES * 64)
depthnet_kernel(DepthArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using Block = typename M::Block;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5;

  using PipeT = Pipe<M, NWAVES, 0>;
  float* bias_lds = reinterpret_cast<float*>(smem + PipeT::kLdsBytes);
  for (int i = threadIdx.x; i < a.bias_floats; i += NWAVES * 64) bias_lds[i] = a.bias[i];
  __syncthreads();

  PipeT ring;
  ring.init(a.stream, smem, a.n_slabs, wave, lane);

  char* stash = a.scratch + (static_cast<size_t>(blockIdx.x) * NWAVES + wave) * 2 * NB * 64 * sizeof(Block);
  constexpr size_t kStashBranch = static_cast<size_t>(NB) * 64 * sizeof(Block);

  const int64_t n_tiles = (a.R + 31) / 32;
  const int64_t n_groups = (n_tiles + NWAVES - 1) / NWAVES;
  for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const int64_t tile = g * NWAVES + wave;
    int64_t r = tile * 32 + (lane & 31);
    const bool valid = r < a.R;
    if (!valid) r = a.R - 1;

    float o[3], d[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { o[c] = a.o[r * 3 + c]; d[c] = a.d[r * 3 + c]; }
    // ray-sphere intersections, utils.py:182-217 (NaN when the line misses, by design)
    float x6[6];
    {
      const float b = 2.0f * ((d[0] * o[0] + d[1] * o[1]) + d[2] * o[2]);
      const float on = sqrtf(__builtin_fmaf(o[2], o[2], __builtin_fmaf(o[1], o[1], o[0] * o[0])));  // torch.norm
      const float c = on * on - a.radius * a.radius;
      const float aa = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2];
      const float sq = sqrtf(b * b - 4.0f * aa * c);
      const float t0 = (-b - sq) / (2.0f * aa), t1 = (-b + sq) / (2.0f * aa);
#pragma unroll
      for (int c3 = 0; c3 < 3; ++c3) { x6[c3] = o[c3] + t0 * d[c3]; x6[3 + c3] = o[c3] + t1 * d[c3]; }
    }

    const float* bias = bias_lds;
    f32x16 acc[NB];
    Block hcur[NB];
    {
      Block e3[2];
      embed3<M, PRECISE_TRIG, 10, 2>(e3, o, h);
      branch<M, NB, 2>(ring, acc, hcur, e3, bias, a.n_layers, h);
      stash_store<M, NB>(stash, hcur, lane);
      embed3<M, PRECISE_TRIG, 10, 2>(e3, d, h);
      branch<M, NB, 2>(ring, acc, hcur, e3, bias, a.n_layers, h);
      stash_store<M, NB>(stash + kStashBranch, hcur, lane);
    }
    {
      Block e6[4];
      embed6<M, PRECISE_TRIG>(e6, x6, h);
      branch<M, NB, 4>(ring, acc, hcur, e6, bias, a.n_layers, h);
      // trunk layer 0, K-segments in the order h_x, e_x, h_o, e_o, h_d, e_d
      init_bias<NB>(acc, bias, h); bias += NB * 32;
      consume<M, NB, NB>(ring, acc, hcur);
      consume<M, NB, 4>(ring, acc, e6);
    }
    {
      Block e3[2];
      stash_load<M, NB>(stash, hcur, lane);
      consume<M, NB, NB>(ring, acc, hcur);
      embed3<M, PRECISE_TRIG, 10, 2>(e3, o, h);
      consume<M, NB, 2>(ring, acc, e3);
      stash_load<M, NB>(stash + kStashBranch, hcur, lane);
      consume<M, NB, NB>(ring, acc, hcur);
      embed3<M, PRECISE_TRIG, 10, 2>(e3, d, h);
      consume<M, NB, 2>(ring, acc, e3);
    }
    to_blocks<M, kLeaky, NB>(hcur, acc);
    for (int i = 1; i < a.n_layers; ++i) {
      init_bias<NB>(acc, bias, h); bias += NB * 32;
      consume<M, NB, NB>(ring, acc, hcur);
      to_blocks<M, kLeaky, NB>(hcur, acc);
    }
    f32x16 acc1[1];
    init_bias<1>(acc1, bias, h);
    consume<M, 1, NB>(ring, acc1, hcur);
    if (valid && h == 0) {
      const float depth = 1.0f / (1.0f + expf(-acc1[0][0]));
      a.z[r] = a.near_ * (1.0f - depth) + a.far_ * depth;  // depth_net.py:168
    }
  }
  ring.finish();
}

int depthnet_program_slabs(int cpb, int NB, int n) {
  auto br = [&](int eblk) {
    return 2 * seg_slabs(cpb, NB, eblk) + (n - 1) * (seg_slabs(cpb, NB, NB) + seg_slabs(cpb, NB, eblk));
  };
  int s = 2 * br(2) + br(4);
  s += 3 * seg_slabs(cpb, NB, NB) + seg_slabs(cpb, NB, 4) + 2 * seg_slabs(cpb, NB, 2);
  s += (n - 1) * seg_slabs(cpb, NB, NB);
  s += seg_slabs(cpb, 1, NB);
  return s;
}

template <class M, int NB, int NWAVES, bool PRECISE>
int launch(const ns_weights* net, DepthArgs& a, hipStream_t stream) {
  const size_t lds = static_cast<size_t>(Pipe<M, NWAVES, 0>::kLdsBytes) + static_cast<size_t>(a.bias_floats) * 4;
  if (lds > 160 * 1024) {
    ns::set_error("ns_depthnet_forward: %zu bytes of LDS needed (too many layers for the resident bias image)", lds);
    return NS_E_UNSUPPORTED;
  }
  auto kern = depthnet_kernel<M, NB, NWAVES, PRECISE>;
  NS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                             static_cast<int>(lds)));
  const int64_t n_tiles = (a.R + 31) / 32;
  const int64_t n_groups = (n_tiles + NWAVES - 1) / NWAVES;
  int cus = ns::cu_count();
  if (cus <= 0) cus = 256;
  if (cus > kDepthnetMaxGrid) cus = kDepthnetMaxGrid;
  const int grid = static_cast<int>(n_groups < cus ? n_groups : cus);
  const size_t need = static_cast<size_t>(grid) * NWAVES * 2 * NB * 64 * sizeof(typename M::Block);
  if (need > net->scratch_bytes) {
    ns::set_error("ns_depthnet_forward: stash too small (%zu > %zu)", need, net->scratch_bytes);
    return NS_E_INVALID;
  }
  kern<<<grid, NWAVES * 64, lds, stream>>>(a);
  NS_LAUNCH_CHECK();
  return NS_OK;
}

}  // namespace

extern "C" {

int ns_depthnet_forward(const ns_weights* net, const float* o_dev, const float* d_dev, int64_t R,
                        float near_, float far_, float sphere_radius, float* z_dev, void* stream) {
  NS_REQUIRE(net && net->kind == NS_KIND_DEPTHNET, "not a DepthNet weight handle");
  NS_REQUIRE(R >= 0, "bad shape");
  if (R == 0) return NS_OK;
  NS_REQUIRE(o_dev && d_dev && z_dev, "null pointer");
  const int NB = net->width / 32;
  const int cpb = net->dtype == NS_DTYPE_F32 ? 4 : 2;
  if (depthnet_program_slabs(cpb, NB, net->depth) != static_cast<int>(net->n_slabs)) {
    ns::set_error("ns_depthnet_forward: packed stream has %u slabs, kernel program expects %d", net->n_slabs,
                  depthnet_program_slabs(cpb, NB, net->depth));
    return NS_E_INVALID;
  }
  DepthArgs a{};
  a.stream = static_cast<const char*>(net->stream_dev);
  a.bias = net->bias_dev; a.n_slabs = net->n_slabs; a.bias_floats = net->bias_floats;
  a.n_layers = net->depth; a.o = o_dev; a.d = d_dev; a.R = R;
  a.near_ = near_; a.far_ = far_; a.radius = sphere_radius; a.z = z_dev;
  a.scratch = static_cast<char*>(net->scratch_dev);
  hipStream_t s = ns::as_stream(stream);
  switch (net->dtype) {
    case NS_DTYPE_F32:
      return NB == 8 ? launch<MmaF32, 8, 4, true>(net, a, s) : launch<MmaF32, 4, 4, true>(net, a, s);
    case NS_DTYPE_BF16:
      return NB == 8 ? launch<MmaBF16, 8, NS_DN_WAVES, false>(net, a, s) : launch<MmaBF16, 4, 8, false>(net, a, s);
    case NS_DTYPE_F16:
      return NB == 8 ? launch<MmaF16, 8, NS_DN_WAVES, false>(net, a, s) : launch<MmaF16, 4, 8, false>(net, a, s);
  }
  return NS_E_UNSUPPORTED;
}

}  // extern "C"
