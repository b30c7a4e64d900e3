// DepthNet forward (depth_net.py:117-169), fp32 parity path, as one persistent MFMA kernel: ray-sphere
// intersection, the three positional encodings, the FOLDED front end (the three affine skip branches -- the reference
// never applies its LeakyReLU there, depth_net.py:140,148,156 -- composed with the first trunk layer at pack time,
// ns_pack.hip), the LeakyReLU trunk and the sigmoid head.  One wave owns 32 rays; HBM traffic per ray is 24 B in,
// 4 B out, nothing else.  The 16-bit paths run ns_depthnet_ob16.hip (v_mfma_f32_16x16x32).
#include "ns_common.h"
#include "ns_mlp_engine.h"
#include "ns_weights.h"

namespace {

using namespace nsmlp;

struct DepthArgs {
  const char* stream;
  const float* bias;
  uint32_t n_slabs;
  int bias_floats;
  int n_layers;   // trunk layers (layer 0 is the folded 252 -> W one)
  const float* o;
  const float* d;
  int64_t R;
  float near_, far_, radius;
  float* z;
};

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
      // folded layer 0 on cat[e_o, e_d, e_x]: 2 + 2 + 4 input blocks (252 -> 256 virtual features)
      Block e[8];
      {
        Block e3[2];
        embed3<M, PRECISE_TRIG, 10, 2>(e3, o, h);
        e[0] = e3[0]; e[1] = e3[1];
        embed3<M, PRECISE_TRIG, 10, 2>(e3, d, h);
        e[2] = e3[0]; e[3] = e3[1];
        Block e6[4];
        embed6<M, PRECISE_TRIG>(e6, x6, h);
        e[4] = e6[0]; e[5] = e6[1]; e[6] = e6[2]; e[7] = e6[3];
      }
      init_bias<NB>(acc, bias, h); bias += NB * 32;
      consume<M, NB, 8>(ring, acc, e);
      to_blocks<M, kLeaky, NB>(hcur, acc);
    }
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
  return seg_slabs(cpb, NB, 8) + (n - 1) * seg_slabs(cpb, NB, NB) + seg_slabs(cpb, 1, NB);
}

template <class M, int NB, int NWAVES, bool PRECISE>
int launch(DepthArgs& a, hipStream_t stream) {
  const size_t lds = static_cast<size_t>(Pipe<M, NWAVES, 0>::kLdsBytes) + static_cast<size_t>(a.bias_floats) * 4;
  if (lds > 160 * 1024) {
    ns::set_error("ns_depthnet_forward: %zu bytes of LDS needed (too many layers for the resident bias image)", lds);
    return NS_E_UNSUPPORTED;
  }
  auto kern = depthnet_kernel<M, NB, NWAVES, PRECISE>;
  NS_HIP(ns::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds));
  const int64_t n_tiles = (a.R + 31) / 32;
  const int64_t n_groups = (n_tiles + NWAVES - 1) / NWAVES;
  int cus = ns::cu_count();
  if (cus <= 0) cus = 256;
  const int grid = static_cast<int>(n_groups < cus ? n_groups : cus);
  kern<<<grid, NWAVES * 64, lds, stream>>>(a);
  NS_LAUNCH_CHECK();
  return NS_OK;
}

}  // namespace

int ns_depthnet_forward_ob16(const ns_weights* net, const float* o_dev, const float* d_dev, int64_t R, float near_,
                             float far_, float sphere_radius, float* z_dev, hipStream_t stream);

extern "C" {

int ns_depthnet_forward(const ns_weights* net, const float* o_dev, const float* d_dev, int64_t R,
                        float near_, float far_, float sphere_radius, float* z_dev, void* stream) {
  NS_REQUIRE(net && net->kind == NS_KIND_DEPTHNET, "not a DepthNet weight handle");
  NS_REQUIRE(R >= 0, "bad shape");
  if (R == 0) return NS_OK;
  NS_REQUIRE(o_dev && d_dev && z_dev, "null pointer");
  hipStream_t s = ns::as_stream(stream);
  if (net->layout == 16) return ns_depthnet_forward_ob16(net, o_dev, d_dev, R, near_, far_, sphere_radius, z_dev, s);
  NS_REQUIRE(net->dtype == NS_DTYPE_F32, "k-major DepthNet streams are fp32 only");
  const int NB = net->width / 32;
  if (depthnet_program_slabs(4, NB, net->depth) != static_cast<int>(net->n_slabs)) {
    ns::set_error("ns_depthnet_forward: packed stream has %u slabs, kernel program expects %d", net->n_slabs,
                  depthnet_program_slabs(4, NB, net->depth));
    return NS_E_INVALID;
  }
  DepthArgs a{};
  a.stream = static_cast<const char*>(net->stream_dev);
  a.bias = net->bias_dev; a.n_slabs = net->n_slabs; a.bias_floats = net->bias_floats;
  a.n_layers = net->depth; a.o = o_dev; a.d = d_dev; a.R = R;
  a.near_ = near_; a.far_ = far_; a.radius = sphere_radius; a.z = z_dev;
  return NB == 8 ? launch<MmaF32, 8, 4, true>(a, s) : launch<MmaF32, 4, 4, true>(a, s);
}

}  // extern "C"
