// Radiance-field MLP forward (run_network + NeRF.forward, Trainer.py:789-806 and
// run_nerf_helpers.py:67-134) as ONE persistent MFMA kernel: positional encoding of points and
// view directions, 8x256 trunk with the input skip, sigma head, (feature o view) / rgb head.
// Per 32-sample tile nothing but 12 B of point, 12 B of direction and 16 B of output touches HBM;
// the 1.2 MB (bf16) weight stream is re-read from L2 by every workgroup through the LDS ring.
#include "ns_common.h"
#include "ns_mlp_engine.h"
#include "ns_weights.h"

namespace {

using namespace nsmlp;


struct NerfArgs {
  const char* stream;
  const float* bias;
  uint32_t n_slabs;
  int bias_floats;
  int D;
  uint32_t skip_mask;     // bit i: layer i + 1 sees cat[x, h]
  int use_viewdirs, out_ch, x_stride;   // x_stride: row length of the pre-embedded input (90, or 63 without view directions)
  // inputs: either pts [S,3] or (o,d [R,3], z [S]); or x [S,90] pre-embedded
  const float* pts;
  const float* o;
  const float* d;
  const float* z;
  const float* viewdirs;  // [R,3]
  const float* x90;
  int64_t S;              // total samples R*N
  int N;                  // samples per ray
  float* raw;             // [S,4]
};

template <class M, int NB, int NWAVES, int LAG, bool PRECISE_TRIG, bool EMBEDDED>
__global__ void __launch_bounds__(NWAVES * 64)
nerf_mlp_kernel(NerfArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using Block = typename M::Block;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5;

  using PipeT = Pipe<M, NWAVES, LAG>;
  // LDS image: [ring: PipeT::RING x 16 KiB][bias floats]
  float* bias_lds = reinterpret_cast<float*>(smem + PipeT::kLdsBytes);
  for (int i = threadIdx.x; i < a.bias_floats; i += NWAVES * 64) bias_lds[i] = a.bias[i];
  __syncthreads();

  PipeT ring;
  ring.init(a.stream, smem, a.n_slabs, wave, lane);

  const int64_t n_tiles = (a.S + 31) / 32;
  const int64_t n_groups = (n_tiles + NWAVES - 1) / NWAVES;
  for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const int64_t tile = g * NWAVES + wave;
    int64_t s = tile * 32 + (lane & 31);
    const bool valid = s < a.S;
    if (!valid) s = a.S - 1;  // clamp: compute on a real sample, mask the store

    Block xe[2];  // embedded point (63 -> 64 virtual features)
    Block ve[1];  // embedded view direction (27 -> 32)
    if constexpr (EMBEDDED) {
      const float* row = a.x90 + s * a.x_stride;
      gather3<M, 10, 2>(xe, row, h);
      if (a.use_viewdirs) gather3<M, 4, 1>(ve, row + 63, h);
    } else {
      const int64_t ray = s / a.N;
      float p[3], v[3];
      if (a.pts) {
#pragma unroll
        for (int c = 0; c < 3; ++c) p[c] = a.pts[s * 3 + c];
      } else {
        const float zz = a.z[s];
#pragma unroll
        for (int c = 0; c < 3; ++c) p[c] = a.o[ray * 3 + c] + a.d[ray * 3 + c] * zz;
      }
      embed3<M, PRECISE_TRIG, 10, 2>(xe, p, h);
      if (a.use_viewdirs) {
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = a.viewdirs[ray * 3 + c];
        embed3<M, PRECISE_TRIG, 4, 1>(ve, v, h);
      }
    }

    const float* bias = bias_lds;
    f32x16 acc[NB];
    Block hcur[NB];
    // layer 0
    init_bias<NB>(acc, bias, h); bias += NB * 32;
    consume<M, NB, 2>(ring, acc, xe);
    to_blocks<M, kRelu, NB>(hcur, acc);
    // layers 1 .. D-1 (the layer after `skip` sees cat[x, h])
    for (int l = 1; l < a.D; ++l) {
      init_bias<NB>(acc, bias, h); bias += NB * 32;
      if ((a.skip_mask >> (l - 1)) & 1u) consume<M, NB, 2>(ring, acc, xe);
      consume<M, NB, NB>(ring, acc, hcur);
      to_blocks<M, kRelu, NB>(hcur, acc);
    }
    f32x16 acc1[1];
    if (!a.use_viewdirs) {
      // output_linear (W -> out_ch, no activation, run_nerf_helpers.py:132-133): rows 0 .. out_ch-1 of one 32-row block;
      // row (q & 3) + 8 (q >> 2) + 4 h sits in register q of lane half h
      init_bias<1>(acc1, bias, h);
      consume<M, 1, NB>(ring, acc1, hcur);
      if (valid) {
        static_for<16>([&](auto q_) {
          constexpr int q = decltype(q_)::value;
          const int row = (q & 3) + 8 * (q >> 2) + 4 * h;
          if (row < a.out_ch) a.raw[s * a.out_ch + row] = acc1[0][q];
        });
      }
      continue;
    }
    // sigma head (W -> 1): row 0 of a 32-row block
    init_bias<1>(acc1, bias, h); bias += 32;
    consume<M, 1, NB>(ring, acc1, hcur);
    const float sigma = acc1[0][0];
    // views o feature: cat[h, dirs27] -> W/2, relu (feature_linear has no activation and is folded into
    // views_linears[0] at pack time, ns_pack.hip)
    f32x16 accv[NB / 2];
    Block hv[NB / 2];
    init_bias<NB / 2>(accv, bias, h); bias += (NB / 2) * 32;
    consume<M, NB / 2, NB>(ring, accv, hcur);
    consume<M, NB / 2, 1>(ring, accv, ve);
    to_blocks<M, kRelu, NB / 2>(hv, accv);
    // rgb (W/2 -> 3): rows 0..2
    init_bias<1>(acc1, bias, h);
    consume<M, 1, NB / 2>(ring, acc1, hv);

    if (valid && h == 0) {
      float4 o4 = make_float4(acc1[0][0], acc1[0][1], acc1[0][2], sigma);
      reinterpret_cast<float4*>(a.raw)[s] = o4;
    }
  }
  ring.finish();
}

// number of slabs one pass of the program consumes (must equal ns_weights::n_slabs)
int nerf_program_slabs(int cpb, int NB, int D, uint32_t skip_mask, int use_viewdirs) {
  int n = seg_slabs(cpb, NB, 2);
  for (int l = 1; l < D; ++l) {
    if ((skip_mask >> (l - 1)) & 1u) n += seg_slabs(cpb, NB, 2);
    n += seg_slabs(cpb, NB, NB);
  }
  if (use_viewdirs) n += seg_slabs(cpb, 1, NB) + seg_slabs(cpb, NB / 2, NB) + seg_slabs(cpb, NB / 2, 1) + seg_slabs(cpb, 1, NB / 2);
  else n += seg_slabs(cpb, 1, NB);
  return n;
}

template <class M, int NB, int NWAVES, int LAG, bool PRECISE, bool EMB>
int launch(const ns_weights* net, NerfArgs& a, hipStream_t stream) {
  const size_t lds = static_cast<size_t>(Pipe<M, NWAVES, LAG>::kLdsBytes) + static_cast<size_t>(a.bias_floats) * 4;
  auto kern = nerf_mlp_kernel<M, NB, NWAVES, LAG, PRECISE, EMB>;
  NS_HIP(ns::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds));
  const int64_t n_tiles = (a.S + 31) / 32;
  const int64_t n_groups = (n_tiles + NWAVES - 1) / NWAVES;
  int cus = ns::cu_count();
  if (cus <= 0) cus = 256;
  const int grid = static_cast<int>(n_groups < cus ? n_groups : cus);
  kern<<<grid, NWAVES * 64, lds, stream>>>(a);
  NS_LAUNCH_CHECK();
  (void)net;
  return NS_OK;
}

template <bool EMB>
int dispatch(const ns_weights* net, NerfArgs& a, hipStream_t stream) {
  const int NB = net->width / 32;
  const int cpb = net->dtype == NS_DTYPE_F32 ? 4 : 2;
  if (nerf_program_slabs(cpb, NB, net->depth, net->skip_mask, net->use_viewdirs) != static_cast<int>(net->n_slabs)) {
    ns::set_error("ns_nerf_forward: packed stream has %u slabs, kernel program expects %d", net->n_slabs,
                  nerf_program_slabs(cpb, NB, net->depth, net->skip_mask, net->use_viewdirs));
    return NS_E_INVALID;
  }
  switch (net->dtype) {
    case NS_DTYPE_F32:
      return NB == 8 ? launch<MmaF32, 8, 4, 0, true, EMB>(net, a, stream) : launch<MmaF32, 4, 4, 0, true, EMB>(net, a, stream);
    default:   // 16-bit handles are packed for, and run in, ns_nerf_mlp_ob16.hip
      break;
  }
  return NS_E_UNSUPPORTED;
}

}  // namespace

int ns_nerf_forward_ob16(const ns_weights* net, const float* pts_dev, const float* o_dev, const float* d_dev,
                         const float* z_dev, const float* viewdirs_dev, const float* x90_dev, int64_t S, int N,
                         float* raw_dev, hipStream_t stream, const ns_composite_args* comp);

extern "C" {

int ns_nerf_out_channels(const ns_weights* net) {
  return (net && net->kind == NS_KIND_NERF) ? net->out_ch : 0;
}

int ns_nerf_forward(const ns_weights* net, const float* pts_dev, const float* o_dev,
                    const float* d_dev, const float* z_dev, const float* viewdirs_dev, int64_t R,
                    int N, float* raw_dev, void* stream) {
  NS_REQUIRE(net && net->kind == NS_KIND_NERF, "not a NeRF weight handle");
  NS_REQUIRE(R >= 0 && N >= 0, "bad shape");
  if (R == 0 || N == 0) return NS_OK;
  NS_REQUIRE(raw_dev && (viewdirs_dev || !net->use_viewdirs), "null pointer");
  NS_REQUIRE(pts_dev || (o_dev && d_dev && z_dev), "need pts or (o, d, z)");
  NS_REQUIRE(net->out_ch != 4 || (reinterpret_cast<uintptr_t>(raw_dev) & 15) == 0, "raw must be 16-byte aligned");
  if (net->layout == 16)
    return ns_nerf_forward_ob16(net, pts_dev, o_dev, d_dev, z_dev, viewdirs_dev, nullptr, R * N, N, raw_dev,
                                ns::as_stream(stream), nullptr);
  NerfArgs a{};
  a.stream = static_cast<const char*>(net->stream_dev);
  a.bias = net->bias_dev; a.n_slabs = net->n_slabs; a.bias_floats = net->bias_floats;
  a.D = net->depth; a.skip_mask = net->skip_mask; a.use_viewdirs = net->use_viewdirs; a.out_ch = net->out_ch;
  a.pts = pts_dev; a.o = o_dev; a.d = d_dev; a.z = z_dev; a.viewdirs = viewdirs_dev; a.x90 = nullptr;
  a.S = R * N; a.N = N; a.raw = raw_dev;
  return dispatch<false>(net, a, ns::as_stream(stream));
}

int ns_nerf_forward_embedded(const ns_weights* net, const float* x_dev, int64_t M, float* raw_dev,
                             void* stream) {
  NS_REQUIRE(net && net->kind == NS_KIND_NERF, "not a NeRF weight handle");
  NS_REQUIRE(M >= 0, "bad shape");
  if (M == 0) return NS_OK;
  NS_REQUIRE(x_dev && raw_dev, "null pointer");
  NS_REQUIRE(net->out_ch != 4 || (reinterpret_cast<uintptr_t>(raw_dev) & 15) == 0, "raw must be 16-byte aligned");
  if (net->layout == 16)
    return ns_nerf_forward_ob16(net, nullptr, nullptr, nullptr, nullptr, nullptr, x_dev, M, 1, raw_dev,
                                ns::as_stream(stream), nullptr);
  NerfArgs a{};
  a.stream = static_cast<const char*>(net->stream_dev);
  a.bias = net->bias_dev; a.n_slabs = net->n_slabs; a.bias_floats = net->bias_floats;
  a.D = net->depth; a.skip_mask = net->skip_mask; a.use_viewdirs = net->use_viewdirs; a.out_ch = net->out_ch;
  a.x_stride = net->use_viewdirs ? 90 : 63;
  a.x90 = x_dev; a.S = M; a.N = 1; a.raw = raw_dev;
  return dispatch<true>(net, a, ns::as_stream(stream));
}

}  // extern "C"
