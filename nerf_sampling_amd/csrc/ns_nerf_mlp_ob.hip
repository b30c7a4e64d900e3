// Radiance-field MLP forward for the 16-bit operand paths (bf16 / f16), output-block-major engine
// (layer_ob in ns_mlp_engine.h; stream layout 1 of ns_pack.hip).  Same operator as ns_nerf_mlp.hip
// (run_network + NeRF.forward, Trainer.py:789-806 and run_nerf_helpers.py:67-134): positional encoding of
// points and view directions, DxW trunk with the input skip, sigma head, feature/view/rgb head, one
// persistent kernel, T tiles of 32 samples per wave.
#include "ns_common.h"
#include "ns_mlp_engine.h"
#include "ns_weights.h"

namespace {

using namespace nsmlp;

#ifndef NS_OB_TILES
#define NS_OB_TILES 2
#endif
#ifndef NS_OB_DEPTH
#define NS_OB_DEPTH 4
#endif
#ifndef NS_OB_WAVES
#define NS_OB_WAVES 4
#endif

struct NerfObArgs {
  const char* stream;
  const float* bias;
  uint32_t n_slabs;
  int bias_floats;
  int D, skip;
  const float* pts;
  const float* o;
  const float* d;
  const float* z;
  const float* viewdirs;
  const float* x90;
  int64_t S;
  int N;
  float* raw;
};

template <class M, int NB, int NWAVES, int T, int G, bool EMBEDDED>
__global__ void __launch_bounds__(NWAVES * 64)
nerf_mlp_ob_kernel(NerfObArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using Block = typename M::Block;
  using PipeT = Pipe<M, NWAVES, 0, NS_OB_DEPTH>;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5;

  float* bias_lds = reinterpret_cast<float*>(smem + PipeT::kLdsBytes);
  for (int i = threadIdx.x; i < a.bias_floats; i += NWAVES * 64) bias_lds[i] = a.bias[i];
  __syncthreads();

  // per-wave LDS stash for the embeddings (point: blocks 0,1; view direction: block 2): they are needed again only
  // at the skip layer and the views layer, and 24 registers per tile matter more than 4 LDS reads per use
  typedef typename M::AFrag __attribute__((address_space(3))) * StashPtr;
  const uint32_t stash_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(NS_LDS_PTR(smem))) + PipeT::kLdsBytes +
                              ((static_cast<uint32_t>(a.bias_floats) * 4u + 15u) & ~15u) +
                              static_cast<uint32_t>(wave) * (T * 3 * 2048) + static_cast<uint32_t>(lane) * 16u;
  auto stash_at = [&](int t, int b, int sub) -> StashPtr {
    return reinterpret_cast<StashPtr>(static_cast<uintptr_t>(stash_base + ((t * 3 + b) * 2 + sub) * 1024));
  };
  auto stash_put = [&](int t, int b, const Block& v) { *stash_at(t, b, 0) = v.v[0]; *stash_at(t, b, 1) = v.v[1]; };
  auto stash_get = [&](int t, int b) -> Block { Block v; v.v[0] = *stash_at(t, b, 0); v.v[1] = *stash_at(t, b, 1); return v; };

  PipeT ring;
  ring.init(a.stream, smem, a.n_slabs, wave, lane);

  // Inputs of the NEXT group are fetched right after layer 0 of the current one, by LDS-DMA into a per-wave
  // staging area (no registers held across the network), so the dependent global loads (sample -> ray -> o, d,
  // viewdirs) are never waited for at a tile boundary: a lone wave per SIMD has nobody to hide them behind
  // (ablation without the loads: -6 % kernel time).  Staging layout per wave: [T][10 values][64 lanes] floats;
  // pts mode: p[0..2], v at 7..9;  (o, d, z) mode: o 0..2, d 3..5, z 6, v 7..9.
  const uint32_t stage_base = stash_base - static_cast<uint32_t>(lane) * 16u - static_cast<uint32_t>(wave) * (T * 3 * 2048) +
                              NWAVES * (T * 3 * 2048) + static_cast<uint32_t>(wave) * (T * 10 * 256);
  auto sample_of = [&](int64_t g, int t, bool& valid) -> int64_t {
    const int64_t tile = (g * NWAVES + wave) * T + t;
    int64_t sidx = tile * 32 + (lane & 31);
    valid = sidx < a.S;
    return valid ? sidx : a.S - 1;   // clamp: compute on a real sample, mask the store
  };
  auto prefetch = [&](int64_t g) {
    if constexpr (!EMBEDDED) {
      static_for<T>([&](auto t_) {
        constexpr int t = decltype(t_)::value;
        bool valid;
        const int64_t sidx = sample_of(g, t, valid);
        const int64_t ray = a.S <= 0x7fffffff ? static_cast<int64_t>(static_cast<uint32_t>(sidx) / static_cast<uint32_t>(a.N))
                                              : sidx / a.N;
        auto put = [&](int slot, const float* src) {
          lds_dma4(src, stage_base + (t * 10 + slot) * 256);
        };
        if (a.pts) {
#pragma unroll
          for (int c = 0; c < 3; ++c) put(c, a.pts + sidx * 3 + c);
        } else {
#pragma unroll
          for (int c = 0; c < 3; ++c) { put(c, a.o + ray * 3 + c); put(3 + c, a.d + ray * 3 + c); }
          put(6, a.z + sidx);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) put(7 + c, a.viewdirs + ray * 3 + c);
      });
    }
  };
  auto staged = [&](int t, int slot) -> float {
    return *reinterpret_cast<const float __attribute__((address_space(3)))*>(
        static_cast<uintptr_t>(stage_base + (t * 10 + slot) * 256 + lane * 4));
  };

  const int64_t n_tiles = (a.S + 31) / 32;
  const int64_t n_groups = (n_tiles + NWAVES * T - 1) / (NWAVES * T);
  prefetch(blockIdx.x);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
    Block xe[T][2];   // embedded point (63 -> 64 virtual features); registers for layer 0 only
    asm volatile("" ::: "memory");   // the staged inputs landed several slab steps ago (in-order vmcnt)
    static_for<T>([&](auto t_) {
      Block ve[1];    // embedded view direction (27 -> 32)
      constexpr int t = decltype(t_)::value;
      if constexpr (EMBEDDED) {
        bool valid;
        const float* row = a.x90 + sample_of(g, t, valid) * 90;
        gather3<M, 10, 2>(xe[t], row, h);
        gather3<M, 4, 1>(ve, row + 63, h);
      } else {
        float p[3], v[3];
        if (a.pts) {
#pragma unroll
          for (int c = 0; c < 3; ++c) p[c] = staged(t, c);
        } else {
          const float zz = staged(t, 6);
#pragma unroll
          for (int c = 0; c < 3; ++c) p[c] = staged(t, c) + staged(t, 3 + c) * zz;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = staged(t, 7 + c);
        embed3<M, false, 10, 2>(xe[t], p, h);
        embed3<M, false, 4, 1>(ve, v, h);
      }
      stash_put(t, 0, xe[t][0]); stash_put(t, 1, xe[t][1]); stash_put(t, 2, ve[0]);
    });

    const float* bias = bias_lds;
    Block hA[T][NB], hB[T][NB];
    f32x16 last[G][T], last1[1][T];
    auto in_x = [&](auto t_, auto kb_) -> const Block& { return xe[decltype(t_)::value][decltype(kb_)::value]; };
    auto in_A = [&](auto t_, auto kb_) -> const Block& { return hA[decltype(t_)::value][decltype(kb_)::value]; };
    auto in_B = [&](auto t_, auto kb_) -> const Block& { return hB[decltype(t_)::value][decltype(kb_)::value]; };
    auto in_xA = [&](auto t_, auto kb_) -> Block {
      constexpr int kb = decltype(kb_)::value;
      if constexpr (kb < 2) return stash_get(decltype(t_)::value, kb); else return hA[decltype(t_)::value][kb - 2];
    };
    auto in_xB = [&](auto t_, auto kb_) -> Block {
      constexpr int kb = decltype(kb_)::value;
      if constexpr (kb < 2) return stash_get(decltype(t_)::value, kb); else return hB[decltype(t_)::value][kb - 2];
    };
    auto finish_A = [&] { convert_last<M, true, T, G, NB>(hA, last); };
    auto finish_B = [&] { convert_last<M, true, T, G, NB>(hB, last); };

    // layer 0: x -> hA
    layer_ob<M, T, G, NB, 2, true>(ring, bias, h, hA, last, in_x); finish_A(); bias += NB * 32;
    // next group's inputs (clamped to the last sample past the end: loaded, never used); the reads of this group's
    // staged values are complete (their results fed the embeddings above)
    prefetch(g + gridDim.x);
    int l = 1;
    // layers 1 .. D-1, two per trip (hA -> hB -> hA); the layer after `skip` sees cat[x, h]
    for (; l + 1 < a.D; l += 2) {
      if (l - 1 == a.skip) layer_ob<M, T, G, NB, NB + 2, true>(ring, bias, h, hB, last, in_xA);
      else layer_ob<M, T, G, NB, NB, true>(ring, bias, h, hB, last, in_A);
      finish_B(); bias += NB * 32;
      if (l == a.skip) layer_ob<M, T, G, NB, NB + 2, true>(ring, bias, h, hA, last, in_xB);
      else layer_ob<M, T, G, NB, NB, true>(ring, bias, h, hA, last, in_B);
      finish_A(); bias += NB * 32;
    }
    if (l < a.D) {  // odd layer left over: hA -> hB, then move back
      if (l - 1 == a.skip) layer_ob<M, T, G, NB, NB + 2, true>(ring, bias, h, hB, last, in_xA);
      else layer_ob<M, T, G, NB, NB, true>(ring, bias, h, hB, last, in_A);
      finish_B(); bias += NB * 32;
      static_for<T>([&](auto t_) { static_for<NB>([&](auto b_) { hA[decltype(t_)::value][decltype(b_)::value] = hB[decltype(t_)::value][decltype(b_)::value]; }); });
    }
    // sigma head (W -> 1): row 0 of a 32-row block
    float sigma[T];
    layer_ob<M, T, 1, 1, NB, false>(ring, bias, h, hB, last1, in_A); bias += 32;
    static_for<T>([&](auto t_) { sigma[decltype(t_)::value] = last1[0][decltype(t_)::value][0]; });
    // feature (W -> W, no activation): hA -> hB
    layer_ob<M, T, G, NB, NB, false>(ring, bias, h, hB, last, in_A); bias += NB * 32;
    convert_last<M, false, T, G, NB>(hB, last);
    // views: cat[feature, dirs27] -> W/2, relu: (hB, ve) -> hA[0 .. NB/2)
    auto in_Bv = [&](auto t_, auto kb_) -> Block {
      constexpr int kb = decltype(kb_)::value;
      if constexpr (kb < NB) return hB[decltype(t_)::value][kb]; else return stash_get(decltype(t_)::value, 2);
    };
    layer_ob<M, T, G, NB / 2, NB + 1, true>(ring, bias, h, hA, last, in_Bv); bias += (NB / 2) * 32;
    convert_last<M, true, T, G, NB / 2>(hA, last);
    // rgb (W/2 -> 3): rows 0..2
    layer_ob<M, T, 1, 1, NB / 2, false>(ring, bias, h, hB, last1, in_A);

    if (h == 0) {
      static_for<T>([&](auto t_) {
        constexpr int t = decltype(t_)::value;
        bool valid;
        const int64_t sidx = sample_of(g, t, valid);
        if (valid) reinterpret_cast<float4*>(a.raw)[sidx] = make_float4(last1[0][t][0], last1[0][t][1], last1[0][t][2], sigma[t]);
      });
    }
  }
  ring.finish();
}

int ob_program_slabs(int NB, int D, int skip) {
  const int cpb = 2;
  int n = ob_layer_slabs(cpb, NB, 2);
  for (int l = 1; l < D; ++l) n += ob_layer_slabs(cpb, NB, (l - 1 == skip) ? NB + 2 : NB);
  n += ob_layer_slabs(cpb, 1, NB) + ob_layer_slabs(cpb, NB, NB) + ob_layer_slabs(cpb, NB / 2, NB + 1) +
       ob_layer_slabs(cpb, 1, NB / 2);
  return n;
}

template <class M, int NB, int NWAVES, int T, int G, bool EMB>
int launch(NerfObArgs& a, hipStream_t stream) {
  const size_t lds = static_cast<size_t>(Pipe<M, NWAVES, 0, NS_OB_DEPTH>::kLdsBytes) +
                     ((static_cast<size_t>(a.bias_floats) * 4 + 15) & ~size_t(15)) + static_cast<size_t>(NWAVES) * T * 3 * 2048 +
                     static_cast<size_t>(NWAVES) * T * 10 * 256;   // ring | bias | embedding stash | input staging
  auto kern = nerf_mlp_ob_kernel<M, NB, NWAVES, T, G, EMB>;
  static bool attr_set = false;
  if (!attr_set) {
    NS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                               static_cast<int>(lds)));
    attr_set = true;
  }
  const int64_t n_tiles = (a.S + 31) / 32;
  const int64_t n_groups = (n_tiles + NWAVES * T - 1) / (NWAVES * T);
  int cus = ns::cu_count();
  if (cus <= 0) cus = 256;
  const int grid = static_cast<int>(n_groups < cus ? n_groups : cus);
  kern<<<grid, NWAVES * 64, lds, stream>>>(a);
  NS_LAUNCH_CHECK();
  return NS_OK;
}

template <class M, bool EMB>
int dispatch_m(const ns_weights* net, NerfObArgs& a, hipStream_t stream) {
  constexpr int T = NS_OB_TILES, NW = NS_OB_WAVES, G = kObGroup;
  if (net->layout != G) {
    ns::set_error("ns_nerf_forward: weight stream packed for %d output blocks in flight, kernel built for %d", net->layout, G);
    return NS_E_INVALID;
  }
#ifdef NS_OB_ONLY_NB8   // tuning builds whose fragment depth does not divide the W=128 program
  if (net->width != 256) return NS_E_UNSUPPORTED;
  return launch<M, 8, NW, T, G, EMB>(a, stream);
#else
  return net->width == 256 ? launch<M, 8, NW, T, G, EMB>(a, stream) : launch<M, 4, NW, T, G, EMB>(a, stream);
#endif
}

}  // namespace

// called by ns_nerf_forward / ns_nerf_forward_embedded for handles packed with layout 1 (arguments validated there)
int ns_nerf_forward_ob(const ns_weights* net, const float* pts_dev, const float* o_dev, const float* d_dev,
                       const float* z_dev, const float* viewdirs_dev, const float* x90_dev, int64_t S, int N,
                       float* raw_dev, hipStream_t stream) {
  const int NB = net->width / 32;
  if (ob_program_slabs(NB, net->depth, net->skip) != static_cast<int>(net->n_slabs)) {
    ns::set_error("ns_nerf_forward: packed stream has %u slabs, kernel program expects %d", net->n_slabs,
                  ob_program_slabs(NB, net->depth, net->skip));
    return NS_E_INVALID;
  }
  NerfObArgs a{};
  a.stream = static_cast<const char*>(net->stream_dev);
  a.bias = net->bias_dev; a.n_slabs = net->n_slabs; a.bias_floats = net->bias_floats;
  a.D = net->depth; a.skip = net->skip;
  a.pts = pts_dev; a.o = o_dev; a.d = d_dev; a.z = z_dev; a.viewdirs = viewdirs_dev; a.x90 = x90_dev;
  a.S = S; a.N = N; a.raw = raw_dev;
  const bool emb = x90_dev != nullptr;
  if (net->dtype == NS_DTYPE_BF16) return emb ? dispatch_m<MmaBF16, true>(net, a, stream) : dispatch_m<MmaBF16, false>(net, a, stream);
  if (net->dtype == NS_DTYPE_F16) return emb ? dispatch_m<MmaF16, true>(net, a, stream) : dispatch_m<MmaF16, false>(net, a, stream);
  return NS_E_UNSUPPORTED;
}
