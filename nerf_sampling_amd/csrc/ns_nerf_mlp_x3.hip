// Radiance-field MLP forward with SPLIT fp16 operands ("f16x3", NS_DTYPE_F16X3): the program of ns_nerf_mlp_ob16.hip
// (same operator, same folded layers, same 16x16x32 engine and weight ring) with every operand carried as a hi + lo
// pair of fp16 values and every product term as three MFMAs (layer_ob16x3 in ns_mlp_engine.h).  fp32-grade results
// (the fp32 parity gates apply: raw <= 2e-5 of scale, config-1 frame 1e-4) at about a third of the fp16 rate, i.e.
// several times the exact-fp32 MFMA path.  A wave owns T = 2 tiles of 16 samples: the hi/lo activation pairs of two
// tiles fill the registers four plain tiles do.
#include "ns_common.h"
#include "ns_mlp_engine.h"
#include "ns_weights.h"

#include <cstdlib>

namespace {

using namespace nsmlp;

constexpr int kT = 2;        // 16-sample tiles per wave (hi + lo activation blocks: half the tiles of the plain kernel)
constexpr int kWaves = 4;    // one wave per SIMD: ~256 AGPRs of activations + accumulators per wave


}  // namespace
#ifndef NS_OB16_ASM_INC
#define NS_OB16_ASM_INC "ns_ob16_asm.inc"
#endif
#include NS_OB16_ASM_INC
namespace {
// one W = 256 hidden layer of the split-operand network as a generated statement (see ns_nerf_mlp_ob16.hip): the (hi, lo)
// blocks of set A / set V are the tuple pairs [tile][kb][half] of the register map
template <bool IN_A, bool SKIP, class PipeT>
__device__ __forceinline__ void hidden_layer_asm_x3(PipeT& ring, const float* bias_lds, int g, Mma16F16x3::Block (&hA)[2][8],
                                                    Mma16F16x3::Block (&hB)[2][8], const Mma16F16x3::Block (&xs)[2][2]) {
  u32x4 A[32], V[32], X[8];
  static_for<2>([&](auto t_) {
    constexpr int t = decltype(t_)::value;
    static_for<8>([&](auto kb_) {
      constexpr int kb = decltype(kb_)::value, k = 2 * (8 * t + kb);
      if constexpr (IN_A) { A[k] = __builtin_bit_cast(u32x4, hA[t][kb].hi); A[k + 1] = __builtin_bit_cast(u32x4, hA[t][kb].lo); }
      else { V[k] = __builtin_bit_cast(u32x4, hB[t][kb].hi); V[k + 1] = __builtin_bit_cast(u32x4, hB[t][kb].lo); }
    });
    static_for<2>([&](auto kb_) {
      constexpr int kb = decltype(kb_)::value, k = 2 * (2 * t + kb);
      X[k] = __builtin_bit_cast(u32x4, xs[t][kb].hi); X[k + 1] = __builtin_bit_cast(u32x4, xs[t][kb].lo);
    });
  });
  hidden_asm_run<Mma16F16x3, 4, IN_A, SKIP>(ring, bias_lds, g, A, V, X);
  static_for<2>([&](auto t_) {
    constexpr int t = decltype(t_)::value;
    static_for<8>([&](auto kb_) {
      constexpr int kb = decltype(kb_)::value, k = 2 * (8 * t + kb);
      if constexpr (IN_A) { hB[t][kb].hi = __builtin_bit_cast(f16x8, V[k]); hB[t][kb].lo = __builtin_bit_cast(f16x8, V[k + 1]); }
      else { hA[t][kb].hi = __builtin_bit_cast(f16x8, A[k]); hA[t][kb].lo = __builtin_bit_cast(f16x8, A[k + 1]); }
    });
  });
}

struct NerfX3Args {
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
  const float* viewdirs;
  const float* x90;
  int64_t S;
  int N;
  float* raw;
  const uint32_t* count_dev;   // NULL, or the number of samples to evaluate, read by the kernel (<= S: the selective guard pass
                               // launches for its capacity and the device knows how many rays were flagged)
};

// PROD: the production network (8 x 256, skips = [4], view directions) as straight-line code over the generated layer
// statements, as in ns_nerf_mlp_ob16.hip
template <int NKB, bool EMBEDDED, bool PROD = false>   // NKB = W / 32 K-blocks of a hidden layer
__global__ void __launch_bounds__(kWaves * 64)
nerf_mlp_x3_kernel(NerfX3Args a) {
  using M = Mma16F16x3;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int T = kT, NWAVES = kWaves, NSB = 2 * NKB;   // 16-row output sub-blocks of a hidden layer
  using Block = typename M::Block;
  using PipeT = Pipe<M, NWAVES, 0, kOb16Depth, kOb16Ahead>;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, g = lane >> 4;
  int64_t S_ = a.S;
  if (a.count_dev) {           // (workgroup-uniform: every wave reads the same word)
    const int64_t c = static_cast<int64_t>(*a.count_dev);
    if (c < S_) S_ = c;
  }
  if (S_ <= 0) return;

  // LDS: [weight ring][bias image][embedding stash: per wave T x 3 blocks x 1 KiB][input staging: per wave 10 x 256 B]
  float* bias_lds = reinterpret_cast<float*>(smem + PipeT::kLdsBytes);
  for (int i = threadIdx.x; i < a.bias_floats; i += NWAVES * 64) bias_lds[i] = a.bias[i];
  __syncthreads();

  typedef typename M::AFrag __attribute__((address_space(3))) * StashPtr;
  const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(NS_LDS_PTR(smem)));
  const uint32_t stash_region = lds0 + PipeT::kLdsBytes + ((static_cast<uint32_t>(a.bias_floats) * 4u + 15u) & ~15u);
  // per wave T x 3 blocks x (hi 1 KiB, lo 1 KiB)
  const uint32_t stash_base = stash_region + static_cast<uint32_t>(wave) * (T * 3 * 2048) + static_cast<uint32_t>(lane) * 16u;
  auto stash_at = [&](int t, int b, int half) -> StashPtr {
    return reinterpret_cast<StashPtr>(static_cast<uintptr_t>(stash_base + ((t * 3 + b) * 2 + half) * 1024));
  };
  auto stash_put = [&](int t, int b, const Block& v) { *stash_at(t, b, 0) = v.hi; *stash_at(t, b, 1) = v.lo; };
  auto stash_get = [&](int t, int b) -> Block { Block v; v.hi = *stash_at(t, b, 0); v.lo = *stash_at(t, b, 1); return v; };
  // staging: value slot k (0..9) of sample j (0..63) of this wave's group at stage_base + k * 256 + j * 4
  const uint32_t stage_base = stash_region + NWAVES * (T * 3 * 2048) + static_cast<uint32_t>(wave) * (10 * 256);

  PipeT ring;
  ring.init(a.stream, smem, a.n_slabs, wave, lane);

  const int64_t n_tiles = (S_ + 15) / 16;
  const int64_t n_groups = (n_tiles + NWAVES * T - 1) / (NWAVES * T);
  // sample held by lane `l16` (0..15) of tile t of this wave in group grp; clamped to a real sample
  auto sample_of = [&](int64_t grp, int t, int l16, bool& valid) -> int64_t {
    const int64_t sidx = ((grp * NWAVES + wave) * T + t) * 16 + l16;
    valid = sidx < S_;
    return valid ? sidx : S_ - 1;
  };
  // Inputs of the NEXT group are fetched right after layer 0 of the current one by LDS-DMA (no registers held across
  // the network): lane j of the wave fetches the ten values of the j-th of the wave's 64 consecutive samples.
  // pts mode: p 0..2, v 7..9;  (o, d, z) mode: o 0..2, d 3..5, z 6, v 7..9.
  auto prefetch = [&](int64_t grp) {
    if constexpr (!EMBEDDED) {
      bool valid;
      const int64_t sidx = sample_of(grp, (lane >> 4) % T, lane & 15, valid);   // (upper lanes re-fetch, harmlessly)
      const int64_t ray = S_ <= 0x7fffffff ? static_cast<int64_t>(static_cast<uint32_t>(sidx) / static_cast<uint32_t>(a.N))
                                            : sidx / a.N;
      auto put = [&](int slot, const float* src) {
        lds_dma4(src, stage_base + slot * 256);
      };
      if (a.pts) {
#pragma unroll
        for (int c = 0; c < 3; ++c) put(c, a.pts + sidx * 3 + c);
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) { put(c, a.o + ray * 3 + c); put(3 + c, a.d + ray * 3 + c); }
        put(6, a.z + sidx);
      }
if (a.use_viewdirs) {
#pragma unroll
  for (int c = 0; c < 3; ++c) put(7 + c, a.viewdirs + ray * 3 + c);
}
    }
  };
  auto staged = [&](int t, int slot) -> float {
    return *reinterpret_cast<const float __attribute__((address_space(3)))*>(
        static_cast<uintptr_t>(stage_base + slot * 256 + (t * 16 + n) * 4));
  };

  prefetch(blockIdx.x);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    Block xe[T][2];   // embedded point (63 -> 64 features); registers for layer 0 only
    asm volatile("" ::: "memory");   // the staged inputs landed several slab steps ago (in-order vmcnt)
    static_for<T>([&](auto t_) {
      constexpr int t = decltype(t_)::value;
      Block ve[1];    // embedded view direction (27 -> 32)
      if constexpr (EMBEDDED) {
        bool valid;
        const float* row = a.x90 + sample_of(grp, t, n, valid) * a.x_stride;
        gather3_16<M, 10, 2>(xe[t], row, g);
        if (a.use_viewdirs) gather3_16<M, 4, 1>(ve, row + 63, g);
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
        embedN_16<M, true, 3, 10, 2>(xe[t], p, g);
        if (a.use_viewdirs) {
#pragma unroll
          for (int c = 0; c < 3; ++c) v[c] = staged(t, 7 + c);
          embedN_16<M, true, 3, 4, 1>(ve, v, g);
        }
      }
      stash_put(t, 0, xe[t][0]); stash_put(t, 1, xe[t][1]);
      if (a.use_viewdirs) stash_put(t, 2, ve[0]);
    });

    const float* bias = bias_lds;
    Block hA[T][NKB], hB[T][NKB];
    f32x4a last[T];
    auto in_x = [&](auto t_, auto kb_) -> const Block& { return xe[decltype(t_)::value][decltype(kb_)::value]; };
    auto in_A = [&](auto t_, auto kb_) -> const Block& { return hA[decltype(t_)::value][decltype(kb_)::value]; };
    auto in_B = [&](auto t_, auto kb_) -> const Block& { return hB[decltype(t_)::value][decltype(kb_)::value]; };
    // the skip layer's embedded point comes back from the LDS stash once per layer (see ns_nerf_mlp_ob16.hip)
    Block xs[T][2];
    auto load_xs = [&] {
      static_for<T>([&](auto t_) {
        constexpr int t = decltype(t_)::value;
        xs[t][0] = stash_get(t, 0); xs[t][1] = stash_get(t, 1);
      });
    };
    auto in_xA = [&](auto t_, auto kb_) -> const Block& {
      constexpr int kb = decltype(kb_)::value;
      if constexpr (kb < 2) return xs[decltype(t_)::value][kb]; else return hA[decltype(t_)::value][kb - 2];
    };
    auto in_xB = [&](auto t_, auto kb_) -> const Block& {
      constexpr int kb = decltype(kb_)::value;
      if constexpr (kb < 2) return xs[decltype(t_)::value][kb]; else return hB[decltype(t_)::value][kb - 2];
    };

    // layer 0: x -> hA
    layer_ob16x3<T, NSB, 2, true>(ring, bias, g, hA, last, in_x); convert_last16x3<true, T, NSB>(hA, last); bias += NSB * 16;
    // next group's inputs (clamped to the last sample past the end: loaded, never used); this group's staged values
    // have been consumed (they fed the embeddings above)
    prefetch(grp + gridDim.x);
    int l = 1;
    if constexpr (PROD) {
      static_assert(NKB == 8 && T == 2 && NWAVES == 4, "the generated streams are W = 256, two split tiles, four waves");
      hidden_layer_asm_x3<true, false>(ring, bias, g, hA, hB, xs); bias += NSB * 16;    // 1
      hidden_layer_asm_x3<false, false>(ring, bias, g, hA, hB, xs); bias += NSB * 16;   // 2
      hidden_layer_asm_x3<true, false>(ring, bias, g, hA, hB, xs); bias += NSB * 16;    // 3
      hidden_layer_asm_x3<false, false>(ring, bias, g, hA, hB, xs); bias += NSB * 16;   // 4
      load_xs();
      hidden_layer_asm_x3<true, true>(ring, bias, g, hA, hB, xs); bias += NSB * 16;     // 5: cat[x, h]
      hidden_layer_asm_x3<false, false>(ring, bias, g, hA, hB, xs); bias += NSB * 16;   // 6
      hidden_layer_asm_x3<true, false>(ring, bias, g, hA, hB, xs); bias += NSB * 16;    // 7: the trunk's output is in hB
      l = 8;
    }
    // layers 1 .. D-1, two per trip (hA -> hB -> hA); the layer after `skip` sees cat[x, h]
    if constexpr (!PROD) {
    for (; l + 1 < a.D; l += 2) {
      if ((a.skip_mask >> (l - 1)) & 1u) { load_xs(); layer_ob16x3<T, NSB, NKB + 2, true>(ring, bias, g, hB, last, in_xA); }
      else layer_ob16x3<T, NSB, NKB, true>(ring, bias, g, hB, last, in_A);
      convert_last16x3<true, T, NSB>(hB, last); bias += NSB * 16;
      if ((a.skip_mask >> l) & 1u) { load_xs(); layer_ob16x3<T, NSB, NKB + 2, true>(ring, bias, g, hA, last, in_xB); }
      else layer_ob16x3<T, NSB, NKB, true>(ring, bias, g, hA, last, in_B);
      convert_last16x3<true, T, NSB>(hA, last); bias += NSB * 16;
    }
    if (l < a.D) {  // odd layer left over: hA -> hB, then move back
      if ((a.skip_mask >> (l - 1)) & 1u) { load_xs(); layer_ob16x3<T, NSB, NKB + 2, true>(ring, bias, g, hB, last, in_xA); }
      else layer_ob16x3<T, NSB, NKB, true>(ring, bias, g, hB, last, in_A);
      convert_last16x3<true, T, NSB>(hB, last); bias += NSB * 16;
      static_for<T>([&](auto t_) { static_for<NKB>([&](auto b_) { hA[decltype(t_)::value][decltype(b_)::value] = hB[decltype(t_)::value][decltype(b_)::value]; }); });
    }
    }
    if (!PROD && !a.use_viewdirs) {
      // output_linear (W -> out_ch, no activation, run_nerf_helpers.py:132-133): one 16-row sub-block, raw accumulators in
      // `last`: row 4 g + r sits in register r of lane group g
      layer_ob16x3<T, 1, NKB, kNone>(ring, bias, g, hB, last, in_A);
      static_for<T>([&](auto t_) {
        constexpr int t = decltype(t_)::value;
        bool valid;
        const int64_t sidx = sample_of(grp, t, n, valid);
        static_for<4>([&](auto r_) {
          constexpr int r = decltype(r_)::value;
          const int row = 4 * g + r;
          if (valid && row < a.out_ch) a.raw[sidx * a.out_ch + row] = last[t][r];
        });
      });
      continue;
    }
    // views o feature (folded at pack time: feature_linear has no activation, run_nerf_helpers.py:119-125) on
    // cat[h, dirs27] -> W/2, relu: (hA, ve) -> hB[0 .. NKB/2); alpha_linear rides along as row 0 of one extra, LAST
    // sub-block, whose raw accumulators come back in `last`: sigma = row 0 (lane group 0, register 0)
    Block vs[T];   // the embedded view direction, once for the layer
    static_for<T>([&](auto t_) { vs[decltype(t_)::value] = stash_get(decltype(t_)::value, 2); });
    auto in_Av = [&](auto t_, auto kb_) -> const Block& {
      constexpr int kb = decltype(kb_)::value;
      if constexpr (kb < NKB) return hA[decltype(t_)::value][kb]; else return vs[decltype(t_)::value];
    };
    float sigma[T];
    if constexpr (PROD) {   // the trunk ended in hB: (hB, ve) -> hA[0 .. NKB/2), then rgb from hA
      auto in_Bv = [&](auto t_, auto kb_) -> const Block& {
        constexpr int kb = decltype(kb_)::value;
        if constexpr (kb < NKB) return hB[decltype(t_)::value][kb]; else return vs[decltype(t_)::value];
      };
      layer_ob16x3<T, NSB / 2 + 1, NKB + 1, kRelu>(ring, bias, g, hA, last, in_Bv); bias += (NSB / 2 + 1) * 16;
      static_for<T>([&](auto t_) { sigma[decltype(t_)::value] = last[decltype(t_)::value][0]; });
      layer_ob16x3<T, 1, NKB / 2, kNone>(ring, bias, g, hB, last, in_A);
    } else {
    layer_ob16x3<T, NSB / 2 + 1, NKB + 1, kRelu>(ring, bias, g, hB, last, in_Av); bias += (NSB / 2 + 1) * 16;
    static_for<T>([&](auto t_) { sigma[decltype(t_)::value] = last[decltype(t_)::value][0]; });
    // rgb (W/2 -> 3): rows 0..2 (lane group 0, registers 0..2)
    layer_ob16x3<T, 1, NKB / 2, kNone>(ring, bias, g, hA, last, in_B);
    }

    if (g == 0) {
      static_for<T>([&](auto t_) {
        constexpr int t = decltype(t_)::value;
        bool valid;
        const int64_t sidx = sample_of(grp, t, n, valid);
        if (valid) reinterpret_cast<float4*>(a.raw)[sidx] = make_float4(last[t][0], last[t][1], last[t][2], sigma[t]);
      });
    }
  }
  ring.finish();
}

int x3_program_slabs(int W, int D, uint32_t skip_mask, int use_viewdirs) {   // two stream chunks (W_hi, W_lo) per K-block
  const int NSB = W / 16, NKB = W / 32, dp = kOb16Depth;
  int n = ob16_layer_slabs(NSB, 2 * 2, dp);
  for (int l = 1; l < D; ++l) n += ob16_layer_slabs(NSB, 2 * (((skip_mask >> (l - 1)) & 1u) ? NKB + 2 : NKB), dp);
  if (use_viewdirs) n += ob16_layer_slabs(NSB / 2 + 1, 2 * (NKB + 1), dp) + ob16_layer_slabs(1, 2 * (NKB / 2), dp);
  else n += ob16_layer_slabs(1, 2 * NKB, dp);
  return n;
}

template <int NKB, bool EMB, bool PROD = false>
int launch(NerfX3Args& a, hipStream_t stream) {
  using M = Mma16F16x3;
  const size_t lds = static_cast<size_t>(Pipe<M, kWaves, 0, kOb16Depth, kOb16Ahead>::kLdsBytes) +
                     ((static_cast<size_t>(a.bias_floats) * 4 + 15) & ~size_t(15)) + static_cast<size_t>(kWaves) * kT * 3 * 2048 +
                     static_cast<size_t>(kWaves) * 10 * 256;   // ring | bias | embedding stash (hi, lo) | input staging
  auto kern = nerf_mlp_x3_kernel<NKB, EMB, PROD>;
  NS_HIP(ns::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds));
  const int64_t n_tiles = (a.S + 15) / 16;
  const int64_t n_groups = (n_tiles + kWaves * kT - 1) / (kWaves * kT);
  int cus = ns::cu_count();
  if (cus <= 0) cus = 256;
  const int grid = static_cast<int>(n_groups < cus ? n_groups : cus);
  kern<<<grid, kWaves * 64, lds, stream>>>(a);
  NS_LAUNCH_CHECK();
  return NS_OK;
}

}  // namespace

// called by ns_nerf_forward_ob16 for NS_DTYPE_F16X3 handles (arguments validated by its callers)
int ns_nerf_forward_x3(const ns_weights* net, const float* pts_dev, const float* o_dev, const float* d_dev,
                       const float* z_dev, const float* viewdirs_dev, const float* x90_dev, int64_t S, int N,
                       float* raw_dev, hipStream_t stream, const uint32_t* count_dev) {
  if (x3_program_slabs(net->width, net->depth, net->skip_mask, net->use_viewdirs) != static_cast<int>(net->n_slabs)) {
    ns::set_error("ns_nerf_forward: packed stream has %u slabs, kernel program expects %d", net->n_slabs,
                  x3_program_slabs(net->width, net->depth, net->skip_mask, net->use_viewdirs));
    return NS_E_INVALID;
  }
  NerfX3Args a{};
  a.stream = static_cast<const char*>(net->stream_dev);
  a.bias = net->bias_dev; a.n_slabs = net->n_slabs; a.bias_floats = net->bias_floats;
  a.D = net->depth; a.skip_mask = net->skip_mask; a.use_viewdirs = net->use_viewdirs; a.out_ch = net->out_ch;
  a.x_stride = net->use_viewdirs ? 90 : 63;
  a.pts = pts_dev; a.o = o_dev; a.d = d_dev; a.z = z_dev; a.viewdirs = viewdirs_dev; a.x90 = x90_dev;
  a.S = S; a.N = N; a.raw = raw_dev; a.count_dev = count_dev;
  const bool emb = x90_dev != nullptr, wide = net->width == 256;
  if (wide && net->depth == 8 && net->skip_mask == (1u << 4) && net->use_viewdirs && !ns::debug_flags().generic_kernels)
    return emb ? launch<8, true, true>(a, stream) : launch<8, false, true>(a, stream);   // the production network
  if (emb) return wide ? launch<8, true>(a, stream) : launch<4, true>(a, stream);
  return wide ? launch<8, false>(a, stream) : launch<4, false>(a, stream);
}
