// DepthNet forward (depth_net.py:117-169) for the 16-bit operand paths (bf16 / f16) on the 16x16x32 engine of
// ns_mlp_engine.h -- the same engine as the radiance-field kernel (ns_nerf_mlp_ob16.hip).  After the pack-time fold
// (ns_pack.hip: the three affine skip branches and the first trunk layer are ONE 252 -> W layer) the network is a
// plain MLP:  cat[gamma(o), gamma(d), gamma(sphere intersections)] (8 K-blocks) -> W, LeakyReLU -> ... -> 1, sigmoid.
// A wave owns T = 4 tiles of 16 rays (T = 2 with split "f16x3" operands, the fp32-grade path on the same engine);
// activations never leave the register file; no global scratch, no spills.  HBM traffic per ray: 24 B in, 4 B out.
// The production shape (10 x 256 trunk, fp16 operands: what the bf16 compute dtype pairs the field with) runs its ten
// 256 x 256 LeakyReLU layers -- the folded input layer has the same 8 K-blocks -- as generated instruction streams
// (tools/gen_ob16_asm.py, act = "leaky"; PROD below), like the radiance-field kernel's hidden layers.
#include "ns_common.h"
#include "ns_mlp_engine.h"
#include "ns_weights.h"

#include "ns_ob16_asm.inc"

#include <type_traits>

namespace {

using namespace nsmlp;

constexpr int kWaves = 4;    // one wave per SIMD
constexpr int kInKB = 8;     // K-blocks of the folded input layer: e_o (2), e_d (2), e_x (4)

// Engine policies: plain 16-bit operands (T = 4 tiles of 16 rays per wave) or split fp16 operands ("f16x3", T = 2:
// every activation block is a hi/lo pair, so half the tiles fit the register file).
template <class M_>
struct Plain16 {
  using M = M_;
  static constexpr int T = 4;
  static constexpr bool kPreciseTrig = false;
  template <int NSB, int NKB, int ACT, class PipeT, class OutT, class InF>
  __device__ static __forceinline__ void layer(PipeT& pipe, const float* bias, int g, OutT& out, f32x4a (&last)[T], InF&& in) {
    layer_ob16<M, T, NSB, NKB, ACT>(pipe, bias, g, out, last, static_cast<InF&&>(in));
  }
  template <int ACT, int NSB, class OutT>
  __device__ static __forceinline__ void convert_last(OutT& out, const f32x4a (&last)[T]) { convert_last16<M, ACT, T, NSB>(out, last); }
};
struct Split16 {
  using M = Mma16F16x3;
  static constexpr int T = 2;
  static constexpr bool kPreciseTrig = true;     // the polynomial sine of the fp32 path (v_sin_f32 is ~1e-6, too coarse here)
  template <int NSB, int NKB, int ACT, class PipeT, class OutT, class InF>
  __device__ static __forceinline__ void layer(PipeT& pipe, const float* bias, int g, OutT& out, f32x4a (&last)[T], InF&& in) {
    layer_ob16x3<T, NSB, NKB, ACT>(pipe, bias, g, out, last, static_cast<InF&&>(in));
  }
  template <int ACT, int NSB, class OutT>
  __device__ static __forceinline__ void convert_last(OutT& out, const f32x4a (&last)[T]) { convert_last16x3<ACT, T, NSB>(out, last); }
};

struct Depth16Args {
  const char* stream;
  const float* bias;
  uint32_t n_slabs;
  int bias_floats;
  int n_layers;   // trunk layers; layer 0 is the folded 252 -> W one
  const float* o;
  const float* d;
  int64_t R;
  float near_, far_, radius;
  float* z;
};

// PROD: ten layers of 8 K-blocks -> 256 (the production DepthNet after the fold) as straight-line code, every layer a
// generated statement: the embedding is set V (v[128:255]), layers alternate V -> A -> V, the head reads set V.
template <class E, int NKB, bool PROD = false>   // E: engine policy; NKB = W / 32 K-blocks of a hidden layer
__global__ void __launch_bounds__(kWaves * 64)
depthnet_ob16_kernel(Depth16Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using M = typename E::M;
  constexpr int T = E::T, NWAVES = kWaves, NSB = 2 * NKB;
  constexpr bool PT = E::kPreciseTrig;
  using Block = typename M::Block;
  using PipeT = Pipe<M, NWAVES, 0, kOb16Depth, kOb16Ahead>;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, g = lane >> 4;

  // LDS: [weight ring][bias image][input staging: per wave 6 x 256 B]
  float* bias_lds = reinterpret_cast<float*>(smem + PipeT::kLdsBytes);
  for (int i = threadIdx.x; i < a.bias_floats; i += NWAVES * 64) bias_lds[i] = a.bias[i];
  __syncthreads();
  const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(NS_LDS_PTR(smem)));
  const uint32_t stage_base = lds0 + PipeT::kLdsBytes + ((static_cast<uint32_t>(a.bias_floats) * 4u + 15u) & ~15u) +
                              static_cast<uint32_t>(wave) * (6 * 256);

  PipeT ring;
  ring.init(a.stream, smem, a.n_slabs, wave, lane);

  const int64_t n_tiles = (a.R + 15) / 16;
  const int64_t n_groups = (n_tiles + NWAVES * T - 1) / (NWAVES * T);
  auto ray_of = [&](int64_t grp, int t, int l16, bool& valid) -> int64_t {
    const int64_t r = ((grp * NWAVES + wave) * T + t) * 16 + l16;
    valid = r < a.R;
    return valid ? r : a.R - 1;
  };
  // the NEXT group's rays are fetched by LDS-DMA right after layer 0 of the current one (no register is held across
  // the network, no global-load wait -- which would also wait for the weight DMA in flight -- at a group boundary):
  // lane j fetches the six values of the j-th of the wave's 16 T consecutive rays; slots o 0..2, d 3..5
  auto prefetch = [&](int64_t grp) {
    bool valid;
    const int64_t r = ray_of(grp, (lane >> 4) % T, lane & 15, valid);   // (T < 4: the upper lanes re-fetch, harmlessly)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      lds_dma4(a.o + r * 3 + c, stage_base + c * 256);
      lds_dma4(a.d + r * 3 + c, stage_base + (3 + c) * 256);
    }
  };
  auto staged = [&](int t, int slot) -> float {
    return *reinterpret_cast<const float __attribute__((address_space(3)))*>(
        static_cast<uintptr_t>(stage_base + slot * 256 + (t * 16 + n) * 4));
  };

  prefetch(blockIdx.x);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    Block e[T][kInKB];
    asm volatile("" ::: "memory");   // the staged inputs landed several slab steps ago (in-order vmcnt)
    static_for<T>([&](auto t_) {
      constexpr int t = decltype(t_)::value;
      float o[3], d[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) { o[c] = staged(t, c); d[c] = staged(t, 3 + c); }
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
      Block e3[2], e6[4];
      embedN_16<M, PT, 3, 10, 2>(e3, o, g);
      e[t][0] = e3[0]; e[t][1] = e3[1];
      embedN_16<M, PT, 3, 10, 2>(e3, d, g);
      e[t][2] = e3[0]; e[t][3] = e3[1];
      embedN_16<M, PT, 6, 10, 4>(e6, x6, g);
      e[t][4] = e6[0]; e[t][5] = e6[1]; e[t][6] = e6[2]; e[t][7] = e6[3];
    });

    const float* bias = bias_lds;
    Block hA[T][NKB], hB[T][NKB];
    f32x4a last[T];
    auto in_e = [&](auto t_, auto kb_) -> const Block& { return e[decltype(t_)::value][decltype(kb_)::value]; };
    auto in_A = [&](auto t_, auto kb_) -> const Block& { return hA[decltype(t_)::value][decltype(kb_)::value]; };
    auto in_B = [&](auto t_, auto kb_) -> const Block& { return hB[decltype(t_)::value][decltype(kb_)::value]; };

    if constexpr (PROD) {
      constexpr bool SPLIT = std::is_same<M, Mma16F16x3>::value;   // (hi, lo) blocks: two tiles fill the registers four plain tiles do
      constexpr int NT = 4;                                        // 8-K-block register tuples per activation set
      static_assert(T == (SPLIT ? 2 : 4) && NKB == 8 && kInKB == 8 && NWAVES == 4, "the generated streams are W = 256, four waves");
      prefetch(grp + gridDim.x);   // before the first statement: compiled code between two statements costs register copies
      u32x4 A[8 * NT], V[8 * NT];
      static_for<T>([&](auto t_) {
        static_for<kInKB>([&](auto kb_) {
          constexpr int t = decltype(t_)::value, kb = decltype(kb_)::value;
          if constexpr (SPLIT) {
            V[2 * (8 * t + kb)] = __builtin_bit_cast(u32x4, e[t][kb].hi); V[2 * (8 * t + kb) + 1] = __builtin_bit_cast(u32x4, e[t][kb].lo);
          } else {
            V[8 * t + kb] = __builtin_bit_cast(u32x4, e[t][kb].v);
          }
        });
      });
      hidden_leaky_asm_run<M, NT, false>(ring, bias, g, A, V); bias += NSB * 16;   // 0 (folded input layer): V -> A
      hidden_leaky_asm_run<M, NT, true>(ring, bias, g, A, V); bias += NSB * 16;    // 1: A -> V
      hidden_leaky_asm_run<M, NT, false>(ring, bias, g, A, V); bias += NSB * 16;   // 2
      hidden_leaky_asm_run<M, NT, true>(ring, bias, g, A, V); bias += NSB * 16;    // 3
      hidden_leaky_asm_run<M, NT, false>(ring, bias, g, A, V); bias += NSB * 16;   // 4
      hidden_leaky_asm_run<M, NT, true>(ring, bias, g, A, V); bias += NSB * 16;    // 5
      hidden_leaky_asm_run<M, NT, false>(ring, bias, g, A, V); bias += NSB * 16;   // 6
      hidden_leaky_asm_run<M, NT, true>(ring, bias, g, A, V); bias += NSB * 16;    // 7
      hidden_leaky_asm_run<M, NT, false>(ring, bias, g, A, V); bias += NSB * 16;   // 8
      hidden_leaky_asm_run<M, NT, true>(ring, bias, g, A, V); bias += NSB * 16;    // 9: the trunk's output is set V
      static_for<T>([&](auto t_) {
        static_for<NKB>([&](auto kb_) {
          constexpr int t = decltype(t_)::value, kb = decltype(kb_)::value;
          if constexpr (SPLIT) {
            hB[t][kb].hi = __builtin_bit_cast(f16x8, V[2 * (8 * t + kb)]); hB[t][kb].lo = __builtin_bit_cast(f16x8, V[2 * (8 * t + kb) + 1]);
          } else {
            hB[t][kb].v = __builtin_bit_cast(typename M::AFrag, V[8 * t + kb]);
          }
        });
      });
      E::template layer<1, NKB, kNone>(ring, bias, g, hA, last, in_B);
    } else {
    // layer 0 (folded): e -> hA
    E::template layer<NSB, kInKB, kLeaky>(ring, bias, g, hA, last, in_e);
    E::template convert_last<kLeaky, NSB>(hA, last); bias += NSB * 16;
    prefetch(grp + gridDim.x);   // clamped to the last ray past the end: loaded, never used
    int l = 1;
    for (; l + 1 < a.n_layers; l += 2) {   // two trunk layers per trip: hA -> hB -> hA
      E::template layer<NSB, NKB, kLeaky>(ring, bias, g, hB, last, in_A);
      E::template convert_last<kLeaky, NSB>(hB, last); bias += NSB * 16;
      E::template layer<NSB, NKB, kLeaky>(ring, bias, g, hA, last, in_B);
      E::template convert_last<kLeaky, NSB>(hA, last); bias += NSB * 16;
    }
    if (l < a.n_layers) {  // odd layer left over: hA -> hB, then move back
      E::template layer<NSB, NKB, kLeaky>(ring, bias, g, hB, last, in_A);
      E::template convert_last<kLeaky, NSB>(hB, last); bias += NSB * 16;
      static_for<T>([&](auto t_) { static_for<NKB>([&](auto b_) { hA[decltype(t_)::value][decltype(b_)::value] = hB[decltype(t_)::value][decltype(b_)::value]; }); });
    }
    // head (W -> 1): row 0 of a 16-row sub-block (lane group 0, register 0), sigmoid, z = near (1 - s) + far s
    E::template layer<1, NKB, kNone>(ring, bias, g, hB, last, in_A);
    }
    if (g == 0) {
      static_for<T>([&](auto t_) {
        constexpr int t = decltype(t_)::value;
        bool valid;
        const int64_t r = ray_of(grp, t, n, valid);
        const float depth = 1.0f / (1.0f + expf(-last[t][0]));
        if (valid) a.z[r] = a.near_ * (1.0f - depth) + a.far_ * depth;  // depth_net.py:168
      });
    }
  }
  ring.finish();
}

// ---- MIXED operands (NS_DTYPE_F16M): the first KX trunk layers on split fp16 operands, the rest on plain fp16 --------------
// Where the fp16 DepthNet loses its depth: per-layer rounding of operands, emulated on the production network's weights
// (tools/depthnet_layer_sensitivity.py): the first three layers contribute 90 % of the depth error's variance -- their inputs are the widest-ranged
// activations of the network -- the last five 3 %.  So the production trunk (ten 256-wide LeakyReLU layers + head) runs its
// first KX layers as the f16x3 statements (hi + lo operand pairs, three MFMAs per product term: two tiles of 16 rays fill the
// registers) and layers KX .. 9 as the fp16 statements (four tiles).  A wave therefore takes its four tiles through the split
// layers in two halves -- the weight stream holds those layers twice in a row -- parks the first half's results (the hi parts:
// exactly the fp16 roundings the plain layers take as input) in LDS, and joins the halves for the plain layers.
// MFMAs per ray: (3 KX + 10 - KX) / 10 of the fp16 kernel's (1.6 x at KX = 3; all-split: 3 x).
template <int KX>
__global__ void __launch_bounds__(kWaves * 64)
depthnet_mix_kernel(Depth16Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(KX >= 1 && KX <= 9 && (KX & 1) == 1, "an odd number of split layers: their output is set A, as layer KX expects");
  using MS = Mma16F16x3;
  using MP = Mma16F16;
  constexpr int T = 4, NWAVES = kWaves, NKB = 8, NSB = 16, NT = 4;
  using PipeT = Pipe<MP, NWAVES, 0, kOb16Depth, kOb16Ahead>;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, g = lane >> 4;

  // LDS: [weight ring][bias image][input staging: per wave 6 x 256 B][parked half: per wave 16 blocks x 1 KiB]
  float* bias_lds = reinterpret_cast<float*>(smem + PipeT::kLdsBytes);
  for (int i = threadIdx.x; i < a.bias_floats; i += NWAVES * 64) bias_lds[i] = a.bias[i];
  __syncthreads();
  const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(NS_LDS_PTR(smem)));
  const uint32_t after_bias = lds0 + PipeT::kLdsBytes + ((static_cast<uint32_t>(a.bias_floats) * 4u + 15u) & ~15u);
  const uint32_t stage_base = after_bias + static_cast<uint32_t>(wave) * (6 * 256);
  const uint32_t park_base = after_bias + NWAVES * (6 * 256) + static_cast<uint32_t>(wave) * (16 * 1024) + static_cast<uint32_t>(lane) * 16u;
  typedef u32x4 __attribute__((address_space(3))) * ParkPtr;
  auto park_at = [&](int k) -> ParkPtr { return reinterpret_cast<ParkPtr>(static_cast<uintptr_t>(park_base + static_cast<uint32_t>(k) * 1024u)); };

  PipeT ring;
  ring.init(a.stream, smem, a.n_slabs, wave, lane);

  const int64_t n_tiles = (a.R + 15) / 16;
  const int64_t n_groups = (n_tiles + NWAVES * T - 1) / (NWAVES * T);
  auto ray_of = [&](int64_t grp, int t, int l16, bool& valid) -> int64_t {
    const int64_t r = ((grp * NWAVES + wave) * T + t) * 16 + l16;
    valid = r < a.R;
    return valid ? r : a.R - 1;
  };
  auto prefetch = [&](int64_t grp) {
    bool valid;
    const int64_t r = ray_of(grp, lane >> 4, lane & 15, valid);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      lds_dma4(a.o + r * 3 + c, stage_base + c * 256);
      lds_dma4(a.d + r * 3 + c, stage_base + (3 + c) * 256);
    }
  };
  auto staged = [&](int t, int slot) -> float {
    return *reinterpret_cast<const float __attribute__((address_space(3)))*>(
        static_cast<uintptr_t>(stage_base + slot * 256 + (t * 16 + n) * 4));
  };

  prefetch(blockIdx.x);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    u32x4 A[8 * NT], V[8 * NT];
    asm volatile("" ::: "memory");   // the staged inputs landed several slab steps ago (in-order vmcnt)
    // ---- the split layers, two tiles at a time
    static_for<2>([&](auto half_) {
      constexpr int half = decltype(half_)::value;
      static_for<2>([&](auto tt_) {
        constexpr int tt = decltype(tt_)::value, t = 2 * half + tt;
        float o[3], d[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { o[c] = staged(t, c); d[c] = staged(t, 3 + c); }
        float x6[6];
        {   // ray-sphere intersections, utils.py:182-217 (NaN when the line misses, by design): the arithmetic of the kernels above
          const float b = 2.0f * ((d[0] * o[0] + d[1] * o[1]) + d[2] * o[2]);
          const float on = sqrtf(__builtin_fmaf(o[2], o[2], __builtin_fmaf(o[1], o[1], o[0] * o[0])));
          const float c = on * on - a.radius * a.radius;
          const float aa = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2];
          const float sq = sqrtf(b * b - 4.0f * aa * c);
          const float t0 = (-b - sq) / (2.0f * aa), t1 = (-b + sq) / (2.0f * aa);
#pragma unroll
          for (int c3 = 0; c3 < 3; ++c3) { x6[c3] = o[c3] + t0 * d[c3]; x6[3 + c3] = o[c3] + t1 * d[c3]; }
        }
        typename MS::Block e3[2], e6[4];
        auto put = [&](int kb, const typename MS::Block& blk) {
          V[2 * (8 * tt + kb)] = __builtin_bit_cast(u32x4, blk.hi); V[2 * (8 * tt + kb) + 1] = __builtin_bit_cast(u32x4, blk.lo);
        };
        embedN_16<MS, true, 3, 10, 2>(e3, o, g);
        put(0, e3[0]); put(1, e3[1]);
        embedN_16<MS, true, 3, 10, 2>(e3, d, g);
        put(2, e3[0]); put(3, e3[1]);
        embedN_16<MS, true, 6, 10, 4>(e6, x6, g);
        put(4, e6[0]); put(5, e6[1]); put(6, e6[2]); put(7, e6[3]);
      });
      if constexpr (half == 1) prefetch(grp + gridDim.x);   // this group's staged values have all been consumed
      const float* bias = bias_lds;
      static_for<KX>([&](auto l_) {
        constexpr int l = decltype(l_)::value;
        hidden_leaky_asm_run<MS, NT, (l & 1) != 0>(ring, bias, g, A, V); bias += NSB * 16;   // V -> A -> V ... : the last one leaves set A
      });
      // the hi parts of the two tiles' 16 blocks: the first half waits in LDS, the second moves up to tiles 2, 3 of the plain set
      if constexpr (half == 0) {
        static_for<16>([&](auto k_) { constexpr int k = decltype(k_)::value; *park_at(k) = A[2 * k]; });
      } else {
        u32x4 hi[16];
        static_for<16>([&](auto k_) { constexpr int k = decltype(k_)::value; hi[k] = A[2 * k]; });
        static_for<16>([&](auto k_) { constexpr int k = decltype(k_)::value; A[16 + k] = hi[k]; A[k] = *park_at(k); });
      }
    });
    // ---- the plain layers KX .. 9 on all four tiles (A[8 t + kb]), then the head on set V
    const float* bias = bias_lds + KX * NSB * 16;
    static_for<10 - KX>([&](auto i_) {
      constexpr int l = KX + decltype(i_)::value;
      hidden_leaky_asm_run<MP, NT, (l & 1) != 0>(ring, bias, g, A, V); bias += NSB * 16;
    });
    typename MP::Block hA[T][NKB], hB[T][NKB];
    f32x4a last[T];
    static_for<T>([&](auto t_) {
      static_for<NKB>([&](auto kb_) {
        constexpr int t = decltype(t_)::value, kb = decltype(kb_)::value;
        hB[t][kb].v = __builtin_bit_cast(typename MP::AFrag, V[8 * t + kb]);
      });
    });
    auto in_B = [&](auto t_, auto kb_) -> const typename MP::Block& { return hB[decltype(t_)::value][decltype(kb_)::value]; };
    layer_ob16<MP, T, 1, NKB, kNone>(ring, bias, g, hA, last, in_B);
    if (g == 0) {
      static_for<T>([&](auto t_) {
        constexpr int t = decltype(t_)::value;
        bool valid;
        const int64_t r = ray_of(grp, t, n, valid);
        const float depth = 1.0f / (1.0f + expf(-last[t][0]));
        if (valid) a.z[r] = a.near_ * (1.0f - depth) + a.far_ * depth;  // depth_net.py:168
      });
    }
  }
  ring.finish();
}

// slabs of the mixed program: the KX split layers twice, the plain layers, the head
int depth16_mix_program_slabs(int kx) {
  const int NSB = 16, NKB = 8, dp = kOb16Depth;
  return 2 * kx * ob16_layer_slabs(NSB, 2 * NKB, dp) + (10 - kx) * ob16_layer_slabs(NSB, NKB, dp) + ob16_layer_slabs(1, NKB, dp);
}

template <int KX>
int launch_mix(Depth16Args& a, hipStream_t stream) {
  const size_t lds = static_cast<size_t>(Pipe<Mma16F16, kWaves, 0, kOb16Depth, kOb16Ahead>::kLdsBytes) +
                     ((static_cast<size_t>(a.bias_floats) * 4 + 15) & ~size_t(15)) + static_cast<size_t>(kWaves) * 6 * 256 +
                     static_cast<size_t>(kWaves) * 16 * 1024;
  if (lds > 160 * 1024) {
    ns::set_error("ns_depthnet_forward: %zu bytes of LDS needed by the mixed-operand kernel", lds);
    return NS_E_UNSUPPORTED;
  }
  auto kern = depthnet_mix_kernel<KX>;
  NS_HIP(ns::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds));
  const int64_t n_tiles = (a.R + 15) / 16;
  const int64_t n_groups = (n_tiles + kWaves * 4 - 1) / (kWaves * 4);
  int cus = ns::cu_count();
  if (cus <= 0) cus = 256;
  const int grid = static_cast<int>(n_groups < cus ? n_groups : cus);
  kern<<<grid, kWaves * 64, lds, stream>>>(a);
  NS_LAUNCH_CHECK();
  return NS_OK;
}

int depth16_program_slabs(int W, int n_layers, int cpk) {   // cpk: stream chunks per K-block (2 for split operands)
  const int NSB = W / 16, NKB = W / 32, dp = kOb16Depth;
  return ob16_layer_slabs(NSB, cpk * kInKB, dp) + (n_layers - 1) * ob16_layer_slabs(NSB, cpk * NKB, dp) +
         ob16_layer_slabs(1, cpk * NKB, dp);
}

template <class E, int NKB, bool PROD = false>
int launch(Depth16Args& a, hipStream_t stream) {
  using M = typename E::M;
  const size_t lds = static_cast<size_t>(Pipe<M, kWaves, 0, kOb16Depth, kOb16Ahead>::kLdsBytes) +
                     ((static_cast<size_t>(a.bias_floats) * 4 + 15) & ~size_t(15)) + static_cast<size_t>(kWaves) * 6 * 256;
  if (lds > 160 * 1024) {
    ns::set_error("ns_depthnet_forward: %zu bytes of LDS needed (too many layers for the resident bias image)", lds);
    return NS_E_UNSUPPORTED;
  }
  auto kern = depthnet_ob16_kernel<E, NKB, PROD>;
  NS_HIP(ns::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds));
  const int64_t n_tiles = (a.R + 15) / 16;
  const int64_t n_groups = (n_tiles + kWaves * E::T - 1) / (kWaves * E::T);
  int cus = ns::cu_count();
  if (cus <= 0) cus = 256;
  const int grid = static_cast<int>(n_groups < cus ? n_groups : cus);
  kern<<<grid, kWaves * 64, lds, stream>>>(a);
  NS_LAUNCH_CHECK();
  return NS_OK;
}

}  // namespace

// called by ns_depthnet_forward for handles packed with layout 16 (arguments validated there)
int ns_depthnet_forward_ob16(const ns_weights* net, const float* o_dev, const float* d_dev, int64_t R, float near_,
                             float far_, float sphere_radius, float* z_dev, hipStream_t stream) {
  if (net->dtype == NS_DTYPE_F16M) {      // mixed operands: the production shape only (the packer refuses others)
    if (net->width != 256 || net->depth != 10 || depth16_mix_program_slabs(NS_F16M_SPLIT_LAYERS) != static_cast<int>(net->n_slabs)) {
      ns::set_error("ns_depthnet_forward: not a mixed-operand stream of the 10 x 256 trunk (%u slabs)", net->n_slabs);
      return NS_E_INVALID;
    }
    Depth16Args m{};
    m.stream = static_cast<const char*>(net->stream_dev);
    m.bias = net->bias_dev; m.n_slabs = net->n_slabs; m.bias_floats = net->bias_floats;
    m.n_layers = net->depth; m.o = o_dev; m.d = d_dev; m.R = R;
    m.near_ = near_; m.far_ = far_; m.radius = sphere_radius; m.z = z_dev;
    return launch_mix<NS_F16M_SPLIT_LAYERS>(m, stream);
  }
  const int cpk = net->dtype == NS_DTYPE_F16X3 ? 2 : 1;
  if (depth16_program_slabs(net->width, net->depth, cpk) != static_cast<int>(net->n_slabs)) {
    ns::set_error("ns_depthnet_forward: packed stream has %u slabs, kernel program expects %d", net->n_slabs,
                  depth16_program_slabs(net->width, net->depth, cpk));
    return NS_E_INVALID;
  }
  Depth16Args a{};
  a.stream = static_cast<const char*>(net->stream_dev);
  a.bias = net->bias_dev; a.n_slabs = net->n_slabs; a.bias_floats = net->bias_floats;
  a.n_layers = net->depth; a.o = o_dev; a.d = d_dev; a.R = R;
  a.near_ = near_; a.far_ = far_; a.radius = sphere_radius; a.z = z_dev;
  const bool wide = net->width == 256;
  if (net->dtype == NS_DTYPE_BF16) return wide ? launch<Plain16<Mma16BF16>, 8>(a, stream) : launch<Plain16<Mma16BF16>, 4>(a, stream);
  if (net->dtype == NS_DTYPE_F16) {
    if (wide && net->depth == 10 && !ns::debug_flags().generic_kernels)      // the production trunk: generated layer streams
      return launch<Plain16<Mma16F16>, 8, true>(a, stream);
    return wide ? launch<Plain16<Mma16F16>, 8>(a, stream) : launch<Plain16<Mma16F16>, 4>(a, stream);
  }
  if (net->dtype == NS_DTYPE_F16X3) {
    if (wide && net->depth == 10 && !ns::debug_flags().generic_kernels) return launch<Split16, 8, true>(a, stream);
    return wide ? launch<Split16, 8>(a, stream) : launch<Split16, 4>(a, stream);
  }
  return NS_E_UNSUPPORTED;
}
