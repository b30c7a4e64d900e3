// Radiance-field MLP forward for the 16-bit operand paths (bf16 / f16) on v_mfma_f32_16x16x32: the shape that
// sustains the most under the MI355X power cap (tools/mfma_peak.hip: 2.14 vs 1.87 PFLOP/s for 32x32x16 with live
// operands).  Same operator as ns_nerf_mlp.hip (run_network + NeRF.forward, Trainer.py:789-806 and
// run_nerf_helpers.py:67-134): positional encoding of points and view directions, DxW trunk with the input skip,
// (feature o view) layer carrying the sigma head as one extra output row, rgb head, one persistent kernel.  A wave owns T = 4 tiles of 16 samples (64 samples);
// every A fragment (16 output rows x 32 input features, 1 KiB) read from LDS feeds 4 MFMAs; layers are walked one
// 16-row output sub-block at a time (layer_ob16 in ns_mlp_engine.h; weight stream layout 16 of ns_pack.hip).
#include "ns_common.h"
#include "ns_composite_ray.h"
#include "ns_mlp_engine.h"
#include "ns_place.h"
#include "ns_weights.h"

// This file is compiled twice: as itself, and with -DNS_OB16_TU_T5 as a second translation unit that holds only the
// five-tile production kernels (the two units build in parallel; each is minutes of register allocation).
namespace nsob16 {
struct Nerf16Args {
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
  // In-kernel compositing (the DepthNet branch of render_rays_test as ONE kernel, nerf_utils.py:836-865): comp != 0 runs
  // raw2outputs (sampling_trainer.py:153-230) on the wave scan of ns_composite_ray.h in the epilogue -- raw then never
  // leaves the CU (raw may be NULL).  comp == 1: depths from the array z [S];  comp == 2: sample_points_around_mean
  // ("uniform", utils.py:231-241) evaluated in-kernel from the DepthNet depth mean [R] -- no z array exists.
  // N is a power of two <= 64 (whole rays per 64-sample chunk) or a multiple of 64 up to 512 (whole chunks per ray).
  int comp;
  int n_shift;             // log2 N when N is a power of two, else -1
  const float* mean;
  float std_, lin_step;    // the grid linspace(-std, std, N - 1) and its step (correctly rounded on the host)
  int white_bkgd;
  float* rgb; int64_t rgb_stride;
  float* disp; int64_t disp_stride;
  float* weights;          // [S] or NULL
  float* z_out;            // [S] or NULL (comp == 2: the depths the kernel placed)
  float* pts_out;          // [S,3] or NULL
  const float* sig_last;   // NULL, or [R,4]: element 3 of row r replaces sigma of ray r's last sample (the guard pass)
  // rays longer than a 64-sample chunk (N = 64 m, m = m_chunks >= 2; 0 otherwise): a workgroup then walks sg_groups CONSECUTIVE
  // groups -- lcm(group samples, N) samples, whole rays -- before it jumps, so that a ray's chunks meet in one workgroup and
  // the transmittance / sums of the ray that is open at a group boundary carry over in LDS
  int m_chunks, sg_groups;
  // the selective guard (single-chunk rays): a ray whose own sigma of the last sample is within fix_thr of zero -- where the step
  // alpha = step(sigma) could flip under the 16-bit rounding -- leaves a 16-float record {tree sums r g b depth acc, T, raw rgb of
  // the last sample, its z and dist, ray index lo / hi} at slot atomicAdd(fix_count) of fix_rec (ns_fix_last_sample re-evaluates
  // sigma through the fp32-grade handle and repeats the last addition)
  float fix_thr;
  uint32_t* fix_count;
  float* fix_rec;
};
// the five-tile production kernel (PROD, 80 samples per wave): defined in the NS_OB16_TU_T5 unit
int launch_prod_t5(int dtype, bool embedded, Nerf16Args& a, hipStream_t stream);
}  // namespace nsob16

namespace {

using namespace nsmlp;

#ifndef NS_NERF16_T
#define NS_NERF16_T 4
#endif
#ifndef NS_NERF16_WAVES
#define NS_NERF16_WAVES 4
#endif
#ifndef NS_OB16_PROD_T
#define NS_OB16_PROD_T 0          // 16-sample tiles per wave in the production kernel: 4, 5, or 0 = chosen per launch
#endif
constexpr int kT = NS_NERF16_T;          // 16-sample tiles per wave
constexpr int kWaves = NS_NERF16_WAVES;  // 4: one wave per SIMD, ~256 AGPRs of activations + accumulators per wave
                                         // (8 waves x T = 2, two per SIMD in 256 registers each: measured slower)

}  // namespace
#ifndef NS_OB16_ASM_INC
#define NS_OB16_ASM_INC "ns_ob16_asm.inc"
#endif
#include NS_OB16_ASM_INC
namespace {
// One W = 256 hidden layer (ReLU) as a generated asm statement: set A (hA, AGPRs) -> set V (hB, VGPRs) or back; SKIP:
// K-blocks 0, 1 are the embedded point xs.  Same chunk walk, ring protocol and arithmetic as layer_ob16<> +
// convert_last16<> (bit-identical results); the ring's bookkeeping is handed over and taken back here.
template <class M, int T, bool IN_A, bool SKIP, class PipeT>
__device__ __forceinline__ void hidden_layer_asm(PipeT& ring, const float* bias_lds, int g, typename M::Block (&hA)[T][8],
                                                 typename M::Block (&hB)[T][8], const typename M::Block (&xs)[T][2]) {
  u32x4 A[8 * T], V[8 * T], X[2 * T];
  static_for<T>([&](auto t_) {
    constexpr int t = decltype(t_)::value;
    static_for<8>([&](auto kb_) {
      constexpr int kb = decltype(kb_)::value;
      if constexpr (IN_A) A[8 * t + kb] = __builtin_bit_cast(u32x4, hA[t][kb].v);
      else V[8 * t + kb] = __builtin_bit_cast(u32x4, hB[t][kb].v);
    });
    static_for<2>([&](auto kb_) { X[2 * t + decltype(kb_)::value] = __builtin_bit_cast(u32x4, xs[t][decltype(kb_)::value].v); });
  });
  hidden_asm_run<M, T, IN_A, SKIP>(ring, bias_lds, g, A, V, X);
  static_for<T>([&](auto t_) {
    constexpr int t = decltype(t_)::value;
    static_for<8>([&](auto kb_) {
      constexpr int kb = decltype(kb_)::value;
      if constexpr (IN_A) hB[t][kb].v = __builtin_bit_cast(typename M::AFrag, V[8 * t + kb]);
      else hA[t][kb].v = __builtin_bit_cast(typename M::AFrag, A[8 * t + kb]);
    });
  });
}
// the other three layers of the production network as generated statements: layer 0, the view layer with
// the sigma sub-block, the rgb head
template <class M, int T, class PipeT>
__device__ __forceinline__ void layer0_asm(PipeT& ring, const float* bias_lds, int g, const typename M::Block (&xe)[T][2],
                                           typename M::Block (&hA)[T][8]) {
  u32x4 X[2 * T], A[8 * T];
  static_for<T>([&](auto t_) {
    constexpr int t = decltype(t_)::value;
    X[2 * t] = __builtin_bit_cast(u32x4, xe[t][0].v); X[2 * t + 1] = __builtin_bit_cast(u32x4, xe[t][1].v);
  });
  layer0_asm_run<M, T>(ring, bias_lds, g, X, A);
  static_for<T>([&](auto t_) {
    constexpr int t = decltype(t_)::value;
    static_for<8>([&](auto kb_) { hA[t][decltype(kb_)::value].v = __builtin_bit_cast(typename M::AFrag, A[8 * t + decltype(kb_)::value]); });
  });
}
template <class M, int T, class PipeT>
__device__ __forceinline__ void views_asm(PipeT& ring, const float* bias_lds, int g, const typename M::Block (&hB)[T][8],
                                          const typename M::Block (&vs)[T], typename M::Block (&hA)[T][8], f32x4a (&last)[T]) {
  u32x4 V[8 * T], D[T], A[4 * T], ACCO[T];
  static_for<T>([&](auto t_) {
    constexpr int t = decltype(t_)::value;
    static_for<8>([&](auto kb_) { V[8 * t + decltype(kb_)::value] = __builtin_bit_cast(u32x4, hB[t][decltype(kb_)::value].v); });
    D[t] = __builtin_bit_cast(u32x4, vs[t].v);
  });
  views_asm_run<M, T>(ring, bias_lds, g, V, D, A, ACCO);
  static_for<T>([&](auto t_) {
    constexpr int t = decltype(t_)::value;
    static_for<4>([&](auto kb_) { hA[t][decltype(kb_)::value].v = __builtin_bit_cast(typename M::AFrag, A[4 * t + decltype(kb_)::value]); });
    last[t] = __builtin_bit_cast(f32x4a, ACCO[t]);
  });
}
template <class M, int T, class PipeT>
__device__ __forceinline__ void rgb_asm(PipeT& ring, const float* bias_lds, int g, const typename M::Block (&hA)[T][8], f32x4a (&last)[T]) {
  u32x4 A[4 * T], ACCO[T];
  static_for<T>([&](auto t_) {
    constexpr int t = decltype(t_)::value;
    static_for<4>([&](auto kb_) { A[4 * t + decltype(kb_)::value] = __builtin_bit_cast(u32x4, hA[t][decltype(kb_)::value].v); });
  });
  rgb_asm_run<M, T>(ring, bias_lds, g, A, ACCO);
  static_for<T>([&](auto t_) { last[decltype(t_)::value] = __builtin_bit_cast(f32x4a, ACCO[decltype(t_)::value]); });
}

using nsob16::Nerf16Args;

// PROD: the production network (8 x 256, skips = [4], view directions: experiments/run.py) as straight-line code whose
// seven hidden layers are the generated asm statements -- no loop over layers, so the activation sets stay in the registers
// the statements pin them to; every other network takes the generic, compiler-scheduled path.
template <class M, int NKB, bool EMBEDDED, bool PROD = false, int TT = kT>   // NKB = W / 32 K-blocks of a hidden layer
__global__ void __launch_bounds__(kWaves * 64)
nerf_mlp_ob16_kernel(Nerf16Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int T = TT, NWAVES = kWaves, NSB = 2 * NKB;   // 16-row output sub-blocks of a hidden layer
  using Block = typename M::Block;
  using PipeT = Pipe<M, NWAVES, 0, kOb16Depth, kOb16Ahead>;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, g = lane >> 4;

  // LDS: [weight ring][bias image][embedding stash: per wave T x 3 blocks x 1 KiB][input staging: per wave 11 rows of 16 T floats]
  //      [compositing records (a.comp): raw float4 per sample of the group | {z, dist} float2 per sample, two group parities
  //       | sigma of a ray's last sample from the guard pass, one float per ray of the group, two parities]
  float* bias_lds = reinterpret_cast<float*>(smem + PipeT::kLdsBytes);
  for (int i = threadIdx.x; i < a.bias_floats; i += NWAVES * 64) bias_lds[i] = a.bias[i];
  __syncthreads();

  typedef typename M::AFrag __attribute__((address_space(3))) * StashPtr;
  const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(NS_LDS_PTR(smem)));
  const uint32_t stash_region = lds0 + PipeT::kLdsBytes + ((static_cast<uint32_t>(a.bias_floats) * 4u + 15u) & ~15u);
  const uint32_t stash_base = stash_region + static_cast<uint32_t>(wave) * (T * 3 * 1024) + static_cast<uint32_t>(lane) * 16u;
  auto stash_at = [&](int t, int b) -> StashPtr {
    return reinterpret_cast<StashPtr>(static_cast<uintptr_t>(stash_base + (t * 3 + b) * 1024));
  };
  auto stash_put = [&](int t, int b, const Block& v) { *stash_at(t, b) = v.v; };
  auto stash_get = [&](int t, int b) -> Block { Block v; v.v = *stash_at(t, b); return v; };
  // staging: value slot k (0..10) of sample j (0 .. 16 T - 1) of this wave's group at stage_base + k * kStageRow + j * 4
  constexpr uint32_t kStageRow = T * 64, kStageRows = 11;
  const uint32_t stage_base = stash_region + NWAVES * (T * 3 * 1024) + static_cast<uint32_t>(wave) * (kStageRows * kStageRow);
  // compositing records (only allocated when a.comp): sample i (0 .. GS - 1) of the open group
  constexpr int GS = NWAVES * T * 16;
  const uint32_t comp_region = stash_region + NWAVES * (T * 3 * 1024) + NWAVES * (kStageRows * kStageRow);
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef float v2f __attribute__((ext_vector_type(2)));
  typedef v4f __attribute__((address_space(3))) * CrawPtr;
  typedef v2f __attribute__((address_space(3))) * CzdPtr;
  // The lane id as a value the compiler cannot hoist: everything the compositing code derives from the lane (LDS record
  // addresses per tile, the scan's lane predicates for six segment widths) would otherwise be computed ONCE before the
  // group loop and kept alive across the ten layer statements -- which leave the compiler 32 VGPRs -- i.e. spilled to
  // scratch and reloaded in the epilogue behind s_waitcnt vmcnt(0), waiting out the weight DMA in flight (measured: +1.5 ms
  // per frame).  Two v_mbcnt per use instead.
  auto opaque_lane = [] {
    uint32_t z = 0;
    asm volatile("" : "+s"(z));
    return static_cast<int>(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z)));
  };
  auto craw_at = [&](int i) -> CrawPtr { return reinterpret_cast<CrawPtr>(static_cast<uintptr_t>(comp_region + static_cast<uint32_t>(i) * 16u)); };
  auto czd_at = [&](uint32_t par, int i) -> CzdPtr {
    return reinterpret_cast<CzdPtr>(static_cast<uintptr_t>(comp_region + GS * 16u + (par * GS + static_cast<uint32_t>(i)) * 8u));
  };
  typedef float __attribute__((address_space(3))) * CsigPtr;
  auto csig_at = [&](uint32_t par, int ray) -> CsigPtr {      // ray: index within the group (<= GS / 2 rays)
    return reinterpret_cast<CsigPtr>(static_cast<uintptr_t>(comp_region + GS * 32u + (par * (GS / 2) + static_cast<uint32_t>(ray)) * 4u));
  };
  // rays of several chunks: per chunk of the group its transmittance factor (cP) and its five sums (cS); the ray that is open
  // at the group's end: {carry, r, g, b, depth, acc}, two parities (copen[par] is read, copen[par ^ 1] written)
  auto cscal_at = [&](int k) -> CsigPtr {
    return reinterpret_cast<CsigPtr>(static_cast<uintptr_t>(comp_region + GS * 36u + static_cast<uint32_t>(k) * 4u));
  };
  constexpr int kCP = 0, kCS = 8, kOPEN = 8 + 8 * 8;          // float offsets inside the 128-float scalar block

  PipeT ring;
  ring.init(a.stream, smem, a.n_slabs, wave, lane);

  const int64_t n_tiles = (a.S + 15) / 16;
  const int64_t n_groups = (n_tiles + NWAVES * T - 1) / (NWAVES * T);
  // sample held by lane `l16` (0..15) of tile t of this wave in group grp; clamped to a real sample
  auto sample_of = [&](int64_t grp, int t, int l16, bool& valid) -> int64_t {
    const int64_t sidx = ((grp * NWAVES + wave) * T + t) * 16 + l16;
    valid = sidx < a.S;
    return valid ? sidx : a.S - 1;
  };
  // Inputs of the NEXT group are fetched right after layer 0 of the current one by LDS-DMA (no registers held across
  // the network): lane j of the wave fetches the ten values of the j-th of the wave's 64 consecutive samples.
  // pts mode: p 0..2, v 7..9;  (o, d, z) mode: o 0..2, d 3..5, z 6, v 7..9;  compositing: slot 6 is the ray's DepthNet depth
  // when the samples are placed in-kernel (comp == 2), slot 10 the NEXT sample's depth when they come from z (comp == 1).
  auto prefetch_round = [&](int64_t grp, int tile0) {     // the active lanes fetch tiles tile0 .. tile0 + 3 (clamped to T - 1)
    bool valid;
    const int tl = tile0 + (lane >> 4);
    const int64_t sidx = sample_of(grp, tl < T ? tl : T - 1, lane & 15, valid);   // (surplus lanes re-fetch, harmlessly)
    const int64_t ray = a.S <= 0x7fffffff ? static_cast<int64_t>(static_cast<uint32_t>(sidx) / static_cast<uint32_t>(a.N))
                                          : sidx / a.N;
    auto put = [&](int slot, const float* src) {
      lds_dma4(src, stage_base + slot * kStageRow + tile0 * 64);
    };
    if (a.pts) {
#pragma unroll
      for (int c = 0; c < 3; ++c) put(c, a.pts + sidx * 3 + c);
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) { put(c, a.o + ray * 3 + c); put(3 + c, a.d + ray * 3 + c); }
      if (a.comp == 2) put(6, a.mean + ray);
      else put(6, a.z + sidx);
      if (a.comp == 1) put(10, a.z + (sidx + 1 < a.S ? sidx + 1 : sidx));
      else if (a.sig_last) put(10, a.sig_last + ray * 4 + 3);
    }
    if (a.use_viewdirs) {
#pragma unroll
      for (int c = 0; c < 3; ++c) put(7 + c, a.viewdirs + ray * 3 + c);
    }
  };
  auto prefetch = [&](int64_t grp) {
    if constexpr (!EMBEDDED) {
      prefetch_round(grp, 0);
      if constexpr (T > 4) {      // the fifth tile: 16 lanes (a staging row holds exactly the wave's 16 T samples)
        if (lane < 16 * (T - 4)) prefetch_round(grp, 4);
      }
    }
  };
  auto staged = [&](int t, int slot) -> float {
    return *reinterpret_cast<const float __attribute__((address_space(3)))*>(
        static_cast<uintptr_t>(stage_base + slot * kStageRow + (t * 16 + n) * 4));
  };

  // group order of a workgroup: sg consecutive groups (one, unless rays span several chunks), then a jump of gridDim.x such runs
  const int sg = a.sg_groups > 1 ? a.sg_groups : 1;
  auto group_after = [&](int64_t grp_, int gi_) -> int64_t {
    return gi_ + 1 == sg ? grp_ + static_cast<int64_t>(gridDim.x - 1) * sg + 1 : grp_ + 1;
  };
  prefetch(static_cast<int64_t>(blockIdx.x) * sg);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  uint32_t par = 0;      // parity of the group pass: which {z, dist} record buffer this group writes (compositing only)
  int gi = 0;            // position of the group in its run
  for (int64_t grp = static_cast<int64_t>(blockIdx.x) * sg, nxt_grp = 0; grp < n_groups;
       grp = nxt_grp, gi = (gi + 1 == sg ? 0 : gi + 1), par ^= 1u) {
    nxt_grp = group_after(grp, gi);
    Block xe[T][2];   // embedded point (63 -> 64 features); registers for layer 0 only
    // Non-finite inputs: the reference's arithmetic turns a NaN / inf coordinate into NaN in all four outputs (sin / cos,
    // nn.Linear and torch.relu all propagate it).  Here the packed-int16 ReLU would drop the NEGATIVE NaNs the matrix
    // cores produce, so the samples are flagged (bit t of `bad`, one register across the network) and written as NaN.
    uint32_t bad = 0;
    auto finite = [](float v) { return __builtin_fabsf(v) < __builtin_inff(); };
    asm volatile("" ::: "memory");   // the staged inputs landed several slab steps ago (in-order vmcnt)
    if constexpr (EMBEDDED) {
      static_for<T>([&](auto t_) {
        constexpr int t = decltype(t_)::value;
        Block ve[1];    // embedded view direction (27 -> 32)
        bool valid;
        const float* row = a.x90 + sample_of(grp, t, n, valid) * a.x_stride;
        gather3_16<M, 10, 2>(xe[t], row, g);
        bool ok = finite(row[0]) && finite(row[1]) && finite(row[2]);
        if (a.use_viewdirs) {
          gather3_16<M, 4, 1>(ve, row + 63, g);
          ok = ok && finite(row[63]) && finite(row[64]) && finite(row[65]);
        }
        if (!ok) bad |= 1u << t;
        stash_put(t, 0, xe[t][0]); stash_put(t, 1, xe[t][1]);
        if (a.use_viewdirs) stash_put(t, 2, ve[0]);
      });
    } else {
      // All staged values of the wave's four tiles come out of LDS in ONE burst (40 reads, one wait) before anything is
      // embedded or stashed: read tile by tile, each read sat right in front of its first use (an exposed LDS latency per
      // value, ~3 % of the kernel: nothing else runs on this SIMD while the lone wave waits).
      float P[T][3], V[T][3];          // (compile-time indices only: a runtime index would park the arrays in scratch)
      if (a.comp) {   // (wave-uniform)
        // Sample placement + the compositing record {z, dist * |d|} of every sample of the wave, ONE SAMPLE PER LANE (64 at a
        // time: the tile layout below holds a sample on four lanes, and a per-tile evaluation would cost T times this):
        // sample i of the wave's 16 T on lane i % 64 of pass i / 64.  The records go to LDS -- the epilogue composites from
        // them, and the tiles read their depth back from there a few lines down (same wave: LDS order suffices).
        const int lo = opaque_lane();
        constexpr int kPasses = (16 * T + 63) / 64;
        // group-level scalars: how many of the group's GS samples exist, and the position of its first sample in its ray
        // (N <= 64 divides the group size: 0;  N = 64 m: the run starts on a ray, every group adds GS mod N)
        const int64_t left = a.S - grp * GS;
        const int rem = left < GS ? static_cast<int>(left) : GS;
        const int jg0 = a.m_chunks ? (gi * GS) % a.N : 0;                 // (wave-uniform 32-bit arithmetic, gi < 8)
#pragma unroll
        for (int pass = 0; pass < kPasses; ++pass) {
          const int i = pass * 64 + lo;
          if (i < 16 * T) {
            auto st = [&](int slot) -> float {
              return *reinterpret_cast<const float __attribute__((address_space(3)))*>(
                  static_cast<uintptr_t>(stage_base + slot * kStageRow + i * 4));
            };
            // every staged value of the pass in one burst, ahead of the arithmetic (slot 10 holds the next depth, the guard's
            // sigma, or nothing: read whatever is there, used only where it is defined)
            const float m_or_z = st(6), d0 = st(3), d1 = st(4), d2 = st(5), s10 = st(10);
            const int ig = wave * (16 * T) + i;                             // the sample's index within the group
            const bool valid = ig < rem;
            int j, ray_in_group;                                            // (a sample past the end: any in-range j, never used)
            if (!a.m_chunks) {                                              // N <= 64, a power of two: whole rays per group
              j = ig & (a.N - 1); ray_in_group = ig >> a.n_shift;
            } else {                                                        // N >= 128, jg0 + ig < GS + N <= 3.5 N
              const int x = jg0 + ig;
              ray_in_group = (x >= a.N) + (x >= 2 * a.N) + (x >= 3 * a.N);
              j = x - ray_in_group * a.N;
            }
            float zz = m_or_z, znext = s10;
            if (a.comp == 2)        // sample_points_around_mean("uniform"): depths j and j + 1 of the ray from its mean
              nsplace::uniform_z_pair(m_or_z, a.std_, a.lin_step, a.N - 1, j, zz, znext);
            const float dist_raw = (j < a.N - 1) ? znext - zz : 1e10f;      // sampling_trainer.py:176-180
            *czd_at(par, ig) = v2f{zz, dist_raw * nscomp::ray_norm(d0, d1, d2)};
            // the guard pass's sigma of this ray's last sample: one slot per ray of the group
            if (a.sig_last && j == a.N - 1) *csig_at(par, ray_in_group) = s10;
            if (valid && (a.z_out || a.pts_out)) {
              const int64_t sidx = grp * GS + ig;
              if (a.z_out) a.z_out[sidx] = zz;
              if (a.pts_out) {
                float* q = a.pts_out + sidx * 3;
                q[0] = st(0) + d0 * zz; q[1] = st(1) + d1 * zz; q[2] = st(2) + d2 * zz;
              }
            }
          }
        }
      }
      static_for<T>([&](auto t_) {
        constexpr int t = decltype(t_)::value;
        if (a.pts) {
          static_for<3>([&](auto c_) { P[t][decltype(c_)::value] = staged(t, decltype(c_)::value); });
        } else {
          const float zz = a.comp ? (*czd_at(par, (wave * T + t) * 16 + n)).x : staged(t, 6);
          static_for<3>([&](auto c_) {
            constexpr int c = decltype(c_)::value;
            P[t][c] = staged(t, c) + staged(t, 3 + c) * zz;
          });
        }
        static_for<3>([&](auto c_) {
          constexpr int c = decltype(c_)::value;
          V[t][c] = a.use_viewdirs ? staged(t, 7 + c) : 0.0f;
        });
      });
      asm volatile("" ::: "memory");   // ... and only then the stash writes below (the compiler cannot tell the two LDS regions apart)
      // A view direction is embedded once per RUN of tiles that share it (a wave's consecutive samples lie on one ray when
      // N is a multiple of 64 and the wave has four tiles, on at most two when it has five): tile t reuses the previous
      // tile's embedding when its direction compares equal BY VALUE, wave-uniformly; a NaN compares unequal and is embedded.
      bool new_view[T];
      static_for<T>([&](auto t_) {
        constexpr int t = decltype(t_)::value;
        if constexpr (t == 0) {
          new_view[0] = true;
        } else {
          const bool same = V[t][0] == V[t - 1][0] && V[t][1] == V[t - 1][1] && V[t][2] == V[t - 1][2];
          new_view[t] = __builtin_amdgcn_ballot_w64(same) != ~0ull;
        }
      });
      Block ve0[1];
      static_for<T>([&](auto t_) {
        constexpr int t = decltype(t_)::value;
        bool ok = finite(P[t][0]) && finite(P[t][1]) && finite(P[t][2]);
        embed3_16<M, false, 10, 2>(xe[t], P[t][0], P[t][1], P[t][2], g);
        stash_put(t, 0, xe[t][0]); stash_put(t, 1, xe[t][1]);
        if (a.use_viewdirs) {
          ok = ok && finite(V[t][0]) && finite(V[t][1]) && finite(V[t][2]);
          if (new_view[t]) embed3_16<M, false, 4, 1>(ve0, V[t][0], V[t][1], V[t][2], g);   // (wave-uniform branch)
          stash_put(t, 2, ve0[0]);
        }
        if (!ok) bad |= 1u << t;
      });
    }

    const float* bias = bias_lds;
    Block hA[T][NKB], hB[T][NKB];
    f32x4a last[T];
    auto in_x = [&](auto t_, auto kb_) -> const Block& { return xe[decltype(t_)::value][decltype(kb_)::value]; };
    auto in_A = [&](auto t_, auto kb_) -> const Block& { return hA[decltype(t_)::value][decltype(kb_)::value]; };
    auto in_B = [&](auto t_, auto kb_) -> const Block& { return hB[decltype(t_)::value][decltype(kb_)::value]; };
    // The skip layer sees cat[x, h]: the embedded point comes back from the per-wave LDS stash ONCE per layer into
    // registers (32 of them, live for that layer only) instead of once per 16-row sub-block -- 8 reads instead of 128
    // per wave pass, none of them right in front of the MFMA that needs it.
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
    if constexpr (PROD) {
      prefetch(nxt_grp);
      layer0_asm<M, T>(ring, bias, g, xe, hA); bias += NSB * 16;
    } else
    { layer_ob16<M, T, NSB, 2, true>(ring, bias, g, hA, last, in_x); convert_last16<M, true, T, NSB>(hA, last); bias += NSB * 16; }
    // next group's inputs (clamped to the last sample past the end: loaded, never used); this group's staged values
    // have been consumed (they fed the embeddings above)
    if constexpr (!PROD) prefetch(nxt_grp);   // (PROD asked before layer 0: compiled code between two statements costs register copies)
    int l = 1;
    if constexpr (PROD) {
      static_assert(NKB == 8 && (T == 4 || T == 5) && NWAVES == 4, "the generated streams are W = 256, four or five tiles, four waves");
      hidden_layer_asm<M, T, true, false>(ring, bias, g, hA, hB, xs); bias += NSB * 16;    // 1
      hidden_layer_asm<M, T, false, false>(ring, bias, g, hA, hB, xs); bias += NSB * 16;   // 2
      hidden_layer_asm<M, T, true, false>(ring, bias, g, hA, hB, xs); bias += NSB * 16;    // 3
      hidden_layer_asm<M, T, false, false>(ring, bias, g, hA, hB, xs); bias += NSB * 16;   // 4
      load_xs();
      hidden_layer_asm<M, T, true, true>(ring, bias, g, hA, hB, xs); bias += NSB * 16;     // 5: cat[x, h]
      hidden_layer_asm<M, T, false, false>(ring, bias, g, hA, hB, xs); bias += NSB * 16;   // 6
      hidden_layer_asm<M, T, true, false>(ring, bias, g, hA, hB, xs); bias += NSB * 16;    // 7: the trunk's output is in hB
      l = 8;
    }
    // layers 1 .. D-1, two per trip (hA -> hB -> hA); the layer after `skip` sees cat[x, h]
    if constexpr (!PROD) {
    for (; l + 1 < a.D; l += 2) {
      if ((a.skip_mask >> (l - 1)) & 1u) { load_xs(); layer_ob16<M, T, NSB, NKB + 2, true>(ring, bias, g, hB, last, in_xA); }
      else layer_ob16<M, T, NSB, NKB, true>(ring, bias, g, hB, last, in_A);
      convert_last16<M, true, T, NSB>(hB, last); bias += NSB * 16;
      if ((a.skip_mask >> l) & 1u) { load_xs(); layer_ob16<M, T, NSB, NKB + 2, true>(ring, bias, g, hA, last, in_xB); }
      else layer_ob16<M, T, NSB, NKB, true>(ring, bias, g, hA, last, in_B);
      convert_last16<M, true, T, NSB>(hA, last); bias += NSB * 16;
    }
    if (l < a.D) {  // odd layer left over: hA -> hB, then move back
      if ((a.skip_mask >> (l - 1)) & 1u) { load_xs(); layer_ob16<M, T, NSB, NKB + 2, true>(ring, bias, g, hB, last, in_xA); }
      else layer_ob16<M, T, NSB, NKB, true>(ring, bias, g, hB, last, in_A);
      convert_last16<M, true, T, NSB>(hB, last); bias += NSB * 16;
      static_for<T>([&](auto t_) { static_for<NKB>([&](auto b_) { hA[decltype(t_)::value][decltype(b_)::value] = hB[decltype(t_)::value][decltype(b_)::value]; }); });
    }
    }
    if (!PROD && !a.use_viewdirs) {
      // output_linear (W -> out_ch, no activation, run_nerf_helpers.py:132-133): ONE 16-row sub-block whose raw accumulators
      // come back in `last`: row 4 g + r sits in register r of lane group g
      layer_ob16<M, T, 1, NKB, kNone>(ring, bias, g, hB, last, in_A);
      static_for<T>([&](auto t_) {
        constexpr int t = decltype(t_)::value;
        bool valid;
        const int64_t sidx = sample_of(grp, t, n, valid);
        static_for<4>([&](auto r_) {
          constexpr int r = decltype(r_)::value;
          const int row = 4 * g + r;
          if (valid && row < a.out_ch) a.raw[sidx * a.out_ch + row] = ((bad >> t) & 1u) ? __builtin_nanf("") : last[t][r];
        });
      });
      continue;
    }
    // views o feature (folded at pack time: feature_linear has no activation, run_nerf_helpers.py:119-125) on
    // cat[h, dirs27] -> W/2, relu: (hA, ve) -> hB[0 .. NKB/2); alpha_linear rides along as row 0 of one extra, LAST
    // sub-block, whose raw accumulators come back in `last`: sigma = row 0 (lane group 0, register 0)
    Block vs[T];   // the embedded view direction, once for the layer's 9 sub-blocks
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
      (void)in_Bv;
      views_asm<M, T>(ring, bias, g, hB, vs, hA, last); bias += (NSB / 2 + 1) * 16;
      static_for<T>([&](auto t_) { sigma[decltype(t_)::value] = last[decltype(t_)::value][0]; });
      rgb_asm<M, T>(ring, bias, g, hA, last);
    } else {
    layer_ob16<M, T, NSB / 2 + 1, NKB + 1, kRelu>(ring, bias, g, hB, last, in_Av); bias += (NSB / 2 + 1) * 16;
    static_for<T>([&](auto t_) { sigma[decltype(t_)::value] = last[decltype(t_)::value][0]; });
    // rgb (W/2 -> 3): rows 0..2 (lane group 0, registers 0..2)
    layer_ob16<M, T, 1, NKB / 2, kNone>(ring, bias, g, hA, last, in_B);
    }

    bool comp = false;
    if constexpr (!EMBEDDED) comp = a.comp != 0;
    const int le = opaque_lane();      // the lane id of the epilogue (see opaque_lane)
    if (le < 16) {                     // lane group g == 0 holds the outputs: rows 0..2 = rgb, sigma from the view layer
      static_for<T>([&](auto t_) {
        constexpr int t = decltype(t_)::value;
        bool valid;
        const int64_t sidx = sample_of(grp, t, le, valid);
        float4 o4 = make_float4(last[t][0], last[t][1], last[t][2], sigma[t]);
        if ((bad >> t) & 1u) { const float q = __builtin_nanf(""); o4 = make_float4(q, q, q, q); }
        if (comp) *craw_at((wave * T + t) * 16 + le) = v4f{o4.x, o4.y, o4.z, o4.w};
        if (valid && a.raw) reinterpret_cast<float4*>(a.raw)[sidx] = o4;
      });
    }
    if constexpr (!EMBEDDED) {
      if (comp && a.m_chunks) {
        // Rays of m = N / 64 chunks (N = 128, 192, ...): a ray's chunks sit on different waves, possibly in different groups
        // of the workgroup's run.  Three phases around two s_barriers, the arithmetic of raw2outputs_kernel's multi-chunk
        // loop (ns_composite_ray.h: chunk_local, then T = carry * excl with the carry multiplied up chunk by chunk, every
        // chunk's sums reduced on their own and added in chunk order):
        //   1  every chunk on its wave: alpha, colours, the chunk's own transmittance scan; its factor P_c -> LDS
        //   2  carry entering the chunk = (the open ray's carry, if the ray began in an earlier group) x P of the ray's
        //      earlier chunks of this group, in order; weights; the chunk's five sums -> LDS
        //   3  lane c of the last wave, for the chunk c that ends a ray or the group: totals in chunk order; a finished ray is written,
        //      the ray that stays open hands {carry, sums} to the next group
        const int m = a.m_chunks;
        const int64_t C0 = grp * T;                      // global index of the group's first chunk
        constexpr int NCW = (T + NWAVES - 1) / NWAVES;   // chunks a wave may own
        nscomp::ChunkLocal L[NCW];
        float zc[NCW];
        bool okc[NCW];
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave's raw records are in LDS
        static_for<NCW>([&](auto ci_) {
          constexpr int ci = decltype(ci_)::value;
          const int c = wave + NWAVES * ci;
          if (c < T) {                                   // (wave-uniform)
            const int i = c * 64 + le;
            const int64_t s_ = grp * GS + i;
            okc[ci] = s_ < a.S;
            const v4f qv = *craw_at(i);
            const v2f zd = *czd_at(par, i);
            float4 q = make_float4(qv.x, qv.y, qv.z, qv.w);
            if (a.sig_last) {                            // the guard pass's sigma for the ray's last sample
              const int x = (gi * GS) % a.N + i;         // position counted from the start of the group's first ray
              const int k = (x >= a.N) + (x >= 2 * a.N) + (x >= 3 * a.N);
              if (x - k * a.N == a.N - 1) q.w = *csig_at(par, k);
            }
            L[ci] = nscomp::chunk_local<64>(okc[ci], le, q, zd.y, 1.0f, 0.0f, false);
            zc[ci] = zd.x;
            if (le == 63) *cscal_at(kCP + c) = L[ci].p;
          }
        });
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        static_for<NCW>([&](auto ci_) {
          constexpr int ci = decltype(ci_)::value;
          const int c = wave + NWAVES * ci;
          if (c < T) {
            const int pos = static_cast<int>((C0 + c) % m);          // the chunk's position in its ray
            const int first = c - pos;                               // the ray's first chunk, as an index of this group
            float carry = first < 0 ? *cscal_at(kOPEN + 8 * par) : 1.0f;
            for (int cc = first < 0 ? 0 : first; cc < c; ++cc) carry = carry * *cscal_at(kCP + cc);
            const float Tr = carry * L[ci].excl;
            const float w = L[ci].alpha * Tr;
            const int64_t s_ = grp * GS + c * 64 + le;
            if (okc[ci] && a.weights) a.weights[s_] = w;
            nscomp::RayAccum A;
            if (okc[ci]) {
              A.r += w * L[ci].cr; A.g += w * L[ci].cg; A.b += w * L[ci].cb;
              A.depth += w * zc[ci];
              A.acc += w;
            }
            nscomp::reduce_sums<64>(A, le);
            if (le == 63) {
              *cscal_at(kCS + 8 * c + 0) = A.r; *cscal_at(kCS + 8 * c + 1) = A.g; *cscal_at(kCS + 8 * c + 2) = A.b;
              *cscal_at(kCS + 8 * c + 3) = A.depth; *cscal_at(kCS + 8 * c + 4) = A.acc;
            }
          }
        });
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (wave == NWAVES - 1 && le < T) {       // (the last wave: with five tiles wave 0 has had two chunks in phases 1 and 2, this one one)
          const int c = le;
          const int pos = static_cast<int>((C0 + c) % m);
          const bool ends = pos == m - 1;
          if (ends || c == T - 1) {
            const int first = c - pos;
            nscomp::RayAccum tot;                        // tot.carry: the transmittance behind chunk c
            if (first < 0) {
              tot.carry = *cscal_at(kOPEN + 8 * par); tot.r = *cscal_at(kOPEN + 8 * par + 1); tot.g = *cscal_at(kOPEN + 8 * par + 2);
              tot.b = *cscal_at(kOPEN + 8 * par + 3); tot.depth = *cscal_at(kOPEN + 8 * par + 4); tot.acc = *cscal_at(kOPEN + 8 * par + 5);
            }
            for (int cc = first < 0 ? 0 : first; cc <= c; ++cc) {
              tot.carry = tot.carry * *cscal_at(kCP + cc);
              tot.r = tot.r + *cscal_at(kCS + 8 * cc); tot.g = tot.g + *cscal_at(kCS + 8 * cc + 1); tot.b = tot.b + *cscal_at(kCS + 8 * cc + 2);
              tot.depth = tot.depth + *cscal_at(kCS + 8 * cc + 3); tot.acc = tot.acc + *cscal_at(kCS + 8 * cc + 4);
            }
            if (ends) {
              const int64_t r = (C0 + c) / m;
              if (r * a.N < a.S) {
                float disp;
                nscomp::finish_totals(tot, a.white_bkgd, disp);
                float* prgb = a.rgb + r * a.rgb_stride;
                prgb[0] = tot.r; prgb[1] = tot.g; prgb[2] = tot.b;
                a.disp[r * a.disp_stride] = disp;
              }
            } else {                                     // the ray goes on in the workgroup's next group
              const uint32_t np = par ^ 1u;
              *cscal_at(kOPEN + 8 * np) = tot.carry; *cscal_at(kOPEN + 8 * np + 1) = tot.r; *cscal_at(kOPEN + 8 * np + 2) = tot.g;
              *cscal_at(kOPEN + 8 * np + 3) = tot.b; *cscal_at(kOPEN + 8 * np + 4) = tot.depth; *cscal_at(kOPEN + 8 * np + 5) = tot.acc;
            }
          }
        }
      } else if (comp) {
        // raw2outputs in the epilogue (sampling_trainer.py:153-230): the group's samples are T chunks of 64 consecutive
        // samples -- whole rays (N <= 64, a power of two) -- one chunk per wave pass, composited by the lane-level code the
        // stand-alone kernel runs (ns_composite_ray.h: same operations in the same order, so bit-identical to it).
        // Four tiles: a wave's chunk is its own 64 samples, the wave's own LDS order suffices; five tiles: chunks straddle
        // waves, one s_barrier (all four waves reach it: the group loop is workgroup-uniform).
        if constexpr (T == 4) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        auto composite = [&](auto sw_, int c) {
          constexpr int SW = decltype(sw_)::value;
          const int i = c * 64 + le;
          const int64_t s = grp * GS + i;
          const bool ok = s < a.S;
          const v4f qv = *craw_at(i);
          const v2f zd = *czd_at(par, i);
          float4 q = make_float4(qv.x, qv.y, qv.z, qv.w);
          if (a.sig_last && (le & (SW - 1)) == SW - 1) q.w = *csig_at(par, i / SW);      // the guard pass's sigma_last
          nscomp::RayAccum A, tree;
          float alpha, w, disp, Tr;
          const int sub = le & (SW - 1);
          nscomp::composite_chunk<SW>(A, ok, sub, q, zd.x, zd.y, 1.0f, 0.0f, false, alpha, w, &Tr);
          if (ok && a.weights) a.weights[s] = w;
          nscomp::composite_finish<SW>(A, a.white_bkgd, disp, sub, &tree);
          if (ok && sub == SW - 1) {
            const int64_t r = s / SW;      // N == SW
            float* prgb = a.rgb + r * a.rgb_stride;
            prgb[0] = A.r; prgb[1] = A.g; prgb[2] = A.b;
            a.disp[r * a.disp_stride] = disp;
            if (a.fix_rec && __builtin_fabsf(q.w) < a.fix_thr) {      // (a NaN sigma compares false: a NaN ray stays NaN)
              float* rec = a.fix_rec + static_cast<size_t>(atomicAdd(a.fix_count, 1u)) * 16;
              reinterpret_cast<float4*>(rec)[0] = make_float4(tree.r, tree.g, tree.b, tree.depth);
              reinterpret_cast<float4*>(rec)[1] = make_float4(tree.acc, Tr, q.x, q.y);
              reinterpret_cast<float4*>(rec)[2] = make_float4(q.z, zd.x, zd.y, __builtin_bit_cast(float, static_cast<uint32_t>(r)));
              rec[12] = __builtin_bit_cast(float, static_cast<uint32_t>(static_cast<uint64_t>(r) >> 32));
            }
          }
        };
        for (int c = wave; c < T; c += NWAVES) {
          switch (a.N) {
            case 64: composite(std::integral_constant<int, 64>{}, c); break;
            case 32: composite(std::integral_constant<int, 32>{}, c); break;
            case 16: composite(std::integral_constant<int, 16>{}, c); break;
            case 8: composite(std::integral_constant<int, 8>{}, c); break;
            case 4: composite(std::integral_constant<int, 4>{}, c); break;
            default: composite(std::integral_constant<int, 2>{}, c); break;
          }
        }
      }
    }
  }
  ring.finish();
}

int ob16_program_slabs(int W, int D, uint32_t skip_mask, int use_viewdirs) {
  const int NSB = W / 16, NKB = W / 32, dp = kOb16Depth;
  int n = ob16_layer_slabs(NSB, 2, dp);
  for (int l = 1; l < D; ++l) n += ob16_layer_slabs(NSB, ((skip_mask >> (l - 1)) & 1u) ? NKB + 2 : NKB, dp);
  if (use_viewdirs) n += ob16_layer_slabs(NSB / 2 + 1, NKB + 1, dp) + ob16_layer_slabs(1, NKB / 2, dp);
  else n += ob16_layer_slabs(1, NKB, dp);
  return n;
}

template <class M, int NKB, bool EMB, bool PROD = false, int TT = kT>
int launch(Nerf16Args& a, hipStream_t stream) {
  const size_t lds = static_cast<size_t>(Pipe<M, kWaves, 0, kOb16Depth, kOb16Ahead>::kLdsBytes) +
                     ((static_cast<size_t>(a.bias_floats) * 4 + 15) & ~size_t(15)) + static_cast<size_t>(kWaves) * TT * 3 * 1024 +
                     static_cast<size_t>(kWaves) * 11 * (TT * 64) +      // ring | bias | embedding stash | input staging
                     (a.comp && !EMB ? static_cast<size_t>(kWaves) * TT * 16 * 36 + 512 : 0);   // | compositing records (32 B per sample + 8 B per ray pair) + chunk scalars
  if (lds > 160 * 1024) {
    ns::set_error("ns_nerf_forward: %zu bytes of LDS needed (too deep a network for the resident bias image)", lds);
    return NS_E_UNSUPPORTED;
  }
  auto kern = nerf_mlp_ob16_kernel<M, NKB, EMB, PROD, TT>;
  NS_HIP(ns::ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds));
  const int64_t n_tiles = (a.S + 15) / 16;
  const int64_t n_groups = (n_tiles + kWaves * TT - 1) / (kWaves * TT);
  int cus = ns::cu_count();
  if (cus <= 0) cus = 256;
  a.sg_groups = 1;
  if (a.m_chunks) {          // runs of lcm(group samples, N) / group samples consecutive groups: whole rays per run
    const int gs = kWaves * TT * 16;
    int x = gs, y = a.N;
    while (y) { const int t = x % y; x = y; y = t; }
    a.sg_groups = a.N / x;   // lcm(gs, N) / gs
  }
  const int64_t n_runs = (n_groups + a.sg_groups - 1) / a.sg_groups;
  const int grid = static_cast<int>(n_runs < cus ? n_runs : cus);
  kern<<<grid, kWaves * 64, lds, stream>>>(a);
  NS_LAUNCH_CHECK();
  return NS_OK;
}

#ifndef NS_OB16_TU_T5
template <class M, bool EMB>
int dispatch_m(const ns_weights* net, Nerf16Args& a, hipStream_t stream) {
#if NS_NERF16_T == 4 && NS_NERF16_WAVES == 4
  if (net->width == 256 && net->depth == 8 && net->skip_mask == (1u << 4) && net->use_viewdirs && !ns::debug_flags().generic_kernels) {
    // the production network: hand-scheduled layers, four or five 16-sample tiles per wave.  Five tiles read 20 % fewer
    // weight fragments and refill bytes per sample (-2.5 % per frame on the final build, profiles/r03c_ab_tiles_final_build.log);
    // the persistent grid runs ceil(groups / CUs) rounds of 256 (320) samples per workgroup, so the choice is made per
    // launch on the rounds' total: a 32768-ray x 64 chunk is exactly 32 rounds of four tiles but 25.6 -> 26 of five.
    int cus = ns::cu_count();
    if (cus <= 0) cus = 256;
    auto rounds = [&](int64_t per_group) { const int64_t g = (a.S + per_group - 1) / per_group; return (g + cus - 1) / cus; };
    const double t4 = static_cast<double>(rounds(kWaves * 4 * 16)) * 4.0, t5 = static_cast<double>(rounds(kWaves * 5 * 16)) * 5.0 * 0.975;
    int tiles = NS_OB16_PROD_T ? NS_OB16_PROD_T : (t5 < t4 ? 5 : 4);
    if (ns::prod_tiles_hint()) tiles = ns::prod_tiles_hint();               // the renderer's hint (host copies in flight)
    if (ns::debug_flags().prod_tiles) tiles = ns::debug_flags().prod_tiles;   // diagnostic override (ns_debug_set)
    if (tiles == 5) return nsob16::launch_prod_t5(M::kDtype, EMB, a, stream);
    return launch<M, 8, EMB, true, 4>(a, stream);
  }
#endif
  return net->width == 256 ? launch<M, 8, EMB>(a, stream) : launch<M, 4, EMB>(a, stream);
}
#endif

}  // namespace

#ifdef NS_OB16_TU_T5
int nsob16::launch_prod_t5(int dtype, bool embedded, Nerf16Args& a, hipStream_t stream) {
  if (dtype == Mma16BF16::kDtype) return embedded ? launch<Mma16BF16, 8, true, true, 5>(a, stream) : launch<Mma16BF16, 8, false, true, 5>(a, stream);
  return embedded ? launch<Mma16F16, 8, true, true, 5>(a, stream) : launch<Mma16F16, 8, false, true, 5>(a, stream);
}
#else
int ns_nerf_forward_x3(const ns_weights* net, const float* pts_dev, const float* o_dev, const float* d_dev,
                       const float* z_dev, const float* viewdirs_dev, const float* x90_dev, int64_t S, int N,
                       float* raw_dev, hipStream_t stream, const uint32_t* count_dev);

// which (network, sample count) pairs the kernel composites itself (see Nerf16Args::comp)
bool ns_nerf_can_composite(const ns_weights* net, int N) {
  return net && net->kind == NS_KIND_NERF && net->layout == 16 && (net->dtype == NS_DTYPE_BF16 || net->dtype == NS_DTYPE_F16) &&
         net->use_viewdirs && net->out_ch == 4 &&
         ((N >= 2 && N <= 64 && (N & (N - 1)) == 0) || (N > 64 && N % 64 == 0 && N <= 512));
}

// called by ns_nerf_forward / ns_nerf_forward_embedded for handles packed with layout 16 (arguments validated there)
int ns_nerf_forward_ob16(const ns_weights* net, const float* pts_dev, const float* o_dev, const float* d_dev,
                         const float* z_dev, const float* viewdirs_dev, const float* x90_dev, int64_t S, int N,
                         float* raw_dev, hipStream_t stream, const ns_composite_args* comp) {
  if (comp) {
    if (!ns_nerf_can_composite(net, N)) {
      ns::set_error("in-kernel compositing needs a 16-bit NeRF handle with view directions and N a power of two in [2, 64] or a "
                    "multiple of 64 up to 512 (N = %d)", N);
      return NS_E_UNSUPPORTED;
    }
    if (pts_dev || x90_dev || !(o_dev && d_dev) || !(z_dev || comp->mean_dev) || !(comp->rgb_dev && comp->disp_dev)) {
      ns::set_error("in-kernel compositing needs rays (o, d), depths (z or mean) and the rgb / disp outputs");
      return NS_E_INVALID;
    }
  }
  if (net->dtype == NS_DTYPE_F16X3)   // split fp16 operands: ns_nerf_mlp_x3.hip
    return ns_nerf_forward_x3(net, pts_dev, o_dev, d_dev, z_dev, viewdirs_dev, x90_dev, S, N, raw_dev, stream, nullptr);
  if (ob16_program_slabs(net->width, net->depth, net->skip_mask, net->use_viewdirs) != static_cast<int>(net->n_slabs)) {
    ns::set_error("ns_nerf_forward: packed stream has %u slabs, kernel program expects %d", net->n_slabs,
                  ob16_program_slabs(net->width, net->depth, net->skip_mask, net->use_viewdirs));
    return NS_E_INVALID;
  }
  Nerf16Args a{};
  a.stream = static_cast<const char*>(net->stream_dev);
  a.bias = net->bias_dev; a.n_slabs = net->n_slabs; a.bias_floats = net->bias_floats;
  a.D = net->depth; a.skip_mask = net->skip_mask; a.use_viewdirs = net->use_viewdirs; a.out_ch = net->out_ch;
  a.x_stride = net->use_viewdirs ? 90 : 63;
  a.pts = pts_dev; a.o = o_dev; a.d = d_dev; a.z = z_dev; a.viewdirs = viewdirs_dev; a.x90 = x90_dev;
  a.S = S; a.N = N; a.raw = raw_dev;
  if (comp) {
    a.comp = comp->mean_dev ? 2 : 1;
    a.mean = comp->mean_dev; a.std_ = comp->std_; a.lin_step = nsplace::linspace_step_of(-comp->std_, comp->std_, N - 1);
    a.n_shift = -1;
    for (int k = 0; k < 31; ++k) if (N == (1 << k)) a.n_shift = k;
    a.m_chunks = N > 64 ? N / 64 : 0;
    a.white_bkgd = comp->white_bkgd;
    a.rgb = comp->rgb_dev; a.rgb_stride = comp->rgb_stride; a.disp = comp->disp_dev; a.disp_stride = comp->disp_stride;
    a.weights = comp->weights_dev; a.z_out = comp->z_out_dev; a.pts_out = comp->pts_out_dev;
    if (comp->fix_rec_dev) {
      if (N > 64 || comp->sigma_last_dev) {
        ns::set_error("the selective guard serves rays of one chunk (N <= 64) and excludes the every-ray guard's sigma array");
        return NS_E_INVALID;
      }
      a.fix_thr = comp->fix_thr; a.fix_count = comp->fix_count_dev; a.fix_rec = comp->fix_rec_dev;
    }
    a.sig_last = comp->sigma_last_dev;
  }
  const bool emb = x90_dev != nullptr;
#ifdef NS_OB16_VARIANT_BUILD   // tools/build_asm_variant.sh: only the kernel under test is instantiated (a 20 s build)
  if (net->dtype == NS_DTYPE_BF16 && !emb && net->width == 256 && net->depth == 8 && net->skip_mask == (1u << 4) && net->use_viewdirs)
    return launch<Mma16BF16, 8, false, true, (NS_OB16_PROD_T ? NS_OB16_PROD_T : 4)>(a, stream);
  return NS_E_UNSUPPORTED;
#endif
  if (net->dtype == NS_DTYPE_BF16) return emb ? dispatch_m<Mma16BF16, true>(net, a, stream) : dispatch_m<Mma16BF16, false>(net, a, stream);
  if (net->dtype == NS_DTYPE_F16) return emb ? dispatch_m<Mma16F16, true>(net, a, stream) : dispatch_m<Mma16F16, false>(net, a, stream);
  return NS_E_UNSUPPORTED;
}
#endif  // NS_OB16_TU_T5
