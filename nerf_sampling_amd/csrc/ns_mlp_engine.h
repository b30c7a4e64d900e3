// Fused-MLP engine for gfx950: activations stay in registers, weights stream through LDS.
//
// Design (MI355X-first, see DESIGN.md "MLP engine"):
//   * One wave owns a tile of 32 samples for the whole network.  Every layer is computed
//     TRANSPOSED, Y^T[n, m] = W[n, k] . X^T[k, m], with v_mfma_f32_32x32x{16_bf16,16_f16,2_f32}:
//     the weight block is the A operand (32 output features x K), the activations are the B
//     operand (K x 32 samples).  A 32x32 fp32 result then has its sample on the lane and its
//     feature index in the 16 accumulator registers, which is exactly the B-operand layout of
//     the next layer's MFMA -- activations never leave the register file, there is no LDS
//     round trip and no barrier between layers.
//   * Weights are pre-packed on the host into 1-KiB "chunks" laid out lane-linearly in the
//     exact order the MFMAs consume them (ns_pack.hip), so that one
//     global_load_lds_dwordx4 per wave-instruction lands a chunk in LDS and one conflict-free
//     ds_read_b128 per lane fetches an A fragment.  Chunks are grouped into 16-KiB slabs; all
//     waves of a workgroup walk the same slab sequence through a ring of 4 LDS slots with ONE
//     s_barrier per slab and counted s_waitcnt vmcnt(N) (never 0 in the loop): while slab t feeds
//     the matrix cores, slab t+1 has landed (so A fragments are read 4 chunks ahead of their MFMA,
//     across slab seams), slab t+2 is in flight and slab t+3 is being issued.
//   * Within a 32-feature block, lane half h (= lane >> 5) holds the 16 features
//     k = 32*blk + (q & 3) + 8*(q >> 2) + 4*h, q = 0..15  (the MFMA C/D register map).  The
//     host packer applies the same map to the weight columns, so any per-lane assignment of
//     embedding features to (blk, q, h) is legal as long as both sides agree.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>
#include <utility>

namespace nsmlp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));


constexpr int kChunkBytes = 1024;
#ifndef NS_SLAB_CHUNKS
#define NS_SLAB_CHUNKS 16
#endif
constexpr int kSlabChunks = NS_SLAB_CHUNKS;
constexpr int kSlabBytes = kChunkBytes * kSlabChunks;
constexpr int kRingBase = 4;     // LDS slots without stagger: open, landed (read-ahead), in flight, being issued
#ifndef NS_FRAG_DEPTH
#define NS_FRAG_DEPTH 2
#endif
constexpr int kFragDepth = NS_FRAG_DEPTH;   // A fragments kept in flight per wave (LDS read-ahead, in chunks)

// ---- compile-time loop with constant indices (keeps register arrays statically indexed) ----
template <int... I, class F>
__host__ __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__host__ __device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

// ---- operand traits --------------------------------------------------------------------------
// CPB = chunks per 32-feature input block; Block = one lane's 16 features of a 32-feature block.
struct MmaBF16 {
  static constexpr int kDtype = 1;
  static constexpr int CPB = 2;
  static constexpr int kElemBytes = 2;
  struct Block { bf16x8 v[2]; };
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  // one dword = two consecutive features.  hipcc (ROCm 7.2) lowers a {(__bf16)a, (__bf16)b} pair to two
  // single-value v_cvt_pk_bf16_f32 + a v_perm_b32; the packed form is ONE instruction (RNE, NaN-preserving).
  // ReLU on the packed pair is a signed 16-bit max with 0 (sign-magnitude floats: negative <=> negative int16).
  template <bool RELU>
  __device__ static __forceinline__ uint32_t pack2(float a, float b) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const f32x2 ab = {a, b};
    uint32_t r = __builtin_bit_cast(uint32_t, __builtin_convertvector(ab, bf16x2));   // one v_cvt_pk_bf16_f32
    if constexpr (RELU) {
      s16x2 q = __builtin_bit_cast(s16x2, r);
      q = __builtin_elementwise_max(q, (s16x2)(0));
      r = __builtin_bit_cast(uint32_t, q);
    }
    return r;
  }
  template <bool RELU = false>
  __device__ static __forceinline__ void from_f32(Block& b, const float (&x)[16]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      u32x4 w = {pack2<RELU>(x[8 * i], x[8 * i + 1]), pack2<RELU>(x[8 * i + 2], x[8 * i + 3]),
                 pack2<RELU>(x[8 * i + 4], x[8 * i + 5]), pack2<RELU>(x[8 * i + 6], x[8 * i + 7])};
      b.v[i] = __builtin_bit_cast(bf16x8, w);
    }
  }
  using AFrag = bf16x8;
  template <int SUB>
  __device__ static __forceinline__ void mma(f32x16& acc, const AFrag& a, const Block& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b.v[SUB], acc, 0, 0, 0);
  }
};

struct MmaF16 {
  static constexpr int kDtype = 2;
  static constexpr int CPB = 2;
  static constexpr int kElemBytes = 2;
  struct Block { f16x8 v[2]; };
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  template <bool RELU>
  __device__ static __forceinline__ uint32_t pack2(float a, float b) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const f32x2 ab = {a, b};
    uint32_t r = __builtin_bit_cast(uint32_t, __builtin_convertvector(ab, f16x2));
    if constexpr (RELU) {
      s16x2 q = __builtin_bit_cast(s16x2, r);
      q = __builtin_elementwise_max(q, (s16x2)(0));
      r = __builtin_bit_cast(uint32_t, q);
    }
    return r;
  }
  template <bool RELU = false>
  __device__ static __forceinline__ void from_f32(Block& b, const float (&x)[16]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      u32x4 w = {pack2<RELU>(x[8 * i], x[8 * i + 1]), pack2<RELU>(x[8 * i + 2], x[8 * i + 3]),
                 pack2<RELU>(x[8 * i + 4], x[8 * i + 5]), pack2<RELU>(x[8 * i + 6], x[8 * i + 7])};
      b.v[i] = __builtin_bit_cast(f16x8, w);
    }
  }
  using AFrag = f16x8;
  template <int SUB>
  __device__ static __forceinline__ void mma(f32x16& acc, const AFrag& a, const Block& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b.v[SUB], acc, 0, 0, 0);
  }
};

struct MmaF32 {
  static constexpr int kDtype = 0;
  static constexpr int CPB = 4;
  static constexpr int kElemBytes = 4;
  struct Block { float v[16]; };
  template <bool RELU = false>
  __device__ static __forceinline__ void from_f32(Block& b, const float (&x)[16]) {
#pragma unroll
    for (int j = 0; j < 16; ++j) b.v[j] = RELU ? (x[j] < 0.0f ? 0.0f : x[j]) : x[j];   // NaN stays NaN, as torch.relu (fmaxf would drop it)
  }
  __device__ static __forceinline__ void relu_packed(Block&) {}
  using AFrag = f32x4;
  template <int SUB>
  __device__ static __forceinline__ void mma(f32x16& acc, const AFrag& a, const Block& b) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b.v[4 * SUB + e], acc, 0, 0, 0);
  }
};

// slabs a segment of NBLK input blocks occupies when feeding NBO output blocks
__host__ __device__ constexpr int seg_slabs(int cpb, int nbo, int nblk) {
  const int kps = kSlabChunks / nbo;
  return (nblk * cpb + kps - 1) / kps;
}

// ---- LDS weight ring --------------------------------------------------------------------------
#define NS_LDS_PTR(p) ((void __attribute__((address_space(3)))*)(p))
#define NS_GLB_PTR(p) ((const void __attribute__((address_space(1)))*)(p))

// LDS-DMA of 16 (or 4) bytes per lane: LDS address = wave-uniform base (M0) + lane * size, global address per lane.
// M0 is compiler-reserved: it cannot be declared as a clobber (hipcc only warns and does not honour it), so every
// statement that writes it saves the old value into a scratch SGPR first and restores it before it ends
// (cdna_hip_programming.md section 5.7): the code generator may keep whatever it likes in M0 around these statements.
__device__ __forceinline__ void lds_dma16(const void* gptr, uint32_t lds_base) {
  const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds_base);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(m0v), "v"(gptr) : "memory");
}
__device__ __forceinline__ void lds_dma4(const void* gptr, uint32_t lds_base) {
  const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds_base);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %2, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(m0v), "v"(gptr) : "memory");
}

// Ring of kRingDepth LDS slots + a kFragDepth-deep register pipeline of A fragments.
//
// Slab timeline (t = slab being multiplied):  slot t%4 is read by the MFMAs, slot (t+1)%4 has landed
// and feeds the fragment read-ahead across the slab seam, slab t+2 is in flight, and slab t+3 is
// issued right after the barrier that opens slab t into the slot slab t-1 just vacated.  Every DS
// read is issued kFragDepth chunks before the MFMA that consumes it, so neither the LDS latency nor
// the barrier sits on the MFMA critical path.
//
// LAG > 0 staggers the second half of the workgroup's waves (the SIMD partners of the first half:
// MI355X_MICROARCH.md "Two waves per SIMD", item 9) LAG slabs behind the first half, so that one
// partner's layer epilogue / tile prologue (VALU, global loads) runs under the other's MFMAs instead
// of both leaving the matrix pipe idle at once.  It costs LAG more ring slots.
template <class M, int NWAVES, int LAG = 0, int DEPTH = kFragDepth, int AHEAD = kRingBase - 1>
struct Pipe {
  static_assert(AHEAD >= 2, "the slab after the open one must have landed, one more must be in flight");
  static constexpr int kDepth = DEPTH;   // A fragments in flight (LDS read-ahead, in chunks)
  using AFrag = typename M::AFrag;
  static constexpr int LPW = kSlabChunks / NWAVES;  // DMA instructions per wave per slab
  static constexpr int RING = AHEAD + 1 + LAG;   // slabs issued ahead of the open one, + the open one
  static constexpr int kLdsBytes = RING * kSlabBytes;
  const char* stream;   // device weight stream, n_slabs * 16 KiB, cyclic
  char* lds;            // ring base in LDS
  uint32_t n_slabs;
  uint32_t issue_slab;  // next slab of the stream to fetch
  uint32_t issue_slot;
  uint32_t read_slot;   // slot of the slab the next begin_slab() opens
  int wave, lane;
  uint32_t lds_off;     // LDS byte address of the ring base
  uint32_t cur;         // this lane's LDS byte address in the open slab (chunk c at +c*1024)
  uint32_t nxt;         // ... and in the following one
  AFrag f[DEPTH];       // fragments of the next DEPTH chunks

  __device__ static __forceinline__ uint32_t next_slot(uint32_t slot) {
    if constexpr ((RING & (RING - 1)) == 0) return (slot + 1) & (RING - 1);   // one s_and instead of compare + select
    else return (slot + 1 == RING) ? 0u : slot + 1;
  }

  // One slab: LPW LDS-DMA instructions per wave.  They are issued through inline asm on purpose: with the
  // __builtin_amdgcn_global_load_lds form the compiler's waitcnt pass sees an LDS store it cannot disambiguate and
  // puts an s_waitcnt vmcnt(0) in front of the next LDS read of EVERY slab step -- the wave then waits out the L2
  // round trip of the slab it has just requested, three slabs early, and the ring prefetches nothing (measured:
  // MFMA pipe busy 52-69 % with it).  The asm form leaves the ordering to the counted wait + barrier of begin_slab().
  __device__ __forceinline__ void issue() {
    // One M0 write per slab: wave w fetches the LPW CONSECUTIVE chunks w*LPW .. w*LPW+LPW-1 with the SGPR-base form of the
    // instruction (address = s[base] + 32-bit lane offset + immediate), the immediate offset stepping through both the
    // global and the LDS address.  Saves a 64-bit VALU add, an M0 write and its wait state per piece.
    if constexpr (LPW * kChunkBytes <= 4096) {
      {
        const char* src = stream + static_cast<size_t>(issue_slab) * kSlabBytes + wave * (LPW * kChunkBytes);
        const uint32_t dst = lds_off + issue_slot * kSlabBytes + wave * (LPW * kChunkBytes);
        const uint32_t lane_off = static_cast<uint32_t>(lane) * 16u;
        const uint32_t m0v = __builtin_amdgcn_readfirstlane(dst);
        const uint64_t base = reinterpret_cast<uint64_t>(src);
        const uint32_t blo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(base));
        const uint32_t bhi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(base >> 32));
        const uint64_t sbase = (static_cast<uint64_t>(bhi) << 32) | blo;
        static_assert(LPW == 1 || LPW == 2 || LPW == 4, "pieces per wave");
        uint32_t keep;      // M0 saved and restored inside the statement (see lds_dma16)
        if constexpr (LPW == 4)
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                       "global_load_lds_dwordx4 %2, %3\n\t"
                       "global_load_lds_dwordx4 %2, %3 offset:1024\n\t"
                       "global_load_lds_dwordx4 %2, %3 offset:2048\n\t"
                       "global_load_lds_dwordx4 %2, %3 offset:3072\n\t"
                       "s_mov_b32 m0, %0" : "=&s"(keep) : "s"(m0v), "v"(lane_off), "s"(sbase) : "memory");
        else if constexpr (LPW == 2)
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                       "global_load_lds_dwordx4 %2, %3\n\t"
                       "global_load_lds_dwordx4 %2, %3 offset:1024\n\t"
                       "s_mov_b32 m0, %0" : "=&s"(keep) : "s"(m0v), "v"(lane_off), "s"(sbase) : "memory");
        else
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "s"(m0v), "v"(lane_off), "s"(sbase) : "memory");
      }
      issue_slab = (issue_slab + 1 == n_slabs) ? 0u : issue_slab + 1;
      issue_slot = next_slot(issue_slot);
      return;
    }
    // wave-uniform part of the address in SGPRs, per-lane part a constant 32-bit offset (lane * 16)
    const char* src = stream + static_cast<size_t>(issue_slab) * kSlabBytes + wave * kChunkBytes;
    const uint32_t dst = lds_off + issue_slot * kSlabBytes + wave * kChunkBytes;
    const uint32_t lane_off = static_cast<uint32_t>(lane) * 16u;
#pragma unroll
    for (int i = 0; i < LPW; ++i) lds_dma16(src + i * NWAVES * kChunkBytes + lane_off, dst + i * NWAVES * kChunkBytes);
    issue_slab = (issue_slab + 1 == n_slabs) ? 0u : issue_slab + 1;
    issue_slot = next_slot(issue_slot);
  }

  // s_waitcnt vmcnt(N) alone (the other counters at "don't wait")
  template <int N>
  __device__ static __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "unexpected DMA count");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
  }

  // A-fragment read: a plain LDS load (ds_read_b128 from a lane-linear image, conflict-free).  Measured
  // alternative (round 1): inline-asm reads with hand-counted lgkmcnt(3) instead of the compiler's
  // lgkmcnt(0) every fourth MFMA -- same kernel time (35.4 vs 35.2 ms), so the compiler-visible form stays.
  template <int OFF>
  __device__ __forceinline__ void load(AFrag& dst, uint32_t addr) const {
    dst = *reinterpret_cast<const AFrag __attribute__((address_space(3)))*>(static_cast<uintptr_t>(addr + OFF));
  }
  __device__ static __forceinline__ void wait_frag(AFrag&) {}

  __device__ __forceinline__ void init(const char* stream_, char* lds_, uint32_t n_slabs_, int wave_, int lane_) {
    stream = stream_; lds = lds_; n_slabs = n_slabs_; wave = wave_; lane = lane_;
    issue_slab = 0; issue_slot = 0; read_slot = 0;
    lds_off = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(NS_LDS_PTR(lds_)));
    static_for<AHEAD>([&](auto) { issue(); });    // slabs 0 .. AHEAD-1
    wait_vm<(AHEAD - 1) * LPW>();                 // my pieces of slab 0 have landed ...
    __builtin_amdgcn_s_barrier();                 // ... and everyone else's
    asm volatile("" ::: "memory");
    lds_off = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(NS_LDS_PTR(lds_)));
    nxt = lds_off + lane * 16;                    // slab 0 is "the following slab" until it is opened
    cur = nxt;
    static_for<DEPTH>([&](auto i_) { load<decltype(i_)::value * kChunkBytes>(f[decltype(i_)::value], nxt); });
    if constexpr (LAG > 0) {
      if (wave >= NWAVES / 2) {                   // trailing half: sit out the first LAG slabs
#pragma unroll 1
        for (int i = 0; i < LAG; ++i) idle_slab();
      }
    }
  }

  // take part in a slab step (wait, barrier, DMA issue) without opening a slab
  __device__ __forceinline__ void idle_slab() {
    wait_vm<(AHEAD - 2) * LPW>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue();
  }

  // after the last tile: the leading half keeps the slab steps going until the trailing half is done
  __device__ __forceinline__ void finish() {
    if constexpr (LAG > 0) {
      if (wave < NWAVES / 2) {
#pragma unroll 1
        for (int i = 0; i < LAG; ++i) idle_slab();
      }
    }
    drain();
  }

  // Open the next slab: its first kFragDepth fragments are already in registers.
  __device__ __forceinline__ void begin_slab() {
    wait_vm<(AHEAD - 2) * LPW>();                 // my pieces of the slab AFTER this one have landed
    __builtin_amdgcn_s_barrier();                 // everyone's; all waves are done with the previous slab
    asm volatile("" ::: "memory");
    issue();                                      // refill the slot the previous slab occupied
    cur = lds_off + read_slot * kSlabBytes + lane * 16;
    read_slot = next_slot(read_slot);
    nxt = lds_off + read_slot * kSlabBytes + lane * 16;
  }

  __device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
};

// ---- one K-segment of a layer: acc[NBO] += W[:, segment] . in[NBLK] -----------------------------
template <class M, int NBO, int NBLK, class PipeT>
__device__ __forceinline__ void consume(PipeT& pipe, f32x16 (&acc)[NBO],
                                        const typename M::Block (&in)[NBLK]) {
  constexpr int KPS = kSlabChunks / NBO;         // chunk rows (K steps) per slab
  constexpr int CHUNKS = NBLK * M::CPB;          // real chunk rows of this segment
  constexpr int SLABS = (CHUNKS + KPS - 1) / KPS;
  static_for<SLABS>([&](auto s_) {
    constexpr int s = decltype(s_)::value;
    constexpr int ROWS = (CHUNKS - s * KPS) < KPS ? (CHUNKS - s * KPS) : KPS;
    constexpr int USED = ROWS * NBO;             // chunks of this slab that carry weights (a prefix)
    static_assert(USED % kFragDepth == 0 && USED >= kFragDepth, "fragment pipeline needs USED % depth == 0");
    pipe.begin_slab();
    static_for<USED>([&](auto p_) {
      constexpr int p = decltype(p_)::value;
      constexpr int kk = p / NBO, nb = p % NBO;
      constexpr int kc = s * KPS + kk;
      PipeT::wait_frag(pipe.f[p % kFragDepth]);
      M::template mma<kc % M::CPB>(acc[nb], pipe.f[p % kFragDepth], in[kc / M::CPB]);
      if constexpr (p + kFragDepth < USED)
        pipe.template load<(p + kFragDepth) * kChunkBytes>(pipe.f[p % kFragDepth], pipe.cur);
      else
        pipe.template load<(p + kFragDepth - USED) * kChunkBytes>(pipe.f[p % kFragDepth], pipe.nxt);
    });
  });
}

// ---- bias init / activation epilogue ------------------------------------------------------------
// bias_lds: floats laid out [nb][h][16] for this layer
template <int NBO>
__device__ __forceinline__ void init_bias(f32x16 (&acc)[NBO], const float* bias_lds, int h) {
  static_for<NBO>([&](auto nb_) {
    constexpr int nb = decltype(nb_)::value;
    const f32x4* b = reinterpret_cast<const f32x4*>(bias_lds + nb * 32 + h * 16);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v = b[g];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[nb][4 * g + e] = v[e];
    }
  });
}

enum Act { kNone = 0, kRelu = 1, kLeaky = 2 };

template <class M, int ACT, int NBO>
__device__ __forceinline__ void to_blocks(typename M::Block (&out)[NBO], const f32x16 (&acc)[NBO]) {
  static_for<NBO>([&](auto nb_) {
    constexpr int nb = decltype(nb_)::value;
    float x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = acc[nb][r];
      if constexpr (ACT == kLeaky) v = v > 0.0f ? v : 0.01f * v;
      x[r] = v;
    }
    M::template from_f32<ACT == kRelu>(out[nb], x);
  });
}

// ---- output-block-major layers (16x16x32 engine below) ------------------------------------------
// The k-major consume<>() above finishes all NBO output blocks of a layer at the same MFMA, so the
// bias loads and f32 -> 16-bit conversions of a whole layer come in one burst.  The 16-bit NeRF kernel
// instead walks a layer one output sub-block at a time (all K chunks of sub-block sb, then sb+1; the
// stream is packed in that order by ns_pack.hip layout 16): only one sub-block's accumulators are live,
// so a wave can own several sample tiles that share every A fragment, and the conversion of sub-block sb
// is issued piecewise between the MFMAs of sub-block sb+1.  (Round 1 also had a 32x32x16 version of this
// engine, two 32-sample tiles per wave; the 16x16x32 one replaced it)

// op(P, frag, load_next) is called for chunk P = 0..TOTAL-1 of the stream, in order, with the chunk's A fragment;
// it must call load_next() exactly once (it re-fills the fragment register with the chunk DEPTH ahead), at the point
// of its instruction stream where the LDS read should issue.
template <int TOTAL, class PipeT, class F>
__device__ __forceinline__ void stream_chunks(PipeT& pipe, F&& op) {
  constexpr int SLABS = (TOTAL + kSlabChunks - 1) / kSlabChunks;
  static_for<SLABS>([&](auto s_) {
    constexpr int s = decltype(s_)::value;
    constexpr int USED = (TOTAL - s * kSlabChunks) < kSlabChunks ? (TOTAL - s * kSlabChunks) : kSlabChunks;
    static_assert(USED % PipeT::kDepth == 0 && USED >= PipeT::kDepth, "fragment pipeline needs USED % depth == 0");
    pipe.begin_slab();
    static_for<USED>([&](auto p_) {
      constexpr int p = decltype(p_)::value;
      // refill the slot of the PREVIOUS chunk (its MFMAs were all issued a step ago, so the LDS read does not have to
      // wait out the write-after-read window of an MFMA that is still fetching its A operand): chunk p + DEPTH - 1
      auto load_next = [&] {
        constexpr int q = p + PipeT::kDepth - 1;
        if constexpr (q < USED) pipe.template load<q * kChunkBytes>(pipe.f[q % PipeT::kDepth], pipe.cur);
        else pipe.template load<(q - USED) * kChunkBytes>(pipe.f[q % PipeT::kDepth], pipe.nxt);
      };
      op(std::integral_constant<int, s * kSlabChunks + p>{}, pipe.f[p % PipeT::kDepth], load_next);
    });
  });
}
// ---- positional-encoding slots -------------------------------------------------------------------
// Lane half h = 0 evaluates sines, h = 1 cosines of the same argument (cos x = sin(x + pi/2)),
// so both halves run the same instruction stream.
struct Rev {  // x / (2 pi) as an unevaluated fp32 pair, for the fast path
  float hi, lo;
};
__device__ __forceinline__ Rev to_rev(float x) {
  // 1/(2 pi) = 0.15915494309189535 = c_hi + c_lo with c_hi = fl32(1/(2 pi))
  constexpr float c_hi = 0.15915493667125702f;
  constexpr float c_lo = 6.4206382432985265e-09f;
  const float hi = x * c_hi;
  const float err = __builtin_fmaf(x, c_hi, -hi);  // exact rounding error of the product
  const float lo = __builtin_fmaf(x, c_lo, err);
  return {hi, lo};
}

// PRECISE = false: v_sin_f32 on the reduced argument (16-bit operand paths: its error is far
// below the operand rounding).  PRECISE = true: odd degree-11 polynomial on [-pi/2, pi/2], ~1e-7
// absolute -- the same class as the 1-ulp vector sin of the reference's CPU path.  Both reduce
// the argument exactly: frequencies are powers of two (run_nerf_helpers.py:32), so
// fract(2^L * x/(2 pi)) is computed without rounding from a two-float x/(2 pi).
template <bool PRECISE>
struct Trig {
  Rev r;
  __device__ __forceinline__ explicit Trig(float x_) : r(to_rev(x_)) {}
  // sin(2^level x) for h = 0, cos(2^level x) for h = 1
  __device__ __forceinline__ float operator()(int level, int h) const {
    const float scale = __builtin_ldexpf(1.0f, level);
    const float a = r.hi * scale;                    // exact (power of two)
    float f = __builtin_amdgcn_fractf(a);            // exact, in [0, 1)
    const float lo = r.lo * scale + 0.25f * static_cast<float>(h);
    if constexpr (!PRECISE) {
      return __builtin_amdgcn_sinf(f + lo);          // v_sin_f32 takes revolutions
    } else {
      f = (f >= 0.5f ? f - 1.0f : f) + lo;           // [-0.5, 0.75)
      f = f > 0.5f ? f - 1.0f : f;                   // [-0.5, 0.5]
      const float g = (f > 0.25f ? 0.5f - f : (f < -0.25f ? -0.5f - f : f));  // sin(pi - x) = sin x
      const float x = g * 6.283185307179586f;
      const float x2 = x * x;
      float p = -2.5052108385441720e-08f;            // -1/11!
      p = __builtin_fmaf(p, x2, 2.7557319223985893e-06f);
      p = __builtin_fmaf(p, x2, -1.9841269841269841e-04f);
      p = __builtin_fmaf(p, x2, 8.3333333333333332e-03f);
      p = __builtin_fmaf(p, x2, -1.6666666666666666e-01f);
      return __builtin_fmaf(p * x2, x, x);
    }
  }
};

// 3-component, L-level embedding into NBLK blocks: slot t = 16*blk + q;
//   t < 3L: level t/3, component t%3;  then the raw components (h=0: c0,c1; h=1: c2,pad)
template <class M, bool PRECISE, int L, int NBLK>
__device__ __forceinline__ void embed3(typename M::Block (&out)[NBLK], const float (&p)[3], int h) {
  const Trig<PRECISE> t0(p[0]), t1(p[1]), t2(p[2]);
  static_for<NBLK>([&](auto b_) {
    constexpr int b = decltype(b_)::value;
    float x[16];
    static_for<16>([&](auto q_) {
      constexpr int t = 16 * b + decltype(q_)::value;
      constexpr int q = decltype(q_)::value;
      if constexpr (t < 3 * L) {
        constexpr int c = t % 3;
        x[q] = (c == 0 ? t0 : (c == 1 ? t1 : t2))(t / 3, h);
      } else if constexpr (t == 3 * L) {
        x[q] = h ? p[2] : p[0];
      } else if constexpr (t == 3 * L + 1) {
        x[q] = h ? 0.0f : p[1];
      } else {
        x[q] = 0.0f;
      }
    });
    M::from_f32(out[b], x);
  });
}

__host__ __device__ inline int embed3_col(int k, int L);

// slot values for a pre-embedded input row (NeRF.forward on [M,90])
template <class M, int L, int NBLK>
__device__ __forceinline__ void gather3(typename M::Block (&out)[NBLK], const float* row, int h) {
  static_for<NBLK>([&](auto b_) {
    constexpr int b = decltype(b_)::value;
    float x[16];
    static_for<16>([&](auto q_) {
      constexpr int q = decltype(q_)::value;
      constexpr int k0 = 32 * b + (q & 3) + 8 * (q >> 2);  // + 4h
      const int c0 = embed3_col(k0, L), c1 = embed3_col(k0 + 4, L);
      const int c = h ? c1 : c0;
      x[q] = c >= 0 ? row[c] : 0.0f;
    });
    M::from_f32(out[b], x);
  });
}

// 6-component, 10-level embedding (DepthNet sphere intersections) into 4 blocks:
//   t < 60: level t/6, component t%6;  t = 60..62: raw (h=0: c0..c2, h=1: c3..c5);  t = 63: pad
template <class M, bool PRECISE>
__device__ __forceinline__ void embed6(typename M::Block (&out)[4], const float (&p)[6], int h) {
  const Trig<PRECISE> tr[6] = {Trig<PRECISE>(p[0]), Trig<PRECISE>(p[1]), Trig<PRECISE>(p[2]),
                               Trig<PRECISE>(p[3]), Trig<PRECISE>(p[4]), Trig<PRECISE>(p[5])};
  static_for<4>([&](auto b_) {
    constexpr int b = decltype(b_)::value;
    float x[16];
    static_for<16>([&](auto q_) {
      constexpr int q = decltype(q_)::value;
      constexpr int t = 16 * b + q;
      if constexpr (t < 60) x[q] = tr[t % 6](t / 6, h);
      else if constexpr (t < 63) x[q] = h ? p[3 + t - 60] : p[t - 60];
      else x[q] = 0.0f;
    });
    M::from_f32(out[b], x);
  });
}

// ---- host/device shared description of the embedding column maps --------------------------------
// reference column (run_nerf_helpers.py:44-45 order) held by virtual feature k of an embedding
// segment, or -1 for padding.  k = 32*blk + (q&3) + 8*(q>>2) + 4*h.
__host__ __device__ inline void k_to_bqh(int k, int& blk, int& q, int& h) {
  blk = k >> 5;
  const int r = k & 31;
  h = (r >> 2) & 1;
  q = (r & 3) + 4 * (r >> 3);
}
__host__ __device__ inline int embed3_col(int k, int L) {
  int blk, q, h;
  k_to_bqh(k, blk, q, h);
  const int t = 16 * blk + q;
  if (t < 3 * L) return 3 + 6 * (t / 3) + 3 * h + (t % 3);
  if (t == 3 * L) return h ? 2 : 0;
  if (t == 3 * L + 1) return h ? -1 : 1;
  return -1;
}
__host__ __device__ inline int embed6_col(int k) {
  int blk, q, h;
  k_to_bqh(k, blk, q, h);
  const int t = 16 * blk + q;
  if (t < 60) return 6 + 12 * (t / 6) + 6 * h + (t % 6);
  if (t < 63) return (h ? 3 : 0) + (t - 60);
  return -1;
}


// =================================================================================================
// 16x16x32 engine (v_mfma_f32_16x16x32_{bf16,f16}).  Measured on MI355X (tools/mfma_peak.hip): under
// the power cap a bare loop of this shape sustains 2.14 PFLOP/s with live operands, the 32x32x16
// shape 1.87 -- the 4-register accumulator moves half the accumulator bytes per flop.  The fused-MLP
// scheme carries over: a wave owns T tiles of 16 samples; lane (n, g) = (lane & 15, lane >> 4) holds,
// for sample n, eight features of every 32-feature K-block; a 16-feature output sub-block finishes
// as 4 fp32 registers per tile, and two consecutive sub-blocks pack into exactly the B operand of
// K-block s of the next layer:
//     element e of lane group g  <->  feature 32 s + 16 (e >> 2) + 4 g + (e & 3).
// The host packer (ns_pack.hip, layout 16) orders weight rows/columns by the same map.
typedef float f32x4a __attribute__((ext_vector_type(4)));
#ifndef NS_OB16_DEPTH
#define NS_OB16_DEPTH 4
#endif
#ifndef NS_OB16_AHEAD
#define NS_OB16_AHEAD 3
#endif
constexpr int kOb16Depth = NS_OB16_DEPTH;   // A-fragment read-ahead of the 16x16x32 kernels (one wave per SIMD)
constexpr int kOb16Ahead = NS_OB16_AHEAD;   // weight slabs in flight ahead of the open one

struct Mma16BF16 {
  static constexpr int kDtype = 1;
  static constexpr bool kPackedLeaky = false;     // (gfx950 has no packed bf16 multiply / max: LeakyReLU runs on the fp32 values)
  using AFrag = bf16x8;
  struct Block { bf16x8 v; };                       // one lane's 8 features of a 32-feature K-block
  template <bool RELU>
  __device__ static __forceinline__ uint32_t pack2(float a, float b) { return MmaBF16::pack2<RELU>(a, b); }
  __device__ static __forceinline__ void mma(f32x4a& acc, const AFrag& a, const Block& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b.v, acc, 0, 0, 0);
  }
  __device__ static __forceinline__ void mma_c(f32x4a& acc, const AFrag& a, const Block& b, const f32x4a& c) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b.v, c, 0, 0, 0);
  }
  __device__ static __forceinline__ Block from_f32(const float (&x)[8]) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 w = {pack2<false>(x[0], x[1]), pack2<false>(x[2], x[3]), pack2<false>(x[4], x[5]), pack2<false>(x[6], x[7])};
    Block b; b.v = __builtin_bit_cast(bf16x8, w); return b;
  }
};
struct Mma16F16 {
  static constexpr int kDtype = 2;
  // LeakyReLU(0.01) of a converted pair on the PACKED fp16 values, max(x, 0.01 x) as v_pk_mul_f16 + v_pk_max_f16: two VALU
  // instructions per pair instead of four on the fp32 values, and exactly what the generated DepthNet layers issue
  // (tools/gen_ob16_asm.py, act = "leaky"), so compiled and generated layers agree bit for bit.  0.01 is 0x211f in fp16
  // (0.0100021): the slope of the negative branch carries a 2e-4 relative error, below fp16's own rounding of the result.
  static constexpr bool kPackedLeaky = true;
  __device__ static __forceinline__ uint32_t leaky2(uint32_t w) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 x = __builtin_bit_cast(h2, w);
    const h2 k = {static_cast<_Float16>(0.01f), static_cast<_Float16>(0.01f)};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(x, x * k));
  }
  using AFrag = f16x8;
  struct Block { f16x8 v; };
  template <bool RELU>
  __device__ static __forceinline__ uint32_t pack2(float a, float b) { return MmaF16::pack2<RELU>(a, b); }
  __device__ static __forceinline__ void mma(f32x4a& acc, const AFrag& a, const Block& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b.v, acc, 0, 0, 0);
  }
  __device__ static __forceinline__ void mma_c(f32x4a& acc, const AFrag& a, const Block& b, const f32x4a& c) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b.v, c, 0, 0, 0);
  }
  __device__ static __forceinline__ Block from_f32(const float (&x)[8]) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 w = {pack2<false>(x[0], x[1]), pack2<false>(x[2], x[3]), pack2<false>(x[4], x[5]), pack2<false>(x[6], x[7])};
    Block b; b.v = __builtin_bit_cast(f16x8, w); return b;
  }
};

// feature index (within a layer input of 32-feature K-blocks) held by element e of lane group g of K-block s
__host__ __device__ constexpr int feature16(int s, int g, int e) { return 32 * s + 16 * (e >> 2) + 4 * g + (e & 3); }

// chunks a layer of NSB 16-row output sub-blocks x NKB K-blocks consumes, padded to the fragment pipeline depth
__host__ __device__ constexpr int ob16_chunks(int nsb, int nkb, int depth) { return ((nsb * nkb + depth - 1) / depth) * depth; }
__host__ __device__ constexpr int ob16_layer_slabs(int nsb, int nkb, int depth) {
  return (ob16_chunks(nsb, nkb, depth) + kSlabChunks - 1) / kSlabChunks;
}

// dword J (0..1) of the finished sub-block SB of one tile goes to dword 2 (SB & 1) + J of K-block SB >> 1.
// ACT: kNone / kRelu (a packed signed-16-bit max after the conversion) / kLeaky (slope 0.01, on the fp32 values:
// max(v, 0.01 v) == v > 0 ? v : 0.01 v for every finite v, and NaN stays NaN).
template <class M, int ACT, int SB, int J>
__device__ __forceinline__ void convert_piece16(typename M::Block& out, const f32x4a& c) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 w = __builtin_bit_cast(u32x4, out.v);
  float a = c[2 * J], b = c[2 * J + 1];
  if constexpr (ACT == kLeaky && M::kPackedLeaky) {
    w[2 * (SB & 1) + J] = M::leaky2(M::template pack2<false>(a, b));
    out.v = __builtin_bit_cast(typename M::AFrag, w);
    return;
  }
  if constexpr (ACT == kLeaky) {
    a = __builtin_fmaxf(a, 0.01f * a);
    b = __builtin_fmaxf(b, 0.01f * b);
  }
  w[2 * (SB & 1) + J] = M::template pack2<ACT == kRelu>(a, b);
  out.v = __builtin_bit_cast(typename M::AFrag, w);
}

// One layer: out[t][sb >> 1] <- act(bias + W . in) for the NSB 16-feature output sub-blocks but the last, whose raw
// accumulators are returned in last[t] (heads read them; hidden layers convert_last16() them).
//   in(t_, kb_) -> Block of tile t, K-block kb (compile-time indices); bias_lds: this layer's biases, natural order.
//   ACT: activation of the converted sub-blocks (enum Act; `true` / `false` of older call sites = kRelu / kNone).
// Stream order: for each sub-block, its NKB chunks (then zero chunks up to a multiple of the pipeline depth).
template <class M, int T, int NSB, int NKB, int ACT, class PipeT, class OutT, class InF>
__device__ __forceinline__ void layer_ob16(PipeT& pipe, const float* bias_lds, int g, OutT& out, f32x4a (&last)[T], InF&& in) {
  constexpr int REAL = NSB * NKB;
  constexpr int TOTAL = ob16_chunks(NSB, NKB, PipeT::kDepth);
  constexpr int PIECES = 2 * T;                       // conversion pieces of one finished sub-block
  constexpr int PPS = (PIECES + NKB - 1) / NKB;       // pieces issued per chunk step of the following sub-block
  constexpr int CONV_END = (PIECES + PPS - 1) / PPS;
  constexpr int BIAS_AT = (NKB - 2) > CONV_END ? (NKB - 2) : (NKB - 1);
  f32x4a c[2][T];
  {
    const f32x4a b0 = *reinterpret_cast<const f32x4a*>(bias_lds + 4 * g);
    static_for<T>([&](auto t_) { c[0][decltype(t_)::value] = b0; });
  }
  // A lone wave issues in order: an instruction placed after the MFMA cluster overlaps only the LAST MFMA's
  // execution (16 cycles for this shape), one placed between two MFMAs hides in the wait for the matrix pipe.  So the
  // step is emitted as MFMA / conversion piece / MFMA / fragment read / MFMA / bias / MFMA with scheduling fences.
  stream_chunks<TOTAL>(pipe, [&](auto P_, const typename M::AFrag& frag_ref, auto&& load_next) {
    constexpr int P = decltype(P_)::value;
    if constexpr (P < REAL) {
      constexpr int sb = P / NKB, kc = P % NKB, par = sb & 1;
      const typename M::AFrag frag = frag_ref;
      static_for<T>([&](auto t_) {
        constexpr int t = decltype(t_)::value;
        M::mma(c[par][t], frag, in(t_, std::integral_constant<int, kc>{}));
        if constexpr (t == 0 && sb > 0) {
          static_for<PPS>([&](auto i_) {
            constexpr int piece = kc * PPS + decltype(i_)::value;
            if constexpr (piece < PIECES) {
              convert_piece16<M, ACT, sb - 1, piece / T>(out[piece % T][(sb - 1) >> 1], c[par ^ 1][piece % T]);
            }
          });
        }
        if constexpr (t == (T > 1 ? 1 : 0)) load_next();
        if constexpr (t == (T > 2 ? 2 : T - 1) && kc == BIAS_AT && sb + 1 < NSB) {
          const f32x4a bn = *reinterpret_cast<const f32x4a*>(bias_lds + 16 * (sb + 1) + 4 * g);
          static_for<T>([&](auto u_) { c[par ^ 1][decltype(u_)::value] = bn; });
        }
      });
    } else {
      load_next();
    }
  });
  static_for<T>([&](auto t_) { last[decltype(t_)::value] = c[(NSB - 1) & 1][decltype(t_)::value]; });
}
template <class M, int ACT, int T, int NSB, class OutT>
__device__ __forceinline__ void convert_last16(OutT& out, const f32x4a (&last)[T]) {
  static_for<T>([&](auto t_) {
    static_for<2>([&](auto j_) {
      convert_piece16<M, ACT, NSB - 1, decltype(j_)::value>(out[decltype(t_)::value][(NSB - 1) >> 1], last[decltype(t_)::value]);
    });
  });
}

// 3-component, L-level embedding of one sample into NKB K-blocks for lane group g = 2u + c: the lane's element e
// of K-block kb is slot q = 16 kb + 8 u + e; q < 3L: level q / 3, component q % 3, sine (c = 0) or cosine (c = 1);
// q = 3L: raw x0 (c = 0) / x2 (c = 1); q = 3L + 1: raw x1 (c = 0) / pad; beyond: pad.
template <class M, bool PRECISE, int L, int NKB>
__device__ __forceinline__ void embed3_16(typename M::Block (&out)[NKB], float p0, float p1, float p2, int g) {
  // everything below selects VALUES, never objects: a `u ? a[i] : b[j]` on arrays or structs makes the compiler select
  // the address instead and park the operands in scratch, whose reload waits on vmcnt, i.e. on the weight DMA
  const Rev r0 = to_rev(p0), r1 = to_rev(p1), r2 = to_rev(p2);
  const bool u = (g >> 1) != 0;
  const int c = g & 1;
  auto trig = [&](float hi, float lo, int level) -> float {
    Trig<PRECISE> t(0.0f);
    t.r.hi = hi; t.r.lo = lo;
    return t(level, c);
  };
  static_for<NKB>([&](auto kb_) {
    constexpr int kb = decltype(kb_)::value;
    float x[8];
    static_for<8>([&](auto e_) {
      constexpr int e = decltype(e_)::value;
      constexpr int q0 = 16 * kb + e, q1 = q0 + 8;       // the slot for u = 0 and for u = 1
      auto value = [&](auto q_) -> float {               // non-trigonometric slots
        constexpr int q = decltype(q_)::value;
        if constexpr (q == 3 * L) return c ? p2 : p0;
        else if constexpr (q == 3 * L + 1) return c ? 0.0f : p1;
        else return 0.0f;
      };
      auto comp_hi = [&](auto q_) -> float { constexpr int k = decltype(q_)::value % 3; return k == 0 ? r0.hi : (k == 1 ? r1.hi : r2.hi); };
      auto comp_lo = [&](auto q_) -> float { constexpr int k = decltype(q_)::value % 3; return k == 0 ? r0.lo : (k == 1 ? r1.lo : r2.lo); };
      using Q0 = std::integral_constant<int, q0>;
      using Q1 = std::integral_constant<int, q1>;
      if constexpr (q1 < 3 * L) {                         // both candidates are sin/cos slots: one evaluation
        const float hi = u ? comp_hi(Q1{}) : comp_hi(Q0{});
        const float lo = u ? comp_lo(Q1{}) : comp_lo(Q0{});
        x[e] = trig(hi, lo, u ? q1 / 3 : q0 / 3);       // (the level is a lane property too: one v_ldexp)
      } else if constexpr (q0 < 3 * L) {
        const float tv = trig(comp_hi(Q0{}), comp_lo(Q0{}), q0 / 3);
        const float ov = value(Q1{});
        x[e] = u ? ov : tv;
      } else {
        const float a = value(Q0{}), b = value(Q1{});
        x[e] = u ? b : a;
      }
    });
    out[kb] = M::from_f32(x);
  });
}
// NC-component, L-level embedding of one ray into NKB K-blocks, same slot scheme as embed3_16: the lane's element e of
// K-block kb is slot q = 16 kb + 8 u + e (g = 2u + c); q < NC L: level q / NC, component q % NC, sine (c = 0) or cosine
// (c = 1); then the raw components two per slot, slot NC L + j: x_j (c = 0) / x_{HALF + j} (c = 1), HALF = ceil(NC / 2);
// beyond: pad.  (NC = 3 reproduces embed3_16; the DepthNet's sphere intersections are NC = 6, L = 10: 4 K-blocks.)
template <class M, bool PRECISE, int NC, int L, int NKB>
__device__ __forceinline__ void embedN_16(typename M::Block (&out)[NKB], const float (&p)[NC], int g) {
  constexpr int HALF = (NC + 1) / 2;
  Rev r[NC];
  static_for<NC>([&](auto i_) { r[decltype(i_)::value] = to_rev(p[decltype(i_)::value]); });
  const bool u = (g >> 1) != 0;
  const int c = g & 1;
  auto trig = [&](float hi, float lo, int level) -> float {
    Trig<PRECISE> t(0.0f);
    t.r.hi = hi; t.r.lo = lo;
    return t(level, c);
  };
  static_for<NKB>([&](auto kb_) {
    constexpr int kb = decltype(kb_)::value;
    float x[8];
    static_for<8>([&](auto e_) {
      constexpr int e = decltype(e_)::value;
      constexpr int q0 = 16 * kb + e, q1 = q0 + 8;       // the slot for u = 0 and for u = 1
      auto value = [&](auto q_) -> float {               // non-trigonometric slots (values, never addresses)
        constexpr int j = decltype(q_)::value - NC * L;
        if constexpr (j >= 0 && j < HALF) {
          const float a = p[j];
          if constexpr (HALF + j < NC) { const float b = p[HALF + j]; return c ? b : a; }
          else return c ? 0.0f : a;
        } else {
          return 0.0f;
        }
      };
      using Q0 = std::integral_constant<int, q0>;
      using Q1 = std::integral_constant<int, q1>;
      if constexpr (q1 < NC * L) {                        // both candidates are sin/cos slots: one evaluation
        const float h0 = r[q0 % NC].hi, h1 = r[q1 % NC].hi, l0 = r[q0 % NC].lo, l1 = r[q1 % NC].lo;
        const float hi = u ? h1 : h0;
        const float lo = u ? l1 : l0;
        x[e] = trig(hi, lo, u ? q1 / NC : q0 / NC);
      } else if constexpr (q0 < NC * L) {
        const float tv = trig(r[q0 % NC].hi, r[q0 % NC].lo, q0 / NC);
        const float ov = value(Q1{});
        x[e] = u ? ov : tv;
      } else {
        const float a = value(Q0{}), b = value(Q1{});
        x[e] = u ? b : a;
      }
    });
    out[kb] = M::from_f32(x);
  });
}
// reference column of feature index k of an embedN_16 segment (column order run_nerf_helpers.py:44-45), or -1
__host__ __device__ inline int embedN_col16(int k, int NC, int L) {
  const int kb = k >> 5, r = k & 31;
  const int e = (r & 3) + 4 * (r >> 4), g = (r >> 2) & 3;     // inverse of feature16()
  const int u = g >> 1, c = g & 1;
  const int q = 16 * kb + 8 * u + e;
  const int half = (NC + 1) / 2;
  if (q < NC * L) return NC + 2 * NC * (q / NC) + NC * c + (q % NC);
  const int j = q - NC * L;
  if (j < half) {
    if (c == 0) return j;
    return half + j < NC ? half + j : -1;
  }
  return -1;
}
// reference column (run_nerf_helpers.py:44-45 order) of feature index k of an embed3_16 segment, or -1 for padding
__host__ __device__ inline int embed3_col16(int k, int L) {
  const int kb = k >> 5, r = k & 31;
  const int e = (r & 3) + 4 * (r >> 4), g = (r >> 2) & 3;     // inverse of feature16()
  const int u = g >> 1, c = g & 1;
  const int q = 16 * kb + 8 * u + e;
  if (q < 3 * L) return 3 + 6 * (q / 3) + 3 * c + (q % 3);
  if (q == 3 * L) return c ? 2 : 0;
  if (q == 3 * L + 1) return c ? -1 : 1;
  return -1;
}
// pre-embedded input row (NeRF.forward on [M, 90]) into the same slots
template <class M, int L, int NKB>
__device__ __forceinline__ void gather3_16(typename M::Block (&out)[NKB], const float* row, int g) {
  static_for<NKB>([&](auto kb_) {
    constexpr int kb = decltype(kb_)::value;
    float x[8];
    static_for<8>([&](auto e_) {
      constexpr int e = decltype(e_)::value;
      const int col = embed3_col16(feature16(kb, g, e), L);
      x[e] = col >= 0 ? row[col] : 0.0f;
    });
    out[kb] = M::from_f32(x);
  });
}

}  // namespace nsmlp

// =================================================================================================
// Split-operand layers on the same 16x16x32 engine ("f16x3"): fp32-grade products at 1/3 of the fp16 MFMA rate.
// Every operand is carried as an unevaluated sum of two fp16 values, x = hi + lo with hi = fp16(x), lo = fp16(x - hi)
// (22 significant bits together), and a product term is three MFMAs: W_hi x_hi + W_hi x_lo + W_lo x_hi (the dropped
// W_lo x_lo is 2^-22 relative).  Accumulation stays fp32 in the matrix core, as in the fp32 MFMA path, which this
// path matches to fp32 rounding (tests: the fp32 gates) at ~4-5x its rate.  fp16's exponent range bounds the
// operands: |x| < 65504 (the packer refuses weights beyond it), and below ~1e-4 the lo half goes subnormal, i.e. the
// absolute error floor is ~3e-8 per operand -- the size of fp32's own rounding of O(1) values.
// Stream order (ns_pack.hip, layout 16 with split operands): per K-block the W_hi chunk then the W_lo chunk, so a
// layer has 2 NKB chunks per sub-block and the generic stream_chunks<> walk applies unchanged.
namespace nsmlp {

struct Mma16F16x3 {
  static constexpr int kDtype = 3;
  using AFrag = f16x8;
  struct Block { f16x8 hi, lo; };                    // one lane's 8 features of a 32-feature K-block, split
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  __device__ static __forceinline__ void mma(f32x4a& acc, const AFrag& a, const f16x8& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  }
  // (a, b) -> packed hi pair, packed lo pair
  __device__ static __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const f32x2 ab = {a, b};
    const f16x2 h = __builtin_convertvector(ab, f16x2);
    const f32x2 back = __builtin_convertvector(h, f32x2);
    const f32x2 rest = ab - back;                     // exact: hi is the nearest fp16, the remainder fits fp32
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(rest, f16x2));
  }
  __device__ static __forceinline__ Block from_f32(const float (&x)[8]) {
    u32x4 h, l;
    static_for<4>([&](auto i_) {
      constexpr int i = decltype(i_)::value;
      uint32_t a, b;
      split2(x[2 * i], x[2 * i + 1], a, b);
      h[i] = a; l[i] = b;
    });
    Block r;
    r.hi = __builtin_bit_cast(f16x8, h);
    r.lo = __builtin_bit_cast(f16x8, l);
    return r;
  }
};

// dword J (0..1) of the finished sub-block SB of one tile -> dword 2 (SB & 1) + J of K-block SB >> 1, both halves
template <int ACT, int SB, int J>
__device__ __forceinline__ void convert_piece16x3(Mma16F16x3::Block& out, const f32x4a& c) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  float a = c[2 * J], b = c[2 * J + 1];
  if constexpr (ACT == kRelu) { a = a < 0.0f ? 0.0f : a; b = b < 0.0f ? 0.0f : b; }   // NaN stays NaN, as torch.relu
  if constexpr (ACT == kLeaky) { a = __builtin_fmaxf(a, 0.01f * a); b = __builtin_fmaxf(b, 0.01f * b); }
  uint32_t h, l;
  Mma16F16x3::split2(a, b, h, l);
  u32x4 wh = __builtin_bit_cast(u32x4, out.hi), wl = __builtin_bit_cast(u32x4, out.lo);
  wh[2 * (SB & 1) + J] = h;
  wl[2 * (SB & 1) + J] = l;
  out.hi = __builtin_bit_cast(f16x8, wh);
  out.lo = __builtin_bit_cast(f16x8, wl);
}

// layer_ob16<> for split operands: stream chunk 2 kc of a sub-block is W_hi of K-block kc (two MFMAs per tile:
// x_hi, x_lo), chunk 2 kc + 1 is W_lo (one MFMA per tile: x_hi).  Conversion pieces (one per tile and dword) are
// spread over the chunk steps of the following sub-block as in layer_ob16<>.
template <int T, int NSB, int NKB, int ACT, class PipeT, class OutT, class InF>
__device__ __forceinline__ void layer_ob16x3(PipeT& pipe, const float* bias_lds, int g, OutT& out, f32x4a (&last)[T], InF&& in) {
  using M = Mma16F16x3;
  constexpr int CPS = 2 * NKB;                        // chunks per sub-block
  constexpr int REAL = NSB * CPS;
  constexpr int TOTAL = ob16_chunks(NSB, CPS, PipeT::kDepth);
  constexpr int PIECES = 2 * T;
  constexpr int PPS = (PIECES + CPS - 1) / CPS;
  constexpr int CONV_END = (PIECES + PPS - 1) / PPS;
  constexpr int BIAS_AT = (CPS - 2) > CONV_END ? (CPS - 2) : (CPS - 1);
  f32x4a c[2][T];
  {
    const f32x4a b0 = *reinterpret_cast<const f32x4a*>(bias_lds + 4 * g);
    static_for<T>([&](auto t_) { c[0][decltype(t_)::value] = b0; });
  }
  stream_chunks<TOTAL>(pipe, [&](auto P_, const typename M::AFrag& frag_ref, auto&& load_next) {
    constexpr int P = decltype(P_)::value;
    if constexpr (P < REAL) {
      constexpr int sb = P / CPS, cc = P % CPS, kc = cc / 2, part = cc % 2, par = sb & 1;
      const typename M::AFrag frag = frag_ref;
      static_for<T>([&](auto t_) {
        constexpr int t = decltype(t_)::value;
        const typename M::Block& xb = in(t_, std::integral_constant<int, kc>{});
        M::mma(c[par][t], frag, xb.hi);
        if constexpr (part == 0) M::mma(c[par][t], frag, xb.lo);
        if constexpr (t == 0 && sb > 0) {
          static_for<PPS>([&](auto i_) {
            constexpr int piece = cc * PPS + decltype(i_)::value;
            if constexpr (piece < PIECES)
              convert_piece16x3<ACT, sb - 1, piece / T>(out[piece % T][(sb - 1) >> 1], c[par ^ 1][piece % T]);
          });
        }
        if constexpr (t == (T > 1 ? 1 : 0)) load_next();
        if constexpr (t == T - 1 && cc == BIAS_AT && sb + 1 < NSB) {
          const f32x4a bn = *reinterpret_cast<const f32x4a*>(bias_lds + 16 * (sb + 1) + 4 * g);
          static_for<T>([&](auto u_) { c[par ^ 1][decltype(u_)::value] = bn; });
        }
      });
    } else {
      load_next();
    }
  });
  static_for<T>([&](auto t_) { last[decltype(t_)::value] = c[(NSB - 1) & 1][decltype(t_)::value]; });
}
template <int ACT, int T, int NSB, class OutT>
__device__ __forceinline__ void convert_last16x3(OutT& out, const f32x4a (&last)[T]) {
  static_for<T>([&](auto t_) {
    static_for<2>([&](auto j_) {
      convert_piece16x3<ACT, NSB - 1, decltype(j_)::value>(out[decltype(t_)::value][(NSB - 1) >> 1], last[decltype(t_)::value]);
    });
  });
}

}  // namespace nsmlp
