// Host-side packers: reference state-dict tensors -> MFMA weight stream + bias image.
//
// The stream is the exact sequence of 1-KiB A-operand chunks the kernels in ns_nerf_mlp.hip /
// ns_depthnet.hip consume (see ns_mlp_engine.h).  The "program" below (order of segment()
// calls) must mirror the order of consume<>() calls in those kernels one for one.
#include <cmath>
#include <cstring>
#include <functional>
#include <vector>

#include "ns_common.h"
#include "ns_mlp_engine.h"
#include "ns_weights.h"

namespace {

using nsmlp::kChunkBytes;
using nsmlp::kSlabBytes;
using nsmlp::kSlabChunks;

inline uint16_t f32_to_bf16_rne(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return static_cast<uint16_t>((u >> 16) | 0x40);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return static_cast<uint16_t>(u >> 16);
}
inline uint16_t f32_to_f16_rne(float f) {
  const _Float16 h = static_cast<_Float16>(f);
  uint16_t u;
  std::memcpy(&u, &h, 2);
  return u;
}

struct Builder {
  int dtype;
  int cpb;        // chunks per 32-feature input block
  int epc;        // elements per lane per chunk (8 for 16-bit, 4 for f32)
  std::vector<uint8_t> bytes;
  std::vector<float> bias;

  explicit Builder(int dt) : dtype(dt), cpb(dt == NS_DTYPE_F32 ? 4 : 2), epc(dt == NS_DTYPE_F32 ? 4 : 8) {}

  void put(uint8_t* lane_base, int elem, float v) const {
    if (dtype == NS_DTYPE_F32) {
      std::memcpy(lane_base + 4 * elem, &v, 4);
    } else {
      const uint16_t h = dtype == NS_DTYPE_BF16 ? f32_to_bf16_rne(v) : f32_to_f16_rne(v);
      std::memcpy(lane_base + 2 * elem, &h, 2);
    }
  }

  // one 1-KiB A-operand chunk: output block nb, input block blk, K sub-step sub
  void fill_chunk(uint8_t* chunk, const float* Wm, int out_f, int in_f, int nb, int blk, int sub,
                  const std::function<int(int)>& colmap) const {
    for (int lane = 0; lane < 64; ++lane) {
      const int r = lane & 31, h = lane >> 5;
      const int n = 32 * nb + r;
      for (int e = 0; e < epc; ++e) {
        const int q = epc * sub + e;
        const int k = 32 * blk + (q & 3) + 8 * (q >> 2) + 4 * h;
        const int col = colmap(k);
        float v = 0.0f;
        if (n < out_f && col >= 0) {
          if (col >= in_f) { std::fprintf(stderr, "ns_pack: column %d out of range %d\n", col, in_f); std::abort(); }
          v = Wm[static_cast<size_t>(n) * in_f + col];
        }
        put(chunk + lane * 16, e, v);
      }
    }
  }

  // Layout 0 (k-major, consume<>()): one K-segment, rows of Wm ([out_f, in_f] row-major) as NBO output
  // blocks, NBLK input blocks whose virtual feature k maps to reference column colmap(k) (or -1 = zero).
  void segment(const float* Wm, int out_f, int in_f, int nbo, int nblk, const std::function<int(int)>& colmap) {
    const int kps = kSlabChunks / nbo;
    const int chunks = nblk * cpb;
    const int slabs = (chunks + kps - 1) / kps;
    for (int s = 0; s < slabs; ++s) {
      const size_t base = bytes.size();
      bytes.resize(base + kSlabBytes, 0);
      for (int kk = 0; kk < kps; ++kk) {
        const int kc = s * kps + kk;
        if (kc >= chunks) continue;
        for (int nb = 0; nb < nbo; ++nb)
          fill_chunk(bytes.data() + base + static_cast<size_t>(kk * nbo + nb) * kChunkBytes, Wm, out_f, in_f, nb,
                     kc / cpb, kc % cpb, colmap);
      }
    }
  }

  struct Seg { int nblk; std::function<int(int)> colmap; };   // a K-segment: nblk 32-feature blocks, column map

  // Layout 16 (16x16x32 engine, layer_ob16<>()): chunk = 16 output rows x 32 input features; lane (n, g) holds
  // row 16 sb + n and the features feature16(kb, g, e), e = 0..7, of K-block kb.
  void fill_chunk16(uint8_t* chunk, const float* Wm, int out_f, int in_f, int sb, int kb,
                    const std::function<int(int)>& colmap) const {
    for (int lane = 0; lane < 64; ++lane) {
      const int n = 16 * sb + (lane & 15), g = lane >> 4;
      for (int e = 0; e < 8; ++e) {
        const int col = colmap(nsmlp::feature16(kb, g, e));
        float v = 0.0f;
        if (n < out_f && col >= 0) {
          if (col >= in_f) { std::fprintf(stderr, "ns_pack: column %d out of range %d\n", col, in_f); std::abort(); }
          v = Wm[static_cast<size_t>(n) * in_f + col];
        }
        put(chunk + lane * 16, e, v);
      }
    }
  }
  void layer_ob16(const float* Wm, int out_f, int in_f, int nsb, const std::vector<Seg>& segs) {
    const size_t base = bytes.size();
    size_t n = 0;
    for (int sb = 0; sb < nsb; ++sb)
      for (const Seg& sg : segs)
        for (int kb = 0; kb < sg.nblk; ++kb) {
          bytes.resize(base + (n + 1) * kChunkBytes, 0);
          fill_chunk16(bytes.data() + base + n * kChunkBytes, Wm, out_f, in_f, sb, kb, sg.colmap);
          ++n;
        }
    // zero chunks up to the fragment pipeline depth (the kernel walks them without MFMAs), then to a slab boundary
    const size_t padded = (n + nsmlp::kOb16Depth - 1) / nsmlp::kOb16Depth * nsmlp::kOb16Depth;
    const size_t slabs = (padded + kSlabChunks - 1) / kSlabChunks;
    bytes.resize(base + slabs * kSlabBytes, 0);
  }
  void add_bias16(const float* b, int out_f, int nsb) {   // natural order, 16 per sub-block
    for (int i = 0; i < 16 * nsb; ++i) bias.push_back(i < out_f ? b[i] : 0.0f);
  }

  void add_bias(const float* b, int out_f, int nbo) {
    for (int nb = 0; nb < nbo; ++nb)
      for (int h = 0; h < 2; ++h)
        for (int r = 0; r < 16; ++r) {
          const int n = 32 * nb + (r & 3) + 8 * (r >> 2) + 4 * h;
          bias.push_back(n < out_f ? b[n] : 0.0f);
        }
  }
};

int finish(Builder& b, ns_weights* w) {
  w->n_slabs = static_cast<uint32_t>(b.bytes.size() / kSlabBytes);
  w->bias_floats = static_cast<int>(b.bias.size());
  NS_HIP(hipMalloc(&w->stream_dev, b.bytes.size()));
  NS_HIP(hipMemcpy(w->stream_dev, b.bytes.data(), b.bytes.size(), hipMemcpyHostToDevice));
  NS_HIP(hipMalloc(reinterpret_cast<void**>(&w->bias_dev), b.bias.size() * sizeof(float)));
  NS_HIP(hipMemcpy(w->bias_dev, b.bias.data(), b.bias.size() * sizeof(float), hipMemcpyHostToDevice));
  return NS_OK;
}

}  // namespace

namespace ns {
// build the host images only (used by the CPU-side layout test through ns_pack_*_host)
}

extern "C" {

int ns_pack_nerf(int D, int W, int skip, const float* const* w, const float* const* b, int dtype,
                 ns_weights** out) {
  NS_REQUIRE(out && w && b, "null pointer");
  *out = nullptr;
  if (!(W == 128 || W == 256) || D < 1 || D > 64 || skip < -1 || (skip >= 0 && skip >= D - 1) ||
      !(dtype == NS_DTYPE_F32 || dtype == NS_DTYPE_BF16 || dtype == NS_DTYPE_F16)) {
    ns::set_error("ns_pack_nerf: unsupported network (W=%d D=%d skip=%d dtype=%d); kernels exist for "
                  "W in {128,256}, one optional skip before the last layer, input_ch 63/27", W, D, skip, dtype);
    return NS_E_UNSUPPORTED;
  }
  for (int i = 0; i < D + 4; ++i) NS_REQUIRE(w[i] && b[i], "null weight tensor");
  const int NB = W / 32;
  Builder bl(dtype);
  auto ident = [](int k) { return k; };
  auto xcol = [](int k) { return nsmlp::embed3_col(k, 10); };
  const int layout = dtype == NS_DTYPE_F32 ? 0 : 16;   // 0 = k-major (fp32 kernel); 16 = 16x16x32 engine (bf16 / f16)
  if (layout == 0) {
    // layer 0: 63 -> W
    bl.add_bias(b[0], W, NB);
    bl.segment(w[0], W, 63, NB, 2, xcol);
    for (int l = 1; l < D; ++l) {
      bl.add_bias(b[l], W, NB);
      if (l - 1 == skip) {  // input = cat[x(63), h(W)]  (run_nerf_helpers.py:118)
        bl.segment(w[l], W, W + 63, NB, 2, xcol);
        bl.segment(w[l], W, W + 63, NB, NB, [](int k) { return 63 + k; });
      } else {
        bl.segment(w[l], W, W, NB, NB, ident);
      }
    }
    const float* const* wf = w + D;
    const float* const* bf = b + D;
    // alpha (W -> 1), feature (W -> W), views ([feature, dirs27] -> W/2), rgb (W/2 -> 3)
    bl.add_bias(bf[1], 1, 1);
    bl.segment(wf[1], 1, W, 1, NB, ident);
    bl.add_bias(bf[0], W, NB);
    bl.segment(wf[0], W, W, NB, NB, ident);
    bl.add_bias(bf[2], W / 2, NB / 2);
    bl.segment(wf[2], W / 2, W + 27, NB / 2, NB, ident);
    bl.segment(wf[2], W / 2, W + 27, NB / 2, 1, [W](int k) { const int c = nsmlp::embed3_col(k, 4); return c < 0 ? -1 : W + c; });
    bl.add_bias(bf[3], 3, 1);
    bl.segment(wf[3], 3, W / 2, 1, NB / 2, ident);
  } else if (layout == 16) {
    // same layers for the 16x16x32 kernel (ns_nerf_mlp_ob16.hip): NSB = out/16 sub-blocks, K-blocks of 32 features;
    // hidden features arrive in feature16() order, i.e. plain feature indices for the column maps
    auto xcol16 = [](int k) { return nsmlp::embed3_col16(k, 10); };
    auto hcol = [](int k) { return 63 + k; };
    const int NSB = W / 16, NKB = W / 32;
    bl.add_bias16(b[0], W, NSB);
    bl.layer_ob16(w[0], W, 63, NSB, {{2, xcol16}});
    for (int l = 1; l < D; ++l) {
      bl.add_bias16(b[l], W, NSB);
      if (l - 1 == skip) bl.layer_ob16(w[l], W, W + 63, NSB, {{2, xcol16}, {NKB, hcol}});
      else bl.layer_ob16(w[l], W, W, NSB, {{NKB, ident}});
    }
    const float* const* wf = w + D;
    const float* const* bf = b + D;
    bl.add_bias16(bf[1], 1, 1);
    bl.layer_ob16(wf[1], 1, W, 1, {{NKB, ident}});
    bl.add_bias16(bf[0], W, NSB);
    bl.layer_ob16(wf[0], W, W, NSB, {{NKB, ident}});
    bl.add_bias16(bf[2], W / 2, NSB / 2);
    bl.layer_ob16(wf[2], W / 2, W + 27, NSB / 2,
                  {{NKB, ident}, {1, [W](int k) { const int c = nsmlp::embed3_col16(k, 4); return c < 0 ? -1 : W + c; }}});
    bl.add_bias16(bf[3], 3, 1);
    bl.layer_ob16(wf[3], 3, W / 2, 1, {{NKB / 2, ident}});
  }

  ns_weights* h = new ns_weights();
  std::memset(h, 0, sizeof(*h));
  h->kind = NS_KIND_NERF; h->dtype = dtype; h->width = W; h->depth = D; h->skip = skip; h->layout = layout;
  int rc = finish(bl, h);
  if (rc != NS_OK) { ns_weights_destroy(h); return rc; }
  *out = h;
  return NS_OK;
}

int ns_pack_depthnet(int n_layers, int width, const float* const* w, const float* const* b, int dtype,
                     ns_weights** out) {
  NS_REQUIRE(out && w && b, "null pointer");
  *out = nullptr;
  if (!(width == 128 || width == 256) || n_layers < 1 || n_layers > 64 ||
      !(dtype == NS_DTYPE_F32 || dtype == NS_DTYPE_BF16 || dtype == NS_DTYPE_F16)) {
    ns::set_error("ns_pack_depthnet: unsupported network (width=%d n_layers=%d dtype=%d); kernels exist for "
                  "uniform hidden width in {128,256}, multires 10", width, n_layers, dtype);
    return NS_E_UNSUPPORTED;
  }
  for (int i = 0; i < 4 * n_layers + 1; ++i) NS_REQUIRE(w[i] && b[i], "null weight tensor");
  const int W = width, NB = W / 32, n = n_layers;
  Builder bl(dtype);
  auto ident = [](int k) { return k; };
  auto col3 = [](int k) { return nsmlp::embed3_col(k, 10); };
  auto col6 = [](int k) { return nsmlp::embed6_col(k); };
  // skip branches (depth_net.py:136-156): layer 0 sees cat[e, e], layers >= 1 cat[h, e]
  auto branch = [&](int first, int eblk, int ecols, const std::function<int(int)>& ecol) {
    bl.add_bias(b[first], W, NB);
    bl.segment(w[first], W, 2 * ecols, NB, eblk, ecol);
    bl.segment(w[first], W, 2 * ecols, NB, eblk, [&](int k) { const int c = ecol(k); return c < 0 ? -1 : ecols + c; });
    for (int i = 1; i < n; ++i) {
      bl.add_bias(b[first + i], W, NB);
      bl.segment(w[first + i], W, W + ecols, NB, NB, ident);
      bl.segment(w[first + i], W, W + ecols, NB, eblk, [&](int k) { const int c = ecol(k); return c < 0 ? -1 : W + c; });
    }
  };
  branch(0, 2, 63, col3);          // origin
  branch(n, 2, 63, col3);          // direction
  branch(2 * n, 4, 126, col6);     // sphere intersections
  // trunk layer 0 on cat[h_o, h_d, h_x, e_o, e_d, e_x] (depth_net.py:158-163); the kernel consumes
  // the K-segments in the order h_x, e_x, h_o, e_o, h_d, e_d
  const int t0 = 3 * n, inT = 3 * W + 252;
  bl.add_bias(b[t0], W, NB);
  bl.segment(w[t0], W, inT, NB, NB, [W](int k) { return 2 * W + k; });
  bl.segment(w[t0], W, inT, NB, 4, [&](int k) { const int c = col6(k); return c < 0 ? -1 : 3 * W + 126 + c; });
  bl.segment(w[t0], W, inT, NB, NB, ident);
  bl.segment(w[t0], W, inT, NB, 2, [&](int k) { const int c = col3(k); return c < 0 ? -1 : 3 * W + c; });
  bl.segment(w[t0], W, inT, NB, NB, [W](int k) { return W + k; });
  bl.segment(w[t0], W, inT, NB, 2, [&](int k) { const int c = col3(k); return c < 0 ? -1 : 3 * W + 63 + c; });
  for (int i = 1; i < n; ++i) {
    bl.add_bias(b[t0 + i], W, NB);
    bl.segment(w[t0 + i], W, W, NB, NB, ident);
  }
  bl.add_bias(b[4 * n], 1, 1);
  bl.segment(w[4 * n], 1, W, 1, NB, ident);

  ns_weights* h = new ns_weights();
  std::memset(h, 0, sizeof(*h));
  h->kind = NS_KIND_DEPTHNET; h->dtype = dtype; h->width = W; h->depth = n; h->skip = -1;
  int rc = finish(bl, h);
  if (rc == NS_OK) {
    // stash for two branch outputs per wave: kDepthnetMaxGrid workgroups x waves x 2 x NB blocks,
    // one block = 64 lanes x (64 B fp32 | 32 B 16-bit); waves per workgroup: 4 (fp32) or 8
    const size_t blk_bytes = (dtype == NS_DTYPE_F32 ? 64 : 32) * 64;
    const size_t waves = dtype == NS_DTYPE_F32 ? 4 : 8;
    h->scratch_bytes = static_cast<size_t>(kDepthnetMaxGrid) * waves * 2 * NB * blk_bytes;
    hipError_t e = hipMalloc(&h->scratch_dev, h->scratch_bytes);
    if (e != hipSuccess) { ns::set_error("ns_pack_depthnet: hipMalloc scratch -> %s", hipGetErrorString(e)); rc = NS_E_HIP; }
  }
  if (rc != NS_OK) { ns_weights_destroy(h); return rc; }
  *out = h;
  return NS_OK;
}

void ns_weights_destroy(ns_weights* w) {
  if (!w) return;
  if (w->stream_dev) (void)hipFree(w->stream_dev);
  if (w->bias_dev) (void)hipFree(w->bias_dev);
  if (w->scratch_dev) (void)hipFree(w->scratch_dev);
  delete w;
}

int64_t ns_weights_stream_bytes(const ns_weights* w) {
  return w ? static_cast<int64_t>(w->n_slabs) * kSlabBytes : 0;
}

}  // extern "C"
