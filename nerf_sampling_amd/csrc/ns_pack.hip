// Host-side packers: reference state-dict tensors -> MFMA weight stream + bias image.
//
// The stream is the exact sequence of 1-KiB A-operand chunks the kernels in ns_nerf_mlp*.hip /
// ns_depthnet*.hip consume (see ns_mlp_engine.h).  The "program" below (order of segment() /
// layer_ob16() calls) must mirror the order of consume<>() / layer_ob16<>() calls in those kernels one for one.
//
// Affine stretches of the reference networks are composed here, once, in fp64 ("folding"):
//   * DepthNet: the three skip branches never apply their activation (depth_net.py:136-160 constructs
//     nn.LeakyReLU(x) and drops it), so each is one affine map of its embedding, and together with the first trunk
//     layer (depth_net.py:158-163) the whole front end is ONE 252 -> C0 layer on cat[e_o, e_d, e_x].
//   * NeRF: feature_linear has no activation and feeds views_linears[0] directly (run_nerf_helpers.py:119-125), so
//     W_views[:, :W] . W_feature is one (W + 27) -> W/2 layer.
// The kernels then execute 19.6 % (DepthNet) / 89 % (NeRF) of the reference's MACs; the result differs from the
// reference's own fp32 chain by rounding only (the fp64 composition is the more exact of the two).
#include <cmath>
#include <cstring>
#include <functional>
#include <utility>
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

  bool split = false;   // NS_DTYPE_F16X3: every layout-16 chunk is followed by the chunk of its fp16 remainders

  explicit Builder(int dt) : dtype(dt == NS_DTYPE_F16X3 ? NS_DTYPE_F16 : dt), cpb(dt == NS_DTYPE_F32 ? 4 : 2),
                             epc(dt == NS_DTYPE_F32 ? 4 : 8), split(dt == NS_DTYPE_F16X3) {}

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
  // remainder chunk of a split-operand stream: lo = fp16(w - float(fp16(w))), same lane layout as the hi chunk
  void fill_chunk16_lo(uint8_t* chunk, const float* Wm, int out_f, int in_f, int sb, int kb,
                       const std::function<int(int)>& colmap) const {
    for (int lane = 0; lane < 64; ++lane) {
      const int n = 16 * sb + (lane & 15), g = lane >> 4;
      for (int e = 0; e < 8; ++e) {
        const int col = colmap(nsmlp::feature16(kb, g, e));
        float v = 0.0f;
        if (n < out_f && col >= 0) {
          const float w = Wm[static_cast<size_t>(n) * in_f + col];
          v = w - static_cast<float>(static_cast<_Float16>(w));
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
          bytes.resize(base + (n + 1 + (split ? 1 : 0)) * kChunkBytes, 0);
          fill_chunk16(bytes.data() + base + n * kChunkBytes, Wm, out_f, in_f, sb, kb, sg.colmap);
          ++n;
          if (split) {
            fill_chunk16_lo(bytes.data() + base + n * kChunkBytes, Wm, out_f, in_f, sb, kb, sg.colmap);
            ++n;
          }
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


// ---- fp64 composition of affine stretches -----------------------------------------------------------
struct Affine {   // y = A x + c, A row-major [out, in]
  int out = 0, in = 0;
  std::vector<double> A, c;
};

// One DepthNet skip branch (depth_net.py:136-156): h = e; layer 0 on cat[e, e]; layers i >= 1 on cat[h, e]; no
// activation.  w[i] is [hs[i], (i ? hs[i-1] : E) + E] row-major.
Affine fold_branch(const float* const* w, const float* const* b, int n, const int* hs, int E) {
  Affine f;
  f.out = hs[0]; f.in = E;
  f.A.assign(static_cast<size_t>(hs[0]) * E, 0.0);
  f.c.assign(hs[0], 0.0);
  for (int r = 0; r < hs[0]; ++r) {
    const float* row = w[0] + static_cast<size_t>(r) * 2 * E;
    for (int k = 0; k < E; ++k) f.A[static_cast<size_t>(r) * E + k] = static_cast<double>(row[k]) + static_cast<double>(row[E + k]);
    f.c[r] = b[0][r];
  }
  for (int i = 1; i < n; ++i) {
    const int H = hs[i - 1], O = hs[i], in_f = H + E;
    Affine g;
    g.out = O; g.in = E;
    g.A.assign(static_cast<size_t>(O) * E, 0.0);
    g.c.assign(O, 0.0);
    for (int r = 0; r < O; ++r) {
      const float* row = w[i] + static_cast<size_t>(r) * in_f;
      double* dst = g.A.data() + static_cast<size_t>(r) * E;
      double cc = b[i][r];
      for (int j = 0; j < H; ++j) {
        const double wj = row[j];
        const double* src = f.A.data() + static_cast<size_t>(j) * E;
        for (int k = 0; k < E; ++k) dst[k] += wj * src[k];
        cc += wj * f.c[j];
      }
      for (int k = 0; k < E; ++k) dst[k] += static_cast<double>(row[H + k]);
      g.c[r] = cc;
    }
    f = std::move(g);
  }
  return f;
}

// dst[r, dst_col0 + k] (+)= sum_j T[r, t_col0 + j] * f.A[j, k];  bias[r] += sum_j T[r, t_col0 + j] * f.c[j]
void compose_into(std::vector<double>& dst, int dst_ld, int dst_col0, std::vector<double>& bias, const float* T, int t_ld,
                  int t_col0, int rows, const Affine& f) {
  for (int r = 0; r < rows; ++r) {
    const float* trow = T + static_cast<size_t>(r) * t_ld + t_col0;
    double* d = dst.data() + static_cast<size_t>(r) * dst_ld + dst_col0;
    double cc = 0.0;
    for (int j = 0; j < f.out; ++j) {
      const double tj = trow[j];
      const double* src = f.A.data() + static_cast<size_t>(j) * f.in;
      for (int k = 0; k < f.in; ++k) d[k] += tj * src[k];
      cc += tj * f.c[j];
    }
    bias[r] += cc;
  }
}

std::vector<float> to_f32(const std::vector<double>& v) {
  std::vector<float> o(v.size());
  for (size_t i = 0; i < v.size(); ++i) o[i] = static_cast<float>(v[i]);
  return o;
}


// views(cat[feature(h), dirs]) = (Wv[:, :W] Wf) h + Wv[:, W:] dirs + (Wv[:, :W] bf + bv): one (W + 27) -> W/2 layer
// (run_nerf_helpers.py:119-125: no activation between feature_linear and views_linears[0]), composed in fp64
void fold_nerf_views(int W, const float* Wf, const float* bfeat, const float* Wv, const float* bv,
                     std::vector<float>& wvf, std::vector<float>& bvf) {
  const int HV = W / 2, KV = W + 27;
  std::vector<double> acc(static_cast<size_t>(HV) * KV, 0.0), bb(HV, 0.0);
  for (int r = 0; r < HV; ++r) {
    const float* vrow = Wv + static_cast<size_t>(r) * KV;
    double* dst = acc.data() + static_cast<size_t>(r) * KV;
    double cc = bv[r];
    for (int j = 0; j < W; ++j) {
      const double vj = vrow[j];
      const float* frow = Wf + static_cast<size_t>(j) * W;
      for (int k = 0; k < W; ++k) dst[k] += vj * static_cast<double>(frow[k]);
      cc += vj * static_cast<double>(bfeat[j]);
    }
    for (int k = W; k < KV; ++k) dst[k] = vrow[k];
    bb[r] = cc;
  }
  wvf = to_f32(acc);
  bvf = to_f32(bb);
}

// DepthNet front end: the three affine skip branches (depth_net.py:136-156) composed with the first trunk layer on
// cat[h_o, h_d, h_x, e_o, e_d, e_x] (depth_net.py:158-163) -> F [C0, 252] on cat[e_o, e_d, e_x], fb [C0]
void fold_depthnet_front(int n_branch, const int* hidden_sizes, int C0, const float* const* w, const float* const* b,
                         std::vector<float>& F32, std::vector<float>& fb32) {
  const int E3 = 63, E6 = 126, EIN = E3 + E3 + E6;
  const Affine fo = fold_branch(w, b, n_branch, hidden_sizes, E3);
  const Affine fd = fold_branch(w + n_branch, b + n_branch, n_branch, hidden_sizes, E3);
  const Affine fx = fold_branch(w + 2 * n_branch, b + 2 * n_branch, n_branch, hidden_sizes, E6);
  const int HL = hidden_sizes[n_branch - 1], t0 = 3 * n_branch, inT = 3 * HL + EIN;
  std::vector<double> F(static_cast<size_t>(C0) * EIN, 0.0), fb(C0, 0.0);
  for (int r = 0; r < C0; ++r) {
    const float* row = w[t0] + static_cast<size_t>(r) * inT + 3 * HL;
    for (int k = 0; k < EIN; ++k) F[static_cast<size_t>(r) * EIN + k] = row[k];
    fb[r] = b[t0][r];
  }
  compose_into(F, EIN, 0, fb, w[t0], inT, 0, C0, fo);
  compose_into(F, EIN, E3, fb, w[t0], inT, HL, C0, fd);
  compose_into(F, EIN, 2 * E3, fb, w[t0], inT, 2 * HL, C0, fx);
  F32 = to_f32(F);
  fb32 = to_f32(fb);
}

// fp16-operand streams (NS_DTYPE_F16, NS_DTYPE_F16X3): a weight beyond fp16's range would become +-inf silently
bool fits_f16(const float* w, size_t n) {
  for (size_t i = 0; i < n; ++i)
    if (!(std::fabs(w[i]) < 65504.0f) && !std::isnan(w[i])) return false;
  return true;
}

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
  if (skip < -1 || skip >= 32 || (skip >= 0 && skip >= D - 1)) {
    ns::set_error("ns_pack_nerf: unsupported network (D=%d skip=%d): the skip must precede the last layer", D, skip);
    if (out) *out = nullptr;
    return NS_E_UNSUPPORTED;
  }
  return ns_pack_nerf_ex(D, W, skip >= 0 ? (1u << skip) : 0u, 1, 4, w, b, dtype, out);
}

int ns_pack_nerf_ex(int D, int W, uint32_t skip_mask, int use_viewdirs, int output_ch, const float* const* w,
                    const float* const* b, int dtype, ns_weights** out) {
  NS_REQUIRE(out && w && b, "null pointer");
  *out = nullptr;
  const int out_ch = use_viewdirs ? 4 : output_ch;
  if (W >= 1 && W <= 256 && W != 128 && W != 256 && D >= 1 && D <= 32) {
    // Any width up to 256 runs on the W = 128 / 256 kernels: every tensor is zero-padded to the next kernel width Wp.  A padded
    // hidden unit has zero weights and a zero bias -- relu(0) = 0 -- and feeds zero columns of the next layer, so every real
    // unit sums the same products plus exact zeros: the same network (run_nerf_helpers.py:87-105 with W -> Wp).
    const int Wp = W <= 128 ? 128 : 256, HV = W / 2, HVp = Wp / 2;
    auto skipped = [skip_mask](int l) { return l >= 1 && ((skip_mask >> (l - 1)) & 1u) != 0; };
    const int n_tensors = use_viewdirs ? D + 4 : D + 1;
    for (int i = 0; i < n_tensors; ++i) NS_REQUIRE(w[i] && b[i], "null weight tensor");
    std::vector<std::vector<float>> pw(n_tensors), pb(n_tensors);
    // rows x cols -> prows x pcols; source column k lands in column k (k < split) or k - split + psplit (the columns behind a
    // block of hidden features that grew from `split` to `psplit`)
    auto pad = [&](int i, int rows, int cols, int prows, int pcols, int split, int psplit) {
      pw[i].assign(static_cast<size_t>(prows) * pcols, 0.0f);
      pb[i].assign(prows, 0.0f);
      for (int r = 0; r < rows; ++r) {
        for (int k = 0; k < cols; ++k)
          pw[i][static_cast<size_t>(r) * pcols + (k < split ? k : k - split + psplit)] = w[i][static_cast<size_t>(r) * cols + k];
        pb[i][r] = b[i][r];
      }
    };
    pad(0, W, 63, Wp, 63, 63, 63);
    for (int l = 1; l < D; ++l) {
      if (skipped(l)) pad(l, W, 63 + W, Wp, 63 + Wp, 63 + W, 63 + Wp);     // cat[x(63), h(W)]: the hidden block is last
      else pad(l, W, W, Wp, Wp, W, Wp);
    }
    if (use_viewdirs) {
      pad(D, W, W, Wp, Wp, W, Wp);                       // feature_linear
      pad(D + 1, 1, W, 1, Wp, W, Wp);                    // alpha_linear
      pad(D + 2, HV, W + 27, HVp, Wp + 27, W, Wp);       // views_linears.0 on cat[feature(W), dirs(27)]
      pad(D + 3, 3, HV, 3, HVp, HV, HVp);                // rgb_linear
    } else {
      pad(D, out_ch, W, out_ch, Wp, W, Wp);              // output_linear
    }
    std::vector<const float*> wp(n_tensors), bp(n_tensors);
    for (int i = 0; i < n_tensors; ++i) { wp[i] = pw[i].data(); bp[i] = pb[i].data(); }
    return ns_pack_nerf_ex(D, Wp, skip_mask, use_viewdirs, output_ch, wp.data(), bp.data(), dtype, out);
  }
  if (!(W == 128 || W == 256) || D < 1 || D > 32 || (skip_mask >> (D - 1)) != 0 || out_ch < 1 || out_ch > 16 ||
      !(dtype == NS_DTYPE_F32 || dtype == NS_DTYPE_BF16 || dtype == NS_DTYPE_F16 || dtype == NS_DTYPE_F16X3)) {
    ns::set_error("ns_pack_nerf: unsupported network (W=%d D=%d skips=0x%x output_ch=%d dtype=%d); kernels exist for "
                  "W <= 256, D <= 32, skips before the last layer, input_ch 63 (/27), output_ch <= 16", W, D, skip_mask,
                  out_ch, dtype);
    return NS_E_UNSUPPORTED;
  }
  auto skipped = [skip_mask](int l) { return l >= 1 && ((skip_mask >> (l - 1)) & 1u) != 0; };   // layer l sees cat[x, h]
  const int n_tensors = use_viewdirs ? D + 4 : D + 1;
  for (int i = 0; i < n_tensors; ++i) NS_REQUIRE(w[i] && b[i], "null weight tensor");
  if (dtype == NS_DTYPE_F16 || dtype == NS_DTYPE_F16X3) {   // fp16 operands: refuse what would silently become +-inf
    bool ok = true;
    for (int l = 0; l < D && ok; ++l) ok = fits_f16(w[l], static_cast<size_t>(W) * (l == 0 ? 63 : (skipped(l) ? W + 63 : W)));
    if (use_viewdirs)
      ok = ok && fits_f16(w[D], static_cast<size_t>(W) * W) && fits_f16(w[D + 1], W) &&
           fits_f16(w[D + 2], static_cast<size_t>(W / 2) * (W + 27)) && fits_f16(w[D + 3], static_cast<size_t>(3) * (W / 2));
    else
      ok = ok && fits_f16(w[D], static_cast<size_t>(out_ch) * W);
    if (!ok) {
      ns::set_error("ns_pack_nerf: a weight exceeds fp16's range (65504); use bf16 or f32 operands for this network");
      return NS_E_UNSUPPORTED;
    }
  }
  const int NB = W / 32;
  Builder bl(dtype);
  auto ident = [](int k) { return k; };
  auto xcol = [](int k) { return nsmlp::embed3_col(k, 10); };
  const int layout = dtype == NS_DTYPE_F32 ? 0 : 16;   // 0 = k-major (fp32 kernel); 16 = 16x16x32 engine (bf16 / f16)
  const int HV = W / 2, KV = W + 27;
  std::vector<float> wvf, bvf;
  if (use_viewdirs) {
    fold_nerf_views(W, w[D], b[D], w[D + 2], b[D + 2], wvf, bvf);
    if ((dtype == NS_DTYPE_F16 || dtype == NS_DTYPE_F16X3) && !fits_f16(wvf.data(), wvf.size())) {
      ns::set_error("ns_pack_nerf: a folded views-layer weight exceeds fp16's range (65504); use bf16 or f32 operands");
      return NS_E_UNSUPPORTED;
    }
  }
  if (layout == 0) {
    // layer 0: 63 -> W
    bl.add_bias(b[0], W, NB);
    bl.segment(w[0], W, 63, NB, 2, xcol);
    for (int l = 1; l < D; ++l) {
      bl.add_bias(b[l], W, NB);
      if (skipped(l)) {  // input = cat[x(63), h(W)]  (run_nerf_helpers.py:118)
        bl.segment(w[l], W, W + 63, NB, 2, xcol);
        bl.segment(w[l], W, W + 63, NB, NB, [](int k) { return 63 + k; });
      } else {
        bl.segment(w[l], W, W, NB, NB, ident);
      }
    }
    const float* const* wf = w + D;
    const float* const* bf = b + D;
    if (use_viewdirs) {
      // alpha (W -> 1), views o feature folded ([h, dirs27] -> W/2), rgb (W/2 -> 3)
      bl.add_bias(bf[1], 1, 1);
      bl.segment(wf[1], 1, W, 1, NB, ident);
      bl.add_bias(bvf.data(), HV, NB / 2);
      bl.segment(wvf.data(), HV, KV, NB / 2, NB, ident);
      bl.segment(wvf.data(), HV, KV, NB / 2, 1, [W](int k) { const int c = nsmlp::embed3_col(k, 4); return c < 0 ? -1 : W + c; });
      bl.add_bias(bf[3], 3, 1);
      bl.segment(wf[3], 3, W / 2, 1, NB / 2, ident);
    } else {
      // output_linear (W -> output_ch), no activation (run_nerf_helpers.py:132-133): rows 0 .. out_ch-1 of one block
      bl.add_bias(bf[0], out_ch, 1);
      bl.segment(wf[0], out_ch, W, 1, NB, ident);
    }
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
      if (skipped(l)) bl.layer_ob16(w[l], W, W + 63, NSB, {{2, xcol16}, {NKB, hcol}});
      else bl.layer_ob16(w[l], W, W, NSB, {{NKB, ident}});
    }
    const float* const* wf = w + D;
    if (use_viewdirs) {
      // views o feature (folded) with alpha_linear riding along as row W/2, i.e. row 0 of one extra 16-row sub-block
      // whose raw accumulators the kernel reads as sigma (no activation: it is the layer's LAST sub-block)
      std::vector<float> wc(static_cast<size_t>(HV + 1) * KV, 0.0f), bc(HV + 1, 0.0f);
      std::memcpy(wc.data(), wvf.data(), wvf.size() * sizeof(float));
      std::memcpy(bc.data(), bvf.data(), bvf.size() * sizeof(float));
      std::memcpy(wc.data() + static_cast<size_t>(HV) * KV, wf[1], static_cast<size_t>(W) * sizeof(float));
      bc[HV] = b[D + 1][0];
      bl.add_bias16(bc.data(), HV + 1, NSB / 2 + 1);
      bl.layer_ob16(wc.data(), HV + 1, KV, NSB / 2 + 1,
                    {{NKB, ident}, {1, [W](int k) { const int c = nsmlp::embed3_col16(k, 4); return c < 0 ? -1 : W + c; }}});
      bl.add_bias16(b[D + 3], 3, 1);
      bl.layer_ob16(wf[3], 3, W / 2, 1, {{NKB / 2, ident}});
    } else {
      bl.add_bias16(b[D], out_ch, 1);
      bl.layer_ob16(wf[0], out_ch, W, 1, {{NKB, ident}});
    }
  }

  ns_weights* h = new ns_weights();
  std::memset(h, 0, sizeof(*h));
  h->kind = NS_KIND_NERF; h->dtype = dtype; h->width = W; h->depth = D; h->layout = layout;
  h->skip_mask = skip_mask; h->use_viewdirs = use_viewdirs ? 1 : 0; h->out_ch = out_ch;
  h->skip = -1;
  for (int i = 0; i < 32; ++i) if ((skip_mask >> i) & 1u) { h->skip = i; break; }
  int rc = finish(bl, h);
  if (rc != NS_OK) { ns_weights_destroy(h); return rc; }
  *out = h;
  return NS_OK;
}

int ns_pack_depthnet_ex(int n_branch, const int* hidden_sizes, int n_trunk, const int* cat_sizes,
                        const float* const* w, const float* const* b, int dtype, ns_weights** out) {
  NS_REQUIRE(out && w && b && hidden_sizes && cat_sizes, "null pointer");
  *out = nullptr;
  if (n_branch < 1 || n_branch > 64 || n_trunk < 1 || n_trunk > 64 ||
      !(dtype == NS_DTYPE_F32 || dtype == NS_DTYPE_BF16 || dtype == NS_DTYPE_F16 || dtype == NS_DTYPE_F16X3 ||
        dtype == NS_DTYPE_F16M)) {
    ns::set_error("ns_pack_depthnet: unsupported network (n_branch=%d n_trunk=%d dtype=%d)", n_branch, n_trunk, dtype);
    return NS_E_UNSUPPORTED;
  }
  int cmax = 0;
  for (int i = 0; i < n_branch; ++i) NS_REQUIRE(hidden_sizes[i] >= 1 && hidden_sizes[i] <= 4096, "bad hidden size");
  for (int i = 0; i < n_trunk; ++i) {
    NS_REQUIRE(cat_sizes[i] >= 1, "bad trunk size");
    if (cat_sizes[i] > cmax) cmax = cat_sizes[i];
  }
  if (cmax > 256) {
    ns::set_error("ns_pack_depthnet: trunk layers wider than 256 (%d) have no kernel (multires 10, trunk widths <= 256)", cmax);
    return NS_E_UNSUPPORTED;
  }
  const int n_tensors = 3 * n_branch + n_trunk + 1;
  for (int i = 0; i < n_tensors; ++i) NS_REQUIRE(w[i] && b[i], "null weight tensor");
  const int E3 = 63, E6 = 126, EIN = E3 + E3 + E6;   // embeddings of origin, direction, sphere intersections: 252
  // 1 + 2. the three affine skip branches composed into the first trunk layer, in fp64
  const int t0 = 3 * n_branch, C0 = cat_sizes[0];
  std::vector<float> F32, fb32;
  fold_depthnet_front(n_branch, hidden_sizes, C0, w, b, F32, fb32);

  // 3. the stream: layer 0 (252 -> Wp), trunk layers 1.. (Wp -> Wp), head (Wp -> 1); every trunk layer is zero-padded
  //    to ONE width Wp in {128, 256}: a padded row has zero weights and bias, LeakyReLU(0) = 0 feeds zero columns
  const int W = cmax <= 128 ? 128 : 256, NB = W / 32;
  const int layout = dtype == NS_DTYPE_F32 ? 0 : 16;
  const bool mixed = dtype == NS_DTYPE_F16M;
  if (mixed && !(W == 256 && n_trunk == 10)) {
    ns::set_error("ns_pack_depthnet: mixed fp16 operands (NS_DTYPE_F16M) are built for the production trunk (ten layers padded to "
                  "256), got %d layers of width <= %d; use NS_DTYPE_F16X3", n_trunk, cmax);
    return NS_E_UNSUPPORTED;
  }
  if (dtype == NS_DTYPE_F16 || dtype == NS_DTYPE_F16X3 || mixed) {   // fp16 operands: refuse what would silently become +-inf
    bool ok = fits_f16(F32.data(), F32.size());
    for (int i = 1; i < n_trunk && ok; ++i) ok = fits_f16(w[t0 + i], static_cast<size_t>(cat_sizes[i]) * cat_sizes[i - 1]);
    ok = ok && fits_f16(w[t0 + n_trunk], cat_sizes[n_trunk - 1]);
    if (!ok) {
      ns::set_error("ns_pack_depthnet: a (folded) weight exceeds fp16's range (65504); use bf16 or f32 operands");
      return NS_E_UNSUPPORTED;
    }
  }
  Builder bl(mixed ? NS_DTYPE_F16 : dtype);
  auto padded_ident = [](int in_f) { return [in_f](int k) { return k < in_f ? k : -1; }; };
  if (mixed) {
    // the mixed program (depthnet_mix_kernel): layers 0 .. KX-1 as split (hi, lo) chunks, TWICE in a row -- a wave takes its four
    // tiles through them two at a time -- then layers KX .. 9 and the head as plain fp16 chunks; one bias image for all
    const int NSB = W / 16, NKB = W / 32, KX = NS_F16M_SPLIT_LAYERS;
    auto layer = [&](int i) {
      if (i == 0)
        bl.layer_ob16(F32.data(), C0, EIN, NSB,
                      {{2, [](int k) { return nsmlp::embedN_col16(k, 3, 10); }},
                       {2, [](int k) { const int c = nsmlp::embedN_col16(k, 3, 10); return c < 0 ? -1 : E3 + c; }},
                       {4, [](int k) { const int c = nsmlp::embedN_col16(k, 6, 10); return c < 0 ? -1 : 2 * E3 + c; }}});
      else
        bl.layer_ob16(w[t0 + i], cat_sizes[i], cat_sizes[i - 1], NSB, {{NKB, padded_ident(cat_sizes[i - 1])}});
    };
    bl.add_bias16(fb32.data(), C0, NSB);
    for (int i = 1; i < n_trunk; ++i) bl.add_bias16(b[t0 + i], cat_sizes[i], NSB);
    bl.add_bias16(b[t0 + n_trunk], 1, 1);
    bl.split = true;
    for (int rep = 0; rep < 2; ++rep)
      for (int i = 0; i < KX; ++i) layer(i);
    bl.split = false;
    for (int i = KX; i < n_trunk; ++i) layer(i);
    bl.layer_ob16(w[t0 + n_trunk], 1, cat_sizes[n_trunk - 1], 1, {{NKB, padded_ident(cat_sizes[n_trunk - 1])}});
  } else if (layout == 0) {
    // k-major (fp32): input blocks e_o (2), e_d (2), e_x (4) in the slot order of embed3 / embed6
    auto col3 = [](int k) { return nsmlp::embed3_col(k, 10); };
    bl.add_bias(fb32.data(), C0, NB);
    bl.segment(F32.data(), C0, EIN, NB, 8, [&](int k) {
      if (k < 64) return col3(k);
      if (k < 128) { const int c = col3(k - 64); return c < 0 ? -1 : E3 + c; }
      const int c = nsmlp::embed6_col(k - 128);
      return c < 0 ? -1 : 2 * E3 + c;
    });
    for (int i = 1; i < n_trunk; ++i) {
      bl.add_bias(b[t0 + i], cat_sizes[i], NB);
      bl.segment(w[t0 + i], cat_sizes[i], cat_sizes[i - 1], NB, NB, padded_ident(cat_sizes[i - 1]));
    }
    bl.add_bias(b[t0 + n_trunk], 1, 1);
    bl.segment(w[t0 + n_trunk], 1, cat_sizes[n_trunk - 1], 1, NB, padded_ident(cat_sizes[n_trunk - 1]));
  } else {
    // 16x16x32 engine: K-blocks e_o (2), e_d (2), e_x (4) in the slot order of embedN_16
    const int NSB = W / 16, NKB = W / 32;
    bl.add_bias16(fb32.data(), C0, NSB);
    bl.layer_ob16(F32.data(), C0, EIN, NSB,
                  {{2, [](int k) { return nsmlp::embedN_col16(k, 3, 10); }},
                   {2, [](int k) { const int c = nsmlp::embedN_col16(k, 3, 10); return c < 0 ? -1 : E3 + c; }},
                   {4, [](int k) { const int c = nsmlp::embedN_col16(k, 6, 10); return c < 0 ? -1 : 2 * E3 + c; }}});
    for (int i = 1; i < n_trunk; ++i) {
      bl.add_bias16(b[t0 + i], cat_sizes[i], NSB);
      bl.layer_ob16(w[t0 + i], cat_sizes[i], cat_sizes[i - 1], NSB, {{NKB, padded_ident(cat_sizes[i - 1])}});
    }
    bl.add_bias16(b[t0 + n_trunk], 1, 1);
    bl.layer_ob16(w[t0 + n_trunk], 1, cat_sizes[n_trunk - 1], 1, {{NKB, padded_ident(cat_sizes[n_trunk - 1])}});
  }

  ns_weights* h = new ns_weights();
  std::memset(h, 0, sizeof(*h));
  h->kind = NS_KIND_DEPTHNET; h->dtype = dtype; h->width = W; h->depth = n_trunk; h->skip = -1; h->layout = layout;
  int rc = finish(bl, h);
  if (rc != NS_OK) { ns_weights_destroy(h); return rc; }
  *out = h;
  return NS_OK;
}

int ns_pack_depthnet(int n_layers, int width, const float* const* w, const float* const* b, int dtype,
                     ns_weights** out) {
  NS_REQUIRE(n_layers >= 1 && n_layers <= 64 && width >= 1, "bad shape");
  std::vector<int> sizes(n_layers, width);
  return ns_pack_depthnet_ex(n_layers, sizes.data(), n_layers, sizes.data(), w, b, dtype, out);
}

int ns_fold_depthnet_front(int n_branch, const int* hidden_sizes, int c0, const float* const* w, const float* const* b,
                           float* F_out, float* bias_out) {
  NS_REQUIRE(hidden_sizes && w && b && F_out && bias_out && n_branch >= 1 && c0 >= 1, "bad argument");
  for (int i = 0; i < 3 * n_branch + 1; ++i) NS_REQUIRE(w[i] && b[i], "null weight tensor");
  std::vector<float> F, fb;
  fold_depthnet_front(n_branch, hidden_sizes, c0, w, b, F, fb);
  std::memcpy(F_out, F.data(), F.size() * sizeof(float));
  std::memcpy(bias_out, fb.data(), fb.size() * sizeof(float));
  return NS_OK;
}

int ns_fold_nerf_views(int W, const float* w_feature, const float* b_feature, const float* w_views, const float* b_views,
                       float* w_out, float* b_out) {
  NS_REQUIRE(W >= 2 && w_feature && b_feature && w_views && b_views && w_out && b_out, "bad argument");
  std::vector<float> wv, bv;
  fold_nerf_views(W, w_feature, b_feature, w_views, b_views, wv, bv);
  std::memcpy(w_out, wv.data(), wv.size() * sizeof(float));
  std::memcpy(b_out, bv.data(), bv.size() * sizeof(float));
  return NS_OK;
}

void ns_weights_destroy(ns_weights* w) {
  if (!w) return;
  if (w->stream_dev) (void)hipFree(w->stream_dev);
  if (w->bias_dev) (void)hipFree(w->bias_dev);
  delete w;
}

int64_t ns_weights_stream_bytes(const ns_weights* w) {
  return w ? static_cast<int64_t>(w->n_slabs) * kSlabBytes : 0;
}

}  // extern "C"
