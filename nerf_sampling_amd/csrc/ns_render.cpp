// The DepthNet branch of render_rays_test (nerf_utils.py:836-865) as one call: a fixed chain of
// kernel launches on the caller's stream, intermediates in a caller-provided workspace.
#include "ns_common.h"
#include "ns_weights.h"

int ns_nerf_forward_ob16(const ns_weights* net, const float* pts_dev, const float* o_dev, const float* d_dev,
                         const float* z_dev, const float* viewdirs_dev, const float* x90_dev, int64_t S, int N,
                         float* raw_dev, hipStream_t stream, const ns_composite_args* comp);
int ns_nerf_forward_x3(const ns_weights* net, const float* pts_dev, const float* o_dev, const float* d_dev,
                       const float* z_dev, const float* viewdirs_dev, const float* x90_dev, int64_t S, int N,
                       float* raw_dev, hipStream_t stream, const uint32_t* count_dev);

namespace {

inline int64_t align256(int64_t x) { return (x + 255) & ~static_cast<int64_t>(255); }

struct Layout {
  int64_t o, d, view, mean, z_last, raw_last, z, raw, total;
};

Layout layout(int64_t R, int N) {
  Layout l{};
  int64_t off = 0;
  l.o = off; off += align256(R * 3 * 4);
  l.d = off; off += align256(R * 3 * 4);
  l.view = off; off += align256(R * 3 * 4);
  l.mean = off; off += align256(R * 4);
  l.z_last = off; off += align256(R * 4);          // the guard pass (ns_render_args::nerf_guard): depth and raw of every
  l.raw_last = off; off += align256(R * 16);       // ray's last sample
  l.z = off; off += align256(R * N * 4);
  l.raw = off; off += align256(R * N * 16);
  l.total = off;
  return l;
}

// The guard pass: the last sample of every ray through a second, fp32-grade (F16X3) handle of the same network; its raw
// lands in raw_last [R,4].  (nerf_utils.py:836-865 composites that sample with dist = 1e10, sampling_trainer.py:176-180.)
int guard_check(const ns_render_args* a, int N) {
  const ns_weights* gnet = a->nerf_guard;
  if (!(gnet->kind == NS_KIND_NERF && gnet->out_ch == 4 && gnet->use_viewdirs && gnet->width == a->nerf->width &&
        gnet->depth == a->nerf->depth && gnet->skip_mask == a->nerf->skip_mask)) {
    ns::set_error("nerf_guard must be another packing of the same network (a NeRF with view directions, same D / W / skips)");
    return NS_E_INVALID;
  }
  if (a->mode != NS_MODE_UNIFORM || N < 2) {
    ns::set_error("nerf_guard: the guard pass is defined for uniform placement with n_samples >= 2");
    return NS_E_UNSUPPORTED;
  }
  return NS_OK;
}
int guard_pass(const ns_render_args* a, const float* o, const float* d, const float* view, const float* mean, int64_t R,
               int N, float* z_last, float* raw_last, void* stream) {
  const ns_weights* gnet = a->nerf_guard;
  int rc = guard_check(a, N);
  if (rc != NS_OK) return rc;
  rc = ns_place_last_sample(mean, R, N, a->std_, z_last, stream);
  if (rc != NS_OK) return rc;
  return ns_nerf_forward(gnet, nullptr, o, d, z_last, view, R, 1, raw_last, stream);
}

}  // namespace

extern "C" {

int64_t ns_render_workspace_bytes(int64_t R, int N) {
  if (R < 0 || N < 1) return 0;
  return layout(R, N).total;
}

int ns_render_rays_depthnet(const ns_render_args* a, void* stream) {
  NS_REQUIRE(a, "null args");
  NS_REQUIRE(a->depthnet && a->nerf, "both networks are required");
  NS_REQUIRE(a->nerf->kind == NS_KIND_NERF && a->nerf->out_ch == 4 && a->nerf->use_viewdirs,
             "the one-call path composites raw [R,N,4] of a network with view directions");
  if (a->o_dev == nullptr ? (a->row1 == a->row0 || a->W == 0) : a->R == 0) return NS_OK;  // nothing to render
  if (a->o_dev == nullptr && a->H == 0 && a->R == 0) return NS_OK;
  NS_REQUIRE(a->workspace_dev && a->rgb_dev && a->disp_dev, "workspace, rgb and disp are required");
  int N = a->mode == NS_MODE_DEPTH_ONLY ? 1 : a->N;
  NS_REQUIRE(N >= 1, "bad sample count");
  int64_t R = a->R;
  if (!a->o_dev) {
    NS_REQUIRE(a->row0 >= 0 && a->row1 <= a->H && a->row0 <= a->row1 && a->W > 0, "bad camera rows");
    R = static_cast<int64_t>(a->row1 - a->row0) * a->W;
  } else {
    NS_REQUIRE(a->d_dev && a->viewdirs_dev, "explicit rays need o, d and viewdirs");
  }
  if (R == 0) return NS_OK;
  const Layout l = layout(R, N);
  char* ws = static_cast<char*>(a->workspace_dev);
  NS_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0, "workspace must be 256-byte aligned");
  const float* o = a->o_dev;
  const float* d = a->d_dev;
  const float* view = a->viewdirs_dev;
  int rc;
  if (!o) {
    float* wo = reinterpret_cast<float*>(ws + l.o);
    float* wd = reinterpret_cast<float*>(ws + l.d);
    float* wv = reinterpret_cast<float*>(ws + l.view);
    rc = ns_get_rays(a->H, a->W, a->fx, a->fy, a->cx, a->cy, a->c2w, a->row0, a->row1, a->near_, a->far_, wo, wd,
                     wv, nullptr, stream);
    if (rc != NS_OK) return rc;
    o = wo; d = wd; view = wv;
  }
  float* mean = reinterpret_cast<float*>(ws + l.mean);
  float* z = a->z_dev ? a->z_dev : reinterpret_cast<float*>(ws + l.z);
  float* raw = reinterpret_cast<float*>(ws + l.raw);
  rc = ns_depthnet_forward(a->depthnet, o, d, R, a->near_, a->far_, a->sphere_radius, mean, stream);
  if (rc != NS_OK) return rc;
  rc = ns_place_samples(a->mode, o, d, mean, a->noise_dev, R, N, a->std_, a->pts_dev, z, stream);
  if (rc != NS_OK) return rc;
  if (a->ev_mlp_begin) NS_HIP(hipEventRecord(static_cast<hipEvent_t>(a->ev_mlp_begin), ns::as_stream(stream)));
  ns::prod_tiles_hint() = (a->z_dev || a->weights_dev || a->pts_dev) ? 4 : 0;   // per-sample outputs: host copies will run beside the next MLP kernel
  rc = ns_nerf_forward(a->nerf, nullptr, o, d, z, view, R, N, raw, stream);
  ns::prod_tiles_hint() = 0;
  if (rc != NS_OK) return rc;
  if (a->ev_mlp_end) NS_HIP(hipEventRecord(static_cast<hipEvent_t>(a->ev_mlp_end), ns::as_stream(stream)));
  if (a->nerf_guard) {
    float* raw_last = reinterpret_cast<float*>(ws + l.raw_last);
    rc = guard_pass(a, o, d, view, mean, R, N, reinterpret_cast<float*>(ws + l.z_last), raw_last, stream);
    if (rc != NS_OK) return rc;
    rc = ns_patch_sigma_last(raw, raw_last, R, N, stream);
    if (rc != NS_OK) return rc;
  }
  return ns_raw2outputs_strided(raw, z, d, nullptr, R, N, a->white_bkgd, a->rgb_dev, a->rgb_stride ? a->rgb_stride : 3,
                                a->disp_dev, a->disp_stride ? a->disp_stride : 1, nullptr, nullptr, nullptr,
                                a->weights_dev, stream);
}

// ---- the same branch as ONE kernel per ray tile: rays -> DepthNet -> [placement + radiance-field
// MLP + compositing in one persistent kernel].  Per-sample data (z, pts, raw, weights) never reaches HBM unless asked for.
int ns_render_fused_supported(const ns_weights* nerf, int mode, int N) {
  return mode == NS_MODE_UNIFORM && ns_nerf_can_composite(nerf, N) ? 1 : 0;
}

int64_t ns_render_fused_workspace_bytes(int64_t R) {
  if (R < 0) return 0;
  // o, d, viewdirs | DepthNet depth | guard: z_last, raw_last | selective guard: counter, records, compact o, d, viewdirs
  return 3 * align256(R * 12) + 2 * align256(R * 4) + align256(R * 16) + 256 + align256(R * 64) + 3 * align256(R * 12);
}

int ns_render_rays_fused(const ns_render_args* a, void* stream) {
  NS_REQUIRE(a, "null args");
  NS_REQUIRE(a->depthnet && a->nerf, "both networks are required");
  if (!ns_render_fused_supported(a->nerf, a->mode, a->N)) {
    ns::set_error("ns_render_rays_fused: uniform placement, a 16-bit NeRF handle with view directions and n_samples a power "
                  "of two in [2, 64] or a multiple of 64 up to 512 are required (mode %d, N %d); use ns_render_rays_depthnet", a->mode, a->N);
    return NS_E_UNSUPPORTED;
  }
  NS_REQUIRE(!a->noise_dev, "uniform placement takes no noise");
  if (a->o_dev == nullptr ? (a->row1 == a->row0 || a->W == 0) : a->R == 0) return NS_OK;  // nothing to render
  NS_REQUIRE(a->workspace_dev && a->rgb_dev && a->disp_dev, "workspace, rgb and disp are required");
  int64_t R = a->R;
  if (!a->o_dev) {
    NS_REQUIRE(a->row0 >= 0 && a->row1 <= a->H && a->row0 <= a->row1 && a->W > 0, "bad camera rows");
    R = static_cast<int64_t>(a->row1 - a->row0) * a->W;
  } else {
    NS_REQUIRE(a->d_dev && a->viewdirs_dev, "explicit rays need o, d and viewdirs");
  }
  if (R == 0) return NS_OK;
  char* ws = static_cast<char*>(a->workspace_dev);
  NS_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0, "workspace must be 256-byte aligned");
  const float* o = a->o_dev;
  const float* d = a->d_dev;
  const float* view = a->viewdirs_dev;
  int rc;
  if (!o) {
    float* wo = reinterpret_cast<float*>(ws);
    float* wd = reinterpret_cast<float*>(ws + align256(R * 12));
    float* wv = reinterpret_cast<float*>(ws + 2 * align256(R * 12));
    rc = ns_get_rays(a->H, a->W, a->fx, a->fy, a->cx, a->cy, a->c2w, a->row0, a->row1, a->near_, a->far_, wo, wd, wv,
                     nullptr, stream);
    if (rc != NS_OK) return rc;
    o = wo; d = wd; view = wv;
  }
  float* mean = reinterpret_cast<float*>(ws + 3 * align256(R * 12));
  rc = ns_depthnet_forward(a->depthnet, o, d, R, a->near_, a->far_, a->sphere_radius, mean, stream);
  if (rc != NS_OK) return rc;
  ns_composite_args c{};
  c.mean_dev = mean; c.std_ = a->std_; c.white_bkgd = a->white_bkgd;
  c.rgb_dev = a->rgb_dev; c.rgb_stride = a->rgb_stride ? a->rgb_stride : 3;
  c.disp_dev = a->disp_dev; c.disp_stride = a->disp_stride ? a->disp_stride : 1;
  c.weights_dev = a->weights_dev; c.z_out_dev = a->z_dev; c.pts_out_dev = a->pts_dev;
  float* z_last = reinterpret_cast<float*>(ws + 3 * align256(R * 12) + align256(R * 4));
  float* raw_last = reinterpret_cast<float*>(ws + 3 * align256(R * 12) + 2 * align256(R * 4));
  char* fix = ws + 3 * align256(R * 12) + 2 * align256(R * 4) + align256(R * 16);
  // (the fix-up launches the split-operand MLP kernel with a device-side count: another packing of the guard handle, e.g. fp32,
  // takes the every-ray pass through the generic dispatch)
  const bool selective = a->nerf_guard && a->guard_threshold > 0.0f && a->N <= 64 && a->nerf_guard->dtype == NS_DTYPE_F16X3 &&
                         a->nerf_guard->layout == 16;
  if (selective) {         // the kernel flags the rays itself; their last samples are re-evaluated after it
    rc = guard_check(a, a->N);
    if (rc != NS_OK) return rc;
    c.fix_thr = a->guard_threshold;
    c.fix_count_dev = reinterpret_cast<uint32_t*>(fix);
    c.fix_rec_dev = reinterpret_cast<float*>(fix + 256);
    NS_HIP(hipMemsetAsync(fix, 0, 256, ns::as_stream(stream)));
  } else if (a->nerf_guard) {     // (before the event pair: the pair times the fused kernel alone)
    rc = guard_pass(a, o, d, view, mean, R, a->N, z_last, raw_last, stream);
    if (rc != NS_OK) return rc;
    c.sigma_last_dev = raw_last;
  }
  if (a->ev_mlp_begin) NS_HIP(hipEventRecord(static_cast<hipEvent_t>(a->ev_mlp_begin), ns::as_stream(stream)));
  ns::prod_tiles_hint() = (a->z_dev || a->weights_dev || a->pts_dev) ? 4 : 0;   // (see ns_common.h)
  rc = ns_nerf_forward_ob16(a->nerf, nullptr, o, d, nullptr, view, nullptr, R * a->N, a->N, nullptr, ns::as_stream(stream), &c);
  ns::prod_tiles_hint() = 0;
  if (rc != NS_OK) return rc;
  if (a->ev_mlp_end) NS_HIP(hipEventRecord(static_cast<hipEvent_t>(a->ev_mlp_end), ns::as_stream(stream)));
  if (selective) {
    float* o_c = reinterpret_cast<float*>(fix + 256 + align256(R * 64));
    float* d_c = reinterpret_cast<float*>(fix + 256 + align256(R * 64) + align256(R * 12));
    float* v_c = reinterpret_cast<float*>(fix + 256 + align256(R * 64) + 2 * align256(R * 12));
    rc = ns_fix_gather(c.fix_rec_dev, c.fix_count_dev, R, o, d, view, o_c, d_c, v_c, z_last, stream);
    if (rc != NS_OK) return rc;
    rc = ns_nerf_forward_x3(a->nerf_guard, nullptr, o_c, d_c, z_last, v_c, nullptr, R, 1, raw_last, ns::as_stream(stream),
                            c.fix_count_dev);
    if (rc != NS_OK) return rc;
    rc = ns_fix_last_sample(c.fix_rec_dev, c.fix_count_dev, R, raw_last, a->N, a->white_bkgd, c.rgb_dev, c.rgb_stride, c.disp_dev,
                            c.disp_stride, a->weights_dev, stream);
    if (rc != NS_OK) return rc;
  }
  return NS_OK;
}

int64_t ns_hier_workspace_bytes(int64_t R, int Nc, int Nf) {
  if (R < 0 || Nc < 3 || Nf < 0) return 0;
  const int64_t Nt = Nc + Nf;
  // o, d, view | z_c [R,Nc] | raw_c [R,Nc,4] | w_c [R,Nc] | z_f [R,Nt] | raw_f [R,Nt,4]
  return 3 * align256(R * 12) + align256(R * Nc * 4) + align256(R * Nc * 16) + align256(R * Nc * 4) +
         align256(R * Nt * 4) + align256(R * Nt * 16);
}

int ns_render_rays_hierarchical(const ns_hier_args* a, void* stream) {
  NS_REQUIRE(a && a->coarse, "null args / coarse network");
  NS_REQUIRE(a->coarse->out_ch == 4 && a->coarse->use_viewdirs && (!a->fine || (a->fine->out_ch == 4 && a->fine->use_viewdirs)),
             "the one-call path composites raw [R,N,4] of networks with view directions");
  if (a->o_dev == nullptr ? (a->row1 == a->row0 || a->W == 0) : a->R == 0) return NS_OK;  // nothing to render
  NS_REQUIRE(a->workspace_dev && a->rgb_dev && a->disp_dev, "workspace, rgb and disp are required");
  NS_REQUIRE(a->Nc >= 3 && a->Nf >= 0, "needs at least 3 coarse samples");
  int64_t R = a->R;
  if (!a->o_dev) {
    NS_REQUIRE(a->row0 >= 0 && a->row1 <= a->H && a->row0 <= a->row1 && a->W > 0, "bad camera rows");
    R = static_cast<int64_t>(a->row1 - a->row0) * a->W;
  } else {
    NS_REQUIRE(a->d_dev && a->viewdirs_dev, "explicit rays need o, d and viewdirs");
  }
  if (R == 0) return NS_OK;
  char* ws = static_cast<char*>(a->workspace_dev);
  NS_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0, "workspace must be 256-byte aligned");
  const int Nc = a->Nc, Nt = a->Nc + a->Nf;
  int64_t off = 0;
  auto take = [&](int64_t bytes) { char* p = ws + off; off += align256(bytes); return reinterpret_cast<float*>(p); };
  float* wo = take(R * 12); float* wd = take(R * 12); float* wv = take(R * 12);
  float* z_c = take(R * Nc * 4); float* raw_c = take(R * Nc * 16); float* w_c = take(R * Nc * 4);
  float* z_f_ws = take(R * Nt * 4); float* raw_f_ws = take(R * Nt * 16);
  const float* o = a->o_dev; const float* d = a->d_dev; const float* view = a->viewdirs_dev;
  int rc;
  if (!o) {
    rc = ns_get_rays(a->H, a->W, a->fx, a->fy, a->cx, a->cy, a->c2w, a->row0, a->row1, a->near_, a->far_, wo, wd, wv,
                     nullptr, stream);
    if (rc != NS_OK) return rc;
    o = wo; d = wd; view = wv;
  }
  // coarse pass (Trainer.py:579-649); its rgb/disp are not part of the 8-tuple and are not produced
  rc = ns_coarse_z_scalar(a->near_, a->far_, R, Nc, a->lindisp, a->t_rand_dev, z_c, stream);
  if (rc != NS_OK) return rc;
  if (a->ev_coarse_begin) NS_HIP(hipEventRecord(static_cast<hipEvent_t>(a->ev_coarse_begin), ns::as_stream(stream)));
  // A 16-bit field composites in its own epilogue (Nerf16Args::comp == 1: depths from the z array): raw [R,N,4] -- 16 bytes
  // per sample written and read back, 2.6 GB per 800 x 800 frame at 64 + 192 samples -- then never exists.  The coarse
  // pass only yields its weights (its colour goes to a scratch corner of the unused raw_c block).
  const bool chain = ns::debug_flags().hier_chain != 0;
  const bool fuse_c = !chain && a->Nf > 0 && ns_nerf_can_composite(a->coarse, Nc);
  if (fuse_c) {
    ns_composite_args c{};
    c.white_bkgd = a->white_bkgd;
    c.rgb_dev = raw_c; c.rgb_stride = 4; c.disp_dev = raw_c + 3; c.disp_stride = 4;
    c.weights_dev = w_c;
    rc = ns_nerf_forward_ob16(a->coarse, nullptr, o, d, z_c, view, nullptr, R * Nc, Nc, nullptr, ns::as_stream(stream), &c);
  } else {
    rc = ns_nerf_forward(a->coarse, nullptr, o, d, z_c, view, R, Nc, raw_c, stream);
  }
  if (rc != NS_OK) return rc;
  if (a->ev_coarse_end) NS_HIP(hipEventRecord(static_cast<hipEvent_t>(a->ev_coarse_end), ns::as_stream(stream)));
  if (!fuse_c) {
    rc = ns_raw2outputs(raw_c, z_c, d, nullptr, R, Nc, a->white_bkgd, nullptr, nullptr, nullptr, nullptr, nullptr, w_c,
                        stream);
    if (rc != NS_OK) return rc;
  }
  if (a->Nf == 0) {  // no importance samples: the coarse pass is the result
    return ns_raw2outputs_strided(raw_c, z_c, d, nullptr, R, Nc, a->white_bkgd, a->rgb_dev,
                                  a->rgb_stride ? a->rgb_stride : 3, a->disp_dev, a->disp_stride ? a->disp_stride : 1,
                                  nullptr, nullptr, nullptr, a->weights_dev, stream);
  }
  // fine pass (Trainer.py:651-710)
  float* z_f = a->z_dev ? a->z_dev : z_f_ws;
  float* raw_f = a->raw_dev ? a->raw_dev : raw_f_ws;
  rc = ns_importance_z(z_c, w_c, R, Nc, a->Nf, a->u_dev, z_f, stream);
  if (rc != NS_OK) return rc;
  if (a->ev_mlp_begin) NS_HIP(hipEventRecord(static_cast<hipEvent_t>(a->ev_mlp_begin), ns::as_stream(stream)));
  const ns_weights* fine = a->fine ? a->fine : a->coarse;
  if (!chain && ns_nerf_can_composite(fine, Nt)) {
    ns_composite_args c{};
    c.white_bkgd = a->white_bkgd;
    c.rgb_dev = a->rgb_dev; c.rgb_stride = a->rgb_stride ? a->rgb_stride : 3;
    c.disp_dev = a->disp_dev; c.disp_stride = a->disp_stride ? a->disp_stride : 1;
    c.weights_dev = a->weights_dev;
    rc = ns_nerf_forward_ob16(fine, nullptr, o, d, z_f, view, nullptr, R * Nt, Nt, a->raw_dev, ns::as_stream(stream), &c);
    if (rc != NS_OK) return rc;
    if (a->ev_mlp_end) NS_HIP(hipEventRecord(static_cast<hipEvent_t>(a->ev_mlp_end), ns::as_stream(stream)));
    return NS_OK;
  }
  rc = ns_nerf_forward(fine, nullptr, o, d, z_f, view, R, Nt, raw_f, stream);
  if (rc != NS_OK) return rc;
  if (a->ev_mlp_end) NS_HIP(hipEventRecord(static_cast<hipEvent_t>(a->ev_mlp_end), ns::as_stream(stream)));
  return ns_raw2outputs_strided(raw_f, z_f, d, nullptr, R, Nt, a->white_bkgd, a->rgb_dev,
                                a->rgb_stride ? a->rgb_stride : 3, a->disp_dev, a->disp_stride ? a->disp_stride : 1,
                                nullptr, nullptr, nullptr, a->weights_dev, stream);
}

int ns_event_create(void** ev) {
  NS_REQUIRE(ev, "null pointer");
  hipEvent_t e;
  NS_HIP(hipEventCreate(&e));
  *ev = e;
  return NS_OK;
}

void ns_event_destroy(void* ev) {
  if (ev) (void)hipEventDestroy(static_cast<hipEvent_t>(ev));
}

int ns_event_record(void* ev, void* stream) {
  NS_REQUIRE(ev, "null event");
  NS_HIP(hipEventRecord(static_cast<hipEvent_t>(ev), ns::as_stream(stream)));
  return NS_OK;
}

int ns_stream_wait_event(void* stream, void* ev) {
  NS_REQUIRE(ev, "null event");
  NS_HIP(hipStreamWaitEvent(ns::as_stream(stream), static_cast<hipEvent_t>(ev), 0));
  return NS_OK;
}

int ns_event_elapsed_ms(void* begin, void* end, float* ms) {
  NS_REQUIRE(begin && end && ms, "null pointer");
  NS_HIP(hipEventSynchronize(static_cast<hipEvent_t>(end)));
  NS_HIP(hipEventElapsedTime(ms, static_cast<hipEvent_t>(begin), static_cast<hipEvent_t>(end)));
  return NS_OK;
}

}  // extern "C"
