// The DepthNet branch of render_rays_test (nerf_utils.py:836-865) as one call: a fixed chain of
// kernel launches on the caller's stream, intermediates in a caller-provided workspace.
#include "ns_common.h"
#include "ns_weights.h"

namespace {

inline int64_t align256(int64_t x) { return (x + 255) & ~static_cast<int64_t>(255); }

struct Layout {
  int64_t o, d, view, mean, z, raw, total;
};

Layout layout(int64_t R, int N) {
  Layout l{};
  int64_t off = 0;
  l.o = off; off += align256(R * 3 * 4);
  l.d = off; off += align256(R * 3 * 4);
  l.view = off; off += align256(R * 3 * 4);
  l.mean = off; off += align256(R * 4);
  l.z = off; off += align256(R * N * 4);
  l.raw = off; off += align256(R * N * 16);
  l.total = off;
  return l;
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
  rc = ns_nerf_forward(a->nerf, nullptr, o, d, z, view, R, N, raw, stream);
  if (rc != NS_OK) return rc;
  if (a->ev_mlp_end) NS_HIP(hipEventRecord(static_cast<hipEvent_t>(a->ev_mlp_end), ns::as_stream(stream)));
  return ns_raw2outputs(raw, z, d, nullptr, R, N, a->white_bkgd, a->rgb_dev, a->disp_dev, nullptr, nullptr, nullptr,
                        a->weights_dev, stream);
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

int ns_event_elapsed_ms(void* begin, void* end, float* ms) {
  NS_REQUIRE(begin && end && ms, "null pointer");
  NS_HIP(hipEventSynchronize(static_cast<hipEvent_t>(end)));
  NS_HIP(hipEventElapsedTime(ms, static_cast<hipEvent_t>(begin), static_cast<hipEvent_t>(end)));
  return NS_OK;
}

}  // extern "C"
