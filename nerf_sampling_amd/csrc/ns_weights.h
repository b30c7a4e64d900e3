// Internal layout of the opaque ns_weights handle (host side).
#pragma once
#include <cstddef>
#include <cstdint>

struct ns_weights {
  int kind;       // 0 = NeRF, 1 = DepthNet
  int dtype;      // NS_DTYPE_*
  int width;      // hidden width the kernels run (128 or 256): every layer is zero-padded to it at pack time
  int depth;      // NeRF: D;  DepthNet: number of trunk layers (the skip branches are folded into trunk layer 0)
  int skip;       // NeRF: first skip index or -1 (legacy view of skip_mask)
  uint32_t skip_mask;   // NeRF: bit i set <=> i in skips, i.e. layer i + 1 sees cat[x, h] (run_nerf_helpers.py:117-118)
  int use_viewdirs;     // NeRF: 1 = alpha / feature / views / rgb head (:120-131), 0 = output_linear (:132-133)
  int out_ch;           // NeRF: channels of raw (4 with view directions; output_ch of output_linear otherwise)
  int layout;     // weight stream order: 0 = k-major slabs (consume<>), 16 = 16x16x32 output-sub-block-major (layer_ob16<>)
  void* stream_dev;      // weight stream, n_slabs * 16 KiB, consumed cyclically by every workgroup
  uint32_t n_slabs;
  float* bias_dev;       // all biases in LDS image order, fp32
  int bias_floats;
};

enum { NS_KIND_NERF = 0, NS_KIND_DEPTHNET = 1 };

// Per-ray outputs of a radiance-field pass that composites in its own epilogue (internal: ns_render.cpp -> the 16-bit MLP
// kernel).  mean_dev != NULL: the kernel also PLACES the samples (sample_points_around_mean "uniform") from the DepthNet
// depth; otherwise depths come from the z array given to the forward call.
struct ns_composite_args {
  const float* mean_dev;   // [R] or NULL
  float std_;
  int white_bkgd;
  float* rgb_dev; int64_t rgb_stride;
  float* disp_dev; int64_t disp_stride;
  float* weights_dev;      // [R,N] or NULL
  float* z_out_dev;        // [R,N] or NULL
  float* pts_out_dev;      // [R,N,3] or NULL
  const float* sigma_last_dev;   // NULL, or [R,4] raw of every ray's LAST sample from the guard pass: its sigma (element 3)
                                 // replaces the kernel's own for that sample (ns_render_args::nerf_guard)
  // the selective guard (ns_render_args::guard_threshold > 0): records of the rays whose own |sigma_last| < fix_thr (16 floats
  // each, Nerf16Args::fix_rec), counted in *fix_count_dev (zeroed by the caller)
  float fix_thr;
  uint32_t* fix_count_dev;
  float* fix_rec_dev;
};
// the selective guard's two small kernels (ns_composite.hip): compact inputs of the flagged rays' last samples for the fp32-grade
// network; then the flagged pixels from the records and the re-evaluated sigma (raw_c [.,4], element 3)
int ns_fix_gather(const float* rec_dev, const uint32_t* count_dev, int64_t cap, const float* o_dev, const float* d_dev,
                  const float* view_dev, float* o_c, float* d_c, float* view_c, float* z_c, void* stream);
int ns_fix_last_sample(const float* rec_dev, const uint32_t* count_dev, int64_t cap, const float* raw_c, int N, int white_bkgd,
                       float* rgb_dev, int64_t rgb_stride, float* disp_dev, int64_t disp_stride, float* weights_dev, void* stream);
// internal helpers of the guard pass (ns_rays.hip, ns_composite.hip)
int ns_place_last_sample(const float* mean_dev, int64_t R, int N, float std_, float* z_last_dev, void* stream);
int ns_patch_sigma_last(float* raw_dev, const float* raw_last_dev, int64_t R, int N, void* stream);
bool ns_nerf_can_composite(const ns_weights* net, int N);
