// Internal layout of the opaque ns_weights handle (host side).
#pragma once
#include <cstddef>
#include <cstdint>

struct ns_weights {
  int kind;       // 0 = NeRF, 1 = DepthNet
  int dtype;      // NS_DTYPE_*
  int width;      // hidden width W (128 or 256)
  int depth;      // NeRF: D;  DepthNet: n_layers
  int skip;       // NeRF: skip index or -1
  int layout;     // weight stream order: 0 = k-major slabs (consume<>), 16 = 16x16x32 output-sub-block-major (layer_ob16<>)
  void* stream_dev;      // weight stream, n_slabs * 16 KiB, consumed cyclically by every workgroup
  uint32_t n_slabs;
  float* bias_dev;       // all biases in LDS image order, fp32
  int bias_floats;
  void* scratch_dev;     // DepthNet: per-workgroup stash for the origin/direction branch outputs
  size_t scratch_bytes;
};

enum { NS_KIND_NERF = 0, NS_KIND_DEPTHNET = 1 };
constexpr int kDepthnetMaxGrid = 512;  // workgroup slots the DepthNet stash is sized for
