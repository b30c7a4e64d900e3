// Internal layout of the opaque ns_weights handle (host side).
#pragma once
#include <cstddef>
#include <cstdint>

struct ns_weights {
  int kind;       // 0 = NeRF, 1 = DepthNet
  int dtype;      // NS_DTYPE_*
  int width;      // hidden width W (128 or 256); DepthNet: the width every trunk layer is zero-padded to
  int depth;      // NeRF: D;  DepthNet: number of trunk layers (the skip branches are folded into trunk layer 0)
  int skip;       // NeRF: skip index or -1
  int layout;     // weight stream order: 0 = k-major slabs (consume<>), 16 = 16x16x32 output-sub-block-major (layer_ob16<>)
  void* stream_dev;      // weight stream, n_slabs * 16 KiB, consumed cyclically by every workgroup
  uint32_t n_slabs;
  float* bias_dev;       // all biases in LDS image order, fp32
  int bias_floats;
};

enum { NS_KIND_NERF = 0, NS_KIND_DEPTHNET = 1 };
