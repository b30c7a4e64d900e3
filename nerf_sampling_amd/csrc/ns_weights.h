// Internal layout of the opaque ns_weights handle (host side).
#pragma once
#include <cstddef>
#include <cstdint>

struct ns_weights {
  int kind;       // 0 = NeRF, 1 = DepthNet
  int dtype;      // NS_DTYPE_*
  int width;      // hidden width W (128 or 256); DepthNet: the width every trunk layer is zero-padded to
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
