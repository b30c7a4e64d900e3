"""Host-side logic that needs no GPU: operand-type tables against the header, the PSNR-guard settings, workspace sizes."""

import os
import re

import pytest

from nerf_sampling_amd import _lib, ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "nerf_sampling_hip.h")).read()


def test_operand_type_codes_match_the_header():
    defines = dict(re.findall(r"#define (NS_DTYPE_[A-Z0-9]+) (\d+)", HEADER))
    assert {k: int(v) for k, v in defines.items()} == {"NS_DTYPE_F32": _lib.DTYPE_F32, "NS_DTYPE_BF16": _lib.DTYPE_BF16,
                                                       "NS_DTYPE_F16": _lib.DTYPE_F16, "NS_DTYPE_F16X3": _lib.DTYPE_F16X3,
                                                       "NS_DTYPE_F16M": _lib.DTYPE_F16M}
    for name, code in (("f32", 0), ("bf16", 1), ("f16", 2), ("f16x3", 3), ("f16m", 4)):
        assert ops.dtype_code(name) == code
    assert int(re.search(r"#define NS_F16M_SPLIT_LAYERS (\d+)", HEADER).group(1)) == 3     # odd: the split layers end in set A


def test_psnr_guard_settings():
    try:
        ops.set_psnr_guard(False, threshold=16.0, depthnet="f16x3")
        assert ops.depthnet_dtype_for("bf16") == "f16" and ops.depthnet_dtype_for("f32") == "f32"   # the plain pairing
        ops.set_psnr_guard(True)
        assert ops.psnr_guard() and ops.depthnet_dtype_for("bf16") == "f16x3" and ops.depthnet_dtype_for("f16") == "f16x3"
        assert ops.depthnet_dtype_for("f32") == "f32" and ops.depthnet_dtype_for("f16x3") == "f16x3"   # fp32-grade fields: no guard
        ops.set_psnr_guard(True, depthnet="f16m")
        assert ops.depthnet_dtype_for("bf16") == "f16m"
        with pytest.raises(ValueError):
            ops.set_psnr_guard(True, depthnet="bf16")
        with pytest.raises(ValueError):
            ops.set_psnr_guard(True, threshold=-1.0)
        ops.set_psnr_guard(True, threshold=0.0)             # 0 = every ray
    finally:
        ops.set_psnr_guard(False, threshold=16.0, depthnet="f16x3")


def test_workspace_sizes_cover_their_parts():
    lib = _lib.load()
    for R in (1, 1000, 640000):
        fused = int(lib.ns_render_fused_workspace_bytes(R))
        # o, d, viewdirs | DepthNet depth | guard: z_last, raw_last | selective guard: counter, 64-B records, compact o, d, viewdirs
        assert fused >= R * (36 + 4 + 4 + 16) + 256 + R * (64 + 36)
        chain = int(lib.ns_render_workspace_bytes(R, 64))
        assert chain >= R * (36 + 4 + 4 + 16) + R * 64 * (4 + 16)
        assert fused < chain or R == 1                       # per-sample arrays are what the one-kernel renderer does without
        hier = int(lib.ns_hier_workspace_bytes(R, 64, 128))
        assert hier >= R * 36 + R * 64 * (4 + 16 + 4) + R * 192 * (4 + 16)
    assert int(lib.ns_render_fused_workspace_bytes(-1)) == 0 and int(lib.ns_hier_workspace_bytes(10, 2, 8)) == 0
