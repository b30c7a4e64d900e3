"""ctypes binding of libnerf_sampling_hip.so (the C ABI declared in include/nerf_sampling_hip.h).

There is NO fallback: if the shared library is missing or fails to load, importing any
operator raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C nerf_sampling_amd/csrc``.
"""

from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NS_LIB_PATH") or os.path.join(_HERE, "libnerf_sampling_hip.so")

NS_OK = 0
DTYPE_F32, DTYPE_BF16, DTYPE_F16, DTYPE_F16X3, DTYPE_F16M = 0, 1, 2, 3, 4
MODE_DEPTH_ONLY, MODE_UNIFORM, MODE_GAUSSIAN = 0, 1, 2

_p = C.c_void_p
_i = C.c_int
_i64 = C.c_int64
_f = C.c_float


class RenderArgs(C.Structure):
    """struct ns_render_args"""

    _fields_ = [
        ("depthnet", _p), ("nerf", _p),
        ("o_dev", _p), ("d_dev", _p), ("viewdirs_dev", _p), ("R", _i64),
        ("H", _i), ("W", _i), ("row0", _i), ("row1", _i),
        ("fx", _f), ("fy", _f), ("cx", _f), ("cy", _f),
        ("c2w", _f * 12),
        ("mode", _i), ("N", _i), ("std_", _f), ("noise_dev", _p),
        ("near_", _f), ("far_", _f), ("sphere_radius", _f), ("white_bkgd", _i),
        ("workspace_dev", _p),
        ("rgb_dev", _p), ("disp_dev", _p), ("z_dev", _p), ("weights_dev", _p), ("pts_dev", _p),
        ("ev_mlp_begin", _p), ("ev_mlp_end", _p),
        ("rgb_stride", _i64), ("disp_stride", _i64),
        ("nerf_guard", _p), ("guard_threshold", _f),
    ]


class GemmProblem(C.Structure):
    """struct ns_gemm_problem"""

    _fields_ = [
        ("A_dev", _p), ("sa0", _i64), ("sa1", _i64),
        ("B_dev", _p), ("sb0", _i64), ("sb1", _i64),
        ("bias_dev", _p),
        ("C_dev", _p), ("ldc", _i64),
        ("M", _i), ("N", _i), ("K", _i),
        ("accumulate", _i), ("act", _i), ("dact", _i),
        ("dact_ref_dev", _p), ("ld_ref", _i64),
        ("a_rowsum_dev", _p),
    ]


class HierArgs(C.Structure):
    """struct ns_hier_args"""

    _fields_ = [
        ("coarse", _p), ("fine", _p),
        ("o_dev", _p), ("d_dev", _p), ("viewdirs_dev", _p), ("R", _i64),
        ("H", _i), ("W", _i), ("row0", _i), ("row1", _i),
        ("fx", _f), ("fy", _f), ("cx", _f), ("cy", _f),
        ("c2w", _f * 12),
        ("Nc", _i), ("Nf", _i), ("lindisp", _i), ("white_bkgd", _i), ("near_", _f), ("far_", _f),
        ("t_rand_dev", _p), ("u_dev", _p), ("workspace_dev", _p),
        ("rgb_dev", _p), ("disp_dev", _p), ("z_dev", _p), ("weights_dev", _p), ("raw_dev", _p),
        ("ev_mlp_begin", _p), ("ev_mlp_end", _p),
        ("rgb_stride", _i64), ("disp_stride", _i64),
        ("ev_coarse_begin", _p), ("ev_coarse_end", _p),
    ]


# name -> (restype, argtypes); every symbol include/nerf_sampling_hip.h declares
SIGNATURES = {
    "ns_last_error": (C.c_char_p, []),
    "ns_version": (_i, []),
    "ns_device_cu_count": (_i, []),
    "ns_debug_set": (_i, [C.c_char_p, _i]),
    "ns_get_rays": (_i, [_i, _i, _f, _f, _f, _f, _p, _i, _i, _f, _f, _p, _p, _p, _p, _p]),
    "ns_sphere_intersect": (_i, [_p, _p, _i64, _f, _p, _p, _p]),
    "ns_solve_quadratic": (_i, [_p, _p, _p, _i64, _p, _p]),
    "ns_posenc": (_i, [_p, _i64, _i, _i, _p, _p]),
    "ns_pack_nerf": (_i, [_i, _i, _i, _p, _p, _i, C.POINTER(_p)]),
    "ns_pack_nerf_ex": (_i, [_i, _i, C.c_uint32, _i, _i, _p, _p, _i, C.POINTER(_p)]),
    "ns_nerf_out_channels": (_i, [_p]),
    "ns_pack_depthnet": (_i, [_i, _i, _p, _p, _i, C.POINTER(_p)]),
    "ns_pack_depthnet_ex": (_i, [_i, _p, _i, _p, _p, _p, _i, C.POINTER(_p)]),
    "ns_fold_depthnet_front": (_i, [_i, _p, _i, _p, _p, _p, _p]),
    "ns_fold_nerf_views": (_i, [_i, _p, _p, _p, _p, _p, _p]),
    "ns_weights_destroy": (None, [_p]),
    "ns_weights_stream_bytes": (_i64, [_p]),
    "ns_depthnet_forward": (_i, [_p, _p, _p, _i64, _f, _f, _f, _p, _p]),
    "ns_place_samples": (_i, [_i, _p, _p, _p, _p, _i64, _i, _f, _p, _p, _p]),
    "ns_nerf_forward": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i, _p, _p]),
    "ns_nerf_forward_embedded": (_i, [_p, _p, _i64, _p, _p]),
    "ns_raw2outputs": (_i, [_p, _p, _p, _p, _i64, _i, _i, _p, _p, _p, _p, _p, _p, _p]),
    "ns_raw2outputs_strided": (_i, [_p, _p, _p, _p, _i64, _i, _i, _p, _i64, _p, _i64, _p, _p, _p, _p, _p]),
    "ns_coarse_z": (_i, [_p, _p, _i64, _i, _i, _p, _p, _p]),
    "ns_sample_pdf": (_i, [_p, _p, _i64, _i, _i, _p, _p, _p]),
    "ns_importance_z": (_i, [_p, _p, _i64, _i, _i, _p, _p, _p]),
    "ns_sort_rows": (_i, [_p, _i64, _i, _p, _p]),
    "ns_points_along_rays": (_i, [_p, _p, _p, _i64, _i, _p, _p]),
    "ns_argmax_gather": (_i, [_p, _p, _p, _i64, _i, _p, _p, _p, _p]),
    "ns_render_workspace_bytes": (_i64, [_i64, _i]),
    "ns_render_rays_depthnet": (_i, [C.POINTER(RenderArgs), _p]),
    "ns_render_fused_supported": (_i, [_p, _i, _i]),
    "ns_render_fused_workspace_bytes": (_i64, [_i64]),
    "ns_render_rays_fused": (_i, [C.POINTER(RenderArgs), _p]),
    "ns_hier_workspace_bytes": (_i64, [_i64, _i, _i]),
    "ns_render_rays_hierarchical": (_i, [C.POINTER(HierArgs), _p]),
    "ns_gemm_strided": (_i, [_p, _i64, _i64, _p, _i64, _i64, _p, _p, _i64, _i, _i, _i, _i, _p]),
    "ns_gemm_fused": (_i, [_p, _i64, _i64, _p, _i64, _i64, _p, _p, _i64, _i, _i, _i, _i, _i, _i, _p, _i64, _p, _p]),
    "ns_gemm_fused_batched": (_i, [C.POINTER(GemmProblem), _i, _p]),
    "ns_colsum": (_i, [_p, _i64, _i, _i, _p, _p]),
    "ns_act_forward": (_i, [_p, _i64, _i, _p]),
    "ns_act_backward": (_i, [_p, _p, _i64, _i, _p]),
    "ns_posenc_backward": (_i, [_p, _p, _i64, _i, _i, _p, _p]),
    "ns_points_backward": (_i, [_p, _p, _i64, _i, _p, _p]),
    "ns_adam_step": (_i, [_p, _p, _p, _p, _i64, _f, _f, _f, _f, _i, _p]),
    "ns_adam_step_dev": (_i, [_p, _p, _p, _p, _i64, _f, _p, _f, _f, _f, _p, _p]),
    "ns_add_i32": (_i, [_p, _i, _p]),
    "ns_adam_step_multi_dev": (_i, [_p, _i, _i64, _f, _p, _f, _f, _f, _p, _p]),
    "ns_event_create": (_i, [C.POINTER(_p)]),
    "ns_event_destroy": (None, [_p]),
    "ns_event_record": (_i, [_p, _p]),
    "ns_stream_wait_event": (_i, [_p, _p]),
    "ns_event_elapsed_ms": (_i, [_p, _p, C.POINTER(_f)]),
}

_lib = None


class NativeLibraryError(RuntimeError):
    pass


def load():
    """dlopen the HIP library and bind every entry point; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} not found: the HIP extension is not built (make -C nerf_sampling_amd/csrc). "
            "There is no CPU fallback."
        )
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise NativeLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != NS_OK:
        msg = load().ns_last_error().decode("utf-8", "replace")
        kind = {-1: ValueError, -2: NotImplementedError}.get(rc, RuntimeError)
        raise kind(f"{what or 'libnerf_sampling_hip'} failed (rc={rc}): {msg}")
