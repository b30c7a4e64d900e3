"""Tensor-level wrappers over the C ABI: torch CUDA (ROCm) tensors in, torch tensors out.

PyTorch is used for device memory and streams only; every arithmetic step runs in
libnerf_sampling_hip.so.  All inputs must be fp32 device tensors; outputs are freshly allocated on
the same device.  Calls are asynchronous on torch's current stream.
"""

from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import check

Tensor = torch.Tensor

_DTYPES = {"f32": _lib.DTYPE_F32, "fp32": _lib.DTYPE_F32, "float32": _lib.DTYPE_F32,
           "bf16": _lib.DTYPE_BF16, "bfloat16": _lib.DTYPE_BF16,
           "f16": _lib.DTYPE_F16, "fp16": _lib.DTYPE_F16, "float16": _lib.DTYPE_F16,
           "f16x3": _lib.DTYPE_F16X3,     # split fp16 operands: fp32-grade results on the 16-bit engine
           "f16m": _lib.DTYPE_F16M}       # DepthNet only (production trunk): first three layers split, the rest plain fp16
_MODES = {"depth_only": _lib.MODE_DEPTH_ONLY, "uniform": _lib.MODE_UNIFORM, "gaussian": _lib.MODE_GAUSSIAN}

_compute_dtype = "f32"


def set_compute_dtype(name: str) -> None:
    """MFMA operand precision used when a network is packed without an explicit dtype."""
    global _compute_dtype
    if name not in _DTYPES:
        raise ValueError(f"unknown dtype {name!r}; expected one of {sorted(_DTYPES)}")
    _compute_dtype = name


def get_compute_dtype() -> str:
    return _compute_dtype


def dtype_code(name: Optional[str]) -> int:
    return _DTYPES[name or _compute_dtype]


# Operand type of the DepthNet when a network is packed WITHOUT an explicit dtype under compute dtype X.  Under bf16 the
# DepthNet runs on f16 operands: it costs nothing (same MFMA rate, 0.64 ms of a 28 ms frame) and its depth error -- which
# moves a ray's whole +-0.1 sampling window -- drops sevenfold (z rms 4.6e-4 instead of 3.3e-3), which is what the scene
# PSNR of a sharp, trained field is sensitive to (per-image delta to fp32: 0.005 dB instead of 0.020 dB,
# tools/scene_psnr_sweep.py).  A DepthNet whose weights exceed fp16's range falls back to bf16 (DepthNet.packed).
# Explicit requests -- packed("bf16") -- are always honoured as given.
_DEPTHNET_PAIRING = {"bf16": "f16"}


def depthnet_dtype_for(name: Optional[str] = None) -> str:
    name = name or _compute_dtype
    if _psnr_guard and name in ("bf16", "f16"):
        return _guard_depthnet
    return _DEPTHNET_PAIRING.get(name, name)


# PSNR guard of the 16-bit compute dtypes.  north_star's acceptance bar is a scene PSNR within 0.05 dB of the reference's; on
# a fitted scene that renders at 28-30 dB, plain 16-bit operands sit AT that bar (tools/scene_psnr_sweep.py: worst per-image
# |delta| 0.13 dB for bf16 + bf16, 0.052 dB for the default bf16 + f16 pairing), and the error has two sources that touch
# 1/64 of the arithmetic: the DepthNet's depth (it moves a ray's whole sampling window) and sigma of the LAST sample of a
# ray, which the reference composites with dist = 1e10 (alpha = step(sigma), sampling_trainer.py:176-180).  With the guard
# on, the DepthNet runs on (partly) split fp16 operands (_guard_depthnet below) and that one sample per ray is evaluated a
# second time through an "f16x3" packing of the field (ns_render_args::nerf_guard); the other N - 1 samples keep the fast path.
_psnr_guard = False
# Which rays the guard re-evaluates (ns_render_args::guard_threshold): only those whose own 16-bit sigma of the last sample lies
# within this distance of zero -- a sigma beyond it composites to alpha = 0 or 1 whatever its last bits are.  16 is > 30 x the
# rms bf16 error of sigma_last on the production network (bench.py: sigma_last_err_rms ~ 0.5); 0 = every ray.
_guard_threshold = 16.0
# The guard's DepthNet operands.  "f16x3" (the default): every layer on split fp16 operands, fp32-grade depths (z rms 8e-7 on the
# fitted scene) at 2.9 x the fp16 kernel's time: PSNR(build || fp32) 63-64 dB, every frame and band within 0.03 dB.  "f16m": the
# first three trunk layers split, the other seven plain fp16 -- the rounding of the first layers' wide-ranged activations is where
# a TRAINED fp16 DepthNet loses its depth (z rms 8.9e-4 -> 1.2e-4 there; tools/depthnet_mix_check.py) -- at 1.7 x: 50-53 dB, whole
# frames within 0.03 dB but a 60-row band can read 0.06 (tools/guard_experiment.py, tests/test_scene_psnr.py): the economy setting,
# built for the production trunk (ten layers, 256 wide; any other shape falls back to "f16x3").
_guard_depthnet = "f16x3"


def set_psnr_guard(on: bool, threshold: Optional[float] = None, depthnet: Optional[str] = None) -> None:
    global _psnr_guard, _guard_threshold, _guard_depthnet
    _psnr_guard = bool(on)
    if threshold is not None:
        if not threshold >= 0.0:
            raise ValueError("guard threshold must be >= 0 (0 = every ray)")
        _guard_threshold = float(threshold)
    if depthnet is not None:
        if depthnet not in ("f16m", "f16x3"):
            raise ValueError("the guard's DepthNet runs on 'f16m' or 'f16x3' operands")
        _guard_depthnet = depthnet


def psnr_guard() -> bool:
    return _psnr_guard


def psnr_guard_handles(depth_net, nerf):
    """(DepthNet handle, field handle, guard handle or None) for the current compute dtype and guard setting: what the
    one-call renderers take.  ``depth_net`` / ``nerf``: this package's DepthNet / NeRF modules."""
    dn, nf = depth_net.packed(), nerf.packed()
    guard = nerf.packed("f16x3") if (_psnr_guard and nf.dtype in ("bf16", "f16")) else None
    return dn, nf, guard


def _dev(t: Tensor, name: str) -> Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (got {t.device}); this path has no CPU fallback")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _ptr(t: Optional[Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(dev) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


# ---- a1 -------------------------------------------------------------------------------------------
def get_rays(H: int, W: int, K, c2w, row0: int = 0, row1: Optional[int] = None, near: float = 0.0,
             far: float = 1.0, device=None, want_batch: bool = False):
    """rays_o, rays_d, viewdirs [R,3] (and ray_batch [R,11]) for image rows [row0,row1)."""
    lib = _lib.load()
    row1 = H if row1 is None else row1
    device = torch.device(device if device is not None else (c2w.device if isinstance(c2w, Tensor) and c2w.is_cuda else "cuda"))
    c2w_h = (c2w.detach().cpu().numpy() if isinstance(c2w, Tensor) else np.asarray(c2w)).astype(np.float32)[:3, :4]
    c2w_h = np.ascontiguousarray(c2w_h)
    R = (row1 - row0) * W
    o = torch.empty((R, 3), dtype=torch.float32, device=device)
    d = torch.empty_like(o)
    v = torch.empty_like(o)
    b = torch.empty((R, 11), dtype=torch.float32, device=device) if want_batch else None
    check(lib.ns_get_rays(H, W, float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2]),
                          c2w_h.ctypes.data_as(C.c_void_p), row0, row1, float(near), float(far),
                          _ptr(o), _ptr(d), _ptr(v), _ptr(b), _stream(device)), "ns_get_rays")
    return (o, d, v, b) if want_batch else (o, d, v)


# ---- a2 -------------------------------------------------------------------------------------------
def sphere_intersect(o: Tensor, d: Tensor, radius: float) -> Tuple[Tensor, Tensor]:
    lib = _lib.load()
    o, d = _dev(o, "origin"), _dev(d, "direction")
    R = o.shape[0]
    t = torch.empty((R, 2), dtype=torch.float32, device=o.device)
    p = torch.empty((R, 2, 3), dtype=torch.float32, device=o.device)
    check(lib.ns_sphere_intersect(_ptr(o), _ptr(d), R, float(radius), _ptr(t), _ptr(p), _stream(o.device)),
          "ns_sphere_intersect")
    return t, p


def solve_quadratic(a: Tensor, b: Tensor, c: Tensor) -> Tensor:
    lib = _lib.load()
    a, b, c = _dev(a, "a"), _dev(b, "b"), _dev(c, "c")
    out = torch.empty((2,) + tuple(a.shape), dtype=torch.float32, device=a.device)
    check(lib.ns_solve_quadratic(_ptr(a), _ptr(b), _ptr(c), a.numel(), _ptr(out), _stream(a.device)),
          "ns_solve_quadratic")
    return out


# ---- a3 -------------------------------------------------------------------------------------------
def posenc(x: Tensor, n_freqs: int) -> Tensor:
    lib = _lib.load()
    x = _dev(x, "x")
    d = x.shape[-1]
    M = x.numel() // d
    out = torch.empty(tuple(x.shape[:-1]) + (d * (1 + 2 * n_freqs),), dtype=torch.float32, device=x.device)
    check(lib.ns_posenc(_ptr(x), M, d, n_freqs, _ptr(out), _stream(x.device)), "ns_posenc")
    return out


# ---- packed networks ------------------------------------------------------------------------------
class PackedWeights:
    """Owner of an ns_weights handle (device weight stream)."""

    def __init__(self, handle: int, kind: str, dtype: str, device):
        self.handle = C.c_void_p(handle)
        self.kind, self.dtype, self.device = kind, dtype, device

    @property
    def stream_bytes(self) -> int:
        return int(_lib.load().ns_weights_stream_bytes(self.handle))

    def __del__(self):
        try:
            if self.handle:
                _lib.load().ns_weights_destroy(self.handle)
                self.handle = None
        except Exception:  # interpreter shutdown
            pass


def _host_ptr_array(tensors: Sequence[Tensor]):
    keep = [np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32)) for t in tensors]
    arr = (C.c_void_p * len(keep))(*[k.ctypes.data for k in keep])
    return arr, keep


def pack_nerf(weights: Sequence[Tensor], biases: Sequence[Tensor], D: int, W: int, skip,
              dtype: Optional[str] = None, device="cuda", use_viewdirs: bool = True, output_ch: int = 4) -> PackedWeights:
    """weights/biases in the order pts_linears.0..D-1, then feature_linear, alpha_linear, views_linears.0, rgb_linear
    (use_viewdirs) or output_linear (use_viewdirs=False: raw has ``output_ch`` channels).  ``skip``: the reference's
    ``skips`` list (or one index / -1): layer i + 1 sees cat[x, h] for every i in it."""
    lib = _lib.load()
    skips = [skip] if isinstance(skip, int) else list(skip)
    skips = [int(i) for i in skips if int(i) >= 0]
    if any(i >= 32 for i in skips):
        raise NotImplementedError("skip indices beyond 31 are not supported")
    mask = 0
    for i in skips:
        mask |= 1 << i
    wa, k1 = _host_ptr_array(weights)
    ba, k2 = _host_ptr_array(biases)
    out = C.c_void_p()
    name = dtype or _compute_dtype
    with torch.cuda.device(device):
        check(lib.ns_pack_nerf_ex(D, W, mask, int(bool(use_viewdirs)), int(output_ch), wa, ba, dtype_code(name),
                                  C.byref(out)), "ns_pack_nerf_ex")
    pw = PackedWeights(out.value, "nerf", name, torch.device(device))
    pw.out_ch = int(lib.ns_nerf_out_channels(pw.handle))
    pw.use_viewdirs = bool(use_viewdirs)
    return pw


def pack_depthnet(weights: Sequence[Tensor], biases: Sequence[Tensor], hidden_sizes: Sequence[int],
                  cat_hidden_sizes: Sequence[int], dtype: Optional[str] = None, device="cuda") -> PackedWeights:
    """order: origin_layers.*, direction_layers.*, intersection_layers.*, cat_layers.{0,2,..}, to_depth.0

    ``hidden_sizes``: the widths of the three skip branches (any); ``cat_hidden_sizes``: the trunk widths (each <= 256).
    The affine branches are folded into the first trunk layer by the packer (ns_pack_depthnet_ex)."""
    lib = _lib.load()
    hs = (C.c_int * len(hidden_sizes))(*[int(v) for v in hidden_sizes])
    cs = (C.c_int * len(cat_hidden_sizes))(*[int(v) for v in cat_hidden_sizes])
    if len(weights) != 3 * len(hidden_sizes) + len(cat_hidden_sizes) + 1 or len(biases) != len(weights):
        raise ValueError("pack_depthnet: expected 3 * len(hidden_sizes) + len(cat_hidden_sizes) + 1 weight tensors")
    wa, k1 = _host_ptr_array(weights)
    ba, k2 = _host_ptr_array(biases)
    out = C.c_void_p()
    name = dtype or _compute_dtype
    with torch.cuda.device(device):
        check(lib.ns_pack_depthnet_ex(len(hidden_sizes), hs, len(cat_hidden_sizes), cs, wa, ba, dtype_code(name),
                                      C.byref(out)), "ns_pack_depthnet_ex")
    return PackedWeights(out.value, "depthnet", name, torch.device(device))


# ---- a4 -------------------------------------------------------------------------------------------
def depthnet_forward(net: PackedWeights, o: Tensor, d: Tensor, near: float = 2.0, far: float = 6.0,
                     sphere_radius: float = 2.0) -> Tensor:
    lib = _lib.load()
    o, d = _dev(o, "rays_o"), _dev(d, "rays_d")
    R = o.shape[0]
    z = torch.empty((R, 1), dtype=torch.float32, device=o.device)
    check(lib.ns_depthnet_forward(net.handle, _ptr(o), _ptr(d), R, float(near), float(far), float(sphere_radius),
                                  _ptr(z), _stream(o.device)), "ns_depthnet_forward")
    return z


# ---- a5 -------------------------------------------------------------------------------------------
def place_samples(o: Tensor, d: Tensor, mean: Tensor, n_samples: int, mode: str, std: float,
                  noise: Optional[Tensor] = None, want_pts: bool = True):
    lib = _lib.load()
    if mode not in _MODES:
        raise ValueError(f"unknown sampling mode {mode!r}")
    o, d, mean = _dev(o, "rays_o"), _dev(d, "rays_d"), _dev(mean, "mean")
    R = o.shape[0]
    N = 1 if mode == "depth_only" else int(n_samples)
    if mode == "gaussian" and noise is None:
        # same draw shape / order as the reference's torch.randn(mean.shape[0], n_samples - 1)
        noise = torch.randn(R, N - 1, device=o.device)
    if noise is not None:
        noise = _dev(noise, "noise")
    z = torch.empty((R, N), dtype=torch.float32, device=o.device)
    pts = torch.empty((R, N, 3), dtype=torch.float32, device=o.device) if want_pts else None
    check(lib.ns_place_samples(_MODES[mode], _ptr(o), _ptr(d), _ptr(mean), _ptr(noise), R, N, float(std),
                               _ptr(pts), _ptr(z), _stream(o.device)), "ns_place_samples")
    return pts, z


def points_along_rays(o: Tensor, d: Tensor, z: Tensor) -> Tensor:
    lib = _lib.load()
    o, d, z = _dev(o, "rays_o"), _dev(d, "rays_d"), _dev(z, "z")
    R, N = z.shape
    pts = torch.empty((R, N, 3), dtype=torch.float32, device=o.device)
    check(lib.ns_points_along_rays(_ptr(o), _ptr(d), _ptr(z), R, N, _ptr(pts), _stream(o.device)),
          "ns_points_along_rays")
    return pts


# ---- a6 / a7 ----------------------------------------------------------------------------------------
def nerf_forward(net: PackedWeights, pts: Tensor, viewdirs: Optional[Tensor]) -> Tensor:
    """pts [R,N,3], viewdirs [R,3] (None for a network without view directions) -> raw [R,N,C], C = net.out_ch"""
    lib = _lib.load()
    pts = _dev(pts, "pts")
    if getattr(net, "use_viewdirs", True):
        viewdirs = _dev(viewdirs, "viewdirs")
    else:
        viewdirs = None
    R, N = pts.shape[0], pts.shape[1]
    raw = torch.empty((R, N, getattr(net, "out_ch", 4)), dtype=torch.float32, device=pts.device)
    check(lib.ns_nerf_forward(net.handle, _ptr(pts), None, None, None, _ptr(viewdirs), R, N, _ptr(raw),
                              _stream(pts.device)), "ns_nerf_forward")
    return raw


def nerf_forward_rays(net: PackedWeights, o: Tensor, d: Tensor, z: Tensor, viewdirs: Optional[Tensor]) -> Tensor:
    """points formed in-kernel as o + d*z; z [R,N] -> raw [R,N,C]"""
    lib = _lib.load()
    o, d, z = _dev(o, "rays_o"), _dev(d, "rays_d"), _dev(z, "z")
    viewdirs = _dev(viewdirs, "viewdirs") if getattr(net, "use_viewdirs", True) else None
    R, N = z.shape
    raw = torch.empty((R, N, getattr(net, "out_ch", 4)), dtype=torch.float32, device=z.device)
    check(lib.ns_nerf_forward(net.handle, None, _ptr(o), _ptr(d), _ptr(z), _ptr(viewdirs), R, N, _ptr(raw),
                              _stream(z.device)), "ns_nerf_forward")
    return raw


def nerf_forward_embedded(net: PackedWeights, x: Tensor) -> Tensor:
    lib = _lib.load()
    x = _dev(x, "x")
    width = 90 if getattr(net, "use_viewdirs", True) else 63
    if x.shape[-1] != width:
        raise NotImplementedError(f"embedded input must be {width} wide (63 point + 27 direction features), got {x.shape[-1]}")
    M = x.numel() // width
    raw = torch.empty(tuple(x.shape[:-1]) + (getattr(net, "out_ch", 4),), dtype=torch.float32, device=x.device)
    check(lib.ns_nerf_forward_embedded(net.handle, _ptr(x), M, _ptr(raw), _stream(x.device)),
          "ns_nerf_forward_embedded")
    return raw


# ---- a8 -------------------------------------------------------------------------------------------
def raw2outputs(raw: Tensor, z: Tensor, rays_d: Tensor, noise: Optional[Tensor] = None,
                white_bkgd: bool = True, want_per_sample: bool = True):
    """-> rgb [R,3], disp [R], acc [R], depth [R], alphas [R,N] | None, weights [R,N] | None"""
    lib = _lib.load()
    if raw.shape[-1] > 4:        # output_ch = 5 networks (no view directions, N_importance > 0): channel 4 is never read
        raw = raw[..., :4]
    raw, z, rays_d = _dev(raw, "raw"), _dev(z, "z_vals"), _dev(rays_d, "rays_d")
    R, N = z.shape
    dev = raw.device
    rgb = torch.empty((R, 3), dtype=torch.float32, device=dev)
    disp = torch.empty((R,), dtype=torch.float32, device=dev)
    acc = torch.empty_like(disp)
    depth = torch.empty_like(disp)
    # one sample: the reference's alphas / weights are [R, 0] (see ns_raw2outputs in the header)
    n_out = 0 if N == 1 else N
    alphas = torch.empty((R, n_out), dtype=torch.float32, device=dev) if want_per_sample else None
    weights = torch.empty((R, n_out), dtype=torch.float32, device=dev) if want_per_sample else None
    if noise is not None:
        noise = _dev(noise, "noise")
    check(lib.ns_raw2outputs(_ptr(raw), _ptr(z), _ptr(rays_d), _ptr(noise), R, N, int(bool(white_bkgd)),
                             _ptr(rgb), _ptr(disp), _ptr(acc), _ptr(depth), _ptr(alphas), _ptr(weights),
                             _stream(dev)), "ns_raw2outputs")
    return rgb, disp, acc, depth, alphas, weights


# ---- a11 ------------------------------------------------------------------------------------------
def coarse_z(near: Tensor, far: Tensor, n_samples: int, lindisp: bool, t_rand: Optional[Tensor] = None) -> Tensor:
    lib = _lib.load()
    near, far = _dev(near.reshape(-1), "near"), _dev(far.reshape(-1), "far")
    R = near.shape[0]
    if t_rand is not None:
        t_rand = _dev(t_rand, "t_rand")
    z = torch.empty((R, n_samples), dtype=torch.float32, device=near.device)
    check(lib.ns_coarse_z(_ptr(near), _ptr(far), R, n_samples, int(bool(lindisp)), _ptr(t_rand), _ptr(z),
                          _stream(near.device)), "ns_coarse_z")
    return z


def sample_pdf(bins: Tensor, weights: Tensor, n_samples: int, u: Optional[Tensor] = None) -> Tensor:
    lib = _lib.load()
    bins, weights = _dev(bins, "bins"), _dev(weights, "weights")
    R, Nb = bins.shape
    if weights.shape[-1] != Nb - 1:
        raise ValueError("weights must have one element fewer than bins")
    if u is not None:
        u = _dev(u, "u")
    out = torch.empty((R, n_samples), dtype=torch.float32, device=bins.device)
    check(lib.ns_sample_pdf(_ptr(bins), _ptr(weights), R, Nb, n_samples, _ptr(u), _ptr(out), _stream(bins.device)),
          "ns_sample_pdf")
    return out


def importance_z(z: Tensor, weights: Tensor, n_importance: int, u: Optional[Tensor] = None) -> Tensor:
    lib = _lib.load()
    z, weights = _dev(z, "z_vals"), _dev(weights, "weights")
    R, Nc = z.shape
    if u is not None:
        u = _dev(u, "u")
    out = torch.empty((R, Nc + n_importance), dtype=torch.float32, device=z.device)
    check(lib.ns_importance_z(_ptr(z), _ptr(weights), R, Nc, n_importance, _ptr(u), _ptr(out), _stream(z.device)),
          "ns_importance_z")
    return out


def sort_rows(x: Tensor) -> Tensor:
    lib = _lib.load()
    x = _dev(x, "x")
    R, N = x.shape
    out = torch.empty_like(x)
    check(lib.ns_sort_rows(_ptr(x), R, N, _ptr(out), _stream(x.device)), "ns_sort_rows")
    return out


def argmax_gather(weights: Tensor, z: Tensor, raw: Optional[Tensor] = None):
    lib = _lib.load()
    weights, z = _dev(weights, "weights"), _dev(z, "z_vals")
    R, N = weights.shape
    dev = weights.device
    max_z = torch.empty((R, 1), dtype=torch.float32, device=dev)
    max_w = torch.empty((R, 1), dtype=torch.float32, device=dev)
    max_rgb = None
    if raw is not None:
        raw = _dev(raw, "raw")
        max_rgb = torch.empty((R, 3), dtype=torch.float32, device=dev)
    check(lib.ns_argmax_gather(_ptr(weights), _ptr(z), _ptr(raw), R, N, _ptr(max_z), _ptr(max_w), _ptr(max_rgb),
                               _stream(dev)), "ns_argmax_gather")
    return max_z, max_w, max_rgb


# ---- a9 fused -------------------------------------------------------------------------------------
class RenderWorkspace:
    """Reusable device workspace for render_rays_depthnet (grown on demand, never shrunk)."""

    def __init__(self):
        self.buf: Optional[Tensor] = None

    def get(self, nbytes: int, device) -> Tensor:
        # + 256: callers align the base pointer up to 256 bytes
        if self.buf is None or self.buf.numel() < nbytes + 256 or self.buf.device != torch.device(device):
            self.buf = torch.empty((nbytes + 256,), dtype=torch.uint8, device=device)
        return self.buf


_default_ws = RenderWorkspace()


def _rgb_disp_outputs(a, R: int, device, shard: Optional[Tensor]):
    """Per-ray outputs of the one-call renderers: fresh packed tensors, or views of an interleaved [.., 4] shard."""
    if shard is None:
        out = {"rgb": torch.empty((R, 3), dtype=torch.float32, device=device),
               "disp": torch.empty((R,), dtype=torch.float32, device=device)}
        a.rgb_dev, a.disp_dev = out["rgb"].data_ptr(), out["disp"].data_ptr()
        return out
    if not (shard.is_cuda and shard.dtype == torch.float32 and shard.dim() == 2 and shard.shape[1] == 4
            and shard.is_contiguous() and shard.shape[0] >= R):
        raise ValueError(f"shard must be a contiguous fp32 device tensor [>= {R}, 4], got {tuple(shard.shape)} {shard.dtype}")
    a.rgb_dev, a.disp_dev = shard.data_ptr(), shard.data_ptr() + 12
    a.rgb_stride = a.disp_stride = 4
    return {"rgb": shard[:R, :3], "disp": shard[:R, 3]}


def render_rays_depthnet(depthnet: PackedWeights, nerf: PackedWeights, *, rays=None, camera=None,
                         n_samples: int, mode: str, std: float, noise: Optional[Tensor] = None,
                         near: float = 2.0, far: float = 6.0, sphere_radius: float = 2.0,
                         white_bkgd: bool = True, extras: bool = False, workspace: Optional[RenderWorkspace] = None,
                         device="cuda", mlp_events=None, shard: Optional[Tensor] = None,
                         one_kernel: Optional[bool] = None, guard: Optional[PackedWeights] = None,
                         guard_threshold: Optional[float] = None):
    """DepthNet -> placement -> NeRF MLP -> compositing as one C call.

    rays = (o, d, viewdirs) device tensors, or camera = (H, W, K, c2w, row0, row1) to generate
    the rays on the device.  Returns dict(rgb [R,3], disp [R], and with extras z/weights/pts).
    ``shard``: a contiguous fp32 [>= R, 4] device tensor; the compositing kernel then writes (r, g, b, disp) of ray i
    straight into shard[i] (the unit parallel.FrameRenderer all-gathers) and rgb / disp are returned as views of it.
    ``one_kernel``: None (default) = ns_render_rays_fused -- placement, MLP and compositing in ONE persistent kernel, per-sample
    data never in HBM -- whenever the configuration supports it (uniform placement, bf16 / f16 field, n_samples a power of
    two <= 64 or a multiple of 64 up to 512), else the five-launch chain ns_render_rays_depthnet; True = require it; False = the chain.  Both produce the
    same bits.
    ``guard``: the SAME radiance field packed "f16x3" (fp32-grade).  The last sample of every ray -- the one the reference
    composites with dist = 1e10, so that alpha = step(sigma) -- is then evaluated a second time through it and its sigma
    replaces the 16-bit one (R of the R * N samples; uniform placement only).  Pair it with an "f16x3" DepthNet handle:
    the two together are the PSNR guard of the 16-bit paths (see psnr_guard_handles).
    ``guard_threshold`` (one-kernel renderer, n_samples <= 64): None = the module setting (set_psnr_guard, 16.0); > 0 = only the
    rays whose own sigma of the last sample lies within it of zero are re-evaluated, after the kernel, on a compacted list
    (the same bits as the every-ray guard wherever |sigma16 - sigma32| stays below it); 0 = every ray, before the kernel.
    """
    lib = _lib.load()
    a = _lib.RenderArgs()
    a.depthnet, a.nerf = depthnet.handle, nerf.handle
    keep = []
    if rays is not None and rays[0].shape[0] == 0:     # empty batch: nothing to launch
        dev0, n0 = rays[0].device, (1 if mode == "depth_only" else int(n_samples))
        out = {"rgb": torch.empty((0, 3), device=dev0), "disp": torch.empty((0,), device=dev0)}
        if extras:
            out.update(z=torch.empty((0, n0), device=dev0), weights=torch.empty((0, 0 if n0 == 1 else n0), device=dev0),
                       pts=torch.empty((0, n0, 3), device=dev0))
        return out
    if rays is not None:
        o, d, v = (_dev(t, n) for t, n in zip(rays, ("rays_o", "rays_d", "viewdirs")))
        keep += [o, d, v]
        device = o.device
        R = o.shape[0]
        a.o_dev, a.d_dev, a.viewdirs_dev, a.R = o.data_ptr(), d.data_ptr(), v.data_ptr(), R
    else:
        H, W, K, c2w, row0, row1 = camera
        c2w_h = (c2w.detach().cpu().numpy() if isinstance(c2w, Tensor) else np.asarray(c2w)).astype(np.float32)[:3, :4]
        a.H, a.W, a.row0, a.row1 = H, W, row0, row1
        a.fx, a.fy, a.cx, a.cy = float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2])
        for i, val in enumerate(c2w_h.reshape(-1)):
            a.c2w[i] = float(val)
        R = (row1 - row0) * W
        a.R = R
    device = torch.device(device)
    N = 1 if mode == "depth_only" else int(n_samples)
    a.mode, a.N, a.std_ = _MODES[mode], N, float(std)
    if mode == "gaussian":
        if noise is None:
            noise = torch.randn(R, N - 1, device=device)
        noise = _dev(noise, "noise")
        keep.append(noise)
        a.noise_dev = noise.data_ptr()
    a.near_, a.far_, a.sphere_radius, a.white_bkgd = float(near), float(far), float(sphere_radius), int(bool(white_bkgd))
    fused_ok = bool(lib.ns_render_fused_supported(nerf.handle, a.mode, N)) and noise is None
    if one_kernel and not fused_ok:
        raise NotImplementedError(f"the one-kernel renderer needs uniform placement, a bf16 / f16 field with view directions and "
                                  f"n_samples a power of two in [2, 64] or a multiple of 64 up to 512 (mode {mode!r}, n_samples {N}, dtype {getattr(nerf, 'dtype', '?')})")
    use_fused = fused_ok if one_kernel is None else bool(one_kernel)
    nbytes = int(lib.ns_render_fused_workspace_bytes(R) if use_fused else lib.ns_render_workspace_bytes(R, N))
    ws = (workspace or _default_ws).get(nbytes, device)
    base = ws.data_ptr()
    a.workspace_dev = (base + 255) & ~255
    out = _rgb_disp_outputs(a, R, device, shard)
    if extras:
        out["z"] = torch.empty((R, N), dtype=torch.float32, device=device)
        # one sample: the reference's weights are [R, 0] (its dists are empty) and ns_raw2outputs writes none
        out["weights"] = torch.empty((R, 0 if N == 1 else N), dtype=torch.float32, device=device)
        out["pts"] = torch.empty((R, N, 3), dtype=torch.float32, device=device)
        a.z_dev, a.pts_dev = out["z"].data_ptr(), out["pts"].data_ptr()
        a.weights_dev = out["weights"].data_ptr() if N > 1 else None
    if mlp_events is not None:
        a.ev_mlp_begin, a.ev_mlp_end = mlp_events[0].handle, mlp_events[1].handle
    if guard is not None:
        a.nerf_guard = guard.handle
        a.guard_threshold = float(_guard_threshold if guard_threshold is None else guard_threshold)
    if use_fused:
        check(lib.ns_render_rays_fused(C.byref(a), _stream(device)), "ns_render_rays_fused")
    else:
        check(lib.ns_render_rays_depthnet(C.byref(a), _stream(device)), "ns_render_rays_depthnet")
    return out


def render_rays_hierarchical(coarse: PackedWeights, fine: Optional[PackedWeights], *, rays=None, camera=None,
                             n_coarse: int = 64, n_importance: int = 128, lindisp: bool = True,
                             white_bkgd: bool = True, near: float = 2.0, far: float = 6.0,
                             t_rand: Optional[Tensor] = None, u: Optional[Tensor] = None, extras: bool = False,
                             workspace: Optional[RenderWorkspace] = None, device="cuda", mlp_events=None,
                             shard: Optional[Tensor] = None, coarse_events=None):
    """Vanilla coarse + fine pass (sample_as_in_NeRF) as one C call; returns the FINE pass outputs.
    ``shard``: as in render_rays_depthnet.  ``mlp_events`` / ``coarse_events``: (begin, end) ops.Event pairs recorded around
    the fine-pass / coarse-pass MLP kernel."""
    lib = _lib.load()
    a = _lib.HierArgs()
    a.coarse = coarse.handle
    a.fine = fine.handle if fine is not None else None
    keep = []
    if rays is not None:
        o, d, v = (_dev(t, n) for t, n in zip(rays, ("rays_o", "rays_d", "viewdirs")))
        keep += [o, d, v]
        device = o.device
        R = o.shape[0]
        a.o_dev, a.d_dev, a.viewdirs_dev, a.R = o.data_ptr(), d.data_ptr(), v.data_ptr(), R
    else:
        H, W, K, c2w, row0, row1 = camera
        c2w_h = (c2w.detach().cpu().numpy() if isinstance(c2w, Tensor) else np.asarray(c2w)).astype(np.float32)[:3, :4]
        a.H, a.W, a.row0, a.row1 = H, W, row0, row1
        a.fx, a.fy, a.cx, a.cy = float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2])
        for i, val in enumerate(c2w_h.reshape(-1)):
            a.c2w[i] = float(val)
        R = (row1 - row0) * W
        a.R = R
    device = torch.device(device)
    a.Nc, a.Nf, a.lindisp, a.white_bkgd = int(n_coarse), int(n_importance), int(bool(lindisp)), int(bool(white_bkgd))
    a.near_, a.far_ = float(near), float(far)
    for name, t in (("t_rand_dev", t_rand), ("u_dev", u)):
        if t is not None:
            t = _dev(t, name)
            keep.append(t)
            setattr(a, name, t.data_ptr())
    nbytes = int(lib.ns_hier_workspace_bytes(R, a.Nc, a.Nf))
    ws = (workspace or _default_ws).get(nbytes, device)
    a.workspace_dev = (ws.data_ptr() + 255) & ~255
    Nt = a.Nc + a.Nf
    out = _rgb_disp_outputs(a, R, device, shard)
    if extras:
        out["z"] = torch.empty((R, Nt), dtype=torch.float32, device=device)
        out["weights"] = torch.empty((R, Nt), dtype=torch.float32, device=device)
        out["raw"] = torch.empty((R, Nt, 4), dtype=torch.float32, device=device)
        a.z_dev, a.weights_dev, a.raw_dev = out["z"].data_ptr(), out["weights"].data_ptr(), out["raw"].data_ptr()
    if mlp_events is not None:
        a.ev_mlp_begin, a.ev_mlp_end = mlp_events[0].handle, mlp_events[1].handle
    if coarse_events is not None:
        a.ev_coarse_begin, a.ev_coarse_end = coarse_events[0].handle, coarse_events[1].handle
    check(lib.ns_render_rays_hierarchical(C.byref(a), _stream(device)), "ns_render_rays_hierarchical")
    return out


class debug_switch:
    """Context manager around ns_debug_set: ``with ops.debug_switch(generic_kernels=1, prod_tiles=4): ...`` (tests compare the
    kernel variants inside one process); every switch is reset to 0 (= the dispatcher's own choice) on exit."""

    def __init__(self, **switches):
        self.switches = switches

    def __enter__(self):
        for k, v in self.switches.items():
            check(_lib.load().ns_debug_set(k.encode(), int(v)), "ns_debug_set")
        return self

    def __exit__(self, *exc):
        for k in self.switches:
            check(_lib.load().ns_debug_set(k.encode(), 0), "ns_debug_set")
        return False


class Event:
    """hipEvent wrapper for timing a kernel on the stream it is launched on."""

    def __init__(self):
        self.handle = C.c_void_p()
        check(_lib.load().ns_event_create(C.byref(self.handle)), "ns_event_create")

    def record(self, device="cuda"):
        check(_lib.load().ns_event_record(self.handle, _stream(torch.device(device))), "ns_event_record")

    def elapsed_ms(self, end: "Event") -> float:
        ms = C.c_float()
        check(_lib.load().ns_event_elapsed_ms(self.handle, end.handle, C.byref(ms)), "ns_event_elapsed_ms")
        return float(ms.value)

    def __del__(self):
        try:
            if self.handle:
                _lib.load().ns_event_destroy(self.handle)
                self.handle = None
        except Exception:
            pass
