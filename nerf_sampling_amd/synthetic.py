"""Deterministic synthetic scenes: seeded network weights and the reference's render cameras.

No dataset or checkpoint ships with the reference, so tests and bench.py
use these generators on both sides of every comparison.  Pure numpy/torch-CPU data generation;
nothing here touches the HIP library or the oracle.
"""

from __future__ import annotations

import math
import os
from typing import Dict

import numpy as np
import torch

Tensor = torch.Tensor
Params = Dict[str, Tensor]


def posenc_dim(d: int, n_freqs: int) -> int:
    return d * (1 + 2 * n_freqs)


def pose_spherical(theta_deg: float, phi_deg: float, radius: float) -> Tensor:
    """Camera-to-world [4,4] of the reference's render path (load_blender.py:10-43)."""
    th = theta_deg / 180.0 * np.pi
    ph = phi_deg / 180.0 * np.pi
    t = torch.tensor([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius], [0, 0, 0, 1]]).float()
    rp = torch.tensor([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0], [0, np.sin(ph), np.cos(ph), 0],
                       [0, 0, 0, 1]]).float()
    rt = torch.tensor([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0],
                       [0, 0, 0, 1]]).float()
    flip = torch.tensor(np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]]), dtype=torch.float32)
    return flip @ (rt @ (rp @ t))


def render_poses(n: int = 40) -> Tensor:
    """The reference's 40 spiral render poses, theta = -180 + 9k (load_blender.py:84-90)."""
    return torch.stack([pose_spherical(a, -30.0, 4.0) for a in np.linspace(-180, 180, n + 1)[:-1]], 0)


def blender_intrinsics(H: int, W: int, camera_angle_x: float = 0.6911112070083618):
    """focal and K as load_blender.py:81-82 and Trainer.py:141-142 (float64 numpy)."""
    focal = 0.5 * W / np.tan(0.5 * camera_angle_x)
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    return focal, K


def _uniform_linear(rng: np.random.Generator, out_f: int, in_f: int, gain: float = 1.0):
    bound = gain / math.sqrt(in_f)
    w = rng.uniform(-bound, bound, size=(out_f, in_f)).astype(np.float32)
    b = rng.uniform(-bound, bound, size=(out_f,)).astype(np.float32)
    return torch.from_numpy(w), torch.from_numpy(b)



def _decay_embedding_columns(w: Tensor, start: int, d: int, n_freqs: int) -> None:
    """Scale the weight columns that multiply sin/cos(2^L x) by 2^-L (in place).

    Trained radiance fields are smooth at the scale of fp32 rounding; an i.i.d. random first layer is
    not (it weights the 2^9 band like the 2^0 band, so a 1-ulp change of a coordinate is amplified
    ~500x and end-to-end comparisons measure conditioning, not kernels).  The 1/f decay gives the
    synthetic scenes a realistic spectrum.  Column order: run_nerf_helpers.py:44-45.
    """
    for level in range(n_freqs):
        lo = start + d + 2 * d * level
        w[:, lo : lo + 2 * d] *= 2.0 ** (-level)


def make_nerf_params(
    seed: int, D: int = 8, W: int = 256, input_ch: int = 63, input_ch_views: int = 27,
    skips=(4,), sigma_gain: float = 1.0, sigma_bias: float = 0.0, hidden_gain: float = 1.0,
    spectral_decay: bool = False, use_viewdirs: bool = True, output_ch: int = 4,
) -> Params:
    """Deterministic NeRF weights, nn.Linear-default-like U(+-1/sqrt(fan_in)) scale.

    Key names follow run_nerf_helpers.py:87-105.  ``sigma_gain``/``sigma_bias`` rescale the
    density head so synthetic scenes have a non-trivial opacity distribution;
    ``hidden_gain`` > 1 keeps activations from shrinking through the trunk.
    """
    rng = np.random.default_rng(seed)
    p: Params = {}
    for i in range(D):
        in_f = input_ch if i == 0 else (W + input_ch if (i - 1) in skips else W)
        w, b = _uniform_linear(rng, W, in_f, hidden_gain)
        if spectral_decay and in_f != W:  # layers that see the embedded point (first 63 columns)
            _decay_embedding_columns(w, 0, 3, (input_ch // 3 - 1) // 2)
        p[f"pts_linears.{i}.weight"], p[f"pts_linears.{i}.bias"] = w, b
    w, b = _uniform_linear(rng, W // 2, input_ch_views + W, hidden_gain)   # views_linears exists in both variants (:94)
    if spectral_decay:
        _decay_embedding_columns(w, W, 3, (input_ch_views // 3 - 1) // 2)
    p["views_linears.0.weight"], p["views_linears.0.bias"] = w, b
    if not use_viewdirs:       # run_nerf_helpers.py:104-105
        w, b = _uniform_linear(rng, output_ch, W, hidden_gain)
        p["output_linear.weight"], p["output_linear.bias"] = w, b
        return p
    w, b = _uniform_linear(rng, W, W, hidden_gain)
    p["feature_linear.weight"], p["feature_linear.bias"] = w, b
    w, b = _uniform_linear(rng, 1, W)
    p["alpha_linear.weight"], p["alpha_linear.bias"] = w * sigma_gain, b * sigma_gain + sigma_bias
    w, b = _uniform_linear(rng, 3, W // 2, hidden_gain)
    p["rgb_linear.weight"], p["rgb_linear.bias"] = w, b
    return p


def make_depthnet_params(
    seed: int, n_layers: int = 10, width: int = 256, multires: int = 10,
    branch_gain: float = 1.0, trunk_gain: float = 1.0, spectral_decay: bool = False,
) -> Params:
    """Deterministic DepthNet weights; key names follow depth_net.py:103-107.

    ``branch_gain`` = sqrt(3) keeps the variance of the (affine) skip branches constant
    with depth, ``trunk_gain`` = sqrt(6) does the same for the LeakyReLU trunk, so the
    predicted depth actually varies across an image instead of collapsing to a constant.
    """
    rng = np.random.default_rng(seed)
    e3, e6 = posenc_dim(3, multires), posenc_dim(6, multires)
    p: Params = {}
    for prefix, e in (("origin_layers", e3), ("direction_layers", e3), ("intersection_layers", e6)):
        for i in range(n_layers):
            in_f = 2 * e if i == 0 else width + e
            w, b = _uniform_linear(rng, width, in_f, branch_gain)
            if spectral_decay:
                dch = 3 if e == e3 else 6
                if i == 0:
                    _decay_embedding_columns(w, 0, dch, multires)
                    _decay_embedding_columns(w, e, dch, multires)
                else:
                    _decay_embedding_columns(w, width, dch, multires)
            p[f"{prefix}.{i}.weight"], p[f"{prefix}.{i}.bias"] = w, b
    for i in range(n_layers):
        in_f = 3 * width + 2 * e3 + e6 if i == 0 else width
        w, b = _uniform_linear(rng, width, in_f, trunk_gain)
        if spectral_decay and i == 0:
            _decay_embedding_columns(w, 3 * width, 3, multires)
            _decay_embedding_columns(w, 3 * width + e3, 3, multires)
            _decay_embedding_columns(w, 3 * width + 2 * e3, 6, multires)
        p[f"cat_layers.{2 * i}.weight"], p[f"cat_layers.{2 * i}.bias"] = w, b
    w, b = _uniform_linear(rng, 1, width, trunk_gain)
    p["to_depth.0.weight"], p["to_depth.0.bias"] = w, b
    return p


def make_depthnet_params_shaped(seed: int, hidden_sizes, cat_hidden_sizes, multires: int = 10,
                                branch_gain: float = 1.0, trunk_gain: float = 1.0) -> Params:
    """make_depthnet_params for arbitrary branch widths (``hidden_sizes``) and trunk widths (``cat_hidden_sizes``), the
    shapes depth_net.py:46-101 builds: branch layer i is [hidden[i], (hidden[i-1] | e) + e], trunk layer 0 is
    [cat[0], 3 hidden[-1] + 2 e3 + e6].  Embedding columns of level L are scaled by 2^-L as in the scenes."""
    rng = np.random.default_rng(seed)
    e3, e6 = posenc_dim(3, multires), posenc_dim(6, multires)
    hs, cs = list(hidden_sizes), list(cat_hidden_sizes)
    p: Params = {}
    for prefix, e in (("origin_layers", e3), ("direction_layers", e3), ("intersection_layers", e6)):
        for i, out_f in enumerate(hs):
            prev = e if i == 0 else hs[i - 1]
            w, b = _uniform_linear(rng, out_f, prev + e, branch_gain)
            dch = 3 if e == e3 else 6
            if i == 0:
                _decay_embedding_columns(w, 0, dch, multires)
            _decay_embedding_columns(w, prev, dch, multires)
            p[f"{prefix}.{i}.weight"], p[f"{prefix}.{i}.bias"] = w, b
    for i, out_f in enumerate(cs):
        in_f = 3 * hs[-1] + 2 * e3 + e6 if i == 0 else cs[i - 1]
        w, b = _uniform_linear(rng, out_f, in_f, trunk_gain)
        if i == 0:
            _decay_embedding_columns(w, 3 * hs[-1], 3, multires)
            _decay_embedding_columns(w, 3 * hs[-1] + e3, 3, multires)
            _decay_embedding_columns(w, 3 * hs[-1] + 2 * e3, 6, multires)
        p[f"cat_layers.{2 * i}.weight"], p[f"cat_layers.{2 * i}.bias"] = w, b
    w, b = _uniform_linear(rng, 1, cs[-1], trunk_gain)
    p["to_depth.0.weight"], p["to_depth.0.bias"] = w, b
    return p


SQRT3, SQRT6 = math.sqrt(3.0), math.sqrt(6.0)

# NeRF constructor variants beyond the production one (run_nerf_helpers.py:67-105), pinned against the reference by
# tests/golden/nerf_variants.npz: tag -> make_nerf_params kwargs (+ the constructor's own arguments)
NERF_VARIANTS = {
    "two_skips": dict(seed=81, D=6, W=128, skips=(1, 3), use_viewdirs=True, hidden_gain=SQRT6, spectral_decay=True),
    "skip_first_and_late": dict(seed=82, D=7, W=256, skips=(0, 2, 5), use_viewdirs=True, hidden_gain=SQRT6, spectral_decay=True),
    "no_viewdirs_5ch": dict(seed=83, D=5, W=128, skips=(2,), use_viewdirs=False, output_ch=5, input_ch_views=0,
                            hidden_gain=SQRT6, spectral_decay=True),
    "no_viewdirs_4ch": dict(seed=84, D=3, W=256, skips=(), use_viewdirs=False, output_ch=4, input_ch_views=0,
                            hidden_gain=SQRT6, spectral_decay=True),
}

# NeRF widths other than the two the kernels are instantiated for (netwidth / netwidth_fine of the reference's configs,
# nerf_utils.py:409-423): the packer zero-pads them to 128 / 256.  Pinned against the reference by tests/golden/nerf_widths.npz.
NERF_WIDTHS = {
    "w64": dict(seed=91, D=4, W=64, skips=(1,), use_viewdirs=True, hidden_gain=SQRT6, spectral_decay=True),
    "w200": dict(seed=92, D=8, W=200, skips=(4,), use_viewdirs=True, hidden_gain=SQRT6, spectral_decay=True),
    "w97_odd": dict(seed=93, D=3, W=97, skips=(), use_viewdirs=True, hidden_gain=SQRT6, spectral_decay=True),
    "w40_no_viewdirs": dict(seed=94, D=4, W=40, skips=(2,), use_viewdirs=False, output_ch=5, input_ch_views=0,
                            hidden_gain=SQRT6, spectral_decay=True),
}

# DepthNet shapes other than one uniform width, pinned against the reference by tests/golden/depthnet_shapes.npz:
# tag -> (hidden_sizes, cat_hidden_sizes, seed).  "default" is the reference's class default (depth_net.py:13-16).
DEPTHNET_SHAPES = {
    "default": ([128] * 6, [128, 128, 128, 128, 256], 71),
    "ragged": ([48, 80], [64, 16, 200, 256], 72),
    "one": ([32], [96], 73),
}

# Canonical synthetic "scenes": seeds and density-head calibration chosen once so that a
# frame has a mix of opaque / transparent rays (probe: tools/make_golden.py --stats).
SCENES = {
    # production sizes (run.py:101-109: n_layers 10, layer_width 256; NeRF 8x256)
    "lego_synth": dict(
        coarse=dict(seed=12, D=8, W=256, hidden_gain=SQRT6, sigma_gain=300.0, sigma_bias=45.0, spectral_decay=True),
        fine=dict(seed=13, D=8, W=256, hidden_gain=SQRT6, sigma_gain=450.0, sigma_bias=115.0, spectral_decay=True),
        depth=dict(seed=7, n_layers=10, width=256, branch_gain=SQRT3, trunk_gain=SQRT6, spectral_decay=True),
    ),
    # reduced sizes for fast CPU tests
    "tiny_synth": dict(
        coarse=dict(seed=22, D=4, W=128, hidden_gain=SQRT6, sigma_gain=300.0, sigma_bias=-30.0, spectral_decay=True),
        fine=dict(seed=23, D=4, W=128, hidden_gain=SQRT6, sigma_gain=300.0, sigma_bias=43.0, spectral_decay=True),
        depth=dict(seed=27, n_layers=3, width=128, branch_gain=SQRT3, trunk_gain=SQRT6, spectral_decay=True),
    ),
}


FITTED_SCENE_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                "fitted_scene")

# "shapes_fit": production-size networks FITTED (tools/fit_scene.py, on an MI355X) to the analytic ground-truth scene of
# nerf_sampling_amd/analytic_scene.py: one NeRF 8x256 used as both network_fn and network_fine, and the 10x256 DepthNet
# trained against it with this repo's own training step.  The weights are committed fixtures (safetensors, fp32).
SCENES["shapes_fit"] = dict(coarse=dict(D=8, W=256), fine=dict(D=8, W=256), depth=dict(n_layers=10, width=256),
                            files=("nerf.safetensors", "depthnet.safetensors"))


def load_fitted_scene(directory: str = None) -> Dict[str, Params]:
    from safetensors.torch import load_file

    directory = directory or FITTED_SCENE_DIR
    nerf_file, depth_file = SCENES["shapes_fit"]["files"]
    for f in (nerf_file, depth_file):
        if not os.path.exists(os.path.join(directory, f)):
            raise FileNotFoundError(f"{os.path.join(directory, f)}: the fitted scene's weights are missing "
                                    "(tools/fit_scene.py writes them)")
    nerf = load_file(os.path.join(directory, nerf_file))
    return {"coarse": nerf, "fine": nerf, "depth": load_file(os.path.join(directory, depth_file))}


def make_scene(name: str = "lego_synth") -> Dict[str, Params]:
    cfg = SCENES[name]
    if "files" in cfg:
        return load_fitted_scene()
    return {
        "coarse": make_nerf_params(**cfg["coarse"]),
        "fine": make_nerf_params(**cfg["fine"]),
        "depth": make_depthnet_params(**cfg["depth"]),
    }
