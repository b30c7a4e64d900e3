"""CPU oracle for the ray-batch render hot path of MarcinKadziolka/nerf-sampling.

TEST INFRASTRUCTURE ONLY.  This module is a plain-PyTorch (CPU, fp32) restatement of
the reference's arithmetic for the hot path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and
only as the checker / the timed CPU port -- never as a product path.  The product
(``nerf_sampling_amd``) fails loudly if its HIP library is missing; it never falls
back to this file.

Pinning: every function below is checked in ``tests/test_oracle_golden.py`` against
golden vectors captured from the reference itself (imported in the build container
by ``tools/make_golden.py``; fixtures under ``tests/golden/``), and the ray-sphere /
quadratic functions additionally against the reference's own known-answer tests
(``nerf_sampling/tests/tests.py:197-331``).

All ``file:line`` citations are relative to the reference tree (``/root/reference``).
Weights are plain ``dict[str, Tensor]`` keyed exactly like the reference state
dicts (``run_nerf_helpers.py:87-105``, ``depth_net.py:103-107``).
"""

from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch

Tensor = torch.Tensor
Params = Dict[str, Tensor]


# --------------------------------------------------------------------------------------
# a1  ray generation                      run_nerf_helpers.py:187-202, nerf_utils.py:156-188
# --------------------------------------------------------------------------------------
def camera_rays(H: int, W: int, K, c2w: Tensor) -> Tuple[Tensor, Tensor]:
    """Pinhole rays for every pixel, row-major (row j, col i).

    run_nerf_helpers.py:187-202.  ``K`` is the float64 numpy intrinsics matrix built at
    Trainer.py:142; its entries enter the fp32 pixel arithmetic as python scalars.
    """
    cols = torch.linspace(0, W - 1, W)
    rows = torch.linspace(0, H - 1, H)
    jj, ii = torch.meshgrid(rows, cols, indexing="ij")  # jj: row index, ii: col index
    cam = torch.stack(
        [(ii - K[0][2]) / K[0][0], -(jj - K[1][2]) / K[1][1], -torch.ones_like(ii)], -1
    )
    rot = c2w[:3, :3]
    rays_d = (cam[..., None, :] * rot).sum(-1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def ray_batch_from_camera(
    H: int, W: int, K, c2w: Tensor, near: float, far: float, use_viewdirs: bool = True,
    c2w_staticcam: Optional[Tensor] = None,
):
    """[R, 11] = [o, d, near, far, unit viewdir] as nerf_utils.py:156-188 (ndc=False).  With ``c2w_staticcam`` the view
    directions are those of ``c2w`` and the rays those of the static camera (:172-176)."""
    rays_o, rays_d = camera_rays(H, W, K, c2w)
    shape = rays_d.shape
    view = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
    view = view.reshape(-1, 3).float()
    if use_viewdirs and c2w_staticcam is not None:
        rays_o, rays_d = camera_rays(H, W, K, c2w_staticcam)
    o = rays_o.reshape(-1, 3).float()
    d = rays_d.reshape(-1, 3).float()
    nr = near * torch.ones_like(d[..., :1])
    fr = far * torch.ones_like(d[..., :1])
    batch = torch.cat([o, d, nr, fr], -1)
    if use_viewdirs:
        batch = torch.cat([batch, view], -1)
    return batch, o, d, shape


from nerf_sampling_amd.synthetic import blender_intrinsics, pose_spherical, render_poses  # noqa: E402,F401


# --------------------------------------------------------------------------------------
# a2  quadratic / ray-sphere intersection                         utils.py:159-217
# --------------------------------------------------------------------------------------
def solve_quadratic(a: Tensor, b: Tensor, c: Tensor) -> Tensor:
    """Roots of a x^2 + b x + c, minus-sqrt root first; NaN if none (utils.py:159-179)."""
    disc = b**2 - 4 * a * c
    sign = torch.stack([torch.ones_like(disc), -torch.ones_like(disc)])
    root = torch.sqrt(disc)
    return (-b - sign * root) / (2 * a)


def sphere_intersections(o: Tensor, d: Tensor, radius: Tensor) -> Tuple[Tensor, Tensor]:
    """t [R,2] and points [R,2,3] where o + t d meets the origin-centred sphere.

    utils.py:182-217.  ``c`` is formed as norm(o)**2 (sqrt, then square) minus r**2.
    """
    b = 2 * (d * o).sum(dim=1)
    c = torch.norm(o, dim=1) ** 2 - radius.reshape(-1) ** 2  # radius is a 1-element tensor
    a = (d * d).sum(dim=1)
    t = solve_quadratic(a, b, c).T
    pts = o.unsqueeze(1) + t.unsqueeze(2) * d.unsqueeze(1)
    return t, pts


# --------------------------------------------------------------------------------------
# a3  positional encoding                                  run_nerf_helpers.py:15-63
# --------------------------------------------------------------------------------------
def posenc(x: Tensor, n_freqs: int) -> Tensor:
    """[x, sin(2^0 x), cos(2^0 x), ..., sin(2^{L-1} x), cos(2^{L-1} x)], each block d wide."""
    bands = 2.0 ** torch.linspace(0.0, n_freqs - 1, steps=n_freqs)
    parts = [x]
    for f in bands:
        parts.append(torch.sin(x * f))
        parts.append(torch.cos(x * f))
    return torch.cat(parts, -1)


def posenc_dim(d: int, n_freqs: int) -> int:
    return d * (1 + 2 * n_freqs)


# --------------------------------------------------------------------------------------
# a4  DepthNet                                                   depth_net.py:10-169
# --------------------------------------------------------------------------------------
def _lin(p: Params, name: str, x: Tensor) -> Tensor:
    return torch.nn.functional.linear(x, p[name + ".weight"], p[name + ".bias"])


def depthnet_layer_counts(p: Params) -> Tuple[int, int]:
    n_branch = len({k.split(".")[1] for k in p if k.startswith("origin_layers.")})
    n_trunk = len({k.split(".")[1] for k in p if k.startswith("cat_layers.")})
    return n_branch, n_trunk


def depthnet_forward(
    p: Params,
    o: Tensor,
    d: Tensor,
    multires: int = 10,
    sphere_radius: float = 2.0,
    near: float = 2,
    far: float = 6,
    return_parts: bool = False,
):
    """Ray (o, d) -> one depth in [near, far], shape [R, 1]  (depth_net.py:117-169).

    The three skip branches are *affine*: the reference constructs ``nn.LeakyReLU(h)``
    without applying it (depth_net.py:140,148,156).  Skip concat order is hidden first,
    embedding second (depth_net.py:139).
    """
    n_branch, n_trunk = depthnet_layer_counts(p)
    e_o = posenc(o, multires)
    e_d = posenc(d, multires)
    _, pts = sphere_intersections(o, d, torch.tensor([sphere_radius]))
    e_x = posenc(torch.flatten(pts, start_dim=1), multires)

    def branch(prefix: str, e: Tensor) -> Tensor:
        h = e
        for i in range(n_branch):
            h = _lin(p, f"{prefix}.{i}", torch.cat([h, e], -1))
        return h

    h_o = branch("origin_layers", e_o)
    h_d = branch("direction_layers", e_d)
    h_x = branch("intersection_layers", e_x)
    y = torch.cat([h_o, h_d, h_x, e_o, e_d, e_x], -1)
    for i in range(n_trunk):
        y = torch.nn.functional.leaky_relu(_lin(p, f"cat_layers.{2 * i}", y), 0.01)
    depth = torch.sigmoid(_lin(p, "to_depth.0", y))
    z = near * (1 - depth) + far * depth
    if return_parts:
        return z, {"e_o": e_o, "e_d": e_d, "e_x": e_x, "h_o": h_o, "h_d": h_d, "h_x": h_x}
    return z


# --------------------------------------------------------------------------------------
# a5  sample placement around the predicted depth                 utils.py:220-244
# --------------------------------------------------------------------------------------
def place_samples(
    o: Tensor,
    d: Tensor,
    mean: Tensor,
    n_samples: int = 32,
    mode: str = "gaussian",
    std: float = 0.1,
    noise: Optional[Tensor] = None,
) -> Tuple[Tensor, Tensor]:
    """pts [R,N,3], z [R,N].  ``noise`` ([R, n-1] standard normal) replaces torch.randn."""
    if mode == "depth_only":
        z = mean
    elif mode == "gaussian":
        if noise is None:
            noise = torch.randn(mean.shape[0], n_samples - 1)
        z, _ = torch.cat([mean + std * noise, mean], dim=-1).sort(dim=-1)
    elif mode == "uniform":
        grid = torch.linspace(-std, std, steps=n_samples - 1)
        z, _ = torch.cat([mean + grid.view(1, -1).expand(mean.size(0), -1), mean], -1).sort(-1)
        z = torch.clip(z, 2, 6)  # hard-coded 2/6, utils.py:240
    else:
        raise ValueError(mode)
    return o[..., None, :] + d[..., None, :] * z[..., :, None], z


# --------------------------------------------------------------------------------------
# a6/a7  run_network + NeRF MLP      Trainer.py:789-806, run_nerf_helpers.py:67-134
# --------------------------------------------------------------------------------------
def nerf_depth(p: Params) -> int:
    return len({k.split(".")[1] for k in p if k.startswith("pts_linears.")})


def nerf_forward(p: Params, x: Tensor, input_ch: int = 63, skips=(4,)) -> Tensor:
    """[M, input_ch + input_ch_views] -> [M, 4] raw (rgb pre-sigmoid, sigma pre-relu); a state dict with an
    ``output_linear`` (use_viewdirs=False, run_nerf_helpers.py:132-133) -> [M, output_ch]."""
    pts, views = x[..., :input_ch], x[..., input_ch:]
    h = pts
    for i in range(nerf_depth(p)):
        h = torch.relu(_lin(p, f"pts_linears.{i}", h))
        if i in skips:
            h = torch.cat([pts, h], -1)  # input first, hidden second (:118)
    if "output_linear.weight" in p:
        return _lin(p, "output_linear", h)
    sigma = _lin(p, "alpha_linear", h)
    feat = _lin(p, "feature_linear", h)  # no relu (:121)
    h = torch.relu(_lin(p, "views_linears.0", torch.cat([feat, views], -1)))
    rgb = _lin(p, "rgb_linear", h)
    return torch.cat([rgb, sigma], -1)


def run_network(
    p: Params,
    pts: Tensor,
    viewdirs: Optional[Tensor],
    multires: int = 10,
    multires_views: int = 4,
    netchunk: int = 1024 * 64,
    skips=(4,),
) -> Tensor:
    """Embed + MLP in netchunk-row slices; [R,N,3] -> [R,N,4] (Trainer.py:789-806)."""
    flat = pts.reshape(-1, pts.shape[-1])
    emb = posenc(flat, multires)
    if viewdirs is not None:
        dirs = viewdirs[:, None].expand(pts.shape).reshape(-1, pts.shape[-1])
        emb = torch.cat([emb, posenc(dirs, multires_views)], -1)
    in_ch = posenc_dim(pts.shape[-1], multires)
    outs = [
        nerf_forward(p, emb[i : i + netchunk], in_ch, skips) for i in range(0, emb.shape[0], netchunk)
    ]
    out = torch.cat(outs, 0)
    return out.reshape(list(pts.shape[:-1]) + [out.shape[-1]])


# --------------------------------------------------------------------------------------
# a8  alpha compositing              nerf_utils.py:27-42, sampling_trainer.py:153-230
# --------------------------------------------------------------------------------------
def raw2outputs(
    raw: Tensor,
    z: Tensor,
    rays_d: Tensor,
    raw_noise_std: float = 0.0,
    white_bkgd: bool = True,
    noise: Optional[Tensor] = None,
):
    """7-tuple (rgb_map, disp_map, acc_map, depth_map, density, alphas, weights).

    ``noise`` ([R,N], already scaled by nothing -- it is multiplied by raw_noise_std
    here) stands in for torch.randn (sampling_trainer.py:188-193).
    """
    dists = z[..., 1:] - z[..., :-1]
    dists = torch.cat([dists, torch.tensor([1e10]).expand(dists[..., :1].shape)], -1)
    dists = dists * torch.norm(rays_d[..., None, :], dim=-1)
    rgb = torch.sigmoid(raw[..., :3])
    add = 0.0
    if raw_noise_std > 0.0:
        if noise is None:
            noise = torch.randn(raw[..., 3].shape)
        add = noise * raw_noise_std
    density = raw[..., 3]
    alphas = 1.0 - torch.exp(-torch.relu(density + add) * dists)
    trans = torch.cumprod(
        torch.cat([torch.ones((alphas.shape[0], 1)), 1.0 - alphas + 1e-10], -1), -1
    )[:, :-1]
    weights = alphas * trans
    rgb_map = torch.sum(weights[..., None] * rgb, -2)
    depth_map = torch.sum(weights * z, -1)
    disp_map = 1.0 / torch.max(
        1e-10 * torch.ones_like(depth_map), depth_map / (torch.sum(weights, -1) + 1e-10)
    )
    acc_map = torch.sum(weights, -1)
    if white_bkgd:
        rgb_map = rgb_map + (1.0 - acc_map[..., None])
    if weights.shape[-1] == 0:
        rgb_map = torch.sum(rgb, -2)
    return rgb_map, disp_map, acc_map, depth_map, density, alphas, weights


# --------------------------------------------------------------------------------------
# a11  vanilla hierarchical path   Trainer.py:553-710, run_nerf_helpers.py:250-293
# --------------------------------------------------------------------------------------
def coarse_z_vals(
    near: Tensor, far: Tensor, n_rays: int, n_samples: int, lindisp: bool,
    perturb: float = 0.0, t_rand: Optional[Tensor] = None,
) -> Tensor:
    """Stratified coarse depths (Trainer.py:603-626).  near/far are [R,1]."""
    t = torch.linspace(0.0, 1.0, steps=n_samples)
    if not lindisp:
        z = near * (1.0 - t) + far * t
    else:
        z = 1.0 / (1.0 / near * (1.0 - t) + 1.0 / far * t)
    z = z.expand([n_rays, n_samples])
    if perturb > 0.0:
        mids = 0.5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat([mids, z[..., -1:]], -1)
        lower = torch.cat([z[..., :1], mids], -1)
        if t_rand is None:
            t_rand = torch.rand(z.shape)
        z = lower + (upper - lower) * t_rand
    return z


def sample_pdf(bins: Tensor, weights: Tensor, n_samples: int, det: bool = True,
               u: Optional[Tensor] = None) -> Tensor:
    """Inverse-CDF sampling (run_nerf_helpers.py:250-293). ``u`` overrides the draws."""
    w = weights + 1e-5
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if u is None:
        if det:
            u = torch.linspace(0.0, 1.0, steps=n_samples)
            u = u.expand(list(cdf.shape[:-1]) + [n_samples])
        else:
            u = torch.rand(list(cdf.shape[:-1]) + [n_samples])
    u = u.contiguous()
    idx = torch.searchsorted(cdf, u, right=True)
    lo = torch.clamp(idx - 1, min=0)
    hi = torch.clamp(idx, max=cdf.shape[-1] - 1)
    cdf_lo, cdf_hi = torch.gather(cdf, -1, lo), torch.gather(cdf, -1, hi)
    bin_lo, bin_hi = torch.gather(bins, -1, lo), torch.gather(bins, -1, hi)
    denom = cdf_hi - cdf_lo
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    return bin_lo + (u - cdf_lo) / denom * (bin_hi - bin_lo)


def hierarchical_render(
    ray_batch: Tensor, p_coarse: Params, p_fine: Optional[Params], n_samples: int,
    n_importance: int, lindisp: bool, white_bkgd: bool, perturb: float = 0.0,
    raw_noise_std: float = 0.0, netchunk: int = 1024 * 64,
    t_rand: Optional[Tensor] = None, u: Optional[Tensor] = None,
):
    """sample_as_in_NeRF: 8-tuple (density, z, pts, rgb_map, weights, alphas, disp, raw).

    nerf_utils.py:497-611 -> Trainer.sample_coarse_points :579-649 (white_bkgd and the
    noise std are passed positionally and therefore honoured here) ->
    Trainer.sample_fine_points :651-710.
    """
    n_rays = ray_batch.shape[0]
    o, d = ray_batch[:, 0:3], ray_batch[:, 3:6]
    view = ray_batch[:, -3:] if ray_batch.shape[-1] > 8 else None
    bounds = ray_batch[..., 6:8].reshape(-1, 1, 2)
    near, far = bounds[..., 0], bounds[..., 1]
    z = coarse_z_vals(near, far, n_rays, n_samples, lindisp, perturb, t_rand)
    pts = o[..., None, :] + d[..., None, :] * z[..., :, None]
    raw = run_network(p_coarse, pts, view, netchunk=netchunk)
    _, _, _, _, _, _, w = raw2outputs(raw, z, d, raw_noise_std, white_bkgd)
    z_mid = 0.5 * (z[..., 1:] + z[..., :-1])
    z_new = sample_pdf(z_mid, w[..., 1:-1], n_importance, det=(perturb == 0.0), u=u).detach()
    z_all, _ = torch.sort(torch.cat([z, z_new], -1), -1)
    pts = o[..., None, :] + d[..., None, :] * z_all[..., :, None]
    raw = run_network(p_fine if p_fine is not None else p_coarse, pts, view, netchunk=netchunk)
    rgb_map, disp, _acc, _depth, density, alphas, weights = raw2outputs(
        raw, z_all, d, raw_noise_std, white_bkgd
    )
    return density, z_all, pts, rgb_map, weights, alphas, disp, raw


# --------------------------------------------------------------------------------------
# a9/a10  ray-batch operators                                  nerf_utils.py:614-876
# --------------------------------------------------------------------------------------
def render_rays_test(
    ray_batch: Tensor, p_coarse: Params, p_fine: Optional[Params], p_depth: Optional[Params],
    n_depth_samples: int, sampling_mode: str, distance: float,
    N_samples: int = 64, N_importance: int = 128, lindisp: bool = True,
    white_bkgd: bool = False, compare_nerf: bool = False, use_nerf_max_pts: bool = False,
    use_full_nerf: bool = False, sphere_radius: float = 2.0, netchunk: int = 1024 * 64,
    noise: Optional[Tensor] = None,
) -> Dict[str, Tensor]:
    """Inference operator (nerf_utils.py:736-876).

    On the DepthNet branch the reference passes ``raw_noise=`` / ``white_bkdg=`` (sic) to
    raw2outputs (:862-863), so the noise std is 0 and the background is white regardless
    of the caller's flags; that is reproduced here.
    """
    o, d = ray_batch[:, 0:3], ray_batch[:, 3:6]
    view = ray_batch[:, -3:] if ray_batch.shape[-1] > 8 else None
    ret: Dict[str, Tensor] = {}
    if compare_nerf or use_nerf_max_pts or use_full_nerf:
        _dens, f_z, f_pts, f_rgb, f_w, _al, f_disp, f_raw = hierarchical_render(
            ray_batch, p_coarse, p_fine, N_samples, N_importance, lindisp, white_bkgd,
            netchunk=netchunk,
        )
        top = f_w.argmax(dim=1, keepdim=True)
        max_z = torch.gather(f_z, 1, top)
        max_w = torch.gather(f_w, 1, top)
        rgb = torch.sigmoid(f_raw[..., :3])
        max_rgb = torch.gather(rgb, 1, top.unsqueeze(-1).expand(-1, 1, 3)).squeeze()
        max_pts = o[..., None, :] + d[..., None, :] * max_z[..., :, None]
        ret["max_z_vals"], ret["max_pts"], ret["max_weights"] = max_z, max_pts, max_w
    if use_nerf_max_pts:
        rgb_map, disp, w, pts, z = max_rgb, torch.zeros_like(max_rgb), max_w, max_pts, max_z
    elif use_full_nerf:
        rgb_map, disp, w, pts, z = f_rgb, f_disp, f_w, f_pts, f_z
    else:
        mean = depthnet_forward(p_depth, o, d, sphere_radius=sphere_radius)
        pts, z = place_samples(o, d, mean, n_depth_samples, sampling_mode, distance, noise)
        raw = run_network(p_fine if p_fine is not None else p_coarse, pts, view, netchunk=netchunk)
        rgb_map, disp, _a, _dm, _de, _al, w = raw2outputs(raw, z, d, 0.0, True)
    ret["depth_net_rgb_map"] = rgb_map
    ret["depth_net_weights"] = w
    ret["depth_net_disp_map"] = disp
    ret["depth_net_z_vals"] = z
    ret["depth_net_pts"] = pts
    return ret


def render_rays(
    ray_batch: Tensor, p_coarse: Params, p_fine: Optional[Params], p_depth: Params,
    N_samples: int = 64, N_importance: int = 128, lindisp: bool = True,
    white_bkgd: bool = False, perturb: float = 0.0, raw_noise_std: float = 0.0,
    sphere_radius: float = 2.0, netchunk: int = 1024 * 64,
    t_rand: Optional[Tensor] = None, u: Optional[Tensor] = None,
) -> Dict[str, Tensor]:
    """Training operator, forward only (nerf_utils.py:614-733)."""
    o, d = ray_batch[:, 0:3], ray_batch[:, 3:6]
    view = ray_batch[:, -3:] if ray_batch.shape[-1] > 8 else None
    _dens, f_z, _pts, _rgb, f_w, _al, _disp, _raw = hierarchical_render(
        ray_batch, p_coarse, p_fine, N_samples, N_importance, lindisp, white_bkgd,
        perturb, raw_noise_std, netchunk, t_rand, u,
    )
    top = f_w.argmax(dim=1, keepdim=True)
    max_z = torch.gather(f_z, 1, top)
    max_pts = o[..., None, :] + d[..., None, :] * max_z[..., :, None]
    z = depthnet_forward(p_depth, o, d, sphere_radius=sphere_radius)
    pts = o[..., None, :] + d[..., None, :] * z[..., :, None]
    raw = run_network(p_fine if p_fine is not None else p_coarse, pts, view, netchunk=netchunk)
    rgb_map, disp, _a, _dm, _de, _al, _w = raw2outputs(raw, z, d, 0.0, True)
    return {
        "depth_net_rgb_map": rgb_map, "depth_net_disp_map": disp, "depth_net_z_vals": z,
        "max_z_vals": max_z, "depth_net_pts": pts, "max_pts": max_pts, "raw": raw,
    }


def render_frame(
    H: int, W: int, K, c2w: Tensor, chunk: int, near: float, far: float, **kw
) -> Tuple[Tensor, Tensor, Dict[str, Tensor]]:
    """render_test: whole frame in ``chunk``-ray slices (nerf_utils.py:191-255, :73-85).

    Per-chunk host copies of weights/disp/z/pts (:867-870) are no-ops on CPU tensors.
    """
    batch, o, d, shape = ray_batch_from_camera(H, W, K, c2w, near, far)
    parts: Dict[str, list] = {}
    for i in range(0, batch.shape[0], chunk):
        r = render_rays_test(batch[i : i + chunk], **kw)
        for k, v in r.items():
            parts.setdefault(k, []).append(v)
    out = {k: torch.cat(v, 0) for k, v in parts.items()}
    out = {k: v.reshape(list(shape[:-1]) + list(v.shape[1:])) for k, v in out.items()}
    rgb = out.pop("depth_net_rgb_map")
    disp = out.pop("depth_net_disp_map")
    out["rays_o"], out["rays_d"] = o, d
    return rgb, disp, out


# --------------------------------------------------------------------------------------
# synthetic seeded scenes live in nerf_sampling_amd/synthetic.py (pure data generators shared by
# tools/make_golden.py, the tests and bench.py); re-exported here for the tests' convenience
# --------------------------------------------------------------------------------------
from nerf_sampling_amd.synthetic import (  # noqa: E402,F401
    SCENES, make_depthnet_params, make_nerf_params, make_scene,
)
