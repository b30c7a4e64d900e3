"""An analytic ground-truth scene: a few coloured spheres and boxes, ray-cast exactly.

No dataset ships with the reference, so the scene-PSNR half of BASELINE.json's metric needs a
ground truth that can be regenerated anywhere: this module IS the dataset.  ``raycast`` returns, for any rays of the
reference's cameras (origin + un-normalised direction, run_nerf_helpers.py:187-202), the exact pixel colour over a white
background (the Blender loader's ``white_bkgd`` compositing, trainers/Blender.py:27-30) and the hit parameter t with
``hit point = o + t d``.  ``tools/fit_scene.py`` fits the radiance field and the DepthNet of ``tests/golden/fitted_scene``
to it; ``tests/test_gpu_scene_psnr.py`` and ``bench.py`` score renders against it.

Pure torch data generation (CPU or GPU tensors); nothing here touches the HIP library or the oracle.
"""

from __future__ import annotations

import math
from typing import Tuple

import torch

Tensor = torch.Tensor

# (centre, radius, albedo) -- everything sits inside the DepthNet's intersection sphere (radius 2) and the cameras'
# [near, far] = [2, 6] range at camera distance 4
SPHERES = (
    ((0.00, 0.00, 0.05), 0.62, (0.85, 0.22, 0.18)),
    ((0.78, 0.35, -0.32), 0.36, (0.20, 0.62, 0.28)),
    ((-0.62, -0.48, -0.22), 0.44, (0.18, 0.32, 0.82)),
    ((-0.30, 0.72, -0.40), 0.27, (0.88, 0.74, 0.16)),
)
# (centre, half extents, albedo)
BOXES = (
    ((0.00, 0.00, -0.78), (1.15, 1.15, 0.11), (0.72, 0.70, 0.66)),     # the slab everything stands on
    ((0.55, -0.62, -0.42), (0.22, 0.22, 0.25), (0.70, 0.30, 0.72)),
)
LIGHT = (0.35, -0.45, 0.82)
AMBIENT, DIFFUSE = 0.32, 0.68
TEXTURE_FREQ, TEXTURE_DEPTH = 5.0, 0.22


def _shade(albedo: Tensor, p: Tensor, n: Tensor) -> Tensor:
    light = torch.tensor(LIGHT, dtype=p.dtype, device=p.device)
    light = light / light.norm()
    lambert = AMBIENT + DIFFUSE * (n * light).sum(-1, keepdim=True).clamp(min=0.0)
    tex = 1.0 - TEXTURE_DEPTH * 0.5 * (1.0 + torch.sin(TEXTURE_FREQ * p[..., 0:1]) * torch.sin(TEXTURE_FREQ * p[..., 1:2])
                                       * torch.sin(TEXTURE_FREQ * p[..., 2:3]))
    return (albedo * lambert * tex).clamp(0.0, 1.0)


def raycast(rays_o: Tensor, rays_d: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """(rgb [R,3] over white, t [R] (inf on a miss), hit [R] bool) for rays o + t d, t > 0; float64 inside."""
    o, d = rays_o.double(), rays_d.double()
    R = o.shape[0]
    best_t = torch.full((R,), math.inf, dtype=torch.float64, device=o.device)
    best_n = torch.zeros((R, 3), dtype=torch.float64, device=o.device)
    best_a = torch.ones((R, 3), dtype=torch.float64, device=o.device)
    dd = (d * d).sum(-1)
    for c, r, a in SPHERES:
        c_ = torch.tensor(c, dtype=torch.float64, device=o.device)
        oc = o - c_
        b = (oc * d).sum(-1)
        disc = b * b - dd * ((oc * oc).sum(-1) - r * r)
        t = (-b - torch.sqrt(disc.clamp(min=0.0))) / dd
        ok = (disc > 0) & (t > 0) & (t < best_t)
        n = (o + t[:, None] * d - c_) / r
        best_t = torch.where(ok, t, best_t)
        best_n = torch.where(ok[:, None], n, best_n)
        best_a = torch.where(ok[:, None], torch.tensor(a, dtype=torch.float64, device=o.device).expand(R, 3), best_a)
    for c, h, a in BOXES:
        c_ = torch.tensor(c, dtype=torch.float64, device=o.device)
        h_ = torch.tensor(h, dtype=torch.float64, device=o.device)
        inv = 1.0 / torch.where(d.abs() < 1e-12, torch.full_like(d, 1e-12), d)
        t0, t1 = (c_ - h_ - o) * inv, (c_ + h_ - o) * inv
        tmin, tmax = torch.minimum(t0, t1), torch.maximum(t0, t1)
        tn, axis = tmin.max(-1)
        tf = tmax.min(-1).values
        ok = (tn < tf) & (tn > 0) & (tn < best_t)
        n = torch.zeros((R, 3), dtype=torch.float64, device=o.device)
        n.scatter_(1, axis[:, None], -torch.sign(torch.gather(d, 1, axis[:, None])))
        best_t = torch.where(ok, tn, best_t)
        best_n = torch.where(ok[:, None], n, best_n)
        best_a = torch.where(ok[:, None], torch.tensor(a, dtype=torch.float64, device=o.device).expand(R, 3), best_a)
    hit = torch.isfinite(best_t)
    p = o + torch.where(hit, best_t, torch.zeros_like(best_t))[:, None] * d
    rgb = torch.where(hit[:, None], _shade(best_a, p, best_n), torch.ones_like(best_a))
    return rgb.float(), best_t.float(), hit


def sdf(p: Tensor) -> Tensor:
    """Signed distance [...,] of points p [..., 3] to the nearest object surface (negative inside)."""
    best = torch.full(p.shape[:-1], math.inf, dtype=p.dtype, device=p.device)
    for c, r, _ in SPHERES:
        c_ = torch.tensor(c, dtype=p.dtype, device=p.device)
        best = torch.minimum(best, (p - c_).norm(dim=-1) - r)
    for c, h, _ in BOXES:
        c_ = torch.tensor(c, dtype=p.dtype, device=p.device)
        h_ = torch.tensor(h, dtype=p.dtype, device=p.device)
        q = (p - c_).abs() - h_
        best = torch.minimum(best, q.clamp(min=0.0).norm(dim=-1) + q.max(-1).values.clamp(max=0.0))
    return best


def depth_target(rays_o: Tensor, rays_d: Tensor, near: float = 2.0, far: float = 6.0, coarse: int = 192, fine: int = 24):
    """(t_target [R], hit [R]): the depth a DepthNet should predict for each ray -- the exact hit parameter where the ray
    hits, and on a miss the parameter of the ray's CLOSEST APPROACH to any surface (argmin of the signed distance along the
    ray, two-level search).  At an object's silhouette against the background the two agree (a tangent ray's hit point is
    its closest approach), so the target is continuous there and only object-over-object occlusion edges keep a jump."""
    _, t_hit, hit = raycast(rays_o, rays_d)
    o, d = rays_o.double(), rays_d.double()
    R = o.shape[0]
    ts = torch.linspace(near, far, coarse, dtype=torch.float64, device=o.device)
    dist = sdf(o[:, None, :] + ts[None, :, None] * d[:, None, :])             # [R, coarse]
    k = dist.argmin(-1)
    step = (far - near) / (coarse - 1)
    t0 = (ts[k] - step).clamp(min=near)
    tf = t0[:, None] + torch.linspace(0.0, 2.0 * step, fine, dtype=torch.float64, device=o.device)[None, :]
    tf = tf.clamp(max=far)
    kf = sdf(o[:, None, :] + tf[..., None] * d[:, None, :]).argmin(-1)
    t_close = torch.gather(tf, 1, kf[:, None])[:, 0]
    return torch.where(hit, t_hit.double(), t_close).float(), hit


def frame(H: int, W: int, K, c2w: Tensor, row0: int = 0, row1: int = None, device="cpu") -> Tuple[Tensor, Tensor, Tensor]:
    """Ground-truth rows [row0, row1) of the H x W frame of camera ``c2w``: (rgb [rows, W, 3], t [rows, W], hit)."""
    row1 = H if row1 is None else row1
    c2w = torch.as_tensor(c2w, dtype=torch.float64, device=device)
    jj, ii = torch.meshgrid(torch.arange(row0, row1, dtype=torch.float64, device=device),
                            torch.arange(W, dtype=torch.float64, device=device), indexing="ij")
    cam = torch.stack([(ii - K[0][2]) / K[0][0], -(jj - K[1][2]) / K[1][1], -torch.ones_like(ii)], -1)
    d = (cam[..., None, :] * c2w[:3, :3]).sum(-1).reshape(-1, 3)
    o = c2w[:3, 3].expand(d.shape)
    rgb, t, hit = raycast(o, d)
    return rgb.reshape(row1 - row0, W, 3), t.reshape(row1 - row0, W), hit.reshape(row1 - row0, W)
