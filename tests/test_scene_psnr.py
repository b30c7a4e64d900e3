"""The PSNR half of BASELINE.json's metric on a FITTED scene.

north_star: "PSNR within 0.05 dB of reference".  The reference's deliverable is a PSNR against ground truth per image
(nerf_utils.py:306-336); no dataset or checkpoint ships with it, so the ground truth here is the analytic scene of
nerf_sampling_amd/analytic_scene.py (exact ray cast) and the networks are the committed fixtures
tests/golden/fitted_scene/*.safetensors, fitted to that scene on an MI355X by tools/fit_scene.py (NeRF 8x256 by torch
autograd; DepthNet 10x256 by this repo's own HIP training step).  The reference arithmetic is the fp32 CPU oracle.
"""

import os

import numpy as np
import pytest
import torch

from nerf_sampling_amd import analytic_scene, synthetic
from oracle import nerf_oracle as O

H = W = 800
POSE = 7            # of the reference's 40 spiral render poses
ROWS = (400, 460)   # 60 rows (48 000 rays) through the spheres and the slab


def _psnr(a, b):
    mse = float(((a - b) ** 2).mean())
    return float("inf") if mse == 0 else -10 * np.log10(mse)


# ---- CPU: the ground truth and the fixture -----------------------------------------------------------------------
def test_analytic_scene_ray_cast():
    """Known answers of the exact ray caster: the red sphere from straight above, a miss, the slab's top face."""
    o = torch.tensor([[0.0, 0.0, 4.0], [3.0, 3.0, 4.0], [1.0, -1.0, 4.0]])
    d = torch.tensor([[0.0, 0.0, -2.0], [0.0, 0.0, -1.0], [0.0, 0.0, -1.0]])       # un-normalised, like rays_d
    rgb, t, hit = analytic_scene.raycast(o, d)
    assert hit.tolist() == [True, False, True]
    assert abs(float(t[0]) - (4.0 - 0.05 - 0.62) / 2.0) < 1e-6          # top of the sphere at z = 0.05 + 0.62, |d| = 2
    assert abs(float(t[2]) - (4.0 - (-0.78 + 0.11))) < 1e-6             # top face of the slab
    assert torch.equal(rgb[1], torch.ones(3)) and float(rgb[0, 0]) > float(rgb[0, 2])   # white miss, red sphere
    _, K = synthetic.blender_intrinsics(64, 64)
    rgb, t, hit = analytic_scene.frame(64, 64, K, synthetic.render_poses(40)[POSE][:3, :4])
    assert rgb.shape == (64, 64, 3) and 0.3 < float(hit.float().mean()) < 0.7
    assert float(rgb.min()) >= 0.0 and float(rgb.max()) <= 1.0 and torch.isinf(t[~hit]).all()
    assert 2.0 < float(t[hit].min()) and float(t[hit].max()) < 6.0      # inside the cameras' [near, far]


def test_fitted_scene_fixture_is_a_trained_field():
    """The committed weights load into the reference's state-dict layout and behave like a trained field on a sparse
    sample of the frame: the oracle's DepthNet render resembles the ground truth, empty space is robustly empty
    (sigma well below zero along background rays), and solid space is opaque."""
    p = synthetic.make_scene("shapes_fit")
    assert len(p["fine"]) == 24 and len(p["depth"]) == 82 and p["fine"]["pts_linears.5.weight"].shape == (256, 319)
    _, K = synthetic.blender_intrinsics(H, W)
    c2w = synthetic.render_poses(40)[POSE][:3, :4]
    batch, o, d, _ = O.ray_batch_from_camera(H, W, K, c2w, 2.0, 6.0)
    idx = torch.randperm(H * W, generator=torch.Generator().manual_seed(11))[:600]      # 600 rays spread over the frame
    gt, t, hit = analytic_scene.raycast(o[idx], d[idx])
    with torch.no_grad():
        out = O.render_rays_test(batch[idx], p["coarse"], p["fine"], p["depth"], 64, "uniform", 0.1)
        # the field along rays that hit nothing: 64 stratified samples over [near, far]
        miss = idx[~hit]
        z = torch.linspace(2.0, 6.0, 64).expand(miss.shape[0], 64)
        pts = o[miss][:, None] + d[miss][:, None] * z[..., None]
        sigma_empty = O.run_network(p["fine"], pts, batch[miss][:, -3:])[..., 3]
    assert _psnr(out["depth_net_rgb_map"], gt) > 17.0                    # a recognisable render (whole-band figure: GPU test)
    assert float((sigma_empty < 0).float().mean()) > 0.97 and float(sigma_empty.median()) < -3.0


# ---- GPU: the acceptance bar ---------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def fitted_band():
    torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
    p = synthetic.make_scene("shapes_fit")
    _, K = synthetic.blender_intrinsics(H, W)
    c2w = synthetic.render_poses(40)[POSE][:3, :4]
    batch, o, d, _ = O.ray_batch_from_camera(H, W, K, c2w, 2.0, 6.0)
    sl = slice(ROWS[0] * W, ROWS[1] * W)
    with torch.no_grad():
        out = O.render_rays_test(batch[sl], p["coarse"], p["fine"], p["depth"], 64, "uniform", 0.1)
    gt = analytic_scene.frame(H, W, K, c2w, ROWS[0], ROWS[1])[0].reshape(-1, 3)
    return dict(K=K, c2w=c2w, rgb=out["depth_net_rgb_map"], gt=gt, psnr=_psnr(out["depth_net_rgb_map"], gt))


#                                     all-ray PSNR(build || oracle) floor: ~3 dB under the value measured on MI355X (round 3)
@pytest.mark.gpu
@pytest.mark.parametrize("dtype,floor_db", [("bf16", 31.0), ("f16", 42.0), ("f16x3", 100.0), ("f32", 100.0),   # measured 34.5, 45.3, 110.8, 108.1
                                            ("bf16 default pairing", 33.0)])
def test_scene_psnr_within_0p05_db_of_the_reference(gpu_modules, fitted_band, dtype, floor_db):
    """PSNR(oracle fp32 || ground truth) vs PSNR(build || ground truth) on 60 rows of an 800x800 frame, DepthNet + 64
    samples/ray (BASELINE configs[1]): |delta| <= 0.05 dB for every operand type, the headline bf16 included; plus the
    all-ray PSNR of the build against the oracle itself."""
    from nerf_sampling_amd import ops

    m = gpu_modules("shapes_fit")
    b = fitted_band
    if dtype == "bf16 default pairing":       # what ops.set_compute_dtype("bf16") gives: bf16 field, f16 DepthNet (ops.depthnet_dtype_for)
        ops.set_compute_dtype("bf16")
        dn, nf = m["depth"].packed(), m["fine"].packed()
        ops.set_compute_dtype("f32")
        assert dn.dtype == "f16" and nf.dtype == "bf16"
    else:
        dn, nf = m["depth"].packed(dtype), m["fine"].packed(dtype)
    out = ops.render_rays_depthnet(dn, nf,
                                   camera=(H, W, b["K"], b["c2w"], ROWS[0], ROWS[1]), n_samples=64, mode="uniform", std=0.1)
    rgb = out["rgb"].cpu()
    mine = _psnr(rgb, b["gt"])
    vs_oracle = _psnr(rgb, b["rgb"])
    err = (rgb - b["rgb"]).abs().max(-1).values
    print(f"scene PSNR [{dtype}]: oracle {b['psnr']:.4f} dB, build {mine:.4f} dB, delta {mine - b['psnr']:+.4f} dB; "
          f"build vs oracle over all rays {vs_oracle:.2f} dB, rays off by > 1e-2: {float((err > 1e-2).float().mean()):.5f}, "
          f"median |err| {float(err.median()):.2e}")
    assert b["psnr"] > 15.0                                 # the fitted scene renders its ground truth recognisably
    assert abs(mine - b["psnr"]) <= 0.05, (dtype, mine, b["psnr"])
    assert vs_oracle > floor_db, (dtype, vs_oracle)
