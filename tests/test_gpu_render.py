"""GPU parity tests, operator level: the mirrored reference API (render_rays_test / render_rays /
render_test / sample_as_in_NeRF) through the HIP kernels against golden dicts captured from the
reference, plus size-independent properties at the full BASELINE size (800x800, 64 samples/ray).

fp32 gate: |rgb - reference| <= 1e-4 (BASELINE.json north_star) on every well-conditioned ray.
A ray is ill-conditioned when the reference's own result moves by more than the gate under a
perturbation of sigma the size of fp32 GEMM rounding noise (the last sample's alpha is
1 - exp(-relu(sigma) * 1e10), a step function of sigma's sign); those rays are counted and must be rare.
"""

import os

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def make_trainer(**over):
    from nerf_sampling_amd.trainers import DepthNetTrainer

    kw = dict(dataset_type="blender", basedir="/tmp", expname="golden", no_batching=True, datadir="/nonexistent",
              half_res=True, white_bkgd=True, N_importance=128, N_samples=64, use_viewdirs=True,
              input_dims_embed=3, device="cuda")
    kw.update(over)
    return DepthNetTrainer(**kw)


def render_kwargs(trainer, m):
    from nerf_sampling_amd.run_nerf_helpers import get_embedder

    embed_fn, _ = get_embedder(trainer.multires, trainer.i_embed, 3)
    embeddirs_fn, _ = get_embedder(trainer.multires_views, trainer.i_embed, 3)
    query = lambda inputs, viewdirs, network_fn: trainer.run_network(  # noqa: E731
        inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, netchunk=trainer.netchunk)
    return dict(network_query_fn=query, perturb=0.0, N_importance=trainer.N_importance, network_fine=m["fine"],
                N_samples=trainer.N_samples, network_fn=m["coarse"], use_viewdirs=True,
                white_bkgd=trainer.white_bkgd, raw_noise_std=0.0, trainer=trainer, lindisp=trainer.lindisp,
                depth_network=m["depth"], model_mode="test")


def npy(x):
    return x.detach().cpu().numpy()


# (max |rgb err|, max relative |disp err|) on the well-conditioned rays, measured on MI355X (round 3, fp32 path against
# the reference goldens); fp32_gate holds each case within 3x of its entry, and everything within north_star's 1e-4
MEASURED_FP32 = {"modes_tiny_synth": (3.7e-6, 2.0e-7), "modes_lego_synth": (4.3e-6, 2.1e-7),
                 "setup_uniform2": (6.6e-7, 2.4e-7), "setup_uniform64": (1.8e-6, 2.4e-7), "setup_depth_only1": (7.8e-7, 1e-7),
                 "train_tiny_synth": (6.6e-7, 1e-7), "train_lego_synth": (4.8e-7, 1e-7)}
WEIGHTS_TOL = 2e-4      # per-sample weights of the DepthNet branch against the reference golden


def frac_bad(a, b, tol):
    err = np.abs(a - b)
    err = err.reshape(err.shape[0], -1).max(axis=1)
    return float(np.mean(~(err <= tol))), err


def fp32_gate(tag, p, rb, n, mode, dist, rgb, disp, exp_rgb, exp_disp, max_ill=0.01, measured=None):
    """north_star's float tolerance on the fp32-grade paths: |rgb - reference| <= 1e-4 and |disp - reference| <= 1e-4
    (relative above 1) on EVERY well-conditioned ray.  The only allowance is an explicit, oracle-derived one: a ray is
    ill-conditioned when the ORACLE's own rgb / disp moves by more than the gate under a sigma shift of 1e-5 max|sigma|
    (~5x the measured fp32 sigma error: the last sample's alpha = 1 - exp(-relu(sigma) 1e10) is a step function of
    sigma_last); those rays are counted and must be rare.  ``measured`` = (max rgb err, max disp err) seen on MI355X
    for this case: the test additionally holds the result within 3x of it (a regression guard far below 1e-4)."""
    o, d, view = rb[:, 0:3], rb[:, 3:6], rb[:, -3:]
    with torch.no_grad():
        mean = O.depthnet_forward(p["depth"], o, d)
        pts, z = O.place_samples(o, d, mean, n, mode, dist)
        raw = O.run_network(p["fine"], pts, view)
        base = O.raw2outputs(raw, z, d, 0.0, True)
        eps = 1e-5 * float(raw[..., 3].abs().max())
        ill = torch.zeros(raw.shape[0], dtype=torch.bool)
        for sgn in (-1.0, 1.0):
            pert = raw.clone()
            pert[..., 3] += sgn * eps
            got = O.raw2outputs(pert, z, d, 0.0, True)
            ill |= (got[0] - base[0]).abs().max(-1).values > 1e-4
            ill |= ((got[1] - base[1]).abs() / base[1].abs().clamp(min=1.0)) > 1e-4
    ill = ill.numpy()
    err_rgb = np.abs(rgb - exp_rgb).reshape(len(ill), -1).max(-1)
    err_disp = np.abs(disp - exp_disp).reshape(-1) / np.maximum(np.abs(exp_disp).reshape(-1), 1.0)
    ok = ~ill & np.isfinite(exp_rgb).reshape(len(ill), -1).all(-1)
    print(f"fp32 gate [{tag}]: ill-conditioned {ill.mean():.4f}; well-conditioned max rgb err {err_rgb[ok].max():.2e}, "
          f"max disp err {err_disp[ok].max():.2e}, median rgb err {np.median(err_rgb[ok]):.2e}")
    assert ill.mean() <= max_ill, (tag, float(ill.mean()))
    assert err_rgb[ok].max() <= 1e-4, (tag, float(err_rgb[ok].max()))
    assert err_disp[ok].max() <= 1e-4, (tag, float(err_disp[ok].max()))
    if measured is not None:
        assert err_rgb[ok].max() <= 3 * measured[0] and err_disp[ok].max() <= 3 * measured[1], (
            tag, float(err_rgb[ok].max()), float(err_disp[ok].max()), measured)


@pytest.fixture(autouse=True)
def _fp32():
    from nerf_sampling_amd import ops

    ops.set_compute_dtype("f32")
    yield
    ops.set_compute_dtype("f32")


@pytest.mark.parametrize("scene", ["tiny_synth", "lego_synth"])
@pytest.mark.parametrize("mode", ["depthnet", "full_nerf", "nerf_max", "compare"])
def test_render_rays_test_modes(golden, gpu_modules, scene, mode):
    from nerf_sampling_amd import nerf_utils

    g = golden("render_rays_test")
    m = gpu_modules(scene)
    flags = {"full_nerf": dict(use_full_nerf=True), "nerf_max": dict(use_nerf_max_pts=True),
             "compare": dict(compare_nerf=True), "depthnet": {}}[mode]
    tr = make_trainer(n_depth_samples=32, sampling_mode="uniform", distance=0.1, **flags)
    res = nerf_utils.render_rays_test(T(g["ray_batch"]).cuda(), **render_kwargs(tr, m))
    prefix = f"{scene}_{mode}_"
    keys = [k[len(prefix):] for k in g if k.startswith(prefix)]
    assert set(keys) == set(res.keys())
    # host/device placement contract of the reference (nerf_utils.py:820-822, 866-870)
    assert res["depth_net_rgb_map"].is_cuda
    for k in keys:
        if k != "depth_net_rgb_map":
            assert not res[k].is_cuda, k
    for k in keys:
        exp = g[prefix + k]
        mine = npy(res[k])
        assert mine.shape[1:] == exp.shape[1:], (k, mine.shape, exp.shape)
        mine = mine[: exp.shape[0]]
        if k in ("depth_net_z_vals", "depth_net_pts", "max_z_vals", "max_pts") and mode == "depthnet":
            np.testing.assert_allclose(mine, exp, rtol=0, atol=3e-4, err_msg=k)   # DepthNet z: 5e-5 of [2,6]
            continue
        # hierarchical modes re-sample by inverting the coarse CDF, which has 1e-5-mass floor bins where a
        # 1e-7 change of the CDF moves a sample by a bin width; gate on the fraction of rays and the median
        hier = mode != "depthnet"
        if not hier and k in ("depth_net_rgb_map", "depth_net_disp_map"):
            continue                                       # gated below at 1e-4 on every well-conditioned ray
        tol = 2e-4 if ("rgb" in k or "disp" in k) else (5e-3 if hier else WEIGHTS_TOL)
        bad, err = frac_bad(mine, exp, tol)
        assert bad <= (0.03 if hier else 0.0), (k, bad, float(err.max()))
        assert np.median(err) < (1e-3 if hier and "rgb" not in k and "disp" not in k else 5e-5), (k, float(np.median(err)))
    if mode == "depthnet":
        fp32_gate(f"render_rays_test {scene}", m["params"], T(g["ray_batch"]), 32, "uniform", 0.1,
                  npy(res["depth_net_rgb_map"]), npy(res["depth_net_disp_map"]), g[prefix + "depth_net_rgb_map"],
                  g[prefix + "depth_net_disp_map"], measured=MEASURED_FP32.get(f"modes_{scene}"))
    if mode == "nerf_max":
        assert res["depth_net_disp_map"].shape == (256, 3)        # reference quirk, nerf_utils.py:826


@pytest.mark.parametrize("ns,mode,dist", [(2, "uniform", 0.01), (64, "uniform", 0.1), (1, "depth_only", 0.1)])
def test_render_rays_test_sampling_setups(golden, gpu_modules, ns, mode, dist):
    from nerf_sampling_amd import nerf_utils

    g = golden("render_rays_test")
    m = gpu_modules("lego_synth")
    tr = make_trainer(n_depth_samples=ns, sampling_mode=mode, distance=dist)
    res = nerf_utils.render_rays_test(T(g["ray_batch"]).cuda(), **render_kwargs(tr, m))
    for k in ("depth_net_rgb_map", "depth_net_disp_map", "depth_net_weights", "depth_net_z_vals"):
        exp = g[f"lego_synth_{mode}{ns}_{dist}_{k}"]
        mine = npy(res[k])
        assert mine.shape == exp.shape, (k, mine.shape, exp.shape)
        if exp.size == 0:
            continue
        if k in ("depth_net_rgb_map", "depth_net_disp_map"):
            continue                                       # gated below at 1e-4 on every well-conditioned ray
        tol = 3e-4 if k == "depth_net_z_vals" else WEIGHTS_TOL
        bad, err = frac_bad(mine, exp, tol)
        assert bad == 0.0 and np.median(err) < 5e-5, (k, bad, float(err.max()))
    fp32_gate(f"sampling set-up {mode}{ns}_{dist}", m["params"], T(g["ray_batch"]), ns, mode, dist,
              npy(res["depth_net_rgb_map"]), npy(res["depth_net_disp_map"]),
              g[f"lego_synth_{mode}{ns}_{dist}_depth_net_rgb_map"], g[f"lego_synth_{mode}{ns}_{dist}_depth_net_disp_map"],
              measured=MEASURED_FP32.get(f"setup_{mode}{ns}"))


@pytest.mark.parametrize("dtype", ["f32", "f16x3"])
def test_frame_config1_fp32_gate(golden, gpu_modules, dtype):
    """BASELINE config 1 (64x64, 32 samples/ray) through render_test, fp32 gate 1e-4 on rgb and disp -- for the
    exact-fp32 MFMA path and for the split-fp16 path (f16x3) on the 16-bit engine, which is held to the same gate."""
    from nerf_sampling_amd import nerf_utils, ops

    ops.set_compute_dtype(dtype)
    g = golden("frame64")
    m = gpu_modules("lego_synth")
    tr = make_trainer(n_depth_samples=32, sampling_mode="uniform", distance=0.1)
    kw = render_kwargs(tr, m)
    kw.update(near=2.0, far=6.0, ndc=False)
    rgb, disp, extras = nerf_utils.render_test(64, 64, g["K"], chunk=1024 * 32, c2w=T(g["c2w"]), **kw)
    assert rgb.shape == (64, 64, 3) and disp.shape == (64, 64)
    assert extras["depth_net_z_vals"].shape == (64, 64, 32) and extras["rays_o"].shape == (4096, 3)
    # sensitivity mask from the oracle: rays whose reference rgb moves > 1e-4 when sigma moves by 1e-4 * |sigma|_max
    p = m["params"]
    batch, o, d, _ = O.ray_batch_from_camera(64, 64, g["K"], T(g["c2w"]), 2.0, 6.0)
    mean = O.depthnet_forward(p["depth"], o, d)
    pts, z = O.place_samples(o, d, mean, 32, "uniform", 0.1)
    raw = O.run_network(p["fine"], pts, batch[:, -3:])
    eps = 1e-5 * float(raw[..., 3].abs().max())   # ~5x the measured fp32 sigma error (tools/gpu_diag.py)
    base = O.raw2outputs(raw, z, d, 0.0, True)[0]
    ill = torch.zeros(raw.shape[0], dtype=torch.bool)
    for sgn in (-1.0, 1.0):
        pert = raw.clone(); pert[..., 3] += sgn * eps
        ill |= ((O.raw2outputs(pert, z, d, 0.0, True)[0] - base).abs().max(-1).values > 1e-4)
    ill = ill.numpy()
    err_rgb = np.abs(npy(rgb) - g["rgb"]).reshape(-1, 3).max(-1)
    err_disp = np.abs(npy(disp) - g["disp"]).reshape(-1) / np.maximum(np.abs(g["disp"]).reshape(-1), 1.0)
    print(f"config1 [{dtype}]: ill-conditioned rays {ill.mean():.4f}, max rgb err on the rest {err_rgb[~ill].max():.2e}, "
          f"overall median {np.median(err_rgb):.2e}")
    assert ill.mean() < 0.01
    assert err_rgb[~ill].max() <= 1e-4
    assert err_disp[~ill].max() <= 1e-4
    assert np.mean(err_rgb > 1e-4) < 0.02
    mse = float(((npy(rgb) - g["rgb"]) ** 2).mean())
    assert -10 * np.log10(max(mse, 1e-20)) > 60.0   # PSNR(build || reference) in dB


@pytest.mark.parametrize("scene", ["tiny_synth", "lego_synth"])
def test_render_rays_train_forward(golden, gpu_modules, scene):
    from nerf_sampling_amd import nerf_utils

    g = golden("render_rays_train")
    m = gpu_modules(scene)
    tr = make_trainer()
    res = nerf_utils.render_rays(T(g["ray_batch"]).cuda(), **render_kwargs(tr, m))
    assert set(res) == {"depth_net_rgb_map", "depth_net_disp_map", "depth_net_z_vals", "max_z_vals",
                        "depth_net_pts", "max_pts", "raw"}
    for k in ("depth_net_pts", "max_pts", "raw"):
        assert not res[k].is_cuda                          # reference moves these to the host (:723-727)
    for k, v in res.items():
        exp = g[f"{scene}_{k}"]
        mine = npy(v)
        assert mine.shape == exp.shape, k
        if k in ("depth_net_z_vals", "depth_net_pts"):
            np.testing.assert_allclose(mine, exp, rtol=0, atol=3e-4)
        elif k == "raw":
            scale = np.abs(exp).reshape(-1, 4).max(0)
            bad, err = frac_bad(mine / scale, exp / scale, 5e-4)   # points differ by the DepthNet z tolerance
            assert bad <= 0.03, (k, bad)
        elif k in ("max_z_vals", "max_pts"):
            bad, err = frac_bad(mine, exp, 2e-3)
            assert bad <= 0.05, (k, bad)                    # argmax over near-tied weights may pick a neighbour
    # the single-sample DepthNet render (rgb = sigmoid(raw rgb) at the predicted depth, disp = 1e10): 1e-4 on every ray
    fp32_gate(f"render_rays (train fwd) {scene}", m["params"], T(g["ray_batch"]), 1, "depth_only", 0.1,
              npy(res["depth_net_rgb_map"]), npy(res["depth_net_disp_map"]), g[f"{scene}_depth_net_rgb_map"],
              g[f"{scene}_depth_net_disp_map"], max_ill=0.0, measured=MEASURED_FP32.get(f"train_{scene}"))


@pytest.mark.parametrize("lindisp", [True, False])
def test_sample_as_in_nerf(golden, gpu_modules, lindisp):
    from nerf_sampling_amd import nerf_utils

    g = golden("hierarchical")
    m = gpu_modules("tiny_synth")
    tr = make_trainer(lindisp=lindisp)
    kw = render_kwargs(tr, m)
    res = nerf_utils.sample_as_in_NeRF(ray_batch=T(g["ray_batch"]).cuda(), network_fn=kw["network_fn"],
                                       network_fine=kw["network_fine"], network_query_fn=kw["network_query_fn"],
                                       N_samples=64, trainer=tr, perturb=0.0, raw_noise_std=0.0, lindisp=lindisp,
                                       white_bkgd=True, kwargs={}, pytest=False)
    names = ("density", "z", "pts", "rgb_map", "weights", "alphas", "disp", "raw")
    assert res[1].shape == (96, 192)
    got = {nm: npy(v) for nm, v in zip(names, res)}
    exp = {nm: g[f"tiny_synth_lin{int(lindisp)}_{nm}"] for nm in names}
    # sample positions: inverse-CDF samples in floor bins are ill-conditioned (see above); the rest is tight
    bad, err = frac_bad(got["z"], exp["z"], 5e-3)
    assert bad <= 0.03 and np.median(np.abs(got["z"] - exp["z"])) < 1e-5, (bad, float(err.max()))
    assert (got["z"][:, 1:] >= got["z"][:, :-1]).all()
    np.testing.assert_allclose(got["pts"], npy(res[1])[..., None] * g["ray_batch"][:, None, 3:6] + g["ray_batch"][:, None, 0:3],
                               rtol=1e-6, atol=1e-6)
    # rendered outputs
    for nm, tol in (("rgb_map", 2e-4), ("disp", 2e-4)):
        scale = np.maximum(np.abs(exp[nm]), 1.0) if nm == "disp" else 1.0
        bad, err = frac_bad(got[nm] / scale, exp[nm] / scale, tol)
        assert bad <= 0.03 and np.median(err) < 5e-5, (nm, bad, float(np.median(err)))
    # per-sample outputs live at the (slightly moved) sample positions: compare where the positions agree
    same = np.abs(got["z"] - exp["z"]) < 1e-6
    assert same.mean() > 0.5
    scale = np.abs(exp["raw"]).reshape(-1, 4).max(0)
    assert (np.abs(got["raw"] - exp["raw"]) / scale)[same].max() < 1e-3
    assert np.allclose(got["density"], got["raw"][..., 3])


def test_fused_matches_operator_chain(gpu_modules):
    """ns_render_rays_depthnet (one C call, camera rays generated on the device) == the operator chain."""
    from nerf_sampling_amd import nerf_utils, ops

    m = gpu_modules("lego_synth")
    H = W = 48
    _, K = O.blender_intrinsics(H, W)
    c2w = O.pose_spherical(-63.0, -30.0, 4.0)[:3, :4]
    tr = make_trainer(n_depth_samples=64, sampling_mode="uniform", distance=0.1)
    kw = render_kwargs(tr, m)
    kw.update(near=2.0, far=6.0, ndc=False)
    rgb, disp, extras = nerf_utils.render_test(H, W, K, chunk=1024 * 32, c2w=c2w, **kw)
    out = ops.render_rays_depthnet(m["depth"].packed("f32"), m["fine"].packed("f32"),
                                   camera=(H, W, K, c2w, 0, H), n_samples=64, mode="uniform", std=0.1, extras=True)
    assert torch.equal(out["rgb"].reshape(H, W, 3), rgb)         # same kernels, same inputs: bit exact
    assert torch.equal(out["disp"].reshape(H, W), disp.cuda())
    assert torch.equal(out["z"].cpu().reshape(H, W, 64), extras["depth_net_z_vals"])
    assert torch.equal(out["weights"].cpu().reshape(H, W, 64), extras["depth_net_weights"])
    # row sharding: two half-frames concatenated == the full frame (what the multi-GPU path relies on)
    top = ops.render_rays_depthnet(m["depth"].packed("f32"), m["fine"].packed("f32"),
                                   camera=(H, W, K, c2w, 0, H // 2), n_samples=64, mode="uniform", std=0.1)
    bot = ops.render_rays_depthnet(m["depth"].packed("f32"), m["fine"].packed("f32"),
                                   camera=(H, W, K, c2w, H // 2, H), n_samples=64, mode="uniform", std=0.1)
    assert torch.equal(torch.cat([top["rgb"], bot["rgb"]]), out["rgb"])


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("scene,rows", [("lego_synth", 47), ("tiny_synth", 5)])
def test_fused_compositing_matches_operator_chain_16bit(gpu_modules, dtype, scene, rows):
    """Both one-call renderers on the 16-bit kernels -- the five-launch chain ns_render_rays_depthnet and the ONE-KERNEL
    ns_render_rays_fused (placement, MLP and compositing in one persistent kernel: no z / raw array in HBM) -- are
    bit-identical to the operator chain depthnet_forward -> place_samples -> nerf_forward -> raw2outputs, for ragged ray
    counts and several N; the one-kernel path on the production network with four and with five tiles per wave (a ray is one
    wave's own chunk, or a chunk that straddles two waves) and on the generic kernel.  N = 128, 192: a ray is several 64-sample
    chunks on different waves -- and, with 192 samples, in different groups of a workgroup's run (the open ray's transmittance and
    sums carry over in LDS)."""
    from nerf_sampling_amd import ops

    m = gpu_modules(scene)
    H, W = rows, 47                              # 2209 / 235 rays: ragged last workgroup
    _, K = O.blender_intrinsics(H, W)
    c2w = O.pose_spherical(40.0, -30.0, 4.0)[:3, :4]
    dn, nf = m["depth"].packed(dtype), m["fine"].packed(dtype)
    o, d, view = ops.get_rays(H, W, K, c2w)[:3]
    mean = ops.depthnet_forward(dn, o, d)
    for n in (64, 32, 96, 192, 16, 2, 128) + ((256, 512) if scene == "tiny_synth" else ()):   # (512: eight chunks per ray, runs of eight groups)
        pts, z = ops.place_samples(o, d, mean, n, "uniform", 0.1)
        raw = ops.nerf_forward_rays(nf, o, d, z, view)
        rgb, disp, acc, depth, alphas, weights = ops.raw2outputs(raw, z, d, None, True)
        variants = [dict(one_kernel=False)]
        if n != 96:                              # (96 is neither a power of two <= 64 nor a multiple of 64: only the chain serves it)
            variants += [dict(one_kernel=True, prod_tiles=t, generic_kernels=g) for t, g in ((0, 0), (4, 0), (5, 0), (0, 1))]
        else:
            with pytest.raises(NotImplementedError):
                ops.render_rays_depthnet(dn, nf, camera=(H, W, K, c2w, 0, H), n_samples=n, mode="uniform", std=0.1, one_kernel=True)
        for v in variants:
            one = v.pop("one_kernel")
            with ops.debug_switch(**v):
                out = ops.render_rays_depthnet(dn, nf, camera=(H, W, K, c2w, 0, H), n_samples=n, mode="uniform", std=0.1,
                                               extras=True, one_kernel=one)
                lean = ops.render_rays_depthnet(dn, nf, camera=(H, W, K, c2w, 0, H), n_samples=n, mode="uniform", std=0.1,
                                                one_kernel=one)       # without per-sample outputs (the benchmark configuration)
                torch.cuda.synchronize()
            tag = (n, one, v)
            assert torch.equal(out["z"], z), tag
            assert torch.equal(out["pts"], pts), tag
            assert torch.equal(out["rgb"].view(torch.int32), rgb.view(torch.int32)), (tag, (out["rgb"] - rgb).abs().max().item())
            assert torch.equal(out["disp"].view(torch.int32), disp.view(torch.int32)), tag
            assert torch.equal(out["weights"].view(torch.int32), weights.view(torch.int32)), tag
            assert torch.equal(lean["rgb"].view(torch.int32), rgb.view(torch.int32)) and torch.equal(lean["disp"].view(torch.int32), disp.view(torch.int32)), tag
    # explicit rays (the operator path) and an interleaved [R, 4] shard as the output, through the one-kernel renderer
    shard = torch.empty((o.shape[0], 4), dtype=torch.float32, device="cuda")
    pts, z = ops.place_samples(o, d, mean, 64, "uniform", 0.1)
    rgb, disp = ops.raw2outputs(ops.nerf_forward_rays(nf, o, d, z, view), z, d, None, True)[:2]
    out = ops.render_rays_depthnet(dn, nf, rays=(o, d, view), n_samples=64, mode="uniform", std=0.1, shard=shard, one_kernel=True)
    assert torch.equal(shard[:, :3], rgb) and torch.equal(shard[:, 3], disp) and out["rgb"].data_ptr() == shard.data_ptr()
    # rays that miss the DepthNet's sphere: NaN depth, NaN samples, NaN pixel -- in both renderers, bit for bit, and only there
    o2, d2 = o.clone(), d.clone()
    miss = torch.arange(0, o.shape[0], 7, device="cuda")
    d2[miss] = torch.tensor([0.0, 0.0, 1.0], device="cuda")          # pointing away from the scene
    for n in (64, 16):
        a = ops.render_rays_depthnet(dn, nf, rays=(o2, d2, view), n_samples=n, mode="uniform", std=0.1, extras=True, one_kernel=True)
        b = ops.render_rays_depthnet(dn, nf, rays=(o2, d2, view), n_samples=n, mode="uniform", std=0.1, extras=True, one_kernel=False)
        for k in ("rgb", "disp", "z", "weights", "pts"):
            assert torch.equal(a[k].view(torch.int32), b[k].view(torch.int32)), (n, k)
        nan_rays = torch.isnan(a["rgb"]).any(-1)
        assert nan_rays[miss].all() and int(nan_rays.sum()) == miss.numel()



def test_one_kernel_renderers_on_a_handful_of_rays(gpu_modules):
    """One, three and seven rays (a single, mostly empty workgroup; a several-chunk ray that is the launch's only run) through
    the one-kernel renderer, its guards and the hierarchical renderer's in-epilogue compositing: the chain's bits."""
    from nerf_sampling_amd import ops

    m = gpu_modules("lego_synth")
    dn, nf, gw, nc = m["depth"].packed("f16"), m["fine"].packed("bf16"), m["fine"].packed("f16x3"), m["coarse"].packed("bf16")
    _, K = O.blender_intrinsics(8, 8)
    o, d, view = ops.get_rays(8, 8, K, O.pose_spherical(10.0, -30.0, 4.0)[:3, :4])[:3]
    for R in (1, 3, 7):
        rays = (o[:R].contiguous(), d[:R].contiguous(), view[:R].contiguous())
        for n in (64, 2, 192, 512):
            for t in (0, 4, 5):
                with ops.debug_switch(prod_tiles=t):
                    a = ops.render_rays_depthnet(dn, nf, rays=rays, n_samples=n, mode="uniform", std=0.1, extras=True, one_kernel=True)
                    b = ops.render_rays_depthnet(dn, nf, rays=rays, n_samples=n, mode="uniform", std=0.1, extras=True, one_kernel=False)
                for k in ("rgb", "disp", "z", "weights", "pts"):
                    assert torch.equal(a[k].view(torch.int32), b[k].view(torch.int32)), (R, n, t, k)
        for thr in (0.0, 16.0, 1e6):
            a = ops.render_rays_depthnet(dn, nf, rays=rays, n_samples=64, mode="uniform", std=0.1, one_kernel=True, guard=gw, guard_threshold=thr)
            b = ops.render_rays_depthnet(dn, nf, rays=rays, n_samples=64, mode="uniform", std=0.1, one_kernel=False, guard=gw)
            assert torch.equal(a["rgb"].view(torch.int32), b["rgb"].view(torch.int32)) or thr == 16.0, (R, thr)
        h1 = ops.render_rays_hierarchical(nc, nf, rays=rays, n_coarse=64, n_importance=128, extras=True)
        with ops.debug_switch(hier_chain=1):
            h0 = ops.render_rays_hierarchical(nc, nf, rays=rays, n_coarse=64, n_importance=128, extras=True)
        for k in ("rgb", "disp", "z", "weights", "raw"):
            assert torch.equal(h1[k].view(torch.int32), h0[k].view(torch.int32)), (R, k)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_psnr_guard_replaces_sigma_of_the_last_sample(gpu_modules, dtype):
    """ns_render_args::nerf_guard on both one-call renderers: the last sample of every ray -- composited with dist = 1e10, so
    alpha = step(sigma) (sampling_trainer.py:176-180) -- is evaluated a second time through an f16x3 packing of the field and
    its sigma replaces the 16-bit one.  Expected result built from the operators: the 16-bit raw with raw[:, -1, 3] taken from
    the f16x3 network at the same point; the one-kernel renderer and the chain agree with it bit for bit."""
    from nerf_sampling_amd import ops

    m = gpu_modules("lego_synth")
    H, W = 37, 47
    _, K = O.blender_intrinsics(H, W)
    c2w = O.pose_spherical(-25.0, -30.0, 4.0)[:3, :4]
    dn, nf, gw = m["depth"].packed("f16x3"), m["fine"].packed(dtype), m["fine"].packed("f16x3")
    o, d, view = ops.get_rays(H, W, K, c2w)[:3]
    mean = ops.depthnet_forward(dn, o, d)
    for n in (64, 32, 192):
        pts, z = ops.place_samples(o, d, mean, n, "uniform", 0.1)
        raw = ops.nerf_forward_rays(nf, o, d, z, view)
        raw_last = ops.nerf_forward_rays(gw, o, d, z[:, -1:].contiguous(), view)
        patched = raw.clone()
        patched[:, -1, 3] = raw_last[:, 0, 3]
        assert float((patched[:, -1, 3] - raw[:, -1, 3]).abs().max()) > 0       # the guard changes something
        rgb, disp, _acc, _depth, _al, weights = ops.raw2outputs(patched, z, d, None, True)
        for one, thr in ((True, 0.0), (False, 0.0), (True, None), (True, 2.0)):
            # thr 0: every ray before the kernel.  None (the module's 16) / 2: the one-kernel renderer flags the rays whose own
            # sigma_last is within thr of zero, re-evaluates those after the kernel and re-adds their last share -- the same bits
            # wherever the 16-bit and the fp32-grade sigma differ by less than thr (n = 192: several chunks, every ray as before)
            out = ops.render_rays_depthnet(dn, nf, camera=(H, W, K, c2w, 0, H), n_samples=n, mode="uniform", std=0.1,
                                           extras=True, one_kernel=one, guard=gw, guard_threshold=thr)
            lim = 16.0 if thr is None else thr
            same = torch.ones(rgb.shape[0], dtype=torch.bool, device="cuda")
            if lim > 0 and n <= 64:
                s16, s32 = raw[:, -1, 3], raw_last[:, 0, 3]
                same = ((s16 > 0) == (s32 > 0)) | (s16.abs() < lim)          # unflagged rays whose step flips keep the 16-bit one
                assert float(same.float().mean()) > (0.999 if thr is None else 0.98), (n, thr, float(same.float().mean()))
                if thr == 2.0:
                    assert int((s16.abs() < lim).sum()) < rgb.shape[0] // 2      # ... and the flagged set is a minority
            tag = (n, one, thr)
            assert torch.equal(out["rgb"][same].view(torch.int32), rgb[same].view(torch.int32)), (tag, (out["rgb"] - rgb).abs().max().item())
            assert torch.equal(out["disp"][same].view(torch.int32), disp[same].view(torch.int32)) and torch.equal(out["z"], z), tag
            assert torch.equal(out["weights"][same].view(torch.int32), weights[same].view(torch.int32)), tag
    # explicit rays, an interleaved [R, 4] shard as the output, per-sample outputs asked for: the fix-up kernel writes through the
    # same strides and patches the last weight
    shard = torch.empty((o.shape[0], 4), dtype=torch.float32, device="cuda")
    every = ops.render_rays_depthnet(dn, nf, rays=(o, d, view), n_samples=64, mode="uniform", std=0.1, extras=True, one_kernel=True,
                                     guard=gw, guard_threshold=0.0)
    sel = ops.render_rays_depthnet(dn, nf, rays=(o, d, view), n_samples=64, mode="uniform", std=0.1, extras=True, one_kernel=True,
                                   guard=gw, guard_threshold=1e6, shard=shard)       # every ray is flagged: the capacity case
    assert torch.equal(shard[:, :3].view(torch.int32), every["rgb"].view(torch.int32)) and sel["rgb"].data_ptr() == shard.data_ptr()
    assert torch.equal(shard[:, 3].view(torch.int32), every["disp"].view(torch.int32))
    assert torch.equal(sel["weights"].view(torch.int32), every["weights"].view(torch.int32))
    # a guard handle that is not a split-fp16 packing (here: exact fp32) takes the every-ray pass whatever the threshold says
    g32 = m["fine"].packed("f32")
    a32 = ops.render_rays_depthnet(dn, nf, rays=(o, d, view), n_samples=64, mode="uniform", std=0.1, one_kernel=True, guard=g32)
    b32 = ops.render_rays_depthnet(dn, nf, rays=(o, d, view), n_samples=64, mode="uniform", std=0.1, one_kernel=False, guard=g32)
    assert torch.equal(a32["rgb"].view(torch.int32), b32["rgb"].view(torch.int32))
    assert float((a32["rgb"] - every["rgb"]).abs().max()) < 1e-3              # (the fp32 and the f16x3 sigma agree to fp32 rounding)
    with pytest.raises((NotImplementedError, ValueError)):                       # the guard pass is defined for uniform placement
        ops.render_rays_depthnet(dn, nf, camera=(H, W, K, c2w, 0, H), n_samples=8, mode="gaussian", std=0.1, guard=gw)
    with pytest.raises(ValueError):                                              # ... and for another packing of the SAME network
        ops.render_rays_depthnet(dn, nf, camera=(H, W, K, c2w, 0, H), n_samples=8, mode="uniform", std=0.1,
                                 guard=gpu_modules("tiny_synth")["fine"].packed("f16x3"))


def step_rule_mask(raw_ref, z_ref, d, rgb_ref, sigma_last_eps, tol=1e-2):
    """Rays whose ORACLE colour moves by more than `tol` when sigma of the LAST sample alone moves by +-sigma_last_eps: the
    reference composites the last sample with dist = 1e10 (sampling_trainer.py:176-180), so alpha_last = step(sigma_last)
    and a ray's colour is discontinuous in sigma_last wherever transmittance is left.  Only that one value is perturbed:
    the mask isolates the step rule and says nothing about a ray that is merely semi-transparent."""
    ill = torch.zeros(raw_ref.shape[0], dtype=torch.bool)
    for sgn in (-1.0, 1.0):
        pert = raw_ref.clone()
        pert[:, -1, 3] += sgn * sigma_last_eps
        ill |= (O.raw2outputs(pert, z_ref, d, 0.0, True)[0] - rgb_ref).abs().max(-1).values > tol
    return ill


def _psnr(a, b):
    mse = float(((a - b) ** 2).mean())
    return float("inf") if mse == 0 else -10 * np.log10(mse)


@pytest.fixture(scope="module")
def oracle_band():
    """BASELINE configs[1] shape: a 50-row band of an 800x800 frame, DepthNet + 64 samples/ray, rendered by the oracle."""
    torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
    p = O.make_scene("lego_synth")
    H = W = 800
    _, K = O.blender_intrinsics(H, W)
    c2w = O.pose_spherical(30.0, -30.0, 4.0)[:3, :4]
    r0, rows = 375, 50
    batch, o, d, _ = O.ray_batch_from_camera(H, W, K, c2w, 2.0, 6.0)
    sl = slice(r0 * W, (r0 + rows) * W)
    batch, o, d = batch[sl], o[sl], d[sl]
    with torch.no_grad():
        z_mean = O.depthnet_forward(p["depth"], o, d)
        pts, z = O.place_samples(o, d, z_mean, 64, "uniform", 0.1)
        raw = O.run_network(p["fine"], pts, batch[:, -3:])
        rgb = O.raw2outputs(raw, z, d, 0.0, True)[0]
    return dict(p=p, H=H, W=W, K=K, c2w=c2w, r0=r0, rows=rows, o=o, d=d, view=batch[:, -3:], z_mean=z_mean, z=z, raw=raw,
                rgb=rgb)


#                                  z rms   sigma/max  all-ray PSNR  step-rule frac  PSNR with oracle sigma_last   (gates <= ~3x measured)
@pytest.mark.parametrize("dtype,g_z,g_sig,g_all,g_step,g_fix", [("bf16", 1.0e-2, 4.0e-2, 21.0, 0.25, 47.0),   # measured 3.3e-3, 1.54e-2, 24.2 dB, 0.106, 52.5 dB
                                                                ("f16", 1.4e-3, 7.0e-3, 28.0, 0.06, 54.0)])
def test_frame16_vs_oracle(gpu_modules, oracle_band, dtype, g_z, g_sig, g_all, g_step, g_fix):
    """The 16-bit paths against the ORACLE at image level on the configs[1] shape, on the STRESS scene (lego_synth: seeded
    random weights whose density crosses zero at the last sample of every fifth ray; the realistic, fitted scene is
    tests/test_scene_psnr.py).  Gated: the stage errors at identical inputs; the all-ray PSNR; and the two figures that
    isolate the reference's last-sample step rule (alpha_last = step(sigma_last)) with NO per-dtype constant -- the
    fraction of rays whose oracle colour flips when sigma_last alone moves by 3x the sigma_last error measured in this
    very run, and the PSNR of the HIP raw composited with only sigma_last taken from the oracle.

    Measured (MI355X): bf16 / f16  z rms 3.3e-3 / 4.6e-4;  sigma noise 1.54e-2 / 2.4e-3 of max |sigma|; all-ray PSNR
    24.2 / 31.3 dB; with the oracle's sigma_last 52.5 / 60.1 dB."""
    from nerf_sampling_amd import ops

    b = oracle_band
    m = gpu_modules("lego_synth")
    dn, nf = m["depth"].packed(dtype), m["fine"].packed(dtype)
    oc, dc, vc = b["o"].cuda(), b["d"].cuda(), b["view"].cuda()
    # stage errors at identical inputs
    z_mean = ops.depthnet_forward(dn, oc, dc)
    z_rms = float((z_mean.cpu() - b["z_mean"]).pow(2).mean().sqrt())
    _, zz = ops.place_samples(oc, dc, z_mean, 64, "uniform", 0.1)
    raw = ops.nerf_forward_rays(nf, oc, dc, zz, vc).cpu()
    with torch.no_grad():
        pts_o, _ = O.place_samples(b["o"], b["d"], z_mean.cpu(), 64, "uniform", 0.1)
        raw_o = O.run_network(b["p"]["fine"], pts_o, b["view"])          # the oracle's MLP at the HIP path's own points
        rgb_o_given_z = O.raw2outputs(raw_o, zz.cpu(), b["d"], 0.0, True)[0]
    sig_max = float(b["raw"][..., 3].abs().max())
    sig_frac = float((raw[..., 3] - raw_o[..., 3]).pow(2).mean().sqrt()) / sig_max
    last_rms = float((raw[:, -1, 3] - raw_o[:, -1, 3]).pow(2).mean().sqrt())
    logit_rms = float((raw[..., :3] - raw_o[..., :3]).pow(2).mean().sqrt())
    # the frame itself: the one-call fused path on camera rays (what bench.py times)
    out = ops.render_rays_depthnet(dn, nf, camera=(b["H"], b["W"], b["K"], b["c2w"], b["r0"], b["r0"] + b["rows"]),
                                   n_samples=64, mode="uniform", std=0.1)
    rgb = out["rgb"].cpu()
    err = (rgb - b["rgb"]).abs().max(-1).values
    ill = step_rule_mask(b["raw"], b["z"], b["d"], b["rgb"], 3.0 * last_rms)
    psnr_all = _psnr(rgb, b["rgb"])
    # the last-sample rule in isolation: HIP raw, oracle's sigma_last
    raw_fix = raw.clone()
    raw_fix[:, -1, 3] = raw_o[:, -1, 3]
    with torch.no_grad():
        rgb_fix = O.raw2outputs(raw_fix, zz.cpu(), b["d"], 0.0, True)[0]
    psnr_fix = _psnr(rgb_fix, rgb_o_given_z)
    print(f"{dtype}: z rms {z_rms:.2e}; sigma noise {sig_frac:.2e} of max|sigma| ({sig_max:.1f}), at the last sample rms "
          f"{last_rms:.3g}; rgb-logit rms {logit_rms:.2e}; PSNR all rays {psnr_all:.2f} dB; step-rule rays "
          f"{float(ill.float().mean()):.4f}; err>1e-2: {float((err > 1e-2).float().mean()):.4f} of rays; with oracle "
          f"sigma_last {psnr_fix:.2f} dB")
    assert z_rms < g_z
    assert sig_frac < g_sig
    assert psnr_all > g_all
    assert float(ill.float().mean()) < g_step
    assert psnr_fix > g_fix                                    # the step rule at the last sample is what differs


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_full_size_frame_properties(gpu_modules, dtype):
    """BASELINE config 2 size (800x800, DepthNet + 64 samples/ray): size-independent properties of the
    16-bit MFMA path (accuracy against the oracle: test_frame16_vs_oracle above)."""
    from nerf_sampling_amd import ops

    m = gpu_modules("lego_synth")
    H = W = 800
    _, K = O.blender_intrinsics(H, W)
    c2w = O.pose_spherical(30.0, -30.0, 4.0)[:3, :4]
    out = ops.render_rays_depthnet(m["depth"].packed(dtype), m["fine"].packed(dtype), camera=(H, W, K, c2w, 0, H),
                                   n_samples=64, mode="uniform", std=0.1, extras=True)
    rgb, w, z = out["rgb"], out["weights"], out["z"]
    assert rgb.shape == (H * W, 3) and torch.isfinite(rgb).all()
    assert float(rgb.min()) >= -1e-5 and float(rgb.max()) <= 1.0 + 1e-5        # white background compositing
    assert (w >= 0).all() and float(w.sum(-1).max()) <= 1.0 + 1e-4              # weights are a sub-partition of 1
    assert (z[:, 1:] >= z[:, :-1]).all() and float(z.min()) >= 2.0 and float(z.max()) <= 6.0
    band = (300, 500)
    lo = ops.render_rays_depthnet(m["depth"].packed(dtype), m["fine"].packed(dtype),
                                  camera=(H, W, K, c2w, band[0], band[1]), n_samples=64, mode="uniform", std=0.1)
    assert torch.equal(lo["rgb"], rgb[band[0] * W : band[1] * W])               # shard == slice of the frame
    ref = ops.render_rays_depthnet(m["depth"].packed("f32"), m["fine"].packed("f32"),
                                   camera=(H, W, K, c2w, band[0], band[1]), n_samples=64, mode="uniform", std=0.1)
    med = float((lo["rgb"] - ref["rgb"]).abs().max(-1).values.median())
    assert med < (1e-3 if dtype == "bf16" else 1.5e-4)


def test_fused_hierarchical_matches_operator_chain(golden, gpu_modules):
    """ns_render_rays_hierarchical (one C call) == sample_as_in_NeRF through the mirrored operators,
    and matches the reference golden within the hierarchical-path conditioning."""
    from nerf_sampling_amd import nerf_utils, ops

    g = golden("hierarchical")
    m = gpu_modules("tiny_synth")
    rb = T(g["ray_batch"]).cuda()
    for lindisp in (True, False):
        tr = make_trainer(lindisp=lindisp)
        kw = render_kwargs(tr, m)
        chain = nerf_utils.sample_as_in_NeRF(ray_batch=rb, network_fn=kw["network_fn"], network_fine=kw["network_fine"],
                                             network_query_fn=kw["network_query_fn"], N_samples=64, trainer=tr,
                                             perturb=0.0, raw_noise_std=0.0, lindisp=lindisp, white_bkgd=True,
                                             kwargs={}, pytest=False)
        fused = ops.render_rays_hierarchical(m["coarse"].packed("f32"), m["fine"].packed("f32"),
                                             rays=(rb[:, 0:3], rb[:, 3:6], rb[:, 8:11]), n_coarse=64, n_importance=128,
                                             lindisp=lindisp, white_bkgd=True, extras=True)
        assert torch.equal(fused["z"], chain[1]) and torch.equal(fused["rgb"], chain[3])
        assert torch.equal(fused["weights"], chain[4]) and torch.equal(fused["raw"], chain[7])
        exp = g[f"tiny_synth_lin{int(lindisp)}_rgb_map"]
        bad, err = frac_bad(npy(fused["rgb"]), exp, 2e-4)
        assert bad <= 0.03 and np.median(err) < 5e-5
    # stratified jitter + random inverse-CDF draws injected (perturb = 1 of the reference, seeded)
    fused = ops.render_rays_hierarchical(m["coarse"].packed("f32"), m["fine"].packed("f32"),
                                         rays=(rb[:, 0:3], rb[:, 3:6], rb[:, 8:11]), n_coarse=64, n_importance=128,
                                         lindisp=True, white_bkgd=True, t_rand=T(g["perturb_t_rand"]).cuda(),
                                         u=T(g["perturb_u"]).cuda(), extras=True)
    bad, err = frac_bad(npy(fused["z"]), g["perturb_z"], 5e-3)
    assert bad <= 0.03
    bad, err = frac_bad(npy(fused["rgb"]), g["perturb_rgb_map"], 2e-4)
    assert bad <= 0.03 and np.median(err) < 5e-5


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_hierarchical_in_kernel_compositing_matches_chain(gpu_modules, dtype):
    """ns_render_rays_hierarchical on 16-bit fields composites both passes in the MLP kernel's epilogue (the coarse pass yields
    only its weights, the fine pass rgb / disp / weights: no raw [R,N,4] array in HBM).  Bit-identical to the same call with the
    raw arrays and the stand-alone compositing kernel (debug switch hier_chain), on every kernel variant, for a ragged ray
    count; 64 + 128 samples (a fine ray is three chunks) and 32 + 32 (one chunk)."""
    from nerf_sampling_amd import ops

    m = gpu_modules("lego_synth")
    H, W = 23, 47
    _, K = O.blender_intrinsics(H, W)
    c2w = O.pose_spherical(-70.0, -30.0, 4.0)[:3, :4]
    nc, nf = m["coarse"].packed(dtype), m["fine"].packed(dtype)
    for n_c, n_i in ((64, 128), (32, 32), (64, 64)):
        kw = dict(camera=(H, W, K, c2w, 0, H), n_coarse=n_c, n_importance=n_i, lindisp=True, white_bkgd=True)
        with ops.debug_switch(hier_chain=1):
            ref = ops.render_rays_hierarchical(nc, nf, extras=True, **kw)
            torch.cuda.synchronize()
        for t, g in ((0, 0), (4, 0), (5, 0), (0, 1)):
            with ops.debug_switch(prod_tiles=t, generic_kernels=g):
                out = ops.render_rays_hierarchical(nc, nf, extras=True, **kw)
                lean = ops.render_rays_hierarchical(nc, nf, **kw)
                torch.cuda.synchronize()
            if g == 0:
                for k in ("z", "raw", "weights", "rgb", "disp"):
                    assert torch.equal(out[k].view(torch.int32), ref[k].view(torch.int32)), (n_c, n_i, t, g, k)
                assert torch.equal(lean["rgb"].view(torch.int32), ref["rgb"].view(torch.int32)), (n_c, n_i, t)
                assert torch.equal(lean["disp"].view(torch.int32), ref["disp"].view(torch.int32)), (n_c, n_i, t)
            else:       # the generic kernel's raw differs from the generated streams' in rounding: compare it with its own chain
                with ops.debug_switch(hier_chain=1, generic_kernels=1):
                    refg = ops.render_rays_hierarchical(nc, nf, extras=True, **kw)
                    torch.cuda.synchronize()
                for k in ("z", "raw", "weights", "rgb", "disp"):
                    assert torch.equal(out[k].view(torch.int32), refg[k].view(torch.int32)), (n_c, n_i, "generic", k)


def test_full_size_hierarchical_config3(gpu_modules):
    """BASELINE config 3 size: 800x800, vanilla 64 + 128 samples/ray (coarse + fine MLP), bf16."""
    from nerf_sampling_amd import ops

    m = gpu_modules("lego_synth")
    H = W = 800
    _, K = O.blender_intrinsics(H, W)
    c2w = O.pose_spherical(99.0, -30.0, 4.0)[:3, :4]
    out = ops.render_rays_hierarchical(m["coarse"].packed("bf16"), m["fine"].packed("bf16"), camera=(H, W, K, c2w, 0, H),
                                       n_coarse=64, n_importance=128, lindisp=True, white_bkgd=True, extras=True)
    rgb, w, z = out["rgb"], out["weights"], out["z"]
    assert rgb.shape == (H * W, 3) and z.shape == (H * W, 192) and torch.isfinite(rgb).all()
    assert float(rgb.min()) >= -1e-5 and float(rgb.max()) <= 1 + 1e-5
    assert (z[:, 1:] >= z[:, :-1]).all() and float(z.min()) >= 2.0 - 1e-5 and float(z.max()) <= 6.0 + 1e-5
    assert (w >= 0).all() and float(w.sum(-1).max()) <= 1 + 1e-4
    half = ops.render_rays_hierarchical(m["coarse"].packed("bf16"), m["fine"].packed("bf16"),
                                        camera=(H, W, K, c2w, 400, 800), n_coarse=64, n_importance=128, lindisp=True)
    assert torch.equal(half["rgb"], rgb[400 * W :])


def test_full_size_config5_fp16(gpu_modules):
    """BASELINE config 5 size: 1600x1600, DepthNet + 192 samples/ray, fp16 MFMA (one GPU's 200-row shard of 8)."""
    from nerf_sampling_amd import ops

    m = gpu_modules("lego_synth")
    H = W = 1600
    _, K = O.blender_intrinsics(H, W)
    c2w = O.pose_spherical(-135.0, -30.0, 4.0)[:3, :4]
    out = ops.render_rays_depthnet(m["depth"].packed("f16"), m["fine"].packed("f16"), camera=(H, W, K, c2w, 600, 800),
                                   n_samples=192, mode="uniform", std=0.1, extras=True)
    rgb, w, z = out["rgb"], out["weights"], out["z"]
    assert rgb.shape == (200 * W, 3) and z.shape == (200 * W, 192) and torch.isfinite(rgb).all()
    assert (z[:, 1:] >= z[:, :-1]).all() and float(z.min()) >= 2.0 and float(z.max()) <= 6.0
    assert (w >= 0).all() and float(w.sum(-1).max()) <= 1 + 1e-4
    # gaussian placement at the same size: sorted, contains the mean, injected noise honoured
    noise = torch.randn(200 * W, 191, device="cuda")
    g1 = ops.render_rays_depthnet(m["depth"].packed("f16"), m["fine"].packed("f16"), camera=(H, W, K, c2w, 600, 800),
                                  n_samples=192, mode="gaussian", std=0.1, noise=noise, extras=True)
    g2 = ops.render_rays_depthnet(m["depth"].packed("f16"), m["fine"].packed("f16"), camera=(H, W, K, c2w, 600, 800),
                                  n_samples=192, mode="gaussian", std=0.1, noise=noise, extras=True)
    assert torch.equal(g1["rgb"], g2["rgb"]) and (g1["z"][:, 1:] >= g1["z"][:, :-1]).all()


@pytest.mark.parametrize("flags", [{}, dict(compare_nerf=True), dict(use_nerf_max_pts=True)])
def test_async_host_copies_equal_blocking_copies(gpu_modules, flags):
    """render_test's per-chunk host copies (nerf_utils.py:820-822, 866-870) through the pinned asynchronous sink return
    exactly what the reference-style blocking `.cpu()` + host concatenation returns: same keys, shapes, devices, values,
    over several ragged chunks, and a second frame does not overwrite the first one's tensors."""
    from nerf_sampling_amd import nerf_utils

    m = gpu_modules("tiny_synth")
    H, W = 23, 31                                  # 713 rays in chunks of 200: 4 chunks, ragged tail
    _, K = O.blender_intrinsics(H, W)
    tr = make_trainer(n_depth_samples=16, sampling_mode="uniform", distance=0.1, **flags)
    kw = render_kwargs(tr, m)
    kw.update(near=2.0, far=6.0, ndc=False)
    c2w = O.pose_spherical(77.0, -30.0, 4.0)[:3, :4]
    a_rgb, a_disp, a = nerf_utils.render_test(H, W, K, chunk=200, c2w=c2w, **kw)
    b_rgb, b_disp, b = nerf_utils.render_test(H, W, K, chunk=200, c2w=c2w, _blocking_host_copies=True, **kw)
    keep = {k: v.clone() for k, v in a.items()}
    nerf_utils.render_test(H, W, K, chunk=200, c2w=O.pose_spherical(-20.0, -30.0, 4.0)[:3, :4], **kw)   # another frame
    assert torch.equal(a_rgb, b_rgb) and a_rgb.is_cuda
    assert torch.equal(a_disp.cpu(), b_disp.cpu()) and a_disp.is_cuda == b_disp.is_cuda
    assert set(a) == set(b)
    for k in a:
        assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype and a[k].is_cuda == b[k].is_cuda, k
        assert torch.equal(a[k].cpu(), b[k].cpu()), k
        assert torch.equal(a[k], keep[k]), k       # untouched by the later frame


def test_config5_band_vs_oracle_f16():
    """BASELINE configs[4] shape (1600x1600, DepthNet + 192 samples/ray, fp16 MFMA) on the FITTED scene: a 6-row band
    against the oracle over ALL rays, and the scene PSNR against the analytic ground truth on both sides -- plain fp16 (gated
    against gross error: 9 600 rays of a 27 dB scene put the 0.05 dB bar inside the noise of a handful of step-rule rays) and
    with the PSNR guard (f16x3 DepthNet + last sample on f16x3), which has to hold the bar."""
    from conftest import _make_modules
    from nerf_sampling_amd import analytic_scene, ops

    torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
    m = _make_modules("shapes_fit")
    p = m["params"]
    H = W = 1600
    N = 192
    _, K = O.blender_intrinsics(H, W)
    c2w = O.render_poses(40)[7][:3, :4]
    r0, rows = 880, 6
    batch, o, d, _ = O.ray_batch_from_camera(H, W, K, c2w, 2.0, 6.0)
    sl = slice(r0 * W, (r0 + rows) * W)
    batch, o, d = batch[sl], o[sl], d[sl]
    with torch.no_grad():
        z_mean = O.depthnet_forward(p["depth"], o, d)
        pts, z = O.place_samples(o, d, z_mean, N, "uniform", 0.1)
        raw = O.run_network(p["fine"], pts, batch[:, -3:])
        rgb_ref = O.raw2outputs(raw, z, d, 0.0, True)[0]
    gt = analytic_scene.frame(H, W, K, c2w, r0, r0 + rows)[0].reshape(-1, 3)
    p_ref = _psnr(rgb_ref, gt)
    for guarded in (False, True):
        dn = m["depth"].packed("f16x3" if guarded else "f16")
        out = ops.render_rays_depthnet(dn, m["fine"].packed("f16"), camera=(H, W, K, c2w, r0, r0 + rows), n_samples=N,
                                       mode="uniform", std=0.1, guard=m["fine"].packed("f16x3") if guarded else None)
        rgb = out["rgb"].cpu()
        err = (rgb - rgb_ref).abs().max(-1).values
        p_build = _psnr(rgb, gt)
        print(f"config5 f16 band (fitted scene){' + PSNR guard' if guarded else ''}: PSNR build vs oracle, all rays "
              f"{_psnr(rgb, rgb_ref):.2f} dB; median |err| {float(err.median()):.2e}; err>1e-2 {float((err > 1e-2).float().mean()):.5f}; "
              f"scene PSNR oracle {p_ref:.4f} dB, build {p_build:.4f} dB, delta {p_build - p_ref:+.4f} dB")
        assert _psnr(rgb, rgb_ref) > (45.0 if guarded else 32.0) and float(err.median()) < 5e-4
        assert abs(p_build - p_ref) <= (0.05 if guarded else 0.4)


@pytest.mark.parametrize("dtype", ["f32", "f16x3", "bf16"])
def test_config3_band_vs_oracle(gpu_modules, dtype):
    """BASELINE configs[2] shape (800x800, vanilla 64 coarse + 128 importance samples, coarse + fine MLP): a 5-row band
    of the one-call hierarchical path against the oracle's sample_as_in_NeRF.  fp32 / f16x3: the hierarchical gates of the
    small golden tests (inverse-CDF samples in floor bins are ill-conditioned: fraction + median); bf16: reported, with
    loose statistical bounds (its parity statement is the kernel-level one)."""
    from nerf_sampling_amd import ops

    m = gpu_modules("lego_synth")
    p = m["params"]
    H = W = 800
    _, K = O.blender_intrinsics(H, W)
    c2w = O.pose_spherical(99.0, -30.0, 4.0)[:3, :4]
    r0, rows = 398, 5
    batch, _, _, _ = O.ray_batch_from_camera(H, W, K, c2w, 2.0, 6.0)
    batch = batch[r0 * W : (r0 + rows) * W]
    torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
    with torch.no_grad():
        _dens, z_ref, _pts, rgb_ref, _w, _al, disp_ref, _raw = O.hierarchical_render(batch, p["coarse"], p["fine"], 64, 128,
                                                                                    True, True)
    out = ops.render_rays_hierarchical(m["coarse"].packed(dtype), m["fine"].packed(dtype), camera=(H, W, K, c2w, r0, r0 + rows),
                                       n_coarse=64, n_importance=128, lindisp=True, white_bkgd=True, extras=True)
    rgb, z = out["rgb"].cpu(), out["z"].cpu()
    assert (z[:, 1:] >= z[:, :-1]).all()
    err = (rgb - rgb_ref).abs().max(-1).values
    zerr = (z - z_ref).abs().max(-1).values
    print(f"config3 [{dtype}] band: rgb median {float(err.median()):.2e} frac>2e-4 {float((err > 2e-4).float().mean()):.4f} "
          f"frac>1e-2 {float((err > 1e-2).float().mean()):.4f} PSNR {_psnr(rgb, rgb_ref):.2f} dB; "
          f"z median {float(zerr.median()):.2e} frac>5e-3 {float((zerr > 5e-3).float().mean()):.4f}")
    if dtype in ("f32", "f16x3"):
        assert float((err > 2e-4).float().mean()) <= 0.03 and float(err.median()) < 5e-5
        assert float((zerr > 5e-3).float().mean()) <= 0.05
    else:   # measured (round 2): median 6.8e-4, PSNR 65.7 dB, no ray off by 1e-2, z median 1.2e-3
        assert float(err.median()) < 2e-3 and _psnr(rgb, rgb_ref) > 56.0 and float((err > 1e-2).float().mean()) < 0.005


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16", "f16x3"])
def test_rays_missing_the_sphere_render_nan(gpu_modules, dtype):
    """The reference does not clamp sqrt of a negative discriminant (utils.py:159-217): a ray that misses the DepthNet's
    sphere gets a NaN depth, NaN sample points and -- through sin/cos, nn.Linear and relu -- a NaN pixel.  Every operand
    type reproduces that, for exactly the rays the oracle does it for."""
    from nerf_sampling_amd import ops

    m = gpu_modules("tiny_synth")
    p = m["params"]
    o = torch.tensor([[0.0, 0.0, 4.0], [0.0, 0.0, 4.0], [3.0, 0.0, 4.0], [0.1, -0.2, 4.0]])
    d = torch.tensor([[0.0, 0.0, -1.0], [0.0, 1.0, 0.0], [0.0, 0.0, -1.0], [0.0, 0.05, -1.0]])     # rays 1 and 2 miss r = 2
    v = torch.nn.functional.normalize(d, dim=-1)
    batch = torch.cat([o, d, torch.full((4, 1), 2.0), torch.full((4, 1), 6.0), v], -1)
    ref = O.render_rays_test(batch, p["coarse"], p["fine"], p["depth"], 8, "uniform", 0.1)
    ref_nan = torch.isnan(ref["depth_net_rgb_map"]).any(-1)
    assert ref_nan.tolist() == [False, True, True, False]
    out = ops.render_rays_depthnet(m["depth"].packed(dtype), m["fine"].packed(dtype), rays=(o.cuda(), d.cuda(), v.cuda()),
                                   n_samples=8, mode="uniform", std=0.1, extras=True)
    assert torch.isnan(out["rgb"].cpu()).all(-1).tolist() == ref_nan.tolist()
    assert torch.isnan(out["z"].cpu()).all(-1).tolist() == ref_nan.tolist()
    assert torch.isfinite(out["rgb"].cpu()[~ref_nan]).all()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("ns,mode", [(16, "uniform"), (1, "depth_only")])
def test_standard_configuration_takes_the_one_call_path_bit_exactly(gpu_modules, dtype, ns, mode):
    """render_rays_test with the query function create_nerf builds (tagged standard) runs its DepthNet branch as ONE C
    call per chunk; an untagged query function runs the operator chain.  Same keys, shapes, placement and bits.  With
    nerf_utils._WHOLE_FRAME_BYTES raised, render_test hands the standard configuration whole frames (or the largest chunks
    under the bound) instead of the reference's chunk size (nerf_utils.py:58-85): rays are independent, same bits again."""
    from nerf_sampling_amd import nerf_utils, ops

    ops.set_compute_dtype(dtype)
    m = gpu_modules("tiny_synth")
    tr = make_trainer(n_depth_samples=ns, sampling_mode=mode, distance=0.1)
    kw = render_kwargs(tr, m)
    kw.update(near=2.0, far=6.0, ndc=False)
    H, W = 19, 23
    _, K = O.blender_intrinsics(H, W)
    c2w = O.pose_spherical(-50.0, -30.0, 4.0)[:3, :4]
    chain = nerf_utils.render_test(H, W, K, chunk=150, c2w=c2w, **kw)
    calls = []
    orig = ops.render_rays_depthnet
    ops.render_rays_depthnet = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        kw2 = dict(kw, network_query_fn=nerf_utils.standard_query_fn(lambda i, v, f: kw["network_query_fn"](i, v, f)))
        fused = nerf_utils.render_test(H, W, K, chunk=150, c2w=c2w, **kw2)
    finally:
        ops.render_rays_depthnet = orig
    assert len(calls) == 3                                        # 437 rays in chunks of 150
    assert torch.equal(chain[0], fused[0]) and torch.equal(chain[1].cpu(), fused[1].cpu())
    assert set(chain[2]) == set(fused[2])
    for k in chain[2]:
        assert chain[2][k].shape == fused[2][k].shape and chain[2][k].is_cuda == fused[2][k].is_cuda, k
        assert torch.equal(chain[2][k].cpu(), fused[2][k].cpu()), k
    # whole-frame calls: room for every ray's per-sample outputs -> one call; room for 200 rays -> 200 + 200 + 37
    keep = nerf_utils._WHOLE_FRAME_BYTES
    for bound, n_calls in ((1 << 30, 1), (200 * 20 * ns, 3)):
        calls.clear()
        nerf_utils._WHOLE_FRAME_BYTES = bound
        ops.render_rays_depthnet = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        try:
            again = nerf_utils.render_test(H, W, K, chunk=150, c2w=c2w, **kw2)
        finally:
            ops.render_rays_depthnet = orig
            nerf_utils._WHOLE_FRAME_BYTES = keep
        assert len(calls) == n_calls and torch.equal(again[0], fused[0])
        for k in chain[2]:
            assert torch.equal(chain[2][k].cpu(), again[2][k].cpu()), k


@pytest.mark.parametrize("scene", ["tiny_synth", "lego_synth"])
def test_gaussian_mode_with_injected_noise_vs_oracle(gpu_modules, scene):
    """sampling_mode = "gaussian" (the -e sweep's second mode, utils.py:226-231): the same standard-normal draws injected
    on both sides; one-call path vs the oracle's render_rays_test, fp32 gate."""
    from nerf_sampling_amd import ops

    m = gpu_modules(scene)
    p = m["params"]
    H = W = 24
    _, K = O.blender_intrinsics(H, W)
    c2w = O.pose_spherical(123.0, -30.0, 4.0)[:3, :4]
    batch, o, d, _ = O.ray_batch_from_camera(H, W, K, c2w, 2.0, 6.0)
    noise = torch.randn(H * W, 31, generator=torch.Generator().manual_seed(9))
    ref = O.render_rays_test(batch, p["coarse"], p["fine"], p["depth"], 32, "gaussian", 0.05, noise=noise)
    out = ops.render_rays_depthnet(m["depth"].packed("f32"), m["fine"].packed("f32"),
                                   rays=(o.cuda(), d.cuda(), batch[:, -3:].cuda()), n_samples=32, mode="gaussian", std=0.05,
                                   noise=noise.cuda(), extras=True)
    assert torch.allclose(out["z"].cpu(), ref["depth_net_z_vals"], rtol=0, atol=3e-4)
    bad, err = frac_bad(npy(out["rgb"]), ref["depth_net_rgb_map"].numpy(), 2e-4)
    assert bad <= 0.03 and np.median(err) < 5e-5, (bad, float(np.median(err)))
    bad, err = frac_bad(npy(out["weights"]), ref["depth_net_weights"].numpy(), 2e-4)
    assert bad <= 0.03, bad


def test_render_rays_test_without_view_directions(gpu_modules):
    """The mirrored operator with a use_viewdirs=False field (output_linear head, 5 raw channels as create_nerf builds with
    N_importance > 0, nerf_utils.py:405-406): ray batches are [R, 8] (no view-direction columns, nerf_utils.py:186-187),
    run_network gets viewdirs=None, raw2outputs reads channels 0..3.  Against the oracle's chain on the same rays."""
    from nerf_sampling_amd import nerf_utils, synthetic
    from nerf_sampling_amd.run_nerf_helpers import NeRF, get_embedder

    kw_net = synthetic.NERF_VARIANTS["no_viewdirs_5ch"]
    p_net = synthetic.make_nerf_params(**kw_net)
    net = NeRF(D=kw_net["D"], W=kw_net["W"], input_ch=63, input_ch_views=0, output_ch=5, skips=list(kw_net["skips"]),
               use_viewdirs=False)
    net.load_state_dict(p_net)
    net = net.cuda()
    m = gpu_modules("tiny_synth")
    tr = make_trainer(n_depth_samples=16, sampling_mode="uniform", distance=0.1, use_viewdirs=False)
    embed_fn, _ = get_embedder(tr.multires, tr.i_embed, 3)
    query = lambda inputs, viewdirs, network_fn: tr.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn,  # noqa: E731
                                                                embeddirs_fn=None, netchunk=tr.netchunk)
    H = W = 12
    _, K = O.blender_intrinsics(H, W)
    c2w = O.pose_spherical(55.0, -30.0, 4.0)[:3, :4]
    batch, o, d, _ = O.ray_batch_from_camera(H, W, K, c2w, 2.0, 6.0, use_viewdirs=False)
    assert batch.shape == (H * W, 8)
    res = nerf_utils.render_rays_test(batch.cuda(), network_fn=net, network_query_fn=query, N_samples=64, trainer=tr,
                                      network_fine=None, depth_network=m["depth"], white_bkgd=True, lindisp=True)
    with torch.no_grad():
        mean = O.depthnet_forward(m["params"]["depth"], o, d)
        pts, z = O.place_samples(o, d, mean, 16, "uniform", 0.1)
        raw = O.run_network(p_net, pts, None, skips=kw_net["skips"])
        assert raw.shape[-1] == 5
        exp = O.raw2outputs(raw[..., :4], z, d, 0.0, True)
    ok = np.isfinite(exp[0].numpy()).all(-1)
    assert np.abs(npy(res["depth_net_rgb_map"]) - exp[0].numpy())[ok].max() < 1e-4
    assert np.abs(npy(res["depth_net_weights"]) - exp[6].numpy())[ok].max() < 2e-4
    assert res["depth_net_z_vals"].shape == (H * W, 16) and not res["depth_net_pts"].is_cuda
