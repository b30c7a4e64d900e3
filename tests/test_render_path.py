"""render_path + Blender loader + checkpoint import."""

import json
import os

import numpy as np
import pytest
import torch


def _write_dataset(root, imgs_by_split, poses_by_split, angle=0.6911112070083618):
    from PIL import Image

    for split, imgs in imgs_by_split.items():
        os.makedirs(os.path.join(root, split), exist_ok=True)
        frames = []
        for i, (im, pose) in enumerate(zip(imgs, poses_by_split[split])):
            Image.fromarray(im).save(os.path.join(root, split, f"r_{i}.png"))
            frames.append({"file_path": f"./{split}/r_{i}", "transform_matrix": np.asarray(pose).tolist()})
        with open(os.path.join(root, f"transforms_{split}.json"), "w") as f:
            json.dump({"camera_angle_x": angle, "frames": frames}, f)


def test_blender_loader_cpu(tmp_path):
    """Loader semantics of load_blender.py:46-103: split indices, testskip, RGBA/255, focal, 2x2 half_res."""
    from nerf_sampling_amd.load_blender import load_blender_data

    rng = np.random.default_rng(0)
    mk = lambda n: [rng.integers(0, 256, size=(8, 8, 4), dtype=np.uint8) for _ in range(n)]  # noqa: E731
    eye = np.eye(4, dtype=np.float32)
    imgs = {"train": mk(3), "val": mk(2), "test": mk(4)}
    poses = {k: [eye * (i + 1) for i in range(len(v))] for k, v in imgs.items()}
    _write_dataset(str(tmp_path), imgs, poses)
    im, po, rp, hwf, split = load_blender_data(str(tmp_path), half_res=False, testskip=2)
    assert im.shape == (3 + 1 + 2, 8, 8, 4) and im.dtype == np.float32
    assert [len(s) for s in split] == [3, 1, 2] and split[2][0] == 4
    np.testing.assert_allclose(im[0], imgs["train"][0] / 255.0, atol=1e-7)
    np.testing.assert_allclose(im[5], imgs["test"][2] / 255.0, atol=1e-7)       # testskip=2 -> frames 0, 2
    assert hwf[0] == 8 and abs(hwf[2] - 0.5 * 8 / np.tan(0.5 * 0.6911112070083618)) < 1e-9
    assert rp.shape == (40, 4, 4)
    im2, _, _, hwf2, _ = load_blender_data(str(tmp_path), half_res=True, testskip=2)
    assert im2.shape == (6, 4, 4, 4) and hwf2[:2] == [4, 4] and abs(hwf2[2] - hwf[2] / 2) < 1e-12
    np.testing.assert_allclose(im2[0, 0, 0], im[0, :2, :2].reshape(4, 4).mean(0), atol=1e-6)


@pytest.mark.gpu
def test_render_only_pipeline_with_reference_checkpoints(tmp_path, gpu_modules):
    """yaml-style kwargs -> DepthNetTrainer.train() with render_only: reads a Blender dataset and .tar
    checkpoints in the reference's layout (utils.py:59-89), renders the test poses, writes NNN.png and
    psnr.txt in the reference's format.  Ground truth = the ORACLE's render of the same poses (CPU fp32, stored as the
    dataset's 8-bit PNGs), so the PSNR the pipeline reports is build-vs-oracle: quantisation noise only."""
    from nerf_sampling_amd import ops
    from nerf_sampling_amd.synthetic import blender_intrinsics, pose_spherical
    from nerf_sampling_amd.utils import load_obj_from_config
    from oracle import nerf_oracle as O

    ops.set_compute_dtype("f32")
    m = gpu_modules("tiny_synth")
    P = m["params"]
    H = W = 32
    _, K = blender_intrinsics(H, W)
    data = str(tmp_path / "data"); logs = str(tmp_path / "logs")
    poses = [pose_spherical(a, -30.0, 4.0).numpy() for a in (10.0, 130.0)]
    frames, oracle_rgb = [], []
    for p in poses:
        with torch.no_grad():
            rgb, _disp, _ = O.render_frame(H, W, K, torch.from_numpy(p)[:3, :4], 1024 * 32, 2.0, 6.0, p_coarse=P["coarse"],
                                           p_fine=P["fine"], p_depth=P["depth"], n_depth_samples=16,
                                           sampling_mode="uniform", distance=0.1)
        rgb = np.nan_to_num(rgb.reshape(H, W, 3).numpy(), nan=1.0)
        oracle_rgb.append(rgb)
        frames.append(np.concatenate([(255 * np.clip(rgb, 0, 1)).round().astype(np.uint8),
                                      np.full((H, W, 1), 255, np.uint8)], -1))
    _write_dataset(data, {"train": frames[:1], "val": frames[:1], "test": frames}, {"train": poses[:1], "val": poses[:1], "test": poses})
    os.makedirs(os.path.join(logs, "exp"), exist_ok=True)
    nerf_ckpt = str(tmp_path / "nerf.tar"); dn_ckpt = str(tmp_path / "depthnet.tar")
    adam = lambda mod: torch.optim.Adam(mod.parameters()).state_dict()  # noqa: E731
    both = list(m["coarse"].parameters()) + list(m["fine"].parameters())
    torch.save({"global_step": 200000, "network_fn_state_dict": m["coarse"].state_dict(),
                "network_fine_state_dict": m["fine"].state_dict(),
                "optimizer_state_dict": torch.optim.Adam(both).state_dict()}, nerf_ckpt)
    torch.save({"global_step": 200000, "depth_network": m["depth"].state_dict(),
                "sampling_optimizer_state_dict": adam(m["depth"])}, dn_ckpt)
    cfg = {"module": "nerf_sampling_amd.trainers.DepthNetTrainer",
           "kwargs": dict(dataset_type="blender", basedir=logs, expname="exp", no_batching=True, datadir=data,
                          half_res=False, white_bkgd=True, testskip=1, device="cuda", render_only=True, render_test=True,
                          N_importance=128, N_samples=64, use_viewdirs=True, input_dims_embed=3, netdepth=4, netwidth=128,
                          netdepth_fine=4, netwidth_fine=128, n_layers=3, layer_width=128, sphere_radius=2.0,
                          ft_path=nerf_ckpt, depth_net_path=dn_ckpt, n_depth_samples=16, sampling_mode="uniform",
                          distance=0.1, save_scene_data=True)}
    trainer = load_obj_from_config(cfg)
    psnr = trainer.train()
    out_dir = os.path.join(logs, "exp", "renderonly_test_200000")
    assert sorted(f for f in os.listdir(out_dir) if f.endswith(".png")) == ["000.png", "001.png"]
    lines = open(os.path.join(out_dir, "psnr.txt")).read().splitlines()
    assert lines[0].startswith("000.png, PSNR: ") and lines[2] == "Avg of 2 images:" and lines[3].startswith("PSNR: ")
    assert psnr > 50.0          # vs the oracle's 8-bit frames: quantisation noise only (~59 dB)
    from PIL import Image

    for i, exp in enumerate(oracle_rgb):   # the written PNGs are the oracle's frames up to one 8-bit step
        png = np.asarray(Image.open(os.path.join(out_dir, f"{i:03d}.png")), dtype=np.int32)
        want = (255 * np.clip(exp, 0, 1)).astype(np.uint8).astype(np.int32)           # to8b truncates (run_nerf_helpers.py:11)
        assert np.abs(png - want).max() <= 1 and np.mean(png != want) < 0.02
    sd = torch.load(os.path.join(out_dir, "scene_data.pt"), weights_only=True)
    assert sd["all_pts"].shape == (2 * H * W * 16, 3) and sd["all_weights"].shape == (2 * H * W * 16,)


@pytest.mark.gpu
def test_render_path_frames_equal_per_frame_render_test(gpu_modules):
    """render_path's software pipeline (frame i consumed under frame i+1, pooled pinned buffers, deferred host copies) returns
    per pose exactly what a stand-alone render_test of that pose returns -- rgbs AND disps (the disparity map reaches the host
    through the frame's sink buffer; round 3 snapshotted it before its device-to-host copies had run)."""
    from nerf_sampling_amd import nerf_utils, ops
    from nerf_sampling_amd.run_nerf_helpers import get_embedder
    from nerf_sampling_amd.synthetic import blender_intrinsics, pose_spherical
    from nerf_sampling_amd.trainers import DepthNetTrainer

    ops.set_compute_dtype("f32")
    m = gpu_modules("tiny_synth")
    tr = DepthNetTrainer(dataset_type="blender", basedir="/tmp", expname="rp", no_batching=True, datadir="", half_res=True,
                         white_bkgd=True, N_importance=128, use_viewdirs=True, input_dims_embed=3, n_depth_samples=16,
                         sampling_mode="uniform", distance=0.1)
    e1, _ = get_embedder(10, 0, 3)
    e2, _ = get_embedder(4, 0, 3)
    query = lambda i, v, f: tr.run_network(i, v, f, embed_fn=e1, embeddirs_fn=e2)  # noqa: E731
    H = W = 48
    focal, K = blender_intrinsics(H, W)
    kw = dict(ndc=False, near=2.0, far=6.0, use_viewdirs=True, network_fn=m["coarse"], network_query_fn=query, N_samples=64,
              trainer=tr, network_fine=m["fine"], depth_network=m["depth"], white_bkgd=True, lindisp=True)
    poses = torch.stack([pose_spherical(a, -30.0, 4.0) for a in (0.0, 72.0, 144.0, 216.0)])
    chunk = 1000                                     # several ragged chunks per frame
    rgbs, disps, _ = nerf_utils.render_path(poses, (H, W, focal), K, chunk, kw, step=0)
    assert rgbs.shape == (4, H, W, 3) and disps.shape == (4, H, W)
    for i, c2w in enumerate(poses):
        rgb, disp, _ = nerf_utils.render_test(H, W, K, chunk=chunk, c2w=c2w[:3, :4], **kw)
        np.testing.assert_array_equal(rgbs[i], rgb.cpu().numpy())
        np.testing.assert_array_equal(disps[i], disp.cpu().numpy())
    assert len({d.tobytes() for d in disps}) == 4    # four different frames, not one buffer seen four times


def _cli_root(tmp_path, m, n_test=2):
    """A reference-layout root (dataset/lego, pretrained/nerf|depth_net/lego/...) holding a random 'lego' dataset of 32x32
    files (the yaml has half_res: True) and the modules' weights as 200000.tar checkpoints."""
    from nerf_sampling_amd.synthetic import pose_spherical

    root = str(tmp_path)
    H = W = 16
    rng = np.random.default_rng(1)
    frames = [np.concatenate([rng.integers(0, 256, (2 * H, 2 * W, 3), dtype=np.uint8), np.full((2 * H, 2 * W, 1), 255, np.uint8)], -1)
              for _ in range(n_test)]
    poses = [pose_spherical(a, -30.0, 4.0).numpy() for a in (0.0, 90.0)[:n_test]]
    _write_dataset(os.path.join(root, "dataset", "lego"), {"train": frames[:1], "val": frames[:1], "test": frames},
                   {"train": poses[:1], "val": poses[:1], "test": poses})
    os.makedirs(os.path.join(root, "pretrained", "nerf", "lego"))
    os.makedirs(os.path.join(root, "pretrained", "depth_net", "lego", "files", "sampler_experiment"))
    both = list(m["coarse"].parameters()) + list(m["fine"].parameters())
    torch.save({"global_step": 200000, "network_fn_state_dict": m["coarse"].state_dict(),
                "network_fine_state_dict": m["fine"].state_dict(),
                "optimizer_state_dict": torch.optim.Adam(both).state_dict()},
               os.path.join(root, "pretrained", "nerf", "lego", "200000.tar"))
    torch.save({"global_step": 200000, "depth_network": m["depth"].state_dict(),
                "sampling_optimizer_state_dict": torch.optim.Adam(m["depth"].parameters()).state_dict()},
               os.path.join(root, "pretrained", "depth_net", "lego", "files", "sampler_experiment", "200000.tar"))
    return root


@pytest.mark.gpu
def test_render_cli_counterpart(tmp_path, gpu_modules):
    """`python -m nerf_sampling_amd.experiments.render -d lego -rt [-nf]`: the reference's CLI flow
    (render.py:135-272) on a synthetic 'lego' with production-size networks and reference-layout paths."""
    from click.testing import CliRunner

    from nerf_sampling_amd.experiments.render import main

    root = _cli_root(tmp_path, gpu_modules("lego_synth"))
    try:
        for flags, exp in (([], "lego_depth_net_render_n_samples_2_distance_0.01_sampling_mode_uniform"),
                           (["-nf"], "lego_nerf_full_render")):
            res = CliRunner().invoke(main, ["-d", "lego", "-rt", "--root", root, "--dtype", "f32"] + flags,
                                     catch_exceptions=False)
            assert res.exit_code == 0, res.output
            assert "Final psnr" in res.output
            out_dir = os.path.join(root, "logs", "lego", exp, "renderonly_test_200000")
            assert os.path.exists(os.path.join(out_dir, "000.png")) and os.path.exists(os.path.join(out_dir, "psnr.txt"))
    finally:
        torch.set_default_device("cpu")   # the CLI switches the global default device like the reference does


@pytest.mark.gpu
def test_render_cli_experiments_sweep(tmp_path, gpu_modules):
    """`render -d lego -rt -e`: the reference's automatic sweep (render.py:232-261) run END TO END -- 2 sampling modes x
    n_samples in {2, 32, 64, 128} x distance in {0.1, 0.3, 0.5, 1} = 32 render-only trainers -- and the
    experiments_results.txt it writes, line by line in the reference's format."""
    import re

    from click.testing import CliRunner

    from nerf_sampling_amd.experiments.render import main

    root = _cli_root(tmp_path, gpu_modules("lego_synth"), n_test=1)
    try:
        res = CliRunner().invoke(main, ["-d", "lego", "-rt", "-e", "--root", root, "--dtype", "f32"], catch_exceptions=False)
        assert res.exit_code == 0, res.output
    finally:
        torch.set_default_device("cpu")
    base = os.path.join(root, "logs", "lego", "experiments")
    lines = open(os.path.join(base, "experiments_results.txt")).read().split("\n")
    want = ["Experiments"]
    for mode in ("uniform", "gaussian"):
        want += ([""] if mode == "uniform" else ["", ""]) + [f"Sampling mode: {mode}", ""]    # "\n\nSampling mode: ..\n\n"
        for n in (2, 32, 64, 128):
            want.append(f"N_samples: {n}:")
            want += [f"    Distance: {d}, PSNR: " for d in (0.1, 0.3, 0.5, 1)]
    assert len(lines) == len(want) + 1 and lines[-1] == ""
    psnrs = []
    for got, exp in zip(lines, want):
        if exp.startswith("    Distance"):
            assert got.startswith(exp) and re.fullmatch(r"-?\d+\.\d\d", got[len(exp):]), (got, exp)
            psnrs.append(float(got[len(exp):]))
        else:
            assert got == exp, (got, exp)
    assert len(psnrs) == 32 and all(np.isfinite(psnrs)) and len(set(psnrs)) > 8      # the settings really differ
    for mode, n, d in (("uniform", 2, 0.1), ("gaussian", 128, 1)):
        out_dir = os.path.join(base, mode, f"lego_depth_net_render_n_samples_{n}_distance_{d}_sampling_mode_{mode}",
                               "renderonly_test_200000")
        assert os.path.exists(os.path.join(out_dir, "000.png")) and os.path.exists(os.path.join(out_dir, "psnr.txt"))
