"""DepthNet training step on the HIP backward kernels against torch-CPU autograd
of the oracle: gradients of every DepthNet parameter, the Adam update, and a short optimisation run."""

import copy
import os

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _kwargs(m, **over):
    from test_gpu_render import make_trainer, render_kwargs

    tr = make_trainer(**over)
    return tr, render_kwargs(tr, m)


def test_gemm_strided_variants():
    from nerf_sampling_amd import autograd as A

    g = torch.Generator().manual_seed(0)
    for (M, N, K) in ((1, 1, 1), (5, 3, 7), (64, 64, 16), (100, 257, 319), (1024, 256, 1020)):
        x = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g); b = torch.randn(N, generator=g)
        dy = torch.randn(M, N, generator=g)
        y = A.linear_forward(x.cuda(), W.cuda(), b.cuda(), A.LEAKY).cpu()
        ref = torch.nn.functional.leaky_relu(x @ W.T + b, 0.01)
        assert torch.allclose(y, ref, rtol=1e-4, atol=1e-4 * K ** 0.5)
        assert torch.allclose(A.linear_backward_input(dy.cuda(), W.cuda()).cpu(), dy @ W, rtol=1e-4, atol=1e-4 * N ** 0.5)
        assert torch.allclose(A.linear_backward_input(dy.cuda(), W.cuda(), n_cols=max(1, K // 2)).cpu(),
                              (dy @ W)[:, : max(1, K // 2)], rtol=1e-4, atol=1e-4 * N ** 0.5)
        dW, db = A.linear_backward_weight(dy.cuda(), x.cuda())
        assert torch.allclose(dW.cpu(), dy.T @ x, rtol=1e-4, atol=1e-4 * M ** 0.5)
        assert torch.allclose(db.cpu(), dy.sum(0), rtol=1e-4, atol=1e-4 * M ** 0.5)


def test_posenc_and_points_backward():
    from nerf_sampling_amd import autograd as A

    x = (torch.rand(50, 3) * 4 - 2).requires_grad_(True)
    de = torch.randn(50, 63)
    (O.posenc(x, 10) * de).sum().backward()
    mine = A.posenc_backward(x.detach().cuda(), de.cuda(), 10).cpu()
    assert torch.allclose(mine, x.grad, rtol=1e-4, atol=1e-2)       # terms up to 2^9 * |de|
    o = torch.randn(20, 3); d = torch.randn(20, 3); z = torch.rand(20, 4, requires_grad=True)
    zc = z.detach().cuda().requires_grad_(True)
    dp = torch.randn(20, 4, 3)
    (A.points_along_rays(o.cuda(), d.cuda(), zc) * dp.cuda()).sum().backward()
    ((o[:, None] + d[:, None] * z[..., None]) * dp).sum().backward()
    assert torch.allclose(zc.grad.cpu(), z.grad, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("scene", ["tiny_synth", "lego_synth"])
def test_depthnet_training_gradients_match_oracle_autograd(golden, gpu_modules, scene):
    from nerf_sampling_amd import nerf_utils, ops

    ops.set_compute_dtype("f32")
    m = dict(gpu_modules(scene))
    m["depth"] = copy.deepcopy(m["depth"])
    for p in m["depth"].parameters():
        p.requires_grad_(True)
    tr, kw = _kwargs(m)
    rb = T(golden("render_rays_train")["ray_batch"])[:48]
    target = torch.rand(48, 3, generator=torch.Generator().manual_seed(4))
    res = nerf_utils.render_rays(rb.cuda(), **kw)
    assert res["depth_net_rgb_map"].requires_grad and res["depth_net_z_vals"].requires_grad
    loss = ((res["depth_net_rgb_map"] - target.cuda()) ** 2).mean() + torch.nn.functional.mse_loss(
        res["depth_net_z_vals"], res["max_z_vals"])
    loss.backward()
    # oracle: same losses through torch-CPU autograd, with the SAME (constant) regression target max_z
    p = {k: v.clone().requires_grad_(True) for k, v in m["params"]["depth"].items()}
    o, d, view = rb[:, 0:3], rb[:, 3:6], rb[:, 8:11]
    z = O.depthnet_forward(p, o, d)
    raw = O.run_network(m["params"]["fine"], o[:, None] + d[:, None] * z[..., None], view)
    rgb = torch.sigmoid(raw[:, 0, :3])               # single-sample compositing (see ns_raw2outputs N == 1)
    loss_o = ((rgb - target) ** 2).mean() + torch.nn.functional.mse_loss(z, res["max_z_vals"].detach().cpu())
    loss_o.backward()
    assert abs(float(loss) - float(loss_o)) < 1e-4 * max(1.0, abs(float(loss_o)))
    worst = 0.0
    for name, mod_p in m["depth"].named_parameters():
        g, go = mod_p.grad.cpu(), p[name].grad
        assert g is not None and go is not None, name
        rel = float((g - go).norm() / (go.norm() + 1e-12))
        worst = max(worst, rel)
        assert rel < 2e-3, (name, rel, float(go.norm()))
    print(f"{scene}: worst relative gradient error over {len(p)} tensors = {worst:.2e}")


def test_hip_adam_matches_torch_adam():
    from nerf_sampling_amd.autograd import HipAdam

    g = torch.Generator().manual_seed(2)
    w0 = torch.randn(300, 17, generator=g)
    a = w0.clone().cuda().requires_grad_(True); b = w0.clone().requires_grad_(True)
    oa, ob = HipAdam([a], lr=1e-3), torch.optim.Adam([b], lr=1e-3)
    for _ in range(5):
        grad = torch.randn(300, 17, generator=g)
        a.grad, b.grad = grad.cuda(), grad.clone()
        oa.step(); ob.step()
    assert torch.allclose(a.detach().cpu(), b.detach(), rtol=1e-5, atol=1e-6)
    sd = oa.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}      # torch's layout: checkpoints round-trip
    # the device-resident step counter / learning rate (what a captured hipGraph of the step replays): same updates
    oa.use_device_step()
    for k in range(4):
        if k == 2:
            oa.param_groups[0]["lr"] = ob.param_groups[0]["lr"] = 3e-4
        grad = torch.randn(300, 17, generator=g)
        a.grad, b.grad = grad.cuda(), grad.clone()
        oa.step(); ob.step()
    assert torch.allclose(a.detach().cpu(), b.detach(), rtol=1e-5, atol=1e-6)
    assert float(oa.state_dict()["state"][0]["step"]) == float(ob.state_dict()["state"][0]["step"]) == 9.0


def test_core_optimization_loop_reduces_loss(gpu_modules):
    """A few DepthNet updates on one fixed ray batch (Trainer.core_optimization_loop): the depth regression loss
    towards the frozen NeRF's max-weight sample goes down and only DepthNet weights move."""
    from nerf_sampling_amd import ops
    from nerf_sampling_amd.autograd import HipAdam

    ops.set_compute_dtype("f32")
    m = dict(gpu_modules("tiny_synth"))
    m["depth"] = copy.deepcopy(m["depth"])
    for p in m["depth"].parameters():
        p.requires_grad_(True)
    tr, kw = _kwargs(m)
    H = W = 24
    _, K = O.blender_intrinsics(H, W)
    tr.H, tr.W, tr.K = H, W, K
    o, d, _ = ops.get_rays(H, W, K, O.pose_spherical(20.0, -30.0, 4.0)[:3, :4])
    batch_rays = torch.stack([o[100:356], d[100:356]], 0)
    target = torch.rand(256, 3, generator=torch.Generator().manual_seed(7)).cuda()
    opt = HipAdam(list(m["depth"].parameters()), lr=1e-3)
    kw.update(near=2.0, far=6.0, ndc=False)
    fine_before = [p.clone() for p in m["fine"].parameters()]
    losses = []
    for i in range(8):
        loss, dn_loss, psnr, _ = tr.core_optimization_loop(opt, kw, batch_rays, i, target)
        losses.append(float(dn_loss))
    print("depth_net_loss:", [round(x, 5) for x in losses])
    assert losses[-1] < losses[0]
    assert all(torch.equal(a, b) for a, b in zip(fine_before, m["fine"].parameters()))


def test_train_loop_end_to_end(tmp_path, gpu_modules):
    """DepthNetTrainer.train(): Blender dataset -> random ray batches -> DepthNet updates -> checkpoints in the
    reference's layout (utils.py:59-89) that reload into a fresh trainer."""
    from test_render_path import _write_dataset

    from nerf_sampling_amd import ops
    from nerf_sampling_amd.trainers import DepthNetTrainer

    ops.set_compute_dtype("f32")
    m = gpu_modules("tiny_synth")
    rng = np.random.default_rng(3)
    H = W = 20
    frames = [np.concatenate([rng.integers(0, 256, (H, W, 3), dtype=np.uint8), np.full((H, W, 1), 255, np.uint8)], -1)
              for _ in range(3)]
    poses = [O.pose_spherical(a, -30.0, 4.0).numpy() for a in (0.0, 120.0, 240.0)]
    data, logs = str(tmp_path / "data"), str(tmp_path / "logs")
    _write_dataset(data, {"train": frames, "val": frames[:1], "test": frames[:1]},
                   {"train": poses, "val": poses[:1], "test": poses[:1]})
    nerf_ckpt = str(tmp_path / "nerf.tar")
    both = list(m["coarse"].parameters()) + list(m["fine"].parameters())
    torch.save({"global_step": 0, "network_fn_state_dict": m["coarse"].state_dict(),
                "network_fine_state_dict": m["fine"].state_dict(),
                "optimizer_state_dict": torch.optim.Adam(both).state_dict()}, nerf_ckpt)
    kw = dict(dataset_type="blender", basedir=logs, expname="exp", no_batching=True, datadir=data, half_res=False,
              white_bkgd=True, testskip=1, device="cuda", N_rand=128, N_importance=128, N_samples=64, use_viewdirs=True,
              input_dims_embed=3, netdepth=4, netwidth=128, netdepth_fine=4, netwidth_fine=128, n_layers=3,
              layer_width=128, sphere_radius=2.0, ft_path=nerf_ckpt, depth_net_lr=1e-3, train_depth_net_only=True,
              i_weights=3, i_print=2, perturb=0.0)
    np.random.seed(0); torch.manual_seed(0)
    tr = DepthNetTrainer(**kw)
    psnr = tr.train(N_iters=7)                   # iterations 1..6, checkpoints at 3 and 6
    assert psnr is not None and np.isfinite(float(psnr))
    ckpts = sorted(f for f in os.listdir(os.path.join(logs, "exp")) if f.endswith(".tar"))
    assert ckpts == ["000003.tar", "000006.tar"]
    lines = open(os.path.join(logs, "exp", "psnr.txt")).read().splitlines()      # i_print = 2: iterations 2, 4, 6
    assert len(lines) == 3 and lines[0].startswith("Iter: 2 Loss: ") and ", Depth Net Loss: " in lines[0] and ", PSNR: " in lines[0]
    ck = torch.load(os.path.join(logs, "exp", "000006.tar"), weights_only=True)
    assert set(ck) == {"global_step", "network_fn_state_dict", "network_fine_state_dict", "optimizer_state_dict",
                       "sampling_optimizer_state_dict", "depth_network"}
    assert "origin_layers.0.weight" in ck["depth_network"] and "cat_layers.4.bias" in ck["depth_network"]
    # NeRF stayed frozen, DepthNet moved
    assert torch.equal(ck["network_fine_state_dict"]["pts_linears.0.weight"].cpu(), m["fine"].pts_linears[0].weight.cpu())
    # resume: a new trainer on the same expdir reloads the newest DepthNet checkpoint and its step
    tr2 = DepthNetTrainer(**kw)
    tr2.create_nerf_model()
    assert tr2.start == ck["global_step"]


def test_run_cli_counterpart(tmp_path, gpu_modules):
    """`python -m nerf_sampling_amd.experiments.run -d lego --iters 2`: production-size DepthNet (10x256, lr 1e-4)
    trained for two steps against the frozen synthetic NeRF, reference directory layout."""
    from click.testing import CliRunner
    from test_render_path import _write_dataset

    from nerf_sampling_amd.experiments.run import main

    m = gpu_modules("lego_synth")
    root = str(tmp_path)
    rng = np.random.default_rng(5)
    frames = [np.concatenate([rng.integers(0, 256, (64, 64, 3), dtype=np.uint8), np.full((64, 64, 1), 255, np.uint8)], -1)
              for _ in range(2)]
    poses = [O.pose_spherical(a, -30.0, 4.0).numpy() for a in (0.0, 90.0)]
    _write_dataset(os.path.join(root, "dataset", "lego"), {"train": frames, "val": frames[:1], "test": frames[:1]},
                   {"train": poses, "val": poses[:1], "test": poses[:1]})
    os.makedirs(os.path.join(root, "pretrained", "nerf", "lego"))
    both = list(m["coarse"].parameters()) + list(m["fine"].parameters())
    torch.save({"global_step": 0, "network_fn_state_dict": m["coarse"].state_dict(),
                "network_fine_state_dict": m["fine"].state_dict(),
                "optimizer_state_dict": torch.optim.Adam(both).state_dict()},
               os.path.join(root, "pretrained", "nerf", "lego", "200000.tar"))
    try:
        res = CliRunner().invoke(main, ["-d", "lego", "--root", root, "--iters", "2", "-ip", "1"], catch_exceptions=False)
        assert res.exit_code == 0, res.output
        assert "[TRAIN] Iter: 2" in res.output
    finally:
        torch.set_default_device("cpu")


def test_use_batching_and_ray_dump(tmp_path, gpu_modules):
    """The use_batching branch of the batch sampler (Trainer.py:232-269, 400-414): rays of all training images shuffled
    once (numpy's generator, as the reference), N_rand rows per step, reshuffle after an epoch; and the safetensors ray
    dump (sampling_trainer.py:124-138) round-tripped through safetensors' own loader."""
    from safetensors.torch import load_file
    from test_render_path import _write_dataset

    from nerf_sampling_amd import ops
    from nerf_sampling_amd.trainers import DepthNetTrainer

    ops.set_compute_dtype("f32")
    m = gpu_modules("tiny_synth")
    rng = np.random.default_rng(4)
    H = W = 12
    frames = [np.concatenate([rng.integers(0, 256, (H, W, 3), dtype=np.uint8), np.full((H, W, 1), 255, np.uint8)], -1)
              for _ in range(3)]
    poses = [O.pose_spherical(a, -30.0, 4.0).numpy() for a in (10.0, 130.0, 250.0)]
    data, logs = str(tmp_path / "data"), str(tmp_path / "logs")
    _write_dataset(data, {"train": frames, "val": frames[:1], "test": frames[:1]},
                   {"train": poses, "val": poses[:1], "test": poses[:1]})
    nerf_ckpt = str(tmp_path / "nerf.tar")
    both = list(m["coarse"].parameters()) + list(m["fine"].parameters())
    torch.save({"global_step": 0, "network_fn_state_dict": m["coarse"].state_dict(),
                "network_fine_state_dict": m["fine"].state_dict(),
                "optimizer_state_dict": torch.optim.Adam(both).state_dict()}, nerf_ckpt)
    kw = dict(dataset_type="blender", basedir=logs, expname="exp", no_batching=False, datadir=data, half_res=False,
              white_bkgd=True, testskip=1, device="cuda", N_rand=100, N_importance=128, N_samples=64, use_viewdirs=True,
              input_dims_embed=3, netdepth=4, netwidth=128, netdepth_fine=4, netwidth_fine=128, n_layers=3,
              layer_width=128, sphere_radius=2.0, ft_path=nerf_ckpt, depth_net_lr=1e-3, train_depth_net_only=True,
              i_weights=1000, i_print=1000, perturb=0.0)
    tr = DepthNetTrainer(**kw)
    assert tr.use_batching
    hwf, ps, i_test, i_val, i_train, images, _ = tr.load_data()
    tr.cast_intrinsics_to_right_types(hwf)
    np.random.seed(11)
    imgs_t, poses_t, rays_rgb, i_batch = tr.prepare_raybatch_tensor_if_batching_random_rays(ps, images, i_train)
    n = len(i_train) * H * W
    assert rays_rgb.shape == (n, 3, 3) and i_batch == 0 and rays_rgb.is_cuda
    # the same permutation the reference's np.random.shuffle(rays_rgb) draws, applied to rays built by the oracle
    np.random.seed(11)
    perm = np.arange(n); np.random.shuffle(perm)
    rows = []
    for i in i_train:
        o, d = O.camera_rays(H, W, tr.K, torch.tensor(np.asarray(ps[i]), dtype=torch.float32)[:3, :4])
        rows.append(torch.stack([o.reshape(-1, 3), d.reshape(-1, 3),
                                 torch.tensor(np.asarray(images[i]), dtype=torch.float32).reshape(-1, images[i].shape[-1])[:, :3]], 1))
    exp = torch.cat(rows, 0)[torch.from_numpy(perm)]
    assert torch.equal(rays_rgb[..., :].cpu()[:, 0], exp[:, 0])              # origins: bit exact
    assert torch.allclose(rays_rgb.cpu(), exp, rtol=0, atol=2e-6)
    seen = 0
    for step in range(6):                                                     # 432 rays / 100 per step: wraps in step 5
        rays_rgb, i_batch, batch_rays, target = tr.sample_random_ray_batch(rays_rgb, i_batch, i_train, imgs_t, poses_t, step)
        assert batch_rays.shape[0] == 2 and batch_rays.shape[2] == 3 and target.shape == (batch_rays.shape[1], 3)
        seen += batch_rays.shape[1]
        if step < 4:
            assert torch.equal(batch_rays[0].cpu(), exp[step * 100 : (step + 1) * 100, 0])
    assert i_batch == 100 and seen == 100 * 4 + 32 + 100                      # reshuffled after the epoch, then continues
    # the training loop itself runs on this branch
    np.random.seed(0); torch.manual_seed(0)
    psnr = DepthNetTrainer(**kw).train(N_iters=4)
    assert psnr is not None and np.isfinite(float(psnr))
    # ray dump
    tr.global_step = 7
    o = torch.randn(5, 3).cuda(); pts = torch.randn(5, 4, 3).cuda(); alpha = torch.rand(5, 4).cuda()
    path = tr.save_rays_data(o, pts[:, ::2], alpha)                            # a non-contiguous view, as callers may pass
    assert path.endswith(os.path.join("exp", "exp_7.safetensors"))
    back = load_file(path)
    assert set(back) == {"origins", "pts", "alpha"}
    assert torch.equal(back["origins"], o.cpu()) and torch.equal(back["pts"], pts[:, ::2].cpu()) and torch.equal(back["alpha"], alpha.cpu())


def test_graphed_step_equals_eager_step(gpu_modules):
    """trainers.GraphedDepthNetStep (forward + backward + Adam as ONE hipGraph replay per step) against
    core_optimization_loop on the same batches: identical losses and bit-identical DepthNet weights and Adam state after
    eight steps -- two eager warm-up steps, the capture, four replays, then a batch of ANOTHER size (the eager fallback after
    a capture: its update goes through the optimizer's eager row table, the graph's gradient pool is left alone) and a
    replay again -- and a step counter that checkpoints like torch's.  A second capture from the same optimizer is refused."""
    from nerf_sampling_amd import ops
    from nerf_sampling_amd.autograd import HipAdam

    ops.set_compute_dtype("f32")
    base = dict(gpu_modules("tiny_synth"))
    H = W = 24
    _, K = O.blender_intrinsics(H, W)
    o, d, _ = ops.get_rays(H, W, K, O.pose_spherical(20.0, -30.0, 4.0)[:3, :4])
    g = torch.Generator().manual_seed(3)
    batches = [(torch.randint(0, H * W, (n,), generator=g).cuda(), torch.rand(n, 3, generator=g).cuda())
               for n in (128, 128, 128, 128, 128, 128, 96, 128)]
    results = {}
    for mode in ("eager", "graph"):
        m = dict(base)
        m["depth"] = copy.deepcopy(base["depth"])
        for p in m["depth"].parameters():
            p.requires_grad_(True)
        tr, kw = _kwargs(m)
        tr.H, tr.W, tr.K = H, W, K
        kw.update(near=2.0, far=6.0, ndc=False)
        opt = HipAdam(list(m["depth"].parameters()), lr=1e-3)
        opt.use_device_step()          # both runs on ns_adam_step_dev (bias corrections evaluated on the device)
        step = (tr.graphed_optimization_loop(opt, kw) if mode == "graph"
                else (lambda rays, i, tgt: tr.core_optimization_loop(opt, kw, rays, i, tgt)))
        losses = []
        for i, (idx, tgt) in enumerate(batches):
            if i == 4:
                opt.param_groups[0]["lr"] = 5e-4           # a learning-rate change reaches the captured update
            loss, dn_loss, psnr, _ = step(torch.stack([o[idx], d[idx]], 0), i, tgt)
            losses.append((float(loss), float(dn_loss), float(psnr)))
        if mode == "graph":
            assert step.graph is not None and step.calls == 8
            with pytest.raises(RuntimeError, match="one hipGraph capture per optimizer"):
                again = tr.graphed_optimization_loop(opt, kw)
                again.warmup = 0
                again(torch.stack([o[batches[0][0]], d[batches[0][0]]], 0), 0, batches[0][1])
        sd = opt.state_dict()
        results[mode] = (losses, [p.detach().clone() for p in m["depth"].parameters()], sd)
    le, pe, sde = results["eager"]
    lg, pg, sdg = results["graph"]
    assert le == lg, (le, lg)
    assert all(torch.equal(a, b) for a, b in zip(pe, pg))
    for k in sde["state"]:
        assert float(sde["state"][k]["step"]) == float(sdg["state"][k]["step"]) == 8.0
        assert torch.equal(sde["state"][k]["exp_avg_sq"], sdg["state"][k]["exp_avg_sq"])
    assert le[-1][1] < le[0][1]                            # and it trains


def test_hip_adam_reloads_its_state_in_device_step_mode(gpu_modules):
    """HipAdam.load_state_dict AFTER use_device_step(): the device step counter (bias corrections), the host count and the
    learning-rate scalar follow the loaded state, whether the checkpoint stores 'step' as a tensor (torch >= 1.12) or as a
    plain int (older checkpoints); the next update equals torch.optim.Adam's from the same state."""
    from nerf_sampling_amd.autograd import HipAdam

    torch.manual_seed(0)
    for step_as_int in (False, True):
        w = torch.nn.Parameter(torch.randn(300, device="cuda"))
        ref_w = torch.nn.Parameter(w.detach().clone())
        ref = torch.optim.Adam([ref_w], lr=2e-3)
        grads = [torch.randn(300, device="cuda") for _ in range(4)]
        for gr in grads[:3]:                               # three steps of torch's Adam make the state to load
            ref_w.grad = gr.clone()
            ref.step()
        sd = copy.deepcopy(ref.state_dict())               # (torch's loader keeps same-device state tensors by reference)
        if step_as_int:
            for st in sd["state"].values():
                st["step"] = int(st["step"])
        opt = HipAdam([w], lr=1e-3)
        opt.use_device_step()                              # device mode first (trainers.GraphedDepthNetStep does this) ...
        with torch.no_grad():
            w.copy_(ref_w)
        opt.load_state_dict(sd)                            # ... the checkpoint afterwards
        assert opt._host_steps == 3 and int(opt._dev_step.item()) == 3 and abs(float(opt._dev_lr.item()) - 2e-3) < 1e-9
        w.grad, ref_w.grad = grads[3].clone(), grads[3].clone()
        opt.step(); ref.step()
        assert torch.allclose(w, ref_w, rtol=0, atol=1e-6), float((w - ref_w).abs().max())       # (an update is ~2e-3)
        assert float(opt.state_dict()["state"][0]["step"]) == 4.0


@pytest.mark.parametrize("use_viewdirs,skips", [(True, [4]), (False, [2]), (False, [])])
def test_nerf_input_gradient_both_heads(use_viewdirs, skips):
    """autograd.NerfInputGrad (the frozen field's gradient w.r.t. its input points, Trainer.py:506-544 via nerf_utils.py:692-715)
    for both heads of the reference's NeRF -- alpha / feature / views / rgb with view directions, output_linear without
    (run_nerf_helpers.py:119-133) -- against torch-CPU autograd of the oracle's forward on the same weights."""
    from nerf_sampling_amd import ops
    from nerf_sampling_amd.autograd import NerfInputGrad
    from nerf_sampling_amd.run_nerf_helpers import NeRF

    ops.set_compute_dtype("f32")
    torch.manual_seed(3)
    D, W = 6, 128
    net = NeRF(D=D, W=W, input_ch=63, input_ch_views=27 if use_viewdirs else 0, output_ch=5, skips=skips, use_viewdirs=use_viewdirs)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(1.5)
    params = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.cuda()
    for p in net.parameters():
        p.requires_grad_(False)
    R, N = 37, 5
    pts = ((torch.rand(R, N, 3) * 2 - 1) * 1.5)
    view = torch.nn.functional.normalize(torch.randn(R, 3), dim=-1)
    C = 4 if use_viewdirs else 5
    gout = torch.randn(R, N, C)
    # reference: plain torch on the CPU
    p_cpu = pts.clone().requires_grad_(True)
    x = O.posenc(p_cpu.reshape(-1, 3), 10)
    h = x
    for i in range(D):
        h = torch.relu(torch.nn.functional.linear(h, params[f"pts_linears.{i}.weight"], params[f"pts_linears.{i}.bias"]))
        if i in skips:
            h = torch.cat([x, h], -1)
    if use_viewdirs:
        v = O.posenc(view[:, None].expand(R, N, 3).reshape(-1, 3), 4)
        sigma = torch.nn.functional.linear(h, params["alpha_linear.weight"], params["alpha_linear.bias"])
        feat = torch.nn.functional.linear(h, params["feature_linear.weight"], params["feature_linear.bias"])
        hv = torch.relu(torch.nn.functional.linear(torch.cat([feat, v], -1), params["views_linears.0.weight"], params["views_linears.0.bias"]))
        raw_ref = torch.cat([torch.nn.functional.linear(hv, params["rgb_linear.weight"], params["rgb_linear.bias"]), sigma], -1)
    else:
        raw_ref = torch.nn.functional.linear(h, params["output_linear.weight"], params["output_linear.bias"])
    (raw_ref.reshape(R, N, C) * gout).sum().backward()
    # build
    p_dev = pts.cuda().requires_grad_(True)
    raw = NerfInputGrad.apply(p_dev, view.cuda() if use_viewdirs else None, net)
    assert raw.shape == (R, N, C)
    scale = raw_ref.abs().max()
    assert float((raw.detach().cpu().reshape(-1, C) - raw_ref.detach()).abs().max() / scale) < 2e-5
    (raw * gout.cuda()).sum().backward()
    gs = p_cpu.grad.abs().max()
    assert float((p_dev.grad.cpu() - p_cpu.grad).abs().max() / gs) < 2e-4, float((p_dev.grad.cpu() - p_cpu.grad).abs().max() / gs)
