#!/usr/bin/env python3
"""Fit the radiance field and the DepthNet of tests/golden/fitted_scene to the analytic ground-truth scene
(nerf_sampling_amd/analytic_scene.py).  Runs on the GPU box:

    python tools/fit_scene.py --phase nerf     --out gpurun_out/fit --seconds 420
    python tools/fit_scene.py --phase depthnet --out gpurun_out/fit --seconds 420
    python tools/fit_scene.py --phase eval     --out gpurun_out/fit

No dataset or checkpoint ships with the reference, so the scene-PSNR half of BASELINE.json's metric ("PSNR within 0.05 dB
of reference") needs a FITTED scene: networks whose density is well below zero in empty space and large inside objects,
as trained fields are, instead of seeded random weights.

* phase nerf: a NeRF 8x256 (the reference's architecture, run_nerf_helpers.py:67-134, same state-dict keys) trained with
  plain torch autograd on rays of random cameras at the reference's camera distance (load_blender.py:84-90: radius 4),
  volume rendering exactly as raw2outputs (sampling_trainer.py:153-230: last distance 1e10, white background), with the
  reference's density-noise regulariser raw_noise_std = 1 (its hotdog / materials configs).  Sample depths: 64 stratified
  in [near, far] plus 64 around the analytic hit depth (a stand-in for the reference's importance pass that needs no
  second network).  ONE network is used as both network_fn and network_fine afterwards.
* phase depthnet: the production DepthNet (10x256, experiments/run.py:101-109) trained with THIS repo's training step
  (trainers.Trainer.core_optimization_loop on the HIP backward kernels): targets are the frozen field's max-weight sample
  of the 64 + 128 vanilla pass, as in the reference (nerf_utils.py:675-690, Trainer.py:506-544).
* phase eval: prints PSNR against the analytic ground truth (fp32 HIP render) and how often the window around the
  predicted depth holds the surface.

Weights are written as safetensors (fp32); copy them to tests/golden/fitted_scene/ to commit.
"""

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from nerf_sampling_amd import analytic_scene, synthetic  # noqa: E402

NEAR, FAR = 2.0, 6.0


def random_pose(rng: np.random.Generator) -> torch.Tensor:
    """A camera on the reference's shell (radius 4) looking at the origin; elevations around its render path's 30 deg."""
    return synthetic.pose_spherical(float(rng.uniform(-180.0, 180.0)), float(rng.uniform(-55.0, -8.0)), 4.0)[:3, :4]


def camera_rays(H, W, K, c2w, idx, device):
    """o, d [n,3] of the pixels ``idx`` (flat, row-major) -- run_nerf_helpers.py:187-202 in fp32 torch."""
    c2w = c2w.to(device)
    jj, ii = (idx // W).float(), (idx % W).float()
    cam = torch.stack([(ii - float(K[0][2])) / float(K[0][0]), -(jj - float(K[1][2])) / float(K[1][1]), -torch.ones_like(ii)], -1)
    d = (cam[:, None, :] * c2w[:3, :3]).sum(-1)
    return c2w[:3, 3].expand(d.shape).contiguous(), d


def posenc(x, n_freqs):
    out = [x]
    for k in range(n_freqs):
        out += [torch.sin(x * 2.0 ** k), torch.cos(x * 2.0 ** k)]
    return torch.cat(out, -1)


class TorchNeRF(nn.Module):
    """The reference's NeRF (use_viewdirs=True), plain torch: trainable twin of nerf_sampling_amd.run_nerf_helpers.NeRF."""

    def __init__(self, D=8, W=256, skips=(4,)):
        super().__init__()
        self.skips = skips
        self.pts_linears = nn.ModuleList([nn.Linear(63, W)] + [nn.Linear(W + 63 if i in skips else W, W) for i in range(D - 1)])
        self.views_linears = nn.ModuleList([nn.Linear(27 + W, W // 2)])
        self.feature_linear, self.alpha_linear, self.rgb_linear = nn.Linear(W, W), nn.Linear(W, 1), nn.Linear(W // 2, 3)

    def forward(self, pts, viewdirs):
        x = posenc(pts, 10)
        v = posenc(viewdirs, 4)
        h = x
        for i, lin in enumerate(self.pts_linears):
            h = torch.relu(lin(h))
            if i in self.skips:
                h = torch.cat([x, h], -1)
        sigma = self.alpha_linear(h)
        h = torch.relu(self.views_linears[0](torch.cat([self.feature_linear(h), v], -1)))
        return torch.cat([self.rgb_linear(h), sigma], -1)


def composite(raw, z, d, noise_std):
    """raw2outputs with white background (sampling_trainer.py:153-230)."""
    dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], 1e10)], -1) * d.norm(dim=-1, keepdim=True)
    sigma = raw[..., 3]
    if noise_std > 0:
        sigma = sigma + torch.randn_like(sigma) * noise_std
    alpha = 1.0 - torch.exp(-torch.relu(sigma) * dists)
    trans = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-10], -1), -1)[:, :-1]
    w = alpha * trans
    rgb = (w[..., None] * torch.sigmoid(raw[..., :3])).sum(1)
    return rgb + (1.0 - w.sum(-1, keepdim=True)), w


def fit_nerf(args):
    dev = torch.device(args.device)
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    net = TorchNeRF().to(dev)
    lr0 = args.nerf_lr
    if args.resume and os.path.exists(os.path.join(args.init, "nerf.safetensors")):
        net.load_state_dict(load(os.path.join(args.init, "nerf.safetensors")))
        print("[nerf] resuming from", args.init, flush=True)
    opt = torch.optim.Adam(net.parameters(), lr=lr0)
    H = W = 800
    _, K = synthetic.blender_intrinsics(H, W)
    B, n_pose = args.rays, 8
    t_start, it, planned = time.time(), 0, None
    log = []
    while True:
        os_, ds_ = [], []
        for _ in range(n_pose):
            idx = torch.randint(0, H * W, (B // n_pose,), device=dev)
            o, d = camera_rays(H, W, K, random_pose(rng), idx, dev)
            os_.append(o), ds_.append(d)
        o, d = torch.cat(os_), torch.cat(ds_)
        with torch.no_grad():
            gt, t_hit, hit = analytic_scene.raycast(o, d)
            # 64 stratified depths + 64 around the analytic hit depth (uniform where the ray misses)
            edges = torch.linspace(NEAR, FAR, 65, device=dev)
            z_s = edges[:-1] + (edges[1:] - edges[:-1]) * torch.rand(B, 64, device=dev)
            centre = torch.where(hit, t_hit, torch.zeros_like(t_hit))[:, None]
            spread = torch.cat([torch.full((B, 32), 0.03, device=dev), torch.full((B, 32), 0.18, device=dev)], -1)
            z_g = centre + spread * torch.randn(B, 64, device=dev)
            z_g = torch.where(hit[:, None], z_g, NEAR + (FAR - NEAR) * torch.rand(B, 64, device=dev))
            z = torch.sort(torch.cat([z_s, z_g.clamp(NEAR, FAR)], -1), -1).values
            pts = o[:, None] + d[:, None] * z[..., None]
            view = (d / d.norm(dim=-1, keepdim=True))[:, None].expand(pts.shape)
        raw = net(pts.reshape(-1, 3), view.reshape(-1, 3)).reshape(B, -1, 4)
        rgb, _ = composite(raw, z, d, args.raw_noise_std)
        loss = ((rgb - gt) ** 2).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        it += 1
        if it in (30, 60):
            if dev.type == "cuda":
                torch.cuda.synchronize()
            if it == 30:
                t30 = time.time()
            else:       # steady-state rate (the first iterations pay library start-up)
                planned = max(200, int((args.seconds - (time.time() - t_start)) / ((time.time() - t30) / 30)) + 60)
                print(f"[nerf] {1e3 * (time.time() - t30) / 30:.1f} ms/iter -> planning {planned} iterations", flush=True)
        if planned:
            for g in opt.param_groups:      # exponential decay 5e-4 -> 2.5e-5 over the run (the reference decays 10x)
                g["lr"] = lr0 * (0.05 ** min(1.0, it / planned))
        if it % 250 == 0:
            psnr = -10 * math.log10(float(loss.detach()))
            log.append((it, psnr))
            print(f"[nerf] it {it} loss {float(loss.detach()):.5f} psnr {psnr:.2f} dB  ({time.time() - t_start:.0f} s)", flush=True)
        if it % 2000 == 0 or (planned and it >= planned):
            save(net.state_dict(), os.path.join(args.out, "nerf.safetensors"))
        if planned and it >= planned:
            break
    json.dump({"iterations": it, "rays_per_iteration": B, "log": log, "raw_noise_std": args.raw_noise_std,
               "resumed": bool(args.resume), "lr0": lr0},
              open(os.path.join(args.out, "nerf_fit.json"), "w"))


def save(state, path):
    from safetensors.torch import save_file

    os.makedirs(os.path.dirname(path), exist_ok=True)
    save_file({k: v.detach().float().cpu().contiguous() for k, v in state.items()}, path + ".tmp")
    os.replace(path + ".tmp", path)


def load(path):
    from safetensors.torch import load_file

    return load_file(path)


def hip_modules(args, depth_state=None):
    from nerf_sampling_amd.depth_net import DepthNet
    from nerf_sampling_amd.run_nerf_helpers import NeRF

    nerf = NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
    nerf.load_state_dict(load(os.path.join(args.out, "nerf.safetensors")))
    nerf = nerf.cuda()
    for p in nerf.parameters():
        p.requires_grad_(False)
    dn = DepthNet(hidden_sizes=[256] * 10, cat_hidden_sizes=[256] * 10, sphere_radius=2.0)
    if depth_state is not None:
        dn.load_state_dict(depth_state)
    return nerf, dn.cuda()


def fit_depthnet(args):
    from nerf_sampling_amd import ops
    from nerf_sampling_amd.autograd import HipAdam
    from nerf_sampling_amd.run_nerf_helpers import get_embedder
    from nerf_sampling_amd.trainers import DepthNetTrainer

    ops.set_compute_dtype("f32")            # the frozen field's target pass in exact fp32
    torch.manual_seed(1)
    rng = np.random.default_rng(1)
    resume = os.path.join(args.out, "depthnet.safetensors")
    init = os.path.join(args.init, "depthnet.safetensors")
    if args.resume and os.path.exists(init):
        print("[depthnet] resuming from", init, flush=True)
    nerf, dn = hip_modules(args, load(init) if (args.resume and os.path.exists(init)) else None)
    tr = DepthNetTrainer(dataset_type="blender", basedir="/tmp", expname="fit", no_batching=True, datadir="", half_res=False,
                         white_bkgd=True, N_importance=128, N_samples=64, use_viewdirs=True, input_dims_embed=3,
                         device="cuda", perturb=0.0, N_rand=args.rays)
    e1, _ = get_embedder(10, 0, 3)
    e2, _ = get_embedder(4, 0, 3)
    query = lambda i, v, f: tr.run_network(i, v, f, embed_fn=e1, embeddirs_fn=e2)  # noqa: E731
    kw = dict(network_query_fn=query, perturb=0.0, N_importance=128, network_fine=nerf, N_samples=64, network_fn=nerf,
              use_viewdirs=True, white_bkgd=True, raw_noise_std=0.0, trainer=tr, lindisp=True, depth_network=dn,
              model_mode="train", near=NEAR, far=FAR, ndc=False)
    H = W = 800
    _, K = synthetic.blender_intrinsics(H, W)
    tr.H, tr.W, tr.K = H, W, K
    opt = HipAdam(list(dn.parameters()), lr=args.depth_lr)
    dev = torch.device("cuda")
    t_start, it, planned, log = time.time(), 0, None, []
    n_pose = 4
    while True:
        os_, ds_ = [], []
        for _ in range(n_pose):
            idx = torch.randint(0, H * W, (args.rays // n_pose,), device=dev)
            o, d = camera_rays(H, W, K, random_pose(rng), idx, dev)
            os_.append(o), ds_.append(d)
        o, d = torch.cat(os_), torch.cat(ds_)
        gt, _, _ = analytic_scene.raycast(o, d)
        loss, dn_loss, psnr, _ = tr.core_optimization_loop(opt, kw, torch.stack([o, d], 0), 100 + it, gt)
        it += 1
        if it in (30, 60):
            torch.cuda.synchronize()
            if it == 30:
                t30 = time.time()
            else:
                planned = max(200, int((args.seconds - (time.time() - t_start)) / ((time.time() - t30) / 30)) + 60)
                print(f"[depthnet] {1e3 * (time.time() - t30) / 30:.1f} ms/iter -> planning {planned} iterations", flush=True)
        if planned:
            for g in opt.param_groups:
                g["lr"] = args.depth_lr * (0.05 ** min(1.0, it / planned))
        if it % 500 == 0:
            log.append((it, float(dn_loss), float(loss)))
            print(f"[depthnet] it {it} depth_net_loss {float(dn_loss):.5f} img_loss {float(loss):.5f} "
                  f"({time.time() - t_start:.0f} s)", flush=True)
        if it % 4000 == 0 or (planned and it >= planned):
            save(dn.state_dict(), resume)
        if planned and it >= planned:
            break
    json.dump({"iterations": it, "rays_per_iteration": args.rays, "lr": args.depth_lr, "log": log, "resumed": bool(args.resume)},
              open(os.path.join(args.out, "depthnet_fit.json"), "w"))


def fit_depthnet_direct(args):
    """DepthNet against ANALYTIC depth targets (analytic_scene.depth_target: the exact hit depth, continued over the
    background by the ray's closest approach to a surface, so silhouettes against the background carry no jump), Huber loss,
    on this repo's own DepthNet backward kernels and Adam (autograd.DepthNetFunction, HipAdam).  No frozen-field pass per
    step, so a step is the DepthNet alone and a run sees two orders of magnitude more rays than phase `depthnet`."""
    from nerf_sampling_amd.autograd import HipAdam
    from nerf_sampling_amd.depth_net import DepthNet

    torch.manual_seed(2)
    rng = np.random.default_rng(2)
    out_path = os.path.join(args.out, "depthnet.safetensors")
    init = os.path.join(args.init, "depthnet.safetensors")
    dn = DepthNet(hidden_sizes=[256] * 10, cat_hidden_sizes=[256] * 10, sphere_radius=2.0)
    if args.resume and os.path.exists(init):
        dn.load_state_dict(load(init))
        print("[direct] resuming from", init, flush=True)
    dn = dn.cuda()
    opt = HipAdam(list(dn.parameters()), lr=args.depth_lr)
    H = W = 800
    _, K = synthetic.blender_intrinsics(H, W)
    dev = torch.device("cuda")
    n_pose, delta = 8, args.huber
    t_start, it, planned, log = time.time(), 0, None, []
    while True:
        os_, ds_ = [], []
        for _ in range(n_pose):
            idx = torch.randint(0, H * W, (args.rays // n_pose,), device=dev)
            o, d = camera_rays(H, W, K, random_pose(rng), idx, dev)
            os_.append(o), ds_.append(d)
        o, d = torch.cat(os_), torch.cat(ds_)
        with torch.no_grad():
            target, hit = analytic_scene.depth_target(o, d, NEAR, FAR)
        z = dn(o, d)[:, 0]
        err = z - target
        a = err.abs()
        loss = torch.where(a < delta, 0.5 * err * err / delta, a - 0.5 * delta).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        dn.repack()
        it += 1
        if it in (30, 60):
            torch.cuda.synchronize()
            if it == 30:
                t30 = time.time()
            else:
                planned = max(200, int((args.seconds - (time.time() - t_start)) / ((time.time() - t30) / 30)) + 60)
                print(f"[direct] {1e3 * (time.time() - t30) / 30:.1f} ms/iter -> planning {planned} iterations", flush=True)
        if planned:
            for g in opt.param_groups:
                g["lr"] = args.depth_lr * (args.lr_floor ** min(1.0, it / planned))
        if it % 1000 == 0:
            with torch.no_grad():
                inwin = float((a[hit] < 0.1).float().mean())
                rms = float((err[hit] ** 2).mean().sqrt())
            log.append((it, float(loss), inwin, rms))
            print(f"[direct] it {it} huber {float(loss):.5f} hit rays: |err|<0.1 {inwin:.4f} rms {rms:.4f} "
                  f"({time.time() - t_start:.0f} s)", flush=True)
        if it % 10000 == 0 or (planned and it >= planned):
            save(dn.state_dict(), out_path)
        if planned and it >= planned:
            break
    json.dump({"iterations": it, "rays_per_iteration": args.rays, "lr": args.depth_lr, "huber_delta": delta, "log": log,
               "resumed": bool(args.resume)}, open(os.path.join(args.out, "depthnet_direct_fit.json"), "w"))


def evaluate(args):
    from nerf_sampling_amd import ops

    ops.set_compute_dtype("f32")
    nerf, dn = hip_modules(args, load(os.path.join(args.out, "depthnet.safetensors")))
    H = W = 800
    _, K = synthetic.blender_intrinsics(H, W)
    poses = synthetic.render_poses(40)[:, :3, :4]
    res = {}
    for dtype in (("f32",) if args.quick else ("f32", "bf16", "f16")):
        nw, dw = nerf.packed(dtype), dn.packed(dtype)
        rows = []
        for k in (0, 7, 13, 21, 34):
            out = ops.render_rays_depthnet(dw, nw, camera=(H, W, K, poses[k], 0, H), n_samples=64, mode="uniform", std=0.1,
                                           device="cuda", extras=True)
            gt, t, hit = analytic_scene.frame(H, W, K, poses[k], device="cuda")
            rgb = out["rgb"].reshape(H, W, 3)
            mse = float(((rgb - gt) ** 2).mean())
            z = out["z"].reshape(H, W, -1)
            inside = ((t >= z[..., 0]) & (t <= z[..., -1]))[hit].float().mean()
            rows.append({"pose": k, "psnr_vs_gt": -10 * math.log10(mse), "surface_in_window": float(inside),
                         "hit_frac": float(hit.float().mean())})
            if dtype == "f32" and k == 7:
                from PIL import Image

                Image.fromarray((rgb.clamp(0, 1).cpu().numpy() * 255).astype(np.uint8)).save(os.path.join(args.out, "pose7_f32.png"))
                Image.fromarray((gt.clamp(0, 1).cpu().numpy() * 255).astype(np.uint8)).save(os.path.join(args.out, "pose7_gt.png"))
        res[dtype] = rows
        print(dtype, json.dumps(rows), flush=True)
    # the ceiling of the field under guided sampling: windows centred on the ANALYTIC depth target (a perfect DepthNet)
    nw = nerf.packed("f32")
    ceil = []
    for k in (0, 7, 13, 21, 34):
        o, d, view = ops.get_rays(H, W, K, poses[k], device="cuda")[:3]
        o, d, view = o.reshape(-1, 3), d.reshape(-1, 3), view.reshape(-1, 3)
        gt, _, _ = analytic_scene.frame(H, W, K, poses[k], device="cuda")
        mse, n = 0.0, 0
        for r0 in range(0, H * W, 160000):
            sl = slice(r0, r0 + 160000)
            tgt, _ = analytic_scene.depth_target(o[sl], d[sl], NEAR, FAR)
            oo, dd, vv = o[sl].contiguous(), d[sl].contiguous(), view[sl].contiguous()
            z = ops.place_samples(oo, dd, tgt[:, None].contiguous(), 64, "uniform", 0.1, want_pts=False)[1]
            raw = ops.nerf_forward_rays(nw, oo, dd, z, vv)
            rgb = ops.raw2outputs(raw, z, dd, None, True, want_per_sample=False)[0]
            mse += float(((rgb - gt.reshape(-1, 3)[sl]) ** 2).sum())
            n += rgb.numel()
        ceil.append({"pose": k, "psnr_vs_gt_perfect_depth": -10 * math.log10(mse / n)})
    res["ceiling"] = ceil
    print("ceiling", json.dumps(ceil), flush=True)
    # what the vanilla 64 + 128 pass of the same field scores (the quality of the field itself)
    ws = ops.RenderWorkspace()
    nw = nerf.packed("f32")
    out = ops.render_rays_hierarchical(nw, nw, camera=(H, W, K, poses[7], 300, 500), n_coarse=64, n_importance=128,
                                       lindisp=True, white_bkgd=True, workspace=ws, device="cuda")
    gt, _, _ = analytic_scene.frame(H, W, K, poses[7], 300, 500, device="cuda")
    res["vanilla_64_128_psnr_vs_gt_pose7_rows300_500"] = -10 * math.log10(float(((out["rgb"].reshape(200, W, 3) - gt) ** 2).mean()))
    print("vanilla:", res["vanilla_64_128_psnr_vs_gt_pose7_rows300_500"], flush=True)
    json.dump(res, open(os.path.join(args.out, "eval.json"), "w"), indent=1)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--phase", required=True, choices=["nerf", "depthnet", "depthnet_direct", "eval"])
    ap.add_argument("--out", default="gpurun_out/fit")
    ap.add_argument("--seconds", type=float, default=420.0)
    ap.add_argument("--rays", type=int, default=2048)
    ap.add_argument("--raw-noise-std", type=float, default=1.0)
    ap.add_argument("--depth-lr", type=float, default=3e-4)
    ap.add_argument("--resume", action="store_true", help="start from the weights under --init")
    ap.add_argument("--init", default=os.path.join(ROOT, "tests", "golden", "fitted_scene"))
    ap.add_argument("--nerf-lr", type=float, default=5e-4)
    ap.add_argument("--quick", action="store_true", help="phase eval: fp32 only")
    ap.add_argument("--huber", type=float, default=0.02, help="phase depthnet_direct: Huber delta in depth units")
    ap.add_argument("--lr-floor", type=float, default=0.03, help="phase depthnet_direct: final / initial learning rate")
    ap.add_argument("--device", default="cuda", help="phase nerf only (plain torch); the other phases need the GPU")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    {"nerf": fit_nerf, "depthnet": fit_depthnet, "depthnet_direct": fit_depthnet_direct, "eval": evaluate}[a.phase](a)
