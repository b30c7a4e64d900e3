"""Mirror of the trainer-side operators the render path calls back into.

Reference: nerf_sampling/nerf_pytorch/trainers/Trainer.py (run_network :789-806,
sample_coarse_points :579-649, sample_fine_points :651-710, _sample_points :553-577, train :712-787,
core_optimization_loop :506-544, the two batch samplers :232-269 / :400-475), trainers/Blender.py,
nerf_sampling/trainers/sampling_trainer.py (DepthNetTrainer.raw2outputs :153-230, create_nerf_model :54-122,
save_rays_data :124-138).  ``train`` renders (render_only) or runs the DepthNet optimisation loop on the HIP backward
kernels (autograd.py); wandb / optuna logging and the mp4 writer are out of scope.
"""

from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch

from . import nerf_utils, ops, utils
from .depth_net import DepthNet
from .run_nerf_helpers import Embedder, NeRF


class Trainer:
    """Attribute bag with the reference's constructor arguments and defaults (Trainer.py:19-130)."""

    def __init__(self, dataset_type, basedir, expname, no_batching, datadir, device="cpu", render_test=False,
                 config_path=None, N_rand=32 * 32 * 4, render_only=False, chunk=1024 * 32, render_factor=0,
                 multires=10, i_embed=0, multires_views=4, netchunk=1024 * 64, lrate=5e-4, lrate_decay=250,
                 use_viewdirs=True, N_importance=0, netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256,
                 ft_path=None, perturb=1.0, raw_noise_std=0.0, N_samples=64, lindisp=True, precrop_iters=0,
                 precrop_frac=0.5, i_weights=10000, i_testset=100, i_video=5000, i_print=100,
                 input_dims_embed: int = 1, save_train_set_render: bool = True, depth_net_lr: float = 0.0001,
                 train_depth_net_only: bool = False, trial=None, single_image=False, single_ray=False,
                 save_scene_data=False, compare_nerf=False, use_nerf_max_pts=False, use_full_nerf=False,
                 hip_graph: bool = True):
        for k, v in list(locals().items()):
            if k != "self":
                setattr(self, k, v)
        self.use_batching = not no_batching
        self.no_reload = False
        self.start = None
        self.K = self.global_step = self.W = self.H = self.c2w = None

    def cast_intrinsics_to_right_types(self, hwf):
        H, W, focal = hwf
        H, W = int(H), int(W)
        if self.K is None:
            self.K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
        self.H, self.W = H, W
        return [H, W, focal]

    # ---- operators called from render_rays / render_rays_test ------------------------------------
    def run_network(self, inputs, viewdirs, fn, embed_fn, embeddirs_fn, netchunk=1024 * 64):
        """Embed + MLP: inputs [R,N,3], viewdirs [R,3] -> [R,N,4]  (Trainer.py:789-806).

        With this package's NeRF module and 10/4-frequency embedders the whole operator is one
        MFMA kernel (no [R*N,90] embedding in memory, no netchunk loop).
        """
        fused = (isinstance(fn, NeRF) and isinstance(embed_fn, Embedder) and embed_fn.num_freqs == 10
                 and embed_fn.input_dims == 3)
        if fused and fn.use_viewdirs:
            fused = (viewdirs is not None and isinstance(embeddirs_fn, Embedder) and embeddirs_fn.num_freqs == 4
                     and embeddirs_fn.input_dims == 3)
        elif fused:                       # output_linear head: the reference feeds the 63 point features only
            fused = viewdirs is None
        if fused:
            if torch.is_grad_enabled() and inputs.requires_grad:   # training: gradient w.r.t. the points only
                from .autograd import NerfInputGrad

                return NerfInputGrad.apply(inputs, viewdirs, fn)
            return ops.nerf_forward(fn.packed(), inputs, viewdirs)
        inputs_flat = torch.reshape(inputs, [-1, inputs.shape[-1]])
        embedded = embed_fn(inputs_flat)
        if viewdirs is not None:
            input_dirs = viewdirs[:, None].expand(inputs.shape)
            embedded = torch.cat([embedded, embeddirs_fn(torch.reshape(input_dirs, [-1, input_dirs.shape[-1]]))], -1)
        outputs_flat = nerf_utils.batchify(fn, netchunk)(embedded)
        return torch.reshape(outputs_flat, list(inputs.shape[:-1]) + [outputs_flat.shape[-1]])

    def _sample_points(self, z_vals_mid, weights, perturb, pytest, rays_d, rays_o, n_importance=None):
        from .run_nerf_helpers import sample_pdf

        if n_importance is None:
            n_importance = self.N_importance
        z_samples = sample_pdf(z_vals_mid, weights[..., 1:-1], n_importance, det=(perturb == 0.0), pytest=pytest)
        z_samples = z_samples.detach()
        return z_samples, ops.points_along_rays(rays_o, rays_d, z_samples)

    def sample_coarse_points(self, near, far, perturb, N_rays, N_samples, viewdirs, network_fn, network_query_fn,
                             rays_o, rays_d, raw_noise_std, white_bkgd, pytest, lindisp, **kwargs):
        """9-tuple in the reference's order (Trainer.py:579-649)."""
        rgb_map = disp_map = acc_map = depth_map = alphas_map = raw = weights = z_vals = None
        if N_samples > 0:
            t_rand = None
            if perturb > 0.0:
                if pytest:
                    np.random.seed(0)
                    t_rand = torch.tensor(np.random.rand(N_rays, N_samples), dtype=torch.float32, device=rays_o.device)
                else:
                    t_rand = torch.rand([N_rays, N_samples], device=rays_o.device)
            z_vals = ops.coarse_z(near, far, N_samples, lindisp, t_rand)
            pts = ops.points_along_rays(rays_o, rays_d, z_vals)
            raw = network_query_fn(pts, viewdirs, network_fn)
            rgb_map, disp_map, acc_map, depth_map, density, alphas, weights = self.raw2outputs(
                raw, z_vals, rays_d, raw_noise_std, white_bkgd, pytest=pytest)
        return rgb_map, disp_map, acc_map, weights, depth_map, z_vals, weights, raw, alphas_map

    def sample_fine_points(self, z_vals, weights, perturb, pytest, rays_d, rays_o, rgb_map, disp_map, acc_map,
                           network_fn, network_fine, network_query_fn, viewdirs, raw_noise_std, white_bkgd):
        """12-tuple in the reference's order (Trainer.py:651-710)."""
        rgb_map_0 = disp_map_0 = acc_map_0 = raw = None
        pts = density = alphas = None
        if self.N_importance > 0:
            rgb_map_0, disp_map_0, acc_map_0 = rgb_map, disp_map, acc_map
            u = None
            if perturb != 0.0:
                if pytest:
                    np.random.seed(0)
                    u = torch.tensor(np.random.rand(z_vals.shape[0], self.N_importance), dtype=torch.float32,
                                     device=z_vals.device)
                else:
                    u = torch.rand([z_vals.shape[0], self.N_importance], device=z_vals.device)
            # z_mid, sample_pdf(weights[1:-1]) and sort(cat[z, samples]) in one kernel
            z_vals = ops.importance_z(z_vals, weights, self.N_importance, u)
            pts = ops.points_along_rays(rays_o, rays_d, z_vals)
            run_fn = network_fn if network_fine is None else network_fine
            raw = network_query_fn(pts, viewdirs, run_fn)
            rgb_map, disp_map, acc_map, depth_map, density, alphas, weights = self.raw2outputs(
                raw, z_vals, rays_d, raw_noise_std, white_bkgd, pytest=pytest)
        return (rgb_map_0, disp_map_0, acc_map_0, rgb_map, disp_map, acc_map, raw, z_vals, pts, density, alphas,
                weights)

    def load_data(self):
        raise NotImplementedError("only the Blender loader is provided (trainers.BlenderTrainer)")

    def render(self, render_test, save_scene_data, images, i_test, render_poses, hwf, render_kwargs_test):
        """renderonly_{test|path}_{step:06d}/ with NNN.png, psnr.txt[, scene_data.pt] -- Trainer.py:181-230
        (the mp4 of the reference needs imageio-ffmpeg and is not written)."""
        with torch.no_grad():
            images = images[i_test] if render_test else None
            testsavedir = os.path.join(self.basedir, self.expname, "renderonly_{}_{:06d}".format(
                "test" if render_test else "path", self.global_step))
            os.makedirs(testsavedir, exist_ok=True)
            _, _, avg_test_psnr = nerf_utils.render_path(
                render_poses, hwf, self.K, self.chunk, render_kwargs_test, step=self.global_step,
                save_scene_data=save_scene_data, gt_imgs=images, savedir=testsavedir, render_factor=self.render_factor)
        return avg_test_psnr

    def train(self, N_iters=200000 + 1):
        """Trainer.train (Trainer.py:712-787): load data, build / reload the networks; render_only: render the test
        (or spiral) poses and return the average PSNR; otherwise the DepthNet optimisation loop (random ray batches from
        one image, or -- use_batching -- from the shuffled rays of all training images), checkpoints every i_weights."""
        hwf, poses, i_test, i_val, i_train, images, render_poses = self.load_data()
        dev = "cuda" if self.device == "cuda" else self.device
        if self.render_test:
            render_poses = torch.tensor(np.array(poses[i_test])).to(dev)
        hwf = self.cast_intrinsics_to_right_types(hwf=hwf)
        os.makedirs(os.path.join(self.basedir, self.expname), exist_ok=True)
        optimizer, sampling_optimizer, render_kwargs_train, render_kwargs_test = self.create_nerf_model()
        if self.train_depth_net_only:
            for k in ("network_fn", "network_fine"):
                if render_kwargs_train[k] is not None:
                    utils.freeze_model(render_kwargs_train[k])
        if self.render_only:
            return self.render(self.render_test, self.save_scene_data, images, i_test, render_poses, hwf,
                               render_kwargs_test)
        images, poses, rays_rgb, i_batch = self.prepare_raybatch_tensor_if_batching_random_rays(poses, images, i_train)
        psnr = None
        # hip_graph (not in the reference; default on): the step is captured as one hipGraph after two eager steps
        step = (self.graphed_optimization_loop(sampling_optimizer, render_kwargs_train)
                if self.hip_graph and dev == "cuda" and hasattr(sampling_optimizer, "use_device_step")
                else lambda rays, it, tgt: self.core_optimization_loop(sampling_optimizer, render_kwargs_train, rays, it, tgt))
        for i in range(self.start + 1, N_iters):
            rays_rgb, i_batch, batch_rays, target_s = self.sample_random_ray_batch(rays_rgb, i_batch, i_train, images,
                                                                                   poses, i)
            loss, depth_net_loss, psnr, _ = step(batch_rays, i, target_s)
            self.update_learning_rate(optimizer)
            if i % self.i_print == 0:
                print(f"[TRAIN] Iter: {i} Loss: {float(loss)} depth_net_loss: {float(depth_net_loss)} "
                      f"PSNR: {float(psnr)}")
                # the reference's progress line, appended to {basedir}/{expname}/psnr.txt (Trainer.py:378-392; wandb is
                # out of scope)
                info = f"Iter: {i} Loss: {float(loss)}, Depth Net Loss: {float(depth_net_loss)}, PSNR: {float(psnr):.5f}"
                with open(os.path.join(self.basedir, self.expname, "psnr.txt"), "a") as file:
                    file.write(f"{info}\n")
            if i % self.i_weights == 0:
                path = os.path.join(self.basedir, self.expname, "{:06d}.tar".format(i))
                utils.save_state(self.global_step, render_kwargs_train["network_fn"],
                                 render_kwargs_train["network_fine"], optimizer, render_kwargs_train["depth_network"],
                                 sampling_optimizer, path)
            self.global_step += 1
        return psnr

    def update_learning_rate(self, optimizer):
        """Trainer.py:546-551 (decays the NeRF optimiser's lr, which never steps when train_depth_net_only)."""
        new_lrate = self.lrate * (0.1 ** (self.global_step / (self.lrate_decay * 1000)))
        for param_group in optimizer.param_groups:
            param_group["lr"] = new_lrate

    def prepare_raybatch_tensor_if_batching_random_rays(self, poses, images, i_train):
        """(images, poses, rays_rgb, i_batch) -- Trainer.py:232-269.  With use_batching the rays of every TRAINING image
        are generated once (ns_get_rays, on the device), joined with their pixel colours into rays_rgb
        [(n_train H W), ro+rd+rgb, 3] and shuffled with numpy's generator: the reference calls np.random.shuffle on the
        array itself, which draws the same permutation as shuffling an index vector of that length."""
        dev = "cuda" if self.device == "cuda" else self.device
        poses_t = torch.tensor(np.asarray(poses), dtype=torch.float32).to(dev)
        if not self.use_batching:
            return images, poses_t, None, None
        rows = []
        for img_i in i_train:
            o, d, _ = ops.get_rays(self.H, self.W, self.K, poses_t[img_i, :3, :4], device=dev)
            rgb = torch.tensor(np.asarray(images[img_i]), dtype=torch.float32, device=o.device).reshape(-1, 3)
            rows.append(torch.stack([o, d, rgb[:, :3]], 1))            # [H*W, 3, 3]
        rays_rgb = torch.cat(rows, 0)
        perm = np.arange(rays_rgb.shape[0])
        np.random.shuffle(perm)
        rays_rgb = rays_rgb[torch.from_numpy(perm).to(rays_rgb.device)]
        images = torch.tensor(np.asarray(images), dtype=torch.float32).to(dev)
        return images, poses_t, rays_rgb, 0

    def sample_random_ray_batch(self, rays_rgb, i_batch, i_train, images, poses, i):
        """use_batching: the next N_rand rows of the shuffled rays_rgb, reshuffled (torch.randperm, as the reference)
        after an epoch; otherwise N_rand random pixels of one random training image -- Trainer.py:400-475."""
        if self.use_batching:
            batch = torch.transpose(rays_rgb[i_batch : i_batch + self.N_rand], 0, 1)     # [ro+rd+rgb, B, 3]
            batch_rays, target_s = batch[:2], batch[2]
            i_batch += self.N_rand
            if i_batch >= rays_rgb.shape[0]:
                print("Shuffle data after an epoch!")
                rays_rgb = rays_rgb[torch.randperm(rays_rgb.shape[0], device=rays_rgb.device)]
                i_batch = 0
            return rays_rgb, i_batch, batch_rays, target_s
        img_i = 42 if self.single_image else np.random.choice(i_train)
        target = torch.tensor(np.asarray(images[img_i]), dtype=torch.float32)
        pose = poses[img_i, :3, :4]
        self.c2w = pose.clone().detach()
        rays_o, rays_d, _ = ops.get_rays(self.H, self.W, self.K, self.c2w)     # [H*W, 3]
        if i < self.precrop_iters:
            dH, dW = int(self.H // 2 * self.precrop_frac), int(self.W // 2 * self.precrop_frac)
            rows = torch.arange(self.H // 2 - dH, self.H // 2 + dH)
            cols = torch.arange(self.W // 2 - dW, self.W // 2 + dW)
        else:
            rows, cols = torch.arange(self.H), torch.arange(self.W)
        coords = torch.stack(torch.meshgrid(rows, cols, indexing="ij"), -1).reshape(-1, 2)
        if self.single_ray:
            select = np.array([91])
        else:
            select = np.random.choice(coords.shape[0], size=[self.N_rand], replace=False)
        sel = coords[select].long()
        flat = (sel[:, 0] * self.W + sel[:, 1]).to(rays_o.device)
        batch_rays = torch.stack([rays_o[flat], rays_d[flat]], 0)
        target_s = target[sel[:, 0], sel[:, 1]].to(rays_o.device)
        return rays_rgb, i_batch, batch_rays, target_s

    def _optimization_step(self, sampling_optimizer, render_kwargs_train, batch_rays, i, target_s, **render_extra):
        """forward + two losses + backward + update; (img_loss, depth_net_loss) as device scalars."""
        from .run_nerf_helpers import img2mse

        rgb, _disp, extras = nerf_utils.render(self.H, self.W, self.K, chunk=self.chunk, rays=batch_rays,
                                               verbose=i < 10, retraw=True, **render_kwargs_train, **render_extra)
        sampling_optimizer.zero_grad()
        img_loss = img2mse(rgb, target_s)
        depth_net_loss = torch.nn.functional.mse_loss(extras["depth_net_z_vals"], extras["max_z_vals"])
        (depth_net_loss + img_loss).backward()
        sampling_optimizer.step()
        return img_loss.detach(), depth_net_loss.detach()

    def core_optimization_loop(self, sampling_optimizer, render_kwargs_train, batch_rays, i, target_s):
        """One DepthNet update: (loss, depth_net_loss, psnr, psnr0) -- Trainer.py:506-544.  The two backward
        calls of the reference accumulate into the same .grad; one backward of the sum is identical."""
        from .run_nerf_helpers import mse2psnr

        img_loss, depth_net_loss = self._optimization_step(sampling_optimizer, render_kwargs_train, batch_rays, i, target_s)
        psnr = mse2psnr(img_loss)
        render_kwargs_train["depth_network"].repack()
        return img_loss, depth_net_loss, psnr, None

    def graphed_optimization_loop(self, sampling_optimizer, render_kwargs_train):
        """core_optimization_loop as ONE hipGraph replay per step (see GraphedDepthNetStep): same arguments after the
        first two, same return value, same updates bit for bit."""
        return GraphedDepthNetStep(self, sampling_optimizer, render_kwargs_train)


class GraphedDepthNetStep:
    """Trainer.core_optimization_loop (Trainer.py:506-544) captured as one hipGraph.

    A training step at the reference's batch size (N_rand = 1024 rays) is ~450 small kernels: the frozen field's
    64 + 128 vanilla pass, the DepthNet forward / backward layer by layer, the NeRF input gradient, 82 Adam updates.
    Eagerly the step is bound by the host side of those launches; captured once (torch.cuda.CUDAGraph = hipGraph on
    ROCm) and replayed, the host issues ONE launch per step.  What makes the step capturable: HipAdam's device-resident
    step counter / learning rate (ns_adam_step_dev), render_rays without its three host copies (nothing in the step
    reads them), and fixed batch shapes.  The first ``warmup`` calls run eagerly (they are real steps and warm every
    lazily-built cache); the next call captures and replays.  A batch of another shape runs eagerly.

    Call: ``step(batch_rays, i, target_s) -> (img_loss, depth_net_loss, psnr, None)`` like core_optimization_loop."""

    def __init__(self, trainer, sampling_optimizer, render_kwargs_train, warmup: int = 2):
        self.tr, self.opt, self.kw, self.warmup = trainer, sampling_optimizer, render_kwargs_train, warmup
        self.calls, self.graph, self.shape = 0, None, None
        self.opt.use_device_step()
        # everything the captured kernels point into must outlive the graph: packed weight streams of the frozen networks
        self._keep = [n.packed() for n in (render_kwargs_train.get("network_fn"), render_kwargs_train.get("network_fine"))
                      if n is not None]

    def _eager(self, batch_rays, i, target_s):
        from .run_nerf_helpers import mse2psnr

        img_loss, dn_loss = self.tr._optimization_step(self.opt, self.kw, batch_rays, i, target_s, _skip_host_copies=True)
        self.kw["depth_network"].repack()
        return img_loss, dn_loss, mse2psnr(img_loss), None

    def _capture(self, batch_rays, target_s):
        self.shape = (tuple(batch_rays.shape), tuple(target_s.shape))
        self.rays, self.target = batch_rays.clone(), target_s.clone()
        self.opt.zero_grad(set_to_none=True)           # backward inside the capture allocates the grads in the graph's pool
        torch.cuda.synchronize()
        self.opt.claim_capture_table()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.img_loss, self.dn_loss = self.tr._optimization_step(self.opt, self.kw, self.rays, 1 << 30, self.target,
                                                                     _skip_host_copies=True)

    def __call__(self, batch_rays, i, target_s):
        from .run_nerf_helpers import mse2psnr

        self.calls += 1
        if self.calls <= self.warmup or (self.shape is not None and
                                         self.shape != (tuple(batch_rays.shape), tuple(target_s.shape))):
            return self._eager(batch_rays, i, target_s)
        if self.graph is None:
            self._capture(batch_rays, target_s)        # records, does not execute
        self.rays.copy_(batch_rays)
        self.target.copy_(target_s)
        self.opt.sync_device_lr()
        self.graph.replay()
        self.opt.note_replayed_step()
        self.kw["depth_network"].repack()
        return self.img_loss, self.dn_loss, mse2psnr(self.img_loss), None


class BlenderTrainer(Trainer):
    def __init__(self, half_res, white_bkgd, testskip=8, near=2.0, far=6.0, **kwargs):
        self.half_res, self.testskip, self.white_bkgd = half_res, testskip, white_bkgd
        self.near, self.far = near, far
        super().__init__(**kwargs)

    def load_data(self):
        """trainers/Blender.py:19-32."""
        from .load_blender import load_blender_data

        images, poses, render_poses, hwf, i_split = load_blender_data(self.datadir, self.half_res, self.testskip)
        i_train, i_val, i_test = i_split
        if self.white_bkgd:
            images = images[..., :3] * images[..., -1:] + (1.0 - images[..., -1:])
        else:
            images = images[..., :3]
        return hwf, poses, i_test, i_val, i_train, images, render_poses.clone().detach()


class DepthNetTrainer(BlenderTrainer):
    """Constructor and operator signatures of sampling_trainer.py:17-52,153-230."""

    def __init__(self, distance=None, sampling_mode=None, n_depth_samples=None,
                 depth_net_path: Optional[str] = None, n_layers: int = 6, layer_width: int = 256,
                 sphere_radius: float = 2.0, **kwargs):
        self.n_layers, self.layer_width = n_layers, layer_width
        self.depth_net_path, self.sphere_radius = depth_net_path, sphere_radius
        self.distance, self.n_depth_samples, self.sampling_mode = distance, n_depth_samples, sampling_mode
        super().__init__(**kwargs)

    def create_nerf_model(self):
        """(optimizer, sampling_optimizer, render_kwargs_train, render_kwargs_test) -- sampling_trainer.py:54-122.

        Checkpoints in the reference's .tar format are read with torch.load(weights_only=True).
        """
        render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer = nerf_utils.create_nerf(self, NeRF)
        bds = {"near": self.near, "far": self.far}
        render_kwargs_train.update(bds)
        render_kwargs_test.update(bds)
        sizes = [self.layer_width for _ in range(self.n_layers)]
        dev = "cuda" if self.device == "cuda" else self.device
        depth_network = DepthNet(hidden_sizes=sizes, cat_hidden_sizes=list(sizes), sphere_radius=self.sphere_radius).to(dev)
        from .autograd import HipAdam

        sampling_optimizer = HipAdam(params=list(depth_network.parameters()), lr=self.depth_net_lr)
        ckpts = []
        if self.depth_net_path is not None and self.depth_net_path != "None":
            ckpts = [self.depth_net_path]
        elif os.path.isdir(os.path.join(self.basedir, self.expname)):
            d = os.path.join(self.basedir, self.expname)
            ckpts = [os.path.join(d, f) for f in sorted(os.listdir(d)) if "tar" in f]
        start = None
        if len(ckpts) > 0 and not self.no_reload:
            ckpt = torch.load(ckpts[-1], weights_only=True, map_location=dev)
            start = ckpt["global_step"]
            utils.load_depth_network(depth_network, sampling_optimizer, ckpt)
        self.global_step = self.start = start if start is not None else 0
        for kw, mode in ((render_kwargs_train, "train"), (render_kwargs_test, "test")):
            kw["depth_network"] = depth_network
            kw["model_mode"] = mode
        return optimizer, sampling_optimizer, render_kwargs_train, render_kwargs_test

    def save_rays_data(self, rays_o, pts, alpha):
        """{basedir}/{expname}/{expname}_{global_step}.safetensors holding origins / pts / alpha, the dump
        experiments/plot.py reads for its point-cloud plots (sampling_trainer.py:124-138)."""
        from safetensors.torch import save_file

        os.makedirs(os.path.join(self.basedir, self.expname), exist_ok=True)
        filename = os.path.join(self.basedir, self.expname, f"{self.expname}_{self.global_step}.safetensors")
        save_file({"origins": rays_o.detach().contiguous(), "pts": pts.detach().contiguous(),
                   "alpha": alpha.detach().contiguous()}, filename)
        return filename

    def raw2outputs(self, raw, z_vals, rays_d, raw_noise_std=0, white_bkgd=True, pytest=False, **kwargs):
        """7-tuple (rgb_map, disp_map, acc_map, depth_map, density, alphas, weights).

        Unknown keyword arguments are swallowed exactly like the reference does (its DepthNet-path
        callers pass the misspelled raw_noise= / white_bkdg=, nerf_utils.py:712-713,862-863).
        """
        noise = None
        if raw_noise_std > 0.0:
            if pytest:
                np.random.seed(0)
                noise = torch.tensor(np.random.rand(*list(raw[..., 3].shape)) * raw_noise_std, dtype=torch.float32,
                                     device=raw.device)
            else:
                noise = torch.randn(raw[..., 3].shape, device=raw.device) * raw_noise_std
        density = raw[..., 3]
        if z_vals.shape[-1] == 0:  # sampling_trainer.py:219-220
            R = raw.shape[0]
            zeros = torch.zeros((R,), device=raw.device)
            return (torch.zeros((R, 3), device=raw.device), torch.full((R,), 1e10, device=raw.device), zeros,
                    zeros.clone(), density, raw[..., 3].clone(), raw[..., 3].clone())
        if torch.is_grad_enabled() and raw.requires_grad:
            if z_vals.shape[-1] != 1 or noise is not None:
                raise NotImplementedError("autograd through compositing is implemented for the single-sample "
                                          "training path only (nerf_utils.py:692-715)")
            from .autograd import SingleSampleComposite

            rgb_map, disp_map, acc_map, depth_map, alphas, weights = SingleSampleComposite.apply(
                raw, z_vals, rays_d, bool(white_bkgd))
            return rgb_map, disp_map, acc_map, depth_map, density, alphas, weights
        rgb_map, disp_map, acc_map, depth_map, alphas, weights = ops.raw2outputs(raw, z_vals, rays_d, noise, white_bkgd)
        return rgb_map, disp_map, acc_map, depth_map, density, alphas, weights
