"""Mirror of the render driver / ray-batch operators of nerf_sampling/nerf_pytorch/nerf_utils.py.

Same function names, arguments and returned keys as the reference (render_rays :614-733,
render_rays_test :736-876, sample_as_in_NeRF :497-611, render / render_test :88-255,
batchify* :45-85, create_nerf :393-494); the arithmetic runs in HIP kernels.
"""

from __future__ import annotations

import os
from typing import Optional

import torch

from . import ops, run_nerf_helpers
from .utils import sample_points_around_mean

DEBUG = False


def raw2alpha(raw, dists):
    """nerf_utils.py:27-42 -- kept for API parity (plain tensor expression, not on the hot path)."""
    return 1.0 - torch.exp(-torch.relu(raw) * dists)


def batchify(fn, chunk):
    if chunk is None:
        return fn

    def ret(inputs):
        return torch.cat([fn(inputs[i : i + chunk]) for i in range(0, inputs.shape[0], chunk)], 0)

    return ret


class _HostSink:
    """Frame-sized pinned host buffers that the per-chunk host copies of render_rays_test (the reference's four
    `.cpu()` calls per chunk, nerf_utils.py:866-870) land in directly, asynchronously, on a side stream.

    The reference serialises the GPU on every chunk (a blocking D2H of ~42 MB at 32768 rays x 64 samples) and then
    concatenates the chunks on the host (another pass over ~0.8 GB per 800x800 frame).  Here the "concatenation" is the
    buffer itself and the copies of chunk i run UNDER THE NeRF-MLP KERNEL OF CHUNK i+1: put() only queues a copy, and
    release(after=ev) starts the queued ones once `ev` -- the event the one-call renderer records right before its MLP
    kernel -- has been reached.  (Measured on MI355X: started right after a chunk the copies run beside the next chunk's
    small kernels -- DepthNet, placement -- and stretch them tenfold, while the MFMA-bound MLP kernel that follows runs
    alone; under the MLP kernel they are nearly free.  The copies go through the SDMA engines -- a kernel trace shows shader
    blits only because the profiler switches the engines off -- at ~27 GB/s beside the MLP kernel, 57 GB/s alone:
    profiles/r04f_api_path_copy_engines.log.)  The caller
    gets the same host tensors (same keys, shapes, dtypes, values).  Buffers come from torch's caching pinned allocator,
    fresh per frame, so tensors returned for one frame are never overwritten by the next (render_path keeps references
    across frames)."""

    def __init__(self, total_rows: int, device, pooled: bool = False):
        self.total, self.device, self.pooled = int(total_rows), torch.device(device), pooled
        self.stream = torch.cuda.Stream(self.device)
        self.bufs, self.keep, self.row0 = {}, [], 0
        self.pending, self.events, self.whole = [], [], []

    def _pinned(self, key, shape, dtype):
        """A pinned host buffer: fresh from torch's caching pinned allocator, or -- pooled=True, render_path's own loop,
        which knows when a frame's host tensors are dead -- one that recycle() handed back (torch's allocator took tens of
        frames to settle on 0.8 GB-per-frame requests: 35-55 ms/frame instead of 31)."""
        # device="cpu" explicitly: the experiment scripts switch torch's DEFAULT device to cuda (utils.py:143-149)
        free = _sink_pool.get((key, tuple(shape), dtype)) if self.pooled else None
        return free.pop() if free else torch.empty(tuple(shape), dtype=dtype, device="cpu", pin_memory=True)

    def recycle(self):
        """Hand this (finished, consumed) frame's pinned buffers to the pool.  Only render_path calls this: the tensors it
        got from render_test for this frame must not be read afterwards."""
        self.finish()
        for key, buf in self.bufs.items():
            _sink_pool.setdefault((key, tuple(buf.shape), buf.dtype), []).append(buf)
        for buf in self.whole:
            _sink_pool.setdefault(("__whole__", tuple(buf.shape), buf.dtype), []).append(buf)
        self.bufs, self.whole = {}, []

    def put(self, key: str, t: torch.Tensor) -> torch.Tensor:
        n = t.shape[0]
        buf = self.bufs.get(key)
        if buf is None:
            buf = self.bufs[key] = self._pinned(key, (self.total,) + tuple(t.shape[1:]), t.dtype)
        dst = buf[self.row0 : self.row0 + n]
        self.pending.append((dst, t))
        return dst

    def new_event_pair(self):
        pair = (ops.Event(), ops.Event())
        self.events.append(pair)           # alive until the frame's copies are done
        return pair

    def release(self, after=None):
        """Start every queued copy on the side stream: after the ops.Event ``after`` (already recorded on the compute
        stream by work submitted earlier), or after everything submitted to the compute stream so far."""
        if not self.pending:
            return
        if after is None:
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(self.device))
            self.stream.wait_event(done)
        else:
            from ._lib import check, load

            check(load().ns_stream_wait_event(ops.C.c_void_p(self.stream.cuda_stream), after.handle), "ns_stream_wait_event")
        with torch.cuda.stream(self.stream):
            for dst, t in self.pending:
                dst.copy_(t, non_blocking=True)
        self.keep.extend(t for _, t in self.pending)      # the device tensors stay alive until finish()
        self.pending.clear()

    def copy_whole(self, t: torch.Tensor) -> torch.Tensor:
        """A pinned host copy of a whole device tensor, queued on the side stream behind everything submitted to the
        compute stream so far; valid after finish()."""
        dst = self._pinned("__whole__", t.shape, t.dtype)
        self.whole.append(dst)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(self.device))
        self.stream.wait_event(done)
        with torch.cuda.stream(self.stream):
            dst.copy_(t, non_blocking=True)
        self.keep.append(t)
        return dst

    def advance(self, n_rows: int):
        self.row0 += int(n_rows)

    def finish(self):
        self.release()
        self.stream.synchronize()
        self.keep.clear()
        self.events.clear()


_sink_pool = {}          # (key, shape, dtype) -> pinned buffers recycled by render_path (emptied when it returns)
_held_sink = None        # render_path's pipeline: the previous frame's sink, its copies queued but not started yet -- they are
                         # released when the NEXT frame's MLP kernel starts (render_rays_test), or by finish()
# A standard-configuration frame may go through render_rays_test in ONE call when its per-sample outputs fit this many bytes
# (NS_WHOLE_FRAME_BYTES; 0 = off, the default: the reference's chunk loop).  Measured on MI355X, 800x800x64, render_path steady
# state (profiles/r04_api_path_whole_frame_vs_chunks.log): one call per frame 35.5 ms, twenty chunks 30.5 ms -- the path is bound
# by the device-to-host copies (0.82 GB per frame at ~27 GB/s beside the MLP kernel), and chunks let them start a chunk
# after the frame does instead of a frame later; the 16-21 ms of Python the chunk loop costs stay hidden under the GPU's 27.
_WHOLE_FRAME_BYTES = int(os.environ.get("NS_WHOLE_FRAME_BYTES", 0))
_pending_sinks = []      # frames whose host copies may still be in flight (only with _defer_host_sync, see _batchify)
_last_sink = None        # the sink of the most recent _batchify call (None: that call made blocking copies)


def drain_host_copies():
    """Wait for the host copies of every frame rendered with ``_defer_host_sync=True``; their host tensors are valid
    afterwards.  render_path calls this before it reads a frame's host tensors and when it is done."""
    for sink in _pending_sinks:
        sink.finish()
    _pending_sinks.clear()


def _batchify(render_fn, rays_flat, chunk, **kwargs):
    """Chunked calls + concatenation (nerf_utils.py:58-85).  Host outputs of render_rays_test land asynchronously in
    frame-sized pinned buffers (_HostSink).  By default the copies are complete when this returns, as with the
    reference's blocking `.cpu()` calls.  With ``_defer_host_sync=True`` (render_path's own loop) the tail of frame i's
    copies stays in flight under frame i+1's kernels: frame i-1 is drained here, frame i by drain_host_copies() or by
    the next frame."""
    global _last_sink, _held_sink
    all_returned = {}
    sink = None
    defer = bool(kwargs.pop("_defer_host_sync", False))
    if render_fn is render_rays_test and rays_flat.is_cuda and not kwargs.get("_blocking_host_copies", False):
        sink = kwargs["_host_sink"] = _HostSink(rays_flat.shape[0], rays_flat.device, pooled=defer)
        # Optionally (_WHOLE_FRAME_BYTES > 0) the standard configuration renders a frame as ONE render_rays_test call: the
        # reference's chunk loop (:58-85) bounds its memory, not its results -- rays are independent, the concatenated chunks
        # ARE the whole-frame tensors.  One call = three launches and four host copies per frame; bigger frames fall back to
        # chunks of the largest size under the bound.  Off by default: see _WHOLE_FRAME_BYTES.
        tr = kwargs.get("trainer")
        net = kwargs.get("network_fine") if kwargs.get("network_fine") is not None else kwargs.get("network_fn")
        if (tr is not None and rays_flat.shape[-1] > 8 and not (tr.compare_nerf or tr.use_nerf_max_pts or tr.use_full_nerf)
                and _one_call_eligible(kwargs.get("depth_network"), net, kwargs.get("network_query_fn"), tr, True)):
            per_ray = 20 * (1 if tr.sampling_mode == "depth_only" else int(tr.n_depth_samples))
            chunk = max(chunk, min(rays_flat.shape[0], _WHOLE_FRAME_BYTES // per_ray))
    for i in range(0, rays_flat.shape[0], chunk):
        returned = render_fn(rays_flat[i : i + chunk], **kwargs)
        for key in returned:
            all_returned.setdefault(key, []).append(returned[key])
        if sink is not None:
            sink.advance(min(chunk, rays_flat.shape[0] - i))
    _last_sink = sink if defer else None
    if sink is not None:
        if defer:
            # render_path's pipeline: this frame's (last chunk's) copies wait for the NEXT frame's MLP kernel -- beside the
            # next frame's small kernels (rays, DepthNet) the blit kernels that move them stretch those tenfold -- or for
            # finish(), whichever comes first
            if _held_sink is not None and _held_sink is not sink:
                _held_sink.release()
            _held_sink = sink
            drain_host_copies()          # the frame before the previous one: long finished by now
            _pending_sinks.append(sink)
        else:
            sink.release()               # the last chunk's copies
            sink.finish()
    return {key: (sink.bufs[key] if sink is not None and key in sink.bufs else torch.cat(all_returned[key], 0))
            for key in all_returned}


def batchify_rays(rays_flat, chunk=1024 * 32, **kwargs):
    return _batchify(render_rays, rays_flat, chunk, **kwargs)


def batchify_rays_test(rays_flat, chunk=1024 * 32, **kwargs):
    return _batchify(render_rays_test, rays_flat, chunk, **kwargs)


def prepare_rays(c2w, c2w_staticcam, use_viewdirs, ndc, H, W, K, near, far, rays):
    """rays [R,11], rays_o, rays_d, shape -- nerf_utils.py:156-188 (ndc=False only).  ``c2w_staticcam`` (:172-176, "visualize
    effect of viewdirs"): the view directions come from ``c2w`` (or ``rays``), origins and directions from the static
    camera."""
    if ndc:
        raise NotImplementedError("NDC rays (LLFF forward-facing scenes) are out of scope (SURVEY.md section 2)")
    if c2w is not None and c2w_staticcam is None:
        rays_o, rays_d, viewdirs, batch = ops.get_rays(H, W, K, c2w, near=near, far=far, want_batch=True)
        sh = (H, W, 3)
        if not use_viewdirs:
            batch = batch[:, :8].contiguous()
        return batch, rays_o, rays_d, sh
    viewdirs = None
    if c2w is not None:
        rays_o, rays_d, viewdirs = ops.get_rays(H, W, K, c2w)          # viewdirs = rays_d / |rays_d|, [H*W, 3]
        sh = (H, W, 3)
    else:
        rays_o, rays_d = rays
        sh = rays_d.shape
        if use_viewdirs:
            viewdirs = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
            viewdirs = torch.reshape(viewdirs, [-1, 3]).float()
    if use_viewdirs and c2w_staticcam is not None:
        rays_o, rays_d, _ = ops.get_rays(H, W, K, c2w_staticcam)
        sh = (H, W, 3)
    rays_o = torch.reshape(rays_o, [-1, 3]).float()
    rays_d = torch.reshape(rays_d, [-1, 3]).float()
    near_t, far_t = near * torch.ones_like(rays_d[..., :1]), far * torch.ones_like(rays_d[..., :1])
    batch = torch.cat([rays_o, rays_d, near_t, far_t], -1)
    if use_viewdirs:
        batch = torch.cat([batch, viewdirs], -1)
    return batch, rays_o, rays_d, sh


def _render(batch_fn, H, W, K, chunk, rays, c2w, ndc, near, far, use_viewdirs, c2w_staticcam, **kwargs):
    rays, rays_o, rays_d, sh = prepare_rays(c2w=c2w, c2w_staticcam=c2w_staticcam, use_viewdirs=use_viewdirs,
                                            ndc=ndc, H=H, W=W, K=K, near=near, far=far, rays=rays)
    all_returned = batch_fn(rays, chunk, **kwargs)
    for key in all_returned:
        all_returned[key] = torch.reshape(all_returned[key], list(sh[:-1]) + list(all_returned[key].shape[1:]))
    key_extract = ["depth_net_rgb_map", "depth_net_disp_map"]
    ret_list = [all_returned[key] for key in key_extract]
    ret_dict = {key: all_returned[key] for key in all_returned if key not in key_extract}
    ret_dict["rays_o"], ret_dict["rays_d"] = rays_o, rays_d
    return ret_list + [ret_dict]


def render(H, W, K, chunk=1024 * 32, rays=None, c2w=None, ndc=True, near=0.0, far=1.0, use_viewdirs=False,
           c2w_staticcam=None, **kwargs):
    """[rgb, disp, extras] through render_rays (nerf_utils.py:88-153)."""
    return _render(batchify_rays, H, W, K, chunk, rays, c2w, ndc, near, far, use_viewdirs, c2w_staticcam, **kwargs)


def render_test(H, W, K, chunk=1024 * 32, rays=None, c2w=None, ndc=True, near=0.0, far=1.0, use_viewdirs=False,
                c2w_staticcam=None, **kwargs):
    """[rgb, disp, extras] through render_rays_test (nerf_utils.py:191-255)."""
    return _render(batchify_rays_test, H, W, K, chunk, rays, c2w, ndc, near, far, use_viewdirs, c2w_staticcam,
                   **kwargs)


def render_path(render_poses, hwf, K, chunk, render_kwargs, step, wandb_log=False, save_scene_data=False,
                gt_imgs=None, savedir=None, render_factor=0):
    """Per-image loop with PSNR / PNG / psnr.txt / scene_data.pt outputs in the reference's format
    (nerf_utils.py:258-360).  wandb logging is out of scope; PNGs are written with PIL."""
    import numpy as np

    H, W, focal = hwf
    if render_factor != 0:
        H, W, focal = H // render_factor, W // render_factor, focal / render_factor
    if wandb_log:
        raise NotImplementedError("wandb logging is out of scope")
    rgbs, disps, all_pts, all_weights = [], [], [], []
    total_psnr, total_mse, psnr_info = 0, 0, None
    n_render_poses = render_poses.shape[0]

    def consume(i, rgb, disp, extras, sink):
        """Everything the reference does with a finished frame (:303-355): host arrays, PSNR, PNG, psnr.txt, scene data."""
        nonlocal total_psnr, total_mse, psnr_info
        if sink is not None:
            sink.finish()                                # this frame's host copies (rgb / disp included) are complete
            rgbs.append(np.array(rgb.numpy()))           # pageable copies: the pinned buffers go back to the cache
            disps.append(np.array(disp.numpy()))
        else:
            rgbs.append(rgb.cpu().numpy())
            disps.append(disp.cpu().numpy())
        if gt_imgs is not None and render_factor == 0:
            psnr = -10.0 * np.log10(np.mean(np.square(rgbs[-1] - np.asarray(gt_imgs[i]))))
            psnr_info = f"{i:03d}.png, PSNR: {psnr}"
            if render_kwargs["trainer"].compare_nerf and extras.get("max_z_vals") is not None:
                mse = torch.nn.functional.mse_loss(extras["max_z_vals"], extras["depth_net_z_vals"])
                total_mse += mse
                psnr_info += f", MSE: {mse}"
            total_psnr += psnr
        if savedir is not None:
            from PIL import Image

            Image.fromarray(run_nerf_helpers.to8b(rgbs[-1])).save(os.path.join(savedir, "{:03d}.png".format(i)))
            if psnr_info is not None:
                f = os.path.join(savedir, "psnr.txt")
                with open(f, "a") as file:
                    file.write(f"{psnr_info}\n")
                if i == n_render_poses - 1:
                    to_write = f"Avg of {n_render_poses} images:\nPSNR: {total_psnr/n_render_poses}\n"
                    if total_mse > 0:
                        to_write += f"MSE: {total_mse/n_render_poses}"
                    with open(f, "a") as file:
                        file.write(to_write)
            if save_scene_data:
                # pageable copies: the frame's pinned buffers go back to the allocator's cache instead of piling up
                # (0.7 GB of page-locked memory per 800x800 frame otherwise)
                for dst, key in ((all_pts, "depth_net_pts"), (all_weights, "depth_net_weights")):
                    flat = torch.flatten(extras[key], end_dim=2)
                    dst.append(torch.empty(flat.shape, dtype=flat.dtype, device="cpu", pin_memory=False).copy_(flat))
        if sink is not None:
            sink.recycle()                               # nothing of this frame's host tensors is read from here on

    # Software pipeline (SURVEY 8f-1 "asynchronous D2H of frames"): frame i+1 is submitted to the GPU BEFORE frame i is
    # consumed on the host, so the PNG / PSNR work and the tail of frame i's host copies run under frame i+1's kernels.
    # rgb / disp go to the host through the frame's copy stream too: a `.cpu()` on the compute stream would wait for
    # frame i+1.  Output files, their order and their contents are the reference's.
    prev = None
    for i, c2w in enumerate(render_poses):
        rgb, disp, extras = render_test(H, W, K, chunk=chunk, c2w=c2w[:3, :4], _defer_host_sync=True, **render_kwargs)
        sink = _last_sink if rgb.is_cuda else None
        if sink is not None:
            # rgb is a device tensor (nerf_utils.py:867 keeps it there); disp already IS this frame's pinned sink buffer
            # (depth_net_disp_map, filled chunk by chunk on the sink's stream): a host-to-host copy_ of it would run at once,
            # ahead of the queued device-to-host copies, and snapshot stale data.  consume() reads it after sink.finish().
            rgb = sink.copy_whole(rgb)
            if disp.is_cuda:
                disp = sink.copy_whole(disp)
        if prev is not None:
            consume(*prev)
        prev = (i, rgb, disp, extras, sink)
    if prev is not None:
        consume(*prev)
    global _held_sink
    _held_sink = None                                    # (finished by consume(): nothing is left in flight)
    _sink_pool.clear()                                   # the pinned buffers go back to torch's cache
    drain_host_copies()
    if save_scene_data and savedir is not None:
        torch.save({"all_pts": torch.cat(all_pts), "all_weights": torch.cat(all_weights)},
                   os.path.join(savedir, "scene_data.pt"))
    return np.stack(rgbs, 0), np.stack(disps, 0), total_psnr / n_render_poses


def create_nerf(args, model):
    """(render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer) -- nerf_utils.py:393-494."""
    embed_fn, input_ch = run_nerf_helpers.get_embedder(args.multires, args.i_embed, args.input_dims_embed)
    input_ch_views, embeddirs_fn = 0, None
    if args.use_viewdirs:
        embeddirs_fn, input_ch_views = run_nerf_helpers.get_embedder(args.multires_views, args.i_embed,
                                                                     args.input_dims_embed)
    output_ch = 5 if args.N_importance > 0 else 4
    skips = [4]
    dev = "cuda" if args.device == "cuda" else args.device
    model_nerf = model(D=args.netdepth, W=args.netwidth, input_ch=input_ch, output_ch=output_ch, skips=skips,
                       input_ch_views=input_ch_views, use_viewdirs=args.use_viewdirs).to(dev)
    grad_vars = list(model_nerf.parameters())
    model_fine = None
    if args.N_importance > 0:
        model_fine = model(D=args.netdepth_fine, W=args.netwidth_fine, input_ch=input_ch, output_ch=output_ch,
                           skips=skips, input_ch_views=input_ch_views, use_viewdirs=args.use_viewdirs).to(dev)
        grad_vars += list(model_fine.parameters())
    network_query_fn = lambda inputs, viewdirs, network_fn: args.run_network(  # noqa: E731
        inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, netchunk=args.netchunk)
    if (isinstance(embed_fn, run_nerf_helpers.Embedder) and isinstance(embeddirs_fn, run_nerf_helpers.Embedder)
            and embed_fn.num_freqs == 10 and embeddirs_fn.num_freqs == 4 and embed_fn.input_dims == 3
            and embeddirs_fn.input_dims == 3):
        standard_query_fn(network_query_fn)
    optimizer = torch.optim.Adam(params=grad_vars, lr=args.lrate, betas=(0.9, 0.999))
    start = 0
    ckpts = []
    if args.ft_path is not None and args.ft_path != "None":
        ckpts = [args.ft_path]
    elif os.path.isdir(os.path.join(args.basedir, args.expname)):
        d = os.path.join(args.basedir, args.expname)
        ckpts = [os.path.join(d, f) for f in sorted(os.listdir(d)) if "tar" in f]
    if len(ckpts) > 0 and not args.no_reload:
        from .utils import load_nerf

        ckpt = torch.load(ckpts[-1], weights_only=True, map_location=dev)
        start = ckpt["global_step"]
        load_nerf(model_nerf, model_fine, optimizer, ckpt)
    render_kwargs_train = {
        "network_query_fn": network_query_fn, "perturb": args.perturb, "N_importance": args.N_importance,
        "network_fine": model_fine, "N_samples": args.N_samples, "network_fn": model_nerf,
        "use_viewdirs": args.use_viewdirs, "white_bkgd": args.white_bkgd, "raw_noise_std": args.raw_noise_std,
        "trainer": args,
    }
    if args.dataset_type != "llff" or getattr(args, "no_ndc", False):
        render_kwargs_train["ndc"] = False
        render_kwargs_train["lindisp"] = args.lindisp
    render_kwargs_test = dict(render_kwargs_train)
    render_kwargs_test["perturb"] = False
    render_kwargs_test["raw_noise_std"] = 0.0
    return render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer


def sample_as_in_NeRF(ray_batch, network_fn, network_fine, network_query_fn, N_samples, trainer, perturb,
                      raw_noise_std, lindisp, white_bkgd, kwargs, pytest):
    """Vanilla coarse+fine pass; 8-tuple (density, z, pts, rgb_map, weights, alphas, disp, raw) -- :497-611."""
    N_rays = ray_batch.shape[0]
    rays_o, rays_d = ray_batch[:, 0:3].contiguous(), ray_batch[:, 3:6].contiguous()
    viewdirs = ray_batch[:, -3:].contiguous() if ray_batch.shape[-1] > 8 else None
    near, far = ray_batch[:, 6].contiguous(), ray_batch[:, 7].contiguous()
    (c_rgb, c_disp, c_acc, c_w, _c_depth, c_z, c_w, _c_raw, _c_alphas) = trainer.sample_coarse_points(
        near=near, far=far, perturb=perturb, N_rays=N_rays, N_samples=N_samples, viewdirs=viewdirs,
        network_fn=network_fn, network_query_fn=network_query_fn, rays_o=rays_o, rays_d=rays_d,
        raw_noise_std=raw_noise_std, white_bkgd=white_bkgd, pytest=pytest, lindisp=lindisp, kwargs=kwargs)
    (_r0, _d0, _a0, f_rgb, f_disp, _f_acc, f_raw, f_z, f_pts, f_density, f_alphas, f_w) = trainer.sample_fine_points(
        z_vals=c_z, weights=c_w, perturb=perturb, pytest=pytest, rays_d=rays_d, rays_o=rays_o, rgb_map=c_rgb,
        disp_map=c_disp, acc_map=c_acc, network_fn=network_fn, network_fine=network_fine,
        network_query_fn=network_query_fn, viewdirs=viewdirs, raw_noise_std=raw_noise_std, white_bkgd=white_bkgd)
    return f_density, f_z, f_pts, f_rgb, f_w, f_alphas, f_disp, f_raw


def render_rays(ray_batch, network_fn, network_query_fn, N_samples, trainer, retraw=True, lindisp=False,
                perturb=0.0, N_importance=0, network_fine=None, white_bkgd=False, raw_noise_std=0.0,
                verbose=False, pytest=False, **kwargs):
    """Training operator, forward (nerf_utils.py:614-733); same keys / host copies as the reference."""
    rays_o, rays_d = ray_batch[:, 0:3].contiguous(), ray_batch[:, 3:6].contiguous()
    viewdirs = ray_batch[:, -3:].contiguous() if ray_batch.shape[-1] > 8 else None
    # the vanilla pass only provides the regression target max_z: frozen networks, detached samples
    # (Trainer.py:569) -- no gradient reaches the DepthNet through it
    with torch.no_grad():
        (_dens, fine_z, _pts, _rgb, fine_w, _al, _disp, _raw) = sample_as_in_NeRF(
            ray_batch=ray_batch, N_samples=N_samples, network_fn=network_fn, network_fine=network_fine,
            network_query_fn=network_query_fn, trainer=trainer, perturb=perturb, raw_noise_std=raw_noise_std,
            lindisp=lindisp, white_bkgd=white_bkgd, pytest=pytest, kwargs=kwargs)
        max_z_vals, _, _ = ops.argmax_gather(fine_w, fine_z)
        max_pts = ops.points_along_rays(rays_o, rays_d, max_z_vals)
    from .autograd import points_along_rays as points_along_rays_ag

    depth_net_z_vals = kwargs["depth_network"](rays_o, rays_d)
    depth_net_pts = points_along_rays_ag(rays_o, rays_d, depth_net_z_vals)
    net = network_fine if network_fine is not None else network_fn
    depth_net_raw = network_query_fn(depth_net_pts, viewdirs, net)
    (rgb_map, disp_map, _acc, _depth, _density, _alphas, _weights) = trainer.raw2outputs(
        raw=depth_net_raw, z_vals=depth_net_z_vals, rays_d=rays_d, raw_noise=raw_noise_std, white_bkdg=white_bkgd,
        pytest=pytest)  # (sic) misspelled keywords, as in the reference: noise 0, white background
    # host copies of the reference (:723-727); `_skip_host_copies` (internal: the graph-captured training step, where a
    # blocking copy cannot be recorded and nothing reads these three) leaves them on the device
    to_host = (lambda t: t) if kwargs.get("_skip_host_copies", False) else (lambda t: t.cpu())
    ret = {"depth_net_rgb_map": rgb_map, "depth_net_disp_map": disp_map, "depth_net_z_vals": depth_net_z_vals,
           "max_z_vals": max_z_vals, "depth_net_pts": to_host(depth_net_pts.detach()), "max_pts": to_host(max_pts)}
    if retraw:
        ret["raw"] = to_host(depth_net_raw.detach())
    return ret


def standard_query_fn(fn):
    """Mark a network_query_fn as the standard one (embed + Trainer.run_network through this package's 10 / 4-frequency
    embedders, exactly what create_nerf builds): render_rays_test may then run its DepthNet branch as one fused C call."""
    fn._ns_standard_query = True
    return fn


def _one_call_eligible(depth_network, net, network_query_fn, trainer, viewdirs) -> bool:
    from .depth_net import DepthNet
    from .trainers import DepthNetTrainer

    return (viewdirs is not None and getattr(network_query_fn, "_ns_standard_query", False)
            and isinstance(depth_network, DepthNet) and isinstance(net, run_nerf_helpers.NeRF) and net.use_viewdirs
            and type(trainer).raw2outputs is DepthNetTrainer.raw2outputs
            and type(trainer).run_network is DepthNetTrainer.run_network
            and trainer.sampling_mode in ("uniform", "gaussian", "depth_only")
            and not (torch.is_grad_enabled() and any(p.requires_grad for p in depth_network.parameters())))


def render_rays_test(ray_batch, network_fn, network_query_fn, N_samples, trainer, retraw=True, lindisp=False,
                     perturb=0.0, N_importance=0, network_fine=None, white_bkgd=False, raw_noise_std=0.0,
                     verbose=False, pytest=False, **kwargs):
    """Inference operator (nerf_utils.py:736-876); same keys, shapes and host copies as the reference."""
    rays_o, rays_d = ray_batch[:, 0:3].contiguous(), ray_batch[:, 3:6].contiguous()
    viewdirs = ray_batch[:, -3:].contiguous() if ray_batch.shape[-1] > 8 else None
    ret = {}
    # host copies (nerf_utils.py:820-822, 866-870): blocking `.cpu()` when called on its own; inside batchify_rays_test
    # they go asynchronously into the frame's pinned buffers (_HostSink)
    sink = kwargs.get("_host_sink")
    released_early = False
    to_host = (lambda key, t: sink.put(key, t)) if sink is not None else (lambda key, t: t.cpu())
    if trainer.compare_nerf or trainer.use_nerf_max_pts or trainer.use_full_nerf:
        (_dens, fine_z, fine_pts, fine_rgb, fine_w, _al, fine_disp, fine_raw) = sample_as_in_NeRF(
            ray_batch=ray_batch, N_samples=N_samples, network_fn=network_fn, network_fine=network_fine,
            network_query_fn=network_query_fn, trainer=trainer, perturb=perturb, raw_noise_std=raw_noise_std,
            lindisp=lindisp, white_bkgd=white_bkgd, pytest=pytest, kwargs=kwargs)
        max_z_vals, max_weights, max_rgb_map = ops.argmax_gather(fine_w, fine_z, fine_raw)
        max_pts = ops.points_along_rays(rays_o, rays_d, max_z_vals)
        ret["max_z_vals"], ret["max_pts"] = to_host("max_z_vals", max_z_vals), to_host("max_pts", max_pts)
        ret["max_weights"] = to_host("max_weights", max_weights)
    if trainer.use_nerf_max_pts:
        rgb_map, disp_map = max_rgb_map, torch.zeros_like(max_rgb_map)  # [R,3] disp: reference quirk, :826
        weights, pts, z_vals = max_weights, max_pts, max_z_vals
    elif trainer.use_full_nerf:
        rgb_map, disp_map, weights, pts, z_vals = fine_rgb, fine_disp, fine_w, fine_pts, fine_z
    elif _one_call_eligible(kwargs.get("depth_network"), network_fine if network_fine is not None else network_fn,
                            network_query_fn, trainer, viewdirs):
        # The standard configuration (this package's DepthNet / NeRF modules, the query function create_nerf builds, the
        # trainer's own raw2outputs): DepthNet -> placement -> MLP -> compositing as ONE C call, bit-identical to the
        # operator chain below (tests/test_gpu_render.py::test_fused_matches_operator_chain and the tagged-path test).
        dn = kwargs["depth_network"]
        net = network_fine if network_fine is not None else network_fn
        ev = sink.new_event_pair() if sink is not None else None
        held = _held_sink
        dn_w, net_w, guard_w = ops.psnr_guard_handles(dn, net)
        if trainer.sampling_mode != "uniform" or trainer.n_depth_samples < 2:
            guard_w = None               # (the guard pass is defined for uniform placement)
        out = ops.render_rays_depthnet(dn_w, net_w, rays=(rays_o, rays_d, viewdirs),
                                       n_samples=trainer.n_depth_samples, mode=trainer.sampling_mode, std=trainer.distance,
                                       near=dn.near, far=dn.far, sphere_radius=float(dn.sphere_radius.reshape(-1)[0]),
                                       white_bkgd=True, extras=True, device=rays_o.device, mlp_events=ev, guard=guard_w)
        if sink is not None:
            sink.release(after=ev[0])    # the PREVIOUS chunk's host copies start with this chunk's MLP kernel
            if held is not None and held is not sink:
                held.release(after=ev[0])          # ... and so do the previous FRAME's (render_path's pipeline)
        released_early = True
        rgb_map, disp_map, weights, pts, z_vals = out["rgb"], out["disp"], out["weights"], out["pts"], out["z"]
    else:
        mean = kwargs["depth_network"](rays_o, rays_d)
        pts, z_vals = sample_points_around_mean(rays_o=rays_o, rays_d=rays_d, mean=mean,
                                                n_samples=trainer.n_depth_samples, mode=trainer.sampling_mode,
                                                std=trainer.distance)
        net = network_fine if network_fine is not None else network_fn
        raw = network_query_fn(pts, viewdirs, net)
        (rgb_map, disp_map, _acc, _depth, _density, _alphas, weights) = trainer.raw2outputs(
            raw=raw, z_vals=z_vals, rays_d=rays_d, raw_noise=raw_noise_std, white_bkdg=white_bkgd, pytest=pytest)
    ret["depth_net_rgb_map"] = rgb_map
    ret["depth_net_weights"] = to_host("depth_net_weights", weights)
    ret["depth_net_disp_map"] = to_host("depth_net_disp_map", disp_map)
    ret["depth_net_z_vals"] = to_host("depth_net_z_vals", z_vals)
    ret["depth_net_pts"] = to_host("depth_net_pts", pts)
    if sink is not None and not released_early:
        sink.release()                   # branches without a single MLP kernel to hide behind: copy right away
    return ret
