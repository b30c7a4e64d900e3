"""Ray-range data parallelism over the GPUs of one node: contiguous image-row shards, one all-gather.

The reference is single-process; rays are independent and the weights are
read-only, so rank r renders rows [r*H/N, (r+1)*H/N) generating its own rays from the broadcast camera
scalars (no scatter), and ONE all_gather_into_tensor of the [rows, W, 4] (rgb + disp) shard assembles the
frame on every rank.  backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests.
"""

from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def row_range(H: int, rank: int, world: int) -> Tuple[int, int, int]:
    """(row0, row1, rows_per_rank): equal contiguous shards; trailing ranks may get fewer (or no) rows."""
    per = (H + world - 1) // world
    r0 = min(H, rank * per)
    return r0, min(H, r0 + per), per


class FrameRenderer:
    """Renders full frames, sharded by rows over the ranks of ``group``.

    ``render_rows(c2w, row0, row1, shard) -> (rgb [R,3], disp [R])`` produces this rank's shard; the default
    is the fused HIP path (ops.render_rays_depthnet) with rays generated on the device.  ``shard`` is this
    rank's interleaved [rows*W, 4] = (r, g, b, disp) buffer, the unit the all-gather moves: a renderer that
    returns views of it (the HIP renderers write it from the compositing kernel) costs no copy; anything else is
    copied in.
    """

    def __init__(self, H: int, W: int, render_rows: Callable, device, group=None):
        self.H, self.W, self.render_rows, self.device, self.group = H, W, render_rows, torch.device(device), group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.row0, self.row1, self.per = row_range(H, self.rank, self.world)
        # two shard / frame buffers: with wait=False the all-gather of frame i runs (on the collective's own stream)
        # under the kernels of frame i+1, which write the other pair
        self.shards = [torch.zeros((self.per * W, 4), dtype=torch.float32, device=self.device) for _ in range(2)]
        self.frames = ([torch.empty((self.world * self.per * W, 4), dtype=torch.float32, device=self.device) for _ in range(2)]
                       if self.world > 1 else [None, None])
        self.cur, self.work = 0, [None, None]

    @property
    def shard(self) -> torch.Tensor:
        return self.shards[self.cur]

    @property
    def rays_per_rank(self) -> int:
        return (self.row1 - self.row0) * self.W

    def render(self, c2w, wait: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
        """rgb [H,W,3], disp [H,W] of the whole frame, on every rank.

        wait=True (default): the tensors are ready for the caller's stream on return.  wait=False: the frame's
        all-gather is left in flight (it overlaps the next frame's kernels); the returned tensors may be read after
        finish(), or after the next-but-one render() -- the frame after next reuses their buffer."""
        b = self.cur = self.cur ^ 1
        if self.work[b] is not None:            # the gather that last used this buffer pair
            self.work[b].wait()
            self.work[b] = None
        shard = self.shards[b]
        n = self.rays_per_rank
        if n > 0:
            rgb, disp = self.render_rows(c2w, self.row0, self.row1, shard)
            if rgb.data_ptr() != shard.data_ptr():               # a renderer that did not write the shard itself
                shard[:n, :3] = rgb
                shard[:n, 3] = disp
        if self.world == 1:
            full = shard
        else:
            work = dist.all_gather_into_tensor(self.frames[b], shard, group=self.group, async_op=True)
            if wait:
                work.wait()
            else:
                self.work[b] = work
            full = self.frames[b]
        full = full[: self.H * self.W]
        return full[:, :3].reshape(self.H, self.W, 3), full[:, 3].reshape(self.H, self.W)

    def finish(self):
        """Wait (on the caller's stream) for every all-gather render(wait=False) left in flight."""
        for b in (0, 1):
            if self.work[b] is not None:
                self.work[b].wait()
                self.work[b] = None


def hip_row_renderer(depthnet, nerf, H: int, W: int, K, n_samples: int, mode: str, std: float, near: float = 2.0,
                     far: float = 6.0, sphere_radius: float = 2.0, device="cuda", events: Optional[list] = None,
                     max_events: int = 128, one_kernel: Optional[bool] = None, guard=None, guard_threshold: Optional[float] = None):
    """render_rows callable over the fused HIP path.  ``events``: list the (begin, end) hipEvent pair of the
    NeRF-MLP kernel of each call is appended to (bench.py's live roofline timing; the caller clears the list to
    start a new measurement).  At most ``max_events`` calls are timed per measurement.  ``one_kernel``: as in
    ops.render_rays_depthnet (None: the one-kernel renderer whenever the configuration supports it); ``guard``: the field
    packed "f16x3" for the PSNR guard pass (ops.render_rays_depthnet), or None; ``guard_threshold``: which rays it re-evaluates
    (None: the module setting, 0: every ray)."""
    from . import ops

    ws = ops.RenderWorkspace()
    ring = []        # hipEvent pairs: all created by the first call, none inside a timed region

    def render_rows(c2w, row0, row1, shard=None):
        ev = None
        if events is not None:
            if not ring:
                ring.extend((ops.Event(), ops.Event()) for _ in range(max_events))
            if len(events) < max_events:        # frames beyond the ring are rendered untimed
                ev = ring[len(events)]
                events.append(ev)
        out = ops.render_rays_depthnet(depthnet, nerf, camera=(H, W, K, c2w, row0, row1), n_samples=n_samples,
                                       mode=mode, std=std, near=near, far=far, sphere_radius=sphere_radius,
                                       workspace=ws, device=device, mlp_events=ev, shard=shard, one_kernel=one_kernel, guard=guard,
                                       guard_threshold=guard_threshold)
        return out["rgb"], out["disp"]

    return render_rows
