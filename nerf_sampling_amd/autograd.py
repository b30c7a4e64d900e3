"""Backward of the DepthNet training step (Trainer.core_optimization_loop, Trainer.py:506-544) on HIP kernels.

What needs gradients in the reference's training step (`train_depth_net_only`, run.py:105):
  * DepthNet: all weights (depth_net.py:117-169), from d(loss)/d(z)
  * the frozen NeRF: only w.r.t. its input point (one sample per ray at the predicted depth,
    nerf_utils.py:692-715) -- its weights are frozen (Trainer.py:724-728)
  * pts = o + d*z, and the single-sample compositing (rgb = sigmoid(raw rgb), see ns_raw2outputs N == 1)
The vanilla coarse+fine pass that produces the target depth runs without gradients on the fused path.

Each torch.autograd.Function below runs its arithmetic in libnerf_sampling_hip.so (ns_gemm_strided,
ns_act_*, ns_posenc[_backward], ns_points_backward); torch is used for tensor storage and `cat`/slicing.
Training batches are N_rand = 1024 rays: launch-bound, so layers are individual fp32 GEMMs here rather than
the fused inference kernels.
"""

from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch

from . import _lib, ops
from ._lib import check
from .ops import _dev, _ptr, _stream

Tensor = torch.Tensor
NONE, RELU, LEAKY, SIGMOID = 0, 1, 2, 3


def _gemm(A: Tensor, sa0: int, sa1: int, B: Tensor, sb0: int, sb1: int, bias: Optional[Tensor], M: int, N: int, K: int,
          out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    lib = _lib.load()
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    check(lib.ns_gemm_strided(_ptr(A), sa0, sa1, _ptr(B), sb0, sb1, _ptr(bias), _ptr(out), out.stride(0), M, N, K,
                              int(accumulate), _stream(A.device)), "ns_gemm_strided")
    return out


def linear_forward(x: Tensor, W: Tensor, b: Optional[Tensor], act: int = NONE) -> Tensor:
    """act(x @ W.T + b): x [M,K], W [N,K] (nn.Linear layout)."""
    x, W = _dev(x, "x"), _dev(W, "weight")
    M, K = x.shape
    N = W.shape[0]
    y = _gemm(x, K, 1, W, K, 1, None if b is None else _dev(b, "bias"), M, N, K)
    if act != NONE:
        check(_lib.load().ns_act_forward(_ptr(y), y.numel(), act, _stream(y.device)), "ns_act_forward")
    return y


def linear_backward_input(dy: Tensor, W: Tensor, n_cols: Optional[int] = None) -> Tensor:
    """dx[:, :n_cols] = dy @ W[:, :n_cols]   (dy [M,N], W [N,K])."""
    dy, W = _dev(dy, "dy"), _dev(W, "weight")
    M, N = dy.shape
    K = W.shape[1]
    return _gemm(dy, N, 1, W, 1, K, None, M, n_cols or K, N)


def linear_backward_weight(dy: Tensor, x: Tensor):
    """dW = dy.T @ x [N,K], db = dy.sum(0) [N]."""
    dy, x = _dev(dy, "dy"), _dev(x, "x")
    M, N = dy.shape
    K = x.shape[1]
    dW = _gemm(dy, 1, N, x, 1, K, None, N, K, M)
    db = torch.empty((N,), dtype=torch.float32, device=dy.device)
    check(_lib.load().ns_colsum(_ptr(dy), N, M, N, _ptr(db), _stream(dy.device)), "ns_colsum")
    return dW, db


def act_backward_(dy: Tensor, y: Tensor, act: int) -> Tensor:
    if act != NONE:
        check(_lib.load().ns_act_backward(_ptr(dy), _ptr(y), dy.numel(), act, _stream(dy.device)), "ns_act_backward")
    return dy


def posenc_backward(x: Tensor, de: Tensor, n_freqs: int) -> Tensor:
    x, de = _dev(x, "x"), _dev(de, "de")
    dx = torch.empty_like(x)
    check(_lib.load().ns_posenc_backward(_ptr(x), _ptr(de), x.shape[0], x.shape[1], n_freqs, _ptr(dx), _stream(x.device)),
          "ns_posenc_backward")
    return dx


# ---- pts = o + d * z -----------------------------------------------------------------------------------
class PointsAlongRays(torch.autograd.Function):
    @staticmethod
    def forward(ctx, o: Tensor, d: Tensor, z: Tensor):
        ctx.save_for_backward(d)
        return ops.points_along_rays(o, d, z)

    @staticmethod
    def backward(ctx, dpts: Tensor):
        (d,) = ctx.saved_tensors
        dpts = _dev(dpts, "dpts")
        R, N = dpts.shape[0], dpts.shape[1]
        dz = torch.empty((R, N), dtype=torch.float32, device=dpts.device)
        check(_lib.load().ns_points_backward(_ptr(dpts), _ptr(_dev(d, "d")), R, N, _ptr(dz), _stream(dpts.device)),
              "ns_points_backward")
        return None, None, dz


def points_along_rays(o: Tensor, d: Tensor, z: Tensor) -> Tensor:
    if torch.is_grad_enabled() and z.requires_grad:
        return PointsAlongRays.apply(o, d, z)
    return ops.points_along_rays(o, d, z)


# ---- single-sample compositing: rgb_map = sigmoid(raw rgb) -------------------------------------------------
class SingleSampleComposite(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw: Tensor, z: Tensor, rays_d: Tensor, white_bkgd: bool):
        rgb, disp, acc, depth, alphas, weights = ops.raw2outputs(raw, z, rays_d, None, white_bkgd)
        ctx.save_for_backward(rgb)
        ctx.mark_non_differentiable(disp, acc, depth, alphas, weights)
        return rgb, disp, acc, depth, alphas, weights

    @staticmethod
    def backward(ctx, drgb, *_unused):
        (rgb,) = ctx.saved_tensors
        g = _dev(drgb, "drgb").clone()
        act_backward_(g, rgb, SIGMOID)                      # d sigmoid = y (1 - y)
        draw = torch.zeros((rgb.shape[0], 1, 4), dtype=torch.float32, device=rgb.device)
        draw[:, 0, :3] = g
        return draw, None, None, None


# ---- frozen NeRF, gradient w.r.t. the input points -------------------------------------------------------
class NerfInputGrad(torch.autograd.Function):
    """forward: the fused MFMA kernel; backward: recompute the layers in fp32 (masks for ReLU), then the
    transposed chain down to the embedded point and through the positional encoding."""

    @staticmethod
    def forward(ctx, pts: Tensor, viewdirs: Tensor, net):
        ctx.net = net
        ctx.save_for_backward(pts, viewdirs)
        return ops.nerf_forward(net.packed(), pts, viewdirs)

    @staticmethod
    def backward(ctx, draw: Tensor):
        pts, viewdirs = ctx.saved_tensors
        net = ctx.net
        R, N = pts.shape[0], pts.shape[1]
        flat = pts.reshape(-1, 3).contiguous()
        dirs = viewdirs[:, None].expand(pts.shape).reshape(-1, 3).contiguous()
        xe, ve = ops.posenc(flat, 10), ops.posenc(dirs, 4)
        skips = net._check_supported()
        if not net.use_viewdirs:
            raise NotImplementedError("the input gradient is implemented for networks with view directions (the reference "
                                      "trains its DepthNet against use_viewdirs=True fields, lego.yaml:11)")
        lins = list(net.pts_linears)
        acts, ins = [], []
        h = xe
        for i, lin in enumerate(lins):                      # recompute (run_nerf_helpers.py:114-118)
            ins.append(h)
            h = linear_forward(h, lin.weight, lin.bias, RELU)
            acts.append(h)
            if i in skips:
                h = torch.cat([xe, h], -1)
        feat = linear_forward(h, net.feature_linear.weight, net.feature_linear.bias)
        vin = torch.cat([feat, ve], -1)
        hv = linear_forward(vin, net.views_linears[0].weight, net.views_linears[0].bias, RELU)
        # backward
        g = _dev(draw, "draw").reshape(-1, 4)
        d_hv = linear_backward_input(g[:, :3].contiguous(), net.rgb_linear.weight)
        act_backward_(d_hv, hv, RELU)
        W = net.W
        d_feat = linear_backward_input(d_hv, net.views_linears[0].weight, n_cols=W)
        d_h = linear_backward_input(d_feat, net.feature_linear.weight)
        d_h = d_h + linear_backward_input(g[:, 3:4].contiguous(), net.alpha_linear.weight)
        d_xe = torch.zeros_like(xe)
        for i in range(len(lins) - 1, -1, -1):
            if i in skips:                                  # output of layer i was concatenated as cat[xe, h]
                d_xe = d_xe + d_h[:, :63]
                d_h = d_h[:, 63:].contiguous()
            act_backward_(d_h, acts[i], RELU)
            d_in = linear_backward_input(d_h, lins[i].weight)
            d_h = d_in
        d_xe = d_xe + d_h                                   # layer 0 input is xe
        dpts = posenc_backward(flat, d_xe.contiguous(), 10)
        return dpts.reshape(R, N, 3), None, None


# ---- DepthNet, gradient w.r.t. its weights -----------------------------------------------------------------
class DepthNetFunction(torch.autograd.Function):
    """forward/backward of depth_net.py:117-169 layer by layer (affine skip branches, LeakyReLU trunk, sigmoid
    head).  params = [w, b] * (4 n + 1) in the order origin, direction, intersection, trunk, head."""

    @staticmethod
    def forward(ctx, o: Tensor, d: Tensor, near: float, far: float, radius: float, n: int, *params: Tensor):
        W_ = params[0::2]
        B_ = params[1::2]
        e_o, e_d = ops.posenc(o, 10), ops.posenc(d, 10)
        _, P = ops.sphere_intersect(o, d, radius)
        e_x = ops.posenc(P.reshape(-1, 6), 10)
        saved_in: List[Tensor] = []

        def branch(first, e):
            h = e
            for i in range(n):
                inp = torch.cat([h, e], -1)
                saved_in.append(inp)
                h = linear_forward(inp, W_[first + i], B_[first + i])
            return h

        h_o, h_d, h_x = branch(0, e_o), branch(n, e_d), branch(2 * n, e_x)
        y = torch.cat([h_o, h_d, h_x, e_o, e_d, e_x], -1)
        trunk_io = []
        for i in range(n):
            out = linear_forward(y, W_[3 * n + i], B_[3 * n + i], LEAKY)
            trunk_io.append((y, out))
            y = out
        s = linear_forward(y, W_[4 * n], B_[4 * n], SIGMOID)
        ctx.n, ctx.scale, ctx.width = n, float(far) - float(near), W_[0].shape[0]
        ctx.saved_in, ctx.trunk_io, ctx.last, ctx.s = saved_in, trunk_io, y, s
        ctx.weights = W_
        return near * (1 - s) + far * s                      # depth_net.py:168

    @staticmethod
    def backward(ctx, dz: Tensor):
        n, W_, width = ctx.n, ctx.weights, ctx.width
        grads: List[Optional[Tensor]] = [None] * (2 * (4 * n + 1))
        g = _dev(dz, "dz") * ctx.scale
        act_backward_(g, ctx.s, SIGMOID)
        grads[2 * 4 * n], grads[2 * 4 * n + 1] = linear_backward_weight(g, ctx.last)
        d_y = linear_backward_input(g, W_[4 * n])
        for i in range(n - 1, -1, -1):
            y_in, y_out = ctx.trunk_io[i]
            act_backward_(d_y, y_out, LEAKY)
            grads[2 * (3 * n + i)], grads[2 * (3 * n + i) + 1] = linear_backward_weight(d_y, y_in)
            # below trunk layer 0 only the three branch outputs (first 3*width columns) need a gradient
            d_y = linear_backward_input(d_y, W_[3 * n + i], n_cols=None if i > 0 else 3 * width)
        for b, first in enumerate((0, n, 2 * n)):
            d_h = d_y[:, b * width : (b + 1) * width].contiguous()
            for i in range(n - 1, -1, -1):
                inp = ctx.saved_in[first + i]
                grads[2 * (first + i)], grads[2 * (first + i) + 1] = linear_backward_weight(d_h, inp)
                if i > 0:
                    d_h = linear_backward_input(d_h, W_[first + i], n_cols=width)
        return (None, None, None, None, None, None, *grads)


def depthnet_forward_train(net, o: Tensor, d: Tensor) -> Tensor:
    n, _width = net._train_shape()
    mods = (list(net.origin_layers) + list(net.direction_layers) + list(net.intersection_layers)
            + [m for m in net.cat_layers if isinstance(m, torch.nn.Linear)] + [net.to_depth[0]])
    params = []
    for m in mods:
        params += [m.weight, m.bias]
    return DepthNetFunction.apply(_dev(o, "rays_o"), _dev(d, "rays_d"), float(net.near), float(net.far),
                                  float(net.sphere_radius.reshape(-1)[0]), n, *params)


# ---- Adam on the HIP kernel, state-dict compatible with torch.optim.Adam -------------------------------------
class HipAdam(torch.optim.Adam):
    """torch.optim.Adam whose step() runs ns_adam_step; state ('step', 'exp_avg', 'exp_avg_sq') and therefore
    state_dict()/load_state_dict() are torch's, so the reference's checkpoints round-trip (utils.py:59-122).

    ``use_device_step()`` moves the step counter (and the learning rate) into device memory: step() then launches
    ns_add_i32 + ns_adam_step_dev and reads nothing from the host, which is what lets trainers.GraphedDepthNetStep
    capture forward + backward + update in one hipGraph.  The per-parameter 'step' entries of the state are brought up
    to date whenever the state is read (state_dict())."""

    _dev_step: Optional[Tensor] = None
    _dev_lr: Optional[Tensor] = None
    _table_event = None
    _host_steps = 0          # steps taken through the device counter (eager calls and graph replays alike)

    def _init_state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0)
            st["exp_avg"] = torch.zeros_like(p)
            st["exp_avg_sq"] = torch.zeros_like(p)
        return st

    def use_device_step(self):
        """Switch to the device-resident step counter (idempotent).  All parameters must share one step count."""
        if self._dev_step is not None:
            return
        params = [p for g in self.param_groups for p in g["params"]]
        dev = params[0].device
        self._host_steps = self._common_step(params)
        self._dev_step = torch.full((1,), self._host_steps, dtype=torch.int32, device=dev)
        self._dev_lr = torch.full((1,), float(self.param_groups[0]["lr"]), dtype=torch.float32, device=dev)
        self._lr_seen = float(self.param_groups[0]["lr"])
        for p in params:
            self._init_state(p)
        # two row tables, [0] for eager steps and [1] for a captured step (a replayed graph re-reads its pinned rows on
        # every replay, so eager steps taken after a capture must not touch them)
        self._table_host = [torch.empty((5 * len(params),), dtype=torch.int64, device="cpu").pin_memory() for _ in range(2)]
        self._table_dev = [torch.empty((5 * len(params),), dtype=torch.int64, device=dev) for _ in range(2)]
        self._table_event = None

    def _common_step(self, params) -> int:
        """The one step count all parameters share ('step' is a tensor in torch's own state, a plain int in older checkpoints)."""
        def as_int(v):
            return int(v.item()) if isinstance(v, torch.Tensor) else int(v)

        steps = {as_int(self.state[p]["step"]) if len(self.state[p]) else 0 for p in params}
        if len(steps) > 1 or len(self.param_groups) != 1:
            raise NotImplementedError("the device step counter needs one parameter group and one common step count")
        return steps.pop() if steps else 0

    def load_state_dict(self, state_dict):
        """torch's loader; in device-step mode the device counter, the host count and the learning-rate scalar are re-seeded
        from the loaded state (a checkpoint loaded AFTER use_device_step() must not keep the old bias corrections)."""
        super().load_state_dict(state_dict)
        for st in self.state.values():                       # torch's step() convention: 'step' is a CPU fp32 tensor
            if "step" in st and not isinstance(st["step"], torch.Tensor):
                st["step"] = torch.tensor(float(st["step"]))
        if self._dev_step is not None:
            params = [p for g in self.param_groups for p in g["params"]]
            self._host_steps = self._common_step(params)
            self._dev_step.fill_(self._host_steps)
            self._lr_seen = None
            self.sync_device_lr()

    def claim_capture_table(self):
        """Called once by the (single) captured step of this optimizer: the captured update re-reads row table [1] on every
        replay, so a second capture from the same optimizer would overwrite the rows the first graph points at."""
        if getattr(self, "_capture_claimed", False):
            raise RuntimeError("this HipAdam already backs a captured step: one hipGraph capture per optimizer "
                               "(build a new optimizer, or reuse the existing GraphedDepthNetStep)")
        self._capture_claimed = True

    def note_replayed_step(self):
        """A captured graph containing step() was replayed once."""
        self._host_steps += 1

    def sync_device_lr(self):
        """Publish param_groups[0]['lr'] to the device scalar the captured update reads (one tiny fill, only on change)."""
        lr = float(self.param_groups[0]["lr"])
        if self._dev_lr is not None and lr != self._lr_seen:
            self._dev_lr.fill_(lr)
            self._lr_seen = lr

    def state_dict(self):
        if self._dev_step is not None:
            for st in self.state.values():
                if "step" in st:
                    st["step"].fill_(float(self._host_steps))
        return super().state_dict()

    @torch.no_grad()
    def step(self, closure=None):
        lib = _lib.load()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            if group.get("weight_decay", 0) or group.get("amsgrad", False) or group.get("maximize", False):
                raise NotImplementedError("HipAdam implements plain Adam only")
            if self._dev_step is not None:
                capturing = torch.cuda.is_current_stream_capturing()
                if not capturing:
                    self.sync_device_lr()
                    self._host_steps += 1
                dev = self._dev_step.device
                check(lib.ns_add_i32(_ptr(self._dev_step), 1, _stream(dev)), "ns_add_i32")
                # one launch for all parameter tensors: {p, g, m, v, n} rows, staged through a pinned host tensor that was
                # allocated BEFORE any capture (use_device_step): an asynchronous copy from pinned memory is something a
                # hipGraph capture records (an allocation is not), and the rows stay valid across replays because the
                # captured backward writes its gradients to the same addresses every time
                rows, keep, max_n = [], [], 0
                for p in group["params"]:
                    if p.grad is None:
                        continue
                    st = self._init_state(p)
                    g = p.grad.contiguous()
                    keep.append(g)
                    rows += [p.data.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()]
                    max_n = max(max_n, p.numel())
                if rows:
                    if not capturing and self._table_event is not None:
                        self._table_event.synchronize()          # the previous step's copy of the rows has been taken
                    n_rows, k = len(rows) // 5, int(capturing)
                    self._table_host[k][: len(rows)].copy_(torch.tensor(rows, dtype=torch.int64, device="cpu"))
                    self._table_dev[k][: len(rows)].copy_(self._table_host[k][: len(rows)], non_blocking=True)
                    if not capturing:
                        self._table_event = torch.cuda.Event()
                        self._table_event.record(torch.cuda.current_stream(dev))
                    self._table_keep = keep                      # gradient tensors the launch reads
                    check(lib.ns_adam_step_multi_dev(_ptr(self._table_dev[k]), n_rows, max_n, float(group["lr"]),
                                                     _ptr(self._dev_lr), float(b1), float(b2), float(group["eps"]),
                                                     _ptr(self._dev_step), _stream(dev)), "ns_adam_step_multi_dev")
                continue
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self._init_state(p)
                g = p.grad.contiguous()
                if self._dev_step is not None:
                    check(lib.ns_adam_step_dev(_ptr(p.data), _ptr(g), _ptr(st["exp_avg"]), _ptr(st["exp_avg_sq"]), p.numel(),
                                               float(group["lr"]), _ptr(self._dev_lr), float(b1), float(b2),
                                               float(group["eps"]), _ptr(self._dev_step), _stream(p.device)),
                          "ns_adam_step_dev")
                    continue
                st["step"] += 1
                check(lib.ns_adam_step(_ptr(p.data), _ptr(g), _ptr(st["exp_avg"]), _ptr(st["exp_avg_sq"]), p.numel(),
                                       float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                       int(st["step"].item()), _stream(p.device)), "ns_adam_step")
        return None
