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
          out: Optional[Tensor] = None, accumulate: bool = False, act: int = 0, dact: int = 0,
          dact_ref: Optional[Tensor] = None, a_rowsum: Optional[Tensor] = None) -> Tensor:
    """C = A B^T (+ bias) through ns_gemm_fused; ``out`` may be a strided [M, N] view (its row stride is the leading
    dimension).  Epilogues: ``act`` on the result; ``dact``: result *= act'(.) evaluated from the activation OUTPUT
    ``dact_ref`` [M, N] (may be strided); ``a_rowsum`` [M] <- sum_k A[i, k]."""
    lib = _lib.load()
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    check(lib.ns_gemm_fused(_ptr(A), sa0, sa1, _ptr(B), sb0, sb1, _ptr(bias), _ptr(out), out.stride(0), M, N, K,
                            int(accumulate), int(act), int(dact), _ptr(dact_ref), 0 if dact_ref is None else dact_ref.stride(0),
                            _ptr(a_rowsum), _stream(A.device)), "ns_gemm_fused")
    return out


def _gemm_batched(problems) -> None:
    """Up to four GEMMs of the same kind in ONE launch (ns_gemm_fused_batched).  Each problem: the keyword arguments of
    ``_gemm`` with ``out`` given (A, sa0, sa1, B, sb0, sb1, bias, M, N, K, out, accumulate, act, dact, dact_ref, a_rowsum)."""
    lib = _lib.load()
    arr = (_lib.GemmProblem * len(problems))()
    dev = problems[0]["A"].device
    for q, pr in zip(arr, problems):
        out, ref = pr["out"], pr.get("dact_ref")
        q.A_dev, q.sa0, q.sa1 = pr["A"].data_ptr(), pr["sa0"], pr["sa1"]
        q.B_dev, q.sb0, q.sb1 = pr["B"].data_ptr(), pr["sb0"], pr["sb1"]
        q.bias_dev = None if pr.get("bias") is None else pr["bias"].data_ptr()
        q.C_dev, q.ldc = out.data_ptr(), out.stride(0)
        q.M, q.N, q.K = pr["M"], pr["N"], pr["K"]
        q.accumulate, q.act, q.dact = int(pr.get("accumulate", False)), int(pr.get("act", 0)), int(pr.get("dact", 0))
        q.dact_ref_dev, q.ld_ref = (None, 0) if ref is None else (ref.data_ptr(), ref.stride(0))
        q.a_rowsum_dev = None if pr.get("a_rowsum") is None else pr["a_rowsum"].data_ptr()
    check(lib.ns_gemm_fused_batched(arr, len(problems), _stream(dev)), "ns_gemm_fused_batched")


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
        xe = ops.posenc(flat, 10)
        skips = net._check_supported()
        lins = list(net.pts_linears)
        M = xe.shape[0]

        def fwd(x, lin, act):              # act(x W^T + b), activation in the GEMM epilogue
            return _gemm(x, x.stride(0), 1, lin.weight, lin.weight.shape[1], 1, lin.bias, M, lin.weight.shape[0], x.shape[1], act=act)

        def bwd(dy, lin, n_cols=None, dref=None):   # dy W[:, :n_cols], times relu'(dref) when the layer below has a ReLU
            Wt = lin.weight
            return _gemm(dy, dy.stride(0), 1, Wt, 1, Wt.shape[1], None, M, n_cols or Wt.shape[1], dy.shape[1],
                         dact=RELU if dref is not None else 0, dact_ref=dref)

        acts = []
        h = xe
        for i, lin in enumerate(lins):                      # recompute (run_nerf_helpers.py:114-118)
            h = fwd(h, lin, RELU)
            acts.append(h)
            if i in skips:
                h = torch.cat([xe, h], -1)
        last = len(lins) - 1
        # d h_last: through the head, then times relu'(trunk output) -- the trunk's last activation.  When that output was
        # concatenated with the embedding (a skip after the last layer cannot occur: _check_supported), plain [M, W].
        if net.use_viewdirs:              # alpha / feature / views / rgb head (run_nerf_helpers.py:119-131)
            dirs = viewdirs[:, None].expand(pts.shape).reshape(-1, 3).contiguous()
            ve = ops.posenc(dirs, 4)
            feat = fwd(h, net.feature_linear, NONE)
            vin = torch.cat([feat, ve], -1)
            hv = fwd(vin, net.views_linears[0], RELU)
            g = _dev(draw, "draw").reshape(-1, 4)
            g_rgb, g_sigma = g[:, :3], g[:, 3:4]                         # strided views, no copies
            d_hv = bwd(g_rgb, net.rgb_linear, dref=hv)
            d_feat = bwd(d_hv, net.views_linears[0], n_cols=net.W)
            d_h = bwd(d_feat, net.feature_linear)
            # + the sigma head, accumulated in place, then the trunk's last ReLU on the sum
            Wa = net.alpha_linear.weight
            _gemm(g_sigma, g_sigma.stride(0), 1, Wa, 1, Wa.shape[1], None, M, Wa.shape[1], 1, out=d_h, accumulate=True,
                  dact=RELU, dact_ref=acts[last])
        else:                             # output_linear head (:132-133): raw = h W_out^T + b, no view directions
            g = _dev(draw, "draw").reshape(-1, net.output_channels)
            d_h = bwd(g, net.output_linear, dref=acts[last])
        d_xe = torch.zeros_like(xe)
        for i in range(last, -1, -1):     # d_h is the gradient w.r.t. the PRE-activation of layer i here
            below = i - 1
            if below >= 0 and below in skips:               # layer i saw cat[xe, h_{i-1}]
                d_in = bwd(d_h, lins[i])                     # [M, 63 + W]: the xe part has no activation
                d_xe = d_xe + d_in[:, :63]
                d_h = d_in[:, 63:].contiguous()
                act_backward_(d_h, acts[below], RELU)
            elif below >= 0:
                d_h = bwd(d_h, lins[i], dref=acts[below])
            else:
                d_h = bwd(d_h, lins[i])                      # layer 0's input is xe
        d_xe = d_xe + d_h
        dpts = posenc_backward(flat, d_xe.contiguous(), 10)
        return dpts.reshape(R, N, 3), None, None


# ---- DepthNet, gradient w.r.t. its weights -----------------------------------------------------------------
class DepthNetFunction(torch.autograd.Function):
    """forward/backward of depth_net.py:117-169 layer by layer (affine skip branches, LeakyReLU trunk, sigmoid
    head).  params = [w, b] * (4 n + 1) in the order origin, direction, intersection, trunk, head.

    Round 4: no elementwise kernels beside the GEMMs.  A branch layer's input cat[h, e] is a row of ONE buffer per branch
    whose embedding columns are filled once, and each layer's GEMM writes its output straight into the next layer's h
    columns (the last one into the trunk's input): no torch.cat.  Activations (forward) and their derivatives (backward: the
    grad-input GEMM of the layer above multiplies by this layer's act') run in the GEMM epilogues, the bias gradient is a
    row sum the grad-weight GEMM takes along, and strided views replace the .contiguous() copies (ns_gemm_fused)."""

    @staticmethod
    def forward(ctx, o: Tensor, d: Tensor, near: float, far: float, radius: float, n: int, *params: Tensor):
        W_ = params[0::2]
        B_ = params[1::2]
        e_o, e_d = ops.posenc(o, 10), ops.posenc(d, 10)
        _, P = ops.sphere_intersect(o, d, radius)
        e_x = ops.posenc(P.reshape(-1, 6), 10)
        M, width, dev = o.shape[0], W_[0].shape[0], o.device
        embs = (e_o, e_d, e_x)
        Es = [e.shape[1] for e in embs]
        # trunk input: cat[h_o, h_d, h_x, e_o, e_d, e_x]; the branches write their last layer into the h columns
        y0 = torch.empty((M, 3 * width + sum(Es)), dtype=torch.float32, device=dev)
        col = 3 * width
        for e, E in zip(embs, Es):
            y0[:, col : col + E] = e
            col += E
        saved_in: List[Optional[Tensor]] = [None] * (3 * n)
        bufs, inp = [], []
        for b, (e, E) in enumerate(zip(embs, Es)):
            inp.append(torch.cat([e, e], -1))                           # layer 0 sees cat[h = e, e]
            bufs.append(torch.empty((max(n - 1, 1), M, width + E), dtype=torch.float32, device=dev))   # inputs of layers 1 .. n-1
            if n > 1:
                bufs[b][:, :, width:] = e                               # one broadcast copy: the e columns of every layer
        for i in range(n):                                              # layer i of the three branches in ONE launch
            probs = []
            for b in range(3):
                x, Wt = inp[b], W_[b * n + i]
                saved_in[b * n + i] = x
                out = y0[:, b * width : (b + 1) * width] if i == n - 1 else bufs[b][i][:, :width]
                probs.append(dict(A=x, sa0=x.stride(0), sa1=1, B=Wt, sb0=Wt.shape[1], sb1=1, bias=B_[b * n + i], M=M, N=width,
                                  K=x.shape[1], out=out))
                if i < n - 1:
                    inp[b] = bufs[b][i]
            _gemm_batched(probs)
        y = y0
        trunk_io = []
        for i in range(n):
            out = _gemm(y, y.stride(0), 1, W_[3 * n + i], W_[3 * n + i].shape[1], 1, B_[3 * n + i], M, width, y.shape[1], act=LEAKY)
            trunk_io.append((y, out))
            y = out
        s = _gemm(y, y.stride(0), 1, W_[4 * n], W_[4 * n].shape[1], 1, B_[4 * n], M, 1, width, act=SIGMOID)
        ctx.n, ctx.scale, ctx.width = n, float(far) - float(near), width
        ctx.saved_in, ctx.trunk_io, ctx.last, ctx.s = saved_in, trunk_io, y, s
        ctx.weights = W_
        return near * (1 - s) + far * s                      # depth_net.py:168

    @staticmethod
    def backward(ctx, dz: Tensor):
        n, W_, width = ctx.n, ctx.weights, ctx.width
        grads: List[Optional[Tensor]] = [None] * (2 * (4 * n + 1))
        dev = ctx.s.device

        def weight_grads(slot, dy, x):
            """dW = dy^T x [N, K] and db = column sums of dy (the row sums of A = dy^T) in ONE launch"""
            M_, N_, K_ = dy.shape[0], dy.shape[1], x.shape[1]
            db = torch.empty((N_,), dtype=torch.float32, device=dev)
            dW = _gemm(dy, 1, dy.stride(0), x, 1, x.stride(0), None, N_, K_, M_, a_rowsum=db)
            grads[2 * slot], grads[2 * slot + 1] = dW, db

        g = _dev(dz, "dz") * ctx.scale
        act_backward_(g, ctx.s, SIGMOID)
        weight_grads(4 * n, g, ctx.last)
        # d(trunk output n-1) = (g W_head) * leaky'(out_{n-1})
        Wh = W_[4 * n]
        d_y = _gemm(g, g.stride(0), 1, Wh, 1, Wh.shape[1], None, g.shape[0], width, 1, dact=LEAKY, dact_ref=ctx.trunk_io[n - 1][1])
        for i in range(n - 1, -1, -1):
            y_in, _y_out = ctx.trunk_io[i]
            weight_grads(3 * n + i, d_y, y_in)
            Wi = W_[3 * n + i]
            if i > 0:      # through the trunk layer below and ITS activation
                d_y = _gemm(d_y, d_y.stride(0), 1, Wi, 1, Wi.shape[1], None, d_y.shape[0], width, width, dact=LEAKY,
                            dact_ref=ctx.trunk_io[i - 1][1])
            else:          # below trunk layer 0 only the three (affine) branch outputs need a gradient
                d_y = _gemm(d_y, d_y.stride(0), 1, Wi, 1, Wi.shape[1], None, d_y.shape[0], 3 * width, width)
        d_h = [d_y[:, b * width : (b + 1) * width] for b in range(3)]   # strided views: no copies
        M = d_y.shape[0]
        for i in range(n - 1, -1, -1):                                  # layer i of the three branches: two launches
            probs = []
            for b in range(3):
                x = ctx.saved_in[b * n + i]
                db = torch.empty((width,), dtype=torch.float32, device=dev)
                dW = torch.empty((width, x.shape[1]), dtype=torch.float32, device=dev)
                grads[2 * (b * n + i)], grads[2 * (b * n + i) + 1] = dW, db
                probs.append(dict(A=d_h[b], sa0=1, sa1=d_h[b].stride(0), B=x, sb0=1, sb1=x.stride(0), M=width, N=x.shape[1], K=M,
                                  out=dW, a_rowsum=db))
            _gemm_batched(probs)
            if i > 0:
                probs, nxt = [], []
                for b in range(3):
                    Wi = W_[b * n + i]
                    out = torch.empty((M, width), dtype=torch.float32, device=dev)
                    nxt.append(out)
                    probs.append(dict(A=d_h[b], sa0=d_h[b].stride(0), sa1=1, B=Wi, sb0=1, sb1=Wi.shape[1], M=M, N=width, K=width, out=out))
                _gemm_batched(probs)
                d_h = nxt
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
