"""Mirror of nerf_sampling/depth_nets/depth_net.py: DepthNet weight container + HIP forward."""

from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from . import ops
from .run_nerf_helpers import get_embedder
from .utils import find_intersection_points_with_sphere


class DepthNet(nn.Module):
    """Same constructor and state-dict keys as the reference (depth_net.py:13-107)."""

    def __init__(self, hidden_sizes: list = [128 for _ in range(6)],
                 cat_hidden_sizes: list = [128, 128, 128, 128, 256], origin_channels: int = 3,
                 direction_channels: int = 3, multires: int = 10, sphere_radius: float = 2.0,
                 near: int = 2, far: int = 6):
        super().__init__()
        self.sphere_radius = torch.tensor([sphere_radius])
        self.near, self.far = near, far
        self.multires = multires
        self.hidden_sizes, self.cat_hidden_sizes = list(hidden_sizes), list(cat_hidden_sizes)
        self.origin_embedder, self.origin_dims = get_embedder(multires=multires, input_dims=origin_channels)
        self.direction_embedder, self.direction_dims = get_embedder(multires=multires, input_dims=direction_channels)
        self.intersection_points_embedder, self.intersection_points_dim = get_embedder(multires=multires, input_dims=6)

        def branch(e_dim):
            layers = [nn.Linear(e_dim + e_dim, hidden_sizes[0])]
            for i, size in enumerate(hidden_sizes[:-1]):
                layers.append(nn.Linear(size + e_dim, hidden_sizes[i + 1]))
            return nn.Sequential(*layers)

        self.origin_layers = branch(self.origin_dims)
        self.direction_layers = branch(self.direction_dims)
        self.intersection_layers = branch(self.intersection_points_dim)
        cat = [nn.Linear(hidden_sizes[-1] * 3 + self.origin_dims + self.direction_dims + self.intersection_points_dim,
                         cat_hidden_sizes[0]), nn.LeakyReLU()]
        for i, size in enumerate(cat_hidden_sizes[:-1]):
            cat += [nn.Linear(size, cat_hidden_sizes[i + 1]), nn.LeakyReLU()]
        self.cat_layers = nn.Sequential(*cat)
        self.to_depth = nn.Sequential(nn.Linear(cat_hidden_sizes[-1], 1), nn.Sigmoid())
        self._packed = {}
        self._unpackable = set()      # operand types these weights were found not to fit (reset with the weights: repack())

    def calculate_intersection_points(self, rays_o, rays_d):
        _, pts = find_intersection_points_with_sphere(rays_o, rays_d, self.sphere_radius)
        return pts

    def _check_supported(self):
        """The HIP kernels run the FOLDED network (ns_pack.hip): any skip-branch widths, trunk widths <= 256."""
        if max(self.cat_hidden_sizes) > 256:
            raise NotImplementedError(
                f"the HIP kernels implement DepthNet trunks up to 256 wide (cat_hidden_sizes={self.cat_hidden_sizes})")
        if self.multires != 10 or self.origin_dims != 63 or self.direction_dims != 63:
            raise NotImplementedError("the HIP kernel is built for multires=10 and 3-channel origins/directions")

    def _train_shape(self):
        """(n_layers, width) for the layer-by-layer training path (autograd.DepthNetFunction), which keeps the literal,
        un-folded chain and is written for one uniform width and as many trunk as branch layers (the production
        shape, experiments/run.py:101-109)."""
        self._check_supported()
        widths = set(self.hidden_sizes) | set(self.cat_hidden_sizes)
        if len(widths) != 1 or len(self.hidden_sizes) != len(self.cat_hidden_sizes):
            raise NotImplementedError(
                "the HIP training step implements DepthNet with one uniform hidden width and as many trunk as branch "
                f"layers (hidden_sizes={self.hidden_sizes}, cat_hidden_sizes={self.cat_hidden_sizes})")
        return len(self.hidden_sizes), widths.pop()

    def packed(self, dtype: Optional[str] = None) -> ops.PackedWeights:
        """Device weight stream (cached).  ``dtype`` given: exactly that operand type.  None: the DepthNet operand type
        paired with the current compute dtype (ops.depthnet_dtype_for: f16 under bf16), falling back to the compute dtype
        itself if the weights do not fit fp16's range."""
        if dtype is None:
            name, paired = ops.get_compute_dtype(), ops.depthnet_dtype_for()
            # ("f16m" exists for the production trunk only: another shape takes every layer split)
            for cand in ((paired, "f16x3") if paired == "f16m" else (paired,)):
                if cand != name and cand not in self._unpackable:
                    try:
                        return self.packed(cand)
                    except NotImplementedError:
                        self._unpackable.add(cand)    # (e.g. weights beyond fp16's range) do not re-fold / re-pack per call
        else:
            name = dtype
        if name not in self._packed:
            self._check_supported()
            mods = (list(self.origin_layers) + list(self.direction_layers) + list(self.intersection_layers)
                    + [m for m in self.cat_layers if isinstance(m, nn.Linear)] + [self.to_depth[0]])
            dev = self.to_depth[0].weight.device
            self._packed[name] = ops.pack_depthnet([m.weight for m in mods], [m.bias for m in mods], self.hidden_sizes,
                                                   self.cat_hidden_sizes, name, dev if dev.type == "cuda" else "cuda")
        return self._packed[name]

    def repack(self):
        self._packed = {}
        self._unpackable = set()

    def __getstate__(self):
        # packed device weight streams are caches of the parameters: never pickled / deep-copied
        state = self.__dict__.copy()
        state["_packed"] = {}
        state["_unpackable"] = set()
        return state

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.repack()
        return r

    def forward(self, rays_o: torch.Tensor, rays_d: torch.Tensor):
        """[R,3], [R,3] -> depth [R,1] in [near, far] (depth_net.py:117-169).

        Under autograd with trainable weights the layer-by-layer differentiable path is used
        (nerf_sampling_amd.autograd.DepthNetFunction); otherwise the fused inference kernel."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from .autograd import depthnet_forward_train

            self.repack()  # weights are about to change
            return depthnet_forward_train(self, rays_o, rays_d)
        return ops.depthnet_forward(self.packed(), rays_o, rays_d, self.near, self.far,
                                    float(self.sphere_radius.reshape(-1)[0]))
