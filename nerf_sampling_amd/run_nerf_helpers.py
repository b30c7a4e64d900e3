"""Mirror of nerf_sampling/nerf_pytorch/run_nerf_helpers.py for the hot path (HIP-backed)."""

from __future__ import annotations

from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import ops

img2mse = lambda x, y: torch.mean((x - y) ** 2)  # noqa: E731   run_nerf_helpers.py:9
_TEN = {}


def mse2psnr(x):
    """-10 log(x) / log(10) as run_nerf_helpers.py:10; the constant tensor is cached per device (creating it per call is
    a blocking host-to-device copy, i.e. a stream synchronisation inside every training step)."""
    ten = _TEN.get(x.device)
    if ten is None:
        ten = _TEN[x.device] = torch.tensor([10.0], device=x.device)
    return -10.0 * torch.log(x) / torch.log(ten)


to8b = lambda x: (255 * np.clip(x, 0, 1)).astype(np.uint8)  # noqa: E731


class Embedder:
    """Positional encoding (run_nerf_helpers.py:15-45); ``embed`` runs the ns_posenc kernel."""

    def __init__(self, **kwargs):
        self.kwargs = kwargs
        if not (kwargs.get("include_input", True) and kwargs.get("log_sampling", True)):
            raise NotImplementedError("only include_input=True, log_sampling=True embedders are built")
        self.input_dims = kwargs["input_dims"]
        self.num_freqs = kwargs["num_freqs"]
        if kwargs["max_freq_log2"] != self.num_freqs - 1:
            raise NotImplementedError("max_freq_log2 must equal num_freqs - 1")
        self.out_dim = self.input_dims * (1 + 2 * self.num_freqs)

    def embed(self, inputs):
        return ops.posenc(inputs, self.num_freqs)

    __call__ = embed


def get_embedder(multires, i=0, input_dims=4):
    """Same signature/returns as run_nerf_helpers.py:48-63: (embed_fn, out_dim)."""
    if i == -1:
        return nn.Identity(), 3
    embedder = Embedder(include_input=True, input_dims=input_dims, max_freq_log2=multires - 1,
                        num_freqs=multires, log_sampling=True, periodic_fns=[torch.sin, torch.cos])
    return embedder, embedder.out_dim


class NeRF(nn.Module):
    """Weight container with the reference's state-dict keys (run_nerf_helpers.py:67-105).

    ``forward`` takes the embedded [M, 90] input like the reference and runs the fused MFMA kernel;
    ``run_network`` bypasses the embedding round trip entirely (trainers.DepthNetTrainer).
    """

    def __init__(self, D=8, W=256, input_ch=3, input_ch_views=3, output_ch=4, skips=[4], use_viewdirs=False):
        super().__init__()
        self.D, self.W = D, W
        self.input_ch, self.input_ch_views = input_ch, input_ch_views
        self.skips, self.use_viewdirs = skips, use_viewdirs
        self.pts_linears = nn.ModuleList(
            [nn.Linear(input_ch, W)]
            + [nn.Linear(W, W) if i not in self.skips else nn.Linear(W + input_ch, W) for i in range(D - 1)]
        )
        self.views_linears = nn.ModuleList([nn.Linear(input_ch_views + W, W // 2)])
        if use_viewdirs:
            self.feature_linear = nn.Linear(W, W)
            self.alpha_linear = nn.Linear(W, 1)
            self.rgb_linear = nn.Linear(W // 2, 3)
        else:
            self.output_linear = nn.Linear(W, output_ch)
        self._packed = {}

    # -- packing ---------------------------------------------------------------------------------
    def _check_supported(self):
        """The HIP kernels cover the reference's whole constructor (run_nerf_helpers.py:67-105) for W <= 256 (the packer
        zero-pads a width to the next kernel width, 128 or 256: the same arithmetic plus exact zeros): any D <= 32, any
        ``skips`` list, both heads (use_viewdirs True / False with output_linear).  Returns the active skips."""
        if self.input_ch != 63 or (self.use_viewdirs and self.input_ch_views != 27):
            raise NotImplementedError("the HIP kernel is built for multires=10 / multires_views=4 (63+27 inputs)")
        if not 2 <= self.W <= 256:
            raise NotImplementedError(f"the HIP kernels are built for 2 <= W <= 256, got {self.W}")
        return sorted({int(s_) for s_ in self.skips if 0 <= int(s_) < self.D - 1})

    @property
    def output_channels(self) -> int:
        return 4 if self.use_viewdirs else self.output_linear.out_features

    def packed(self, dtype: Optional[str] = None) -> ops.PackedWeights:
        """Device weight stream for the MFMA kernels (cached per dtype; call ``repack`` after updates)."""
        name = dtype or ops.get_compute_dtype()
        if name not in self._packed:
            skips = self._check_supported()
            mods = list(self.pts_linears) + ([self.feature_linear, self.alpha_linear, self.views_linears[0], self.rgb_linear]
                                             if self.use_viewdirs else [self.output_linear])
            dev = self.pts_linears[0].weight.device
            self._packed[name] = ops.pack_nerf([m.weight for m in mods], [m.bias for m in mods], self.D, self.W, skips, name,
                                               dev if dev.type == "cuda" else "cuda", use_viewdirs=self.use_viewdirs,
                                               output_ch=self.output_channels)
        return self._packed[name]

    def repack(self):
        self._packed = {}

    def __getstate__(self):
        # packed device weight streams are caches of the parameters: never pickled / deep-copied
        state = self.__dict__.copy()
        state["_packed"] = {}
        return state

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.repack()
        return r

    def forward(self, x):
        return ops.nerf_forward_embedded(self.packed(), x)


def get_rays(H, W, K, c2w):
    """rays_o, rays_d [H,W,3] (run_nerf_helpers.py:187-202)."""
    o, d, _ = ops.get_rays(H, W, K, c2w)
    return o.reshape(H, W, 3), d.reshape(H, W, 3)


def sample_pdf(bins, weights, N_samples, det=False, pytest=False):
    """Inverse-CDF sampling (run_nerf_helpers.py:250-293); random draws come from torch's generator."""
    u = None
    if pytest:
        np.random.seed(0)
        shape = list(bins.shape[:-1]) + [N_samples]
        u_np = np.broadcast_to(np.linspace(0.0, 1.0, N_samples), shape) if det else np.random.rand(*shape)
        u = torch.tensor(np.ascontiguousarray(u_np), dtype=torch.float32, device=bins.device)
    elif not det:
        u = torch.rand(list(bins.shape[:-1]) + [N_samples], device=bins.device)
    return ops.sample_pdf(bins, weights, N_samples, u)
