"""Mirror of the hot-path part of nerf_sampling/nerf_pytorch/utils.py (HIP-backed)."""

from __future__ import annotations

import importlib
from typing import Literal, Union

import torch

from . import ops


def load_obj_from_config(cfg: dict):
    """Create an object from {module, kwargs} (utils.py:12-21)."""
    module_name, class_name = cfg["module"].rsplit(".", maxsplit=1)
    cls = getattr(importlib.import_module(module_name), class_name)
    return cls(**cfg["kwargs"])


def override_config(config, update):
    """utils.py:125-140: overwrite existing keys only; unknown key -> KeyError."""
    for key, value in update.items():
        if key in config:
            config[key] = value
        else:
            raise KeyError(f"Key {key} does not exist in config")


def set_global_device(device: Union[Literal["cuda"], Literal["cpu"]]):
    """utils.py:143-149."""
    if device == "cuda":
        if torch.cuda.is_available():
            torch.set_default_device(device="cuda")
    elif device == "cpu":
        torch.set_default_device(device="cpu")


def freeze_model(model):
    for p in model.parameters():
        p.requires_grad = False


def unfreeze_model(model):
    for p in model.parameters():
        p.requires_grad = True


def save_state(global_step, network_fn, network_fine, optimizer, depth_network, sampling_optimizer, path) -> None:
    """Checkpoint in the reference's layout (utils.py:59-89)."""
    data = {
        "global_step": global_step,
        "network_fn_state_dict": network_fn.state_dict(),
        "optimizer_state_dict": optimizer.state_dict(),
        "sampling_optimizer_state_dict": sampling_optimizer.state_dict(),
        "depth_network": depth_network.state_dict(),
    }
    if network_fine is not None:
        data["network_fine_state_dict"] = network_fine.state_dict()
    torch.save(data, path)


def load_nerf(network_fn, network_fine, optimizer, ckpt):
    """Checkpoint keys as written by the reference's save_state (utils.py:59-106)."""
    if optimizer is not None:
        optimizer.load_state_dict(ckpt["optimizer_state_dict"])
    network_fn.load_state_dict(ckpt["network_fn_state_dict"])
    if network_fine is not None:
        network_fine.load_state_dict(ckpt["network_fine_state_dict"])


def load_depth_network(depth_network, sampling_optimizer, ckpt):
    if sampling_optimizer is not None:
        sampling_optimizer.load_state_dict(ckpt["sampling_optimizer_state_dict"])
    depth_network.load_state_dict(ckpt["depth_network"])


def solve_quadratic_equation(a: torch.Tensor, b: torch.Tensor, c: torch.Tensor):
    """[2, ...] roots, minus-sqrt root first, NaN where none exists (utils.py:159-179)."""
    return ops.solve_quadratic(a, b, c)


def find_intersection_points_with_sphere(origin, direction, sphere_radius):
    """t [n,2], points [n,2,3] (utils.py:182-217); sphere_radius is a 1-element tensor or a float."""
    r = float(sphere_radius.reshape(-1)[0]) if isinstance(sphere_radius, torch.Tensor) else float(sphere_radius)
    return ops.sphere_intersect(origin, direction, r)


def sample_points_around_mean(rays_o, rays_d, mean, n_samples=32, mode="gaussian", std=0.1):
    """pts [R,N,3], z_vals [R,N] (utils.py:220-244)."""
    return ops.place_samples(rays_o, rays_d, mean.reshape(-1), n_samples, mode, std)
