import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(autouse=True)
def _reset_process_wide_switches():
    """The compute dtype and the PSNR guard are process-wide settings (the CLIs set them): no test inherits another's."""
    yield
    if "nerf_sampling_amd.ops" in sys.modules:
        ops = sys.modules["nerf_sampling_amd.ops"]
        ops.set_psnr_guard(False)
        ops.set_compute_dtype("f32")


@pytest.fixture(scope="session")
def golden():
    """Loader for the committed fixtures captured from the reference (tools/make_golden.py)."""
    cache = {}

    def load(name):
        if name not in cache:
            with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
                cache[name] = {k: z[k] for k in z.files}
        return cache[name]

    return load


@pytest.fixture(scope="session")
def scenes():
    from oracle import nerf_oracle as O

    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = O.make_scene(name)
        return cache[name]

    return get


def _make_modules(scene_name):
    """This package's NeRF / DepthNet modules holding the oracle's seeded synthetic weights."""
    import torch

    from nerf_sampling_amd.depth_net import DepthNet
    from nerf_sampling_amd.run_nerf_helpers import NeRF
    from oracle import nerf_oracle as O

    cfg, params = O.SCENES[scene_name], O.make_scene(scene_name)
    out = {}
    for which in ("coarse", "fine"):
        net = NeRF(D=cfg[which]["D"], W=cfg[which]["W"], input_ch=63, input_ch_views=27, output_ch=5,
                   skips=[4], use_viewdirs=True)
        net.load_state_dict(params[which])
        out[which] = net.to("cuda")
    dn = DepthNet(hidden_sizes=[cfg["depth"]["width"]] * cfg["depth"]["n_layers"],
                  cat_hidden_sizes=[cfg["depth"]["width"]] * cfg["depth"]["n_layers"], sphere_radius=2.0)
    dn.load_state_dict(params["depth"])
    out["depth"] = dn.to("cuda")
    out["params"] = params
    for k in ("coarse", "fine", "depth"):      # inference fixtures: frozen (the training test unfreezes DepthNet)
        for p in out[k].parameters():
            p.requires_grad_(False)
    return out


@pytest.fixture(scope="session")
def gpu_modules():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = _make_modules(name)
        return cache[name]

    return get
