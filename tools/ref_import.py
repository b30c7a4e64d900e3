"""Import the read-only reference (/root/reference) in THIS container only.

Used by tools/make_golden.py to capture golden vectors.  Useless on the GPU box (/root/reference does not exist
there) and never imported by the product, tests or bench.

The reference's hot-path modules import four arithmetic-free packages that are not
installed here (imageio, cv2, wandb, optuna) at module level; empty stand-in modules are
registered for those names so the import succeeds.
"""

import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    sys.dont_write_bytecode = True
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    for name in ("imageio", "cv2", "wandb"):
        if name not in sys.modules:
            try:
                __import__(name)
            except ImportError:
                _stub(name)
    if "optuna" not in sys.modules:
        try:
            __import__("optuna")
        except ImportError:
            opt = _stub("optuna")
            opt.samplers = _stub("optuna.samplers")
            trial = _stub("optuna.trial", Trial=type("Trial", (), {}))
            opt.trial = trial
            exc = _stub("optuna.exceptions", TrialPruned=type("TrialPruned", (Exception,), {}))
            opt.exceptions = exc
    import nerf_sampling.nerf_pytorch.run_nerf_helpers as helpers
    import nerf_sampling.nerf_pytorch.utils as utils
    import nerf_sampling.depth_nets.depth_net as depth_net
    import nerf_sampling.nerf_pytorch.nerf_utils as nerf_utils
    import nerf_sampling.nerf_pytorch.load_blender as load_blender
    import nerf_sampling.trainers.sampling_trainer as sampling_trainer

    return types.SimpleNamespace(
        helpers=helpers, utils=utils, depth_net=depth_net, nerf_utils=nerf_utils,
        load_blender=load_blender, sampling_trainer=sampling_trainer,
    )
