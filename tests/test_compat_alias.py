"""The `nerf_sampling` alias package (compat/): the reference's import names resolve to this build's modules, so its
yaml plugin entry (lego.yaml:32 `module: "nerf_sampling.trainers.DepthNetTrainer"`) and the imports of its experiment
scripts and tests (experiments/render.py:10-15, tests/tests.py:6-12) work unchanged.  Runs in a subprocess so that the
alias never shadows the real reference inside this test session (tools/make_golden.py imports that one)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r"""
import sys
import nerf_sampling, nerf_sampling_amd
assert nerf_sampling.__file__.replace("\\", "/").endswith("compat/nerf_sampling/__init__.py"), nerf_sampling.__file__
from nerf_sampling.definitions import ROOT_DIR
from nerf_sampling.nerf_pytorch.utils import load_obj_from_config, override_config, set_global_device
from nerf_sampling.nerf_pytorch import utils
from nerf_sampling.nerf_pytorch.run_nerf_helpers import NeRF
from nerf_sampling.depth_nets import depth_net
from nerf_sampling.nerf_pytorch.utils import find_intersection_points_with_sphere, solve_quadratic_equation
from nerf_sampling.nerf_pytorch import nerf_utils
from nerf_sampling.nerf_pytorch.trainers import BlenderTrainer
import nerf_sampling.nerf_pytorch.trainers.Trainer as T
import nerf_sampling.trainers.sampling_trainer as S
from nerf_sampling.trainers import DepthNetTrainer
import nerf_sampling_amd.trainers as A, nerf_sampling_amd.utils as U, nerf_sampling_amd.depth_net as D
assert DepthNetTrainer is A.DepthNetTrainer and S.DepthNetTrainer is A.DepthNetTrainer and T.Trainer is A.Trainer
assert utils is U and depth_net is D and depth_net.DepthNet is D.DepthNet and NeRF is nerf_sampling_amd.run_nerf_helpers.NeRF
assert nerf_utils.render_rays is nerf_sampling_amd.nerf_utils.render_rays
assert ROOT_DIR == "/data/ns_root"
tr = load_obj_from_config({"module": "nerf_sampling.trainers.DepthNetTrainer",
                           "kwargs": dict(dataset_type="blender", basedir="/tmp", expname="x", no_batching=True, datadir="",
                                          half_res=True, white_bkgd=True, n_depth_samples=8, sampling_mode="uniform",
                                          distance=0.1)})
assert type(tr) is A.DepthNetTrainer and tr.n_depth_samples == 8
# reference test tests.py:14-26 reads the same against these names
m = NeRF()
utils.freeze_model(m); assert all(not p.requires_grad for p in m.parameters())
utils.unfreeze_model(m); assert all(p.requires_grad for p in m.parameters())
print("alias ok")
"""


def test_alias_package_resolves_to_this_build():
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "compat"), ROOT]), NERF_SAMPLING_ROOT="/data/ns_root")
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "alias ok" in r.stdout, r.stderr[-3000:]
