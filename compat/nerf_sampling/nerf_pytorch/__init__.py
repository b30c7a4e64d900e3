"""nerf_sampling.nerf_pytorch.* -> nerf_sampling_amd.* (utils, nerf_utils, run_nerf_helpers, load_blender, trainers)."""
import sys

from nerf_sampling_amd import load_blender, nerf_utils, run_nerf_helpers, utils

for _name, _mod in (("utils", utils), ("nerf_utils", nerf_utils), ("run_nerf_helpers", run_nerf_helpers),
                    ("load_blender", load_blender)):
    sys.modules[f"{__name__}.{_name}"] = _mod
