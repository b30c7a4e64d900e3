"""nerf_sampling.nerf_pytorch.trainers: BlenderTrainer, and the submodule names Trainer / Blender (as in the reference,
`trainers.Trainer` is the MODULE holding class Trainer)."""
import sys

from nerf_sampling_amd import trainers as _t
from nerf_sampling_amd.trainers import BlenderTrainer

Trainer = Blender = _t
sys.modules[f"{__name__}.Trainer"] = _t
sys.modules[f"{__name__}.Blender"] = _t
__all__ = ["BlenderTrainer"]
