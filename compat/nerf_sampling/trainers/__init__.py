"""nerf_sampling.trainers.DepthNetTrainer -- the plugin entry the reference's yaml names (lego.yaml: `module:`)."""
import sys

from nerf_sampling_amd import trainers as _t
from nerf_sampling_amd.trainers import DepthNetTrainer

sys.modules[f"{__name__}.sampling_trainer"] = _t
__all__ = ["DepthNetTrainer"]
