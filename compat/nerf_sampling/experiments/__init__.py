"""nerf_sampling.experiments.{run,render} -> this build's counterparts of the reference scripts (same flags)."""
import sys

from nerf_sampling_amd.experiments import render, run

sys.modules[f"{__name__}.render"] = render
sys.modules[f"{__name__}.run"] = run
