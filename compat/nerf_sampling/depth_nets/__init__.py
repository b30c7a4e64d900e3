"""nerf_sampling.depth_nets.depth_net -> nerf_sampling_amd.depth_net"""
import sys

from nerf_sampling_amd import depth_net

sys.modules[f"{__name__}.depth_net"] = depth_net
