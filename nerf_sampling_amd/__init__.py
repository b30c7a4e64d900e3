"""nerf_sampling_amd: MI355X-native (gfx950) ray-batch render path with DepthNet-guided sampling.

Drop-in for the hot path of MarcinKadziolka/nerf-sampling: the modules below mirror the
reference's operator interface (same names, arguments, return keys) and run every arithmetic
step in hand-written HIP kernels through the C ABI of ``libnerf_sampling_hip.so``
(``include/nerf_sampling_hip.h``).  There is no CPU fallback.

    reference module                                   this package
    nerf_sampling.nerf_pytorch.run_nerf_helpers   ->   nerf_sampling_amd.run_nerf_helpers
    nerf_sampling.nerf_pytorch.utils              ->   nerf_sampling_amd.utils
    nerf_sampling.nerf_pytorch.nerf_utils         ->   nerf_sampling_amd.nerf_utils
    nerf_sampling.depth_nets.depth_net            ->   nerf_sampling_amd.depth_net
    nerf_sampling.trainers (DepthNetTrainer)      ->   nerf_sampling_amd.trainers
"""

from .ops import get_compute_dtype, set_compute_dtype  # noqa: F401

__all__ = ["set_compute_dtype", "get_compute_dtype"]
