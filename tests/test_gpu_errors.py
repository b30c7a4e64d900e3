"""Error behaviour at the boundary: loud failures, never a silent fallback (gpu)."""

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_cpu_tensors_are_rejected():
    from nerf_sampling_amd import ops

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.posenc(torch.zeros(4, 3), 10)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.raw2outputs(torch.zeros(2, 4, 4), torch.zeros(2, 4), torch.zeros(2, 3))


def test_unsupported_network_shapes_raise():
    from nerf_sampling_amd.depth_net import DepthNet
    from nerf_sampling_amd.run_nerf_helpers import NeRF

    dn = DepthNet().cuda()        # class defaults: cat sizes [128,128,128,128,256] -- not a uniform width
    with pytest.raises(NotImplementedError):
        dn(torch.zeros(4, 3).cuda(), torch.ones(4, 3).cuda())
    net = NeRF(D=8, W=96, input_ch=63, input_ch_views=27, use_viewdirs=True).cuda()   # width without a kernel
    with pytest.raises(NotImplementedError):
        net(torch.zeros(8, 90).cuda())
    net = NeRF(D=8, W=256, input_ch=63, input_ch_views=27, use_viewdirs=False).cuda()
    with pytest.raises(NotImplementedError):
        net(torch.zeros(8, 90).cuda())


def test_default_depthnet_shape_contract():
    """tests.py:188-194: output is [n_rays, 1] (with a supported configuration)."""
    from nerf_sampling_amd.depth_net import DepthNet

    dn = DepthNet(hidden_sizes=[128] * 2, cat_hidden_sizes=[128] * 2).cuda()
    o = torch.tensor([[0.0, 0.0, 4.0]] * 5).cuda(); d = torch.tensor([[0.0, 0.0, -1.0]] * 5).cuda()
    z = dn(o, d)
    assert z.shape == (5, 1) and ((z >= 2) & (z <= 6)).all()


def test_bad_arguments_raise_value_error():
    from nerf_sampling_amd import ops

    with pytest.raises(ValueError):
        ops.place_samples(torch.zeros(2, 3).cuda(), torch.ones(2, 3).cuda(), torch.ones(2).cuda(), 1, "uniform", 0.1)
    with pytest.raises(ValueError):
        ops.place_samples(torch.zeros(2, 3).cuda(), torch.ones(2, 3).cuda(), torch.ones(2).cuda(), 8, "nope", 0.1)
    with pytest.raises(ValueError):
        ops.sort_rows(torch.zeros(1, 4096).cuda())
