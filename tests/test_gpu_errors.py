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

    dn = DepthNet(hidden_sizes=[64, 64], cat_hidden_sizes=[128, 512]).cuda()    # a trunk layer wider than 256
    with pytest.raises(NotImplementedError):
        dn(torch.zeros(4, 3).cuda(), torch.ones(4, 3).cuda())
    dn = DepthNet(hidden_sizes=[64, 64], cat_hidden_sizes=[128, 128], multires=6).cuda()   # not the multires the kernel embeds
    with pytest.raises(NotImplementedError):
        dn(torch.zeros(4, 3).cuda(), torch.ones(4, 3).cuda())
    net = NeRF(D=8, W=320, input_ch=63, input_ch_views=27, use_viewdirs=True).cuda()  # wider than the kernels (128 / 256)
    with pytest.raises(NotImplementedError):
        net(torch.zeros(8, 90).cuda())
    net = NeRF(D=8, W=96, input_ch=63, input_ch_views=27, use_viewdirs=True).cuda()   # narrower: zero-padded to 128 at pack time
    assert net(torch.zeros(8, 90).cuda()).shape == (8, 4)
    net = NeRF(D=8, W=256, input_ch=21, input_ch_views=27, use_viewdirs=True).cuda()   # not the multires the kernel embeds
    with pytest.raises(NotImplementedError):
        net(torch.zeros(8, 48).cuda())
    net = NeRF(D=8, W=256, input_ch=63, input_ch_views=0, use_viewdirs=False).cuda()    # output_linear head: 63 features in
    with pytest.raises(NotImplementedError):
        net(torch.zeros(8, 90).cuda())
    assert net(torch.zeros(8, 63).cuda()).shape == (8, 4)


def test_fp16_operand_range_is_checked_at_pack_time():
    """fp16-operand handles (f16, f16x3) refuse weights beyond fp16's range instead of packing +-inf; bf16 / f32 take them."""
    from nerf_sampling_amd.run_nerf_helpers import NeRF

    net = NeRF(D=2, W=128, input_ch=63, input_ch_views=27, skips=[], use_viewdirs=True)
    with torch.no_grad():
        net.alpha_linear.weight[0, 5] = 1.0e5
    net = net.cuda()
    for dt in ("f16", "f16x3"):
        with pytest.raises(NotImplementedError, match="fp16"):
            net.packed(dt)
    net.packed("bf16"); net.packed("f32")


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


def test_empty_and_degenerate_inputs(gpu_modules):
    """R = 0 everywhere, N = 0 compositing, one ray / one sample: shapes as the reference produces them."""
    from nerf_sampling_amd import ops
    from nerf_sampling_amd.trainers import DepthNetTrainer

    m = gpu_modules("tiny_synth")
    e3 = torch.zeros(0, 3).cuda()
    assert ops.sphere_intersect(e3, e3, 2.0)[1].shape == (0, 2, 3)
    assert ops.posenc(e3, 10).shape == (0, 63)
    assert ops.depthnet_forward(m["depth"].packed("f32"), e3, e3).shape == (0, 1)
    assert ops.place_samples(e3, e3, torch.zeros(0).cuda(), 8, "uniform", 0.1)[1].shape == (0, 8)
    assert ops.nerf_forward(m["fine"].packed("f32"), torch.zeros(0, 8, 3).cuda(), e3).shape == (0, 8, 4)
    out = ops.raw2outputs(torch.zeros(0, 8, 4).cuda(), torch.zeros(0, 8).cuda(), e3)
    assert out[0].shape == (0, 3) and out[5].shape == (0, 8)
    assert ops.sort_rows(torch.zeros(0, 5).cuda()).shape == (0, 5)
    assert ops.render_rays_depthnet(m["depth"].packed("f32"), m["fine"].packed("f32"), rays=(e3, e3, e3), n_samples=8,
                                    mode="uniform", std=0.1)["rgb"].shape == (0, 3)
    tr = DepthNetTrainer(dataset_type="blender", basedir="/tmp", expname="x", no_batching=True, datadir="",
                         half_res=True, white_bkgd=True)
    res = tr.raw2outputs(torch.zeros(5, 0, 4).cuda(), torch.zeros(5, 0).cuda(), torch.ones(5, 3).cuda())
    assert res[0].shape == (5, 3) and float(res[0].abs().max()) == 0.0 and res[6].shape == (5, 0)   # sum over no samples
    one = ops.render_rays_depthnet(m["depth"].packed("f32"), m["fine"].packed("f32"),
                                   rays=(torch.tensor([[0.0, 0.0, 4.0]]).cuda(), torch.tensor([[0.0, 0.0, -1.0]]).cuda(),
                                         torch.tensor([[0.0, 0.0, -1.0]]).cuda()), n_samples=1, mode="depth_only", std=0.1,
                                   extras=True)
    assert one["rgb"].shape == (1, 3) and one["z"].shape == (1, 1) and torch.isfinite(one["rgb"]).all()
    assert one["weights"].shape == (1, 0)       # the reference's weights for one sample are [R, 0], never uninitialised


def test_gaussian_mode_through_the_operator_api(gpu_modules):
    """sampling_mode='gaussian' (the -e sweep's second mode): noise comes from torch's device generator with the
    reference's call shape, so a fixed seed reproduces the frame."""
    from test_gpu_render import make_trainer, render_kwargs

    from nerf_sampling_amd import nerf_utils, ops

    ops.set_compute_dtype("f32")
    m = gpu_modules("tiny_synth")
    tr = make_trainer(n_depth_samples=128, sampling_mode="gaussian", distance=1.0)
    kw = render_kwargs(tr, m)
    o, d, v, batch = ops.get_rays(12, 12, __import__("oracle.nerf_oracle", fromlist=["x"]).blender_intrinsics(12, 12)[1],
                                  __import__("oracle.nerf_oracle", fromlist=["x"]).pose_spherical(5.0, -30.0, 4.0)[:3, :4],
                                  near=2.0, far=6.0, want_batch=True)
    torch.manual_seed(3)
    a = nerf_utils.render_rays_test(batch, **kw)
    torch.manual_seed(3)
    b = nerf_utils.render_rays_test(batch, **kw)
    assert torch.equal(a["depth_net_rgb_map"], b["depth_net_rgb_map"])
    z = a["depth_net_z_vals"]
    assert z.shape == (144, 128) and (z[:, 1:] >= z[:, :-1])[~torch.isnan(z[:, 0])].all()
