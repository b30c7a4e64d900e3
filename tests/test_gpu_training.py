"""DepthNet training step on the HIP backward kernels against torch-CPU autograd
of the oracle: gradients of every DepthNet parameter, the Adam update, and a short optimisation run."""

import copy

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _kwargs(m, **over):
    from test_gpu_render import make_trainer, render_kwargs

    tr = make_trainer(**over)
    return tr, render_kwargs(tr, m)


def test_gemm_strided_variants():
    from nerf_sampling_amd import autograd as A

    g = torch.Generator().manual_seed(0)
    for (M, N, K) in ((1, 1, 1), (5, 3, 7), (64, 64, 16), (100, 257, 319), (1024, 256, 1020)):
        x = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g); b = torch.randn(N, generator=g)
        dy = torch.randn(M, N, generator=g)
        y = A.linear_forward(x.cuda(), W.cuda(), b.cuda(), A.LEAKY).cpu()
        ref = torch.nn.functional.leaky_relu(x @ W.T + b, 0.01)
        assert torch.allclose(y, ref, rtol=1e-4, atol=1e-4 * K ** 0.5)
        assert torch.allclose(A.linear_backward_input(dy.cuda(), W.cuda()).cpu(), dy @ W, rtol=1e-4, atol=1e-4 * N ** 0.5)
        assert torch.allclose(A.linear_backward_input(dy.cuda(), W.cuda(), n_cols=max(1, K // 2)).cpu(),
                              (dy @ W)[:, : max(1, K // 2)], rtol=1e-4, atol=1e-4 * N ** 0.5)
        dW, db = A.linear_backward_weight(dy.cuda(), x.cuda())
        assert torch.allclose(dW.cpu(), dy.T @ x, rtol=1e-4, atol=1e-4 * M ** 0.5)
        assert torch.allclose(db.cpu(), dy.sum(0), rtol=1e-4, atol=1e-4 * M ** 0.5)


def test_posenc_and_points_backward():
    from nerf_sampling_amd import autograd as A

    x = (torch.rand(50, 3) * 4 - 2).requires_grad_(True)
    de = torch.randn(50, 63)
    (O.posenc(x, 10) * de).sum().backward()
    mine = A.posenc_backward(x.detach().cuda(), de.cuda(), 10).cpu()
    assert torch.allclose(mine, x.grad, rtol=1e-4, atol=1e-2)       # terms up to 2^9 * |de|
    o = torch.randn(20, 3); d = torch.randn(20, 3); z = torch.rand(20, 4, requires_grad=True)
    zc = z.detach().cuda().requires_grad_(True)
    dp = torch.randn(20, 4, 3)
    (A.points_along_rays(o.cuda(), d.cuda(), zc) * dp.cuda()).sum().backward()
    ((o[:, None] + d[:, None] * z[..., None]) * dp).sum().backward()
    assert torch.allclose(zc.grad.cpu(), z.grad, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("scene", ["tiny_synth", "lego_synth"])
def test_depthnet_training_gradients_match_oracle_autograd(golden, gpu_modules, scene):
    from nerf_sampling_amd import nerf_utils, ops

    ops.set_compute_dtype("f32")
    m = dict(gpu_modules(scene))
    m["depth"] = copy.deepcopy(m["depth"])
    for p in m["depth"].parameters():
        p.requires_grad_(True)
    tr, kw = _kwargs(m)
    rb = T(golden("render_rays_train")["ray_batch"])[:48]
    target = torch.rand(48, 3, generator=torch.Generator().manual_seed(4))
    res = nerf_utils.render_rays(rb.cuda(), **kw)
    assert res["depth_net_rgb_map"].requires_grad and res["depth_net_z_vals"].requires_grad
    loss = ((res["depth_net_rgb_map"] - target.cuda()) ** 2).mean() + torch.nn.functional.mse_loss(
        res["depth_net_z_vals"], res["max_z_vals"])
    loss.backward()
    # oracle: same losses through torch-CPU autograd, with the SAME (constant) regression target max_z
    p = {k: v.clone().requires_grad_(True) for k, v in m["params"]["depth"].items()}
    o, d, view = rb[:, 0:3], rb[:, 3:6], rb[:, 8:11]
    z = O.depthnet_forward(p, o, d)
    raw = O.run_network(m["params"]["fine"], o[:, None] + d[:, None] * z[..., None], view)
    rgb = torch.sigmoid(raw[:, 0, :3])               # single-sample compositing (see ns_raw2outputs N == 1)
    loss_o = ((rgb - target) ** 2).mean() + torch.nn.functional.mse_loss(z, res["max_z_vals"].detach().cpu())
    loss_o.backward()
    assert abs(float(loss) - float(loss_o)) < 1e-4 * max(1.0, abs(float(loss_o)))
    worst = 0.0
    for name, mod_p in m["depth"].named_parameters():
        g, go = mod_p.grad.cpu(), p[name].grad
        assert g is not None and go is not None, name
        rel = float((g - go).norm() / (go.norm() + 1e-12))
        worst = max(worst, rel)
        assert rel < 2e-3, (name, rel, float(go.norm()))
    print(f"{scene}: worst relative gradient error over {len(p)} tensors = {worst:.2e}")


def test_hip_adam_matches_torch_adam():
    from nerf_sampling_amd.autograd import HipAdam

    g = torch.Generator().manual_seed(2)
    w0 = torch.randn(300, 17, generator=g)
    a = w0.clone().cuda().requires_grad_(True); b = w0.clone().requires_grad_(True)
    oa, ob = HipAdam([a], lr=1e-3), torch.optim.Adam([b], lr=1e-3)
    for _ in range(5):
        grad = torch.randn(300, 17, generator=g)
        a.grad, b.grad = grad.cuda(), grad.clone()
        oa.step(); ob.step()
    assert torch.allclose(a.detach().cpu(), b.detach(), rtol=1e-5, atol=1e-6)
    sd = oa.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}      # torch's layout: checkpoints round-trip


def test_core_optimization_loop_reduces_loss(gpu_modules):
    """A few DepthNet updates on one fixed ray batch (Trainer.core_optimization_loop): the depth regression loss
    towards the frozen NeRF's max-weight sample goes down and only DepthNet weights move."""
    from nerf_sampling_amd import ops
    from nerf_sampling_amd.autograd import HipAdam

    ops.set_compute_dtype("f32")
    m = dict(gpu_modules("tiny_synth"))
    m["depth"] = copy.deepcopy(m["depth"])
    for p in m["depth"].parameters():
        p.requires_grad_(True)
    tr, kw = _kwargs(m)
    H = W = 24
    _, K = O.blender_intrinsics(H, W)
    tr.H, tr.W, tr.K = H, W, K
    o, d, _ = ops.get_rays(H, W, K, O.pose_spherical(20.0, -30.0, 4.0)[:3, :4])
    batch_rays = torch.stack([o[100:356], d[100:356]], 0)
    target = torch.rand(256, 3, generator=torch.Generator().manual_seed(7)).cuda()
    opt = HipAdam(list(m["depth"].parameters()), lr=1e-3)
    kw.update(near=2.0, far=6.0, ndc=False)
    fine_before = [p.clone() for p in m["fine"].parameters()]
    losses = []
    for i in range(8):
        loss, dn_loss, psnr, _ = tr.core_optimization_loop(opt, kw, batch_rays, i, target)
        losses.append(float(dn_loss))
    print("depth_net_loss:", [round(x, 5) for x in losses])
    assert losses[-1] < losses[0]
    assert all(torch.equal(a, b) for a, b in zip(fine_before, m["fine"].parameters()))
