"""Host-side pack-time folds (ns_pack.hip) against the oracle's literal chains, on the CPU (no GPU needed):

* DepthNet: the three affine skip branches + first trunk layer -> one 252 -> C0 layer (depth_net.py:136-163);
* NeRF: views_linears[0] o feature_linear -> one (W + 27) -> W/2 layer (run_nerf_helpers.py:119-125).

The folds are composed in fp64 and stored in fp32; the oracle evaluates the literal fp32 chain, so the two differ by
fp32 rounding of the chain (checked against an fp64 evaluation of the same chain for a tight bound).
"""
import ctypes as C

import numpy as np
import pytest
import torch

from nerf_sampling_amd import _lib, synthetic
from oracle import nerf_oracle as O


def _ptrs(arrs):
    keep = [np.ascontiguousarray(a, dtype=np.float32) for a in arrs]
    return (C.c_void_p * len(keep))(*[k.ctypes.data for k in keep]), keep


def _fold_depthnet(p, hidden, c0):
    lib = _lib.load()
    n = len(hidden)
    names = ([f"origin_layers.{i}" for i in range(n)] + [f"direction_layers.{i}" for i in range(n)]
             + [f"intersection_layers.{i}" for i in range(n)] + ["cat_layers.0"])
    wa, k1 = _ptrs([p[k + ".weight"].numpy() for k in names])
    ba, k2 = _ptrs([p[k + ".bias"].numpy() for k in names])
    hs = (C.c_int * n)(*hidden)
    F = np.zeros((c0, 252), np.float32)
    fb = np.zeros((c0,), np.float32)
    _lib.check(lib.ns_fold_depthnet_front(n, hs, c0, wa, ba, F.ctypes.data_as(C.c_void_p), fb.ctypes.data_as(C.c_void_p)),
               "ns_fold_depthnet_front")
    return F, fb


def _rays(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    o = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1) * 4.03
    d = torch.nn.functional.normalize(-o + 0.3 * torch.randn(n, 3, generator=g), dim=-1)
    return o, d


@pytest.mark.parametrize("hidden,cat", [([32, 32, 32], [32, 32, 32]), ([128] * 6, [128, 128, 128, 128, 256]),
                                        ([48, 80], [64, 16, 200, 256])])
def test_depthnet_front_fold_matches_literal_chain(hidden, cat):
    """first trunk layer pre-activation: folded F . cat[e_o, e_d, e_x] + f == the literal 3-branch chain"""
    from nerf_sampling_amd.depth_net import DepthNet

    torch.manual_seed(7)
    p = {k: v.detach().clone() for k, v in DepthNet(hidden_sizes=hidden, cat_hidden_sizes=cat).state_dict().items()}
    F, fb = _fold_depthnet(p, hidden, cat[0])
    o, d = _rays(257)
    _, parts = O.depthnet_forward(p, o, d, return_parts=True)
    e = torch.cat([parts["e_o"], parts["e_d"], parts["e_x"]], -1)
    full = torch.cat([parts["h_o"], parts["h_d"], parts["h_x"], e], -1)
    lit32 = torch.nn.functional.linear(full, p["cat_layers.0.weight"], p["cat_layers.0.bias"])
    got = e.double() @ torch.from_numpy(F).double().T + torch.from_numpy(fb).double()
    # fp64 evaluation of the literal chain: the fold must agree to fp32 rounding of F itself
    pd = {k: v.double() for k, v in p.items()}
    _, pp = O.depthnet_forward(pd, o.double(), d.double(), return_parts=True)
    lit64 = torch.nn.functional.linear(torch.cat([pp["h_o"], pp["h_d"], pp["h_x"], pp["e_o"], pp["e_d"], pp["e_x"]], -1),
                                       pd["cat_layers.0.weight"], pd["cat_layers.0.bias"])
    e64 = torch.cat([pp["e_o"], pp["e_d"], pp["e_x"]], -1)
    got64 = e64 @ torch.from_numpy(F).double().T + torch.from_numpy(fb).double()
    scale = float(lit64.abs().max())
    assert float((got64 - lit64).abs().max()) <= 2e-6 * scale            # fp32 storage of F, 252 terms
    assert float((got - lit32.double()).abs().max()) <= 2e-5 * scale      # the reference's own fp32 chain rounding


def test_depthnet_fold_production_scene():
    p = synthetic.make_scene("lego_synth")["depth"]
    n, w = synthetic.SCENES["lego_synth"]["depth"]["n_layers"], synthetic.SCENES["lego_synth"]["depth"]["width"]
    F, fb = _fold_depthnet(p, [w] * n, w)
    o, d = _rays(128, 3)
    pd = {k: v.double() for k, v in p.items()}
    _, pp = O.depthnet_forward(pd, o.double(), d.double(), return_parts=True)
    lit64 = torch.nn.functional.linear(torch.cat([pp["h_o"], pp["h_d"], pp["h_x"], pp["e_o"], pp["e_d"], pp["e_x"]], -1),
                                       pd["cat_layers.0.weight"], pd["cat_layers.0.bias"])
    e64 = torch.cat([pp["e_o"], pp["e_d"], pp["e_x"]], -1)
    got64 = e64 @ torch.from_numpy(F).double().T + torch.from_numpy(fb).double()
    assert float((got64 - lit64).abs().max()) <= 2e-6 * float(lit64.abs().max())


@pytest.mark.parametrize("scene", ["tiny_synth", "lego_synth"])
def test_nerf_views_fold_matches_literal_chain(scene):
    lib = _lib.load()
    p = synthetic.make_scene(scene)["fine"]
    W = synthetic.SCENES[scene]["fine"]["W"]
    arrs = [np.ascontiguousarray(p[k].numpy(), dtype=np.float32) for k in
            ("feature_linear.weight", "feature_linear.bias", "views_linears.0.weight", "views_linears.0.bias")]
    wo = np.zeros((W // 2, W + 27), np.float32)
    bo = np.zeros((W // 2,), np.float32)
    _lib.check(lib.ns_fold_nerf_views(W, *[a.ctypes.data_as(C.c_void_p) for a in arrs], wo.ctypes.data_as(C.c_void_p),
                                      bo.ctypes.data_as(C.c_void_p)), "ns_fold_nerf_views")
    g = torch.Generator().manual_seed(1)
    h = torch.relu(torch.randn(300, W, generator=g, dtype=torch.float64))
    v = torch.randn(300, 27, generator=g, dtype=torch.float64)
    pd = {k: t.double() for k, t in p.items()}
    feat = torch.nn.functional.linear(h, pd["feature_linear.weight"], pd["feature_linear.bias"])
    lit = torch.nn.functional.linear(torch.cat([feat, v], -1), pd["views_linears.0.weight"], pd["views_linears.0.bias"])
    got = torch.cat([h, v], -1) @ torch.from_numpy(wo).double().T + torch.from_numpy(bo).double()
    assert float((got - lit).abs().max()) <= 2e-6 * float(lit.abs().max())


def test_bench_flop_accounting():
    """bench.py's three FLOP counts per NeRF sample: the reference's arithmetic (593 408 MAC), the folded
    network without padding (roofline.frac's basis: never above the peak) and the MFMAs the kernels issue."""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    f = bench.nerf_flop_per_sample(8, 256, 4, "bf16")
    assert f["reference"] == 2 * 593_408 and f["useful"] == 2 * 527_872 and f["executed"] == 2 * 535_040
    assert bench.nerf_flop_per_sample(8, 256, 4, "f16x3")["executed"] == 3 * f["executed"]
    f32 = bench.nerf_flop_per_sample(8, 256, 4, "f32")
    assert f32["useful"] == f["useful"] <= f32["executed"] <= f["reference"]
    blk = bench.roofline_block("bf16", 27.97, 640_000, 64, 8, 256, 4)
    assert 0.60 < blk["frac"] < 0.63 and blk["frac"] < blk["executed_mfma_frac"] < blk["effective_frac_on_reference_flops"] <= 1.0
