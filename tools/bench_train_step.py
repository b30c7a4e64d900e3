"""Time one DepthNet training step (Trainer.core_optimization_loop) at the reference's batch size
(N_rand = 1024 rays, production networks) on one MI355X.  Prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_sampling_amd import ops, synthetic
from nerf_sampling_amd.autograd import HipAdam
from nerf_sampling_amd.depth_net import DepthNet
from nerf_sampling_amd.run_nerf_helpers import NeRF, get_embedder
from nerf_sampling_amd.trainers import DepthNetTrainer

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
graph = "--graph" in sys.argv
ops.set_compute_dtype(dtype)
cfg, params = synthetic.SCENES["lego_synth"], synthetic.make_scene("lego_synth")
nets = {}
for which in ("coarse", "fine"):
    n = NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
    n.load_state_dict(params[which]); n = n.cuda()
    for p in n.parameters(): p.requires_grad_(False)
    nets[which] = n
dn = DepthNet(hidden_sizes=[256] * 10, cat_hidden_sizes=[256] * 10); dn.load_state_dict(params["depth"]); dn = dn.cuda()
tr = DepthNetTrainer(dataset_type="blender", basedir="/tmp", expname="b", no_batching=True, datadir="", half_res=True,
                     white_bkgd=True, N_importance=128, N_samples=64, use_viewdirs=True, input_dims_embed=3, device="cuda",
                     perturb=1.0)
e1, _ = get_embedder(10, 0, 3); e2, _ = get_embedder(4, 0, 3)
q = lambda i, v, f: tr.run_network(i, v, f, embed_fn=e1, embeddirs_fn=e2)
kw = dict(network_query_fn=q, perturb=1.0, N_importance=128, network_fine=nets["fine"], N_samples=64, network_fn=nets["coarse"],
          use_viewdirs=True, white_bkgd=True, raw_noise_std=0.0, trainer=tr, lindisp=True, depth_network=dn,
          model_mode="train", near=2.0, far=6.0, ndc=False)
H = W = 400
_, K = synthetic.blender_intrinsics(H, W)
tr.H, tr.W, tr.K = H, W, K
o, d, _ = ops.get_rays(H, W, K, synthetic.pose_spherical(30.0, -30.0, 4.0)[:3, :4])
opt = HipAdam(list(dn.parameters()), lr=1e-4)
g = torch.Generator().manual_seed(0)
run = tr.graphed_optimization_loop(opt, kw) if graph else (lambda rays, i, tgt: tr.core_optimization_loop(opt, kw, rays, i, tgt))
idxs = [torch.randint(0, H * W, (1024,), generator=g).cuda() for _ in range(8)]
tgts = [torch.rand(1024, 3, generator=g).cuda() for _ in range(8)]
def step(i):
    idx = idxs[i % 8]
    return run(torch.stack([o[idx], d[idx]], 0), i, tgts[i % 8])
for i in range(5): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter(); K_ = 30
for i in range(K_): loss = step(i)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K_
print(json.dumps({"metric": "DepthNet training step (1024 rays: frozen 64+128 NeRF pass + DepthNet fwd/bwd + Adam)",
                  "hip_graph": graph, "ms_per_iter": 1e3 * dt, "iters_per_s": 1 / dt, "rays_per_s": 1024 / dt, "dtype_frozen_nerf": dtype}))
