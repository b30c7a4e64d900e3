"""Which layers of the fitted DepthNet lose its depth under fp16 operands: CPU emulation -- operands of chosen trunk layers rounded to fp16,
fp32 accumulation -- on 10 000 rays of the fitted scene (oracle weights).  python tools/depthnet_layer_sensitivity.py"""
import sys, torch, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
from oracle import nerf_oracle as O
from nerf_sampling_amd import synthetic
torch.set_num_threads(8)
_c, fine, dn, params = bench.build_modules("shapes_fit", torch.device("cpu"))
p = params["depth"]
H=W=800
_, K = synthetic.blender_intrinsics(H, W)
poses = synthetic.render_poses(40)[:, :3, :4]
batch,_,_,_ = O.ray_batch_from_camera(H, W, K, poses[3], 2.0, 6.0)
rb = batch[375*W:425*W:4]   # 10000 rays
o, d = rb[:,0:3], rb[:,3:6]
n_branch, n_trunk = O.depthnet_layer_counts(p)
def r16(x): return x.half().float()
def lin(name, x, q):
    w, b = p[name+".weight"], p[name+".bias"]
    if q: return r16(x) @ r16(w).t() + b
    return x @ w.t() + b
def run(qset):
    with torch.no_grad():
        z, parts = O.depthnet_forward(p, o, d, return_parts=True)
        y = torch.cat([parts[k] for k in ("h_o","h_d","h_x","e_o","e_d","e_x")], -1).double().float()
        # folded front is fp64-composed in the kernel; emulate layer 0 on the literal input
        for i in range(n_trunk):
            y = torch.nn.functional.leaky_relu(lin(f"cat_layers.{2*i}", y, i in qset), 0.01)
        depth = torch.sigmoid(lin("to_depth.0", y, "out" in qset))
        return 2.0*(1-depth)+6.0*depth, z
zq, zref = run(set(range(n_trunk))|{"out"})
print("all f16: rms", float((zq-zref).pow(2).mean().sqrt()), "max", float((zq-zref).abs().max()))
for i in list(range(n_trunk))+["out"]:
    zq,_ = run({i}); print("only layer", i, "f16: rms", float((zq-zref).pow(2).mean().sqrt()))
for k in (1,2,3,5):
    zq,_ = run(set(range(k, n_trunk))|{"out"}); print(f"first {k} layers exact, rest f16: rms", float((zq-zref).pow(2).mean().sqrt()))
    zq,_ = run(set(range(0, n_trunk-k))); print(f"last {k} layers+out exact, rest f16: rms", float((zq-zref).pow(2).mean().sqrt()))
