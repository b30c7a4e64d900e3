"""Same-box A/B of the one-kernel renderer against the five-launch chain at the configs[4] shape (1600 x 1600, 192 samples/ray, f16:
a ray = three 64-sample chunks on different waves): ms per frame, two rounds.  python tools/ab_multichunk.py"""
import sys, os, time, json
sys.path.insert(0, os.getcwd())
import torch, bench
from nerf_sampling_amd import ops, synthetic
from nerf_sampling_amd.parallel import hip_row_renderer
dev = torch.device("cuda", 0)
_c, fine, dn, _p = bench.build_modules("shapes_fit", dev)
H = W = 1600
_, K = synthetic.blender_intrinsics(H, W)
poses = synthetic.render_poses(40)[:, :3, :4]
nw, dw = fine.packed("f16"), dn.packed("f16")
def sync(): torch.cuda.synchronize()
for rep in range(2):
    for label, one, tiles in (("fused, five tiles", None, 0), ("fused, four tiles", None, 4), ("chain", False, 0)):
        with ops.debug_switch(prod_tiles=tiles):
            t = bench.Timed(H, W, hip_row_renderer(dw, nw, H, W, K, 192, "uniform", 0.1, device=dev, events=[], one_kernel=one), [], dev)
            el, _, _ = t.run(poses, 3, 1, sync)
        print(label, round(1e3 * el / 3, 2), flush=True)
