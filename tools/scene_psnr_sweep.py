"""Scene PSNR of the fitted scene over whole frames and poses, per operand type (MI355X).

PSNR against the analytic ground truth (nerf_sampling_amd/analytic_scene.py) of full 800x800 frames, DepthNet + 64
samples/ray, for the fp32 HIP path (which matches the CPU oracle to > 100 dB: tests/test_scene_psnr.py) and the 16-bit
paths, incl. the mixed set-up "bf16 radiance field + f16 DepthNet".  Prints one JSON line: per pose and dtype the PSNR
and its delta to fp32; the reference computes its PSNR per image (nerf_utils.py:306-336), so this is the figure
north_star's 0.05 dB bar applies to."""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nerf_sampling_amd import analytic_scene, ops, synthetic

dev = torch.device("cuda", 0)
_c, fine, dn, _p = bench.build_modules("shapes_fit", dev)
H = W = 800
_, K = synthetic.blender_intrinsics(H, W)
poses = synthetic.render_poses(40)[:, :3, :4]
combos = {"f32": ("f32", "f32"), "f16x3": ("f16x3", "f16x3"), "f16": ("f16", "f16"), "bf16": ("bf16", "bf16"),
          "bf16+f16dn": ("bf16", "f16"), "bf16+f16x3dn": ("bf16", "f16x3")}
sel = [int(a) for a in sys.argv[1:]] or list(range(0, 40, 4))
out = {}
for k in sel:
    gt = analytic_scene.frame(H, W, K, poses[k], device="cuda")[0].reshape(-1, 3)
    row = {}
    for name, (nd, dd) in combos.items():
        rgb = ops.render_rays_depthnet(dn.packed(dd), fine.packed(nd), camera=(H, W, K, poses[k], 0, H), n_samples=64,
                                       mode="uniform", std=0.1, device=dev)["rgb"]
        row[name] = -10 * math.log10(float(((rgb - gt) ** 2).mean()))
    out[k] = {n: round(v, 4) for n, v in row.items()}
    out[k]["delta"] = {n: round(row[n] - row["f32"], 4) for n in combos if n != "f32"}
worst = {n: max(abs(out[k]["delta"][n]) for k in sel) for n in combos if n != "f32"}
print(json.dumps({"poses": sel, "per_pose": out, "worst_abs_delta_db": worst}))
