"""Which component of the PSNR guard buys what, per pose of the fitted scene: whole-frame and band scene-PSNR deltas to fp32 arithmetic and
PSNR(build vs fp32) for the bf16 field with an f16 / f16m / f16x3 DepthNet, with and without the sigma_last guard."""
import json, math, os, sys
sys.path.insert(0, os.getcwd())
import torch
import bench
from nerf_sampling_amd import analytic_scene, ops, synthetic
dev = torch.device("cuda", 0)
_c, fine, dn, _p = bench.build_modules("shapes_fit", dev)
H = W = 800
_, K = synthetic.blender_intrinsics(H, W)
poses = synthetic.render_poses(40)[:, :3, :4]
combos = {"f32": ("f32", "f32", False), "plain": ("bf16", "f16", False), "sig_only": ("bf16", "f16", True), "dn_only": ("bf16", "f16x3", False), "full": ("bf16", "f16x3", True),
          "mix_only": ("bf16", "f16m", False), "mix_full": ("bf16", "f16m", True)}
out = {}
for k in (0, 3, 7, 13, 21, 34):
    gt = analytic_scene.frame(H, W, K, poses[k], device="cuda")[0].reshape(-1, 3)
    row = {}; band = {}
    for name, (nd, dd, g) in combos.items():
        rgb = ops.render_rays_depthnet(dn.packed(dd), fine.packed(nd), camera=(H, W, K, poses[k], 0, H), n_samples=64,
                                       mode="uniform", std=0.1, device=dev, guard=fine.packed("f16x3") if g else None)["rgb"]
        row[name] = -10 * math.log10(float(((rgb - gt) ** 2).mean()))
        b = slice(375 * W, 425 * W)
        band[name] = -10 * math.log10(float(((rgb[b] - gt[b]) ** 2).mean()))
        if name == "f32": ref = rgb
        else: row[name + "_vs_f32_db"] = -10 * math.log10(float(((rgb - ref) ** 2).mean()))
    out[k] = {"frame_delta": {n: round(row[n] - row["f32"], 4) for n in combos if n != "f32"}, "band_delta": {n: round(band[n] - band["f32"], 4) for n in combos if n != "f32"},
              "vs_f32_db": {n: round(row[n + "_vs_f32_db"], 2) for n in combos if n != "f32"}, "f32": round(row["f32"], 3), "f32_band": round(band["f32"], 3)}
    print(k, json.dumps(out[k]), flush=True)
# sigma_last distribution (bf16) on pose 3
o, d, view = ops.get_rays(H, W, K, poses[3])[:3]
mean = ops.depthnet_forward(dn.packed("f16"), o, d)
pts, z = ops.place_samples(o, d, mean, 64, "uniform", 0.1)
raw = ops.nerf_forward_rays(fine.packed("bf16"), o, d, z[:, -1:].contiguous(), view)
s = raw[:, 0, 3]
for thr in (1, 2, 4, 8, 16, 32):
    print("frac |sigma_last| <", thr, float((s.abs() < thr).float().mean()))
