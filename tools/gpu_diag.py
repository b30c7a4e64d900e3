"""Error statistics of each stage with identical inputs (run on the GPU box; prints only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import nerf_oracle as O
from nerf_sampling_amd import ops
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import _make_modules

def stats(name, a, b, scale=None):
    a = a.detach().cpu().double().numpy(); b = b.detach().cpu().double().numpy()
    ok = ~np.isnan(b)
    err = np.abs(a - b)[ok]
    sc = np.abs(b[ok]).max() if scale is None else scale
    print(f"{name:42s} max {err.max():.3e} rms {np.sqrt((err**2).mean()):.3e} median {np.median(err):.3e}  (scale {sc:.3g}, max/scale {err.max()/sc:.2e})")

for scene in ("tiny_synth", "lego_synth"):
    m = _make_modules(scene); p = m["params"]
    H = W = 64
    _, K = O.blender_intrinsics(H, W)
    c2w = O.pose_spherical(30.0, -30.0, 4.0)[:3, :4]
    batch, o, d, _ = O.ray_batch_from_camera(H, W, K, c2w, 2.0, 6.0)
    go, gd, gv, gb = ops.get_rays(H, W, K, c2w, near=2.0, far=6.0, want_batch=True)
    print(scene, "rays bit-exact:", torch.equal(gb.cpu(), batch))
    z_ref = O.depthnet_forward(p["depth"], o, d)
    for dt in ("f32", "f16", "bf16"):
        z = ops.depthnet_forward(m["depth"].packed(dt), o.cuda(), d.cuda())
        stats(f"depthnet z [{dt}]", z, z_ref)
    pts, zz = O.place_samples(o, d, z_ref, 64, "uniform", 0.1)
    raw_ref = O.run_network(p["fine"], pts, batch[:, -3:])
    for dt in ("f32", "f16", "bf16"):
        raw = ops.nerf_forward(m["fine"].packed(dt), pts.cuda(), batch[:, -3:].cuda())
        stats(f"nerf raw rgb [{dt}]", raw[..., :3], raw_ref[..., :3])
        stats(f"nerf raw sigma [{dt}]", raw[..., 3], raw_ref[..., 3])
        out = ops.raw2outputs(raw, zz.cuda(), d.cuda(), None, True)
        ref = O.raw2outputs(raw_ref, zz, d, 0.0, True)
        stats(f"  -> rgb_map via own raw [{dt}]", out[0], ref[0], 1.0)
        e = (out[0].cpu() - ref[0]).abs().max(-1).values.numpy()
        print(f"     rays with rgb err >1e-4: {np.mean(e>1e-4):.4f}  >1e-3: {np.mean(e>1e-3):.4f}  PSNR {-10*np.log10(((out[0].cpu()-ref[0])**2).mean().item()+1e-30):.1f} dB")
    # end to end, fused
    for dt in ("f32", "f16", "bf16"):
        out = ops.render_rays_depthnet(m["depth"].packed(dt), m["fine"].packed(dt), camera=(H, W, K, c2w, 0, H),
                                       n_samples=64, mode="uniform", std=0.1)
        ref = O.render_frame(H, W, K, c2w, 1 << 15, 2.0, 6.0, p_coarse=p["coarse"], p_fine=p["fine"], p_depth=p["depth"],
                             n_depth_samples=64, sampling_mode="uniform", distance=0.1)
        e = (out["rgb"].cpu() - ref[0].reshape(-1, 3)).abs().max(-1).values.numpy()
        mse = ((out["rgb"].cpu() - ref[0].reshape(-1, 3)) ** 2).mean().item()
        print(f"end-to-end [{dt}] rgb: max {e.max():.3e} median {np.median(e):.3e} frac>1e-4 {np.mean(e>1e-4):.4f} frac>1e-3 {np.mean(e>1e-3):.4f} PSNR {-10*np.log10(mse+1e-30):.1f} dB")
