"""Where does the 16-bit frame differ from the oracle?  (run on the GPU box; prints only)

For a band of an 800x800 / N = 64 frame (BASELINE configs[1] shape) and each 16-bit dtype:
  1. DepthNet z: HIP vs oracle.
  2. NeRF raw at the HIP path's own sample points: HIP vs oracle  -> the measured sigma / rgb noise.
  3. composited colour: HIP vs oracle end to end; HIP vs "oracle given the HIP z" (isolates MLP + compositing).
  4. the last-sample rule (sampling_trainer.py:176-180: dist_last = 1e10 => alpha_last = step(sigma_last)):
     rays whose sign(sigma_last) differs between HIP and oracle, and what is left when the HIP raw is composited
     with the ORACLE's sigma_last (if that matches the oracle, the step rule is the whole story).
  5. the conditioning mask of tests/test_gpu_render.py at the measured noise.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from conftest import _make_modules  # noqa: E402
from nerf_sampling_amd import ops  # noqa: E402
from oracle import nerf_oracle as O  # noqa: E402


def psnr(a, b):
    mse = float(((a - b) ** 2).mean())
    return float("inf") if mse == 0 else -10 * np.log10(mse)


def main():
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    scene = sys.argv[1] if len(sys.argv) > 1 else "lego_synth"
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    m = _make_modules(scene)
    p = m["params"]
    H = W = 800
    N = 64
    _, K = O.blender_intrinsics(H, W)
    c2w = O.pose_spherical(30.0, -30.0, 4.0)[:3, :4]
    r0 = H // 2 - rows // 2
    batch, o, d, _ = O.ray_batch_from_camera(H, W, K, c2w, 2.0, 6.0)
    sl = slice(r0 * W, (r0 + rows) * W)
    batch, o, d = batch[sl], o[sl], d[sl]
    view = batch[:, -3:]
    with torch.no_grad():
        z_ref = O.depthnet_forward(p["depth"], o, d)
        pts_ref, zz_ref = O.place_samples(o, d, z_ref, N, "uniform", 0.1)
        raw_ref = O.run_network(p["fine"], pts_ref, view)
        rgb_ref, disp_ref, acc_ref, _, _, _, w_ref = O.raw2outputs(raw_ref, zz_ref, d, 0.0, True)
    T_last = (1.0 - acc_ref + w_ref[:, -1]).clamp(min=0)  # transmittance reaching the last sample (approx.)
    print(f"{scene}: {o.shape[0]} rays; oracle sigma: max|.| {float(raw_ref[..., 3].abs().max()):.3g}, "
          f"rms {float(raw_ref[..., 3].pow(2).mean().sqrt()):.3g}; acc mean {float(acc_ref.mean()):.3f}")
    for dt in ("f32", "f16", "bf16"):
        dn, nf = m["depth"].packed(dt), m["fine"].packed(dt)
        oc, dc, vc = o.cuda(), d.cuda(), view.cuda()
        z = ops.depthnet_forward(dn, oc, dc)
        ez = (z.cpu() - z_ref).abs()
        print(f"[{dt}] depthnet z: rms {float(ez.pow(2).mean().sqrt()):.3e} max {float(ez.max()):.3e}")
        pts, zz = ops.place_samples(oc, dc, z, N, "uniform", 0.1)
        raw = ops.nerf_forward_rays(nf, oc, dc, zz, vc).cpu()
        with torch.no_grad():
            pts_o, _ = O.place_samples(o, d, z.cpu(), N, "uniform", 0.1)
            raw_o = O.run_network(p["fine"], pts_o, view)            # oracle MLP at the HIP path's points
            rgb_o_given_z = O.raw2outputs(raw_o, zz.cpu(), d, 0.0, True)[0]
        esig = raw[..., 3] - raw_o[..., 3]
        ergb = raw[..., :3] - raw_o[..., :3]
        sig_rms = float(esig.pow(2).mean().sqrt())
        print(f"[{dt}] raw at own points: sigma err rms {sig_rms:.3e} max {float(esig.abs().max()):.3e} "
              f"(p99 {float(esig.abs().flatten().kthvalue(int(0.99 * esig.numel())).values):.3e}); "
              f"rgb-logit err rms {float(ergb.pow(2).mean().sqrt()):.3e} max {float(ergb.abs().max()):.3e}")
        out = ops.render_rays_depthnet(dn, nf, rays=(oc, dc, vc), n_samples=N, mode="uniform", std=0.1)
        rgb = out["rgb"].cpu()
        e_end = (rgb - rgb_ref).abs().max(-1).values
        e_mlp = (rgb - rgb_o_given_z).abs().max(-1).values
        print(f"[{dt}] rgb end-to-end vs oracle:     PSNR {psnr(rgb, rgb_ref):.2f} dB  median {float(e_end.median()):.2e} "
              f"frac>1e-2 {float((e_end > 1e-2).float().mean()):.4f} frac>1e-3 {float((e_end > 1e-3).float().mean()):.4f}")
        print(f"[{dt}] rgb vs oracle given HIP z:     PSNR {psnr(rgb, rgb_o_given_z):.2f} dB  median {float(e_mlp.median()):.2e} "
              f"frac>1e-2 {float((e_mlp > 1e-2).float().mean()):.4f}")
        print(f"[{dt}] oracle given HIP z vs oracle:  PSNR {psnr(rgb_o_given_z, rgb_ref):.2f} dB")
        # 4. last-sample rule
        flip = (raw[:, -1, 3] > 0) != (raw_o[:, -1, 3] > 0)
        raw_fix = raw.clone()
        raw_fix[:, -1, 3] = raw_o[:, -1, 3]
        with torch.no_grad():
            rgb_fix = O.raw2outputs(raw_fix, zz.cpu(), d, 0.0, True)[0]
        e_fix = (rgb_fix - rgb_o_given_z).abs().max(-1).values
        print(f"[{dt}] sign(sigma_last) flipped on {float(flip.float().mean()):.4f} of rays; of the rays with err>1e-2 "
              f"{float(flip[e_mlp > 1e-2].float().mean()) if (e_mlp > 1e-2).any() else float('nan'):.4f} are flipped")
        print(f"[{dt}] HIP raw composited with the ORACLE's sigma_last: PSNR {psnr(rgb_fix, rgb_o_given_z):.2f} dB "
              f"max {float(e_fix.max()):.2e} frac>1e-2 {float((e_fix > 1e-2).float().mean()):.4f}")
        print(f"[{dt}] PSNR on un-flipped rays: {psnr(rgb[~flip], rgb_o_given_z[~flip]):.2f} dB (vs oracle given z), "
              f"{psnr(rgb[~flip], rgb_ref[~flip]):.2f} dB (end to end)")
        # 5. conditioning mask at k x measured sigma noise
        for k in (1.0, 2.0, 3.0):
            ill = torch.zeros(raw_ref.shape[0], dtype=torch.bool)
            with torch.no_grad():
                for sgn in (-1.0, 1.0):
                    pert = raw_ref.clone()
                    pert[..., 3] += sgn * k * sig_rms
                    ill |= (O.raw2outputs(pert, zz_ref, d, 0.0, True)[0] - rgb_ref).abs().max(-1).values > 1e-2
            well = ~ill
            print(f"[{dt}] mask k={k:.0f}: ill {float(ill.float().mean()):.4f}; well-conditioned PSNR "
                  f"{psnr(rgb[well], rgb_ref[well]):.2f} dB max {float(e_end[well].max()):.2e}; "
                  f"err>1e-2 inside ill {float((e_end[ill] > 1e-2).float().mean()) if ill.any() else float('nan'):.3f}, "
                  f"outside {float((e_end[well] > 1e-2).float().mean()):.4f}")


if __name__ == "__main__":
    main()
