#!/usr/bin/env python3
"""Headline benchmark: rays/sec at 800x800, DepthNet + 64 samples/ray (BASELINE.json configs[1]); PSNR vs reference.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N rank processes, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one full 800x800 frame of the reference's spiral render path (pose k of 40): ray generation,
DepthNet, sample placement, NeRF MLP and compositing, all on the device, plus -- for N > 1 -- the single
all-gather that assembles the frame.  Frames are row-sharded over the N ranks (total work fixed: strong
scaling).  Inputs are 16 camera scalars, so nothing is staged from the host inside the timed region.

Scenes (no dataset / checkpoint ships with the reference): "shapes_fit" (default) = networks FITTED to the analytic
ground-truth scene of nerf_sampling_amd/analytic_scene.py (tools/fit_scene.py; weights under tests/golden/fitted_scene):
the realistic case, the only one a scene PSNR exists for; "lego_synth" = seeded random weights (a conditioning stress
test: its density crosses zero on a fifth of the rays).  The timed arithmetic is identical for both.

Rank 0 prints ONE JSON line.  Beside the contract's fields it carries: `roofline` (dominant kernel, HIP events inside the
timed region), `cpu_baseline` (the oracle on the host cores over a band of the same frame), the accuracy of the timed
dtype against that same oracle pass (all rays first; then diagnostics that isolate the reference's last-sample step rule),
`scene_psnr` (oracle vs build against the analytic ground truth: north_star's 0.05 dB bar), `other_configs` (the
parity-grade dtypes and the other BASELINE shapes, a few steps each) and `api_path` (the mirrored reference API with its
host copies).  None of those legs runs inside the timed region.
"""

import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32", "f16x3"])
    ap.add_argument("--size", type=int, default=800)
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--scene", default="shapes_fit", choices=["shapes_fit", "lego_synth"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=50, help="rows of the frame timed on the CPU (x 800 rays)")
    ap.add_argument("--mode", default="depthnet", choices=["depthnet", "full_nerf"],
                    help="depthnet = BASELINE configs[1] (headline); full_nerf = configs[2], vanilla 64+128 coarse+fine")
    ap.add_argument("--api-path", dest="api_path", action="store_true", default=True,
                    help="also time nerf_utils.render_test (the mirrored reference API with per-chunk host copies)")
    ap.add_argument("--no-api-path", dest="api_path", action="store_false")
    ap.add_argument("--other-configs", dest="other_configs", action="store_true", default=True,
                    help="also run f16x3 / f32 at the headline shape, configs[2] (vanilla 64+128) and the configs[4] shape")
    ap.add_argument("--no-other-configs", dest="other_configs", action="store_false")
    ap.add_argument("--chain", action="store_true", help="render with the five-launch chain (ns_render_rays_depthnet) instead of "
                    "the one-kernel renderer (ns_render_rays_fused): same bits, for A/B timing")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                    "the multi-rank path on a one-GPU box, all ranks sharing cuda:0)")
    return ap.parse_args(argv)


def _launch_ranks(n: int, port: int):
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), NS_BENCH_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr,
                                      stderr=subprocess.PIPE if r == 0 else None, text=(r == 0)))
    return procs


def kernel_sources_sha256():
    """identity of the kernel sources (csrc/*): ties profiles/*_traffic_*.json to the build it was measured on"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "nerf_sampling_amd", "csrc", "*.*"))):
        if f.endswith((".hip", ".h", ".cpp", ".inc")):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without an outer launcher: start N fresh rank processes (one per GPU, RCCL rendezvous
    on 127.0.0.1) and relay rank 0's JSON line.  The parent never touches the GPU (it has not even imported torch), so
    nothing is exec'ed or forked from a process with an initialised HIP runtime.  Any failing rank fails the run.
    Rank 0's pipes are drained while the ranks run (a full pipe would block it inside a collective); a rendezvous port
    lost to another process between probing and binding is retried on a fresh port; a failed run ends the surviving ranks
    with SIGTERM first, SIGKILL only after a grace period."""
    for attempt in range(3):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        procs = _launch_ranks(n, port)
        out0, err0 = [], []
        readers = [threading.Thread(target=lambda f, sink: sink.extend(f), args=(procs[0].stdout, out0), daemon=True),
                   threading.Thread(target=lambda f, sink: sink.extend(f), args=(procs[0].stderr, err0), daemon=True)]
        for t in readers:
            t.start()
        failed = False
        while any(p_.poll() is None for p_ in procs):
            if any(p_.poll() not in (None, 0) for p_ in procs):
                failed = True
                deadline = time.time() + 5.0          # let the others report and leave their collectives
                for p_ in procs:
                    if p_.poll() is None:
                        p_.terminate()
                while time.time() < deadline and any(p_.poll() is None for p_ in procs):
                    time.sleep(0.05)
                for p_ in procs:
                    if p_.poll() is None:
                        p_.kill()
                break
            time.sleep(0.05)
        rcs = [p_.wait() for p_ in procs]
        for t in readers:
            t.join(timeout=5.0)
        err_text = "".join(err0)
        sys.stderr.write(err_text)
        if failed and attempt < 2 and ("EADDRINUSE" in err_text or "Address already in use" in err_text
                                       or "address already in use" in err_text):
            sys.stderr.write(f"bench.py: rendezvous port {port} was taken, retrying on a new one\n")
            continue
        for line in out0:          # the JSON line to stdout, library chatter (gloo prints there) to stderr
            (sys.stdout if line.startswith("{") else sys.stderr).write(line if line.endswith("\n") else line + "\n")
        sys.stdout.flush()
        if failed or any(rcs):
            sys.stderr.write(f"bench.py: rank exit codes {rcs}\n")
            return 1
        return 0
    return 1


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ:
    _a = parse_args()
    if _a.gpus > 1:
        sys.exit(spawn_ranks(_a.gpus))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3,  # dense MFMA, MI355X_MICROARCH.md
               "f16x3": 2500.0}  # split fp16 operands run on the fp16 MFMA pipe (3 MFMAs per product term: see executed_*)
MEASURED_MFMA_CEILING = {"bf16": 2140.0, "f16": None, "f32": 155.0}   # TFLOP/s (bf16: bare 16x16x32 loop), DESIGN.md section 6
KERNEL_NAME = {"f32": "nerf_mlp_kernel", "f16x3": "nerf_mlp_x3_kernel"}
DEPTHNET_FLOP_PER_RAY = 2 * 3_330_304      # reference arithmetic, SURVEY.md section 8(a) a4


def nerf_flop_per_sample(D, W, skip, dtype):
    """FLOPs of one NeRF-MLP sample, three ways:
      reference -- the reference's arithmetic (593 408 MAC at 8x256), no padding credit;
      useful    -- what the kernels compute after the pack-time fold of feature_linear into views_linears
                   (DESIGN.md section 4.0), WITHOUT padding: the basis of roofline.achieved / frac (<= 1 by construction);
      executed  -- MFMAs issued x FLOP per MFMA, zero padding included (K 63->64, 319->320, 283->288, the 16-row alpha /
                   rgb blocks) and x3 for the split-operand path: matrix-pipe utilisation."""
    trunk = 63 * W + sum((W + (63 if l - 1 == skip else 0)) * W for l in range(1, D))
    reference = trunk + W + W * W + (W + 27) * (W // 2) + (W // 2) * 3
    useful = reference - W * W
    if dtype == "f32":
        nb = W // 32
        blocks = nb * 2 + sum(nb * (nb + (2 if l - 1 == skip else 0)) for l in range(1, D))
        blocks += nb + (nb // 2) * (nb + 1) + nb // 2            # alpha, views o feature, rgb
        executed = blocks * 32 * 32                              # one 32-row x 32-feature block per sample column
    else:
        nsb, nkb = W // 16, W // 32
        chunks = nsb * 2 + sum(nsb * (nkb + (2 if l - 1 == skip else 0)) for l in range(1, D))
        chunks += (nsb // 2 + 1) * (nkb + 1) + nkb // 2
        executed = chunks * 16 * 32 * (3 if dtype == "f16x3" else 1)   # a chunk = 16 rows x 32 features
    return {"reference": 2 * reference, "useful": 2 * useful, "executed": 2 * executed}


def build_modules(scene_name, device):
    from nerf_sampling_amd import synthetic
    from nerf_sampling_amd.depth_net import DepthNet
    from nerf_sampling_amd.run_nerf_helpers import NeRF

    cfg, params = synthetic.SCENES[scene_name], synthetic.make_scene(scene_name)
    nets = {}
    for which in ("coarse", "fine"):
        net = NeRF(D=cfg[which]["D"], W=cfg[which]["W"], input_ch=63, input_ch_views=27, output_ch=5, skips=[4],
                   use_viewdirs=True)
        net.load_state_dict(params[which])
        if os.environ.get("NS_BENCH_ZERO_NERF"):   # diagnostic only: all-zero MLP operands (is the kernel power-limited?)
            with torch.no_grad():
                for p_ in net.parameters():
                    p_.zero_()
        nets[which] = net.to(device)
    n, w = cfg["depth"]["n_layers"], cfg["depth"]["width"]
    dn = DepthNet(hidden_sizes=[w] * n, cat_hidden_sizes=[w] * n, sphere_radius=2.0)
    dn.load_state_dict(params["depth"])
    return nets["coarse"], nets["fine"], dn.to(device), params


def host_cores() -> int:
    """CPU cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(n, 32)  # one GPU's share of the host; more threads than that only oversubscribes


def psnr(a, b):
    mse = float(((a - b) ** 2).mean())
    return None if mse == 0 else -10.0 * float(np.log10(mse))


def cpu_baseline(params, H, W, K, c2w, n_samples, rows, budget_s=24.0):
    """The oracle (CPU port of the reference path) timed on the host cores over a band of rows of the same frame with
    the reference's own chunk structure: chunk = 32768 rays per render_rays_test call (Trainer.py:31), the NeRF MLP in
    netchunk = 65536-row slices (Trainer.py:36).  Each chunk runs exactly what nerf_oracle.render_rays_test runs
    (depthnet_forward -> place_samples -> run_network -> raw2outputs); the per-chunk pieces are kept so the accuracy
    leg below can build its diagnostics from the same pass instead of a second one."""
    from oracle import nerf_oracle as O

    cores = host_cores()
    torch.set_num_threads(cores)
    batch, _, _, _ = O.ray_batch_from_camera(H, W, K, c2w, 2.0, 6.0)
    batch = batch[rows[0] * W : rows[1] * W]
    chunk, done, parts = 32768, 0, []
    with torch.no_grad():
        O.render_rays_test(batch[:1024], p_coarse=params["coarse"], p_fine=params["fine"], p_depth=params["depth"],
                           n_depth_samples=n_samples, sampling_mode="uniform", distance=0.1)  # warm the thread pool
        t0 = time.perf_counter()
        while done < batch.shape[0] and time.perf_counter() - t0 < budget_s:
            rb = batch[done : done + chunk]
            o, d, view = rb[:, 0:3], rb[:, 3:6], rb[:, -3:]
            mean = O.depthnet_forward(params["depth"], o, d)
            pts, z = O.place_samples(o, d, mean, n_samples, "uniform", 0.1)
            raw = O.run_network(params["fine"], pts, view, netchunk=1024 * 64)
            rgb = O.raw2outputs(raw, z, d, 0.0, True)[0]
            parts.append((raw, z, d, rgb, o, view))
            done = min(batch.shape[0], done + chunk)
        dt = time.perf_counter() - t0
    ref = {k: torch.cat([p_[i] for p_ in parts], 0) for i, k in enumerate(("raw", "z", "d", "rgb", "o", "view"))}
    ref["rows"] = (rows[0], rows[0] + done // W)     # whole rows the oracle finished
    return {"value": done / dt, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": f"{done} rays from rows {rows[0]}-{rows[1]} of the same {H}x{W} frame, fp32 torch-CPU oracle in the "
                      f"reference's chunks (32768 rays / 65536 MLP rows), {dt:.1f} s"}, ref


def step_rule_mask(O, raw, z, d, rgb_ref, eps):
    """Rays whose ORACLE colour moves by > 1e-2 when sigma of the LAST sample alone moves by +-eps: the reference composites
    that sample with dist = 1e10 (sampling_trainer.py:176-180), alpha_last = step(sigma_last), so colour is discontinuous in
    sigma_last wherever transmittance is left."""
    ill = torch.zeros(raw.shape[0], dtype=torch.bool)
    with torch.no_grad():
        for sgn in (-1.0, 1.0):
            pert = raw.clone()
            pert[:, -1, 3] += sgn * eps
            ill |= (O.raw2outputs(pert, z, d, 0.0, True)[0] - rgb_ref).abs().max(-1).values > 1e-2
    return ill


def accuracy_vs_oracle(ops, depth_w, nerf_w, ref, H, W, K, c2w, n_samples, dtype, device, pose_k, scene):
    """Second half of the cpu_baseline leg (the oracle as CHECKER, outside the timed region, rank 0 at N = 1 only).

    Primary figure: PSNR(build || oracle) over ALL rays of the band the CPU leg rendered, plus the error quantiles.
    Diagnostics, all derived from THIS run (no per-dtype constants): the HIP MLP is also run at the oracle's own sample
    points, which gives the sigma error of the kernel itself (rms, and at the last sample); the reference composites the
    last sample with dist = 1e10 (sampling_trainer.py:176-180), alpha_last = step(sigma_last), so a ray's colour is
    discontinuous in sigma_last wherever transmittance is left: `step_rule_frac` counts the rays whose ORACLE colour
    moves by > 1e-2 when sigma_last ALONE moves by 3x the measured last-sample error, and
    `psnr_mlp_with_oracle_sigma_last_db` composites the HIP raw with only sigma_last taken from the oracle."""
    from oracle import nerf_oracle as O

    r0, r1 = ref["rows"]
    n = (r1 - r0) * W
    if n == 0:
        return {}, None
    raw, z, d, rgb_ref, o, view = (ref[k][:n] for k in ("raw", "z", "d", "rgb", "o", "view"))
    rgb = ops.render_rays_depthnet(depth_w, nerf_w, camera=(H, W, K, c2w, r0, r1), n_samples=n_samples, mode="uniform",
                                   std=0.1, device=device)["rgb"].cpu()
    err = (rgb - rgb_ref).abs().max(-1).values
    out = {"psnr_vs_oracle_db": psnr(rgb, rgb_ref),
           "abs_err_vs_oracle": {"median": float(err.median()), "p99": float(err.quantile(0.99)), "max": float(err.max()),
                                 "rays_over_1e-2": float((err > 1e-2).float().mean()),
                                 "rays_over_1e-4": float((err > 1e-4).float().mean())}}
    # the MLP kernel alone, at the oracle's own points
    raw_hip = ops.nerf_forward_rays(nerf_w, o.to(device), d.to(device), z.to(device), view.to(device)).cpu()
    d_sig = raw_hip[..., 3] - raw[..., 3]
    sig_max = float(raw[..., 3].abs().max())
    last_rms = float(d_sig[:, -1].pow(2).mean().sqrt())
    with torch.no_grad():
        rgb_mlp = O.raw2outputs(raw_hip, z, d, 0.0, True)[0]
        fix = raw_hip.clone()
        fix[:, -1, 3] = raw[:, -1, 3]
        rgb_fix = O.raw2outputs(fix, z, d, 0.0, True)[0]
    ill = step_rule_mask(O, raw, z, d, rgb_ref, 3.0 * last_rms)
    mse_vs_oracle = float(((rgb - rgb_ref) ** 2).mean())
    out.update({
        # The scene PSNR at which an UNCORRELATED error of this size moves the PSNR by exactly 0.05 dB:
        # 10 log10(1 + mse_err / mse_scene) = 0.05  <=>  mse_scene = mse_err / 0.01158.  On a scene that renders BELOW this
        # figure the 0.05 dB bar cannot fail whatever the error looks like; above it, it can.
        "breakeven_scene_psnr_db": (-10.0 * float(np.log10(mse_vs_oracle / 0.011579))) if mse_vs_oracle > 0 else None,
        "sigma_err_rms_over_max_sigma": float(d_sig.pow(2).mean().sqrt()) / max(sig_max, 1e-30),
        "sigma_last_err_rms": last_rms, "max_abs_sigma": sig_max,
        "psnr_mlp_at_oracle_points_db": psnr(rgb_mlp, rgb_ref),
        "psnr_mlp_with_oracle_sigma_last_db": psnr(rgb_fix, rgb_ref),
        "step_rule_frac": float(ill.float().mean()),
        "psnr_vs_oracle_outside_step_rule_db": psnr(rgb[~ill], rgb_ref[~ill]) if (~ill).any() else None,
        "rays_over_1e-2_outside_step_rule": float(((err > 1e-2) & ~ill).float().mean()),
        "accuracy_sample": f"rows {r0}-{r1} of pose {pose_k} ({scene}) against the fp32 CPU oracle on identical rays and "
                           f"weights; step rule: oracle colour moves > 1e-2 when sigma_last alone moves by +-{3 * last_rms:.3g} "
                           f"(3x this run's rms {dtype} error of sigma_last)"})
    return out, rgb


def scene_psnr(rgb_build, ref, H, W, K, c2w, dtype):
    """north_star's acceptance bar: PSNR against GROUND TRUTH of the reference arithmetic (the oracle) and of the build,
    same rays; |delta| <= 0.05 dB.  Ground truth = the analytic scene the networks were fitted to."""
    from nerf_sampling_amd import analytic_scene

    r0, r1 = ref["rows"]
    n = (r1 - r0) * W
    gt = analytic_scene.frame(H, W, K, c2w, r0, r1)[0].reshape(-1, 3)
    p_ref, p_build = psnr(ref["rgb"][:n], gt), psnr(rgb_build, gt)
    return {"ground_truth": "nerf_sampling_amd/analytic_scene.py (exact ray cast)", "rows": [r0, r1],
            "oracle_fp32_db": p_ref, f"build_{dtype}_db": p_build, "delta_db": p_build - p_ref,
            "within_0.05_db": abs(p_build - p_ref) <= 0.05, "rays": n}


def api_path_rate(coarse, fine, dn, dtype, H, W, K, poses, n_samples, device, frames=6, blocking=False):
    """What a user of the mirrored reference API gets: nerf_utils.render_test (render_rays_test in 32768-ray chunks,
    per-sample extras, the reference's per-chunk host copies of weights / disp / z / pts, nerf_utils.py:866-870) timed
    over whole frames INCLUDING those device-to-host copies (~0.8 GB per 800x800x64 frame).  blocking=True: the copies
    as the reference issues them (`.cpu()` per chunk + host concatenation); False: this build's pinned async sink."""
    from nerf_sampling_amd import nerf_utils, ops
    from nerf_sampling_amd.run_nerf_helpers import get_embedder
    from nerf_sampling_amd.trainers import DepthNetTrainer

    ops.set_compute_dtype(dtype)
    tr = DepthNetTrainer(dataset_type="blender", basedir="/tmp", expname="bench_api", no_batching=True, datadir="",
                         half_res=False, white_bkgd=True, N_importance=128, N_samples=64, use_viewdirs=True,
                         input_dims_embed=3, device="cuda", n_depth_samples=n_samples, sampling_mode="uniform", distance=0.1)
    embed_fn, _ = get_embedder(tr.multires, tr.i_embed, 3)
    embeddirs_fn, _ = get_embedder(tr.multires_views, tr.i_embed, 3)
    # the query function exactly as nerf_utils.create_nerf builds (and tags) it for a user of the mirrored API
    query = nerf_utils.standard_query_fn(
        lambda i_, v_, f_: tr.run_network(i_, v_, f_, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, netchunk=tr.netchunk))
    kw = dict(network_query_fn=query, perturb=0.0, N_importance=128, network_fine=fine, N_samples=64, network_fn=coarse,
              use_viewdirs=True, white_bkgd=True, raw_noise_std=0.0, trainer=tr, lindisp=True, depth_network=dn,
              model_mode="test", near=2.0, far=6.0, ndc=False, _blocking_host_copies=blocking)
    with torch.no_grad():
        for i in range(2):
            rgb, disp, extras = nerf_utils.render_test(H, W, K, chunk=tr.chunk, c2w=poses[i], **kw)
        if not blocking:     # torch's pinned-memory cache takes a few frames of the same loop to reach its steady size
            nerf_utils.render_path(poses[:4], [H, W, float(K[0][0])], K, tr.chunk, kw, step=0)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        # the loop experiments/render.py runs: render_path = render_test per pose + rgb / disp to numpy (no PNGs here)
        nerf_utils.render_path(poses[4 : 4 + frames], [H, W, float(K[0][0])], K, tr.chunk, kw, step=0)
        torch.cuda.synchronize(device)
        dt = (time.perf_counter() - t0) / frames
    host_bytes = sum(v.numel() * v.element_size() for v in extras.values() if isinstance(v, torch.Tensor) and not v.is_cuda)
    ops.set_compute_dtype("f32")
    return {"rays_per_s": H * W / dt, "ms_per_frame": 1e3 * dt, "host_bytes_per_frame": host_bytes, "frames": frames}


def d2h_ceiling(device, nbytes=256 << 20, reps=4):
    """Pinned hipMemcpyAsync device-to-host bandwidth of this box (GB/s): the ceiling of the API path's host copies."""
    src = torch.empty(nbytes, dtype=torch.uint8, device=device)
    dst = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
    dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(reps):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize(device)
    return reps * nbytes / (time.perf_counter() - t0) / 1e9


class Timed:
    """One renderer (rows_fn + its event list) measured like the headline: warm-up, then K frames between syncs."""

    def __init__(self, H, W, rows_fn, events, device):
        from nerf_sampling_amd.parallel import FrameRenderer

        self.events, self.renderer, self.device = events, FrameRenderer(H, W, rows_fn, device), device

    def run(self, poses, steps, warmup, sync):
        for i in range(warmup):
            self.renderer.render(poses[i % 40])
        self.events.clear()
        sync()
        t0 = time.perf_counter()
        for i in range(steps):
            rgb, disp = self.renderer.render(poses[(warmup + i) % 40], wait=False)   # frame i's all-gather under frame i+1
        self.renderer.finish()
        sync()
        elapsed = time.perf_counter() - t0
        return elapsed, rgb, disp

    def kernel_ms(self):
        return float(np.mean([b.elapsed_ms(e) for b, e in self.events])) if self.events else float("nan")


def hier_rows_fn(ops, coarse_w, fine_w, H, W, K, device, events, max_events=128, coarse_events=None):
    """``events`` / ``coarse_events``: lists the (begin, end) hipEvent pairs of the fine-pass / coarse-pass MLP kernel of each
    call are appended to."""
    ws, ring, cring = ops.RenderWorkspace(), [], []

    def rows_fn(c2w, row0, row1, shard=None):
        if not ring:
            ring.extend((ops.Event(), ops.Event()) for _ in range(max_events))
            cring.extend((ops.Event(), ops.Event()) for _ in range(max_events if coarse_events is not None else 0))
        ev = cev = None
        if len(events) < max_events:
            ev = ring[len(events)]
            if coarse_events is not None:
                cev = cring[len(events)]
                coarse_events.append(cev)
            events.append(ev)
        out = ops.render_rays_hierarchical(coarse_w, fine_w, camera=(H, W, K, c2w, row0, row1), n_coarse=64,
                                           n_importance=128, lindisp=True, white_bkgd=True, workspace=ws,
                                           device=device, mlp_events=ev, shard=shard, coarse_events=cev)
        return out["rgb"], out["disp"]

    return rows_fn


def roofline_block(dtype, kernel_ms, rays_per_launch, samples, D, W_, skip):
    """roofline of the dominant kernel (NeRF MLP), per launch.  `achieved` / `frac` count the USEFUL arithmetic the
    kernel performs (folded network, no padding): never above the peak.  The reference's own (unfolded) arithmetic over
    the same time is `effective_*`; the MFMAs actually issued (padding, and x3 for split operands) are `executed_*`."""
    f = nerf_flop_per_sample(D, W_, skip, dtype)
    n = rays_per_launch * samples
    rate = lambda flop: n * flop / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else float("nan")  # noqa: E731
    peak = PEAK_TFLOPS[dtype]
    blk = {"bound": "mfma", "kernel": KERNEL_NAME.get(dtype, "nerf_mlp_ob16_kernel"), "achieved": rate(f["useful"]),
           "peak": peak, "unit": "TFLOP/s", "frac": rate(f["useful"]) / peak, "traffic": None, "kernel_ms": kernel_ms,
           "algorithmic_flop_per_launch": n * f["useful"],
           "flop_basis": "folded network (feature_linear composed into views_linears at pack time), no padding",
           "effective_tflops_on_reference_flops": rate(f["reference"]),
           "effective_frac_on_reference_flops": rate(f["reference"]) / peak,
           "reference_flop_per_launch": n * f["reference"],
           "executed_flop_per_launch": n * f["executed"], "executed_tflops": rate(f["executed"]),
           "executed_mfma_frac": rate(f["executed"]) / peak,
           # context, not the contract's peak: what a bare v_mfma_f32_16x16x32 loop whose A and B operands change on
           # every MFMA sustains on this part under its power cap (tools/mfma_peak.hip)
           "measured_mfma_ceiling": MEASURED_MFMA_CEILING.get(dtype)}
    assert not (blk["frac"] > 1.0), blk
    return blk


def other_configs(ops, synthetic, nets, params, ref, H, W, K, poses, pose_k, device, sync, headline_dtype, samples):
    """The parity-grade dtypes at the headline shape and the other BASELINE shapes, a few steps each (never `value`)."""
    from nerf_sampling_amd.parallel import hip_row_renderer
    from oracle import nerf_oracle as O

    coarse, fine, dn = nets
    out = []
    cfg = {"D": fine.D, "W": fine.W, "skip": 4}
    # -- f16x3 / f32 (and the other 16-bit type) at configs[1]: error against the band the cpu_baseline leg rendered
    r0, r1 = ref["rows"]
    n = (r1 - r0) * W
    from nerf_sampling_amd import analytic_scene
    gt_band = analytic_scene.frame(H, W, K, poses[pose_k], r0, r1)[0].reshape(-1, 3)
    # (label, field operands, guarded, timed steps); "guarded" = ops.set_psnr_guard: f16x3 DepthNet + every ray's last sample on f16x3
    # (label, field operands, guard = None or (DepthNet operands, threshold), timed steps).  The guard (ops.set_psnr_guard): the
    # DepthNet on "f16x3" (every layer split, the default) or "f16m" (the first three) operands + sigma of the last sample
    # re-evaluated on f16x3 -- threshold 16: only for the rays whose own sigma there is within 16 of zero, after the kernel; 0: every ray
    for label, dtype, guard, steps in ((f"{headline_dtype} + PSNR guard", headline_dtype, ("f16x3", 16.0), 5),
                                       (f"{headline_dtype} + PSNR guard, economy setting (f16m DepthNet: three of its ten layers split)",
                                        headline_dtype, ("f16m", 16.0), 4),
                                       (f"{headline_dtype} + PSNR guard on every ray", headline_dtype, ("f16x3", 0.0), 4),
                                       ("f16x3", "f16x3", None, 4), ("f32", "f32", None, 3),
                                       ("f16" if headline_dtype == "bf16" else "bf16",) * 2 + (None, 5)):
        guarded, thr = guard is not None, (guard[1] if guard else None)
        nw = fine.packed(dtype)
        dw = dn.packed(guard[0] if guarded else ops.depthnet_dtype_for(dtype))
        gw = fine.packed("f16x3") if guarded else None
        events = []
        t = Timed(H, W, hip_row_renderer(dw, nw, H, W, K, samples, "uniform", 0.1, device=device, events=events, guard=gw,
                                         guard_threshold=thr), events, device)
        elapsed, _, _ = t.run(poses, steps, 1, sync)
        rl = roofline_block(dtype, t.kernel_ms(), H * W, samples, cfg["D"], cfg["W"], cfg["skip"])
        rgb = ops.render_rays_depthnet(dw, nw, camera=(H, W, K, poses[pose_k], r0, r1), n_samples=samples, mode="uniform",
                                       std=0.1, device=device, guard=gw, guard_threshold=thr)["rgb"].cpu()
        err = (rgb - ref["rgb"][:n]).abs().max(-1).values
        mse_err = float(((rgb - ref["rgb"][:n]) ** 2).mean())
        p_ref, p_build = psnr(ref["rgb"][:n], gt_band), psnr(rgb, gt_band)
        # the same step-rule accounting as the headline: sigma_last error of THIS dtype measured at the oracle's points
        raw_o, z_o, d_o, o_o, v_o = (ref[k][:n] for k in ("raw", "z", "d", "o", "view"))
        raw_hip = ops.nerf_forward_rays(nw, o_o.to(device), d_o.to(device), z_o.to(device), v_o.to(device)).cpu()
        last_rms = float((raw_hip[:, -1, 3] - raw_o[:, -1, 3]).pow(2).mean().sqrt())
        ill = step_rule_mask(O, raw_o, z_o, d_o, ref["rgb"][:n], 3.0 * last_rms)
        out.append({"config": f"configs[1] shape ({H}x{W}, DepthNet + {samples} samples), {label}", "dtype": dtype,
                    "depthnet_operands": dw.dtype, "psnr_guard": guarded, "guard_threshold": thr,
                    "scene_psnr": {"oracle_fp32_db": p_ref, "build_db": p_build, "delta_db": p_build - p_ref,
                                   "within_0.05_db": abs(p_build - p_ref) <= 0.05},
                    "breakeven_scene_psnr_db": (-10.0 * float(np.log10(mse_err / 0.011579))) if mse_err > 0 else None,
                    "sigma_last_err_rms": last_rms, "step_rule_frac": float(ill.float().mean()),
                    "rays_over_1e-4_outside_step_rule": float(((err > 1e-4) & ~ill).float().mean()),
                    "max_abs_err_outside_step_rule": float(err[~ill].max()) if (~ill).any() else None,
                    "steps": steps, "rays_per_s": H * W * steps / elapsed, "ms_per_step": 1e3 * elapsed / steps,
                    "kernel_ms": rl["kernel_ms"], "frac": rl["frac"], "executed_mfma_frac": rl["executed_mfma_frac"],
                    "effective_frac_on_reference_flops": rl["effective_frac_on_reference_flops"],
                    "max_abs_err_vs_oracle": float(err.max()), "rays_over_1e-4": float((err > 1e-4).float().mean()),
                    "psnr_vs_oracle_db": psnr(rgb, ref["rgb"][:n]), "oracle_rays": n})
    # -- the headline shape through the five-launch chain (ns_render_rays_depthnet): the same pixels, the A/B of the one-kernel
    #    renderer on THIS box, and the MLP kernel timed alone (the one-kernel renderer's launch includes placement + compositing)
    nw, dw = fine.packed(headline_dtype), dn.packed(ops.depthnet_dtype_for(headline_dtype))
    events = []
    t = Timed(H, W, hip_row_renderer(dw, nw, H, W, K, samples, "uniform", 0.1, device=device, events=events, one_kernel=False),
              events, device)
    steps = 8
    elapsed, rgb_chain, _ = t.run(poses, steps, 2, sync)
    rl = roofline_block(headline_dtype, t.kernel_ms(), H * W, samples, cfg["D"], cfg["W"], cfg["skip"])
    t1 = Timed(H, W, hip_row_renderer(dw, nw, H, W, K, samples, "uniform", 0.1, device=device, events=[]), [], device)
    _, rgb_one, _ = t1.run(poses, steps, 2, sync)       # the same frames (same poses in the same order) through the one-kernel renderer
    out.append({"config": f"configs[1] shape ({H}x{W}, DepthNet + {samples} samples), {headline_dtype}, five-launch chain "
                          "(ns_render_rays_depthnet: placement, MLP and compositing as separate kernels)", "dtype": headline_dtype,
                "steps": steps, "rays_per_s": H * W * steps / elapsed, "ms_per_step": 1e3 * elapsed / steps,
                "kernel": "NeRF MLP alone", "kernel_ms": rl["kernel_ms"], "frac": rl["frac"],
                "executed_mfma_frac": rl["executed_mfma_frac"],
                "effective_frac_on_reference_flops": rl["effective_frac_on_reference_flops"],
                "pixels_bit_identical_to_the_one_kernel_renderer": bool(torch.equal(rgb_chain, rgb_one))})
    # -- configs[2]: vanilla hierarchical 64 + 128, coarse + fine network, headline dtype; oracle on a thin band
    cw, fw = coarse.packed(headline_dtype), fine.packed(headline_dtype)
    events, cevents = [], []
    t = Timed(H, W, hier_rows_fn(ops, cw, fw, H, W, K, device, events, coarse_events=cevents), events, device)
    steps = 4
    elapsed, _, _ = t.run(poses, steps, 1, sync)
    cevents = cevents[-len(events):]                 # the warm-up frame's pair was dropped from `events` by Timed.run
    rl = roofline_block(headline_dtype, t.kernel_ms(), H * W, 192, cfg["D"], cfg["W"], cfg["skip"])
    with ops.debug_switch(hier_chain=1):             # same box, same frames: raw [R,N,4] in HBM + the stand-alone compositing kernel
        tc = Timed(H, W, hier_rows_fn(ops, cw, fw, H, W, K, device, [], coarse_events=[]), [], device)
        elapsed_chain, rgb_hc, _ = tc.run(poses, steps, 1, sync)
    _, rgb_hf, _ = Timed(H, W, hier_rows_fn(ops, cw, fw, H, W, K, device, [], coarse_events=[]), [], device).run(poses, steps, 1, sync)
    coarse_ms = float(np.mean([b.elapsed_ms(e) for b, e in cevents]))
    rlc = roofline_block(headline_dtype, coarse_ms, H * W, 64, cfg["D"], cfg["W"], cfg["skip"])
    b0 = H // 2
    batch, _, _, _ = O.ray_batch_from_camera(H, W, K, poses[pose_k], 2.0, 6.0)
    with torch.no_grad():
        exp = O.hierarchical_render(batch[b0 * W : (b0 + 4) * W], params["coarse"], params["fine"], 64, 128, True, True)[3]
    got = ops.render_rays_hierarchical(cw, fw, camera=(H, W, K, poses[pose_k], b0, b0 + 4), n_coarse=64, n_importance=128,
                                       lindisp=True, white_bkgd=True, device=device)["rgb"].cpu()
    err = (got - exp).abs().max(-1).values
    out.append({"config": f"configs[2] ({H}x{W}, vanilla hierarchical 64 + 128, coarse + fine MLP), {headline_dtype}",
                "dtype": headline_dtype, "steps": steps, "rays_per_s": H * W * steps / elapsed,
                "ms_per_step": 1e3 * elapsed / steps, "kernel_ms": rl["kernel_ms"], "kernel": "fine pass (192 samples/ray)",
                "frac": rl["frac"], "executed_mfma_frac": rl["executed_mfma_frac"],
                "coarse_pass": {"kernel": "coarse pass (64 samples/ray)", "kernel_ms": coarse_ms, "frac": rlc["frac"],
                                "executed_mfma_frac": rlc["executed_mfma_frac"],
                                "effective_frac_on_reference_flops": rlc["effective_frac_on_reference_flops"]},
                "ms_outside_the_two_mlp_kernels": 1e3 * elapsed / steps - rl["kernel_ms"] - coarse_ms,
                "compositing": "in the MLP kernels' epilogues (no raw [R,N,4] array)",
                "ms_per_step_with_raw_arrays_and_compositing_launches": 1e3 * elapsed_chain / steps,
                "pixels_bit_identical_to_that_chain": bool(torch.equal(rgb_hc, rgb_hf)),
                "effective_frac_on_reference_flops": rl["effective_frac_on_reference_flops"],
                "max_abs_err_vs_oracle": float(err.max()), "median_abs_err_vs_oracle": float(err.median()),
                "psnr_vs_oracle_db": psnr(got, exp), "oracle_rays": int(exp.shape[0])})
    # -- configs[4] shape: 1600 x 1600, DepthNet + 192 samples, fp16, the whole frame on this GPU
    H5 = W5 = 2 * H
    _, K5 = synthetic.blender_intrinsics(H5, W5)
    nw, dw = fine.packed("f16"), dn.packed("f16")
    events = []
    t = Timed(H5, W5, hip_row_renderer(dw, nw, H5, W5, K5, 192, "uniform", 0.1, device=device, events=events), events, device)
    steps = 3
    elapsed, _, _ = t.run(poses, steps, 1, sync)
    rl = roofline_block("f16", t.kernel_ms(), H5 * W5, 192, cfg["D"], cfg["W"], cfg["skip"])
    tc = Timed(H5, W5, hip_row_renderer(dw, nw, H5, W5, K5, 192, "uniform", 0.1, device=device, events=[], one_kernel=False), [], device)
    elapsed_chain5, rgb5c, _ = tc.run(poses, steps, 1, sync)
    _, rgb5f, _ = Timed(H5, W5, hip_row_renderer(dw, nw, H5, W5, K5, 192, "uniform", 0.1, device=device, events=[]), [], device).run(poses, steps, 1, sync)
    b0 = H5 // 2
    with torch.no_grad():
        batch, _, _, _ = O.ray_batch_from_camera(H5, W5, K5, poses[pose_k], 2.0, 6.0)
        exp = O.render_rays_test(batch[b0 * W5 : (b0 + 2) * W5], params["coarse"], params["fine"], params["depth"], 192,
                                 "uniform", 0.1)["depth_net_rgb_map"]
    got = ops.render_rays_depthnet(dw, nw, camera=(H5, W5, K5, poses[pose_k], b0, b0 + 2), n_samples=192, mode="uniform",
                                   std=0.1, device=device)["rgb"].cpu()
    err = (got - exp).abs().max(-1).values
    out.append({"config": f"configs[4] shape ({H5}x{W5}, DepthNet + 192 samples), f16, one GPU", "dtype": "f16",
                "steps": steps, "rays_per_s": H5 * W5 * steps / elapsed, "ms_per_step": 1e3 * elapsed / steps,
                "renderer": "one kernel (a ray = three 64-sample chunks on different waves)",
                "ms_per_step_five_launch_chain": 1e3 * elapsed_chain5 / steps,
                "pixels_bit_identical_to_the_chain": bool(torch.equal(rgb5c, rgb5f)),
                "kernel_ms": rl["kernel_ms"], "frac": rl["frac"], "executed_mfma_frac": rl["executed_mfma_frac"],
                "effective_frac_on_reference_flops": rl["effective_frac_on_reference_flops"],
                "max_abs_err_vs_oracle": float(err.max()), "median_abs_err_vs_oracle": float(err.median()),
                "psnr_vs_oracle_db": psnr(got, exp), "oracle_rays": int(exp.shape[0])})
    return out


def main():
    args = parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:   # checked before anything touches the GPU
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to print a mislabelled number")
    n_dev = torch.cuda.device_count()
    dev_index = local_rank if (world > 1 and args.backend == "nccl") else min(local_rank, max(n_dev - 1, 0))
    if world > 1:
        torch.cuda.set_device(dev_index)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)
    device = torch.device("cuda", dev_index)
    torch.cuda.set_device(device)
    rccl_world = dist.get_world_size() if world > 1 else 1
    if rccl_world != args.gpus:
        raise SystemExit(f"bench.py: process group has {rccl_world} ranks, --gpus {args.gpus}")

    from nerf_sampling_amd import ops, synthetic
    from nerf_sampling_amd.parallel import hip_row_renderer

    H = W = args.size
    _, K = synthetic.blender_intrinsics(H, W)
    poses = synthetic.render_poses(40)[:, :3, :4]
    coarse, fine, dn, params = build_modules(args.scene, device)
    # the DepthNet's operand type under compute dtype X (ops.depthnet_dtype_for: f16 under bf16, same MFMA rate)
    ops.set_compute_dtype(args.dtype)
    depth_dtype = ops.depthnet_dtype_for(args.dtype)
    nerf_w, depth_w = fine.packed(args.dtype), dn.packed()
    depth_dtype = depth_w.dtype
    ops.set_compute_dtype("f32")
    events = []
    if args.mode == "depthnet":
        rows_fn = hip_row_renderer(depth_w, nerf_w, H, W, K, args.samples, "uniform", 0.1, device=device, events=events,
                                   one_kernel=False if args.chain else None)
        samples_in_timed_kernel = args.samples
    else:
        rows_fn = hier_rows_fn(ops, coarse.packed(args.dtype), nerf_w, H, W, K, device, events)
        samples_in_timed_kernel = 192  # the fine pass (64 + 128 samples) is the event-timed launch

    def sync():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    timed = Timed(H, W, rows_fn, events, device)
    elapsed, rgb, disp = timed.run(poses, args.steps, args.warmup, sync)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert os.environ.get("NS_BENCH_NOCHECK") or torch.isfinite(rgb).all()

    # dominant kernel (NeRF MLP), timed with HIP events on its own stream inside the timed region
    roofline = roofline_block(args.dtype, timed.kernel_ms(), timed.renderer.rays_per_rank, samples_in_timed_kernel,
                              fine.D, fine.W, 4)
    # HBM bytes per launch of that kernel: from the separate rocprofv3 --pmc passes under profiles/ (PMC counters cannot
    # be read from inside the process); only quoted for the exact workload and build they were collected on
    tname = "r04f_traffic_nerf_mlp.json"
    tpath = os.path.join(ROOT, "profiles", tname)
    if (world == 1 and args.mode == "depthnet" and args.dtype == "bf16" and args.size == 800 and args.samples == 64
            and not args.chain and os.path.exists(tpath)):
        rec = json.load(open(tpath))
        # the record names the kernel sources it was measured on: a build from other sources gets traffic = null, not a
        # stale number (tools/profile_round.sh rewrites the record)
        if rec.get("kernel_sources_sha256") == kernel_sources_sha256():
            roofline["traffic"] = rec["hbm_bytes_per_launch"]
            roofline["hbm_bytes_per_frame_all_kernels"] = rec.get("hbm_bytes_per_frame_all_kernels")
            roofline["kernels_per_frame"] = rec.get("kernels_per_frame")
            roofline["traffic_source"] = (f"profiles/{tname} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE in separate "
                                          "passes on this build's kernel sources; not live)")
        else:
            roofline["traffic"] = None
            roofline["traffic_source"] = f"profiles/{tname} was measured on other kernel sources: not quoted"

    one_kernel = (args.mode == "depthnet" and not args.chain
                  and bool(ops._lib.load().ns_render_fused_supported(nerf_w.handle, 1, args.samples)))
    renderer_note = ("ns_render_rays_fused: ray generation, DepthNet, ONE kernel for placement + MLP + compositing (3 launches "
                     "per frame; z / raw never in HBM)" if one_kernel else
                     "ns_render_rays_depthnet: ray generation, DepthNet, placement, MLP, compositing (5 launches per frame)"
                     if args.mode == "depthnet" else "ns_render_rays_hierarchical")
    if rank == 0:
        rays = H * W * args.steps
        scene_note = ("networks fitted to the analytic ground-truth scene (tools/fit_scene.py)" if args.scene == "shapes_fit"
                      else "seeded synthetic weights")
        out = {
            "metric": "rays/sec at 800x800, 64 samples/ray; PSNR vs reference",
            "value": rays / elapsed, "unit": "rays/s", "n_gpus": world, "rccl_world": rccl_world,
            "backend": args.backend if world > 1 else None, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": (f"Lego-shaped {H}x{W} frame, DepthNet (10x256, {depth_dtype} operands) + {args.samples} uniform "
                                    f"samples/ray (std 0.1) through the NeRF 8x256 fine MLP ({args.dtype} operands)"
                                    if args.mode == "depthnet" else
                                    f"Lego-shaped {H}x{W} frame, vanilla hierarchical 64 coarse + 128 importance "
                                    f"samples/ray (coarse + fine NeRF 8x256)")
                                   + f", {scene_note} ({args.scene}), spiral render poses of load_blender.py",
                       "rays_per_step": H * W, "samples_per_ray": args.samples, "depthnet_operands": depth_dtype,
                       "renderer": renderer_note,
                       "parallelism": f"rows sharded over {world} GPU(s), one all-gather per frame"},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline and args.mode == "depthnet":
            mid = H // 2
            rows = (max(0, mid - args.cpu_rows // 2), min(H, mid - args.cpu_rows // 2 + args.cpu_rows))
            pose_k = args.warmup % 40
            out["cpu_baseline"], ref = cpu_baseline(params, H, W, K, poses[pose_k], args.samples, rows)
            # "PSNR vs reference" half of the metric, outside the timed region, from the same oracle pass
            if not os.environ.get("NS_BENCH_NOCHECK"):
                acc, rgb_band = accuracy_vs_oracle(ops, depth_w, nerf_w, ref, H, W, K, poses[pose_k], args.samples,
                                                   args.dtype, device, pose_k, args.scene)
                out.update(acc)
                if args.scene == "shapes_fit":
                    out["scene_psnr"] = scene_psnr(rgb_band, ref, H, W, K, poses[pose_k], args.dtype)
                if args.other_configs:
                    out["other_configs"] = other_configs(ops, synthetic, (coarse, fine, dn), params, ref, H, W, K, poses,
                                                         pose_k, device, sync, args.dtype, args.samples)
            # the mirrored-API path with its host copies (SURVEY 8d "reported separately"); never `value`
            if args.api_path:
                a_ = api_path_rate(coarse, fine, dn, args.dtype, H, W, K, poses, args.samples, device)
                b_ = api_path_rate(coarse, fine, dn, args.dtype, H, W, K, poses, args.samples, device, blocking=True)
                ceil = d2h_ceiling(device)
                a_["d2h_gb_per_s"] = a_["host_bytes_per_frame"] / (a_["ms_per_frame"] * 1e-3) / 1e9
                a_["d2h_frac"] = a_["d2h_gb_per_s"] / ceil
                a_["frac_of_fused_rate"] = a_["rays_per_s"] / out["value"]
                out["api_path"] = {"async_pinned": a_, "blocking_as_reference": b_, "d2h_ceiling_gb_per_s": ceil,
                                   "d2h_ceiling_sample": "pinned hipMemcpyAsync device-to-host, 4 x 256 MiB, this box"}
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
