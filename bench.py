#!/usr/bin/env python3
"""Headline benchmark: rays/sec at 800x800, DepthNet + 64 samples/ray (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N rank processes, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one full 800x800 frame of the reference's spiral render path (pose k of 40): ray generation,
DepthNet, sample placement, NeRF MLP and compositing, all on the device, plus -- for N > 1 -- the single
all-gather that assembles the frame.  Frames are row-sharded over the N ranks (total work fixed: strong
scaling).  Weights are the seeded synthetic "lego_synth" scene (no dataset / checkpoint ships with the
reference); inputs are 16 camera scalars, so nothing is staged from the host inside the timed region.
Rank 0 prints ONE JSON line.
"""

import argparse
import json
import os
import socket
import subprocess
import sys
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
    ap.add_argument("--scene", default="lego_synth")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=50, help="rows of the frame timed on the CPU (x 800 rays)")
    ap.add_argument("--mode", default="depthnet", choices=["depthnet", "full_nerf"],
                    help="depthnet = BASELINE configs[1] (headline); full_nerf = configs[2], vanilla 64+128 coarse+fine")
    ap.add_argument("--api-path", dest="api_path", action="store_true", default=True,
                    help="also time nerf_utils.render_test (the mirrored reference API with per-chunk host copies)")
    ap.add_argument("--no-api-path", dest="api_path", action="store_false")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                    "the multi-rank path on a one-GPU box, all ranks sharing cuda:0)")
    return ap.parse_args(argv)


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without an outer launcher: start N fresh rank processes (one per GPU, RCCL rendezvous
    on 127.0.0.1) and relay rank 0's JSON line.  The parent never touches the GPU (it has not even imported torch), so
    nothing is exec'ed or forked from a process with an initialised HIP runtime.  Any failing rank fails the run."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), NS_BENCH_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    # poll: a rank that dies early must not leave the others waiting in the rendezvous for ever
    failed = False
    while any(p_.poll() is None for p_ in procs):
        if any(p_.poll() not in (None, 0) for p_ in procs):
            failed = True
            time.sleep(2.0)   # let the others report, then end them
            for p_ in procs:
                if p_.poll() is None:
                    p_.kill()
            break
        time.sleep(0.05)
    out0 = procs[0].stdout.read() if procs[0].stdout else ""
    rcs = [p_.wait() for p_ in procs]
    for line in (out0 or "").splitlines():          # the JSON line to stdout, library chatter (gloo prints there) to stderr
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    if failed or any(rcs):
        sys.stderr.write(f"bench.py: rank exit codes {rcs}\n")
        return 1
    return 0


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ:
    _a = parse_args()
    if _a.gpus > 1:
        sys.exit(spawn_ranks(_a.gpus))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# algorithmic work, SURVEY.md section 8(a): MACs counted on the reference's arithmetic, no padding credit
NERF_FLOP_PER_SAMPLE = 2 * 593_408
DEPTHNET_FLOP_PER_RAY = 2 * 3_330_304
PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3,  # dense MFMA, MI355X_MICROARCH.md
               "f16x3": 2500.0}  # split fp16 operands run on the fp16 MFMA pipe (3 MFMAs per product term: see executed_*)


MEASURED_MFMA_CEILING = {"bf16": 2140.0, "f16": None, "f32": 155.0}   # TFLOP/s (bf16: bare 16x16x32 loop), DESIGN.md section 6


def build_modules(scene_name, device):
    from nerf_sampling_amd import synthetic
    from nerf_sampling_amd.depth_net import DepthNet
    from nerf_sampling_amd.run_nerf_helpers import NeRF

    cfg, params = synthetic.SCENES[scene_name], synthetic.make_scene(scene_name)
    fine = NeRF(D=cfg["fine"]["D"], W=cfg["fine"]["W"], input_ch=63, input_ch_views=27, output_ch=5, skips=[4],
                use_viewdirs=True)
    fine.load_state_dict(params["fine"])
    if os.environ.get("NS_BENCH_ZERO_NERF"):   # diagnostic only: all-zero MLP operands (is the kernel power-limited?)
        with torch.no_grad():
            for p_ in fine.parameters():
                p_.zero_()
    n, w = cfg["depth"]["n_layers"], cfg["depth"]["width"]
    dn = DepthNet(hidden_sizes=[w] * n, cat_hidden_sizes=[w] * n, sphere_radius=2.0)
    dn.load_state_dict(params["depth"])
    return fine.to(device), dn.to(device), params


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


def cpu_baseline(params, H, W, K, c2w, n_samples, rows, budget_s=24.0):
    """The oracle (CPU port of the reference path) timed on the host cores over a band of rows of the same frame with
    the reference's own chunk structure: chunk = 32768 rays per render_rays_test call (Trainer.py:31), the NeRF MLP in
    netchunk = 65536-row slices (Trainer.py:36).  Each chunk runs exactly what nerf_oracle.render_rays_test runs
    (depthnet_forward -> place_samples -> run_network -> raw2outputs); the per-chunk pieces are kept so the accuracy
    leg below can build its conditioning mask from the same pass instead of a second one."""
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
            parts.append((raw, z, d, rgb))
            done = min(batch.shape[0], done + chunk)
        dt = time.perf_counter() - t0
    ref = {k: torch.cat([p_[i] for p_ in parts], 0) for i, k in enumerate(("raw", "z", "d", "rgb"))}
    ref["rows"] = (rows[0], rows[0] + done // W)     # whole rows the oracle finished
    return {"value": done / dt, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": f"{done} rays from rows {rows[0]}-{rows[1]} of the same {H}x{W} frame, fp32 torch-CPU oracle in the "
                      f"reference's chunks (32768 rays / 65536 MLP rows), {dt:.1f} s"}, ref


# rms error of the 16-bit HIP sigma against the oracle's at the same points, as a fraction of max |sigma| of the band:
# measured and pinned (+-30 %) by tests/test_gpu_render.py::test_frame16_vs_oracle
SIGMA_NOISE_FRAC = {"bf16": 1.55e-2, "f16": 2.4e-3, "f32": 1.0e-5, "f16x3": 1.0e-5}


def accuracy_vs_oracle(ops, depth_w, nerf_w, ref, H, W, K, c2w, n_samples, dtype, device):
    """Second half of the cpu_baseline leg (the oracle as CHECKER, outside the timed region, rank 0 at N = 1 only):
    PSNR(build || oracle) on the band the CPU leg rendered.  The reference composites the last sample with
    dist = 1e10 (sampling_trainer.py:176-180): alpha_last = step(sigma_last), so a ray's colour is discontinuous in
    sigma_last; rays whose ORACLE colour moves by > 1e-2 under a sigma shift of 3x the dtype's measured sigma noise are
    ill-conditioned and reported as a fraction; the PSNR is quoted on all rays and on the well-conditioned ones."""
    from oracle import nerf_oracle as O

    r0, r1 = ref["rows"]
    n = (r1 - r0) * W
    if n == 0:
        return {}
    raw, z, d, rgb_ref = (ref[k][:n] for k in ("raw", "z", "d", "rgb"))
    rgb = ops.render_rays_depthnet(depth_w, nerf_w, camera=(H, W, K, c2w, r0, r1), n_samples=n_samples, mode="uniform",
                                   std=0.1, device=device)["rgb"].cpu()
    eps = 3.0 * SIGMA_NOISE_FRAC[dtype] * float(raw[..., 3].abs().max())
    ill = torch.zeros(n, dtype=torch.bool)
    with torch.no_grad():
        for sgn in (-1.0, 1.0):
            pert = raw.clone()
            pert[..., 3] += sgn * eps
            ill |= (O.raw2outputs(pert, z, d, 0.0, True)[0] - rgb_ref).abs().max(-1).values > 1e-2

    def psnr(a, b):
        mse = float(((a - b) ** 2).mean())
        return None if mse == 0 else -10.0 * float(np.log10(mse))

    err = (rgb - rgb_ref).abs().max(-1).values
    return {"psnr_vs_oracle_db": psnr(rgb, rgb_ref), "psnr_vs_oracle_wellconditioned_db": psnr(rgb[~ill], rgb_ref[~ill]),
            "ill_conditioned_frac": float(ill.float().mean()),
            "max_abs_err_wellconditioned": float(err[~ill].max()) if (~ill).any() else None,
            "rays_off_by_1e-2_outside_mask": float(((err > 1e-2) & ~ill).float().mean()),
            "accuracy_sample": f"rows {r0}-{r1} of pose {{pose}} against the fp32 CPU oracle on identical rays and weights; "
                               f"mask: oracle colour moves > 1e-2 under a sigma shift of +-{eps:.3g} (3x measured {dtype} noise)"}


def api_path_rate(fine, dn, params, scene, dtype, H, W, K, poses, n_samples, device, frames=6, blocking=False):
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
    kw = dict(network_query_fn=query, perturb=0.0, N_importance=128, network_fine=fine, N_samples=64, network_fn=fine,
              use_viewdirs=True, white_bkgd=True, raw_noise_std=0.0, trainer=tr, lindisp=True, depth_network=dn,
              model_mode="test", near=2.0, far=6.0, ndc=False, _blocking_host_copies=blocking)
    with torch.no_grad():
        for i in range(2 if blocking else 4):    # the pinned-buffer cache of the async sink takes three frames to fill
            nerf_utils.render_test(H, W, K, chunk=tr.chunk, c2w=poses[i], **kw)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for i in range(frames):
            rgb, disp, extras = nerf_utils.render_test(H, W, K, chunk=tr.chunk, c2w=poses[(4 + i) % 40], **kw)
        torch.cuda.synchronize(device)
        dt = (time.perf_counter() - t0) / frames
    host_bytes = sum(v.numel() * v.element_size() for v in extras.values() if isinstance(v, torch.Tensor) and not v.is_cuda)
    ops.set_compute_dtype("f32")
    return {"rays_per_s": H * W / dt, "ms_per_frame": 1e3 * dt, "host_bytes_per_frame": host_bytes, "frames": frames}


def nerf_executed_flop_per_sample(D, W, skip, dtype):
    """FLOPs the kernel really issues per sample (MFMA count x FLOP per MFMA, padding included): the folded program of
    ns_nerf_mlp_ob16.hip (16-bit; 16x16x32 MFMAs, 4 tiles per chunk) or ns_nerf_mlp.hip (fp32; k-major 32-row blocks)."""
    if dtype == "f32":
        nb = W // 32
        blocks = nb * 2 + sum(nb * (nb + (2 if l - 1 == skip else 0)) for l in range(1, D))
        blocks += nb + (nb // 2) * (nb + 1) + nb // 2            # alpha, views o feature, rgb
        return blocks * 2 * 32 * 32                                # one 32-row x 32-feature block pair per sample column
    nsb, nkb = W // 16, W // 32
    chunks = nsb * 2 + sum(nsb * (nkb + (2 if l - 1 == skip else 0)) for l in range(1, D))
    chunks += (nsb // 2 + 1) * (nkb + 1) + nkb // 2
    return chunks * 2 * 16 * 32 * (3 if dtype == "f16x3" else 1)   # a chunk = 16 rows x 32 features, per sample column


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
    from nerf_sampling_amd.parallel import FrameRenderer, hip_row_renderer

    H = W = args.size
    _, K = synthetic.blender_intrinsics(H, W)
    poses = synthetic.render_poses(40)[:, :3, :4]
    fine, dn, params = build_modules(args.scene, device)
    nerf_w, depth_w = fine.packed(args.dtype), dn.packed(args.dtype)
    events = []
    if args.mode == "depthnet":
        rows_fn = hip_row_renderer(depth_w, nerf_w, H, W, K, args.samples, "uniform", 0.1, device=device, events=events)
        samples_in_timed_kernel = args.samples
    else:
        from nerf_sampling_amd.run_nerf_helpers import NeRF

        cfg = synthetic.SCENES[args.scene]["coarse"]
        coarse = NeRF(D=cfg["D"], W=cfg["W"], input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
        coarse.load_state_dict(params["coarse"])
        coarse_w = coarse.to(device).packed(args.dtype)
        ws = ops.RenderWorkspace()

        def rows_fn(c2w, row0, row1):
            ev = (ops.Event(), ops.Event())
            events.append(ev)
            out = ops.render_rays_hierarchical(coarse_w, nerf_w, camera=(H, W, K, c2w, row0, row1), n_coarse=64,
                                               n_importance=128, lindisp=True, white_bkgd=True, workspace=ws,
                                               device=device, mlp_events=ev)
            return out["rgb"], out["disp"]

        samples_in_timed_kernel = 192  # the fine pass (64 + 128 samples) is the event-timed launch
    renderer = FrameRenderer(H, W, rows_fn, device)

    def sync():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    for i in range(args.warmup):
        renderer.render(poses[i % 40])
    events.clear()
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        rgb, disp = renderer.render(poses[(args.warmup + i) % 40])
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert os.environ.get("NS_BENCH_NOCHECK") or torch.isfinite(rgb).all()

    # dominant kernel (NeRF MLP), timed with HIP events on its own stream inside the timed region
    mlp_ms = float(np.mean([b.elapsed_ms(e) for b, e in events])) if events else float("nan")
    rays_per_launch = renderer.rays_per_rank
    mlp_flop = rays_per_launch * samples_in_timed_kernel * NERF_FLOP_PER_SAMPLE
    achieved = mlp_flop / (mlp_ms * 1e-3) / 1e12 if mlp_ms > 0 else float("nan")
    peak = PEAK_TFLOPS[args.dtype]
    # HBM bytes per launch of that kernel: from the separate rocprofv3 --pmc passes under profiles/ (not live);
    # only quoted for the exact workload they were collected on
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r02d_traffic_nerf_mlp.json")
    if (world == 1 and args.mode == "depthnet" and args.dtype == "bf16" and args.size == 800 and args.samples == 64
            and os.path.exists(tpath)):
        traffic = json.load(open(tpath))["hbm_bytes_per_launch"]

    cfg_fine = synthetic.SCENES[args.scene]["fine"]
    executed = rays_per_launch * samples_in_timed_kernel * nerf_executed_flop_per_sample(cfg_fine["D"], cfg_fine["W"], 4,
                                                                                         args.dtype)
    executed_rate = executed / (mlp_ms * 1e-3) / 1e12 if mlp_ms > 0 else float("nan")

    if rank == 0:
        rays = H * W * args.steps
        out = {
            "metric": "rays/sec at 800x800, 64 samples/ray; PSNR vs reference",
            "value": rays / elapsed, "unit": "rays/s", "n_gpus": world, "rccl_world": rccl_world,
            "backend": args.backend if world > 1 else None, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": (f"Lego-shaped {H}x{W} frame, DepthNet (10x256) + {args.samples} uniform samples/ray "
                                    f"(std 0.1) through the NeRF 8x256 fine MLP" if args.mode == "depthnet" else
                                    f"Lego-shaped {H}x{W} frame, vanilla hierarchical 64 coarse + 128 importance "
                                    f"samples/ray (coarse + fine NeRF 8x256)")
                                   + f", seeded synthetic weights ({args.scene}), spiral render poses of load_blender.py",
                       "rays_per_step": H * W, "samples_per_ray": args.samples,
                       "parallelism": f"rows sharded over {world} GPU(s), one all-gather per frame"},
            "roofline": {"bound": "mfma", "kernel": {"f32": "nerf_mlp_kernel", "f16x3": "nerf_mlp_x3_kernel"}.get(args.dtype, "nerf_mlp_ob16_kernel"), "achieved": achieved, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                         "kernel_ms": mlp_ms, "algorithmic_flop_per_launch": mlp_flop,
                         # `achieved` counts the REFERENCE's arithmetic (593 408 MAC / sample); the kernel
                         # issues fewer MFMAs than that since feature_linear is folded into views_linears at pack time:
                         # executed_* = MFMAs issued x FLOP per MFMA (padding included), i.e. matrix-pipe utilisation
                         "executed_flop_per_launch": executed, "executed_tflops": executed_rate,
                         "executed_mfma_frac": executed_rate / peak,
                         # context, not the contract's peak: what a bare v_mfma_f32_16x16x32 loop whose A and B operands
                         # change on every MFMA sustains on this part under its power cap (tools/mfma_peak.hip)
                         "measured_mfma_ceiling": MEASURED_MFMA_CEILING.get(args.dtype)},
        }
        if world == 1 and not args.no_cpu_baseline and args.mode == "depthnet":
            mid = H // 2
            rows = (max(0, mid - args.cpu_rows // 2), min(H, mid - args.cpu_rows // 2 + args.cpu_rows))
            pose_k = args.warmup % 40
            out["cpu_baseline"], ref = cpu_baseline(params, H, W, K, poses[pose_k], args.samples, rows)
            # "PSNR vs reference" half of the metric, outside the timed region, from the same oracle pass
            if not os.environ.get("NS_BENCH_NOCHECK"):
                acc = accuracy_vs_oracle(ops, depth_w, nerf_w, ref, H, W, K, poses[pose_k], args.samples, args.dtype, device)
                if "accuracy_sample" in acc:
                    acc["accuracy_sample"] = acc["accuracy_sample"].format(pose=pose_k)
                out.update(acc)
            # the mirrored-API path with its host copies (SURVEY 8d "reported separately"); never `value`
            if args.api_path:
                out["api_path"] = {
                    "async_pinned": api_path_rate(fine, dn, params, args.scene, args.dtype, H, W, K, poses, args.samples, device),
                    "blocking_as_reference": api_path_rate(fine, dn, params, args.scene, args.dtype, H, W, K, poses,
                                                           args.samples, device, blocking=True)}
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
