"""Per-frame wall times of the mirrored-API path (nerf_utils.render_test, 800x800, N = 64, bf16) with the asynchronous
pinned host sink: shows warm-up effects of the pinned allocator and the steady state."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nerf_sampling_amd import nerf_utils, ops, synthetic
from nerf_sampling_amd.run_nerf_helpers import get_embedder
from nerf_sampling_amd.trainers import DepthNetTrainer

dev = torch.device("cuda", 0)
_coarse, fine, dn, params = bench.build_modules("shapes_fit", dev)
H = W = 800
_, K = synthetic.blender_intrinsics(H, W)
poses = synthetic.render_poses(40)[:, :3, :4]
ops.set_compute_dtype("bf16")
tr = DepthNetTrainer(dataset_type="blender", basedir="/tmp", expname="x", no_batching=True, datadir="", half_res=False,
                     white_bkgd=True, N_importance=128, N_samples=64, use_viewdirs=True, input_dims_embed=3, device="cuda",
                     n_depth_samples=64, sampling_mode="uniform", distance=0.1)
e1, _ = get_embedder(10, 0, 3); e2, _ = get_embedder(4, 0, 3)
q = lambda i, v, f: tr.run_network(i, v, f, embed_fn=e1, embeddirs_fn=e2, netchunk=tr.netchunk)
kw = dict(network_query_fn=q, perturb=0.0, N_importance=128, network_fine=fine, N_samples=64, network_fn=fine, use_viewdirs=True,
          white_bkgd=True, raw_noise_std=0.0, trainer=tr, lindisp=True, depth_network=dn, model_mode="test", near=2.0, far=6.0,
          ndc=False)
with torch.no_grad():
    for label, blocking, tagged in (("async, one-call branch (create_nerf's query fn)", False, True), ("async, operator chain", False, False),
                                    ("blocking, operator chain (as the reference)", True, False)):
        kw["network_query_fn"] = nerf_utils.standard_query_fn(lambda i, v, f: q(i, v, f)) if tagged else q
        ts = []
        for i in range(10):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            rgb, disp, ex = nerf_utils.render_test(H, W, K, chunk=tr.chunk, c2w=poses[i], _blocking_host_copies=blocking, **kw)
            torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
        print(f"{label:48s}", " ".join(f"{t:6.1f}" for t in ts))
    # the loop experiments/render.py runs: render_path keeps frame i's host copies in flight under frame i+1's kernels
    kw["network_query_fn"] = nerf_utils.standard_query_fn(lambda i, v, f: q(i, v, f))
    n = int(os.environ.get("NS_API_FRAMES", "8"))
    # host side of one frame: time until render_test has SUBMITTED everything (no synchronisation)
    torch.cuda.synchronize()
    sub = []
    for i in range(6):
        t0 = time.perf_counter()
        rgb, disp, ex = nerf_utils.render_test(H, W, K, chunk=tr.chunk, c2w=poses[i], _defer_host_sync=True, **kw)
        sub.append(1e3 * (time.perf_counter() - t0))
    nerf_utils.drain_host_copies(); torch.cuda.synchronize()
    print("host time to submit a frame (ms; includes waiting for the previous frame's copies):", " ".join(f"{t:5.1f}" for t in sub))
    for rep in range(3):      # the first pass still grows torch's pinned-memory cache (one more frame's buffers in flight)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        nerf_utils.render_path(poses[3 : 3 + n], [H, W, float(K[0][0])], K, tr.chunk, kw, step=0)
        torch.cuda.synchronize()
        print(f"render_path pass {rep}, {n} frames: {1e3 * (time.perf_counter() - t0) / n:.1f} ms/frame")
