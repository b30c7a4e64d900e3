"""Per-rank frame time for an N-way row shard, measured on ONE GPU (no collective): what one rank of an N-GPU
run spends per frame, i.e. the compute side of the strong-scaling curve (tail effects of the persistent kernels)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nerf_sampling_amd import synthetic
from nerf_sampling_amd.parallel import hip_row_renderer, row_range

dev = torch.device("cuda", 0)
_c, fine, dn, _ = bench.build_modules("shapes_fit", dev)
H = W = 800
_, K = synthetic.blender_intrinsics(H, W)
poses = synthetic.render_poses(40)[:, :3, :4]
rows = hip_row_renderer(dn.packed("bf16"), fine.packed("bf16"), H, W, K, 64, "uniform", 0.1, device=dev)
full = None
for n in (1, 2, 4, 8):
    r0, r1, per = row_range(H, n - 1, n)
    shard = torch.zeros((per * W, 4), dtype=torch.float32, device=dev)   # as FrameRenderer: the kernels write the gather shard
    for i in range(3): rows(poses[i], r0, r1, shard)
    torch.cuda.synchronize(); t0 = time.perf_counter(); k = 20
    for i in range(k): rows(poses[i % 40], r0, r1, shard)
    torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0) / k
    full = full or ms
    print(f"N={n}: rows {r0}-{r1}: {ms:.3f} ms per frame-shard; compute-side speedup {full / ms:.2f}x of {n}")
