"""Row-sharded frame assembly over 2 ranks (gloo, CPU): the N>1 path of nerf_sampling_amd.parallel.

The per-rank renderer is stubbed with the CPU oracle on a tiny frame (tests may use the oracle); what is
under test is the sharding arithmetic and the single all-gather, which are device-agnostic.
"""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nerf_sampling_amd.parallel import FrameRenderer, row_range


def test_row_range_partitions_every_row_once():
    for H in (1, 5, 7, 64, 800, 801):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                r0, r1, per = row_range(H, r, world)
                assert 0 <= r0 <= r1 <= H and r1 - r0 <= per
                seen += list(range(r0, r1))
            assert seen == list(range(H)), (H, world)


def _oracle_rows(H, W):
    from oracle import nerf_oracle as O

    sc = O.make_scene("tiny_synth")
    _, K = O.blender_intrinsics(H, W)

    def render_rows(c2w, row0, row1, shard=None):
        batch, _, _, _ = O.ray_batch_from_camera(H, W, K, c2w, 2.0, 6.0)
        res = O.render_rays_test(batch[row0 * W : row1 * W], sc["coarse"], sc["fine"], sc["depth"], 4, "uniform", 0.1)
        return res["depth_net_rgb_map"], res["depth_net_disp_map"]

    return render_rows


def _worker(rank, world, port, H, W, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from oracle import nerf_oracle as O

        fr = FrameRenderer(H, W, _oracle_rows(H, W), "cpu")
        assert fr.world == world and fr.rank == rank
        c2w = O.pose_spherical(40.0, -30.0, 4.0)[:3, :4]
        rgb, disp = fr.render(c2w)
        rgb2, _ = fr.render(c2w)              # buffers are reused across frames
        assert torch.allclose(rgb, rgb2, rtol=0, atol=0, equal_nan=True)  # a corner ray misses the sphere: NaN, as the reference
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), rgb=rgb.numpy(), disp=disp.numpy())
        # pipelined mode: the all-gather of frame i stays in flight under frame i+1 (two shard / frame buffer pairs);
        # after finish() the last two frames are intact and equal the synchronous renders
        poses = [O.pose_spherical(a, -30.0, 4.0)[:3, :4] for a in (40.0, 75.0, 110.0)]
        sync = [tuple(t.clone() for t in fr.render(p_)) for p_ in poses]
        pipe = [fr.render(p_, wait=False) for p_ in poses]
        fr.finish()
        for k in (1, 2):
            assert torch.equal(pipe[k][0], sync[k][0]) or torch.allclose(pipe[k][0], sync[k][0], equal_nan=True)
            assert torch.allclose(pipe[k][1], sync[k][1], equal_nan=True)
        assert pipe[0][0].data_ptr() == pipe[2][0].data_ptr()      # frame 2 reused frame 0's buffers
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("H,W,world", [(8, 6, 2), (7, 5, 2), (3, 4, 4)])
def test_two_rank_frame_equals_single_process(tmp_path, H, W, world):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(world, port, H, W, str(tmp_path)), nprocs=world, join=True)
    from oracle import nerf_oracle as O

    c2w = O.pose_spherical(40.0, -30.0, 4.0)[:3, :4]
    rgb, disp = _oracle_rows(H, W)(c2w, 0, H)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        assert got["rgb"].shape == (H, W, 3) and got["disp"].shape == (H, W)
        np.testing.assert_allclose(got["rgb"].reshape(-1, 3), rgb.numpy(), rtol=0, atol=2e-5)  # CPU GEMM rounding depends on the batch split
        np.testing.assert_allclose(got["disp"].reshape(-1), disp.numpy(), rtol=2e-5, atol=1e-6)


def _stub_rows(H, W):
    """A per-rank renderer that costs nothing: (r, g, b, disp) of pixel p under pose c2w is a closed-form function of p and of
    the pose, written straight into the rank's shard like the HIP renderers do -- what is under test at world 8 is the row
    partition, the shard / frame buffer rotation and the all-gather at the real frame sizes, not the arithmetic."""
    def render_rows(c2w, row0, row1, shard=None):
        n = (row1 - row0) * W
        p = torch.arange(row0 * W, row1 * W, dtype=torch.float64)
        key = float(c2w[0, 3]) + 2.0 * float(c2w[1, 3])
        vals = torch.stack([torch.sin(p * 1e-3 + key), torch.cos(p * 7e-4 - key), (p % 251.0) / 251.0, 1.0 / (1.0 + p * 1e-6 + key * key)], -1).float()
        shard[:n] = vals
        return shard[:n, :3], shard[:n, 3]

    return render_rows


def _worker8(rank, world, port, H, W, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        from nerf_sampling_amd.synthetic import pose_spherical

        fr = FrameRenderer(H, W, _stub_rows(H, W), "cpu")
        assert (fr.world, fr.rank) == (world, rank) and fr.rays_per_rank == (H // world) * W
        poses = [pose_spherical(a, -30.0, 4.0)[:3, :4] for a in (0.0, 45.0, 90.0, 135.0, 180.0)]
        sync = [tuple(t.clone() for t in fr.render(p_)) for p_ in poses]
        pipe = [fr.render(p_, wait=False) for p_ in poses]       # the all-gather of frame i in flight under frame i + 1
        fr.finish()
        for k in (3, 4):                                         # the last two frames own the two buffer pairs
            assert torch.equal(pipe[k][0], sync[k][0]) and torch.equal(pipe[k][1], sync[k][1]), (rank, k)
        full = torch.empty((H * W, 4))                           # the same frames rendered whole, no process group
        for k, p_ in enumerate(poses):
            _stub_rows(H, W)(p_, 0, H, full)
            assert torch.equal(sync[k][0].reshape(-1, 3), full[:, :3]) and torch.equal(sync[k][1].reshape(-1), full[:, 3]), (rank, k)
        if rank == world - 1:
            open(os.path.join(out_dir, "ok"), "w").write(f"{H}x{W} world {world}")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("H", [800, 1600])
def test_world_8_frame_assembly_at_the_benchmark_sizes(tmp_path, H):
    """Eight ranks (gloo, CPU), the frames of BASELINE configs[3] / [4] (800 x 800, 1600 x 1600): every rank assembles the
    whole frame, pipelined mode (wait=False + finish) equals the synchronous one, five frames through the two buffer pairs.
    RCCL itself at world > 1 is exercised only by the driver's multi-GPU run (one GPU per box here)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker8, args=(8, port, H, H, str(tmp_path)), nprocs=8, join=True)
    assert open(os.path.join(str(tmp_path), "ok")).read() == f"{H}x{H} world 8"


def test_bench_refuses_a_world_size_mismatch():
    """bench.py --gpus N under a launcher that made a different world size exits non-zero instead of printing a
    mislabelled number (checked before anything touches a GPU, so this runs on the CPU box)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()
