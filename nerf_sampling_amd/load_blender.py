"""Blender-synthetic dataset loader (mirror of nerf_sampling/nerf_pytorch/load_blender.py:46-103).

Host I/O only.  PNGs are read with PIL (the reference uses imageio + cv2,
which are not installed here); half_res uses the exact 2x2 box average that cv2.INTER_AREA performs for
a factor-2 downscale.
"""

from __future__ import annotations

import json
import os

import numpy as np
import torch

from .synthetic import pose_spherical


def _imread(path):
    from PIL import Image

    with Image.open(path) as im:
        return np.array(im.convert("RGBA"))


def load_blender_data(basedir, half_res=False, testskip=1):
    """-> imgs [N,H,W,4] float32, poses [N,4,4], render_poses [40,4,4], [H, W, focal], i_split."""
    splits = ["train", "val", "test"]
    metas = {}
    for s in splits:
        with open(os.path.join(basedir, f"transforms_{s}.json"), "r") as fp:
            metas[s] = json.load(fp)
    all_imgs, all_poses, counts = [], [], [0]
    for s in splits:
        meta = metas[s]
        skip = 1 if (s == "train" or testskip == 0) else testskip
        imgs, poses = [], []
        for frame in meta["frames"][::skip]:
            imgs.append(_imread(os.path.join(basedir, frame["file_path"] + ".png")))
            poses.append(np.array(frame["transform_matrix"]))
        imgs = (np.array(imgs) / 255.0).astype(np.float32)
        poses = np.array(poses).astype(np.float32)
        counts.append(counts[-1] + imgs.shape[0])
        all_imgs.append(imgs)
        all_poses.append(poses)
    i_split = [np.arange(counts[i], counts[i + 1]) for i in range(3)]
    imgs = np.concatenate(all_imgs, 0)
    poses = np.concatenate(all_poses, 0)
    H, W = imgs[0].shape[:2]
    camera_angle_x = float(meta["camera_angle_x"])
    focal = 0.5 * W / np.tan(0.5 * camera_angle_x)
    render_poses = torch.stack([pose_spherical(a, -30.0, 4.0) for a in np.linspace(-180, 180, 40 + 1)[:-1]], 0)
    if half_res:
        if H % 2 or W % 2:
            raise ValueError("half_res needs even image sizes")
        H, W, focal = H // 2, W // 2, focal / 2.0
        imgs = imgs.reshape(imgs.shape[0], H, 2, W, 2, 4).mean(axis=(2, 4)).astype(np.float64)
    return imgs, poses, render_poses, [H, W, focal], i_split
