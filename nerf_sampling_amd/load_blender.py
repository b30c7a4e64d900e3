"""Blender-synthetic dataset reader.

The on-disk format (what nerf_sampling/nerf_pytorch/load_blender.py:46-103 consumes; host I/O only, SURVEY.md section 8f
row 1): a directory with ``transforms_train.json``, ``transforms_val.json``, ``transforms_test.json``; each holds
``camera_angle_x`` (horizontal field of view, radians) and ``frames``, a list of ``{"file_path": "./<split>/r_<i>",
"transform_matrix": 4x4 camera-to-world}``; the image of a frame is ``<file_path>.png``, 8-bit RGBA, all of one size.

What the trainer expects back (same tuple as the reference): every image as float32 RGBA in [0, 1] in ONE array ordered
train | val | test, the matching poses, the 40 spiral render poses, ``[H, W, focal]`` with ``focal = W / (2 tan(fov / 2))``,
and the three index ranges.  ``testskip`` thins val and test (never train; 0 means 1).  ``half_res`` halves the images
with the exact 2x2 box mean -- what an area-interpolated factor-2 resize computes -- and, as in the reference, returns them
as float64.

Written around the format: the three indexes are read first, the output arrays are allocated once at their final size, and
each PNG is decoded straight into its row (PIL; the reference's imageio / cv2 are not installed here).
"""

from __future__ import annotations

import json
import os
from typing import List, NamedTuple

import numpy as np
import torch

from .synthetic import pose_spherical

SPLITS = ("train", "val", "test")


class _Frame(NamedTuple):
    png: str
    c2w: list


class _SplitIndex(NamedTuple):
    frames: List[_Frame]
    fov_x: float


def _read_index(basedir: str, split: str, stride: int) -> _SplitIndex:
    """The frames of one split that will be loaded (every ``stride``-th) and the split's field of view."""
    with open(os.path.join(basedir, f"transforms_{split}.json")) as fp:
        doc = json.load(fp)
    picked = doc["frames"][:: max(1, stride)]
    return _SplitIndex([_Frame(os.path.join(basedir, f["file_path"] + ".png"), f["transform_matrix"]) for f in picked],
                       float(doc["camera_angle_x"]))


def _decode_rgba(path: str, out: np.ndarray) -> None:
    """PNG -> out [H, W, 4] float32 in [0, 1]."""
    from PIL import Image

    with Image.open(path) as im:
        px = np.asarray(im.convert("RGBA"))
    if px.shape != out.shape:
        raise ValueError(f"{path}: {px.shape[1]}x{px.shape[0]} image in a dataset of {out.shape[1]}x{out.shape[0]} images")
    np.divide(px, 255.0, out=out, casting="unsafe")


def _image_size(path: str):
    from PIL import Image

    with Image.open(path) as im:
        return im.height, im.width


def load_blender_data(basedir, half_res=False, testskip=1):
    """-> imgs [N,H,W,4], poses [N,4,4] float32, render_poses [40,4,4], [H, W, focal], i_split (train, val, test)."""
    index = {s: _read_index(basedir, s, 1 if s == "train" else testskip) for s in SPLITS}
    sizes = [len(index[s].frames) for s in SPLITS]
    total = sum(sizes)
    if total == 0:
        raise ValueError(f"{basedir}: no frames in any split")
    first = next(f for s in SPLITS for f in index[s].frames)
    H, W = _image_size(first.png)
    imgs = np.empty((total, H, W, 4), dtype=np.float32)
    poses = np.empty((total, 4, 4), dtype=np.float32)
    row = 0
    for s in SPLITS:
        for frame in index[s].frames:
            _decode_rgba(frame.png, imgs[row])
            poses[row] = np.asarray(frame.c2w, dtype=np.float32)
            row += 1
    bounds = np.cumsum([0] + sizes)
    i_split = [np.arange(bounds[k], bounds[k + 1]) for k in range(len(SPLITS))]
    # one camera for the whole scene: the reference takes the field of view of the split it read last
    focal = 0.5 * W / np.tan(0.5 * index[SPLITS[-1]].fov_x)
    render_poses = torch.stack([pose_spherical(float(theta), -30.0, 4.0) for theta in -180.0 + 9.0 * np.arange(40)], 0)
    if half_res:
        if H % 2 or W % 2:
            raise ValueError("half_res needs even image sizes")
        H, W, focal = H // 2, W // 2, focal / 2.0
        imgs = imgs.reshape(total, H, 2, W, 2, 4).mean(axis=(2, 4)).astype(np.float64)
    return imgs, poses, render_poses, [H, W, focal], i_split
