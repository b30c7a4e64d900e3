"""Raw output of the production NeRF (8 x 256, seeded weights) on seeded points, written to a file: run once per library
(NS_LIB_PATH) and `cmp` the files -- variants of the hand-scheduled kernel must be bit-identical to the in-tree build.
    python tools/raw_dump.py OUT.npy [dtype]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerf_sampling_amd import ops, synthetic  # noqa: E402
from nerf_sampling_amd.run_nerf_helpers import NeRF  # noqa: E402

dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
params = synthetic.make_nerf_params(seed=77, D=8, W=256, skips=(4,), hidden_gain=6 ** 0.5, sigma_gain=30.0, spectral_decay=True)
net = NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
net.load_state_dict(params)
net = net.cuda()
gen = torch.Generator().manual_seed(3)
outs = []
for R, N in ((5, 7), (4100, 64), (1500, 192)):
    pts = ((torch.rand(R, N, 3, generator=gen) * 2 - 1) * 2.5).cuda()
    view = torch.nn.functional.normalize(torch.randn(R, 3, generator=gen), dim=-1).cuda()
    outs.append(ops.nerf_forward(net.packed(dtype), pts, view).cpu().numpy().reshape(-1))
raw = np.concatenate(outs)
assert np.isfinite(raw).all()
np.save(sys.argv[1], raw)
print("raw checksum", float(np.abs(raw).sum()), raw.shape)
