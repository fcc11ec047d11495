#!/usr/bin/env python
"""Host-side (Python) cost of one frame: cProfile over a few render_fast calls, GPU kept busy asynchronously.
    python tools/host_profile.py        (on the GPU box)"""
import cProfile
import os
import pstats
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from transhuman_amd import synth  # noqa: E402
from transhuman_amd.config import get_cfg  # noqa: E402
from transhuman_amd.networks.renderer.if_clight_renderer import Renderer  # noqa: E402

dev = torch.device("cuda:0")
cfg = get_cfg()
cfg.N_samples, cfg.num_class = 64, 500
b_cpu = synth.make_batch(512, 512, 3, seed=0, all_rays=True)
body = b_cpu["tar_smpl_vertice_smplcoord"][0].numpy()
net = bench.build_net(dev)
r = Renderer(net, vertex_can=body.astype(np.float64) * 1.02 + 0.001, pc2voxel_ind=bench.load_assign(500, body))
b = synth.batch_to(b_cpu, dev)
for _ in range(3):
    r.render_fast(b)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    r.render_fast(b)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
