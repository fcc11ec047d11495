#!/usr/bin/env python
"""The host's own cost per frame of the timed loop of bench.py (Renderer.render_sequence), for the whole frame or for one
rank's shard of an N-rank job: wall time of the queueing loop minus the time spent in blocking waits (th_host_wait_read),
then the same loop under cProfile (inflated ~2x, but it shows where the Python / ctypes time goes).

    python tools/host_cost.py [--world 8] [--rank 3] [--steps 40]        (on the GPU box)
"""
import argparse
import cProfile
import itertools
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from transhuman_amd import synth, hip  # noqa: E402
from transhuman_amd.config import get_cfg  # noqa: E402
from transhuman_amd.dist import shard_ray_indices, TokenExchange  # noqa: E402
from transhuman_amd.networks.renderer.if_clight_renderer import Renderer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=1)
ap.add_argument("--rank", type=int, default=0)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--top", type=int, default=32)
args = ap.parse_args()

dev = torch.device("cuda:0")
cfg = get_cfg()
cfg.N_samples, cfg.num_class = 64, 500
b_cpu = synth.make_batch(512, 512, 3, seed=0, all_rays=True)
body = b_cpu["tar_smpl_vertice_smplcoord"][0].numpy()
net = bench.build_net(dev)
r = Renderer(net, vertex_can=body.astype(np.float64) * 1.02 + 0.001, pc2voxel_ind=bench.load_assign(500, body))
b = synth.batch_to(b_cpu, dev)
idx = shard_ray_indices(512, 512, args.world, args.rank, tile=8, tile_major=True).to(dev)
shard = dict(b)
for k in ("ray_o", "ray_d", "near", "far"):
    shard[k] = b[k][:, idx].contiguous()
tx = TokenExchange(emulate=(args.world, args.rank)) if args.world > 1 else None
seq = r.render_sequence(itertools.repeat(shard), small_frame_rays=-1 if args.world > 1 else 2400, token_exchange=tx)
for _ in range(5):
    next(seq)
torch.cuda.synchronize()
hip.host_wait_read(dev)
t0 = time.perf_counter()
for _ in range(args.steps):
    next(seq)
host = time.perf_counter() - t0
wait = hip.host_wait_read(dev)
torch.cuda.synchronize()
total = time.perf_counter() - t0
print(f"world {args.world} rank {args.rank}: frame {total / args.steps * 1e3:.3f} ms, host loop {host / args.steps * 1e3:.3f} ms of "
      f"which blocked {wait / args.steps:.3f} ms -> pure host {(host * 1e3 - wait) / args.steps:.3f} ms/frame")
pr = cProfile.Profile()
pr.enable()
for _ in range(args.steps):
    next(seq)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
print(f"--- cProfile over {args.steps} frames (times inflated by the profiler), by internal time ---")
st.sort_stats("tottime").print_stats(args.top)
print("--- by cumulative time ---")
st.sort_stats("cumulative").print_stats(args.top)
