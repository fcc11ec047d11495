#!/usr/bin/env python
"""What bounds K5 (pixgather_s256_kernel; TH_K5_GENERIC=1: pixgather_kernel<true>)?  The same launch (the headline frame's 2.09 M valid samples x 3 views, split
rows out) on the frame's own sample list and on lists with the SAME instruction stream but other footprints:
  rowmajor / tile8 / morton / ...: the depth-major list of the frame with its hit rays in that image order (bench.py: tile8)
  shuffled the same samples in random order                      (no locality between the rows of a wave)
  window   the first 4096 samples of the list, repeated          (footprint: a few hundred texel rows, L2-resident)
  one      one sample, repeated                                  (every corner load hits L1)
If `one` / `window` are not much faster than `frame`, the kernel is bound by what it issues (instructions through the
texture path, stores), not by where the texels come from.
    python tools/k5_locality.py            (on the GPU box)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from transhuman_amd import synth, hip  # noqa: E402
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
frame = r.prepare_frame(b)
pts = hip.Points(b["ray_o"][0], b["ray_d"][0], b["near"][0], b["far"][0], n_samples=64)
mask, _ = hip.hull_mask(pts, b["tar_smpl_vertice"][0])
def morton(x, y):
    k = torch.zeros_like(x)
    for i in range(10):
        k |= ((x >> i) & 1) << (2 * i) | ((y >> i) & 1) << (2 * i + 1)
    return k


def sample_list(ray_key):
    """the frame's sample list (depth-major inside groups of 16 consecutive hit rays) with the rays ordered by ray_key"""
    hit = torch.nonzero(mask.any(1)).reshape(-1)
    hit = hit[torch.argsort(ray_key[hit], stable=True)]
    m = mask[hit]
    cr, ss = torch.nonzero(m, as_tuple=True)                         # cr: index into the ordered hit-ray list
    order = torch.argsort((cr // 16) * (64 * 16) + ss * 16 + (cr % 16))
    rr, ss = hit[cr[order]], ss[order]
    z = pts.near[rr] * pts.omt[ss] + pts.far[rr] * pts.t[ss]
    return (pts.ray_o[rr] + pts.ray_d[rr] * z[:, None]).contiguous()


ray = torch.arange(512 * 512, device=dev)
px, py = ray % 512, ray // 512
world = sample_list(ray)
P = world.shape[0]
print(f"{P} valid samples")
lib = hip.load_library()
V, H, W = frame.map.V, frame.map.H, frame.map.W
out = torch.zeros((P, V, 544), dtype=torch.float16, device=dev)
g = torch.Generator(device=dev).manual_seed(0)
cases = {
    "rowmajor": world,
    "tile8": sample_list(((py // 8) * 64 + px // 8) * 64 + (py % 8) * 8 + px % 8),
    "morton": sample_list(morton(px, py)),
    "tile8-mort": sample_list(morton(px // 8, py // 8) * 64 + (py % 8) * 8 + px % 8),
    "tile4x4": sample_list(((py // 4) * 128 + px // 4) * 16 + (py % 4) * 4 + px % 4),
    "shuffled": world[torch.randperm(P, device=dev, generator=g)].contiguous(),
    "window": world[:4096].repeat((P + 4095) // 4096, 1)[:P].contiguous(),
    "one": world[P // 2:P // 2 + 1].repeat(P, 1).contiguous(),
}

if len(sys.argv) > 1:                      # e.g. tile8,window,one (PMC runs: tools/k5_pmc.sh)
    cases = {k: cases[k] for k in sys.argv[1].split(",")}
for name, w in cases.items():
    ts = []
    for it in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hip._check(lib.th_pixel_gather_split(hip.ctx(dev), hip._p(frame.map), V, H, W, hip._p(w), None, P, hip._p(frame.cams),
                                             hip._p(frame.scale), hip._p(out), 272, hip._stream()))
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = float(np.median(ts[1:]))
    rows = P * V
    print(f"{name:9s} {t:6.3f} ms   {rows / t / 1e3:7.1f} M rows/s   loads {rows * 4 * 1040 / t / 1e9:6.2f} TB/s of corner bytes, "
          f"stores {rows * 1088 / t / 1e9:5.2f} TB/s")
