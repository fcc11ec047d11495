#!/usr/bin/env python
"""Which producer is not reproducible?  Render the headline frame N times with render_fast (K5 beside K4 beside TransHE),
snapshot the shading pool after every frame and compare it byte for byte with the first frame's: region A of the pool is
[pixel rows | neighbour records + tile headers | positional encodings | raw_c].
    python tools/pool_diff.py [N]        (on the GPU box)"""
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

N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0")
cfg = get_cfg()
cfg.N_samples, cfg.num_class = 64, 500
b_cpu = synth.make_batch(512, 512, 3, seed=0, all_rays=True)
body = b_cpu["tar_smpl_vertice_smplcoord"][0].numpy()
net = bench.build_net(dev)
r = Renderer(net, vertex_can=body.astype(np.float64) * 1.02 + 0.001, pc2voxel_ind=bench.load_assign(500, body))
b = synth.batch_to(b_cpu, dev)
al = lambda x: (x + 255) // 256 * 256
ref = None
for i in range(N):
    out = r.render_fast(b)
    torch.cuda.synchronize()
    n = int(r.last_stats["valid_samples"])
    pool = hip._pool_cache[str(dev)]
    a_f = al(n * 3 * 272 * 4)
    a_h = al((n + 32) * 64 + (n // 32 + 2) * 512)
    a_pe = al(n * 256)
    used = a_f + a_h + a_pe + al(n * 16)
    snap = pool[:used].clone()
    img = torch.cat([out["rgb_map"][0], out["acc_map"][0][:, None]], 1).clone()
    if ref is None:
        ref, ref_img = snap, img
        print("valid samples", n, "pool bytes used", used)
        continue
    parts = []
    CH = 1 << 28
    for o in range(0, used, CH):
        w = torch.nonzero(snap[o:o + CH] != ref[o:o + CH]).reshape(-1)
        if w.numel():
            parts.append(w + o)
    d = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.int64)
    print(f"frame {i}: image rays that differ {int((img != ref_img).any(1).sum())}; pool bytes that differ {d.numel()}")
    if d.numel():
        d = d.cpu().numpy()
        for name, lo, hi, unit in (("pixel rows (K5)", 0, a_f, 3 * 1088), ("records/headers (K4)", a_f, a_f + a_h, 64),
                                   ("positional enc (K4)", a_f + a_h, a_f + a_h + a_pe, 256), ("raw_c (K6)", a_f + a_h + a_pe, used, 16)):
            m = d[(d >= lo) & (d < hi)] - lo
            if m.size:
                items = np.unique(m // unit)
                print(f"   {name}: {m.size} bytes in {items.size} samples, first samples {items[:12].tolist()}, "
                      f"pos in 32-batch {[int(x) % 32 for x in items[:12]]}, byte offsets inside the first: "
                      f"{(m[m // unit == items[0]] % unit)[:8].tolist()} .. {(m[m // unit == items[0]] % unit)[-3:].tolist()}")
                if unit == 3 * 1088:
                    for it in items[:3]:
                        o = lo + int(it) * unit
                        a = snap[o:o + unit].view(torch.float16).float().cpu().numpy()
                        bq = ref[o:o + unit].view(torch.float16).float().cpu().numpy()
                        w = np.nonzero(a != bq)[0]
                        print("      sample", int(it), "halves", w[:6].tolist(), "now", a[w[:6]].tolist(), "ref", bq[w[:6]].tolist())
