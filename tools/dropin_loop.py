#!/usr/bin/env python
"""N calls of Renderer.render_fast on the headline frame (the reference's own call pattern, run.py:96-118), each followed by the
host wait a caller's `.cpu()` is -- for a rocprofv3 kernel trace of the single-frame chain (tools/trace_timeline.py)."""
import sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo')
import bench
from transhuman_amd import synth
from transhuman_amd.config import get_cfg
from transhuman_amd.networks.renderer.if_clight_renderer import Renderer
cfg = get_cfg(); cfg.N_samples = 64; cfg.num_class = 500
dev = torch.device('cuda:0')
b = synth.make_batch(512, 512, 3, seed=0); body = b["tar_smpl_vertice_smplcoord"][0].numpy()
a = bench.load_assign(500, body); net = bench.build_net(dev)
r = Renderer(net, vertex_can=body.astype(np.float64) * 1.02 + 0.001, pc2voxel_ind=a)
bd = synth.batch_to(b, dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
import os, contextlib
ctx = torch.cuda.stream(torch.cuda.Stream()) if os.environ.get('STREAM') == '1' else contextlib.nullcontext()
with torch.no_grad(), ctx:
    for _ in range(3):
        r.render_fast(bd)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        o = r.render_fast(bd)
        torch.cuda.synchronize()
    print('render_fast + sync ms', (time.perf_counter() - t) / n * 1e3)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        o = r.render_fast(bd)
    torch.cuda.synchronize()
    print('render_fast back to back ms', (time.perf_counter() - t) / n * 1e3)
