#!/usr/bin/env python
"""Host timeline of one Renderer.render_fast call on the headline frame: when (us after the call's start) the host enters and
leaves each of the calls that queue the frame's stages -- next to tools/dropin_loop.py's device timeline this shows which
stage starts late because the host got there late."""
import sys, time, torch, numpy as np, collections
sys.path.insert(0, '/root/repo')
import bench
from transhuman_amd import synth, hip
from transhuman_amd.config import get_cfg
from transhuman_amd.networks.renderer.if_clight_renderer import Renderer
cfg = get_cfg(); cfg.N_samples = 64; cfg.num_class = 500
dev = torch.device('cuda:0')
b = synth.make_batch(512, 512, 3, seed=0); body = b["tar_smpl_vertice_smplcoord"][0].numpy()
a = bench.load_assign(500, body); net = bench.build_net(dev)
r = Renderer(net, vertex_can=body.astype(np.float64) * 1.02 + 0.001, pc2voxel_ind=a)
bd = synth.batch_to(b, dev)
T0 = [0.0]
log = collections.defaultdict(list)


def wrap(obj, name, tag=None):
    fn = getattr(obj, name)
    tag = tag or name

    def w(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            log[tag].append(((t - T0[0]) * 1e6, (time.perf_counter() - T0[0]) * 1e6))
    setattr(obj, name, w)


for n in ("render_prepass", "upsample_concat_split", "map_fold", "map_box", "paint_group_nhwc", "segment_mean", "segment_mean_rot",
          "render_pregather", "render_rays", "vit_forward", "pack_cams", "feat_scale", "Points", "Frame", "render_predemand"):
    if hasattr(hip, n):
        wrap(hip, n)
wrap(net.encoder, "trunk")
wrap(r, "prepare_frame")
with torch.no_grad():
    for _ in range(4):
        r.render_fast(bd)
    torch.cuda.synchronize()
    log.clear()
    N = 10
    tot = 0.0
    for _ in range(N):
        T0[0] = time.perf_counter()
        r.render_fast(bd)
        tot += time.perf_counter() - T0[0]
        torch.cuda.synchronize()
print("render_fast host time per call (us):", tot / N * 1e6)
rows = sorted(((np.mean([x[0] for x in v]), np.mean([x[1] for x in v]), k, len(v) // N) for k, v in log.items()))
for s, e, k, c in rows:
    print(f"{s:9.1f} -> {e:9.1f}  (+{e - s:7.1f})  x{c}  {k}")
