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
import os
os.environ["TH_STEM_GRAPH"] = "0"; os.environ["TH_VIT_GRAPH"] = "0"
for i in range(3):
    torch.cuda.synchronize(); t = time.perf_counter(); r.render_fast(bd); torch.cuda.synchronize()
    print("eager frame", i, (time.perf_counter() - t) * 1e3, "ms")
os.environ["TH_STEM_GRAPH"] = "1"; os.environ["TH_VIT_GRAPH"] = "1"
for i in range(3):
    torch.cuda.synchronize(); t = time.perf_counter(); r.render_fast(bd); torch.cuda.synchronize()
    print("graph frame", i, (time.perf_counter() - t) * 1e3, "ms")
# which frame captures what: count capture_begin calls per frame on a fresh renderer
import collections
net2 = bench.build_net(dev)
r2 = Renderer(net2, vertex_can=body.astype(np.float64) * 1.02 + 0.001, pc2voxel_ind=a)
cnt = collections.Counter()
real = torch.cuda.CUDAGraph.capture_begin
cur = [0]
def cb(self, *a, **k):
    import traceback
    who = [f.name for f in traceback.extract_stack() if f.name in ("_trunk_graphed", "_vit_forward_graphed")]
    cnt[(cur[0], who[-1] if who else "?")] += 1
    return real(self, *a, **k)
torch.cuda.CUDAGraph.capture_begin = cb
for i in range(4):
    cur[0] = i
    torch.cuda.synchronize(); t = time.perf_counter(); r2.render_fast(bd); torch.cuda.synchronize()
    print("fresh renderer frame", i, (time.perf_counter() - t) * 1e3, "ms")
print("captures per frame:", dict(cnt))
