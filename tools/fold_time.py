import sys, time, torch, numpy as np
sys.path.insert(0, "/root/repo")
import bench
from transhuman_amd import synth, hip
from transhuman_amd.config import get_cfg
from transhuman_amd.networks.renderer.if_clight_renderer import Renderer
dev = torch.device("cuda:0")
cfg = get_cfg(); cfg.N_samples, cfg.num_class = 64, 500
net = bench.build_net(dev)
WIDTH = int(sys.argv[1]) if len(sys.argv) > 1 else 512      # image width: the map's row pitch is WIDTH KiB
bc = synth.make_batch(512, WIDTH, 3, seed=0, all_rays=True)
body = bc["tar_smpl_vertice_smplcoord"][0].numpy()
r = Renderer(net, vertex_can=body.astype(np.float64) * 1.02 + 0.001, pc2voxel_ind=bench.load_assign(500, body))
b = synth.batch_to(bc, dev)
f = r.prepare_frame(b)
m = f.map
print("box", m.box.cpu().numpy().tolist())
def tm(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("map_fold ms", tm(lambda: hip.map_fold(net, m)))
