import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from transhuman_amd import synth, hip
from transhuman_amd.config import get_cfg
from transhuman_amd.networks.renderer.if_clight_renderer import Renderer
dev = torch.device("cuda:0")
cfg = get_cfg(); cfg.N_samples, cfg.num_class = 64, 500
b_cpu = synth.make_batch(512, 512, 3, seed=0, all_rays=True)
body = b_cpu["tar_smpl_vertice_smplcoord"][0].numpy()
net = bench.build_net(dev)
r = Renderer(net, vertex_can=body.astype(np.float64) * 1.02 + 0.001, pc2voxel_ind=bench.load_assign(500, body))
b = synth.batch_to(b_cpu, dev)
for _ in range(3): r.prepare_frame(b)
torch.cuda.synchronize()
hs, gs = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); f = r.prepare_frame(b); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    hs.append(t1 - t0); gs.append(t2 - t0)
print("prepare_frame host queue ms", np.median(hs) * 1e3, "to completion ms", np.median(gs) * 1e3)
# pieces
enc = net.encoder
imgs = b["input_imgs"][0].reshape(-1, *b["input_imgs"][0].shape[2:])
def tm(fn, n=10):
    fn(); torch.cuda.synchronize(); h=[]; g=[]
    for _ in range(n):
        torch.cuda.synchronize(); t0=time.perf_counter(); fn(); t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
        h.append(t1-t0); g.append(t2-t0)
    return round(np.median(h)*1e3,3), round(np.median(g)*1e3,3)
print("trunk", tm(lambda: enc.trunk(imgs)))
grouped = r.last_grouped
pe = r._pe_norm(3, dev)
print("vit", tm(lambda: net.ViT(grouped, pe, mask=None)))
