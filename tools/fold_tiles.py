import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
import bench
from transhuman_amd import synth, hip
from transhuman_amd.config import get_cfg
from transhuman_amd.networks.renderer.if_clight_renderer import Renderer
dev = torch.device("cuda:0")
cfg = get_cfg(); cfg.N_samples, cfg.num_class = 64, 500
net = bench.build_net(dev)
bc = synth.make_batch(512, 512, 3, seed=0, all_rays=True)
body = bc["tar_smpl_vertice_smplcoord"][0].numpy()
r = Renderer(net, vertex_can=body.astype(np.float64) * 1.02 + 0.001, pc2voxel_ind=bench.load_assign(500, body))
b = synth.batch_to(bc, dev)
f = r.prepare_frame(b)
sp = hip.map_spans(f.map.box, 512).cpu().numpy()      # [V,H,2]
w = np.maximum(sp[..., 1] - sp[..., 0] + 1, 0)
print("texels", int(w.sum()), "rows", int((w > 0).sum()), "mean width", float(w[w > 0].mean()))
print("old tiles (96 from span start)", int(np.ceil(w / 96).sum()))
for RT, BW in ((3, 32), (2, 48), (1, 96), (6, 16), (3, 64)):
    n = 0
    for v in range(sp.shape[0]):
        for y0 in range(0, 512, RT):
            rows = [(sp[v, y, 0], sp[v, y, 1]) for y in range(y0, min(y0 + RT, 512)) if sp[v, y, 1] >= sp[v, y, 0]]
            if not rows: continue
            a = min(x for x, _ in rows); bb = max(x for _, x in rows)
            n += bb // BW - a // BW + 1
    print("tiles", RT, "rows x", BW, "cols:", n, "slots", n * RT * BW)
