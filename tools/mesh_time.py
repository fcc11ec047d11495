#!/usr/bin/env python
"""Where the 256^3 mesh workload (if_mesh_renderer.Renderer.render) spends its time on the GPU box: whole render, sigma
grid alone, frame constants, marching cubes:  python tools/mesh_time.py"""
import sys, time, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench
from transhuman_amd import hip, synth
from transhuman_amd.config import get_cfg
from transhuman_amd.networks.renderer.if_mesh_renderer import Renderer as MeshRenderer
dev = torch.device("cuda:0")
net = bench.build_net(dev)
cfg = get_cfg(); cfg.N_samples = 64; cfg.num_class = 500; cfg.mesh_th = 0.5
bc = synth.make_batch(512, 512, 3, seed=0, all_rays=True)
body = bc["tar_smpl_vertice_smplcoord"][0].numpy()
assign = bench.load_assign(500, body)
mr = MeshRenderer(net, vertex_can=body.astype(np.float64) * 1.02 + 0.001, pc2voxel_ind=assign)
mb = dict(bc); mb["pts"] = synth.make_grid_pts(bc, 256); mbd = synth.batch_to(mb, dev)
def T(f, n=4, w=2):
    for _ in range(w): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): o = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, o
ms, out = T(lambda: mr.render(mbd)); print("render ms", ms)
flat = mbd["pts"].reshape(-1, 3)
fr = mr.prepare_frame(mbd)
ms2, sig = T(lambda: hip.eval_sigma_grid(net, fr, flat)); print("sigma only ms", ms2)
ms3, _ = T(lambda: mr.prepare_frame(mbd)); print("prepare_frame ms", ms3)
cube = out["cube"]
ms4, _ = T(lambda: hip.marching_cubes(torch.as_tensor(cube, device=dev) if not torch.is_tensor(cube) else cube.to(dev), 0.5)); print("mc ms", ms4, type(cube))
