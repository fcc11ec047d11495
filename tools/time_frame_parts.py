import sys, time, torch, numpy as np
sys.path.insert(0,'/root/repo')
import bench
from transhuman_amd import synth, hip
from transhuman_amd.config import get_cfg
from transhuman_amd.networks.renderer.if_clight_renderer import Renderer
cfg=get_cfg(); cfg.N_samples=64; cfg.num_class=500
dev=torch.device('cuda:0')
b=synth.make_batch(512,512,3,seed=0); body=b["tar_smpl_vertice_smplcoord"][0].numpy()
a=bench.load_assign(500, body); net=bench.build_net(dev)
r=Renderer(net, vertex_can=body.astype(np.float64)*1.02+0.001, pc2voxel_ind=a)
bd=synth.batch_to(b,dev)
for _ in range(3): f=r.prepare_frame(bd)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(10): f=r.prepare_frame(bd)
torch.cuda.synchronize(); print('prepare_frame ms', (time.perf_counter()-t)*100)
enc=net.encoder; imgs=bd["input_imgs"][0][0]
def tm(fn,n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
print('trunk ms', tm(lambda: enc.trunk(imgs)))
lat=enc.trunk(imgs)
print('upsample ms', tm(lambda: hip.upsample_concat_nhwc(imgs,lat[0],lat[1],lat[2],enc.upsample_color.weight,enc.upsample_color.bias)))
for _ in range(3): o=r.render_fast(bd, frame=f)
print('render_rays ms', tm(lambda: r.render_fast(bd, frame=f)))
