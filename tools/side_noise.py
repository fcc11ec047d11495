"""What does a kernel boundary on ANOTHER stream cost the shading of a frame?  render_rays of one fixed frame in a loop on the
current stream; on a side stream N spin kernels per frame (torch.cuda._sleep: one thread each) whose total length is constant.
    python tools/side_noise.py
(the side kernels are issued AFTER render_rays has queued the frame, so the host's launch calls are not in the measurement)
"""
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
frame = r.prepare_frame(b)
pts = hip.Points(b["ray_o"][0], b["ray_d"][0], b["near"][0], b["far"][0], n_samples=64)
side = torch.cuda.Stream(dev)
tiny = torch.zeros(64, device=dev)
TOTAL = int(float(os.environ.get("NOISE_MS", "12")) * 2.0e6)      # spin cycles per frame (~2 GHz)

def run(n, kind, frames=6):
    ts = []
    for it in range(frames + 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        side.wait_stream(torch.cuda.current_stream())
        hip.render_rays(net, frame, pts)          # returns with the whole shading queued (one host wait for the sample count)
        with torch.cuda.stream(side):
            for _ in range(n):
                if kind == "sleep":
                    torch.cuda._sleep(TOTAL // max(n, 1))
                else:
                    tiny.add_(1.0)
        torch.cuda.current_stream().synchronize()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        if it >= 2:
            ts.append((t1 - t0) * 1e3)
    return float(np.median(ts))

def run_graph(n, frames=6):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        torch.cuda._sleep(1000)
        with torch.cuda.graph(g, stream=side):
            for _ in range(n):
                torch.cuda._sleep(TOTAL // max(n, 1))
    torch.cuda.synchronize()
    ts = []
    for it in range(frames + 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        side.wait_stream(torch.cuda.current_stream())
        hip.render_rays(net, frame, pts)
        with torch.cuda.stream(side):
            g.replay()
        torch.cuda.current_stream().synchronize()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        if it >= 2:
            ts.append((t1 - t0) * 1e3)
    return float(np.median(ts))

print("alone", round(run(0, "sleep"), 3))
for n in (64, 256, 1024):
    print("graph of sleep kernels per frame", n, "shading ms", round(run_graph(n), 3))
for n in (1, 16, 64, 256, 1024):
    print("sleep kernels per frame", n, "shading ms", round(run(n, "sleep"), 3))
for n in (64, 256, 1024):
    print("tiny add kernels per frame", n, "shading ms", round(run(n, "add"), 3))
print("alone", round(run(0, "sleep"), 3))
