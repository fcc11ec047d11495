#!/usr/bin/env python
"""TransHE forward as eager launches vs one replayed hipGraph (torch.cuda.CUDAGraph capture of the same C-ABI launches):
what the ~60 dependent launches cost as launches.   python tools/vit_graph_time.py [N_c ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from transhuman_amd import hip
dev = torch.device("cuda:0")
net = bench.build_net(dev)


def timed(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        e0.record(); f(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


for nc in [int(a) for a in sys.argv[1:]] or [500, 1500]:
    g = torch.randn(3, nc, 192, device=dev)
    pe = torch.rand(3, nc, 3, device=dev) * 2 - 1
    eager = timed(lambda: net.ViT(g, pe, mask=None))
    out_ref = net.ViT(g, pe, mask=None).clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): net.ViT(g, pe, mask=None)
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(gr):
            out = net.ViT(g, pe, mask=None)
        rep = timed(gr.replay)
        gr.replay(); torch.cuda.synchronize()
        print("N_c", nc, "eager ms", round(eager, 3), "graph replay ms", round(rep, 3), "max diff", float((out - out_ref).abs().max()))
    except Exception as e:
        print("N_c", nc, "eager ms", round(eager, 3), "capture failed:", repr(e)[:300])
