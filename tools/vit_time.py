#!/usr/bin/env python
"""ViT (TransHE) forward time on the GPU box: python tools/vit_time.py [N_c ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from transhuman_amd import hip
dev = torch.device("cuda:0")
net = bench.build_net(dev)
for nc in [int(a) for a in sys.argv[1:]] or [500, 1500]:
    g = torch.randn(3, nc, 192, device=dev)
    pe = torch.rand(3, nc, 3, device=dev) * 2 - 1
    for _ in range(3): net.ViT(g, pe, mask=None)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter(); net.ViT(g, pe, mask=None); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("N_c", nc, "vit ms", round(float(np.median(ts)) * 1e3, 3))
